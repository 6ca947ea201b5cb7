"""CPU: the product path never imports, links or executes anything under oracle/ (the oracle is the checker only)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = ["envgs_amd", "diff_surfel_rasterization_wet", "diff_surfel_rasterization_wet_ch05", "diff_surfel_rasterization_wet_ch07",
           "diff_surfel_tracing", "include"]


def test_product_sources_do_not_reference_the_oracle():
    offenders = []
    for top in PRODUCT:
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            if "_build" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if not f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for m in re.finditer(r"^\s*(from\s+oracle|import\s+oracle|#include\s+\".*oracle)|liboracle|oracle/_ref", txt, re.M):
                    offenders.append((os.path.join(dp, f), m.group(0)))
    assert not offenders, offenders


def test_only_allowed_callers_import_the_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            assert "from oracle" not in open(os.path.join(ROOT, f)).read(), f


def test_library_does_not_link_the_oracle():
    so = os.path.join(ROOT, "envgs_amd", "libenvgs_hip.so")
    if os.path.exists(so):
        blob = open(so, "rb").read()
        assert b"liboracle" not in blob and b"orc_render_fwd" not in blob and b"trc_forward" not in blob
