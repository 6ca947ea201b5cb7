"""CPU half of tests/test_reference_kernel_golden.py: the fixture-dump script for the REAL extensions must parse, document itself, and REFUSE to
pin this repository's own same-named drop-in packages (a fixture dumped from them would compare the HIP path with itself)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "golden", "dump_reference_kernel_golden.py")


def test_help_and_refusal(tmp_path):
    r = subprocess.run([sys.executable, SCRIPT, "--help"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "--allow-local" in r.stdout
    out = str(tmp_path / "x.pt")
    env = dict(os.environ); env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, SCRIPT, "--out", out], capture_output=True, text=True, cwd=str(tmp_path), timeout=300, env=env)
    assert r.returncode != 0 and not os.path.exists(out)
    assert "diff_surfel_rasterization_wet" in (r.stderr + r.stdout)           # the missing / refused package is named


def test_the_script_imports_nothing_of_the_reference():
    src = open(SCRIPT).read()
    for word in ("import easyvolcap", "from easyvolcap", "sys.path.insert(0, \"/root/reference"):
        assert word not in src
