"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; argument validation that needs no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", h)).read()
            names += re.findall(r"ENVGS_API\s+[\w\s\*]+?\b(envgs_\w+)\s*\(", txt)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from envgs_amd import build, _lib
    build.build_library()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from envgs_amd import _lib
    decl = _declared_symbols()
    assert len(decl) >= 12
    for name in decl:
        assert hasattr(lib, name), "libenvgs_hip.so does not export %s" % name
        assert name in _lib.SYMBOLS, "envgs_amd/_lib.py has no prototype for %s" % name
    # and nothing is bound that the headers do not declare
    assert sorted(_lib.SYMBOLS) == decl


def test_struct_layouts_match_headers():
    from envgs_amd import _lib
    assert ctypes.sizeof(_lib.RasterCfg) == 8 * 4 + 3 * 4 + 4
    assert ctypes.sizeof(_lib.TraceCfg) == 11 * 4 + 2 * 4 + 4
    hdr = open(os.path.join(ROOT, "include", "envgs_raster.h")).read()
    fields = re.findall(r"^\s+(?:int32_t|float)\s+([\w, ]+);", hdr[hdr.index("typedef struct envgs_raster_cfg"):hdr.index("} envgs_raster_cfg")], re.M)
    flat = [f.strip() for grp in fields for f in grp.split(",")]
    assert flat == [n for n, _ in _lib.RasterCfg._fields_]
    hdr = open(os.path.join(ROOT, "include", "envgs_trace.h")).read()
    fields = re.findall(r"^\s+(?:int32_t|float)\s+([\w, ]+);", hdr[hdr.index("typedef struct envgs_trace_cfg"):hdr.index("} envgs_trace_cfg")], re.M)
    flat = [f.strip() for grp in fields for f in grp.split(",")]
    assert flat == [n for n, _ in _lib.TraceCfg._fields_]
    body = hdr[hdr.index("typedef struct envgs_trace_lists"):hdr.index("} envgs_trace_lists")]
    names = re.findall(r"^\s+(?:u?int\d+_t|size_t|void|float)\s+\*?\s*(\w+);", body, re.M)
    assert names == [n for n, _ in _lib.TraceLists._fields_]


def test_bad_arguments_are_rejected_before_any_gpu_work(lib):
    from envgs_amd import _lib
    n = ctypes.c_uint32(7)
    null = [None] * 15
    for bad in (dict(channels=4), dict(sh_degree=5), dict(width=0)):
        kw = dict(P=10, sh_degree=0, sh_coeffs=0, channels=3, width=64, height=64, bg_len=3, debug=0, scale_modifier=1.0, tanfovx=1.0, tanfovy=1.0, feature_f16=0)
        kw.update(bad)
        cfg = _lib.RasterCfg(*[kw[k] for k, _ in _lib.RasterCfg._fields_])
        rc = lib.envgs_raster_project(cfg, *null, None, 0, n, None)
        assert rc == -1
    tcfg = _lib.TraceCfg(10, 10, 9, 0, 0, 1, 0, 3, 0, 0, 0, 1.0, 0.0, 0)      # sh_degree 9
    assert lib.envgs_trace_forward(tcfg, *([None] * 22), None, None) == -1
    assert lib.envgs_bvh_build(-1, None, None, None, None, 0, 0, None) == -1
    assert lib.envgs_prof_kernel_name(6) == b"composite_bwd" and lib.envgs_prof_kernel_name(999) == b""


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    import diff_surfel_rasterization_wet as pkg
    import diff_surfel_tracing as tpkg
    from envgs_amd import _lib
    st = pkg.GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                                           viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                           prefiltered=False, debug=False)
    R = pkg.GaussianRasterizer(raster_settings=st)
    a = dict(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        R(**a, shs=None, colors_precomp=None, cov3D_precomp=None)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        R(**a, shs=torch.zeros(4, 16, 3), colors_precomp=None, cov3D_precomp=torch.zeros(4, 9))
    with pytest.raises(RuntimeError, match="no CPU path"):          # CPU tensors: refuse, never fall back
        R(**a, shs=torch.zeros(4, 16, 3), colors_precomp=None, cov3D_precomp=None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        tpkg.SurfelTracer().build_acceleration_structure(torch.zeros(16, 3), torch.zeros(8, 3, dtype=torch.int32))
    # missing library -> loud failure at first use
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libenvgs_hip.so")
    with pytest.raises(RuntimeError, match="is missing"):
        _lib.load()


def test_product_library_is_trimmed_of_the_diagnostic_kernels():
    """the superseded A/B kernels (three earlier collection kernels, the per-ray atomic-flush list backward) are compiled only with
    -DENVGS_DIAG into libenvgs_hip_diag.so, which tests and `bench.py --diag` select explicitly; the product library does not contain them
    (and rejects the debug switches that would ask for them).  Both builds export the whole C-ABI."""
    from envgs_amd import _lib, build
    prod = open(build.LIB, "rb").read(); diag = open(build.LIB_DIAG, "rb").read()
    for name in (b"collect_hits_packet4", b"collect_hits_packet", b"composite_lists_bwd"):
        assert name not in prod and name in diag, name
    for name in (b"collect_hits_coop", b"sort_composite_fwd", b"batch_surfel_bwd", b"composite_bwd"):
        assert name in prod and name in diag, name
    assert b"collect_hits_coopILb1E" not in prod and b"collect_hits_coopILb1E" in diag        # round 5's deferred-exact-test form: measured slower, A/B only
    old = _lib.select("diag")
    try:
        lib = _lib.load()
        for sym in _lib.SYMBOLS:
            assert hasattr(lib, sym)
    finally:
        _lib.select(old)
    assert _lib.load() is not lib or old == "diag"


def test_importing_the_tracing_package_changes_no_process_state(monkeypatch):
    """the rocBLAS selection for torch's own matmuls is OPT-IN (ENVGS_PREFER_ROCBLAS=1 / envgs_amd.prefer_rocblas()), never a
    side effect of `import diff_surfel_tracing`."""
    import importlib
    import envgs_amd
    calls = []
    monkeypatch.setattr(envgs_amd, "prefer_rocblas", lambda: calls.append(1))
    monkeypatch.delenv("ENVGS_PREFER_ROCBLAS", raising=False)
    import diff_surfel_tracing
    importlib.reload(diff_surfel_tracing)
    assert calls == []
    monkeypatch.setenv("ENVGS_PREFER_ROCBLAS", "1")
    importlib.reload(diff_surfel_tracing)
    assert calls == [1]


def test_step_glue_leaves_the_public_feature_storage_choice_alone():
    """envgs_step's passes only override the process-wide feature storage when bench / tests pinned it (tri-state FEATURE_F16)."""
    import envgs_amd
    from envgs_amd import envgs_step, raster
    try:
        envgs_amd.set_feature_storage("f16")
        assert envgs_step.FEATURE_F16["on"] is None
        envgs_step._select_storage()
        assert raster.FEATURE_STORAGE["f16"] is True
        envgs_step.FEATURE_F16["on"] = False
        envgs_step._select_storage()
        assert raster.FEATURE_STORAGE["f16"] is False
    finally:
        envgs_step.FEATURE_F16["on"] = None
        envgs_amd.set_feature_storage("f32")
