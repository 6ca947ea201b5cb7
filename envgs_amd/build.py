"""Builds libenvgs_hip.so (all HIP kernels + the C-ABI of include/*.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built .so travels to
the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).  No torch headers are involved:
the library is plain HIP behind an extern "C" boundary and is loaded with ctypes (envgs_amd/_lib.py).
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libenvgs_hip.so")
# Diagnostic build: the same library plus the superseded A/B kernels (three earlier collection kernels, the per-ray atomic-flush list
# backward) behind envgs_debug_set -- compiled with -DENVGS_DIAG from the three sources that mention them; tests and `bench.py --diag` load it,
# the product library does not contain them.
LIB_DIAG = os.path.join(HERE, "libenvgs_hip_diag.so")
DIAG_SOURCES = ("trace_collect.hip", "trace_surfel_bwd.hip", "trace_api.hip", "raster_render.hip")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++20", "-fPIC", "-munsafe-fp-atomics", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function"]
# per-file extra flags; raster_project.hip feeds bit-exact integer keys -> no FMA contraction there; raster_project_bwd.hip (R8, HBM-bound,
# one lane per surfel) follows the oracle's operation order statement by statement, so without contraction its 3-term dot products against the
# pixel-scale projection matrix round exactly as the oracle's do
# tracer kernels: the SLP vectoriser pairs scalar fp32 ops into v_pk_* and then spends two v_mov per packed op assembling register pairs
# (batch_surfel_bwd: 89 v_mov per entry, 255 VGPRs; without it 17 and 221)
_NO_SLP = ["-fno-slp-vectorize"]
EXTRA = {"raster_project.hip": ["-ffp-contract=off"], "raster_project_bwd.hip": ["-ffp-contract=off"], "trace_kbuffer.hip": _NO_SLP, "trace_collect.hip": _NO_SLP, "trace_lists.hip": _NO_SLP,
         "trace_surfel_bwd.hip": _NO_SLP, "trace_api.hip": _NO_SLP}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hdr_m, diag=False):
    obj = os.path.join(OBJ, src.replace(".hip", ".diag.o" if diag else ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_m):
        return obj, False
    cmd = ["hipcc"] + COMMON + EXTRA.get(src, []) + (["-DENVGS_DIAG"] if diag else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
    return obj, True


def build_library(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _headers_mtime()
    jobs = [(s, False) for s in _sources()] + [(s, True) for s in DIAG_SOURCES]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda j: _compile(j[0], force, hdr_m, j[1]), jobs))
    by = {j: r for j, r in zip(jobs, res)}
    for lib, diag in ((LIB, False), (LIB_DIAG, True)):
        objs = [by[(s, diag and s in DIAG_SOURCES)][0] for s in _sources()]
        changed = any(by[(s, diag and s in DIAG_SOURCES)][1] for s in _sources())
        if force or changed or not os.path.exists(lib):
            cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
            if verbose:
                print("built", lib)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in os.sys.argv, verbose=True)
