"""Times the two tiny-K matmuls of the reference's caller code (gaussian2d_utils.py:1123, optix_utils.py:59) under the BLAS backends
torch offers on ROCm.  Run on the GPU box:  python scratch/blas_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
H = W = 800
n = torch.randn(3, H, W, device=dev)
R = torch.randn(3, 3, device=dev)
P = 163840
T = torch.randn(P, 4, 4, device=dev)
s3 = torch.randn(P, 4, 4, device=dev)


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


def normal_ref(): return (n.permute(1, 2, 0) @ R.T).permute(2, 0, 1)
def normal_elem(): return torch.stack([n[0] * R[c, 0] + n[1] * R[c, 1] + n[2] * R[c, 2] for c in range(3)], 0)
def disks_ref():
    Tt = T[:, None].expand(-1, 4, -1, -1)
    return (Tt.reshape(-1, 4, 4) @ s3.reshape(-1, 4, 1))


for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print(lib, "unavailable", e); continue
    print("%-9s normal matmul %.3f ms   elementwise %.3f ms   get_disks bmm %.3f ms" % (lib, t(normal_ref), t(normal_elem), t(disks_ref)))
for k in ("TORCH_BLAS_PREFER_HIPBLASLT",):
    import os; print(k, os.environ.get(k))
