"""Time the reference's loss expressions (restated with torch conv2d, as ssim_utils.py does) against the fused kernel."""
import sys, os, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import loss as eloss
from oracle.loss_oracle import WINDOW_F32
dev = torch.device("cuda", 0)
win = torch.tensor(WINDOW_F32, device=dev).reshape(1, 1, 1, 11).repeat(3, 1, 1, 1)
def blur(t):
    t = F.conv2d(t, win.transpose(2, 3), padding="same", groups=3)
    return F.conv2d(t, win, padding="same", groups=3)
def ref(x, y):
    X, Y = x[None], y[None]
    mu1, mu2 = blur(X), blur(Y)
    s1 = blur(X * X) - mu1 * mu1; s2 = blur(Y * Y) - mu2 * mu2; s12 = blur(X * Y) - mu1 * mu2
    C1, C2 = 1e-4, 9e-4
    ssim = (((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))).mean()
    return 0.8 * (x - y).abs().mean() + 0.2 * (1 - ssim)
x = torch.rand(3, 800, 800, device=dev, requires_grad=True); y = torch.rand(3, 800, 800, device=dev)
for fn, nm in ((ref, "torch (reference expressions)"), (eloss.l1_ssim_loss, "fused HIP")):
    for _ in range(3):
        x.grad = None; fn(x, y).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        x.grad = None; fn(x, y).backward()
    torch.cuda.synchronize()
    print("%-32s fwd+bwd %.3f ms  loss %.7f" % (nm, (time.perf_counter() - t0) / 30 * 1e3, float(fn(x, y))))
