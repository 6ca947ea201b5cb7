"""Fused image loss  w_l1 * mean|x - y| + w_ssim * (1 - ssim(x, y))  (include/envgs_loss.h).

`l1_ssim_loss(x, y)` with the default weights is the supervision EnvGS trains with (configs/models/envgs.yaml:70-72; L1 and SSIM branches of
easyvolcap/models/supervisors/volumetric_video_supervisor.py:40-66,112-144 -> loss_utils.l1 :319-333 and ssim_utils.ssim :107-167).
x, y: (C, H, W) -- any strides, any float dtype; the gradient flows to x only (y is the ground truth)."""
import torch

from . import _lib


def _stream(dev):
    return _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, w_l1, w_ssim):
        lib = _lib.load()
        if x.device.type != "cuda":
            raise RuntimeError("l1_ssim_loss needs GPU tensors; there is no CPU path")
        if x.shape != y.shape or x.dim() != 3:
            raise ValueError("l1_ssim_loss expects two (C, H, W) images of the same shape")
        C, H, W = x.shape
        if H < 11 or W < 11:
            raise ValueError("SSIM needs H, W >= 11 (the reference skips it below that)")
        dev = x.device
        xc = x.detach().to(torch.float32).contiguous(); yc = y.detach().to(torch.float32).contiguous()
        need = ctx.needs_input_grad[0]
        nb = lib.envgs_l1_ssim_partial_count(C, H, W)
        partial = torch.empty(nb, 2, dtype=torch.float32, device=dev)
        maps = torch.empty(3, C, H, W, dtype=torch.float32, device=dev) if need else None
        p = _lib.ptr
        _lib.check(lib.envgs_l1_ssim_forward(C, H, W, p(xc), p(yc), p(maps), p(partial), _stream(dev)), "envgs_l1_ssim_forward")
        sums = partial.double().sum(0) / float(C * H * W)              # (mean ssim, mean |x - y|): tile sums added in double
        loss = (w_l1 * sums[1] + w_ssim * (1.0 - sums[0])).to(torch.float32)
        ctx.saved = (xc, yc, maps, float(w_l1), float(w_ssim), x.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xc, yc, maps, w_l1, w_ssim, dt = ctx.saved
        C, H, W = xc.shape
        dx = torch.empty_like(xc)
        go = g.detach().to(torch.float32).reshape(1).contiguous()
        p = _lib.ptr
        _lib.check(lib.envgs_l1_ssim_backward(C, H, W, p(xc), p(yc), p(maps), p(go), w_l1, w_ssim, p(dx), _stream(xc.device)), "envgs_l1_ssim_backward")
        return dx.to(dt), None, None, None


def l1_ssim_loss(x, y, w_l1=0.8, w_ssim=0.2):
    return _L1SSIM.apply(x, y, w_l1, w_ssim)
