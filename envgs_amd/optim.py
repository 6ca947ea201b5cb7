"""Sparse fused Adam over the per-Gaussian tensors (include/envgs_optim.h).

`FusedAdam` is the MI355X counterpart of the reference's `MyFusedAdam` (easyvolcap/runners/optimizers.py:17-75 on top of
easyvolcap/utils/src/fused_adam.cu): plain Adam, no weight decay, no amsgrad, elements whose gradient is exactly zero are skipped
(their moments and the parameter stay untouched).  One launch per step for up to 24 tensors instead of one launch per tensor.
State layout is torch.optim.Adam's (`step`, `exp_avg`, `exp_avg_sq`), so checkpoints interchange and the densification code that edits
optimizer state in place (gaussian2d_utils.py:526-621) keeps working.
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        from .tracing import join_deferred_gradients
        join_deferred_gradients()                  # (SurfelTracer.set_deferred_surfel_gradients: the surfel gradients may still be in flight on the library's stream)
        batches = {}
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise RuntimeError("FusedAdam mirrors MyFusedAdam: no weight decay / amsgrad / maximize")
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters on the GPU; there is no CPU path")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (p.device, float(beta1), float(beta2), float(group["eps"]))
                batches.setdefault(key, []).append((p, g, st, float(group["lr"])))
        for (dev, beta1, beta2, eps), items in batches.items():
            stream = _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for i in range(0, len(items), 24):
                chunk = items[i:i + 24]
                arr = (_lib.AdamTensor * len(chunk))()
                for j, (p, g, st, lr) in enumerate(chunk):
                    arr[j] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), lr,
                                             float(st["step"]))
                _lib.check(lib.envgs_fused_adam(len(chunk), arr, beta1, beta2, eps, stream), "envgs_fused_adam")
        return loss
