"""The CPU oracle dressed up as the two drop-in packages (same constructors / call signatures), so the SAME caller code
(envgs_amd/envgs_step.py) can run once over the HIP extensions and once over the oracle.  Test infrastructure only."""
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from envgs_amd.raster import GaussianRasterizationSettings
from envgs_amd.tracing import SurfelTracingSettings
from oracle import raster as orc, trace as otr


def _np(t):
    return None if t is None else t.detach().cpu().float().numpy()


def make_raster_pkg(C):
    class _F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, st):
            fwd = orc.raster_forward(_np(means3D), _np(opacities), _np(st.viewmatrix), _np(st.projmatrix), _np(st.campos),
                                     st.image_width, st.image_height, scales=_np(scales), rotations=_np(rotations),
                                     colors_precomp=_np(colors), bg=_np(st.bg), scale_modifier=st.scale_modifier)
            ctx.fwd = fwd
            return (torch.from_numpy(fwd["out_color"]), torch.from_numpy(fwd["radii"]), torch.from_numpy(fwd["allmap"]),
                    torch.from_numpy(fwd["weight"].astype(np.float32))[:, None])

        @staticmethod
        def backward(ctx, g_color, g_radii, g_allmap, g_w):
            z = lambda g, s: np.zeros(s, np.float32) if g is None else _np(g)
            f = ctx.fwd
            b = orc.raster_backward(f, z(g_color, f["out_color"].shape), z(g_allmap, f["allmap"].shape))
            t = torch.from_numpy
            return (t(b["dmeans3D"]), t(b["dmeans2D"]), t(b["dcolors"]), t(b["dopacities"])[:, None], t(b["dscales"]), t(b["drots"]), None)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.st = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            assert shs is None and cov3D_precomp is None
            return _F.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.st)

    return SimpleNamespace(GaussianRasterizationSettings=GaussianRasterizationSettings, GaussianRasterizer=GaussianRasterizer)


def make_trace_pkg():
    class _F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, ray_o, ray_d, means3D, grads3D, shs, opacities, scales, rotations, ts, sff):
            lead = tuple(ray_o.shape[:-1])
            deg = int(ts.sh_degree.item()) if torch.is_tensor(ts.sh_degree) else int(ts.sh_degree)
            fwd = otr.trace_forward(_np(ray_o), _np(ray_d), _np(means3D), _np(scales), _np(rotations), _np(opacities), shs=_np(shs),
                                    sh_degree=deg, bg=_np(ts.bg), max_trace_depth=ts.max_trace_depth,
                                    specular_threshold=ts.specular_threshold, start_from_first=sff, scale_modifier=ts.scale_modifier)
            ctx.fwd, ctx.lead = fwd, lead
            t = torch.from_numpy
            return (t(fwd["rgb"]).reshape(lead + (3,)), t(fwd["dpt"]).reshape(lead + (1,)), t(fwd["acc"]).reshape(lead + (1,)),
                    t(fwd["norm"]).reshape(lead + (3,)), t(fwd["dist"]).reshape(lead + (1,)), t(fwd["aux"]).reshape(lead + (2,)),
                    t(fwd["mid"]).reshape(lead + (-1,)), t(fwd["wet"].astype(np.float32))[:, None])

        @staticmethod
        def backward(ctx, g_rgb, g_dpt, g_acc, g_norm, g_dist, g_aux, g_mid, g_wet):
            f = ctx.fwd
            R = f["rgb"].shape[0]
            z = lambda g, c: np.zeros((R, c), np.float32) if g is None else _np(g).reshape(R, c)
            b = otr.trace_backward(f, z(g_rgb, 3), z(g_dpt, 1)[:, 0], z(g_acc, 1)[:, 0], z(g_norm, 3), z(g_aux, 2))
            t = lambda a: torch.from_numpy(a.astype(np.float32))
            lead = ctx.lead
            # grads3D is the densification sink (optix_utils.py:134-136): it receives dL/dmeans3D, as the HIP package delivers it
            return (t(b["dray_o"]).reshape(lead + (3,)), t(b["dray_d"]).reshape(lead + (3,)), t(b["dmeans3D"]), t(b["dmeans3D"]), t(b["dshs"]),
                    t(b["dopacities"])[:, None], t(b["dscales"]), t(b["drots"]), None, None)

    class SurfelTracer(nn.Module):
        def build_acceleration_structure(self, vertices, faces=None, rebuild=True):
            pass

        def forward(self, ray_o, ray_d, v=None, *, means3D, grads3D=None, shs=None, colors_precomp=None, others_precomp=None,
                    opacities=None, scales=None, rotations=None, cov3D_precomp=None, tracer_settings=None, start_from_first=True):
            assert colors_precomp is None and others_precomp is None
            if grads3D is None:
                grads3D = torch.zeros_like(means3D)
            return _F.apply(ray_o, ray_d, means3D, grads3D, shs, opacities, scales, rotations, tracer_settings, start_from_first)

    return SimpleNamespace(SurfelTracer=SurfelTracer, SurfelTracingSettings=SurfelTracingSettings)
