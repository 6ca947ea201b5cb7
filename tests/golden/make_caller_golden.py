"""Generate tests/golden/caller_golden.pt by RUNNING the reference's own caller code -- render() (easyvolcap/utils/gaussian2d_utils.py:
1003-1155) and HardwareRendering.render_gaussians() (easyvolcap/utils/optix_utils.py:87-267) -- on CPU in the authoring container, with
RECORDING stand-ins for the four extension packages (diff_surfel_rasterization_wet{,_ch05,_ch07}, diff_surfel_tracing) that forward to
the CPU oracle (tests/oracle_packages.py).  /root/reference does not exist on the GPU box; the fixture is data only:

  contract   : for every extension call the reference made -- settings field names / python types / tensor shapes+dtypes, every keyword
               with shape / dtype / requires_grad / contiguity (or None), the outputs' shapes / dtypes; and the import statements used
  tensors    : the seeded model parameters, the exact tensors the reference handed to the extensions, what the (oracle) extensions
               returned, and the output dicts render() / render_gaussians() built from them

Consumers: tests/test_caller_contract.py (drop-in signatures accept exactly this; the HIP packages reproduce the boundary outputs;
envgs_amd/envgs_step.py's re-derived caller reproduces the reference's output dicts).  Re-run: python tests/golden/make_caller_golden.py"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CALLS = []


def _desc(v):
    if v is None:
        return None
    if torch.is_tensor(v):
        return dict(kind="tensor", shape=list(v.shape), dtype=str(v.dtype).replace("torch.", ""), requires_grad=bool(v.requires_grad),
                    contiguous=bool(v.is_contiguous()))
    return dict(kind=type(v).__name__, value=(v if isinstance(v, (int, float, bool)) else None))


def _recording_raster_pkg(name, C):
    from tests.oracle_packages import make_raster_pkg
    inner = make_raster_pkg(C)
    mod = types.ModuleType(name)

    class GaussianRasterizationSettings(inner.GaussianRasterizationSettings):
        pass

    def settings(**kw):
        CALLS.append(dict(package=name, what="settings", fields={k: _desc(v) for k, v in kw.items()}, order=list(kw)))
        return inner.GaussianRasterizationSettings(**kw)

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.inner = inner.GaussianRasterizer(raster_settings=raster_settings)
            self.settings = raster_settings

        def forward(self, **kw):
            rec = dict(package=name, what="call", kwargs={k: _desc(v) for k, v in kw.items()}, order=list(kw))
            outs = self.inner(**kw)
            rec["outputs"] = [_desc(o) for o in outs]
            rec["tensors"] = dict(kwargs={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()},
                                  settings={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.settings._asdict().items()},
                                  outputs=[o.detach().clone() for o in outs])
            CALLS.append(rec)
            return outs

    mod.GaussianRasterizationSettings = settings
    mod.GaussianRasterizer = GaussianRasterizer
    return mod


def _recording_trace_pkg():
    from tests.oracle_packages import make_trace_pkg
    inner = make_trace_pkg()
    mod = types.ModuleType("diff_surfel_tracing")

    def settings(**kw):
        CALLS.append(dict(package="diff_surfel_tracing", what="settings", fields={k: _desc(v) for k, v in kw.items()}, order=list(kw)))
        return inner.SurfelTracingSettings(**kw)

    class SurfelTracer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.tracer_id = sum(1 for c_ in CALLS if c_.get("what") == "SurfelTracer()")      # which live tracer object a record belongs to
            CALLS.append(dict(package="diff_surfel_tracing", what="SurfelTracer()", args=[], tracer=self.tracer_id))
            self.inner = inner.SurfelTracer()

        def build_acceleration_structure(self, *a, **kw):
            CALLS.append(dict(package="diff_surfel_tracing", what="build_acceleration_structure", args=[_desc(x) for x in a], kwargs={k: _desc(v) for k, v in kw.items()},
                              tracer=self.tracer_id))
            return self.inner.build_acceleration_structure(*a, **kw)

        def forward(self, *a, **kw):
            rec = dict(package="diff_surfel_tracing", what="call", args=[_desc(x) for x in a], kwargs={k: _desc(v) for k, v in kw.items()}, order=list(kw),
                       tracer=self.tracer_id)
            outs = self.inner(*a, **kw)
            rec["outputs"] = [_desc(o) for o in outs]
            st = kw["tracer_settings"]
            rec["tensors"] = dict(args=[(x.detach().clone() if torch.is_tensor(x) else x) for x in a],
                                  kwargs={k: (v.detach().clone() if torch.is_tensor(v) else (None if k == "tracer_settings" else v)) for k, v in kw.items()},
                                  settings={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st._asdict().items()},
                                  outputs=[o.detach().clone() for o in outs])
            CALLS.append(rec)
            return outs

    mod.SurfelTracingSettings = settings
    mod.SurfelTracer = SurfelTracer
    return mod


def main():
    for m in ("pdbr", "pdbr.utils", "ruamel", "ruamel.yaml", "plyfile"):
        sys.modules[m] = MagicMock()
    sys.modules["ujson"] = json
    for name, C in (("diff_surfel_rasterization_wet", 3), ("diff_surfel_rasterization_wet_ch05", 5), ("diff_surfel_rasterization_wet_ch07", 7)):
        sys.modules[name] = _recording_raster_pkg(name, C)
    sys.modules["diff_surfel_tracing"] = _recording_trace_pkg()
    sys.path.insert(0, "/root/reference")
    from easyvolcap.utils import gaussian2d_utils as g2d
    from easyvolcap.utils import optix_utils
    from easyvolcap.utils.base_utils import dotdict
    from easyvolcap.utils.math_utils import normalize
    # the only adaptation: there is no GPU here, so the reference's `device='cuda'` DEFAULT of dpt2norm is redirected to the CPU
    # (render() and render_gaussians() look the name up in their own modules; the function bodies run unchanged)
    _dpt2norm = g2d.dpt2norm
    g2d.dpt2norm = lambda camera, dpt, device="cpu": _dpt2norm(camera, dpt, "cpu")
    optix_utils.dpt2norm = g2d.dpt2norm

    torch.manual_seed(0)
    H, W = 48, 64
    P, Pe = 300, 200
    K = torch.tensor([[1111.1 * W / 800.0, 0, W / 2], [0, 1111.1 * W / 800.0, H / 2], [0, 0, 1]])
    c = torch.tensor([2.6, 2.2, 1.8])
    fwd = -c / c.norm(); up = torch.tensor([0., 0., 1.])
    right = torch.linalg.cross(fwd, up); right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd]); T = -(R @ c).reshape(3, 1)
    n, f = torch.tensor(2.0), torch.tensor(6.0)
    batch = dotdict(H=[H], W=[W], K=K[None], R=R[None], T=T[None], n=n[None], f=f[None],
                    meta=dotdict(H=torch.tensor([H]), W=torch.tensor([W]), K=K[None], R=R[None], T=T[None], n=n[None], f=f[None]))
    cam = g2d.prepare_gaussian_camera(batch)

    def model(P_, spread, scale_lo, scale_hi, reflection):
        m = g2d.GaussianModel(xyz=(torch.rand(P_, 3) * 2 - 1) * spread, colors=torch.rand(P_, 3), init_occ=0.1,
                              init_scale=torch.log(torch.rand(P_, 2) * (scale_hi - scale_lo) + scale_lo), sh_degree=3, init_sh_degree=3,
                              render_reflection=reflection, xyz_lr_scheduler=None, max_gs=10 ** 6, max_gs_threshold=0.9, spatial_scale=1.0)
        with torch.no_grad():
            m._opacity.add_(torch.randn_like(m._opacity) + 2.0)
            m._features_rest.add_(0.1 * torch.randn_like(m._features_rest))
            m._rotation.copy_(torch.randn_like(m._rotation))
            if reflection:
                m._specular.add_(torch.randn_like(m._specular)); m._roughness.add_(0.3 * torch.randn_like(m._roughness))
        return m

    pcd = model(P, 1.0, 0.05, 0.25, True)
    env = model(Pe, 8.0, 0.8, 2.5, False)
    pipe = dotdict(convert_SHs_python=True, compute_cov3D_python=False, depth_ratio=0.0, debug=False)
    pipe_env = dotdict(convert_SHs_python=False, compute_cov3D_python=False, depth_ratio=0.0, debug=False)
    bg = torch.zeros(3); env_bg = torch.tensor([0.1, 0.2, 0.3])

    # --- the reference's own base pass -------------------------------------------------------------------------------------
    out_base = g2d.render(cam, pcd, pipe, bg, 1.0, None, device="cpu")
    # --- reflected rays exactly as EnvGSSampler.get_reflect_rays builds them (envgs_sampler.py:420-431; (B,P,3) maps) --------
    from easyvolcap.utils.ray_utils import get_rays
    ray_o, ray_d = get_rays(H, W, K, R, T, z_depth=True, correct_pix=True)
    norm_map = out_base.rend_normal.permute(1, 2, 0).reshape(1, H * W, 3)
    dpt_map = out_base.surf_depth.permute(1, 2, 0).reshape(1, H * W, 1)
    nrm = normalize(norm_map)
    ref_d = ray_d.reshape(1, -1, 3) - 2 * torch.sum(ray_d.reshape(1, -1, 3) * nrm, dim=-1, keepdim=True) * nrm
    ref_o = ray_o.reshape(1, -1, 3) + ray_d.reshape(1, -1, 3) * dpt_map
    ref_o, ref_d = ref_o.reshape(H, W, 3), ref_d.reshape(H, W, 3)
    # --- the reference's own env pass ---------------------------------------------------------------------------------------
    hw = optix_utils.HardwareRendering()
    hw.train()
    out_env = hw.render_gaussians(cam, ref_o, ref_d, env, pipe_env, env_bg, 0, start_from_first=False, scaling_modifier=1.0, override_color=None, batch=batch)
    spec = out_base.specular.permute(1, 2, 0)
    rgb = (1 - spec) * out_base.render.permute(1, 2, 0) + spec * out_env.render.permute(1, 2, 0)         # envgs_sampler.py:474
    loss = (rgb * torch.linspace(0.5, 1.5, 3)).sum() / (H * W) + (out_base.rend_normal * out_base.surf_normal).sum() / (H * W)
    loss.backward()

    contract = [{k: v for k, v in c_.items() if k != "tensors"} for c_ in CALLS]
    tens = [c_.get("tensors") for c_ in CALLS]
    td = lambda d: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items() if torch.is_tensor(v)}
    raw = lambda m: {k: getattr(m, k).detach().clone() for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_specular", "_roughness")
                     if hasattr(m, k) and torch.is_tensor(getattr(m, k))}
    acts = lambda m: dict(xyz=m.get_xyz.detach().clone(), features=m.get_features.detach().clone(), opacity=m.get_opacity.detach().clone(),
                          scaling=m.get_scaling.detach().clone(), rotation=m.get_rotation.detach().clone(),
                          **({"specular": m.get_specular.detach().clone(), "roughness": m.get_roughness.detach().clone()} if m.render_reflection else {}))
    grads = lambda m: {k: getattr(m, k).grad.detach().clone() for k in raw(m) if getattr(m, k).grad is not None}
    fixture = dict(contract=contract, call_tensors=tens, H=H, W=W, K=K, R=R, T=T, n=n, f=f,
                   camera={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in cam.items()},
                   pcd_raw=raw(pcd), env_raw=raw(env), pcd_act=acts(pcd), env_act=acts(env), active_sh_degree=int(pcd.active_sh_degree.item()),
                   bg=bg, env_bg=env_bg, rays=(ray_o.reshape(H, W, 3).clone(), ray_d.reshape(H, W, 3).clone()), ref_rays=(ref_o.detach().clone(), ref_d.detach().clone()),
                   out_base=td(out_base), out_env=td(out_env), rgb=rgb.detach().clone(), pcd_grad=grads(pcd), env_grad=grads(env),
                   imports=["from diff_surfel_rasterization_wet_ch05 import GaussianRasterizationSettings, GaussianRasterizer  (gaussian2d_utils.py:1013)",
                            "from diff_surfel_tracing import SurfelTracer, SurfelTracingSettings  (optix_utils.py:7)"])
    torch.save(fixture, os.path.join(HERE, "caller_golden.pt"))
    with open(os.path.join(HERE, "caller_contract.json"), "w") as fh:
        json.dump(contract, fh, indent=1)
    print("calls recorded:", [(c_["package"], c_["what"]) for c_ in CALLS])
    print("out_base keys:", sorted(out_base.keys()))
    print("out_env keys:", sorted(out_env.keys()))
    print("fixture bytes:", os.path.getsize(os.path.join(HERE, "caller_golden.pt")))


if __name__ == "__main__":
    main()
