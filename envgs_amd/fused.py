"""Fused caller-side glue (include/envgs_glue.h): drop-in replacements for the torch expressions the
reference evaluates between its two extension calls.  Optional -- the extensions do not need them.

    sh_colors(means3D, shs, campos, sh_degree, specular, roughness) -> (P, 3+S+1)
        == cat([clamp_min(eval_sh(deg, shs^T, normalize(xyz - campos)) + 0.5, 0), specular, roughness], -1)     gaussian2d_utils.py:1071-1084
    reflect(allmap, ray_o, ray_d, viewmatrix, depth_ratio=0.0) -> normal_world (3,H,W), depth (1,H,W), ref_o (H,W,3), ref_d (H,W,3)
        == gaussian2d_utils.py:1119-1136 + envgs_sampler.py:420-431
"""
import torch

from . import _lib
from .raster import sh_degree_of


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _stream(dev):
    return _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _ShColors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, shs, campos, deg, specular, roughness):
        lib = _lib.load()
        if means3D.device.type != "cuda":
            raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
        means3D, shs, specular, roughness = _f32c(means3D), _f32c(shs), _f32c(specular), _f32c(roughness)
        campos = _f32c(campos).reshape(-1)
        P, M, S = means3D.shape[0], shs.shape[1], specular.shape[1]
        colors = torch.empty(P, 3 + S + 1, dtype=torch.float32, device=means3D.device)
        clamped = torch.empty(P, 3, dtype=torch.uint8, device=means3D.device)
        p = _lib.ptr
        _lib.check(lib.envgs_sh_colors_forward(P, deg, M, S, p(means3D), p(shs), p(campos), p(specular), p(roughness), p(colors), p(clamped),
                                               _stream(means3D.device)), "envgs_sh_colors_forward")
        ctx.save_for_backward(means3D, shs, campos, clamped)
        ctx.meta = (P, deg, M, S)
        return colors

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        means3D, shs, campos, clamped = ctx.saved_tensors
        P, deg, M, S = ctx.meta
        g = _f32c(g)
        dm = torch.empty_like(means3D); dsh = torch.empty_like(shs)
        dsp = torch.empty(P, S, dtype=torch.float32, device=g.device); dr = torch.empty(P, 1, dtype=torch.float32, device=g.device)
        p = _lib.ptr
        _lib.check(lib.envgs_sh_colors_backward(P, deg, M, S, p(means3D), p(shs), p(campos), p(clamped), p(g), p(dm), p(dsh), p(dsp), p(dr),
                                                _stream(g.device)), "envgs_sh_colors_backward")
        return dm, dsh, None, None, dsp, dr


def sh_colors(means3D, shs, campos, sh_degree, specular, roughness):
    return _ShColors.apply(means3D, shs, campos, sh_degree_of(sh_degree), specular, roughness)


class _Reflect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, ray_o, ray_d, viewmatrix, depth_ratio):
        ctx.set_materialize_grads(False)          # outputs the loss does not use arrive as None (= NULL upstream pointer), not as buffers of zeros
        lib = _lib.load()
        if allmap.device.type != "cuda":
            raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
        allmap, ray_o, ray_d, viewmatrix = _f32c(allmap), _f32c(ray_o), _f32c(ray_d), _f32c(viewmatrix)
        _, H, W = allmap.shape
        f32 = dict(dtype=torch.float32, device=allmap.device)
        nw = torch.empty(3, H, W, **f32); dep = torch.empty(1, H, W, **f32)
        ref_o = torch.empty(H, W, 3, **f32); ref_d = torch.empty(H, W, 3, **f32)
        p = _lib.ptr
        _lib.check(lib.envgs_reflect_forward(H, W, float(depth_ratio), p(allmap), p(ray_o), p(ray_d), p(viewmatrix), p(nw), p(dep), p(ref_o),
                                             p(ref_d), _stream(allmap.device)), "envgs_reflect_forward")
        ctx.save_for_backward(allmap, ray_o, ray_d, viewmatrix)
        ctx.ratio = float(depth_ratio)
        ctx.need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return nw, dep, ref_o, ref_d

    @staticmethod
    def backward(ctx, g_nw, g_dep, g_ro, g_rd):
        lib = _lib.load()
        allmap, ray_o, ray_d, viewmatrix = ctx.saved_tensors
        _, H, W = allmap.shape
        c = lambda g: None if g is None else _f32c(g)
        g_nw, g_dep, g_ro, g_rd = c(g_nw), c(g_dep), c(g_ro), c(g_rd)
        dall = torch.empty_like(allmap)
        dro = torch.empty_like(ray_o) if ctx.need[0] else None
        drd = torch.empty_like(ray_d) if ctx.need[1] else None
        p = _lib.ptr
        _lib.check(lib.envgs_reflect_backward(H, W, ctx.ratio, p(allmap), p(ray_o), p(ray_d), p(viewmatrix), p(g_nw), p(g_dep), p(g_ro), p(g_rd),
                                              p(dall), p(dro), p(drd), _stream(allmap.device)), "envgs_reflect_backward")
        return dall, dro, drd, None, None


def reflect(allmap, ray_o, ray_d, viewmatrix, depth_ratio=0.0):
    return _Reflect.apply(allmap, ray_o, ray_d, viewmatrix, depth_ratio)


class _SurfaceNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, viewmatrix, fx, fy, depth_ratio):
        ctx.set_materialize_grads(False)          # outputs the loss does not use arrive as None (= NULL upstream pointer), not as buffers of zeros
        lib = _lib.load()
        if allmap.device.type != "cuda":
            raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
        allmap, viewmatrix = _f32c(allmap), _f32c(viewmatrix)
        _, H, W = allmap.shape
        f32 = dict(dtype=torch.float32, device=allmap.device)
        sd = torch.empty(1, H, W, **f32); sn = torch.empty(3, H, W, **f32)
        p = _lib.ptr
        _lib.check(lib.envgs_surface_normal_forward(H, W, float(depth_ratio), float(fx), float(fy), p(allmap), p(viewmatrix), p(sd), p(sn),
                                                    _stream(allmap.device)), "envgs_surface_normal_forward")
        ctx.save_for_backward(allmap, viewmatrix)
        ctx.cfg = (float(depth_ratio), float(fx), float(fy))
        return sd, sn

    @staticmethod
    def backward(ctx, g_sd, g_sn):
        lib = _lib.load()
        allmap, viewmatrix = ctx.saved_tensors
        _, H, W = allmap.shape
        ratio, fx, fy = ctx.cfg
        c = lambda g: None if g is None else _f32c(g)
        g_sd, g_sn = c(g_sd), c(g_sn)
        dall = torch.empty_like(allmap)
        p = _lib.ptr
        _lib.check(lib.envgs_surface_normal_backward(H, W, ratio, fx, fy, p(allmap), p(viewmatrix), p(g_sd), p(g_sn), p(dall),
                                                     _stream(allmap.device)), "envgs_surface_normal_backward")
        return dall, None, None, None, None


def surface_normal(allmap, cam, depth_ratio=0.0):
    """surf_depth (1,H,W), surf_normal (3,H,W) of render()'s tail (gaussian2d_utils.py:1125-1142, dpt2norm :1190-1206) in one kernel each way;
    cam: the prepare_gaussian_camera namespace (image size, FoVx / FoVy, world_view_transform)."""
    import math
    fx = cam.image_width / (2.0 * math.tan(cam.FoVx / 2.0))
    fy = cam.image_height / (2.0 * math.tan(cam.FoVy / 2.0))
    return _SurfaceNormal.apply(allmap, cam.world_view_transform, fx, fy, depth_ratio)


_FACES = {}


def surfel_quads(means3D, scales, rotations):
    """get_disks (optix_utils.py:39-69) in one launch: v (4P,3) f32, f (2P,3) i32.  No gradient flows through the quads (they only seed the
    acceleration structure); the face table depends on P alone and is cached."""
    lib = _lib.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
    P = means3D.shape[0]
    m, s, q = _f32c(means3D.detach()), _f32c(scales.detach()), _f32c(rotations.detach())
    v = torch.empty(4 * P, 3, dtype=torch.float32, device=dev)
    f = _FACES.get((dev.index, P))
    fresh = f is None
    if fresh:
        f = _FACES[(dev.index, P)] = torch.empty(2 * P, 3, dtype=torch.int32, device=dev)
        if len(_FACES) > 8:
            _FACES.pop(next(iter(_FACES)))
    p = _lib.ptr
    _lib.check(lib.envgs_surfel_quads(P, p(m), p(s), p(q), p(v), p(f) if fresh else None, _stream(dev)), "envgs_surfel_quads")
    return v, f


class _Blend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, rgb_env):
        lib = _lib.load()
        if img.device.type != "cuda":
            raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
        img, rgb_env = _f32c(img), _f32c(rgb_env)
        C, H, W = img.shape
        rgb = torch.empty(H, W, 3, dtype=torch.float32, device=img.device)
        p = _lib.ptr
        _lib.check(lib.envgs_blend_forward(H, W, C, p(img), p(rgb_env), p(rgb), _stream(img.device)), "envgs_blend_forward")
        ctx.save_for_backward(img, rgb_env)
        return rgb

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        img, rgb_env = ctx.saved_tensors
        C, H, W = img.shape
        g = _f32c(g)
        dimg = torch.empty_like(img)
        denv = torch.empty_like(rgb_env) if ctx.needs_input_grad[1] else None
        p = _lib.ptr
        _lib.check(lib.envgs_blend_backward(H, W, C, p(img), p(rgb_env), p(g), p(dimg), p(denv), _stream(img.device)), "envgs_blend_backward")
        return dimg, denv


def blend(img, rgb_env):
    """(1 - s) * img[:3] + s * rgb_env  (H,W,3) from the -ch05 / -ch07 rasterizer output img (C,H,W) = [rgb | specular C-4 | roughness] and the
    traced colour rgb_env (H,W,3): envgs_sampler.py:474, one launch each way instead of ~20 (include/envgs_glue.h)."""
    return _Blend.apply(img, rgb_env)



class _BounceRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_o, ray_d, dpt, acc, norm, sel):
        lib = _lib.load()
        if ray_o.device.type != "cuda":
            raise RuntimeError("envgs_amd.fused needs tensors on the GPU; there is no CPU path")
        ray_o, ray_d, dpt, acc, norm = (_f32c(t) for t in (ray_o, ray_d, dpt, acc, norm))
        sel = sel.contiguous()
        n = int(sel.numel())
        o2 = torch.empty(n, 3, dtype=torch.float32, device=ray_o.device); d2 = torch.empty_like(o2)
        p = _lib.ptr
        _lib.check(lib.envgs_bounce_rays_forward(n, p(sel), p(ray_o), p(ray_d), p(dpt), p(acc), p(norm), p(o2), p(d2), _stream(ray_o.device)),
                   "envgs_bounce_rays_forward")
        ctx.save_for_backward(ray_o, ray_d, dpt, acc, norm, sel)
        return o2, d2

    @staticmethod
    def backward(ctx, g_o2, g_d2):
        lib = _lib.load()
        ray_o, ray_d, dpt, acc, norm, sel = ctx.saved_tensors
        n = int(sel.numel())
        need = ctx.needs_input_grad
        outs = [torch.zeros_like(t) if need[i] else None for i, t in enumerate((ray_o, ray_d, dpt, acc, norm))]
        p = _lib.ptr
        _lib.check(lib.envgs_bounce_rays_backward(n, p(sel), p(ray_o), p(ray_d), p(dpt), p(acc), p(norm), p(_f32c(g_o2)), p(_f32c(g_d2)),
                                                  *[p(t) for t in outs], _stream(ray_o.device)), "envgs_bounce_rays_backward")
        return (*outs, None)


def bounce_rays(ray_o, ray_d, dpt, acc, norm, sel):
    """Rays of the next bounce stage from the rows `sel` (unique int64 indices) of a stage's rays (R,3) and outputs dpt / acc (R,1), norm (R,3):
    o2 = o + d dpt/acc, d2 = d - 2 (d.n) n with n = norm/|norm| -- (n,3) each; differentiable in everything but `sel`; one launch each way
    (include/envgs_glue.h: envgs_bounce_rays_forward)."""
    return _BounceRays.apply(ray_o, ray_d, dpt, acc, norm, sel)


class _BounceBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, aux, col_next, sel):
        lib = _lib.load()
        rgb, aux, col_next = _f32c(rgb), _f32c(aux), _f32c(col_next)
        sel = sel.contiguous()
        col = rgb.clone()
        p = _lib.ptr
        _lib.check(lib.envgs_bounce_blend_forward(int(sel.numel()), p(sel), p(rgb), p(aux), p(col_next), p(col), _stream(rgb.device)),
                   "envgs_bounce_blend_forward")
        ctx.save_for_backward(rgb, aux, col_next, sel)
        return col

    @staticmethod
    def backward(ctx, g_col):
        lib = _lib.load()
        rgb, aux, col_next, sel = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_col = _f32c(g_col)
        g_rgb = g_col.clone() if need[0] else None
        g_aux = torch.zeros_like(aux) if need[1] else None
        g_next = torch.empty_like(col_next) if need[2] else None
        p = _lib.ptr
        _lib.check(lib.envgs_bounce_blend_backward(int(sel.numel()), p(sel), p(rgb), p(aux), p(col_next), p(g_col), p(g_rgb), p(g_aux), p(g_next),
                                                   _stream(rgb.device)), "envgs_bounce_blend_backward")
        return g_rgb, g_aux, g_next, None


def bounce_blend(rgb, aux, col_next, sel):
    """A stage's colour with the next stage blended in at the rows that bounced: rgb (R,3) with rows sel replaced by
    (1 - s) rgb[sel] + s col_next, s = aux[sel, 0]; differentiable in rgb, aux, col_next (include/envgs_glue.h: envgs_bounce_blend_forward)."""
    return _BounceBlend.apply(rgb, aux, col_next, sel)


def bounce_pack_mid(mid, k, stages, idx, ray_o, ray_d, dpt, acc, norm, aux, rgb):
    """Write stage k's 16 `mid` channels [o | d | dpt | acc | norm | aux | rgb] into mid (R, 16 * stages) at rows idx (None: row i).  No gradient."""
    lib = _lib.load()
    p = _lib.ptr
    ts = [_f32c(t.detach()) for t in (ray_o, ray_d, dpt, acc, norm, aux, rgb)]
    _lib.check(lib.envgs_bounce_pack_mid(int(ts[0].shape[0]), p(idx.contiguous()) if idx is not None else None, int(stages), int(k), *[p(t) for t in ts],
                                         p(mid), _stream(mid.device)), "envgs_bounce_pack_mid")
