"""PyTorch-eager (dense rays x surfels) restatement of the surfel tracer, float64 + autograd.
TEST INFRASTRUCTURE ONLY: validates the analytic backward of oracle/surfel_trace_oracle.c.
Same definitions as that file's header (reference boundary: easyvolcap/utils/optix_utils.py:188-201)."""
import torch

from .eager import _rotmat, C0, C1, C2, C3

NEAR_N, FAR_N = 0.2, 100.0


def _sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)            # (R, nb)


def trace(ray_o, ray_d, means3D, scales, rotations, opacities, *, shs=None, colors_precomp=None, others=None, sh_degree=0,
          bg=None, start_from_first=True, scale_modifier=1.0, tmin=None):
    """One stage of tracing.  Returns rgb (R,3), dpt (R), acc (R), norm (R,3), aux (R,2), wet (P).  tmin overrides the start_from_first rule
    (bounce stages start at 1e-3)."""
    dt = means3D.dtype
    R_, P = ray_o.shape[0], means3D.shape[0]
    Rm = _rotmat(rotations)
    a, b, n = Rm[:, :, 0], Rm[:, :, 1], Rm[:, :, 2]
    su, sv = scales[:, 0] * scale_modifier, scales[:, 1] * scale_modifier
    denom = ray_d @ n.t()                                               # (R,P)
    num = (n * means3D).sum(-1)[None] - ray_o @ n.t()
    ok = denom != 0
    t = num / torch.where(ok, denom, torch.ones_like(denom))
    if tmin is None:
        tmin = NEAR_N if start_from_first else 0.0
    q = ray_o[:, None] + t[..., None] * ray_d[:, None] - means3D[None]  # (R,P,3)
    u = (q * a[None]).sum(-1) / su[None]; v = (q * b[None]).sum(-1) / sv[None]
    G = torch.exp(-0.5 * (u * u + v * v))
    araw = opacities.reshape(-1)[None] * G
    alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()
    ok = ok & (t.detach() > tmin) & (u.detach().abs() <= 3) & (v.detach().abs() <= 3) & (alpha.detach() >= 1 / 255)
    # order by (t, id): stable sort on t with ids ascending
    tkey = torch.where(ok, t.detach(), torch.full_like(t, float("inf")))
    order = torch.sort(tkey, dim=1, stable=True).indices                # (R,P)
    gat = lambda x: torch.gather(x, 1, order)
    ok_s, al_s, t_s = gat(ok), gat(alpha), gat(t)
    al_s = torch.where(ok_s, al_s, torch.zeros_like(al_s))
    Tin = torch.cumprod(1 - al_s, dim=1)
    stop = ok_s & (Tin.detach() < 1e-4)
    alive = torch.cumsum(stop.to(torch.int32), dim=1) == 0
    al_s = al_s * alive.to(dt)
    Tin = torch.cumprod(1 - al_s, dim=1)
    Tex = torch.cat([torch.ones_like(Tin[:, :1]), Tin[:, :-1]], dim=1)
    w = al_s * Tex                                                      # (R,P) in sorted order
    Tfin = Tin[:, -1] if P > 0 else torch.ones(R_, dtype=dt)
    dirs = ray_d / ray_d.norm(dim=-1, keepdim=True)
    if shs is not None:
        nb = (sh_degree + 1) ** 2
        basis = _sh_basis(sh_degree, dirs)                              # (R,nb)
        col = torch.clamp_min(torch.einsum("rk,pkc->rpc", basis, shs[:, :nb]) + 0.5, 0.0)   # (R,P,3)
    else:
        col = colors_precomp[None].expand(R_, P, 3)
    col_s = torch.gather(col, 1, order[..., None].expand(-1, -1, 3))
    bgv = torch.zeros(3, dtype=dt)
    if bg is not None: bgv[:len(bg)] = bg.to(dt)
    rgb = (w[..., None] * col_s).sum(1) + Tfin[:, None] * bgv[None]
    t_safe = torch.where(ok_s, t_s, torch.zeros_like(t_s))
    dpt = (w * t_safe).sum(1)
    acc = w.sum(1)
    sgn = torch.where(denom.detach() < 0, 1.0, -1.0).to(dt)
    nf = sgn[..., None] * n[None]
    nf_s = torch.gather(nf, 1, order[..., None].expand(-1, -1, 3))
    norm = (w[..., None] * nf_s).sum(1)
    if others is not None:
        ot = others[None].expand(R_, P, 2)
        aux = (w[..., None] * torch.gather(ot, 1, order[..., None].expand(-1, -1, 2))).sum(1)
    else:
        aux = torch.zeros(R_, 2, dtype=dt)
    wet = torch.zeros(P, dtype=dt).index_add(0, order.reshape(-1), w.detach().reshape(-1))
    return rgb, dpt, acc, norm, aux, wet


def trace_bounces(ray_o, ray_d, means3D, scales, rotations, opacities, *, max_trace_depth, specular_threshold, **kw):
    """max_trace_depth > 0 as ONE differentiable expression (float64 autograd gives the true derivative of the blended colour):
    stage k+1 starts at o + d * dpt_k / acc_k along d - 2 (d.n) n (n = normalised accumulated normal) where aux_k[0] > threshold and
    acc_k > 0.5, t_min = 1e-3; rgb = (1 - s_0) rgb_0 + s_0 ((1 - s_1) rgb_1 + ...), s_k = aux_k[0]  (surfel_trace_oracle.c header;
    reference call sites gaussian2d_sampler.py:413-426, optix_utils.py:117-118).  Returns (rgb, dpt_0, acc_0, norm_0, aux_0, wet summed over the stages, n_stages)."""
    first = trace(ray_o, ray_d, means3D, scales, rotations, opacities, **kw)
    stages = [dict(o=ray_o, d=ray_d, out=first, sel=None)]
    kw2 = dict(kw); kw2.pop("start_from_first", None)
    for k in range(1, max_trace_depth + 1):
        p = stages[-1]
        rgb, dpt, acc, norm, aux, _ = p["out"]
        nl = norm.detach().norm(dim=-1)
        sel = ((aux.detach()[:, 0] > specular_threshold) & (acc.detach() > 0.5) & (nl > 0)).nonzero(as_tuple=False)[:, 0]
        if sel.numel() == 0:
            break
        po, pd = p["o"][sel], p["d"][sel]
        nh = norm[sel] / norm[sel].norm(dim=-1, keepdim=True)
        o2 = po + pd * (dpt[sel] / acc[sel])[:, None]
        d2 = pd - 2.0 * (pd * nh).sum(-1, keepdim=True) * nh
        stages.append(dict(o=o2, d=d2, out=trace(o2, d2, means3D, scales, rotations, opacities, start_from_first=False, tmin=1e-3, **kw2), sel=sel))
    col = stages[-1]["out"][0]
    for k in range(len(stages) - 2, -1, -1):
        p, c = stages[k], stages[k + 1]
        s = p["out"][4][c["sel"], 0:1]
        col = p["out"][0].index_put((c["sel"],), (1.0 - s) * p["out"][0][c["sel"]] + s * col)
    rgb0, dpt0, acc0, norm0, aux0, wet0 = first
    wet = wet0
    for st in stages[1:]:            # wet: blend weights summed over ALL stages (a surfel blended only by bounce rays is visible too)
        wet = wet + st["out"][5]
    return col, dpt0, acc0, norm0, aux0, wet, len(stages)
