"""PyTorch-eager CPU restatement of the surfel rasterizer (project + depth sort + per-pixel alpha blend).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two uses:
  * float64 + autograd: validates the hand-written backward of oracle/surfel_raster_oracle.c
    (tests/test_oracle_grad.py) -- every gradient of R7/R8 is a true derivative except the two documented
    quirks (0.99 alpha cap treated as identity; means2D.grad is a densification proxy), which are
    reproduced here with a straight-through cap and excluded respectively.
  * float32, no grad: BASELINE config 1 ("2k surfels, 256x256, PyTorch-eager CPU project+alpha-blend"),
    timed by bench.py as cpu_baseline.

Same algorithm and constants as surfel_raster_oracle.c; dense (surfel x pixel) evaluation in pixel chunks
instead of tiles, with the same tile-membership mask so results are comparable bit-for-bit in structure.
Boundary followed: easyvolcap/utils/gaussian2d_utils.py:1025-1061,1089-1144 (reference call site).
"""
import torch

TILE = 16
NEAR_N, FAR_N = 0.2, 100.0
FILTER_SIZE, FILTER_INV_SQ = 0.707106, 2.0
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _sh_rgb(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = C0 * shs[:, 0]
    if deg > 0:
        r = r - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14]
             + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def project(means3D, scales, rotations, opacities, viewmatrix, projmatrix, W, H, scale_modifier=1.0,
            transmat_precomp=None):
    """R1 (differentiable where the reference is).  Returns dict; `keep` = surfels surviving all culls."""
    dt = means3D.dtype
    V, FP = viewmatrix.to(dt), projmatrix.to(dt)
    PM = torch.stack([W / 2 * FP[:, 0] + (W - 1) / 2 * FP[:, 3], H / 2 * FP[:, 1] + (H - 1) / 2 * FP[:, 3], FP[:, 3]], dim=1)
    p_view = means3D @ V[:3, :3] + V[3, :3]
    if transmat_precomp is None:
        R = _rotmat(rotations)
        a = R[:, :, 0] * (scales[:, 0:1] * scale_modifier)
        b = R[:, :, 1] * (scales[:, 1:2] * scale_modifier)
        Trow = lambda c: torch.stack([a @ PM[:3, c], b @ PM[:3, c], means3D @ PM[:3, c] + PM[3, c]], dim=-1)
        Tu, Tv, Tw = Trow(0), Trow(1), Trow(2)
        normal = R[:, :, 2] @ V[:3, :3]
    else:
        Tm = transmat_precomp.reshape(-1, 3, 3)
        Tu, Tv, Tw = Tm[:, 0], Tm[:, 1], Tm[:, 2]
        normal = torch.zeros_like(means3D); normal[:, 2] = 1
    cosv = -(p_view * normal).sum(-1)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0).to(dt)[:, None].detach()
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    d = (t * Tw * Tw).sum(-1)
    f = t[None] / d[:, None]
    cx = (f * Tu * Tw).sum(-1); cy = (f * Tv * Tw).sum(-1)
    hx = cx * cx - (f * Tu * Tu).sum(-1); hy = cy * cy - (f * Tv * Tv).sum(-1)
    ext = torch.sqrt(torch.clamp_min(torch.stack([hx, hy], -1), 1e-4)).detach()
    radius = torch.ceil(torch.clamp_min(ext.max(-1).values, 3.0 * FILTER_SIZE))
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    cxd, cyd = cx.detach().float(), cy.detach().float()
    rad = radius.float()
    x0 = ((cxd - rad) / TILE).to(torch.int32).clamp(0, gx); y0 = ((cyd - rad) / TILE).to(torch.int32).clamp(0, gy)
    x1 = ((cxd + rad + TILE - 1) / TILE).to(torch.int32).clamp(0, gx); y1 = ((cyd + rad + TILE - 1) / TILE).to(torch.int32).clamp(0, gy)
    keep = (p_view[:, 2] > NEAR_N) & (cosv != 0) & (d != 0) & ((x1 - x0) * (y1 - y0) > 0)
    return dict(Tu=Tu, Tv=Tv, Tw=Tw, normal=normal, xy=torch.stack([cx, cy], -1), depth=p_view[:, 2], radius=radius,
                rect=(x0, y0, x1, y1), keep=keep)


def rasterize(means3D, opacities, viewmatrix, projmatrix, campos, W, H, *, scales=None, rotations=None,
              transmat_precomp=None, shs=None, colors_precomp=None, sh_degree=0, bg=None, scale_modifier=1.0,
              pix_chunk=4096):
    """Returns (out_color (C,H,W), radii (P), allmap (7,H,W), weight (P)).  Differentiable."""
    dt = means3D.dtype
    pr = project(means3D, scales, rotations, opacities, viewmatrix, projmatrix, W, H, scale_modifier, transmat_precomp)
    if shs is not None:
        dirs = means3D - campos.to(dt)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        colors = _sh_rgb(int(sh_degree), shs, dirs)
    else:
        colors = colors_precomp
    C = colors.shape[1]
    bgv = torch.zeros(C, dtype=dt)
    if bg is not None: bgv[:min(C, len(bg))] = bg.to(dt)[:C]
    keep = pr["keep"]
    idx = torch.nonzero(keep)[:, 0]
    depth32 = pr["depth"].detach().float()[idx]
    order = idx[torch.sort(depth32.view(torch.int32), stable=True).indices]      # positive floats: bit order == value order
    Tu, Tv, Tw = pr["Tu"][order], pr["Tv"][order], pr["Tw"][order]
    nrm, xy, opa, col = pr["normal"][order], pr["xy"][order], opacities.reshape(-1)[order], colors[order]
    x0, y0, x1, y1 = [r[order] for r in pr["rect"]]
    G = order.shape[0]
    HW = H * W
    out_color = torch.zeros(C, HW, dtype=dt); allmap = torch.zeros(7, HW, dtype=dt)
    weight = torch.zeros(means3D.shape[0], dtype=dt)
    pix_all = torch.arange(HW)
    for s in range(0, HW, pix_chunk):
        pid = pix_all[s:s + pix_chunk]
        px = (pid % W).to(dt)[None]; py = (pid // W).to(dt)[None]
        tx = ((pid % W) // TILE)[None]; ty = ((pid // W) // TILE)[None]
        member = (tx >= x0[:, None]) & (tx < x1[:, None]) & (ty >= y0[:, None]) & (ty < y1[:, None])
        k = [px * Tw[:, i:i + 1] - Tu[:, i:i + 1] for i in range(3)]
        l = [py * Tw[:, i:i + 1] - Tv[:, i:i + 1] for i in range(3)]
        p0 = k[1] * l[2] - k[2] * l[1]; p1 = k[2] * l[0] - k[0] * l[2]; p2 = k[0] * l[1] - k[1] * l[0]
        ok = member & (p2 != 0)
        p2s = torch.where(p2 != 0, p2, torch.ones_like(p2))
        sx, sy = p0 / p2s, p1 / p2s
        rho3d = sx * sx + sy * sy
        dx, dy = xy[:, 0:1] - px, xy[:, 1:2] - py
        rho2d = FILTER_INV_SQ * (dx * dx + dy * dy)
        use3d = rho3d <= rho2d
        rho = torch.where(use3d, rho3d, rho2d)
        dep = torch.where(use3d, sx * Tw[:, 0:1] + sy * Tw[:, 1:2] + Tw[:, 2:3], Tw[:, 2:3].expand_as(sx))
        Gv = torch.exp(-0.5 * rho)
        a_raw = opa[:, None] * Gv
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()          # straight-through cap (2DGS quirk)
        ok = ok & (dep >= NEAR_N) & (alpha.detach() >= 1.0 / 255.0)
        alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
        one_m = 1 - alpha
        Tincl = torch.cumprod(one_m, dim=0)
        Texcl = torch.cat([torch.ones_like(Tincl[:1]), Tincl[:-1]], dim=0)
        stop = ok & (Tincl.detach() < 1e-4)
        alive = (torch.cumsum(stop.to(torch.int32), dim=0) == 0)
        alpha = alpha * alive.to(dt)
        Tincl = torch.cumprod(1 - alpha, dim=0)
        Texcl = torch.cat([torch.ones_like(Tincl[:1]), Tincl[:-1]], dim=0)
        w = alpha * Texcl
        Tfin = Tincl[-1] if G > 0 else torch.ones(pid.shape[0], dtype=dt)
        deps = torch.where(ok, dep, torch.ones_like(dep))
        m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / deps)
        A = 1 - Texcl
        M1 = torch.cumsum(m * w, 0) - m * w
        M2 = torch.cumsum(m * m * w, 0) - m * m * w
        dist = ((m * m * A + M2 - 2 * m * M1) * w).sum(0)
        contrib = (alpha.detach() > 0)
        medsel = contrib & (Texcl.detach() > 0.5)
        ar = torch.arange(G)[:, None].expand_as(medsel)
        medidx = torch.where(medsel, ar, torch.full_like(ar, -1)).max(0).values
        med = torch.where(medidx >= 0, torch.gather(dep, 0, medidx.clamp_min(0)[None])[0], torch.zeros_like(Tfin))
        out_color[:, pid] = (col.t() @ w) + Tfin[None] * bgv[:, None]
        allmap[0, pid] = (w * deps).sum(0)
        allmap[1, pid] = 1 - Tfin
        allmap[2:5, pid] = nrm.t() @ w
        allmap[5, pid] = med
        allmap[6, pid] = dist
        weight = weight.index_add(0, order, w.detach().sum(1))
    radii = torch.where(keep, pr["radius"].to(torch.int32), torch.zeros_like(pr["radius"], dtype=torch.int32))
    return out_color.reshape(C, H, W), radii, allmap.reshape(7, H, W), weight
