"""Densify / prune support on the per-Gaussian SoA (include/envgs_densify.h).

`prune_rows` = every `tensor[mask]` of the reference's pruning (`_prune_optimizer` / `prune_stats`,
easyvolcap/utils/gaussian2d_utils.py:536-560,640-648) in one scan + one gather launch; `prune_optimizer` / `cat_tensors_to_optimizer`
keep the reference's contract (one parameter per group, fresh `nn.Parameter`, Adam moments carried over / zero-extended);
`knn3_mean_dist2` = `simple_knn.distCUDA2` (gaussian2d_utils.py:432-440).
"""
import torch
from torch import nn

from . import _lib


def _stream(dev):
    return _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def prune_rows(tensors, keep):
    """[t[keep] for t in tensors] for tensors sharing their first dimension; keep: (P,) bool.  One host sync (the kept count)."""
    lib = _lib.load()
    if keep.device.type != "cuda":
        raise RuntimeError("prune_rows needs GPU tensors; there is no CPU path")
    dev = keep.device
    P = keep.shape[0]
    keep8 = keep.to(torch.uint8).contiguous()
    pos = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev)
    tb = lib.envgs_compact_temp_bytes(P)
    temp = torch.empty(max(tb, 1), dtype=torch.uint8, device=dev)
    p = _lib.ptr
    _lib.check(lib.envgs_compact_scan(P, p(keep8), p(pos), p(nk), p(temp), tb, _stream(dev)), "envgs_compact_scan")
    n = int(nk.item()) & 0xFFFFFFFF
    outs, srcs = [], []
    for t in tensors:
        if t.shape[0] != P or t.device != dev:
            raise RuntimeError("prune_rows: every tensor must have %d rows on %s" % (P, dev))
        if t.element_size() * (t[0].numel() if P else 1) % 4:
            raise RuntimeError("prune_rows: rows must be a multiple of 4 bytes")
        srcs.append(t.detach().contiguous())
        outs.append(torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev))
    if n == 0:
        return outs
    for i in range(0, len(srcs), 32):
        chunk = list(zip(srcs[i:i + 32], outs[i:i + 32]))
        arr = (_lib.RowsTensor * len(chunk))()
        for j, (s, o) in enumerate(chunk):
            arr[j] = _lib.RowsTensor(s.data_ptr(), o.data_ptr(), s.element_size() * (s[0].numel() if P else 0))
        _lib.check(lib.envgs_compact_gather(len(chunk), arr, P, p(keep8), p(pos), _stream(dev)), "envgs_compact_gather")
    return outs


def prune_optimizer(optimizer, keep):
    """`_prune_optimizer` of the reference for an optimizer whose groups hold ONE per-Gaussian parameter each: every parameter and its
    Adam moments are compacted together; returns {group name: new nn.Parameter} (state moved to the new parameter)."""
    items = []
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        prm = group["params"][0]
        st = optimizer.state.get(prm, None)
        items.append((group, prm, st))
    flat = []
    for group, prm, st in items:
        flat.append(prm.data)
        if st is not None and "exp_avg" in st:
            flat += [st["exp_avg"], st["exp_avg_sq"]]
    outs = iter(prune_rows(flat, keep))
    result = {}
    for group, prm, st in items:
        new = nn.Parameter(next(outs).requires_grad_(True))
        if st is not None:
            if "exp_avg" in st:
                st["exp_avg"] = next(outs); st["exp_avg_sq"] = next(outs)
            del optimizer.state[prm]
            optimizer.state[new] = st
        group["params"][0] = new
        result[group.get("name", str(len(result)))] = new
    return result


def cat_tensors_to_optimizer(tensors_dict, optimizer):
    """`cat_tensors_to_optimizer` of the reference (gaussian2d_utils.py:562-588): append rows to each named group, zero moments."""
    result = {}
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        ext = tensors_dict.get(group.get("name"))
        if ext is None:
            continue
        prm = group["params"][0]
        st = optimizer.state.get(prm, None)
        new = nn.Parameter(torch.cat((prm.data, ext), dim=0).requires_grad_(True))
        if st is not None:
            if "exp_avg" in st:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del optimizer.state[prm]
            optimizer.state[new] = st
        group["params"][0] = new
        result[group["name"]] = new
    return result


def knn3_mean_dist2(xyz):
    """simple_knn.distCUDA2: mean squared distance to the 3 nearest neighbours, (P,) float32."""
    lib = _lib.load()
    if xyz.device.type != "cuda":
        raise RuntimeError("knn3_mean_dist2 needs a GPU tensor; there is no CPU path")
    x = xyz.detach().to(torch.float32).contiguous()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(lib.envgs_knn3_mean_dist2(x.shape[0], _lib.ptr(x), _lib.ptr(out), _stream(x.device)), "envgs_knn3_mean_dist2")
    return out


def torch_rows(tensors, keep):
    """`prune_rows` written with torch indexing.  NOT a product path: tests pass it as `row_ops` to exercise `SurfelSet`'s host logic on CPU."""
    return [t.detach()[keep] for t in tensors]


class SurfelSet:
    """One Gaussian set's raw parameters, their optimizer groups and the densification statistics, with the densify / prune schedule of the
    reference's `GaussianModel` (easyvolcap/utils/gaussian2d_utils.py:622-909) rebuilt over the compaction kernels:

      * every removal is ONE `prune_rows` call over parameters + both Adam moments + the four statistics (the reference: `_prune_optimizer`
        :536-560 + `prune_stats` :640-648, 28 boolean-mask gathers);
      * every clone / split gathers all selected rows of all parameters in one call and appends them with one `torch.cat` per tensor;
      * selection masks are the reference's expressions (cited per method); random split offsets use the same `torch.normal(means, stds)` call.

    raw: {"_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"[, "_specular", "_roughness"]} (ckpt.PT_PARAMS).
    optimizer: groups named `prefix + name`, one parameter each (the reference's layout, gaussian2d_utils.py:562-588); may be None.
    row_ops: the row gather; defaults to the HIP `prune_rows` (GPU tensors only)."""

    STATS = ("xyz_gradient_accum", "denom", "max_radii2D", "xyz_weight_accum")

    def __init__(self, raw, optimizer=None, prefix="", spatial_scale=1.0, max_gs=None, max_gs_threshold=1.0, row_ops=None, generator=None):
        self.names = [k for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_specular", "_roughness") if k in raw]
        self.optimizer, self.prefix, self.spatial_scale = optimizer, prefix, float(spatial_scale)
        self.max_gs, self.max_gs_threshold = max_gs, max_gs_threshold
        self.rows = row_ops or prune_rows
        # split offsets are random: with data parallelism every rank must draw the SAME ones, i.e. pass generators seeded
        # identically on all ranks (or seed the global generator identically before every densification)
        self.generator = generator
        self.p = {}
        groups = {g.get("name"): g for g in optimizer.param_groups} if optimizer is not None else {}
        for k in self.names:
            g = groups.get(prefix + k)
            self.p[k] = g["params"][0] if g is not None else nn.Parameter(raw[k].detach().clone().requires_grad_(True))
        self.reset_stats()
        self.log = []                                # (event, count) pairs, the numbers the reference prints

    # ---- views -------------------------------------------------------------------------------------------------------------------------------
    @property
    def number(self):
        return self.p["_xyz"].shape[0]

    @property
    def device(self):
        return self.p["_xyz"].device

    def scaling(self):
        return torch.exp(self.p["_scaling"].detach())

    def opacity(self):
        return torch.sigmoid(self.p["_opacity"].detach())

    def raw(self):
        return {k: v.detach() for k, v in self.p.items()}

    # ---- statistics (gaussian2d_utils.py:622-637, 901-909) --------------------------------------------------------------------------------------
    def reset_stats(self):
        P, dev = self.number, self.device
        self.stats = {"xyz_gradient_accum": torch.zeros(P, 1, device=dev), "denom": torch.zeros(P, 1, device=dev),
                      "max_radii2D": torch.zeros(P, device=dev), "xyz_weight_accum": torch.zeros(P, 1, device=dev)}

    def add_densification_stats(self, viewspace_grad, update_filter, weight_accumulate=None, radii=None):
        """viewspace_grad: the `.grad` of the rasterizer's means2D (P,3); update_filter: (P,) bool; radii (optional): the max-radius update the
        sampler does next to it (gaussian2d_sampler.py:330-332)."""
        s = self.stats
        s["denom"][update_filter] += 1
        s["xyz_gradient_accum"][update_filter] += torch.norm(viewspace_grad[update_filter], dim=-1, keepdim=True)
        if weight_accumulate is not None:
            s["xyz_weight_accum"][update_filter] += weight_accumulate[update_filter]
        if radii is not None:
            s["max_radii2D"][update_filter] = torch.max(s["max_radii2D"][update_filter], radii[update_filter].to(s["max_radii2D"].dtype))

    def _avg(self, key):
        avg = self.stats[key] / self.stats["denom"]
        avg[avg.isnan()] = 0.0
        return avg

    def gradient_avg(self):
        return self._avg("xyz_gradient_accum")

    def weight_avg(self):
        return self._avg("xyz_weight_accum")

    # ---- row surgery ---------------------------------------------------------------------------------------------------------------------------
    def _groups(self):
        """[(name, group or None, parameter, adam state or None)] in parameter order."""
        out = []
        gs = {g.get("name"): g for g in self.optimizer.param_groups} if self.optimizer is not None else {}
        for k in self.names:
            g = gs.get(self.prefix + k)
            prm = self.p[k]
            st = self.optimizer.state.get(prm, None) if g is not None else None
            out.append((k, g, prm, st if st else None))
        return out

    def _install(self, k, g, old, st, new_data, m=None, v=None):
        new = nn.Parameter(new_data.requires_grad_(True))
        if g is not None:
            if st is not None:
                if m is not None:
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                del self.optimizer.state[old]
                self.optimizer.state[new] = st
            g["params"][0] = new
        self.p[k] = new

    def remove(self, mask):
        """prune_points + prune_stats (:553-560, :640-648): drop the rows where `mask` is set -- parameters, Adam moments and statistics, one gather."""
        keep = ~mask
        items = self._groups()
        flat = []
        for k, g, prm, st in items:
            flat.append(prm.data)
            if st is not None and "exp_avg" in st:
                flat += [st["exp_avg"], st["exp_avg_sq"]]
        flat += [self.stats[s] for s in self.STATS]
        outs = iter(self.rows(flat, keep))
        for k, g, prm, st in items:
            data = next(outs)
            if st is not None and "exp_avg" in st:
                self._install(k, g, prm, st, data, next(outs), next(outs))
            else:
                self._install(k, g, prm, st, data)
        for s in self.STATS:
            self.stats[s] = next(outs)
        self._scene_changed()

    @staticmethod
    def _scene_changed():
        """Rows were removed / appended or the opacities reset: a tracer's stored topology no longer describes this set even when the count
        happens to be what it was -- its next request is a full build, not a refit (tracing.invalidate_all_structures)."""
        from . import tracing
        tracing.invalidate_all_structures()

    def _append(self, new, selected_stats, split, ratio):
        """densification_postfix + densify_stats (:590-621, :650-663): new rows get zero moments; their statistics are the parents' (gradient and
        radius scaled by `ratio`, weight multiplied by the current maximum -- the reference's expression, kept as is)."""
        for k, g, prm, st in self._groups():
            ext = new[k]
            if st is not None and "exp_avg" in st:
                m = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                v = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                self._install(k, g, prm, st, torch.cat((prm.data, ext), dim=0), m, v)
            else:
                self._install(k, g, prm, st, torch.cat((prm.data, ext), dim=0))
        ga, dn, mr, wa = selected_stats
        s = self.stats
        wmax = s["xyz_weight_accum"].max()
        s["xyz_gradient_accum"] = torch.cat([s["xyz_gradient_accum"], ga.repeat(split, 1) * ratio], dim=0)
        s["denom"] = torch.cat([s["denom"], dn.repeat(split, 1)], dim=0)
        s["max_radii2D"] = torch.cat([s["max_radii2D"], mr.repeat(split) * ratio], dim=0)
        s["xyz_weight_accum"] = torch.cat([s["xyz_weight_accum"], wa.repeat(split, 1) * wmax], dim=0)
        self._scene_changed()

    def _selected(self, mask):
        outs = self.rows([self.p[k].data for k in self.names] + [self.stats[s] for s in self.STATS], mask)
        return dict(zip(self.names, outs[:len(self.names)])), outs[len(self.names):]

    def clone(self, mask):
        """:665-677."""
        sel, st = self._selected(mask)
        self._append(sel, st, 1, 1.0)

    def split(self, mask, N=2, ratio=0.8):
        """:679-706: N children per selected surfel, offsets ~ N(0, diag(sx, sy, 0)) in the surfel's frame, scales / (ratio N); parents removed."""
        from .synth import build_rotation
        sel, st = self._selected(mask)
        scal = torch.exp(sel["_scaling"])
        stds = scal.repeat(N, 1)
        stds = torch.cat([stds, torch.zeros_like(stds[:, :1])], dim=-1)
        samples = torch.normal(torch.zeros_like(stds), stds, generator=self.generator)
        rots = build_rotation(sel["_rotation"]).repeat(N, 1, 1)
        new = {k: v.repeat(*([N] + [1] * (v.dim() - 1))) for k, v in sel.items()}
        new["_xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + sel["_xyz"].repeat(N, 1)
        new["_scaling"] = torch.log(scal.repeat(N, 1) / (ratio * N))
        self._append(new, st, N, 1.0 / (ratio * N))
        n_split = int(mask.sum().item())
        self.remove(torch.cat((mask, torch.zeros(n_split * N, device=mask.device, dtype=torch.bool))))

    def replace(self, name, tensor):
        """replace_tensor_to_optimizer (:517-534): new values, zeroed moments."""
        for k, g, prm, st in self._groups():
            if k == name:
                if st is not None and "exp_avg" in st:
                    self._install(k, g, prm, st, tensor.detach().clone(), torch.zeros_like(tensor), torch.zeros_like(tensor))
                else:
                    self._install(k, g, prm, st, tensor.detach().clone())

    def reset_opacity(self, value=0.01):
        """:512-515."""
        o = self.p["_opacity"].detach()
        self.replace("_opacity", torch.min(o, torch.logit(torch.ones_like(o) * value)))
        self._scene_changed()           # (leaf boxes are bounded by the alpha >= 1/255 disc: every box shrinks at once)

    def reset_specular(self, value=0.001, reset_all=False):
        """:505-510."""
        s = self.p["_specular"].detach()
        cap = torch.logit(torch.ones_like(s) * value)
        self.replace("_specular", cap if reset_all else torch.min(s, cap))

    # ---- the schedule (gaussian2d_utils.py:718-899) ------------------------------------------------------------------------------------------------
    def densify_and_clone(self, grad_threshold, size_threshold):
        high = (self.gradient_avg() >= grad_threshold).squeeze(-1)
        mask = (torch.max(self.scaling(), dim=1).values <= size_threshold * self.spatial_scale) & high
        n = int(mask.sum().item())
        self.log.append(("clone", n))
        if n > 0:
            self.clone(mask)

    def densify_and_split(self, grad_threshold, size_threshold, split_screen_threshold=None, N=2):
        high = (self.gradient_avg() >= grad_threshold).squeeze(-1)
        mask = torch.max(self.scaling(), dim=1).values > size_threshold * self.spatial_scale
        if split_screen_threshold is not None:
            mask = mask | (self.stats["max_radii2D"] > split_screen_threshold)
        mask = mask & high
        n = int(mask.sum().item())
        self.log.append(("split", n))
        if n > 0:
            self.split(mask, N)

    def prune_min_opacity_and_gradients(self, min_opacity=None, min_gradient=None):
        P, dev = self.number, self.device
        occ = (self.opacity() < min_opacity).squeeze(-1) if min_opacity is not None else torch.zeros(P, dtype=torch.bool, device=dev)
        if min_gradient is not None:
            grd = ((self.gradient_avg() <= min_gradient) & (self.stats["denom"] != 0)).squeeze(-1)
        else:
            grd = torch.zeros(P, dtype=torch.bool, device=dev)
        mask = occ | grd
        n = int(mask.sum().item())
        self.log.append(("prune_occ_grad", n))
        if n > 0:
            self.remove(mask)

    def prune_max_scene_and_screen(self, max_scene_threshold=None, max_screen_threshold=None, min_weight_threshold=None):
        P, dev = self.number, self.device
        none = torch.zeros(P, dtype=torch.bool, device=dev)
        screens = self.stats["max_radii2D"] > max_screen_threshold if max_screen_threshold is not None else none
        scenes = torch.max(self.scaling(), dim=-1).values > self.spatial_scale * max_scene_threshold if max_scene_threshold is not None else none
        if min_weight_threshold is not None:
            w = self.weight_avg()
            light = (w < torch.quantile(w, min_weight_threshold)).squeeze(-1)
        else:
            light = torch.ones(P, dtype=torch.bool, device=dev)
        big = screens | scenes
        prune = big & light
        split = (big & ~light)[~prune]
        n_prune, n_split = int(prune.sum().item()), int(split.sum().item())
        self.log.append(("prune_large", n_prune)); self.log.append(("split_large", n_split))
        if n_prune > 0:
            self.remove(prune)
        if n_split > 0:
            self.split(split, 5, 0.5)

    def prune_visibility(self):
        n_prune = self.number - int(self.max_gs * self.max_gs_threshold)
        if n_prune > 0:
            _, idx = torch.topk(self.weight_avg()[..., 0], n_prune, largest=False)
            mask = torch.zeros(self.number, dtype=torch.bool, device=self.device)
            mask[idx] = True
            self.remove(mask)
            self.log.append(("prune_visibility", n_prune))

    def densify_and_prune(self, min_opacity, min_gradient, densify_grad_threshold, densify_size_threshold, split_screen_threshold=None,
                          max_scene_threshold=None, max_screen_threshold=None, min_weight_threshold=None, prune_visibility=False, prune_large_gs=False):
        """:866-899, the same order: clone, split, prune by opacity / gradient, [prune or split the oversized], [prune the least visible], reset."""
        self.densify_and_clone(densify_grad_threshold, densify_size_threshold)
        self.densify_and_split(densify_grad_threshold, densify_size_threshold, split_screen_threshold)
        self.prune_min_opacity_and_gradients(min_opacity, min_gradient)
        if prune_large_gs:
            self.prune_max_scene_and_screen(max_scene_threshold, max_screen_threshold, min_weight_threshold)
        if prune_visibility:
            self.prune_visibility()
        self.reset_stats()
