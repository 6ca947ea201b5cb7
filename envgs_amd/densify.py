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
