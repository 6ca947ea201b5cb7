"""GPU: structural invariants of the acceleration structure (csrc/trace_bvh.hip), read back and walked on the host: the binary LBVH is a
tree over exactly the P surfels whose stored child boxes are the unions of the leaf boxes below them, and the 4-wide nodes the packet
traversal walks hold exactly the grandchildren of each binary node.  (That rays find the right surfels is what tests/test_trace_parity.py
checks; this pins the structure itself, including the degenerate sizes.)"""
import numpy as np
import pytest
import torch

from envgs_amd import synth, tracing

pytestmark = pytest.mark.gpu


def _leaf_boxes(v):
    q = v.reshape(-1, 4, 3)
    return q.min(axis=1), q.max(axis=1)


def _build(P, seed, clustered=False):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(P, 3, generator=g) * 2 - 1
    if clustered:
        xyz[: P // 2] = xyz[0]                                   # identical Morton codes: the id tie-break must still give a tree
    scales = torch.rand(P, 2, generator=g) * 0.05 + 0.005
    q = torch.randn(P, 4, generator=g); q = q / q.norm(dim=-1, keepdim=True)
    v, f = synth.get_disks(xyz, scales, q)
    nodes, n = tracing.build_bvh(v.cuda())
    torch.cuda.synchronize()
    assert n == P
    return nodes.cpu().numpy(), v.numpy()


@pytest.mark.parametrize("P,clustered", [(1, False), (2, False), (3, False), (5, False), (64, False), (1000, False), (777, True),
                                         (12000, True), (40000, True)])      # half of the surfels on ONE Morton code: a 6 000- / 20 000-entry bucket of the key sort (LDS / chunked)
def test_binary_and_wide_nodes(P, clustered):
    flat, v = _build(P, 11 + P, clustered)
    ni = max(P - 1, 1)
    assert flat.size == 48 * ni
    nodes = flat[:16 * ni].reshape(ni, 16)
    wide = flat[16 * ni:].reshape(ni, 32)
    llo, lhi = _leaf_boxes(v)
    refs = nodes[:, 12:14].copy().view(np.int32)
    boxes = nodes[:, :12].reshape(ni, 2, 2, 3)                       # node, side, lo/hi, xyz
    # --- binary tree: every surfel exactly once, every internal node (but the root) exactly once, boxes = unions of the leaves below
    seen_leaf = np.zeros(P, np.int32); seen_node = np.zeros(ni, np.int32)
    sub = {}

    def walk(i):
        seen_node[i] += 1
        lo = np.full(3, np.inf, np.float32); hi = np.full(3, -np.inf, np.float32)
        for side in range(2):
            c = int(refs[i, side])
            if P == 1 and side == 1:
                continue                                             # the single-surfel tree: the right child is an unreachable far point
            if c < 0:
                sid = ~c
                seen_leaf[sid] += 1
                # a leaf box is the quad's box, padded outwards by a few ulps-of-extent (conservative slab tests)
                clo, chi = boxes[i, side, 0], boxes[i, side, 1]
                tol = 1e-4 * (np.abs(llo[sid]) + np.abs(lhi[sid]) + (lhi[sid] - llo[sid])) + 1e-7
                assert (clo <= llo[sid]).all() and (chi >= lhi[sid]).all() and (llo[sid] - clo <= tol).all() and (chi - lhi[sid] <= tol).all()
            else:
                clo, chi = walk(c)                                   # an inner box is EXACTLY the union of the stored boxes below it
                assert np.array_equal(boxes[i, side, 0], clo) and np.array_equal(boxes[i, side, 1], chi)
            lo = np.minimum(lo, clo); hi = np.maximum(hi, chi)
        sub[i] = (lo, hi)
        return lo, hi

    import sys
    sys.setrecursionlimit(10000)
    walk(0)
    assert (seen_leaf == 1).all() and (seen_node == 1).all()
    # --- wide nodes: slot boxes / refs are the grandchildren (a leaf child stays), unused slots are far-away points
    w8 = wide.reshape(ni, 4, 8)                                     # per slot: lo.x hi.x lo.y hi.y lo.z hi.z ref 0
    wl = np.stack([w8[:, :, 0], w8[:, :, 2], w8[:, :, 4]], axis=1)   # (node, axis, slot)
    wh = np.stack([w8[:, :, 1], w8[:, :, 3], w8[:, :, 5]], axis=1)
    wr = np.ascontiguousarray(w8[:, :, 6]).view(np.int32)
    assert (w8[:, :, 7] == 0).all()
    for i in range(ni):
        exp = []
        for side in range(2):
            c = int(refs[i, side])
            if c < 0:
                exp.append((boxes[i, side, 0], boxes[i, side, 1], c))
            else:
                for s2 in range(2):
                    exp.append((boxes[c, s2, 0], boxes[c, s2, 1], int(refs[c, s2])))
        for k in range(4):
            if k < len(exp):
                assert np.array_equal(wl[i, :, k], exp[k][0]) and np.array_equal(wh[i, :, k], exp[k][1]) and int(wr[i, k]) == exp[k][2]
            else:
                assert (wl[i, :, k] == np.float32(1e30)).all() and (wh[i, :, k] == np.float32(1e30)).all()
    # every surfel is reachable exactly once through the wide nodes too
    seen = np.zeros(P, np.int32)
    stack = [0]
    while stack:
        i = stack.pop()
        for k in range(4):
            if wl[i, 0, k] == np.float32(1e30) and wh[i, 0, k] == np.float32(1e30):
                continue                                             # unused slot (or the single-surfel tree's unreachable twin)
            c = int(wr[i, k])
            if c < 0:
                seen[~c] += 1
            else:
                stack.append(c)
    assert (seen == 1).all()
