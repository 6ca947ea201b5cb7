"""PLY checkpoints in the reference's layout (envgs_amd/ckpt.py): header, byte layout, round trip, tolerant reader."""
import numpy as np
import pytest
import torch

from envgs_amd import ckpt


def _model(P=37, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g), features_rest=torch.randn(P, 15, 3, generator=g),
                opacity=torch.randn(P, 1, generator=g), scaling=torch.randn(P, 2, generator=g), rotation=torch.randn(P, 4, generator=g))


def test_ply_layout_and_round_trip(tmp_path):
    m = _model()
    path = str(tmp_path / "gs.ply")
    ckpt.save_ply(path, **m)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:] if l]
    assert names == ckpt.attribute_names() and len(names) == 61 and all(l.startswith("property float ") for l in lines[3:] if l)
    tab = np.frombuffer(body, dtype="<f4").reshape(37, 61)
    assert np.array_equal(tab[:, 0:3], m["xyz"].numpy()) and not tab[:, 3:6].any()                              # normals are zeros
    assert np.array_equal(tab[:, 6:9], m["features_dc"].transpose(1, 2).flatten(1).numpy())                     # channel-major features
    assert np.array_equal(tab[:, 9:54], m["features_rest"].transpose(1, 2).flatten(1).numpy())
    assert np.array_equal(tab[:, 54:55], m["opacity"].numpy()) and np.array_equal(tab[:, 55:57], m["scaling"].numpy()) and np.array_equal(tab[:, 57:61], m["rotation"].numpy())
    back = ckpt.load_ply(path, max_sh_degree=3)
    for k, v in m.items():
        assert back[k].shape == v.shape and torch.equal(back[k], v), k


def test_ply_bounds_and_degree_check(tmp_path):
    m = _model(P=200, seed=1)
    path = str(tmp_path / "b.ply")
    ckpt.save_ply(path, bounds=(torch.tensor([-0.5, -0.5, -0.5]), torch.tensor([0.5, 0.5, 0.5])), **m)
    back = ckpt.load_ply(path)
    keep = ((m["xyz"] >= -0.5) & (m["xyz"] <= 0.5)).all(-1)
    assert 0 < int(keep.sum()) < 200 and torch.equal(back["xyz"], m["xyz"][keep]) and torch.equal(back["rotation"], m["rotation"][keep])
    with pytest.raises(ValueError):
        ckpt.load_ply(path, max_sh_degree=2)                                                                    # 45 f_rest properties != 24


def test_ply_reader_tolerates_order_extras_and_ascii(tmp_path):
    m = _model(P=5, seed=2)
    names = ckpt.attribute_names()
    cols = torch.cat([m["xyz"], torch.zeros(5, 3), m["features_dc"].transpose(1, 2).flatten(1), m["features_rest"].transpose(1, 2).flatten(1), m["opacity"],
                      m["scaling"], m["rotation"]], 1).numpy()
    order = list(reversed(range(len(names))))                                                                   # properties in reverse order + an extra one
    path = str(tmp_path / "r.ply")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\ncomment made by a test\nelement vertex 5\nproperty uchar flag\n" +
                 "".join("property float %s\n" % names[j] for j in order) + "end_header\n").encode())
        for i in range(5):
            f.write(np.uint8(7).tobytes() + cols[i, order].astype("<f4").tobytes())
    back = ckpt.load_ply(path)
    for k, v in m.items():
        assert torch.equal(back[k], v), k
    apath = str(tmp_path / "a.ply")
    with open(apath, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 5\n" + "".join("property float %s\n" % n for n in names) + "end_header\n")
        for i in range(5):
            f.write(" ".join(repr(float(x)) for x in cols[i]) + "\n")
    back = ckpt.load_ply(apath)
    for k, v in m.items():
        assert torch.allclose(back[k], v, rtol=0, atol=0), k
