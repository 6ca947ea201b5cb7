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


def _raw_set(P, K, seed, reflective):
    g = torch.Generator().manual_seed(seed)
    d = {"_xyz": torch.randn(P, 3, generator=g), "_features_dc": torch.randn(P, 1, 3, generator=g), "_features_rest": torch.randn(P, K - 1, 3, generator=g),
         "_scaling": torch.randn(P, 2, generator=g) - 4, "_rotation": torch.randn(P, 4, generator=g), "_opacity": torch.randn(P, 1, generator=g)}
    if reflective:
        d["_specular"] = torch.randn(P, 1, generator=g); d["_roughness"] = torch.zeros(P, 1)
    return d


def test_pt_checkpoint_round_trip_and_reference_key_layout(tmp_path):
    pcd, env = _raw_set(50, 16, 0, True), _raw_set(30, 16, 1, False)
    pcd["max_radii2D"] = torch.arange(50.0)
    p = str(tmp_path / "latest.pt")
    ckpt.save_model_pt(p, {"pcd": pcd, "env": env}, epoch=12)
    blob = torch.load(p, weights_only=True)
    assert blob["epoch"] == 12
    keys = set(blob["model"])
    # the names the reference's trainer would look up (net_utils.py:363-377 loads by state_dict key)
    assert {"sampler.pcd._xyz", "sampler.pcd._features_dc", "sampler.pcd._specular", "sampler.env._scaling", "sampler.env.active_sh_degree",
            "sampler.pcd.xyz_weight_accum", "sampler.pcd.denom"} <= keys and "sampler.env._specular" not in keys
    assert blob["model"]["sampler.env.active_sh_degree"].dtype == torch.long and int(blob["model"]["sampler.env.active_sh_degree"]) == 3
    sets, epoch = ckpt.load_model_pt(p)
    assert epoch == 12 and set(sets) == {"pcd", "env"}
    for name, src in (("pcd", pcd), ("env", env)):
        for k, v in src.items():
            assert torch.equal(sets[name][k], v), (name, k)
    assert torch.equal(sets["pcd"]["max_radii2D"], torch.arange(50.0)) and float(sets["env"]["denom"].abs().sum()) == 0


def test_pt_checkpoint_reads_a_trainer_style_file(tmp_path):
    """Extra entries (networks, optimizer state), a DDP prefix and a base-only model are all accepted; a truncated set is refused."""
    pcd = _raw_set(20, 16, 2, True)
    model = {"module.sampler.pcd." + k: v for k, v in pcd.items()}
    model["module.network.some_mlp.weight"] = torch.zeros(4, 4)
    p = str(tmp_path / "ddp.pt")
    torch.save({"model": model, "epoch": 3, "optimizer": {"state": {}, "param_groups": []}}, p)
    sets, epoch = ckpt.load_model_pt(p)
    assert epoch == 3 and set(sets) == {"pcd"} and torch.equal(sets["pcd"]["_rotation"], pcd["_rotation"])
    del model["module.sampler.pcd._opacity"]
    torch.save({"model": model, "epoch": 3}, p)
    with pytest.raises(KeyError):
        ckpt.load_model_pt(p)


def test_activate_matches_the_reference_getters():
    raw = _raw_set(40, 16, 5, True)
    a = ckpt.activate(raw)
    assert a["shs"].shape == (40, 16, 3) and torch.equal(a["shs"][:, :1], raw["_features_dc"])
    assert torch.allclose(a["scales"], raw["_scaling"].exp()) and torch.allclose(a["rotations"].norm(dim=-1), torch.ones(40), atol=1e-6)
    assert torch.allclose(a["opacities"], torch.sigmoid(raw["_opacity"])) and torch.allclose(a["roughness"], torch.full((40, 1), 0.5))
    assert "specular" not in ckpt.activate(_raw_set(4, 16, 6, False))


def test_pt_checkpoint_against_the_reference_models_own_state_dict(tmp_path):
    """tests/golden/model_golden.pt was written from the reference's GaussianModel instances (tests/golden/make_ckpt_golden.py): its key names,
    shapes and dtypes are what a reference checkpoint holds, its `activated` entries what the reference's getters return."""
    import os
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_golden.pt")
    gold = torch.load(gold_path, weights_only=True)
    sets, epoch = ckpt.load_model_pt(gold_path)
    assert epoch == 7 and set(sets) == {"pcd", "env"}
    for name in ("pcd", "env"):
        assert set(sets[name]) == set(ckpt.PT_PARAMS + ckpt.PT_BUFFERS)                 # nothing of the reference's module is dropped
        for k, v in sets[name].items():
            assert torch.equal(v, gold["model"][ckpt.PT_SETS[name] + k])
        a = ckpt.activate(sets[name])
        assert set(a) == set(gold["activated"][name])
        for k, v in gold["activated"][name].items():
            assert a[k].shape == v.shape and torch.allclose(a[k], v, rtol=1e-6, atol=1e-7), (name, k)
    # the writer reproduces the reference's layout: same keys, shapes, dtypes and values
    p = str(tmp_path / "again.pt")
    ckpt.save_model_pt(p, sets, epoch=7)
    mine = torch.load(p, weights_only=True)["model"]
    assert set(mine) == set(gold["model"])
    for k, v in gold["model"].items():
        assert mine[k].dtype == v.dtype and mine[k].shape == v.shape and torch.equal(mine[k], v), k
    # and the defaults it invents for absent buffers are the reference constructor's
    bare = {k: v for k, v in sets["env"].items() if k in ckpt.PT_PARAMS}
    ckpt.save_model_pt(p, {"env": bare})
    mine = torch.load(p, weights_only=True)["model"]
    for k in ckpt.PT_BUFFERS:
        g = gold["model"]["sampler.env." + k]
        assert mine["sampler.env." + k].dtype == g.dtype and mine["sampler.env." + k].shape == g.shape, k
    assert int(mine["sampler.env.active_sh_degree"]) == 3


def test_ply_content_is_what_the_reference_hands_to_plyfile(tmp_path):
    """tests/golden/ply_golden.npz: the reference's own GaussianModel.save_ply was run (gaussian2d_utils.py:935-960) and the structured array it
    hands to plyfile -- field names, order, dtype, every value -- was recorded; its load_ply (:962-1000) was run on a file written by
    envgs_amd.ckpt.save_ply and gave back the saved parameters (asserted by the generator).  Here: save_ply writes exactly that table behind a
    PLY 1.0 binary_little_endian header with one `property float <name>` line per recorded field, and load_ply inverts it."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply_golden.npz"))
    raw = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("raw")}
    path = str(tmp_path / "golden.ply")
    ckpt.save_ply(path, raw["_xyz"], raw["_features_dc"], raw["_features_rest"], raw["_opacity"], raw["_scaling"], raw["_rotation"])
    names = [str(n) for n in g["names"]]
    data = open(path, "rb").read()
    head, body = data.split(b"end_header\n", 1)
    lines = head.decode("ascii").splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "element %s %d" % (str(g["element_name"]), g["table"].shape[0])
    assert lines[3:] == ["property float %s" % n for n in names]                    # the reference's construct_list_of_attributes, in its order
    assert body == np.ascontiguousarray(g["table"].astype("<f4")).tobytes()          # the bytes of the recorded structured array (all fields f4: no padding)
    tab = ckpt.read_vertex_table(path)
    assert list(tab.dtype.names) == names
    back = ckpt.load_ply(path, max_sh_degree=3)
    loaded = {k[6:]: g[k] for k in g.files if k.startswith("loaded")}
    for ours, ref in (("xyz", "_xyz"), ("features_dc", "_features_dc"), ("features_rest", "_features_rest"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                      ("rotation", "_rotation")):
        assert np.array_equal(back[ours].numpy(), loaded[ref]), ours                # what the reference's load_ply made of the same file
    assert int(g["active_sh_degree_after_load"]) == 3
