"""Checkpoints in the reference's two formats (the file-format half): PLY point clouds without `plyfile`, and the trainer's
`.pt` model files (`save_model_pt` / `load_model_pt` / `activate`, below; key layout PINNED by tests/golden/model_golden.pt, which was written from the
reference's own GaussianModel instances).

`save_ply` / `load_ply` follow `GaussianModel.save_ply` / `load_ply` (easyvolcap/utils/gaussian2d_utils.py:918-1000): one `vertex` element of float32
properties  x y z nx ny nz f_dc_0..2 f_rest_0..(3K-4) opacity scale_0..1 rot_0..3  (K = (sh_degree + 1)^2), binary little-endian, features stored
channel-major (`features.transpose(1, 2).flatten(1)`), raw (pre-activation) opacity / scaling / rotation.  The reader accepts any property order
and extra properties, as the reference's does (it looks properties up by name).  Host-side I/O: numpy only.
PLY CONTENT PINNED (tests/golden/ply_golden.npz, tests/golden/make_ply_golden.py): the reference's own save_ply was run and the structured array
it hands to `plyfile` -- field names, order, dtype, every value -- recorded; its load_ply was run on a file written by `save_ply` below and returned
the saved parameters.  What stays from the PLY 1.0 specification rather than from a reference-written file is only plyfile's header TEXT
(`format binary_little_endian 1.0`, one `property float <name>` line per field, no comments): plyfile is not installed here."""
import numpy as np
import torch


def attribute_names(n_dc=3, n_rest=45, n_scale=2, n_rot=4):
    """construct_list_of_attributes, gaussian2d_utils.py:918-929."""
    return (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(n_dc)] + ["f_rest_%d" % i for i in range(n_rest)] + ["opacity"] +
            ["scale_%d" % i for i in range(n_scale)] + ["rot_%d" % i for i in range(n_rot)])


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, bounds=None):
    """xyz (P,3), features_dc (P,1,3), features_rest (P,K-1,3), opacity (P,1), scaling (P,2), rotation (P,4): the model's raw parameters.
    bounds = (min xyz, max xyz) keeps only the points inside, like the reference."""
    t = lambda a: a.detach().cpu().float()
    xyz = t(xyz)
    mask = torch.ones(xyz.shape[0], dtype=torch.bool)
    if bounds is not None:
        mask = ((xyz >= t(bounds[0])) & (xyz <= t(bounds[1]))).all(dim=-1)
    f_dc = t(features_dc)[mask].transpose(1, 2).flatten(start_dim=1)
    f_rest = t(features_rest)[mask].transpose(1, 2).flatten(start_dim=1)
    cols = [xyz[mask], torch.zeros_like(xyz[mask]), f_dc, f_rest, t(opacity)[mask], t(scaling)[mask], t(rotation)[mask]]
    table = torch.cat(cols, dim=1).numpy().astype("<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0] + "".join("property float %s\n" % n for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_vertex_table(path):
    """The `vertex` element of a binary or ascii PLY as a numpy structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, n, props, in_vertex, first = None, 0, [], False, True
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header of %s is not terminated" % path)
            w = line.decode("ascii").split()
            if not w or w[0] == "comment":
                continue
            if w[0] == "format":
                fmt = w[1]
            elif w[0] == "element":
                in_vertex = w[1] == "vertex"
                if in_vertex:
                    if not first:
                        raise ValueError("the vertex element must come first")
                    n = int(w[2])
                first = False
            elif w[0] == "property" and in_vertex:
                if w[1] == "list":
                    raise ValueError("list properties are not supported on vertices")
                props.append((w[2], _PLY_TYPES[w[1]]))
            elif w[0] == "end_header":
                break
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(n)]
            out = np.empty(n, dtype=[(nm, "<" + ty if ty[1] != "1" else ty) for nm, ty in props])
            for j, (nm, ty) in enumerate(props):
                out[nm] = [float(r[j]) for r in rows]
            return out
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, (end + ty) if ty[1] != "1" else ty) for nm, ty in props])
        return np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)


def load_ply(path, max_sh_degree=3, device="cpu"):
    """-> dict(xyz (P,3), features_dc (P,1,3), features_rest (P,K-1,3), opacity (P,1), scaling (P,S), rotation (P,4)) float32 tensors,
    with the reference's checks (gaussian2d_utils.py:963-1000)."""
    v = read_vertex_table(path)
    names = v.dtype.names
    col = lambda nm: np.asarray(v[nm], dtype=np.float32)
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1)[:, :, None]              # (P,3,1)
    by_index = lambda prefix: sorted([nm for nm in names if nm.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
    rest_names = by_index("f_rest_")
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError("%d f_rest properties, expected %d for SH degree %d" % (len(rest_names), 3 * (max_sh_degree + 1) ** 2 - 3, max_sh_degree))
    rest = np.stack([col(nm) for nm in rest_names], axis=1).reshape(xyz.shape[0], 3, (max_sh_degree + 1) ** 2 - 1) if rest_names else np.zeros((xyz.shape[0], 3, 0), np.float32)
    scales = np.stack([col(nm) for nm in by_index("scale_")], axis=1)
    rots = np.stack([col(nm) for nm in by_index("rot")], axis=1)
    tt = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return dict(xyz=tt(xyz), features_dc=tt(dc).transpose(1, 2).contiguous(), features_rest=tt(rest).transpose(1, 2).contiguous(),
                opacity=tt(col("opacity")[:, None]), scaling=tt(scales), rotation=tt(rots))


# ---- `.pt` model checkpoints -----------------------------------------------------------------------------------------------------------------
# The reference's trainer writes torch.save({'model': model.state_dict(), 'epoch': e, ['optimizer', 'scheduler', 'moderator']})
# (easyvolcap/utils/net_utils.py:486-504).  The EnvGS sampler owns two GaussianModel modules, `sampler.pcd` (base set) and `sampler.env`
# (environment set) (models/samplers/gaussian2d_sampler.py:148, envgs_sampler.py:165; the optimizer prefixes 'sampler.pcd.' / 'sampler.env.' of
# envgs_sampler.py:231,344), each with the raw parameters of gaussian2d_utils.py:453-467 and the buffers of :294,309-312.
PT_PARAMS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_specular", "_roughness")
PT_BUFFERS = ("active_sh_degree", "max_radii2D", "xyz_gradient_accum", "denom", "xyz_weight_accum")
PT_SETS = {"pcd": "sampler.pcd.", "env": "sampler.env."}


def save_model_pt(path, sets, epoch=-1, extra=None):
    """sets = {"pcd": {...}, "env": {...}}: raw (pre-activation) tensors keyed by PT_PARAMS / PT_BUFFERS names.  Missing buffers are written as
    the zeros the reference's constructor registers; `active_sh_degree` defaults to the degree the feature count implies."""
    model = {}
    for name, d in sets.items():
        prefix = PT_SETS[name]
        P = d["_xyz"].shape[0]
        for k in PT_PARAMS:
            if k in d:
                model[prefix + k] = d[k].detach().cpu().contiguous()
        deg = int(round((d["_features_rest"].shape[1] + 1) ** 0.5)) - 1
        defaults = {"active_sh_degree": torch.full((1,), deg, dtype=torch.long), "max_radii2D": torch.zeros(P),
                    "xyz_gradient_accum": torch.zeros(P, 1), "denom": torch.zeros(P, 1), "xyz_weight_accum": torch.zeros(P, 1)}
        for k in PT_BUFFERS:
            model[prefix + k] = (d[k].detach().cpu() if k in d else defaults[k])
    blob = {"model": model, "epoch": int(epoch)}
    if extra:
        blob.update(extra)
    torch.save(blob, path)


def load_model_pt(path, device="cpu"):
    """-> ({"pcd": {...}, "env": {...}}, epoch).  Accepts the reference's file: keys outside the two Gaussian sets (networks, optimizer, moderator)
    are ignored, a DDP 'module.' prefix is dropped (net_utils.py:493 notes the incorrect naming), a file without an environment set (a plain
    2DGS run) returns only "pcd"."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    model = blob["model"] if "model" in blob else blob
    sets = {}
    for key, val in model.items():
        if key.startswith("module."):
            key = key[len("module."):]
        for name, prefix in PT_SETS.items():
            if key.startswith(prefix) and key[len(prefix):] in PT_PARAMS + PT_BUFFERS:
                sets.setdefault(name, {})[key[len(prefix):]] = val.to(device)
    for name, d in sets.items():
        missing = [k for k in PT_PARAMS[:6] if k not in d]
        if missing:
            raise KeyError("checkpoint %s: set '%s' lacks %s" % (path, name, missing))
        P = d["_xyz"].shape[0]
        for k in PT_PARAMS:
            if k in d and d[k].shape[0] != P:
                raise ValueError("checkpoint %s: %s%s has %d rows, _xyz has %d" % (path, PT_SETS[name], k, d[k].shape[0], P))
    return sets, int(blob.get("epoch", -1)) if isinstance(blob, dict) else -1


def activate(raw):
    """Raw parameters -> the renderer's inputs, with the reference's activations (gaussian2d_utils.py:330-352 setup_functions, :363-390 getters):
    exp scaling, normalised rotation, sigmoid opacity / specular / roughness, features = cat(dc, rest)."""
    out = dict(means3D=raw["_xyz"], shs=torch.cat([raw["_features_dc"], raw["_features_rest"]], dim=1).contiguous(),
               scales=torch.exp(raw["_scaling"]), rotations=torch.nn.functional.normalize(raw["_rotation"], dim=-1),
               opacities=torch.sigmoid(raw["_opacity"]))
    if "_specular" in raw:
        out["specular"] = torch.sigmoid(raw["_specular"])
    if "_roughness" in raw:
        out["roughness"] = torch.sigmoid(raw["_roughness"])
    return out
