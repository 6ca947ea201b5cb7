"""PLY point-cloud checkpoints in the reference's layout (the file-format half), without `plyfile`.

`save_ply` / `load_ply` follow `GaussianModel.save_ply` / `load_ply` (easyvolcap/utils/gaussian2d_utils.py:918-1000): one `vertex` element of float32
properties  x y z nx ny nz f_dc_0..2 f_rest_0..(3K-4) opacity scale_0..1 rot_0..3  (K = (sh_degree + 1)^2), binary little-endian, features stored
channel-major (`features.transpose(1, 2).flatten(1)`), raw (pre-activation) opacity / scaling / rotation.  The reader accepts any property order
and extra properties, as the reference's does (it looks properties up by name).  Host-side I/O: numpy only.
"PLY layout unpinned": the header text plyfile would write is restated from the PLY specification, not diffed against a file the reference wrote."""
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
