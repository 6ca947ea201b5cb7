"""Generate tests/golden/ply_golden.npz by RUNNING the reference's own GaussianModel.save_ply / load_ply (easyvolcap/utils/gaussian2d_utils.py:
935-1000) in the authoring container.  `plyfile` is not installed here, so the two calls the reference makes into it are RECORDED instead:

  save_ply : `PlyElement.describe(elements, 'vertex')` receives the numpy structured array the reference assembled -- property names, order,
             dtypes and every value (the channel-major flattening of the SH features, the zero normals, raw opacity / scaling / rotation).
             That array IS the file's content; what plyfile adds is the header text (PLY 1.0, binary_little_endian, one `property float
             <name>` line per field -- from the format's specification, the one thing this fixture cannot pin) and `elements.tobytes()`.
  load_ply : `PlyData.read(path)` is answered with envgs_amd.ckpt's own parse of a file envgs_amd.ckpt.save_ply wrote from the same parameters;
             the reference's load_ply then rebuilds its parameters from it, by property name.  They must equal what was saved.

The fixture is data: the seeded raw parameters, the recorded structured array (names + a (P, n_fields) float32 table), and the parameters the
reference's load_ply produced.  Consumer: tests/test_ckpt.py.  Re-run: python tests/golden/make_ply_golden.py"""
import json
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    for m in ("pdbr", "pdbr.utils", "ruamel", "ruamel.yaml", "diff_surfel_tracing"):
        sys.modules[m] = MagicMock()
    sys.modules["ujson"] = json
    rec = {}
    ply = types.ModuleType("plyfile")

    class PlyElement:
        def __init__(self, data, name):
            self.data, self.name = data, name
            self.properties = [types.SimpleNamespace(name=n) for n in data.dtype.names]

        @staticmethod
        def describe(elements, name):
            rec["describe"] = (elements.copy(), name)
            return PlyElement(elements, name)

        def __getitem__(self, k):
            return self.data[k]

    class PlyData:
        def __init__(self, elements):
            self.elements = elements

        def write(self, path):
            rec["write_path"] = path

        @staticmethod
        def read(path):
            from envgs_amd import ckpt
            return PlyData([PlyElement(ckpt.read_vertex_table(path), "vertex")])
    ply.PlyElement, ply.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = ply
    torch.Tensor.cuda = lambda self, *a, **k: self
    _tensor = torch.tensor
    torch.tensor = lambda *a, **k: _tensor(*a, **{kk: ("cpu" if kk == "device" else vv) for kk, vv in k.items()})     # load_ply: device="cuda"
    sys.path.insert(0, "/root/reference")
    from easyvolcap.utils import gaussian2d_utils as g2d
    g2d.PlyData, g2d.PlyElement = PlyData, PlyElement          # load_ply uses module-level names (save_ply imports them locally)
    torch.manual_seed(0)
    P = 29
    m = g2d.GaussianModel(xyz=torch.rand(P, 3) * 2 - 1, colors=torch.rand(P, 3), init_occ=0.1, init_scale=torch.rand(P, 2) * 0.05 + 0.01,
                          sh_degree=3, init_sh_degree=3, render_reflection=False, xyz_lr_scheduler=None)
    with torch.no_grad():
        for p in m.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.3)
    raw = {k: getattr(m, k).detach().clone() for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}
    tmp = tempfile.mkdtemp()
    m.save_ply(os.path.join(tmp, "ref", "gs.ply"))
    elements, name = rec["describe"]
    names = list(elements.dtype.names)
    assert name == "vertex" and all(elements.dtype[n] == np.dtype("f4") for n in names)
    table = np.stack([elements[n] for n in names], axis=1).astype(np.float32)
    # the reverse direction: the reference's load_ply reads a file written by envgs_amd.ckpt.save_ply
    from envgs_amd import ckpt
    ours = os.path.join(tmp, "ours.ply")
    ckpt.save_ply(ours, raw["_xyz"], raw["_features_dc"], raw["_features_rest"], raw["_opacity"], raw["_scaling"], raw["_rotation"])
    m2 = g2d.GaussianModel(xyz=torch.rand(P, 3), colors=torch.rand(P, 3), init_occ=0.1, init_scale=torch.rand(P, 2) * 0.05 + 0.01,
                           sh_degree=3, init_sh_degree=0, render_reflection=False, xyz_lr_scheduler=None)
    m2.load_ply(ours)
    loaded = {k: getattr(m2, k).detach().clone() for k in raw}
    for k in raw:
        assert torch.equal(loaded[k], raw[k]), k                 # the reference reads back exactly what was saved
    np.savez_compressed(os.path.join(HERE, "ply_golden.npz"), names=np.array(names), table=table, element_name=name,
                        **{"raw" + k: v.numpy() for k, v in raw.items()}, **{"loaded" + k: v.numpy() for k, v in loaded.items()},
                        active_sh_degree_after_load=int(m2.active_sh_degree.item()))
    print("fields:", len(names), names[:8], "...", names[-7:], "table", table.shape, "sh degree after load", int(m2.active_sh_degree.item()))


if __name__ == "__main__":
    main()
