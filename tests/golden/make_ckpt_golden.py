"""Generate tests/golden/model_golden.pt by INSTANTIATING the reference's GaussianModel (authoring container only; /root/reference does not
exist on the GPU box): two small models, base (render_reflection) and environment, hung under `sampler.pcd` / `sampler.env` of a bare
torch.nn.Module the way Gaussian2DSampler / EnvGSSampler own them (gaussian2d_sampler.py:148, envgs_sampler.py:165), and written exactly like
net_utils.save_model does: torch.save({'model': state_dict, 'epoch': e}).  The fixture is data: the reference's own key names, shapes, dtypes
and initial values, plus the activated tensors its getters return.  Re-run:  python tests/golden/make_ckpt_golden.py"""
import json
import os
import sys
from unittest.mock import MagicMock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for m in ("pdbr", "pdbr.utils", "ruamel", "ruamel.yaml", "plyfile", "diff_surfel_tracing"):
        sys.modules[m] = MagicMock()
    sys.modules["ujson"] = json
    sys.path.insert(0, "/root/reference")
    from easyvolcap.utils import gaussian2d_utils as g2d
    torch.manual_seed(0)
    mk = lambda P, refl, deg: g2d.GaussianModel(xyz=torch.rand(P, 3) * 2 - 1, colors=torch.rand(P, 3), init_occ=0.1, init_scale=torch.rand(P, 2) * 0.05 + 0.01,
                                                sh_degree=3, init_sh_degree=deg, render_reflection=refl, xyz_lr_scheduler=None)
    root = torch.nn.Module()
    root.sampler = torch.nn.Module()
    root.sampler.pcd = mk(24, True, 0)
    root.sampler.env = mk(16, False, 3)
    with torch.no_grad():                                         # not the constant initial values: something every activation changes
        for m in (root.sampler.pcd, root.sampler.env):
            for p in m.parameters():
                if p.requires_grad:
                    p.add_(torch.randn_like(p) * 0.3)
    sd = {k: v.detach().clone() for k, v in root.state_dict().items()}
    act = {}
    for name, m in (("pcd", root.sampler.pcd), ("env", root.sampler.env)):
        act[name] = {"means3D": m.get_xyz, "shs": m.get_features, "scales": m.get_scaling, "rotations": m.get_rotation, "opacities": m.get_opacity,
                     "specular": m.get_specular, "roughness": m.get_roughness}
        act[name] = {k: v.detach().clone() for k, v in act[name].items()}
    torch.save({"model": sd, "epoch": 7, "activated": act}, os.path.join(HERE, "model_golden.pt"))
    for k, v in sd.items():
        print(k, tuple(v.shape), v.dtype)


if __name__ == "__main__":
    main()
