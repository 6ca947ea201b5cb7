"""The reference's SAMPLER code at the drop-in boundary (round-2 VERDICT item 2).

tests/golden/sampler_golden.pt records the extension calls that the reference's own EnvGSSampler.forward / Gaussian2DSampler.forward made
when they were run -- unchanged, on CPU, over recording stand-ins that forward to the oracle -- across the iteration schedule of
configs/models/envgs.yaml (tests/golden/make_sampler_golden.py; /root/reference is not read here).

CPU: the recorded sequence has the shape SURVEY.md 3.6 describes (raster only below iteration 3000, then raster + build + trace; P changing
     between calls; SH degree steps; filtered (1,S,3) rays with camera-sized settings; a second tracer with two bounces and others_precomp),
     and the drop-in packages' settings types accept exactly the recorded fields.
GPU: the whole sequence is REPLAYED through the HIP packages -- every recorded call with the very tensors the reference passed, in order, on
     two live SurfelTracer objects -- and what comes back equals what the oracle stand-ins returned to the reference (index work bit-exact,
     values within 1e-4 on the pixels / rays the audits call determined)."""
import os

import numpy as np
import pytest
import torch

from tests.util import check_close, record, record_fragile, FRAGILE_RAYS_MAX

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    return torch.load(os.path.join(HERE, "golden", "sampler_golden.pt"), weights_only=False)


def _calls(fx, step):
    a, b = step["call_range"]
    return [(fx["contract"][i], fx["call_tensors"][i]) for i in range(a, b)]


def test_recorded_schedule_has_the_expected_call_sequence(fx):
    steps = {s["tag"]: s for s in fx["steps"]}
    seq = lambda s: [(c["package"].replace("diff_surfel_", ""), c["what"]) for c, _ in _calls(fx, s)]
    RASTER = [("rasterization_wet_ch05", "settings"), ("rasterization_wet_ch05", "call")]
    TRACE = [("tracing", "settings"), ("tracing", "build_acceleration_structure"), ("tracing", "call")]     # optix_utils.py:104-119, :73-81, :188-201
    for s in fx["steps"]:
        if s["tag"].startswith("gaussian2d"):
            assert seq(s) == TRACE                                        # camera rays traced over the base set: no raster call at all
        elif s["iter"] < 3000:
            assert seq(s) == RASTER, s["tag"]                             # envgs_sampler.py:545: no reflection before render_reflection_start_iter
        else:
            assert seq(s) == RASTER + TRACE, s["tag"]
    # P changes between calls (densify / prune replace the parameters), for both sets
    assert steps["base_densify_prune"]["P_after"][0] != steps["base_densify_prune"]["P_before"][0]
    assert steps["env_and_base_densify"]["P_after"][1] != steps["env_and_base_densify"]["P_before"][1]
    # the SH degree buffers step (1-element int64 tensors handed over in the settings)
    assert steps["raster_only"]["sh_degree"] == (0, 0) and steps["sh_steps_and_normal_prop"]["sh_degree"] == (3, 1)
    for s in fx["steps"]:
        for c, t in _calls(fx, s):
            if c["what"] == "settings":
                f = c["fields"]["sh_degree"]
                assert f["kind"] == "tensor" and f["shape"] == [1] and f["dtype"] == "int64"
    # specular-filtered rays: a (1,S,3) tensor, S < H*W, while the settings still describe the camera (envgs_sampler.py:436-447)
    H, W = fx["H"], fx["W"]
    cs = _calls(fx, steps["specular_filtered_rays"])
    (tc, tt), = [(c, t) for c, t in cs if c["package"] == "diff_surfel_tracing" and c["what"] == "call"]
    (sc, _), = [(c, t) for c, t in cs if c["package"] == "diff_surfel_tracing" and c["what"] == "settings"]
    S = tc["args"][0]["shape"][1]
    assert tc["args"][0]["shape"] == [1, S, 3] and 0 < S < H * W and [o["shape"][:2] for o in tc["outputs"][:7]] == [[1, S]] * 7
    assert sc["fields"]["image_height"]["value"] == H and sc["fields"]["image_width"]["value"] == W
    assert int(steps["specular_filtered_rays"]["outputs"]["ref_msk"].sum()) == S
    # the other sampler: its own tracer object, two bounces, others_precomp, start_from_first
    cs = _calls(fx, steps["gaussian2d_traced_two_bounces"])
    (tc, tt), = [(c, t) for c, t in cs if c["what"] == "call"]
    (sc, _), = [(c, t) for c, t in cs if c["what"] == "settings"]
    assert tc["tracer"] == 1 and sc["fields"]["max_trace_depth"]["value"] == 2 and tc["kwargs"]["others_precomp"]["shape"][1] == 2
    assert tc["kwargs"]["start_from_first"]["value"] is True and tc["outputs"][6]["shape"][-1] == 48
    assert all(c["tracer"] == 0 for c, _ in _calls(fx, steps["envgs_after_the_other_tracer"]) if "tracer" in c)


def test_drop_in_settings_accept_the_recorded_fields(fx):
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    seen = set()
    for c, t in zip(fx["contract"], fx["call_tensors"]):
        if c["what"] != "settings" or c["package"] in seen:
            continue
        seen.add(c["package"])
        cls = tpkg.SurfelTracingSettings if c["package"] == "diff_surfel_tracing" else pkg.GaussianRasterizationSettings
        assert list(cls._fields) == c["order"]                            # same names, same order as the reference passed them
    assert len(seen) == 2
    # keyword sets of the two call forms
    import inspect
    for c in fx["contract"]:
        if c["what"] == "call" and c["package"] == "diff_surfel_tracing":
            sig = inspect.signature(tpkg.SurfelTracer.forward)
            assert set(c["order"]) <= set(sig.parameters)
        if c["what"] == "call" and c["package"].startswith("diff_surfel_rasterization"):
            sig = inspect.signature(pkg.GaussianRasterizer.forward)
            assert set(c["order"]) <= set(sig.parameters)


@pytest.mark.gpu
def test_hip_packages_replay_the_recorded_sampler_sequence(fx):
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from oracle import raster as orc, trace as otr
    dev = torch.device("cuda:0")
    on = lambda v: v.to(dev) if torch.is_tensor(v) else v
    tracers = {}
    n_r = n_t = 0
    for step in fx["steps"]:
        test = "sampler_replay." + step["tag"]
        cs = _calls(fx, step)
        pending = {}
        for c, t in cs:
            if c["package"].startswith("diff_surfel_rasterization") and c["what"] == "call":
                st = pkg.GaussianRasterizationSettings(**{k: on(v) for k, v in t["settings"].items()})
                kw = {k: on(v) for k, v in t["kwargs"].items()}
                with torch.no_grad():
                    outs = pkg.GaussianRasterizer(raster_settings=st)(**kw)
                s, k = t["settings"], t["kwargs"]
                ref = orc.raster_forward(k["means3D"].numpy(), k["opacities"].numpy(), s["viewmatrix"].numpy(), s["projmatrix"].numpy(), s["campos"].numpy(),
                                         int(s["image_width"]), int(s["image_height"]), scales=k["scales"].numpy(), rotations=k["rotations"].numpy(),
                                         colors_precomp=k["colors_precomp"].numpy(), bg=s["bg"].numpy(), scale_modifier=float(s["scale_modifier"]))
                assert np.array_equal(ref["out_color"], t["outputs"][0].numpy())           # the recorded outputs ARE the oracle's on these inputs
                aud = orc.raster_audit(ref)
                ok = ~aud["fragile"]; nfr = int((~ok).sum())
                assert outs[0].shape == tuple(t["outputs"][0].shape) and outs[3].shape == tuple(t["outputs"][3].shape)
                assert torch.equal(outs[1].cpu(), t["outputs"][1])                          # radii: bit-exact
                check_close(test, "raster.image", outs[0].cpu().numpy()[:, ok], t["outputs"][0].numpy()[:, ok], excluded=nfr)
                for ch in range(6):
                    check_close(test, "raster.allmap%d" % ch, outs[2].cpu().numpy()[ch][ok], t["outputs"][2].numpy()[ch][ok], excluded=nfr)
                clean = ~aud["tainted"]
                check_close(test, "raster.weight", outs[3].cpu().numpy()[clean], t["outputs"][3].numpy()[clean], excluded=int((~clean).sum()))
                n_r += 1
            elif c["what"] == "build_acceleration_structure":
                pending[c["tracer"]] = True
            elif c["package"] == "diff_surfel_tracing" and c["what"] == "call":
                tid = c["tracer"]
                if tid not in tracers:
                    tracers[tid] = tpkg.SurfelTracer()                                       # two live tracers over the sequence: each keeps its own state
                tracer = tracers[tid]
                ts = tpkg.SurfelTracingSettings(**{k_: on(v) for k_, v in t["settings"].items()})
                ro, rd, v = [on(x) for x in t["args"]]
                kw = {k_: on(v_) for k_, v_ in t["kwargs"].items() if k_ != "tracer_settings"}
                P = kw["means3D"].shape[0]
                assert pending.pop(tid, False), "the reference rebuilds the structure before every traced call in training (optix_utils.py:73-81)"
                faces = torch.stack([torch.arange(4 * P).reshape(P, 4)[:, :3], torch.arange(4 * P).reshape(P, 4)[:, 1:]], dim=1).reshape(-1, 3).int().to(dev)
                tracer.build_acceleration_structure(v.detach().clone(), faces, rebuild=True)
                with torch.no_grad():
                    outs = tracer(ro, rd, v, tracer_settings=ts, **kw)
                k = t["kwargs"]
                depth = int(t["settings"]["max_trace_depth"]); thr = float(t["settings"]["specular_threshold"])
                oth = None if k["others_precomp"] is None else k["others_precomp"].numpy()
                deg = int(t["settings"]["sh_degree"].item())
                args = (k["means3D"].numpy(), k["scales"].numpy(), k["rotations"].numpy(), k["opacities"].numpy())
                R = ro.reshape(-1, 3).shape[0]
                a = otr.trace_audit(t["args"][0].reshape(-1, 3).numpy(), t["args"][1].reshape(-1, 3).numpy(), *args, others=oth,
                                    start_from_first=bool(k["start_from_first"]), bounce_thr=(thr if depth > 0 else None), shs=k["shs"].numpy(), sh_degree=deg)
                frag = a["fragile"].copy()
                rmid = t["outputs"][6].numpy().reshape(R, -1)
                for b in range(1, depth + 1):                                                # bounce stages: audit the rays the oracle traced there
                    ran = np.abs(rmid[:, 16 * b + 3:16 * b + 6]).sum(-1) > 0
                    if ran.any():
                        ab = otr.trace_audit(rmid[ran, 16 * b:16 * b + 3], rmid[ran, 16 * b + 3:16 * b + 6], *args, others=oth, start_from_first=False, tmin=1e-3,
                                             bounce_thr=(thr if b < depth else None), shs=k["shs"].numpy(), sh_degree=deg)
                        frag[np.nonzero(ran)[0][ab["fragile"]]] = True
                okr = ~frag
                record_fragile(test, "trace.fragile_rays", frag, FRAGILE_RAYS_MAX)
                lead = tuple(ro.shape[:-1])
                assert outs[0].shape == lead + (3,) and outs[6].shape == lead + (16 * (depth + 1),) and outs[7].shape == (P, 1)
                for i, nm in ((0, "rgb"), (1, "dpt"), (2, "acc"), (3, "norm")):
                    got = outs[i].detach().cpu().numpy().reshape(R, -1)[okr]; want = t["outputs"][i].numpy().reshape(R, -1)[okr]
                    check_close(test, "trace." + nm, got, want, excluded=int((~okr).sum()))
                if depth > 0:                                                                # every stage's record, and the weights over all stages
                    check_close(test, "trace.mid", outs[6].cpu().numpy().reshape(R, -1)[okr], rmid[okr], excluded=int((~okr).sum()))
                n_t += 1
    assert n_r == sum(1 for c in fx["contract"] if c["what"] == "call" and c["package"].startswith("diff_surfel_rast")) and n_r >= 10
    assert n_t == sum(1 for c in fx["contract"] if c["what"] == "call" and c["package"] == "diff_surfel_tracing") and n_t >= 8 and len(tracers) == 2
