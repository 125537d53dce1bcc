"""TEST INFRASTRUCTURE — generate the golden fixtures in tests/golden by running the REAL
reference (imported from /root/reference, CPU) on the hashed inputs of ``oracle.cases``.

Run in the build container only:   python -m oracle.gen_golden [--big]
Only outputs (data) are stored; weights / inputs / eps are regenerated from their names.

  ops_c8.npz      G1/G2: single-module calls (gru cells, dual cells, p_model+sample, encoder,
                  decoder, ConvNeXt block, DeepLabHead, SpatialGRU, ode_step euler/midpoint)
  fpode.npz       G3/G5: NNFOwithBayesianJumps.forward + FuturePredictionODE.forward outputs
  schedules.json  G4: (kind, dt) op lists + selection indices produced by the reference's own
                  control flow (temporal_ode_bayes.py:508-620) with its numerics stubbed out
  big_stats.json  G7: statistics of the C=64, 200x200 forward (tensors are 72 MB)
  voxelize.npz    N2: mmdet3d Voxelization (hard voxelisation) outputs from the reference's C++ CPU kernel
  unused_cells.npz a16: Dual_GRU / BiGRU (layers/temporal.py:59-249) and the dual cells with several present frames
  lift_splat.npz  N1: streamingflow.bev_pool / projection_to_birds_eye_view / get_geometry /
                  create_frustum / pose_vec2mat / mmdet3d bev_pool (+ QuickCumsum) outputs
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

from workloads import hashfill
from . import cases, refimport

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().contiguous().numpy().astype(np.float32)


def build_ref(m, C, solver="euler", impute=True, variable=True, delta_t=0.05):
    cfg = refimport.make_cfg(C, impute=impute, solver=solver, variable=variable)
    net = m.FuturePredictionODE(in_channels=C, latent_dim=C, n_future=4, cfg=cfg, mixture=True,
                                n_gru_blocks=2, n_res_layers=1, delta_t=delta_t).eval()
    sd = cases.fpode_state_dict(net.state_dict())
    net.load_state_dict(sd)
    return net, sd


def gen_ops(m):
    """G1/G2 at C=8, latent 12x12 (BEV 48x48)."""
    C, h, w = 8, 12, 12
    out = {}
    net, _ = build_ref(m, C)
    ode = net.gru_ode
    x = hashfill.normal("op_x", (1, C, h, w), 11)
    s = hashfill.normal("op_s", (1, C, h, w), 12) * 0.5
    with torch.no_grad():
        out["spatial_gru_cell"] = _np(net.spatial_grus[0].gru_cell(x, s))
        out["dual_ode_cell"] = _np(ode.gru_c(x, s))
        out["dual_obs_cell"] = _np(ode.gru_obs(s, None, x)[0])
        with refimport.patched_standard_normal(hashfill.HashedNoise(cases.EPS_SEED)):
            y0, q = ode.infer_state(s)
        out["infer_state_y"], out["infer_state_q"] = _np(y0), _np(q)
        bev = hashfill.normal("op_bev", (1, 2, C, 4 * h, 4 * w), 13)
        out["srvp_encode"] = _np(ode.srvp_encode(bev)[0])
        lat = hashfill.normal("op_lat", (1, 2, C, h, w), 14) * 0.5
        out["srvp_decode"] = _np(ode.srvp_decode(lat))
        frames = hashfill.normal("op_frames", (3, C, 4 * h, 4 * w), 15)
        out["convnext_block"] = _np(net.res_blocks[0](frames))
        out["deeplab_head"] = _np(net.res_blocks[1](frames))
        seq = hashfill.normal("op_seq", (1, 3, C, 4 * h, 4 * w), 16)
        out["spatial_gru_seq"] = _np(net.spatial_grus[1](seq, seq[:, 0]))
        for solver in ("euler", "midpoint"):
            for impute in (True, False):
                ode.solver, ode.impute = solver, impute
                for dt in (0.05, torch.tensor(0.37, dtype=torch.float64)):
                    with refimport.patched_standard_normal(hashfill.HashedNoise(cases.EPS_SEED)):
                        st, inp, ct, _, _ = ode.ode_step(s, x, dt, 0.0)
                    tag = f"ode_step_{solver}_{'imp' if impute else 'noimp'}_{float(dt):.2f}"
                    out[tag + "_state"], out[tag + "_input"] = _np(st), _np(inp)
    np.savez_compressed(os.path.join(OUT, "ops_c8.npz"), **out)
    print("ops_c8.npz", {k: v.shape for k, v in out.items()})


def gen_gate_bias(m):
    """gru_bias_init != 0 (a constant added to the gate pre-activations, temporal.py:50-51, temporal_ode_bayes.py:54-55,
    :139-140, :154-155): the reference's own cells with hashed weights -> tests/golden/gate_bias.npz."""
    C, h, w = 8, 12, 12
    x = hashfill.normal("gb_x", (1, C, h, w), 21)
    s = hashfill.normal("gb_s", (1, C, h, w), 22) * 0.5
    out = {}
    with torch.no_grad():
        for tag, gb in (("pos", 0.7), ("neg", -1.3)):
            cell = m.temporal.SpatialGRU(C, C, gru_bias_init=gb).eval()
            cell.load_state_dict(hashfill.fill_state_dict(cell.state_dict(), seed=31, gain=0.6))
            out[f"spatial_gru_cell_{tag}"] = _np(cell.gru_cell(x, s))
            dual = m.tob.DualGRUODECell(C, C, gru_bias_init=gb).eval()
            dual.load_state_dict(hashfill.fill_state_dict(dual.state_dict(), seed=32, gain=0.6))
            out[f"dual_ode_cell_{tag}"] = _np(dual(x, s))
            obs = m.tob.DualGRUCell(C, C, gru_bias_init=gb).eval()
            obs.load_state_dict(hashfill.fill_state_dict(obs.state_dict(), seed=33, gain=0.6))
            out[f"dual_cell_{tag}"] = _np(obs(x, s))
    np.savez_compressed(os.path.join(OUT, "gate_bias.npz"), **out)
    print("gate_bias.npz", list(out))


def gen_fpode(m, table=None, fname="fpode.npz", keep_decoded=True):
    out = {}
    for name, (C, H, W, ts, solver, impute, variable, eps0) in (table or cases.FPODE_CASES).items():
        cts, lts, tts, dt = cases.timeset(ts)
        net, _ = build_ref(m, C, solver, impute, variable, dt)
        cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
        x_in = cases.present_input(cam, lid)
        grabbed = {}
        hook = net.gru_ode.register_forward_hook(lambda mod, a, o: grabbed.__setitem__("nnfo", o))
        with torch.no_grad(), refimport.patched_standard_normal(hashfill.HashedNoise(cases.EPS_SEED, zero=eps0)):
            y, aux = net(x_in, cam, lid, cts, lts, tts)
        hook.remove()
        assert aux == 0
        out[name + "/out"] = _np(y)
        out[name + "/nnfo_state"] = _np(grabbed["nnfo"][0])
        if keep_decoded:
            out[name + "/nnfo_x"] = _np(grabbed["nnfo"][2])
        print(name, y.shape, float(y.abs().max()))
    np.savez_compressed(os.path.join(OUT, fname), **out)


def reference_schedule(m, times, T, delta_t, variable):
    """Run the reference's own ``NNFOwithBayesianJumps.forward`` control flow with every numeric
    sub-call stubbed (state value == number of ops applied so far)."""
    tob = m.tob
    ode = tob.NNFOwithBayesianJumps(8, 8, refimport.make_cfg(8, variable=variable))
    ops = []

    class FakeObs(torch.nn.Module):
        def forward(self, state, p, X_obs):
            ops.append(["jump", None])
            return state + 1, None

    def fake_step(state, inp, dt, current_time):
        ops.append(["step", float(dt)])
        current_time += dt
        return state + 1, inp, current_time, torch.zeros(1, dtype=torch.float64), torch.zeros(1)

    ode.gru_obs = FakeObs()
    ode.ode_step = fake_step
    ode.srvp_encode = lambda x: (x, None)
    ode.srvp_decode = lambda x, skip=None: x
    ode.infer_state = lambda x, deterministic=False: (x, None)
    n = len(times)
    obs = torch.zeros(1, n, 1, 1, 1)
    inp = torch.zeros(1, 1, 1, 1, 1)
    _, _, x = ode.forward(times, inp, obs, delta_t, T)
    sel = [int(v) for v in x.reshape(-1).tolist()]
    return ops, sel


def gen_schedules(m):
    from .ref_torch import merge_observations
    out = {}
    for name in cases.TIMESETS:
        cts, lts, tts, dt = cases.timeset(name)
        C = 1
        cam = torch.zeros(1, cts.shape[1], C, 1, 1)
        lid = torch.zeros(1, lts.shape[1], C, 1, 1)
        # the merge itself is FuturePredictionODE.forward:37-49; reproduce it with the
        # reference's exact statements (dict of 0-d tensor keys, sorted) to get `times`.
        obs_feature_with_time = {}
        for index in range(cts.shape[1]):
            obs_feature_with_time[cts[0, index]] = ("cam", index)
        for index in range(lts.shape[1]):
            obs_feature_with_time[lts[0, index]] = ("lidar", index)
        obs = dict(sorted(obs_feature_with_time.items(), key=lambda v: v[0]))
        times = torch.tensor(list(obs.keys()))
        order = [[s, i] for (s, i) in obs.values()]
        for variable in (True, False):
            ops, sel = reference_schedule(m, times, tts[0], dt, variable)
            out[f"{name}/{'variable' if variable else 'fixed'}"] = {
                "camera_ts": cts[0].tolist(), "lidar_ts": lts[0].tolist(), "target_ts": tts[0].tolist(),
                "delta_t": dt, "variable": variable, "obs_order": order,
                "ops": ops, "select_nops": sel,
                "n_steps": sum(o[0] == "step" for o in ops), "n_jumps": sum(o[0] == "jump" for o in ops)}
            print(name, variable, out[f"{name}/{'variable' if variable else 'fixed'}"]["n_steps"], sel)
    with open(os.path.join(OUT, "schedules.json"), "w") as f:
        json.dump(out, f, indent=0)


def gen_beverse(m):
    """G6: BEVerse FuturePrediction / SpatialDistributionModule / DistributionModule and the unused
    streamingflow DistributionModule; incl. BASELINE config 1's FuturePrediction(32, 16, 3, 3) case."""
    refimport.install()
    from mmdet3d.models.beverse.models import motion_modules as RM
    from streamingflow.models import distributions as RD
    out = {}
    for tag, (C, lat, h, w, T) in cases.BEVERSE_CASES.items():
        fp = RM.FuturePrediction(C, lat, 3, 3).eval()
        fp.load_state_dict(hashfill.fill_state_dict(fp.state_dict(), seed=2))
        x = hashfill.normal("bv_x", (1, T, lat, h, w), 21)
        hid = hashfill.normal("bv_h", (1, C, h, w), 22)
        s_t = hashfill.normal("bv_s", (1, 1, C, h, w), 23)
        sdm = RM.SpatialDistributionModule(C, lat, -5.0, 5.0).eval()
        sdm.load_state_dict(hashfill.fill_state_dict(sdm.state_dict(), seed=2, gain=3.0))
        dm = RM.DistributionModule(C, lat, -0.05, 0.05).eval()
        dm.load_state_dict(hashfill.fill_state_dict(dm.state_dict(), seed=2, gain=3.0))
        sfd = RD.DistributionModule(C, lat).eval()
        sfd.load_state_dict(hashfill.fill_state_dict(sfd.state_dict(), seed=2, gain=2.0))
        with torch.no_grad():
            out[tag + "/future_prediction"] = _np(fp(x, hid))
            mu, ls = sdm(s_t)
            out[tag + "/spatial_mu"], out[tag + "/spatial_log_sigma"] = _np(mu), _np(ls)
            mu, ls = dm(s_t)
            out[tag + "/dist_mu"], out[tag + "/dist_log_sigma"] = _np(mu), _np(ls)
            out[tag + "/sf_dist"] = _np(sfd(s_t))
            for meth, key in (("MIXGAUSSIAN", "sf_dist_mix"), ("BERNOULLI", "sf_dist_bern")):
                mod = RD.DistributionModule(C, lat, method=meth).eval()
                mod.load_state_dict(hashfill.fill_state_dict(mod.state_dict(), seed=2, gain=2.0))
                out[tag + "/" + key] = _np(mod(s_t))
        print(tag, {k.split("/")[1]: v.shape for k, v in out.items() if k.startswith(tag)})
    # a16: single-branch cells defined (unused) in temporal_ode_bayes.py
    tob = m.tob
    x = hashfill.normal("sg_x", (2, 16, 12, 10), 31)
    st = hashfill.normal("sg_s", (2, 16, 12, 10), 32) * 0.5
    for name, cls in (("gru_ode_cell", tob.SpatialGRUODECell), ("gru_cell", tob.SpatialGRUCell)):
        mod = cls(16, 16).eval()
        mod.load_state_dict(hashfill.fill_state_dict(mod.state_dict(), seed=4))
        with torch.no_grad():
            out["single/" + name] = _np(mod(x, st))
    np.savez_compressed(os.path.join(OUT, "beverse.npz"), **out)


def _big_stats(m, C, H, W, ts, solver, impute, variable):
    cts, lts, tts, dt = cases.timeset(ts)
    net, _ = build_ref(m, C, solver, impute, variable, dt)
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    x_in = cases.present_input(cam, lid)
    grabbed = {}
    net.gru_ode.register_forward_hook(lambda mod, a, o: grabbed.__setitem__("nnfo", o))
    with torch.no_grad(), refimport.patched_standard_normal(hashfill.HashedNoise(cases.EPS_SEED)):
        y, _ = net(x_in, cam, lid if lts.shape[1] else torch.zeros((1, 0, C, H, W)), cts, lts, tts)
    stats = {}
    for k, t in (("out", y), ("nnfo_state", grabbed["nnfo"][0]), ("nnfo_x", grabbed["nnfo"][2])):
        flat = t.reshape(-1).double()
        idx = torch.linspace(0, flat.numel() - 1, 256, dtype=torch.float64).long().clamp(max=flat.numel() - 1)
        stats[k] = {"shape": list(t.shape), "mean": flat.mean().item(), "absmax": flat.abs().max().item(),
                    "std": flat.std().item(), "sum": flat.sum().item(),
                    "sample_idx": idx.tolist(), "samples": flat[idx].tolist()}
        print(k, stats[k]["shape"], stats[k]["mean"], stats[k]["absmax"], stats[k]["std"])
    return stats


def gen_big(m, only_cases=False):
    """G7: full-size forwards — statistics only.  Top level: C=64, BEV 200x200, shipped schedule (BASELINE config 2);
    "cases": config 4 (19 frames) and config 1 (C=32, 4 fixed Euler steps), oracle.cases.BIG_CASES."""
    path = os.path.join(OUT, "big_stats.json")
    if only_cases and os.path.exists(path):
        stats = json.load(open(path))
    else:
        stats = _big_stats(m, 64, 200, 200, "shipped", "euler", True, True)
    stats["cases"] = {tag: _big_stats(m, *cfg) for tag, cfg in cases.BIG_CASES.items()}
    with open(path, "w") as f:
        json.dump(stats, f)


def _stats_of(t):
    flat = t.reshape(-1).double()
    idx = torch.linspace(0, flat.numel() - 1, 256, dtype=torch.float64).long().clamp(max=flat.numel() - 1)
    return {"shape": list(t.shape), "mean": flat.mean().item(), "absmax": flat.abs().max().item(), "std": flat.std().item(),
            "sum": flat.sum().item(), "sample_idx": idx.tolist(), "samples": flat[idx].tolist()}


def gen_big_stream(m):
    """Round 3: BASELINE config 5 at full size.  Adds to big_stats.json (existing entries untouched): "cases" entries of
    oracle.cases.BIG_STREAM_CASES from the real reference (euler, midpoint), "oracle_cases" entries of BIG_ORACLE_CASES from
    oracle/ref_torch.py (rk4 is build-defined: the reference has no such solver)."""
    from . import ref_torch as R
    path = os.path.join(OUT, "big_stats.json")
    stats = json.load(open(path))
    for tag, cfg in cases.BIG_STREAM_CASES.items():
        if tag not in stats["cases"]:
            stats["cases"][tag] = _big_stats(m, *cfg)
            json.dump(stats, open(path, "w"))
    stats.setdefault("oracle_cases", {})
    for tag, (C, H, W, ts, solver, impute, variable) in cases.BIG_ORACLE_CASES.items():
        if tag in stats["oracle_cases"]:
            continue
        cts, lts, tts, dt = cases.timeset(ts)
        _, sd = build_ref(m, C, "euler", impute, variable, dt)       # the reference class only provides the state_dict
        cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
        with torch.no_grad():
            y, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2, solver, impute, variable,
                                                    hashfill.HashedNoise(cases.EPS_SEED))
        stats["oracle_cases"][tag] = {"out": _stats_of(y)}
        print(tag, stats["oracle_cases"][tag]["out"]["mean"], stats["oracle_cases"][tag]["out"]["absmax"])
        json.dump(stats, open(path, "w"))


def gen_lift():
    """N1: camera lift-splat.  The reference's own Python runs (bev_pool.py, streamingflow.bev_pool,
    projection_to_birds_eye_view, get_geometry, create_frustum, pose_vec2mat); only the CUDA kernel
    behind ``bev_pool_ext.bev_pool_forward`` is the oracle's restatement (no nvcc here)."""
    from types import SimpleNamespace as NS
    from . import lift_splat as LS
    R = refimport.lift_splat_reference(LS.bev_pool_kernel)
    out = {}

    def fake_self(start, res, dim, discount=0.5):
        me = NS(bev_start_position=start, bev_resolution=res, bev_dimension=dim, discount=discount)
        me.bev_pool = lambda g, x: R.model.bev_pool(me, g, x)
        return me
    with torch.no_grad():
        for tag in cases.LIFT_POOL_CASES:
            geo, x, start, res, dim = cases.lift_pool_inputs(tag)
            if tag == "empty":      # no point inside the grid: the reference fails (bev_pool.py:45)
                try:
                    R.model.bev_pool(fake_self(start, res, dim), geo.clone(), x.clone())
                    raise AssertionError("expected the reference to raise on an empty frame")
                except IndexError:
                    continue
            pooled, kept = R.model.bev_pool(fake_self(start, res, dim), geo.clone(), x.clone())
            out["pool_" + tag] = _np(pooled)
            out["pool_kept_" + tag] = kept.numpy().astype(np.int32)
            mine, kept2 = LS.sf_bev_pool(geo, x, start, res, dim, stable=False)
            assert torch.equal(kept, kept2) and torch.equal(pooled, mine), tag    # same argsort => same bits
            print("pool", tag, tuple(pooled.shape), int(kept.shape[0]), "kept of", x.numel() // x.shape[-1])
        for tag in cases.LIFT_CASES:
            feat, depth, geo, ego, (start, res, dim), discount = cases.lift_inputs(tag)
            b, s, n, C, fH, fW = feat.shape
            x = LS.depth_outer(feat.reshape(b * s * n, C, fH, fW), depth.reshape(b * s * n, -1, fH, fW))
            x = x.reshape(b, s, n, *x.shape[1:])
            ref = R.model.projection_to_birds_eye_view(fake_self(start, res, dim, discount), x.clone(), geo.clone(), ego.clone())
            out["proj_" + tag] = _np(ref)
            mine = LS.projection_to_birds_eye_view(x, geo, ego, start, res, dim, discount, stable=False)
            assert torch.equal(ref, mine), tag
            print("proj", tag, tuple(ref.shape), float(ref.abs().max()))
        # frustum + geometry of a small rig through the reference's create_frustum / get_geometry
        cfg = NS(IMAGE=NS(FINAL_DIM=(32, 48)), LIFT=NS(D_BOUND=[2.0, 10.0, 1.0]))
        me = NS(cfg=cfg, encoder_downsample=8)
        fr = R.model.create_frustum(me).data
        assert torch.equal(fr, LS.create_frustum((32, 48), 8, [2.0, 10.0, 1.0]))
        out["frustum"] = _np(fr)
        intr = torch.tensor([[20.0, 0.0, 24.0], [0.0, 20.0, 16.0], [0.0, 0.0, 1.0]]).repeat(1, 2, 1, 1)
        ang = hashfill.uniform("lift_extr_r", (1, 2, 3), -1.0, 1.0, seed=31)
        extr = R.geometry.pose_vec2mat(torch.cat([hashfill.uniform("lift_extr_t", (1, 2, 3), -1.0, 1.0, seed=32), ang], -1))
        assert torch.equal(extr, LS.pose_vec2mat(torch.cat([hashfill.uniform("lift_extr_t", (1, 2, 3), -1.0, 1.0, seed=32), ang], -1)))
        me.frustum = fr
        g = R.model.get_geometry(me, intr, extr)
        assert torch.equal(g, LS.get_geometry(fr, intr, extr))
        out["geometry"] = _np(g)
        out["pose_vec2mat"] = _np(extr)
        # the pooling op itself, with duplicated coordinates, vs the reference's QuickCumsum implementation
        n, c = 3000, 8
        coords = (hashfill.uniform("lift_op_coords", (n, 4), 0.0, 1.0, seed=33) * torch.tensor([7.0, 6.0, 2.0, 2.0])).long()
        feats = hashfill.normal("lift_op_feats", (n, c), seed=34)
        a = R.bev_pool_py.bev_pool(feats, coords, 2, 2, 7, 6)
        assert torch.equal(a, LS.bev_pool_op(feats, coords, 2, 2, 7, 6, stable=False))
        ranks = coords[:, 0] * (6 * 2 * 2) + coords[:, 1] * (2 * 2) + coords[:, 2] * 2 + coords[:, 3]
        idx = ranks.argsort()
        xq, gq = R.bev_pool_py.QuickCumsum.apply(feats[idx], coords[idx], ranks[idx])
        o = torch.zeros(2, 2, 7, 6, c)
        o[gq[:, 3], gq[:, 2], gq[:, 0], gq[:, 1]] = xq
        print("bev_pool op vs QuickCumsum:", float((o.permute(0, 4, 1, 2, 3) - a).abs().max()))
        assert float((o.permute(0, 4, 1, 2, 3) - a).abs().max()) < 1e-4
        out["op_bev_pool"] = _np(a)
        out["op_quickcumsum"] = _np(o.permute(0, 4, 1, 2, 3))
    np.savez_compressed(os.path.join(OUT, "lift_splat.npz"), **out)
    print("lift_splat.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def gen_voxel():
    """N2 (voxelise half): the reference's Voxelization module on the reference's own C++ CPU kernel
    (oracle/build_ref.py compiles mmdet3d/ops/voxel/src/voxelization{,_cpu}.cpp in place)."""
    from . import voxelize as VZ
    R = refimport.voxel_reference()
    out = {}
    for tag, (n, F, vs, rng, mp, mv) in cases.VOXEL_CASES.items():
        pts = cases.voxel_points(tag)
        m = R.Voxelization(list(vs), list(rng), mp, (mv, mv)).eval()
        v, c, k = m(pts)
        out["voxels_" + tag], out["coors_" + tag], out["num_" + tag] = _np(v), c.numpy().astype(np.int32), k.numpy().astype(np.int32)
        v2, c2, k2 = VZ.hard_voxelize(pts.numpy(), vs, rng, mp, mv)
        assert np.array_equal(v.numpy(), v2) and np.array_equal(c.numpy(), c2) and np.array_equal(k.numpy(), k2), tag
        # the dynamic branch of the same module (max_points = -1): per-point coordinates
        dc = R.Voxelization(list(vs), list(rng), -1, (mv, mv)).eval()(pts)
        out["dyn_coors_" + tag] = dc.numpy().astype(np.int32)
        assert np.array_equal(out["dyn_coors_" + tag], VZ.dynamic_voxelize(pts.numpy(), vs, rng)), tag
        print("voxel", tag, tuple(v.shape), int(k.sum()), "points kept of", n)
    np.savez_compressed(os.path.join(OUT, "voxelize.npz"), **out)


def gen_decoder():
    """N3 (Decoder): the reference's Decoder class on the oracle's restatement of torchvision's resnet18 trunk."""
    from . import decoder_ref as DR
    D = refimport.decoder_reference()
    out, keys = {}, {}
    for tag, (cin, ncls, npres, nhd, gate, (b, s, h, w)) in cases.DECODER_CASES.items():
        m = D(cin, ncls, npres, nhd, gate).eval()
        sd = cases.decoder_state_dict(m.state_dict())
        m.load_state_dict(sd)
        x = hashfill.normal("dec_x_" + tag, (b, s, cin, h, w), seed=62)
        with torch.no_grad():
            ref = m(x)
            mine = DR.decoder_forward(sd, x, npres)
        for k, v in ref.items():
            if v is None:
                assert mine[k] is None
                continue
            assert torch.equal(v, mine[k]), (tag, k)
            out[f"{tag}.{k}"] = _np(v)
        keys[tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print("decoder", tag, {k: tuple(v.shape) for k, v in ref.items() if v is not None})
    np.savez_compressed(os.path.join(OUT, "decoder.npz"), **out)
    with open(os.path.join(OUT, "decoder_state_dict_keys.json"), "w") as f:
        json.dump(keys, f)


def gen_temporal():
    """N3 (TemporalModel): the reference class itself."""
    from . import temporal_model_ref as TR
    T = refimport.temporal_model_reference()
    out, keys = {}, {}
    for tag, (cin, rf, start, extra, inb, pyr, (b, s, h, w)) in cases.TEMPORAL_CASES.items():
        m = T(cin, rf, (h, w), start_out_channels=start, extra_in_channels=extra, n_spatial_layers_between_temporal_layers=inb,
              use_pyramid_pooling=pyr).eval()
        sd = cases.decoder_state_dict(m.state_dict(), seed=71)
        m.load_state_dict(sd)
        x = hashfill.normal("tm_x_" + tag, (b, s, cin, h, w), seed=72)
        with torch.no_grad():
            ref = m(x)
            mine = TR.temporal_model_forward(sd, x, (h, w))
        assert torch.equal(ref, mine), tag
        out[tag] = _np(ref)
        keys[tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print("temporal", tag, tuple(ref.shape), float(ref.abs().max()))
    np.savez_compressed(os.path.join(OUT, "temporal_model.npz"), **out)
    with open(os.path.join(OUT, "temporal_model_state_dict_keys.json"), "w") as f:
        json.dump(keys, f)


def gen_eval():
    """N4: the reference's instance post-processing (utils/instance.py, as is) and metrics (metrics.py on a restated
    pytorch-lightning Metric base) on synthetic decoder outputs."""
    R = refimport.eval_reference()
    I, M = R.instance, R.metrics
    out = {}
    for seed in (0, 1, 2):
        o, labels = cases.eval_scene(seed)
        centers = I.find_instance_centers(o["instance_center"][0, 0].clone(), conf_threshold=0.1)
        out[f"centers_{seed}"] = centers.numpy()
        fg = torch.argmax(o["segmentation"][0, 0], 0) == 1
        inst0, _ = I.get_instance_segmentation_and_centers(o["instance_center"][0, 0].clone(), o["instance_offset"][0, 0], fg)
        out[f"inst0_{seed}"] = inst0.numpy()
        cons, traj = I.predict_instance_segmentation_and_trajectories({k: v.clone() for k, v in o.items()}, compute_matched_centers=True)
        out[f"consistent_{seed}"] = cons.numpy()
        for k, v in traj.items():
            out[f"traj_{seed}_{k}"] = np.ascontiguousarray(v)
        iou = M.IntersectionOverUnion(2)
        seg_pred = torch.argmax(o["segmentation"], dim=2, keepdim=True)
        iou(seg_pred, labels["segmentation"])
        iou(seg_pred[:, 1:], labels["segmentation"][:, 1:])
        out[f"iou_tp_{seed}"], out[f"iou_fp_{seed}"] = iou.true_positive.numpy(), iou.false_positive.numpy()
        out[f"iou_fn_{seed}"], out[f"iou_{seed}"] = iou.false_negative.numpy(), iou.compute().numpy()
        pq = M.PanopticMetric(2)
        pq(cons, labels["instance"])
        res = pq.compute()
        for k in ("true_positive", "false_positive", "false_negative"):
            out[f"pq_{k}_{seed}"] = getattr(pq, k).numpy()
        out[f"pq_iou_{seed}"], out[f"pq_{seed}"] = pq.iou.numpy(), res["pq"].numpy()
        print("eval", seed, "centers", len(centers), "ids", int(cons.max()), "iou", iou.compute().tolist(), "pq", res["pq"].tolist())
    np.savez_compressed(os.path.join(OUT, "eval.npz"), **out)


def gen_labels():
    """N4 (label warping): the reference's geometry helpers themselves (utils/geometry.py) under the restated
    dictionary plumbing of TrainingModule.prepare_future_labels."""
    from types import SimpleNamespace as NS
    from . import labels_ref as LR
    refimport.install()
    import importlib
    G = importlib.import_module("streamingflow.utils.geometry")
    cfg = NS(LIFT=NS(GT_DEPTH=True, D_BOUND=[2.0, 50.0, 1.0]), SEMANTIC_SEG=NS(PEDESTRIAN=NS(ENABLED=False)),
             INSTANCE_SEG=NS(ENABLED=True), INSTANCE_FLOW=NS(ENABLED=True))
    out = {}
    for seed in (0, 1):
        batch = cases.label_batch(seed)
        lab = LR.prepare_future_labels(G, batch, cfg, 3, (50.0, 50.0), 8)
        for k, v in lab.items():
            out[f"{seed}.{k}"] = v.numpy()
        x = batch["centerness"][:, 0]
        out[f"{seed}.warp_bilinear"] = G.warp_features(x, batch["future_egomotion"][:, 0], mode="bilinear", spatial_extent=(50.0, 50.0)).numpy()
        print("labels", seed, {k: tuple(v.shape) for k, v in lab.items()})
    np.savez_compressed(os.path.join(OUT, "labels.npz"), **out)


def gen_unused_cells(m):
    """The recurrent modules the reference defines but never constructs (layers/temporal.py:59-249) and the dual cells'
    several-present-frames path (temporal_ode_bayes.py:101-109, :248-256): the reference's own classes, hashed weights."""
    C = 8
    I = cases.unused_cell_inputs()
    out = {}
    fill = lambda mod, key: mod.load_state_dict(hashfill.fill_state_dict(mod.state_dict(), seed=cases.UNUSED_CELL_SEEDS[key], gain=0.6))
    with torch.no_grad():
        for mix in (True, False):
            net = m.temporal.Dual_GRU(C, C, n_future=3, mixture=mix, gru_bias_init=0.3).eval()
            fill(net, "dual_gru")
            out[f"dual_gru_mix{int(mix)}_p1"] = _np(net(I["x1"], I["st1"]))
            out[f"dual_gru_mix{int(mix)}_p3"] = _np(net(I["x1"], I["st3"]))
        net = m.temporal.Dual_GRU(2 * C, C, n_future=2, mixture=True).eval()       # input wider than the state
        fill(net, "dual_gru_wide")
        out["dual_gru_wide"] = _np(net(I["x1w"], I["st1"]))
        net = m.temporal.BiGRU(C, gru_bias_init=-0.2).eval()
        fill(net, "bigru")
        out["bigru"] = _np(net(I["seq"]))
        net = m.tob.DualGRUODECell(C, C).eval()
        fill(net, "dual_ode")
        out["dual_ode_p2"] = _np(net(I["x1b1"], I["st2b1"]))
        net = m.tob.DualGRUCell(C, C).eval()
        fill(net, "dual_obs")
        out["dual_obs_p3"] = _np(net(I["x1"], I["st3"]))
    np.savez_compressed(os.path.join(OUT, "unused_cells.npz"), **out)
    print("unused_cells.npz", {k: v.shape for k, v in out.items()})
    keys = {"Dual_GRU(16, 8, 2)": m.temporal.Dual_GRU(2 * C, C, n_future=2), "BiGRU(8)": m.temporal.BiGRU(C),
            "Bottleblock(16, 8)": m.convolutions.Bottleblock(2 * C, C), "Bottleblock(8)": m.convolutions.Bottleblock(C)}
    json.dump({k: {n: list(t.shape) for n, t in mod.state_dict().items()} for k, mod in keys.items()},
              open(os.path.join(OUT, "unused_cells_keys.json"), "w"), indent=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also generate the C=64 200x200 statistics (slow)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    todo = a.only.split(",") if a.only else ["ops", "fpode", "schedules", "beverse", "lift", "voxel", "decoder", "temporal", "eval", "labels"]
    if "lift" in todo:
        gen_lift()
    if "voxel" in todo:
        gen_voxel()
    if "decoder" in todo:
        gen_decoder()
    if "temporal" in todo:
        gen_temporal()
    if "eval" in todo:
        gen_eval()
    if "labels" in todo:
        gen_labels()
    if not set(todo) - {"lift", "voxel", "decoder", "temporal", "eval", "labels"} and not a.big:
        return
    m = refimport.modules()
    if "ops" in todo:
        gen_ops(m)
    if "fpode" in todo:
        gen_fpode(m)
    if "schedules" in todo:
        gen_schedules(m)
    if "beverse" in todo:
        gen_beverse(m)
    if "gate_bias" in todo:
        gen_gate_bias(m)
    if "unused_cells" in todo:
        gen_unused_cells(m)
    if "fpode_stream" in todo:
        gen_fpode(m, cases.FPODE_STREAM_CASES, "fpode_stream.npz", keep_decoded=False)
    if a.big or "big" in todo:
        gen_big(m)
    if "big_cases" in todo:
        gen_big(m, only_cases=True)
    if "big_stream" in todo:
        gen_big_stream(m)


if __name__ == "__main__":
    sys.exit(main())
