"""TEST INFRASTRUCTURE — shared definitions of the parity cases.

Both ``oracle/gen_golden.py`` (which runs the real reference in the build container) and the tests
(which run the oracle / the HIP path) regenerate inputs and weights from these definitions with
``workloads.hashfill``; only expected outputs are stored under ``tests/golden``.
"""
import torch

from workloads import hashfill
from workloads.synthetic import (EPS_SEED, TIMESETS, WEIGHT_GAIN_DEFAULT, WEIGHT_GAINS, WEIGHT_SEED, bev_inputs,       # noqa: F401  (re-exported)
                                 fpode_state_dict, present_input, timeset, VOXEL_SHIPPED, decoder_state_dict)


# Full-forward golden cases: name -> (C, H, W, timeset, solver, impute, variable, eps zero?)
FPODE_CASES = {
    "c8_16_shipped":        (8, 16, 16, "shipped", "euler", True, True, False),
    "c8_16_shipped_eps0":   (8, 16, 16, "shipped", "euler", True, True, True),
    "c8_16_fixed":          (8, 16, 16, "shipped", "euler", True, False, False),
    "c8_16_noimpute":       (8, 16, 16, "shipped", "euler", False, True, False),
    "c8_16_midpoint":       (8, 16, 16, "shipped", "midpoint", True, True, False),
    "c8_16_midpoint_fixed": (8, 16, 16, "dt_010", "midpoint", True, False, False),
    "c8_48_shipped":        (8, 48, 48, "shipped", "euler", True, True, False),
    "c8_48_config1":        (8, 48, 48, "config1", "euler", True, False, False),
    "c16_24_irregular":     (16, 24, 24, "irregular", "euler", True, True, False),
    "c8_16_lidar_only":     (8, 16, 16, "lidar_only", "euler", True, True, False),
}


# BASELINE config 5 (streaming fine interval: 40 targets at 0.05 s -> 46 ODE steps) and config 4 (8 s horizon:
# 19 frames), small enough to keep whole outputs: tests/golden/fpode_stream.npz.  rk4 has no reference
# implementation (build-defined): checked against the oracle only.
FPODE_STREAM_CASES = {
    "c8_16_stream40_euler":    (8, 16, 16, "stream40", "euler", True, True, False),
    "c8_16_stream40_midpoint": (8, 16, 16, "stream40", "midpoint", True, True, False),
    "c16_24_stream40_euler":   (16, 24, 24, "stream40", "euler", True, True, False),
    "c8_16_future16_euler":    (8, 16, 16, "future16", "euler", True, True, False),
    "c8_16_stream40_noimpute": (8, 16, 16, "stream40", "midpoint", False, True, False),
}
# full-size statistics (tests/golden/big_stats.json "cases"): tag -> (C, H, W, timeset, solver, impute, variable)
BIG_CASES = {
    "config4_future16": (64, 200, 200, "future16", "euler", True, True),
    "config1_c32":      (32, 200, 200, "config1", "euler", True, False),
}


# BASELINE config 5 at full size (C=64, BEV 200x200, the 46-step streaming schedule, 43 decoded frames): statistics only —
# euler / midpoint from the real reference, rk4 (build-defined solver) from the oracle (big_stats.json "cases" / "oracle_cases")
BIG_STREAM_CASES = {
    "config5_stream40_euler":    (64, 200, 200, "stream40", "euler", True, True),
    "config5_stream40_midpoint": (64, 200, 200, "stream40", "midpoint", True, True),
}
BIG_ORACLE_CASES = {
    "config5_stream40_rk4":      (64, 200, 200, "stream40", "rk4", True, True),
}


# BEVerse-named secondary classes: tag -> (in_channels, latent_dim, h, w, n_future)
BEVERSE_CASES = {
    "config1_c32": (32, 16, 50, 50, 4),     # BASELINE config 1: FuturePrediction(32, 16, 3, 3) at 50x50
    "odd_c16": (16, 8, 13, 21, 2),          # odd sizes: zero-pad + ceil pooling of the skip path
}


# ---- camera lift-splat ------------------------------------------------------------
# tag -> b, s, n_cam, D, fH, fW, C, X_BOUND, Y_BOUND, Z_BOUND, discount
LIFT_CASES = {
    "tiny":     (2, 3, 2, 5, 3, 4, 8, (-4.0, 4.0, 0.5), (-4.0, 4.0, 0.5), (-10.0, 10.0, 20.0), 0.5),
    "wide_c64": (1, 3, 3, 6, 4, 7, 64, (-6.0, 6.0, 0.5), (-5.0, 5.0, 0.5), (-10.0, 10.0, 20.0), 0.5),
    "one_frame_c20": (1, 1, 2, 4, 3, 5, 20, (-3.0, 3.0, 0.25), (-2.0, 2.0, 0.5), (-10.0, 10.0, 20.0), 0.9),
}
# bev_pool alone also with several height slices (projection_to_birds_eye_view needs Z == 1)
LIFT_POOL_CASES = {
    "z4_b2": (2, 2, 4, 3, 5, 16, (-3.0, 3.0, 0.5), (-2.0, 2.0, 0.25), (-2.0, 2.0, 1.0)),   # B, N, D, fH, fW, C, bounds
    "z1_b1": (1, 3, 5, 4, 4, 8, (-4.0, 4.0, 0.5), (-4.0, 4.0, 0.5), (-10.0, 10.0, 20.0)),
    "empty": (1, 1, 2, 2, 2, 8, (-1.0, 1.0, 0.5), (-1.0, 1.0, 0.5), (-10.0, 10.0, 20.0)),     # every point outside
}
LIFT_MARGIN = 2e-3      # in cells: no test point closer than this to a cell boundary (see safe_geometry)


def lift_bounds(xb, yb, zb):
    from . import lift_splat as LS
    return LS.calculate_birds_eye_view_parameters(list(xb), list(yb), list(zb))


def safe_geometry(geo, final_of, lo, res):
    """Nudge points whose *final* (float64-evaluated) cell coordinate lies within LIFT_MARGIN of a cell
    boundary, so that fp32 rounding differences between devices in the ego-motion matmul cannot move
    a test point into a neighbouring cell.  ``final_of(geo64) -> final positions`` (same shape)."""
    geo = geo.clone()
    for _ in range(50):
        q = (final_of(geo.double()) - lo.double()) / res.double()
        bad = (q - q.round()).abs() < LIFT_MARGIN
        if not bool(bad.any()):
            return geo
        geo = torch.where(bad, geo + 0.037 * res, geo)
    raise RuntimeError("could not move the test geometry off the cell boundaries")


def lift_pool_inputs(tag):
    """Inputs of ``streamingflow.bev_pool``: (geom_feats [B,N,D,H,W,3], x [B,N,D,H,W,C], start, res, dim)."""
    B, N, D, fH, fW, C, xb, yb, zb = LIFT_POOL_CASES[tag]
    res, start, dim = lift_bounds(xb, yb, zb)
    lo = start - res / 2.0
    span = torch.tensor([xb[1] - xb[0], yb[1] - yb[0], zb[1] - zb[0]])
    u = hashfill.uniform("lift_pool_geo_" + tag, (B, N, D, fH, fW, 3), 0.0, 1.0, seed=21)
    if tag == "empty":
        geo = lo + span * (1.5 + u)                     # all beyond the far corner
    else:
        geo = lo + span * (u * 1.3 - 0.15)              # ~23 % outside, some in (-1, 0) cells (truncate to 0)
    geo = safe_geometry(geo, lambda g: g, lo, res)
    x = hashfill.normal("lift_pool_x_" + tag, (B, N, D, fH, fW, C), seed=22)
    return geo, x, start, res, dim


def lift_inputs(tag):
    """Inputs of the camera branch after the image encoder: feat [b,s,n,C,fH,fW], depth logits
    [b,s,n,D,fH,fW], geometry [b,s,n,D,fH,fW,3], future_egomotion [b,s,6], (start, res, dim), discount."""
    from . import lift_splat as LS
    b, s, n, D, fH, fW, C, xb, yb, zb, discount = LIFT_CASES[tag]
    res, start, dim = lift_bounds(xb, yb, zb)
    lo = start - res / 2.0
    span = torch.tensor([xb[1] - xb[0], yb[1] - yb[0], zb[1] - zb[0]])
    u = hashfill.uniform("lift_geo_" + tag, (b, s, n, D, fH, fW, 3), 0.0, 1.0, seed=23)
    geo = lo + span * (u * 1.3 - 0.15)
    ego = torch.cat([hashfill.uniform("lift_ego_t_" + tag, (b, s, 3), -0.8, 0.8, seed=24),
                     hashfill.uniform("lift_ego_r_" + tag, (b, s, 3), -0.08, 0.08, seed=25)], -1)
    mat = LS.pose_vec2mat(ego.double())

    def final_of(g):
        return torch.stack([LS.warp_geometry(g[i], mat[i, :, :3, :3], mat[i, :, :3, 3]) for i in range(b)])
    geo = safe_geometry(geo, final_of, lo, res)
    feat = hashfill.normal("lift_feat_" + tag, (b, s, n, C, fH, fW), seed=26)
    depth = hashfill.normal("lift_depth_" + tag, (b, s, n, D, fH, fW), seed=27, std=2.0)
    return feat, depth, geo, ego, (start, res, dim), discount


# fused path: the camera rig itself is the input.  tag -> b, s, n_cam, C, final_dim, downsample, D_BOUND, bounds, discount
LIFT_RIG_CASES = {
    "rig_small": (2, 3, 2, 8, (32, 48), 8, (2.0, 8.0, 1.0), (-6.0, 6.0, 0.5), (-6.0, 6.0, 0.5), (-10.0, 10.0, 20.0), 0.5),
    "rig_c64":   (1, 2, 3, 64, (24, 40), 8, (1.0, 7.0, 0.5), (-5.0, 5.0, 0.25), (-5.0, 5.0, 0.5), (-10.0, 10.0, 20.0), 0.7),
    "e2e_c16":   (1, 3, 2, 16, (32, 48), 8, (2.0, 8.0, 1.0), (-4.0, 4.0, 0.5), (-4.0, 4.0, 0.5), (-10.0, 10.0, 20.0), 0.5),
}


def lift_rig_inputs(tag):
    """feat [b,s,n,C,fH,fW], depth logits [b,s,n,D,fH,fW], intrinsics [b,s,n,3,3], extrinsics [b,s,n,4,4],
    future_egomotion [b,s,6], frustum [D,fH,fW,3], (start, res, dim), discount.  The rig is shifted
    (deterministically) until no frustum point ends within LIFT_MARGIN/4 cells of a cell boundary, so
    the fp32 rounding of the composed transform cannot change a cell."""
    from . import lift_splat as LS
    b, s, n, C, final_dim, down, d_bound, xb, yb, zb, discount = LIFT_RIG_CASES[tag]
    res, start, dim = lift_bounds(xb, yb, zb)
    lo = start - res / 2.0
    fr = LS.create_frustum(final_dim, down, list(d_bound))
    D, fH, fW, _ = fr.shape
    H, W = final_dim
    f = 0.9 * W
    intr = torch.tensor([[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]]).repeat(b, s, n, 1, 1)
    intr = intr * (1.0 + 0.05 * hashfill.uniform("rig_intr_" + tag, (b, s, n, 1, 1), -1.0, 1.0, seed=41))
    intr[..., 2, 2] = 1.0
    # cameras look along +x / -x / +y ... : camera z (depth) -> ego x, camera x -> ego -y, camera y -> ego -z
    base = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    ego = torch.cat([hashfill.uniform("rig_ego_t_" + tag, (b, s, 3), -0.6, 0.6, seed=44),
                     hashfill.uniform("rig_ego_r_" + tag, (b, s, 3), -0.06, 0.06, seed=45)], -1)
    for attempt in range(200):
        yaw = hashfill.uniform("rig_yaw_%s_%d" % (tag, attempt), (b, s, n), -3.1, 3.1, seed=42)
        ang = torch.stack([torch.zeros_like(yaw), torch.zeros_like(yaw), yaw], -1)
        rot = LS.euler2mat(ang).matmul(base)
        trans = hashfill.uniform("rig_t_%s_%d" % (tag, attempt), (b, s, n, 3), -0.5, 0.5, seed=43)
        extr = torch.zeros(b, s, n, 4, 4)
        extr[..., :3, :3], extr[..., :3, 3], extr[..., 3, 3] = rot, trans, 1.0
        g = LS.get_geometry(fr.double(), intr.double().view(b * s, n, 3, 3), extr.double().view(b * s, n, 4, 4)).view(b, s, n, D, fH, fW, 3)
        mat = LS.pose_vec2mat(ego.double())
        fin = torch.stack([LS.warp_geometry(g[i], mat[i, :, :3, :3], mat[i, :, :3, 3]) for i in range(b)])
        q = (fin - lo.double()) / res.double()
        if float((q - q.round()).abs().min()) >= LIFT_MARGIN / 4:
            break
    else:
        raise RuntimeError("no safe rig found")
    feat = hashfill.normal("rig_feat_" + tag, (b, s, n, C, fH, fW), seed=46)
    depth = hashfill.normal("rig_depth_" + tag, (b, s, n, D, fH, fW), seed=47, std=2.0)
    return feat, depth, intr, extr, ego, fr, (start, res, dim), discount


# ---- LiDAR hard voxelisation --------------------------------------------------------
# tag -> n_points, F, voxel_size, point_cloud_range, max_points, max_voxels.  Cubic grids: the reference's
# CPU kernel (the fixture generator) is only memory-safe there (see oracle/voxelize.py).
VOXEL_CASES = {
    "cube16":      (3000, 5, (0.5, 0.5, 0.5), (-4.0, -4.0, -4.0, 4.0, 4.0, 4.0), 3, 200),
    "cube8_dense": (5000, 4, (1.0, 1.0, 1.0), (-4.0, -4.0, -4.0, 4.0, 4.0, 4.0), 10, 5000),
    "cap50":       (2000, 5, (0.5, 0.5, 0.5), (-4.0, -4.0, -4.0, 4.0, 4.0, 4.0), 2, 50),
    "one_point":   (1, 3, (0.5, 0.5, 0.5), (-4.0, -4.0, -4.0, 4.0, 4.0, 4.0), 4, 10),
}
# VOXEL_SHIPPED (the shipped configuration, streamingflow.py:111) lives in workloads.synthetic


def voxel_points(tag):
    n, F, vs, rng, mp, mv = VOXEL_CASES[tag]
    p = hashfill.uniform("voxel_pts_" + tag, (n, F), -5.0, 5.0, seed=51)
    if n >= 16:      # points exactly on cell / range boundaries: floor() and the open upper bound
        edge = torch.tensor([-4.0, 4.0, 0.0, 0.5, -0.5, 3.5, 3.9999998, -4.0000005])
        p[:8, 0] = edge
        p[8:16, 1] = edge
        p[4:12, 2] = edge
    return p


# ---- BEV Decoder --------------------------------------------------------------------------
# tag -> in_channels, n_classes, n_present, n_hdmap, predict_gate, (b, s, h, w)
DECODER_CASES = {
    "shipped_gates_small": (64, 2, 3, 2, dict(perceive_hdmap=False, predict_pedestrian=False, predict_instance=True,
                                               predict_future_flow=True, planning=False), (1, 3, 24, 32)),
    "all_heads_c32": (32, 2, 2, 2, dict(perceive_hdmap=True, predict_pedestrian=True, predict_instance=True,
                                         predict_future_flow=True, planning=True), (2, 2, 16, 16)),
}


# ---- TemporalModel --------------------------------------------------------------------------
# tag -> in_channels, receptive_field, start_out_channels, extra_in_channels, inbetween, pyramid, (b, s, h, w)
TEMPORAL_CASES = {
    "camera_c70": (70, 3, 64, 0, 0, True, (1, 3, 24, 16)),        # 64 + 6 ego-pose channels (config.py:141)
    "lidar_c256_b2": (256, 3, 64, 0, 0, True, (2, 3, 16, 16)),
    "rf2_nopool_c16": (16, 2, 16, 0, 0, False, (1, 2, 12, 20)),
    "inbetween2_c24": (24, 3, 16, 0, 2, True, (1, 3, 12, 12)),      # Bottleneck3D layers between the temporal blocks
}


# ---- LiDAR SparseEncoder (second half) -------------------------------------------------------
# tag -> cfg overrides, n_voxels per sample, batch.  Small grids so that the dense formulation fits.
SPARSE_CASES = {
    "grid32x24x27": (dict(sparse_shape=[32, 24, 27]), 260, 2),
    "thin_c8": (dict(sparse_shape=[24, 24, 41], base_channels=8, output_channels=16,
                     encoder_channels=[[8, 8, 16], [16, 16, 16], [16, 16, 32], [32, 32]]), 150, 1),
    # the class-default layout (block_type='conv_module': plain conv modules, each later stage opens with the strided conv)
    "conv_module": (dict(sparse_shape=[24, 24, 41], block_type="conv_module", base_channels=16, output_channels=32,
                         encoder_channels=[[16], [32, 32, 32], [64, 64, 64], [64, 64, 64]],
                         encoder_paddings=[[1], [1, 1, 1], [1, 1, 1], [[0, 1, 1], 1, 1]]), 140, 1),
    # the shipped channel widths / depth (41 z cells) on an x-y crop of the shipped 1600 x 1600 grid that the dense
    # conv3d formulation can still hold: 12 000 voxels in two samples
    "crop96_shipped_widths": (dict(sparse_shape=[96, 96, 41]), 6000, 2),
}


def sparse_cfg(tag):
    from . import sparse_encoder_ref as SR
    cfg = SR.default_cfg()
    cfg.update(SPARSE_CASES[tag][0])
    return cfg


def sparse_inputs(tag):
    """(voxel_features [N, Cin] f32, coors [N, 4] int32 (batch, x, y, z), batch_size): unique random sites,
    clustered so that neighbourhoods are populated."""
    cfg = sparse_cfg(tag)
    _, n, B = SPARSE_CASES[tag]
    X, Y, Z = cfg["sparse_shape"]
    feats, coords = [], []
    for b in range(B):
        u = hashfill.uniform(f"sp_sites_{tag}_{b}", (4 * n, 3), 0.0, 1.0, seed=81)
        c = (u * torch.tensor([X * 0.6, Y * 0.6, Z * 0.8])).long() + torch.tensor([int(X * 0.2), int(Y * 0.2), 0])
        key = (c[:, 0] * Y + c[:, 1]) * Z + c[:, 2]
        seen, keep = set(), []
        for i, k in enumerate(key.tolist()):
            if k not in seen:
                seen.add(k)
                keep.append(i)
            if len(keep) == n:
                break
        c = c[keep]
        coords.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.long), c], 1))
        feats.append(hashfill.normal(f"sp_feats_{tag}_{b}", (c.shape[0], cfg["in_channels"]), seed=82))
    return torch.cat(feats, 0), torch.cat(coords, 0).int(), B


def sparse_state_dict(tag, seed=83):
    from . import sparse_encoder_ref as SR
    shapes = SR.state_dict_shapes(sparse_cfg(tag))
    return hashfill.fill_state_dict({k: torch.empty(v) if v else torch.tensor(0) for k, v in shapes.items()}, seed=seed, gain=1.6)


# ---- evaluation harness: synthetic decoder outputs with a few moving blobs -------------------
def eval_scene(seed=0, b=1, s=4, h=48, w=40, n_obj=5):
    """(output dict like Decoder.forward's, labels dict): Gaussian centre heat maps, offsets pointing at the
    centres, a small constant flow, ground-truth instance ids / segmentation slightly perturbed."""
    g = torch.Generator().manual_seed(100 + seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float), torch.arange(w, dtype=torch.float), indexing="ij")
    seg = torch.zeros(b, s, 2, h, w)
    center = torch.zeros(b, s, 1, h, w)
    offset = torch.zeros(b, s, 2, h, w)
    flow = torch.zeros(b, s, 2, h, w)
    gt_inst = torch.zeros(b, s, h, w, dtype=torch.long)
    for bi in range(b):
        pos = torch.stack([torch.rand(n_obj, generator=g) * (h - 16) + 8, torch.rand(n_obj, generator=g) * (w - 16) + 8], 1)
        vel = (torch.rand(n_obj, 2, generator=g) - 0.5) * 3.0
        for t in range(s):
            fg = torch.zeros(h, w, dtype=torch.bool)
            for k in range(n_obj):
                cy, cx = pos[k] + vel[k] * t
                m = ((yy - cy).abs() <= 2.5) & ((xx - cx).abs() <= 1.5)
                fg |= m
                gt_inst[bi, t][((yy - cy - 0.4).abs() <= 2.5) & ((xx - cx + 0.3).abs() <= 1.5)] = k + 1
                center[bi, t, 0] = torch.maximum(center[bi, t, 0], torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 6.0))
                offset[bi, t, 0][m] = (cy.round() - yy)[m]
                offset[bi, t, 1][m] = (cx.round() - xx)[m]
                flow[bi, t, 0][m] = vel[k, 0]
                flow[bi, t, 1][m] = vel[k, 1]
            seg[bi, t, 1][fg] = 4.0
            seg[bi, t, 0][~fg] = 4.0
    noise = torch.randn(center.shape, generator=g) * 0.01
    out = {"segmentation": seg, "instance_center": (center + noise).clamp(0, 1), "instance_offset": offset, "instance_flow": flow}
    labels = {"segmentation": (gt_inst > 0).long().unsqueeze(2), "instance": gt_inst}
    return out, labels


def label_batch(seed=0, b=2, T=7, h=64, w=64):
    """A dataset-like batch: label sequences of the eval_scene kind + ego-motion + depth maps."""
    g = torch.Generator().manual_seed(300 + seed)
    out, labels = eval_scene(seed, b=b, s=T, h=h, w=w, n_obj=6)
    ego = torch.cat([(torch.rand((b, T, 3), generator=g) - 0.5) * torch.tensor([6.0, 4.0, 0.0]),
                     (torch.rand((b, T, 3), generator=g) - 0.5) * torch.tensor([0.0, 0.0, 0.12])], -1)
    return {"segmentation": labels["segmentation"], "instance": labels["instance"], "centerness": out["instance_center"],
            "offset": out["instance_offset"], "flow": out["instance_flow"], "future_egomotion": ego,
            "depths": torch.rand((b, T, 2, 32, 48), generator=g) * 60.0}


# ---- a16: modules the reference defines but never constructs (tests/golden/unused_cells.npz) ------------------------
def unused_cell_inputs():
    """Hashed inputs of tests/golden/unused_cells.npz (shared by the generator and the GPU test)."""
    C, h, w = 8, 12, 12
    return {
        "x1": hashfill.normal("uc_x1", (2, 1, C, h, w), 41),
        "x1w": hashfill.normal("uc_x1w", (2, 1, 2 * C, h, w), 42),
        "st1": hashfill.normal("uc_st1", (2, 1, C, h, w), 43) * 0.5,
        "st3": hashfill.normal("uc_st3", (2, 3, C, h, w), 44) * 0.5,
        "seq": hashfill.normal("uc_seq", (2, 3, C, h, w), 45) * 0.5,
        "x1b1": hashfill.normal("uc_x1b1", (1, 1, C, h, w), 46),
        "st2b1": hashfill.normal("uc_st2b1", (1, 2, C, h, w), 47) * 0.5,
    }


UNUSED_CELL_SEEDS = {"dual_gru": 51, "dual_gru_wide": 52, "bigru": 53, "dual_ode": 54, "dual_obs": 55}


