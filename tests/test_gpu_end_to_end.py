"""GPU: the whole model after the image backbone (streamingflow_amd.models.streamingflow) against the chain of
oracles: lift-splat -> TemporalModel, voxelise -> SparseEncoder -> TemporalModel, FuturePredictionODE, Decoder.
Tolerance 1e-3 max-abs on the decoder outputs (north star)."""
from types import SimpleNamespace as NS

import pytest
import torch

from util import cases, hashfill, maxabs

pytestmark = pytest.mark.gpu


def small_cfg():
    from streamingflow_amd.models.streamingflow import default_cfg
    C = 16
    lidar = {"voxelize": {"max_num_points": 10, "point_cloud_range": [-4.0, -4.0, -5.0, 4.0, 4.0, 3.0], "voxel_size": [0.0625, 0.0625, 0.2],
                          "max_voxels": [2000, 3000]},
             "backbone": {"in_channels": 5, "sparse_shape": [128, 128, 41], "output_channels": 16, "base_channels": 8,
                          "order": ["conv", "norm", "act"], "encoder_channels": [[8, 8, 16], [16, 16, 16], [16, 16, 32], [32, 32]],
                          "encoder_paddings": [[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], "block_type": "basicblock"}}
    cfg = default_cfg(LIDAR_ENCODER=lidar)
    cfg.IMAGE.FINAL_DIM = (32, 48)
    cfg.LIFT = NS(X_BOUND=[-4.0, 4.0, 0.5], Y_BOUND=[-4.0, 4.0, 0.5], Z_BOUND=[-10.0, 10.0, 20.0], D_BOUND=[2.0, 8.0, 1.0], DISCOUNT=0.5)
    cfg.MODEL.ENCODER.OUT_CHANNELS = C
    cfg.MODEL.TEMPORAL_MODEL.START_OUT_CHANNELS = C
    cfg.MODEL.DISTRIBUTION.LATENT_DIM = C
    cfg.MODEL.SMALL_ENCODER.FILTER_SIZE = C
    return cfg, lidar


def test_whole_model_after_the_image_backbone():
    from streamingflow_amd.models.streamingflow import streamingflow
    from oracle import decoder_ref as DR, lift_splat as LS, ref_torch as R, sparse_encoder_ref as SR, temporal_model_ref as TR, voxelize as VZ
    cfg, lidar = small_cfg()
    net = streamingflow(cfg).eval()
    sd = hashfill.fill_state_dict(net.state_dict(), seed=91, gain=0.9)
    pre = "future_prediction_ode."
    sd.update({pre + k: v for k, v in cases.fpode_state_dict({k[len(pre):]: v for k, v in net.state_dict().items() if k.startswith(pre)}).items()})
    pb = "encoders.lidar.backbone."
    sd.update(hashfill.fill_state_dict({k: v for k, v in net.state_dict().items() if k.startswith(pb)}, seed=92, gain=1.6))
    for k, v in net.state_dict().items():          # grid parameters / frustum are geometry, not weights
        if k.startswith(("bev_", "lift.", "frustum")):
            sd[k] = v
    net.load_state_dict(sd)
    net = net.cuda()
    net.future_prediction_ode.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED)

    feat, depth, intr, extr, ego, fr, (start, res, dim), discount = cases.lift_rig_inputs("e2e_c16")
    b, s, n, C, fH, fW = feat.shape
    T = 2
    pts = [torch.cat([hashfill.uniform(f"e2e_pts_{t}", (1, 600, 3), -1.0, 1.0, seed=93) * torch.tensor([4.4, 4.4, 3.0]) + torch.tensor([0.0, 0.0, -1.0]),
                      hashfill.uniform(f"e2e_ptf_{t}", (1, 600, 2), 0.0, 1.0, seed=94)], -1) for t in range(T)]
    cts = torch.tensor([[-1.0, -0.5, 0.0]], dtype=torch.float64)
    lts = torch.tensor([[-0.3, 0.0]], dtype=torch.float64)
    tts = torch.tensor([[0.0, 0.5, 1.0]], dtype=torch.float64)
    out = net((feat.cuda(), depth.cuda()), intr.cuda(), extr.cuda(), ego.cuda(), None, cts, [p.cuda() for p in pts], lts, tts)

    # ---- the same through the oracles (CPU) -----------------------------------------------------------------------
    with torch.no_grad():
        g = LS.get_geometry(fr, intr.view(b * s, n, 3, 3), extr.view(b * s, n, 4, 4)).view(b, s, n, *fr.shape)
        x = LS.depth_outer(feat.reshape(b * s * n, C, fH, fW), depth.reshape(b * s * n, -1, fH, fW)).reshape(b, s, n, -1, fH, fW, C)
        bev = LS.projection_to_birds_eye_view(x, g, ego, start, res, dim, discount)
        egos = ego.view(b, s, 6, 1, 1).expand(b, s, 6, *bev.shape[-2:])
        egos = torch.cat([torch.zeros_like(egos[:, :1]), egos[:, : s - 1]], 1)
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        cam_states = TR.temporal_model_forward(sub("temporal_model."), torch.cat([bev, egos], 2), tuple(bev.shape[-2:]))
        vz = lidar["voxelize"]
        f, c, _ = VZ.sf_voxelize([p[0].numpy() for p in pts], vz["voxel_size"], vz["point_cloud_range"], vz["max_num_points"], vz["max_voxels"][1])
        bcfg = dict(lidar["backbone"])
        lid = SR.sparse_encoder_forward(sub("encoders.lidar.backbone."), f.numpy(), c.numpy(), T, bcfg)
        lid_states = TR.temporal_model_forward(sub("temporal_model_lidar."), lid.view(1, T, *lid.shape[1:]), tuple(lid.shape[-2:]))
        present = cam_states[:, -1:].contiguous()
        states, _ = R.future_prediction_ode_forward(sub("future_prediction_ode."), present, cam_states, lid_states, cts, lts, tts,
                                                    cfg.MODEL.FUTURE_PRED.DELTA_T, 2, "euler", True, True, hashfill.HashedNoise(cases.EPS_SEED))
        want = DR.decoder_forward(sub("decoder."), states, cfg.TIME_RECEPTIVE_FIELD)
    worst = 0.0
    for k, v in want.items():
        if v is None:
            continue
        e = maxabs(out[k], v)
        print(k, tuple(v.shape), "max-abs", e, "scale", float(v.abs().max()))
        worst = max(worst, e)
    assert worst <= 1e-3
    assert float(want["segmentation"].abs().max()) > 1e-2        # a non-trivial signal reached the heads
