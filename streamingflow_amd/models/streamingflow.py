"""``streamingflow`` — the whole model after the image backbone, wired from the MI355X-native parts.

Mirrors ``streamingflow.forward`` (streamingflow/models/streamingflow.py:208-269) with the reference's
attribute names (``temporal_model``, ``encoders.lidar.{voxelize,backbone}``, ``temporal_model_lidar``,
``future_prediction_ode``, ``decoder``, ``bev_resolution`` / ``bev_start_position`` / ``bev_dimension``,
``frustum``), so a released ``model.*`` checkpoint slice loads with ``strict=False``.  The one part that
is NOT here is the EfficientNet image ``Encoder`` (streamingflow/models/encoder.py; needs
``efficientnet_pytorch``, out of scope — SURVEY.md §8): assign any module returning
``(features [b*n, C, fH, fW], depth_logits [b*n, D, fH, fW])`` to ``self.encoder``, or pass that pair in
place of ``image``.

Camera:  (features, depth logits, rig) -> LiftSplat.lift_splat -> + ego-pose channels -> TemporalModel
LiDAR:   point clouds -> hard voxelisation + mean -> SparseEncoder -> TemporalModel
both  -> FuturePredictionODE -> Decoder.
"""
from types import SimpleNamespace as NS

import torch
import torch.nn as nn

from .decoder import Decoder
from .future_prediction_ode import FuturePredictionODE
from .lift_splat import LiftSplat
from .sparse_encoder import SparseEncoder
from .temporal_model import TemporalModel
from ..voxelize import Voxelization, voxelize

# the LiDAR encoder configuration hard-coded at streamingflow.py:111
LIDAR_ENCODER = {"voxelize": {"max_num_points": 10, "point_cloud_range": [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0],
                              "voxel_size": [0.0625, 0.0625, 0.2], "max_voxels": [120000, 160000]},
                 "backbone": {"in_channels": 5, "sparse_shape": [1600, 1600, 41], "output_channels": 128,
                              "order": ["conv", "norm", "act"],
                              "encoder_channels": [[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                              "encoder_paddings": [[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], "block_type": "basicblock"}}


def default_cfg(**over):
    """The configuration keys this stage reads, with the reference's defaults (streamingflow/config.py)."""
    cfg = NS(TIME_RECEPTIVE_FIELD=3, N_FUTURE_FRAMES=4,
             IMAGE=NS(FINAL_DIM=(224, 480)),
             LIFT=NS(X_BOUND=[-50.0, 50.0, 0.5], Y_BOUND=[-50.0, 50.0, 0.5], Z_BOUND=[-10.0, 10.0, 20.0], D_BOUND=[2.0, 50.0, 1.0],
                     DISCOUNT=0.5),
             MODEL=NS(ENCODER=NS(DOWNSAMPLE=8, OUT_CHANNELS=64),
                      MODALITY=NS(USE_CAMERA=True, USE_LIDAR=True),
                      TEMPORAL_MODEL=NS(NAME="temporal_block", START_OUT_CHANNELS=64, EXTRA_IN_CHANNELS=0, INBETWEEN_LAYERS=0,
                                        PYRAMID_POOLING=True, INPUT_EGOPOSE=True),
                      DISTRIBUTION=NS(LATENT_DIM=64),
                      FUTURE_PRED=NS(N_GRU_BLOCKS=2, N_RES_LAYERS=1, MIXTURE=True, DELTA_T=0.05, USE_VARIABLE_ODE_STEP=True),
                      IMPUTE=True, SOLVER="euler", SMALL_ENCODER=NS(FILTER_SIZE=64, SKIPCO=False)),
             SEMANTIC_SEG=NS(VEHICLE=NS(WEIGHTS=[1.0, 2.0]), PEDESTRIAN=NS(ENABLED=False), HDMAP=NS(ENABLED=False, ELEMENTS=["lane_divider", "drivable_area"])),
             INSTANCE_SEG=NS(ENABLED=True), INSTANCE_FLOW=NS(ENABLED=True), PLANNING=NS(ENABLED=False),
             LIDAR_ENCODER=LIDAR_ENCODER)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


class streamingflow(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.lift = LiftSplat.from_cfg(cfg)
        # the reference keeps these on the top-level module (streamingflow.py:31-33, :41)
        self.bev_resolution, self.bev_start_position, self.bev_dimension = self.lift.bev_resolution, self.lift.bev_start_position, self.lift.bev_dimension
        self.frustum = self.lift.frustum
        self.encoder_out_channels = cfg.MODEL.ENCODER.OUT_CHANNELS
        self.use_lidar, self.use_camera = cfg.MODEL.MODALITY.USE_LIDAR, cfg.MODEL.MODALITY.USE_CAMERA
        self.receptive_field, self.n_future = cfg.TIME_RECEPTIVE_FIELD, cfg.N_FUTURE_FRAMES
        self.latent_dim = cfg.MODEL.DISTRIBUTION.LATENT_DIM
        self.bev_size = (int(self.bev_dimension[0]), int(self.bev_dimension[1]))
        self.encoder = None                       # image backbone: not part of this build (see module docstring)
        tm = cfg.MODEL.TEMPORAL_MODEL
        if tm.NAME != "temporal_block":
            raise NotImplementedError("TEMPORAL_MODEL.NAME == 'temporal_block' only")
        kw = dict(input_shape=self.bev_size, start_out_channels=tm.START_OUT_CHANNELS, extra_in_channels=tm.EXTRA_IN_CHANNELS,
                  n_spatial_layers_between_temporal_layers=tm.INBETWEEN_LAYERS, use_pyramid_pooling=tm.PYRAMID_POOLING)
        if self.use_camera:
            self.temporal_model = TemporalModel(self.encoder_out_channels + (6 if tm.INPUT_EGOPOSE else 0), self.receptive_field, **kw)
        self.future_pred_in_channels = tm.START_OUT_CHANNELS
        if self.n_future > 0:
            fp = cfg.MODEL.FUTURE_PRED
            self.future_prediction_ode = FuturePredictionODE(in_channels=self.future_pred_in_channels, latent_dim=self.latent_dim,
                                                             n_future=self.n_future, cfg=cfg, mixture=fp.MIXTURE, n_gru_blocks=fp.N_GRU_BLOCKS,
                                                             n_res_layers=fp.N_RES_LAYERS, delta_t=fp.DELTA_T)
        self.decoder = Decoder(in_channels=self.future_pred_in_channels, n_classes=len(cfg.SEMANTIC_SEG.VEHICLE.WEIGHTS),
                               n_present=self.receptive_field, n_hdmap=len(cfg.SEMANTIC_SEG.HDMAP.ELEMENTS),
                               predict_gate={"perceive_hdmap": cfg.SEMANTIC_SEG.HDMAP.ENABLED,
                                             "predict_pedestrian": cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED,
                                             "predict_instance": cfg.INSTANCE_SEG.ENABLED,
                                             "predict_future_flow": cfg.INSTANCE_FLOW.ENABLED, "planning": cfg.PLANNING.ENABLED})
        if self.use_lidar:
            enc = getattr(cfg, "LIDAR_ENCODER", LIDAR_ENCODER)
            self.encoders = nn.ModuleDict({"lidar": nn.ModuleDict({
                "voxelize": Voxelization(**{k: (tuple(v) if k == "max_voxels" else v) for k, v in enc["voxelize"].items()}),
                "backbone": SparseEncoder(**enc["backbone"])})})
            self.voxelize_reduce = True
            self.lidar_channels = enc["backbone"]["output_channels"] * self._lidar_depth(enc["backbone"])
            self.temporal_model_lidar = TemporalModel(self.lidar_channels, self.receptive_field, **kw)

    @staticmethod
    def _lidar_depth(b):
        z = b["sparse_shape"][2]
        pads = b["encoder_paddings"]
        for i in range(len(b["encoder_channels"]) - 1):
            p = pads[i][-1]
            pz = p[2] if isinstance(p, (list, tuple)) else p
            z = (z + 2 * pz - 3) // 2 + 1
        return (z - 3) // 2 + 1                       # conv_out: kernel (1,1,3), stride (1,1,2)

    # ---- streamingflow.py:170-206 -------------------------------------------------------------------
    def voxelize(self, points):
        return voxelize(points, self.encoders["lidar"]["voxelize"], self.voxelize_reduce)

    def extract_lidar_features(self, x):
        feats, coords, sizes = self.voxelize(x)
        return self.encoders["lidar"]["backbone"](feats, coords, len(x))

    def calculate_birds_eye_view_features(self, image, intrinsics, extrinsics, future_egomotion):
        """streamingflow.py:430-448 minus the image backbone: ``image`` is (features [b,s,n,C,fH,fW],
        depth logits [b,s,n,D,fH,fW]) or, with ``self.encoder`` set, the images [b,s,n,3,H,W]."""
        if isinstance(image, (tuple, list)):
            feat, depth = image
        else:
            if self.encoder is None:
                raise RuntimeError("no image backbone: set `.encoder` or pass (features, depth_logits) instead of images")
            b, s, n = image.shape[:3]
            feat, depth = self.encoder(image.reshape(b * s * n, *image.shape[3:]))
            feat, depth = feat.view(b, s, n, *feat.shape[1:]), depth.view(b, s, n, *depth.shape[1:])
        x = self.lift.lift_splat(feat, depth, intrinsics, extrinsics, future_egomotion)
        return x, depth, None

    def forward(self, image, intrinsics, extrinsics, future_egomotion, padded_voxel_points=None, camera_timestamp=None, points=None,
                lidar_timestamp=None, target_timestamp=None):
        output = {}
        future_egomotion = future_egomotion[:, : self.receptive_field].contiguous()
        camera_states = lidar_states = states = None
        if self.use_lidar:
            pts = torch.stack(points).permute(1, 0, 2, 3)                       # B, T, num_point, C
            B, T, num_point, C = pts.shape
            pts = pts.contiguous().view(B * T, num_point, C).to(torch.float32)
            feature = self.extract_lidar_features([pts[i] for i in range(pts.shape[0])])
            _, C, H_det, W_det = feature.shape
            lidar_states = self.temporal_model_lidar(feature.view(B, T, C, H_det, W_det))
            states = lidar_states
        if self.use_camera:
            rf = self.receptive_field
            img = tuple(t[:, :rf].contiguous() for t in image) if isinstance(image, (tuple, list)) else image[:, :rf].contiguous()
            x, depth, cam_front = self.calculate_birds_eye_view_features(img, intrinsics[:, :rf].contiguous(), extrinsics[:, :rf].contiguous(),
                                                                         future_egomotion)
            output = {**output, "depth_prediction": depth, "cam_front": cam_front}
            if self.cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE:
                b, s, c = future_egomotion.shape
                h, w = x.shape[-2:]
                ego = future_egomotion.view(b, s, c, 1, 1).expand(b, s, c, h, w)
                ego = torch.cat([torch.zeros_like(ego[:, :1]), ego[:, : (rf - 1)]], dim=1)
                x = torch.cat([x, ego], dim=-3)
            camera_states = self.temporal_model(x)
            states = camera_states
        if self.n_future > 0:
            present_state = states[:, -1:].contiguous()
            states, _ = self.future_prediction_ode(present_state, camera_states, lidar_states, camera_timestamp, lidar_timestamp,
                                                   target_timestamp)
        bev_output = self.decoder(states)
        return {**output, **bev_output}
