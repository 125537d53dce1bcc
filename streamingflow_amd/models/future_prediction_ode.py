"""MI355X-native ``FuturePredictionODE`` (streamingflow/models/future_prediction_ode.py:9-64):
GRU-ODE temporal propagator over camera/LiDAR BEV states followed by the spatial-GRU head.

Drop-in for the reference module (same constructor, ``forward`` signature, return value and
``state_dict`` keys — a ``model.future_prediction_ode.*`` checkpoint slice loads unchanged); the
arithmetic runs on libsfnative (HIP, gfx950).  Internally everything is NHWC fp32; the NCHW
reference layout exists only at ``forward``'s boundary.
"""
import torch
import torch.nn as nn

from .. import runtime, schedule as sched
from ..layers.convolutions import Block, DeepLabHead
from ..layers.temporal import SpatialGRU
from ..layers.temporal_ode_bayes import NNFOwithBayesianJumps


class FuturePredictionODE(nn.Module):
    def __init__(self, in_channels, latent_dim, n_future, cfg, mixture=True, n_gru_blocks=2, n_res_layers=1,
                 delta_t=0.05):
        super().__init__()
        self.n_spatial_gru = n_gru_blocks
        self.delta_t = delta_t
        self.gru_ode = NNFOwithBayesianJumps(input_size=in_channels, hidden_size=latent_dim, cfg=cfg,
                                             mixing=int(mixture))
        grus, blocks = [], []
        for i in range(n_gru_blocks):
            grus.append(SpatialGRU(in_channels, in_channels))
            last = i == n_gru_blocks - 1
            blocks.append(DeepLabHead(in_channels, in_channels, 128) if last else
                          nn.Sequential(*[Block(in_channels) for _ in range(n_res_layers)]))
        self.spatial_grus = nn.ModuleList(grus)
        self.res_blocks = nn.ModuleList(blocks)

    def observations(self, camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs):
        """Merge + time-sort one sample's observations (:36-49); returns (times, [n_obs,H,W,C])."""
        cam_ts = camera_timestamp[bs].tolist() if camera_states is not None else []
        lid_ts = lidar_timestamp[bs].tolist() if lidar_states is not None else []
        times, order = sched.merge_observations(cam_ts, lid_ts)
        frames = [(camera_states if src == 0 else lidar_states)[bs, i] for src, i in order]
        return times, runtime.to_nhwc(torch.stack(frames, dim=0))

    def head_nhwc(self, x):
        """x: [T, H, W, C] decoded predictions of one sample -> [T, H, W, C] (:56-62)."""
        hidden = x[0]
        for gru, blk in zip(self.spatial_grus, self.res_blocks):
            x = gru.forward_nhwc(x, hidden)
            if isinstance(blk, DeepLabHead):
                x = blk.forward_nhwc(x)
            else:
                for b in blk:
                    x = b.forward_nhwc(x)
        return x

    def forward(self, future_prediction_input, camera_states, lidar_states, camera_timestamp, lidar_timestamp,
                target_timestamp):
        some = camera_states if camera_states is not None else lidar_states
        runtime.require_cuda(some)
        outs = []
        for bs in range(some.shape[0]):
            times, obs = self.observations(camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs)
            _, x, _ = self.gru_ode.forward_nhwc(times, obs, self.delta_t, target_timestamp[bs].tolist())
            outs.append(runtime.to_nchw(self.head_nhwc(x)))
        return torch.stack(outs, dim=0), 0
