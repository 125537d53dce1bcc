"""MI355X-native ``FuturePredictionODE`` (streamingflow/models/future_prediction_ode.py:9-64):
GRU-ODE temporal propagator over camera/LiDAR BEV states followed by the spatial-GRU head.

Drop-in for the reference module (same constructor, ``forward`` signature, return value and
``state_dict`` keys — a ``model.future_prediction_ode.*`` checkpoint slice loads unchanged); the
arithmetic runs on libsfnative (HIP, gfx950).  Internally everything is NHWC fp32; the NCHW
reference layout exists only at ``forward``'s boundary.
"""
import os

import torch
import torch.nn as nn

from .. import _lib, runtime, schedule as sched
from ..layers.convolutions import Block, DeepLabHead
from ..layers.temporal import SpatialGRU
from ..layers.temporal_ode_bayes import NNFOwithBayesianJumps


# SF_HEAD_PLANAR=0: the head writes [pixel][channel] frames and T strided transposes follow (the form before round 5; A/B aid)
_HEAD_PLANAR = os.environ.get("SF_HEAD_PLANAR", "1") != "0"


# SF_FOLD_DECODER=0: the last SpatialGRU runs its 1x1 conv_decoder on every frame (T launches) instead of handing its hidden states
# to a DeepLabHead packed with the decoder folded into its input convolutions (A/B aid)
_FOLD_DECODER = os.environ.get("SF_FOLD_DECODER", "1") != "0"


class _FoldedTail(runtime.PackedModule):
    """Packed copies for the last (SpatialGRU, DeepLabHead) pair of the head: the GRU without its decoder and the head with
    ``conv_decoder`` composed into its branch and pooling weights (DeepLabHead.pack_after).  Holds no parameters of its own — the two
    modules are referenced, not registered — and re-packs when any of theirs changes."""

    def __init__(self, gru, head):
        super().__init__()
        self.__dict__["_gru"], self.__dict__["_head"] = gru, head

    def _param_signature(self):
        from .. import packing
        sig = [packing.math_mode(), packing.winograd(), self._head.training]
        for m in (self._gru, self._head):
            for t in list(m.parameters()) + list(m.buffers()):
                sig.append((t.data_ptr(), t._version, t.device))
        return tuple(sig)

    def packed(self):
        sig = self._param_signature()
        cache = self.__dict__.get("_sf_pack")
        if cache is None or cache[0] != sig:
            runtime.require_cuda(self._gru.conv_decoder.weight)
            with torch.no_grad():
                pk = (self._gru.pack_states_only(), self._head.pack_after(self._gru.conv_decoder.weight))
            self.__dict__["_sf_pack"] = cache = (sig, pk, 0)
        return cache[1]


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

    def __getstate__(self):
        return runtime.strip_runtime_state(self.__dict__)      # (the folded tail holds device pointers: rebuilt on the next forward)

    def observations(self, camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs, with_order=False):
        """Merge + time-sort one sample's observations (:36-49); returns (times, [frames NCHW])
        (+ the (source tensor, frame index) of every observation with ``with_order``)."""
        cam_ts = camera_timestamp[bs].tolist() if camera_states is not None else []
        lid_ts = lidar_timestamp[bs].tolist() if lidar_states is not None else []
        times, order = sched.merge_observations(cam_ts, lid_ts)
        frames = [(camera_states if src == 0 else lidar_states)[bs, i] for src, i in order]
        return (times, frames, order) if with_order else (times, frames)

    def head_nhwc(self, x, res=None):
        """x: [T, B, H, W, C] decoded predictions -> [T, B, H, W, C] (:56-62).  With ``res`` (a [B, T, C, H, W] tensor) the last
        block's classifier writes frame (t, b) straight into res[b, t] in the reference's layout and None is returned."""
        T, B, H, W, C = x.shape
        hidden = x[0]
        last = len(self.spatial_grus) - 1
        for i, (gru, blk) in enumerate(zip(self.spatial_grus, self.res_blocks)):
            gst = hst = None
            if (_FOLD_DECODER and i == last and isinstance(blk, DeepLabHead) and gru.conv_decoder.bias is None
                    and gru.input_size == blk.in_channels):
                # the decoder of the last GRU is linear and only the head reads it: composed into the head's input weights
                tail = self.__dict__.get("_folded_tail")
                if tail is None or tail._gru is not gru or tail._head is not blk:
                    tail = self.__dict__["_folded_tail"] = _FoldedTail(gru, blk)
                gpk, hpk = tail.packed()
                gst, hst = gpk.struct, hpk.struct
            x = gru.forward_nhwc(x, hidden, gst)
            x = x.view(T * B, H, W, x.shape[-1])
            if isinstance(blk, DeepLabHead):
                if res is not None and i == last:
                    blk.forward_nhwc_into_planar(x, res, B, C * H * W, T * C * H * W, hst)
                    return None
                x = blk.forward_nhwc(x, hst)
            else:
                for b in blk:
                    x = b.forward_nhwc(x)
            x = x.view(T, B, H, W, C)
        return x

    def _gather_obs(self, frames, sources, states):
        """[n_obs, B, H, W, C] NHWC from the per-sample observation frames.  When every sample of the group takes
        observation o from the same (tensor, frame index) and the group is a run of consecutive samples — the
        normal case — each observation is ONE strided transpose straight out of camera_states / lidar_states."""
        B, n_obs = len(frames), len(frames[0])
        src0 = sources[0][0]
        same = all(s[0] == src0 for s in sources) and all(s[1] == sources[0][1] + k for k, s in enumerate(sources))
        ok = same and all(states[src] is not None and states[src].is_contiguous() and states[src].dtype == torch.float32
                          for src, _ in src0)
        if not ok:
            stacked = torch.stack([frames[b][o] for o in range(n_obs) for b in range(B)], dim=0)
            obs = runtime.to_nhwc(stacked)
            return obs.view(n_obs, B, *obs.shape[1:])
        _, C, H, W = states[src0[0][0]].shape[1:]
        dev = states[src0[0][0]].device
        obs = torch.empty((n_obs, B, H, W, C), dtype=torch.float32, device=dev)
        L = _lib.lib()
        b0 = sources[0][1]
        for o, (src, i) in enumerate(src0):
            t = states[src]
            first = t[b0, i]
            _lib.check(L.sf_nchw_to_nhwc_strided(runtime.ptr(first), t.shape[1] * C * H * W, runtime.ptr(obs[o]), C * H * W, B, C, H * W,
                                                 runtime.stream_ptr(dev)), "nchw_to_nhwc_strided")
        return obs

    def _run_group(self, frames, scs, sources=None, states=None, out=None, members=None):
        """frames[b][o]: NCHW observation o of sample b (time order); scs: one Schedule per sample,
        all with the same structure.  Returns [B, T, C, H, W] (written into out[members] when the group is a run of
        consecutive samples of the final tensor)."""
        B = len(frames)
        obs = self._gather_obs(frames, sources, states) if sources is not None else None
        if obs is None:
            n_obs = len(frames[0])
            stacked = torch.stack([frames[b][o] for o in range(n_obs) for b in range(B)], dim=0)
            obs = runtime.to_nhwc(stacked)
            obs = obs.view(n_obs, B, *obs.shape[1:])
        _, x = self.gru_ode.forward_nhwc(scs if B > 1 else scs[0], obs)
        if (out is True and _HEAD_PLANAR and isinstance(self.res_blocks[-1], DeepLabHead)
                and self.res_blocks[-1][4].out_channels == x.shape[-1]):
            # the group is the whole batch, in order: the head's classifier writes [B, T, C, H, W] itself
            T, _, H, W, C = x.shape
            res = torch.empty((B, T, C, H, W), dtype=torch.float32, device=x.device)
            self.head_nhwc(x, res)
            return res
        y = self.head_nhwc(x)                                   # [T, B, H, W, C]
        T, _, H, W, C = y.shape
        if out is True:          # the group is the whole batch, in order: write [B, T, C, H, W] directly
            res = torch.empty((B, T, C, H, W), dtype=torch.float32, device=y.device)
            L = _lib.lib()
            for t in range(T):                                  # frame (t, b) -> res[b, t]: one strided transpose per t
                _lib.check(L.sf_nhwc_to_nchw_strided(runtime.ptr(y[t]), H * W * C, runtime.ptr(res[0, t]), T * C * H * W, B, C,
                                                     H * W, runtime.stream_ptr(y.device)), "nhwc_to_nchw_strided")
            return res
        res = runtime.to_nchw(y.view(T * B, H, W, C))
        return res.view(T, B, *res.shape[1:]).permute(1, 0, 2, 3, 4)

    def forward(self, future_prediction_input, camera_states, lidar_states, camera_timestamp, lidar_timestamp,
                target_timestamp):
        """Same contract as the reference.  Samples whose schedules have the same structure (same
        sequence of jumps / steps and the same target selection — the normal case within a batch)
        are pushed through the encoder, the rollout and the head together; step sizes may differ
        per sample.  Samples with a different structure are processed in their own group."""
        some = camera_states if camera_states is not None else lidar_states
        runtime.require_cuda(some)
        runtime.require_no_grad(future_prediction_input, camera_states, lidar_states)
        b = some.shape[0]
        groups, meta = {}, []
        states = (camera_states, lidar_states)
        for bs in range(b):
            times, frames, order = self.observations(camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs, with_order=True)
            sc = self.gru_ode.make_schedule(times, self.delta_t, target_timestamp[bs].tolist())
            groups.setdefault(sc.key(), []).append(bs)
            meta.append((frames, sc, (tuple(order), bs)))
        outs = [None] * b
        MAX_GROUP = 64       # libsfnative sizes its per-image SE scratch for 64 images per call
        chunks = [m[i:i + MAX_GROUP] for m in groups.values() for i in range(0, len(m), MAX_GROUP)]
        for members in chunks:
            fr, scs, srcs = [meta[i][0] for i in members], [meta[i][1] for i in members], [meta[i][2] for i in members]
            whole = len(chunks) == 1 and members == list(range(b))
            y = self._run_group(fr, scs, srcs, states, True if whole else None, members)
            if whole:
                return y, 0
            for k, i in enumerate(members):
                outs[i] = y[k]
        return torch.stack(outs, dim=0), 0
