"""BASELINE config 3 without the image backbone: camera features + depth logits (6 cams x 28 x 60, 3 frames) and 5 LiDAR
frames of 350 000 points -> lift-splat / voxelise + SparseEncoder -> TemporalModels -> FuturePredictionODE (2 s future) ->
Decoder.  One JSON object.  Usage: python3 tools/e2ebench.py"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(reps=3, dev=None):
    from streamingflow_amd import _lib, runtime
    from streamingflow_amd.models.streamingflow import default_cfg, streamingflow
    from workloads import hashfill, synthetic as cases
    import liftbench
    import voxelbench
    dev = dev or torch.device("cuda", 0)
    import streamingflow_amd as sfa
    mode = os.environ.get("SF_MATH_MODE")
    if mode:
        sfa.set_math_mode(mode)
    cfg = default_cfg()
    net = streamingflow(cfg).eval()
    sd = hashfill.fill_state_dict(net.state_dict(), seed=91, gain=0.9)
    pre = "future_prediction_ode."
    sd.update({pre + k: v for k, v in cases.fpode_state_dict({k[len(pre):]: v for k, v in net.state_dict().items() if k.startswith(pre)}).items()})
    for k, v in net.state_dict().items():
        if k.startswith(("bev_", "lift.", "frustum")):
            sd[k] = v
    net.load_state_dict(sd)
    net = net.to(dev)
    b, s, n, C, D, fH, fW = 1, 3, 6, 64, 48, 28, 60
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((b, s, n, C, fH, fW), generator=g).to(dev)
    logits = (torch.randn((b, s, n, D, fH, fW), generator=g) * 2).to(dev)
    intr, extr, ego = liftbench.synthetic_rig(b, s, n, dev)
    pts = [voxelbench.cloud(seed=10 + t)[None].to(dev) for t in range(5)]
    cts = torch.tensor([[-1.0, -0.5, 0.0]], dtype=torch.float64)
    lts = torch.tensor([[-0.8, -0.6, -0.4, -0.2, 0.0]], dtype=torch.float64)
    tts = torch.tensor([[-1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0]], dtype=torch.float64)

    def fwd():
        return net((feat, logits), intr, extr, ego, None, cts, pts, lts, tts)
    L = _lib.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
    ms = ctypes.c_float()
    out = fwd()
    fwd()
    torch.cuda.synchronize()
    L.sf_event_record(e0, runtime.stream_ptr(dev))
    for _ in range(reps):
        fwd()
    L.sf_event_record(e1, runtime.stream_ptr(dev))
    L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return {"workload": "config 3 minus the EfficientNet image backbone: 3 camera frames (6 x 48 x 28 x 60 frustum, C=64) + 5 LiDAR frames "
                        "x 350000 points -> 7 target frames of 200x200 decoder outputs",
            "ms_per_sample": ms.value / reps, "samples_per_s": 1e3 * reps / ms.value,
            "outputs": {k: list(v.shape) for k, v in out.items() if torch.is_tensor(v) and k != "depth_prediction"}}


if __name__ == "__main__":
    print(json.dumps(run()))
