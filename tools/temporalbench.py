"""TemporalModel (N3) at the shipped sizes: camera branch (70 = 64 + 6 ego-pose channels) and LiDAR branch (256),
3 frames of 200 x 200.  One JSON object.  Usage: python3 tools/temporalbench.py [--cpu]"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(reps=10, cpu=False, dev=None):
    from streamingflow_amd import _lib, runtime
    from streamingflow_amd.models.temporal_model import TemporalModel
    from workloads import synthetic as cases
    dev = dev or torch.device("cuda", 0)
    L = _lib.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
    ms = ctypes.c_float()
    out = {}
    for name, cin in (("camera_c70", 70), ("lidar_c256", 256)):
        m = TemporalModel(cin, 3, (200, 200), start_out_channels=64).eval()
        sd = cases.decoder_state_dict(m.state_dict(), seed=71)
        m.load_state_dict(sd)
        m = m.to(dev)
        x = torch.randn((1, 3, cin, 200, 200), generator=torch.Generator().manual_seed(1)).to(dev)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        L.sf_event_record(e0, runtime.stream_ptr(dev))
        for _ in range(reps):
            m(x)
        L.sf_event_record(e1, runtime.stream_ptr(dev))
        L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        r = {"ms_per_call": ms.value / reps, "samples_per_s": 1e3 * reps / ms.value}
        if cpu and name == "camera_c70":
            from oracle import temporal_model_ref as TR      # the checker, timed as the reported CPU baseline only
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            with torch.no_grad():
                t0 = time.perf_counter()
                TR.temporal_model_forward(sd, x.cpu(), (200, 200))
                tc = time.perf_counter() - t0
            r["cpu_baseline"] = {"value": 1.0 / tc, "unit": "samples/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                                 "sample": f"1 sample (3 frames), oracle/temporal_model_ref.py on torch CPU, {tc:.2f} s"}
        out[name] = r
    out["workload"] = "TemporalModel(receptive_field=3, pyramid pooling) + DeepLabHead on 3 frames of 200x200"
    return out


if __name__ == "__main__":
    print(json.dumps(run(cpu="--cpu" in sys.argv)))
