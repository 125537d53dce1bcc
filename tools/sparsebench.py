"""LiDAR branch at the shipped size: hard voxelisation (350 000 x 5 points) -> per-voxel mean -> SparseEncoder on the
1600 x 1600 x 41 grid -> BEV [1, 256, 200, 200].  One JSON object.  Usage: python3 tools/sparsebench.py [--cpu]"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(reps=5, cpu=False, dev=None):
    from streamingflow_amd import _lib, runtime
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    from streamingflow_amd.voxelize import Voxelization, voxelize
    from workloads import hashfill, synthetic as cases
    import voxelbench
    dev = dev or torch.device("cuda", 0)
    cfg = dict(cases.SPARSE_SHIPPED)
    m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                      encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock").eval()
    sd = hashfill.fill_state_dict(m.state_dict(), seed=83, gain=1.6)      # same (key, shape, seed) -> same weights as oracle.cases.sparse_state_dict
    m.load_state_dict(sd)
    m = m.to(dev)
    vs, rng, mp, mv = cases.VOXEL_SHIPPED
    vz = Voxelization(list(vs), list(rng), mp, (120000, mv)).eval()
    pts = voxelbench.cloud().to(dev)
    feats, coords, sizes = voxelize([pts], vz)
    L = _lib.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
    ms = ctypes.c_float()

    def timed(fn, r):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        L.sf_event_record(e0, runtime.stream_ptr(dev))
        for _ in range(r):
            fn()
        L.sf_event_record(e1, runtime.stream_ptr(dev))
        L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        return ms.value / r
    enc_ms = timed(lambda: m(feats, coords, 1, nhwc=True), reps)
    all_ms = timed(lambda: m(*voxelize([pts], vz)[:2], 1, nhwc=True), reps)
    out = m(feats, coords, 1)
    r = {"workload": f"350000x5 points -> {feats.shape[0]} voxels on 1600x1600x41 -> BEV {tuple(out.shape)}",
         "sparse_encoder_ms": enc_ms, "voxelize_plus_encoder_ms": all_ms, "clouds_per_s": 1e3 / all_ms,
         "occupied_bev_fraction": float((out.abs().amax(1) > 0).float().mean())}
    if cpu:
        from oracle import sparse_encoder_ref as SR      # the checker, timed as the reported CPU baseline only
        n = 20000                     # bounded sample: the numpy oracle is ~linear in the number of voxels
        t0 = time.perf_counter()
        SR.sparse_encoder_forward(sd, feats[:n].cpu().numpy(), coords[:n].cpu().numpy(), 1, cfg)
        tc = time.perf_counter() - t0
        r["cpu_baseline"] = {"value": 1.0 / (tc * feats.shape[0] / n), "unit": "clouds/s", "cores": 1, "kind": "port",
                             "sample": f"first {n} of {feats.shape[0]} voxels through oracle/sparse_encoder_ref.py (numpy), {tc:.1f} s, "
                                       "scaled linearly to the full cloud"}
    return r


if __name__ == "__main__":
    print(json.dumps(run(cpu="--cpu" in sys.argv)))
