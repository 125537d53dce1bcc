"""Camera lift-splat (N1) at the shipped size: 6 cameras x 48 depth bins x 28 x 60 rays, C = 64, BEV 200 x 200,
3 past frames per sample.  Prints one JSON object; imported by bench.py, run directly under rocprofv3
(tools/pmc_lift.sh).  Usage: python3 tools/liftbench.py [--reps N] [--cpu]"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0


def synthetic_rig(b, s, n, dev):
    """A plausible surround rig: n cameras at yaw 2*pi*k/n, ~1.5 m from the ego centre, small ego motion."""
    import math
    gen = torch.Generator().manual_seed(3)
    intr = torch.tensor([[380.0, 0.0, 240.0], [0.0, 380.0, 112.0], [0.0, 0.0, 1.0]]).repeat(b, s, n, 1, 1)
    base = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    extr = torch.zeros(b, s, n, 4, 4)
    for k in range(n):
        a = 2 * math.pi * k / n
        rz = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
        extr[:, :, k, :3, :3] = rz @ base
        extr[:, :, k, :3, 3] = torch.tensor([1.5 * math.cos(a), 1.5 * math.sin(a), 1.5])
    extr[..., 3, 3] = 1.0
    ego = torch.cat([torch.rand((b, s, 3), generator=gen) * 2 - 1, (torch.rand((b, s, 3), generator=gen) * 2 - 1) * 0.05], -1)
    return intr.to(dev), extr.to(dev), ego.to(dev)


def run(reps=20, cpu=False, dev=None):
    from streamingflow_amd import _lib, runtime
    from streamingflow_amd.models.lift_splat import LiftSplat
    dev = dev or torch.device("cuda", 0)
    b, s, n, D, fH, fW, C = 1, 3, 6, 48, 28, 60, 64
    m = LiftSplat().to(dev)
    X, Y = 200, 200
    gen = torch.Generator().manual_seed(4)
    feat = torch.randn((b, s, n, C, fH, fW), generator=gen).to(dev)
    logits = (torch.randn((b, s, n, D, fH, fW), generator=gen) * 2).to(dev)
    intr, extr, ego = synthetic_rig(b, s, n, dev)
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

    fused_ms = timed(lambda: m.lift_splat(feat, logits, intr, extr, ego, nhwc=True), reps)
    # the drop-in path on the materialised tensors (what the reference feeds its bev_pool)
    geo = m.get_geometry(intr.view(b * s, n, 3, 3), extr.view(b * s, n, 4, 4)).view(b, s, n, D, fH, fW, 3)
    prob = logits.view(b * s * n, D, fH, fW).softmax(1)
    x = (prob.unsqueeze(1) * feat.view(b * s * n, C, fH, fW).unsqueeze(2)).permute(0, 2, 3, 4, 1).reshape(b, s, n, D, fH, fW, C).contiguous()
    dropin_ms = timed(lambda: m.projection_to_birds_eye_view(x, geo, ego, nhwc=True), max(3, reps // 4))
    a = m.lift_splat(feat, logits, intr, extr, ego, nhwc=True)
    c = m.projection_to_birds_eye_view(x, geo, ego, nhwc=True)
    diff_cells = float(((a - c).abs().amax(-1) > 1e-4).float().mean())
    # kernels alone
    npts = b * s * n * D * fH * fW
    ncells = b * s * X * Y
    order = torch.empty((npts,), dtype=torch.int32, device=dev)
    start = torch.empty((ncells + 1,), dtype=torch.int32, device=dev)
    ws = runtime.workspace(L.sf_lift_index_ws_bytes(npts, ncells), dev)
    lo, res, dim = m._grid_args()
    aff = m.rig_affines(intr, extr, ego).view(-1, 12)
    fr = m.frustum
    us, vs, ds = fr[0, 0, :, 0].contiguous(), fr[0, :, 0, 1].contiguous(), fr[:, 0, 0, 2].contiguous()
    pr = runtime.ptr
    st = runtime.stream_ptr(dev)
    index_ms = timed(lambda: L.sf_lift_index_rig_fwd(pr(aff), pr(us), pr(vs), pr(ds), b * s, n, D, fH, fW, lo, res, dim, pr(order),
                                                     pr(start), pr(ws), ws.numel() * 4, st), reps)
    rays = runtime.to_nhwc(feat.reshape(b * s * n, C, fH, fW)).view(-1, C)
    probc = prob.contiguous()
    out = torch.empty((X * Y, C), device=dev)
    prev = torch.randn((X * Y, C), device=dev)
    pool_fused_ms = timed(lambda: L.sf_lift_pool_fused_fwd(pr(rays), pr(probc), D, fH * fW, pr(order), pr(start), X * Y, C, pr(prev),
                                                           0.5, pr(out), st), reps)
    xf = x.view(-1, C)
    pool_mat_ms = timed(lambda: L.sf_lift_pool_fwd(pr(xf), pr(order), pr(start), X * Y, C, pr(prev), 0.5, pr(out), st), reps)
    kept = int(start[X * Y].item())
    lens = (start[1:X * Y + 1] - start[:X * Y]).float()
    pts_frame = n * D * fH * fW
    # algorithmic bytes per frame.  reference formulation: x rows of the kept points + geometry + output (+ blend read)
    by_mat = 4.0 * kept * C + 12.0 * pts_frame + 4.0 * X * Y * C * 2
    by_fused = 4.0 * (n * fH * fW * C + pts_frame) + 4.0 * kept + 4.0 * X * Y * C * 2      # features + depth + point list + out/prev
    r = {"workload": f"{n} cameras x {D} depths x {fH}x{fW} rays = {pts_frame} points/frame, C={C}, BEV {X}x{Y}, {s} frames/sample",
         "fused_ms_per_sample": fused_ms, "fused_frames_per_s": s * b / (fused_ms * 1e-3),
         "dropin_ms_per_sample": dropin_ms, "fused_vs_dropin_cells_differing": diff_cells,
         "kernels_us": {"index_all_frames(quantise+radix_sort+row_starts)": index_ms * 1e3,
                        "pool_fused_one_frame": pool_fused_ms * 1e3, "pool_materialised_one_frame": pool_mat_ms * 1e3},
         "points_kept_frame0": kept, "points_per_cell_max": float(lens.max()), "cells_occupied": float((lens > 0).float().mean()),
         "roofline": {"bound": "hbm", "kernel": "lift_pool_kernel<0> (materialised x, the reference's formulation)",
                      "algorithmic_bytes": by_mat, "achieved": by_mat / (pool_mat_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": by_mat / (pool_mat_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
         "fused_pool": {"algorithmic_bytes": by_fused, "gbs": by_fused / (pool_fused_ms * 1e-3) / 1e9,
                        "note": "features (2.6 MB) and depth stay cache resident: latency/L2-bound, not HBM-bound"}}
    if cpu:
        from oracle import lift_splat as LS
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        xc, gc, ec = x[:, :1].cpu(), geo[:, :1].cpu(), ego[:, :1].cpu()
        t0 = time.perf_counter()
        LS.projection_to_birds_eye_view(xc, gc, ec, m.bev_start_position.cpu(), m.bev_resolution.cpu(), m.bev_dimension.cpu(), 0.5)
        tc = time.perf_counter() - t0
        r["cpu_baseline"] = {"value": 1.0 / tc, "unit": "frames/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                             "sample": f"1 frame (projection_to_birds_eye_view on materialised x), oracle/lift_splat.py, {tc:.2f} s"}
    return r


if __name__ == "__main__":
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    print(json.dumps(run(reps, "--cpu" in sys.argv)))
