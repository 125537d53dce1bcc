#!/usr/bin/env python
"""bench.py — ODE-steps/s of the MI355X-native GRU-ODE future-state path (BASELINE.json metric).

One "step" of this bench = one pass of the hot path over one batch of synthetic input: a full
``FuturePredictionODE.forward`` on ``--batch`` samples (default 32) of BASELINE config 2 (C=64, BEV
200x200, 3 camera + 5 LiDAR observations, 7 targets, variable-step Euler: 10 ODE steps + 8
Bayesian jumps, SmallEncoder on 8 frames, SmallDecoder + 2x(SpatialGRU + res block) head on 7 frames
per sample), inputs resident in HBM.  The reference's forward takes a batch and loops over it one
sample at a time; here samples with the same schedule structure go through the kernels together.
``value`` = ODE steps integrated per second over all ranks = n_ode_steps * batch * K * N / t;
the single-sample (batch 1) latency of the same forward is reported next to it.

N > 1: one process per GPU (torchrun, backend nccl = RCCL over xGMI), every rank runs its own samples (the
reference is batch-1, samples shard with no data-path collective: weak scaling); barrier + synchronize on both sides
of the timed region, max over ranks.  What the north star names for the multi-GPU case — the all-gather of per-sample
BEV grids — runs INSIDE the timed region: after every forward each rank hands one sample's [T, C, H, W] grid to
``streamingflow_amd.dist.gather_predictions`` on a side stream, overlapped with the next forward (``--no-gather`` to
leave it out); its stand-alone time, byte count and the RCCL world size are reported (``multi_gpu``).

Extra objects on the JSON line: ``roofline`` (dominant kernel of the forward, per-launch hipEvent timing in a dedicated
pass of the same workload through libsfnative's profiler), ``roofline_ode_step`` (the kernel group of ONE GRU-ODE
step — the unit SURVEY.md §8d defines: 728 C^2 h w FLOP, 16 C h w + eps + weights bytes — from >= 20 individually timed
hipGraph replays: median and p95), ``batch1_forward`` (the same forward at the reference's own batch size) and
``cpu_baseline`` (the oracle — a torch-CPU port of the reference path — on the host cores, rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--timeset", default="shipped", help="workloads.synthetic.TIMESETS key (default: BASELINE config 2)")
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--batch", type=int, default=32, help="samples per forward on each GPU (the reference API "
                    "takes a batch and loops over it; here same-structure samples run through the kernels together)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary sections (next-row components, ODE step alone)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the all-gather of per-sample BEV grids out of the timed region")
    ap.add_argument("--headline-only", action="store_true", help="only the timed workload and its roofline pass (profiling aid: the rocprofv3 "
                    "kernel statistics of such a run average over exactly the launches the roofline object describes)")
    ap.add_argument("--dry", action="store_true", help="no GPU: the launcher, the rendezvous, the sample sharding, the timed-region collectives "
                    "and the JSON relay on the gloo backend with a stub forward (tests/test_bench_launcher.py); the line says \"dry\": true")
    return ap.parse_args()


def launch_ranks(a, argv):
    """``python bench.py --gpus N`` (N > 1) outside torchrun: start the N ranks as children — one process per GPU through
    ``python -m torch.distributed.run`` on 127.0.0.1 — BEFORE this process touches the GPU (it never does: it only relays
    the children's output and exits with their return code; no exec)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SF_BENCH_LAUNCHED_BY"] = "bench.py"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line_ok = False
    for line in p.stdout:                      # relay as it comes; the JSON line of rank 0 must report all N ranks
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                line_ok = line_ok or json.loads(line).get("n_gpus") == a.gpus
            except ValueError:
                pass
    rc = p.wait()
    if rc == 0 and not line_ok:
        sys.stderr.write(f"bench.py: the {a.gpus} ranks exited 0 without a JSON line for n_gpus={a.gpus}\n")
        rc = 3
    return rc


def dry_run(a, world):
    """The N > 1 control flow without a GPU (gloo, CPU tensors): rendezvous, sample sharding, barrier-bracketed timed
    region with the all-gather inside, MAX over ranks, counters all-reduce, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from streamingflow_amd import dist as sfd
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n_ode = max(1, a.batch), 10
    grid = torch.full((2, 4, 8, 8), float(rank))

    def forward():
        time.sleep(0.002)
        return [grid]

    def fence():
        if world > 1:
            dist.barrier()
    got = None
    for _ in range(a.warmup):
        forward()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = forward()
        if world > 1:
            got = sfd.gather_predictions({rank: y[0]}, world)
    fence()
    el = time.perf_counter() - t0
    multi = None
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        assert all(float(g.flatten()[0]) == r for r, g in enumerate(got)), "all-gather returned a wrong slot"
        cnt = sfd.reduce_counters(torch.ones(8))
        assert float(cnt[0]) == world
        multi = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(), "gather_in_timed_region": True,
                 "gather_bytes_per_rank": grid.numel() * 4, "gather_bytes_total": grid.numel() * 4 * world, "gather_ms_standalone": None}
    if rank == 0:
        print(json.dumps({"metric": "ODE-steps/s (DRY RUN: stub forward, no GPU)", "dry": True, "value": n_ode * B * a.steps * world / el,
                          "unit": "ODE-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "launched_by": os.environ.get("SF_BENCH_LAUNCHED_BY", "torchrun"), "multi_gpu": multi}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _family(calls_per_forward, ms_per_forward, flops, nbytes, ms):
    """one launch family of the per-launch profiler: rate, the roof that bounds it (by arithmetic intensity) and the fraction of THAT roof"""
    t = ms * 1e-3
    tfl, gbs = flops / t / 1e12, nbytes / t / 1e9
    intensity = flops / nbytes if nbytes else float("inf")
    bound = "mfma" if intensity >= PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    return {"calls_per_forward": calls_per_forward, "ms_per_forward": ms_per_forward, "tflops": tfl, "algorithmic_gbs": gbs,
            "flop_per_byte": intensity, "bound": bound, "frac_of_bound": tfl / PEAK_F32_MFMA_TFLOPS if bound == "mfma" else gbs / PEAK_HBM_GBS}


def _all_ranks(value, world, device):
    """every rank's integer `value` (all-gather of one int64 each)"""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [int(g.item()) for g in got]


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # not under torchrun: be the launcher (nothing below runs in this process)
        sys.exit(launch_ranks(a, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:                                     # never fall through to a different rank count than asked for
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world}")
    if a.dry:
        return dry_run(a, world)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SF_BENCH_BACKEND=gloo + SF_BENCH_ONE_DEVICE=1: exercise the multi-rank path on a 1-GPU box
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    if os.environ.get("SF_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world and dist.get_backend() == backend

    import streamingflow_amd as sfa
    from streamingflow_amd import _lib, schedule as S
    from workloads import hashfill, synthetic as cases

    C, H, W = 64, 200, 200
    cts, lts, tts, dt = cases.timeset(a.timeset)
    cfg = cases.make_cfg(C, impute=True, solver=a.solver, variable=True)
    net = sfa.FuturePredictionODE(C, C, 4, cfg, n_gru_blocks=2, n_res_layers=1, delta_t=dt).eval()
    sd = cases.fpode_state_dict(net.state_dict())        # random-init weights (hashed, reproducible)
    net.load_state_dict(sd)
    net = net.to(dev)
    B = max(1, a.batch)
    cams, lids = zip(*[cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1], seed=rank * 1000 + i) for i in range(B)])
    cam, lid = cams[0], lids[0]
    cam_d, lid_d = torch.cat(cams, 0).to(dev), torch.cat(lids, 0).to(dev)     # B samples per rank
    cts_b, lts_b, tts_b = cts.repeat(B, 1), lts.repeat(B, 1), tts.repeat(B, 1)
    x_in = cases.present_input(cam_d, lid_d)
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True, a.solver)
    n_ode = sc.n_steps

    def forward():
        return net(x_in, cam_d, lid_d, cts_b, lts_b, tts_b)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from streamingflow_amd import dist as sfd
    do_gather = world > 1 and not a.no_gather      # nccl: RCCL all-gather on the device; gloo: the same call, staged through the host
    side = torch.cuda.Stream(device=dev) if do_gather else None
    gathered = [None]

    def gather(y):
        """one sample's BEV grid per rank -> every rank (RCCL all_gather_into_tensor on a side stream, overlapped with the
        next forward; sample index = rank, so each rank owns exactly one slot)"""
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            y[0].record_stream(side)
            gathered[0] = sfd.gather_predictions({rank: y[0]}, world)

    for _ in range(a.warmup):
        y, _ = forward()
        if do_gather:
            gather(y)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y, _ = forward()
        if do_gather:
            gather(y)
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = 1e3 * el / a.steps
    value = n_ode * B * a.steps * world / el

    multi = None
    if world > 1:
        grid = y[0]
        gbytes = grid.numel() * 4
        gms = None
        if do_gather:
            ts = []
            for _ in range(5):
                fence()
                tg = time.perf_counter()
                sfd.gather_predictions({rank: grid}, world)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - tg))
            gms = sorted(ts)[2]
            ok = all(g.shape == grid.shape for g in gathered[0]) and torch.equal(gathered[0][rank], grid)
            assert ok, "all-gather returned a wrong slot"
        # what gathering the WHOLE shard (every sample's grid, not one per rank) would cost: measured on up to 4 samples per rank, outside the
        # timed region (the figure for B samples is linear in the bytes)
        shard_ms = shard_n = None
        try:
            shard_n = min(B, 4)
            loc = {rank + world * i: y[i] for i in range(shard_n)}      # sample i of this rank's shard: index i * W + rank (dist.shard_indices)
            sfd.gather_predictions(loc, world * shard_n)
            fence()
            tg = time.perf_counter()
            sfd.gather_predictions(loc, world * shard_n)
            torch.cuda.synchronize()
            shard_ms = 1e3 * (time.perf_counter() - tg)
        except Exception as ex:
            shard_ms = repr(ex)
        cnt = torch.ones(8, device=dev if backend == "nccl" else "cpu")
        sfd.reduce_counters(cnt)
        assert float(cnt[0]) == world
        multi = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(), "gather_in_timed_region": do_gather,
                 "gather_bytes_per_rank": gbytes, "gather_bytes_total": gbytes * world, "gather_ms_standalone": gms,
                 "gather_estimate_ms": gbytes / 153e9 * 1e3,      # one xGMI hop at ~153 GB/s per link
                 "whole_shard_gather": {"samples_per_rank_measured": shard_n, "bytes_per_rank_measured": None if shard_n is None else gbytes * shard_n,
                                        "ms_standalone": shard_ms, "samples_per_rank_in_a_forward": B, "bytes_per_rank_in_a_forward": gbytes * B,
                                        "ms_per_forward_if_gathered_linear_estimate": (shard_ms * B / shard_n) if isinstance(shard_ms, float) and shard_n else None,
                                        "xgmi_estimate_ms_per_forward": gbytes * B * (world - 1) / (min(world - 1, 7) * 153e9) * 1e3 if world > 1 else None,
                                        "what": "all_gather_into_tensor of every sample's grid of the shard (not in the timed region: the reference evaluates its metrics per sample and "
                                                "all-reduces counters; SURVEY 8e)"},
                 "devices": sorted({int(v) for v in _all_ranks(local, world, dev if backend == "nccl" else "cpu")}),
                 "what": "all_gather_into_tensor of one sample's [T, C, H, W] fp32 BEV grid per rank (streamingflow_amd.dist.gather_predictions), side stream, overlapped with the next forward"
                         + ("" if backend == "nccl" else " [gloo: staged through the host, the host blocks on it]") + "; metric counters: all_reduce(SUM)"}

    # ---- single-sample latency of the same forward (batch 1) -----------------------------------------
    def forward1():
        return net(x_in[:1], cam_d[:1], lid_d[:1], cts, lts, tts)
    single_ms = single_eager_ms = None
    if not a.headline_only:
        def time_forward1(n=5):
            for _ in range(2):
                forward1()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                forward1()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        # the module's defaults (what a caller of evaluate.py gets): rollout replayed from a hipGraph, noise drawn in the
        # sampling epilogue (Philox) ...
        single_ms = time_forward1()
        # ... and the round-1..3 form of the same call: rollout enqueued launch by launch, eps from one torch.randn
        net.gru_ode.use_graph, net.gru_ode.in_kernel_noise = False, False
        single_eager_ms = time_forward1()
        net.gru_ode.use_graph, net.gru_ode.in_kernel_noise = None, None

    L = _lib.lib()
    rollout = step_only = roof_step = None
    if world == 1 and not a.headline_only:      # secondary sections: single-GPU runs only (at N > 1 no rank may lag behind the others)
        # ---- ODE rollout alone (the serial chain the north star names), same stream, hipEvents -------
        ode = net.gru_ode
        hx = torch.randn((len(times), H // 4, W // 4, C), device=dev) * 0.5
        eps = torch.randn((sc.n_draws, H // 4, W // 4, C), device=dev)
        ode.use_graph = False                   # eager first (the module default replays single-latent rollouts from a graph)
        for _ in range(2):
            ode.rollout_nhwc(hx, sc, eps)
        torch.cuda.synchronize()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
        from streamingflow_amd import runtime
        reps = 10
        L.sf_event_record(e0, runtime.stream_ptr(dev))
        for _ in range(reps):
            ode.rollout_nhwc(hx, sc, eps)
        L.sf_event_record(e1, runtime.stream_ptr(dev))
        ms = ctypes.c_float()
        L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        rollout_ms = ms.value / reps
        # the same rollout replayed as one captured hipGraph
        ode.use_graph = True
        for _ in range(2):
            ode.rollout_nhwc(hx, sc, eps)
        torch.cuda.synchronize()
        L.sf_event_record(e0, runtime.stream_ptr(dev))
        for _ in range(reps):
            ode.rollout_nhwc(hx, sc, eps)
        L.sf_event_record(e1, runtime.stream_ptr(dev))
        L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        rollout_graph_ms = ms.value / reps
        ode.use_graph = None
        # BASELINE config 5: streaming 0.05 s x 40 targets -> 46 ODE steps + 8 jumps, the whole rollout one hipGraph
        stream40 = None
        try:
            c5, l5, t5, dt5 = cases.timeset("stream40")
            tm5, _ = S.merge_observations(c5[0].tolist(), l5[0].tolist())
            sc5 = S.build_schedule(tm5, t5[0].tolist(), dt5, True, a.solver)
            eps5 = torch.randn((sc5.n_draws, H // 4, W // 4, C), device=dev)
            ode.use_graph = True
            for _ in range(2):
                ode.rollout_nhwc(hx, sc5, eps5)
            torch.cuda.synchronize()
            L.sf_event_record(e0, runtime.stream_ptr(dev))
            for _ in range(reps):
                ode.rollout_nhwc(hx, sc5, eps5)
            L.sf_event_record(e1, runtime.stream_ptr(dev))
            L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            ode.use_graph = None
            m5 = ms.value / reps
            stream40 = {"hipgraph_replay_ms": m5, "ode_steps": sc5.n_steps, "jumps": sc5.n_jumps, "solver": a.solver,
                        "us_per_op": 1e3 * m5 / len(sc5.ops), "ode_steps_per_s": sc5.n_steps / (m5 * 1e-3)}
        except Exception as ex:
            stream40 = {"error": repr(ex)}
        flops_step = 728.0 * C * C * (H // 4) * (W // 4)          # Euler step, algorithmic
        flops_jump = flops_step
        rollout = {"ms": rollout_ms, "hipgraph_replay_ms": rollout_graph_ms, "batch": 1, "ode_steps": sc.n_steps, "jumps": sc.n_jumps,
                   "ode_steps_per_s_single_sample_graph": sc.n_steps / (rollout_graph_ms * 1e-3),
                   "us_per_op": 1e3 * rollout_ms / max(1, len(sc.ops)),
                   "ops_per_s": len(sc.ops) / (rollout_ms * 1e-3),
                   "tflops": (sc.n_steps * flops_step + sc.n_jumps * flops_jump) / (rollout_ms * 1e-3) / 1e12,
                   "config5_stream40": stream40}

        # ---- one fused GRU-ODE step alone (SURVEY §8d unit of work), hipEvents on the launch stream -----
        def time_step(Bs, h, w, reps):
            s_in = torch.randn((Bs, h, w, C), device=dev) * 0.5
            p_in = torch.randn((Bs, h, w, C), device=dev) * 0.5
            e_in = torch.randn((S.DRAWS_PER_STEP[a.solver], Bs, h, w, C), device=dev)
            s_o, p_o = torch.empty_like(s_in), torch.empty_like(p_in)
            coef = torch.from_numpy(S.Schedule(dts=[float(dt)]).coef_array()).to(dev)
            wsb = L.sf_ode_step_ws_bytes(C, Bs, h, w)
            ws = runtime.workspace(wsb, dev)
            pr = runtime.ptr

            def one():
                _lib.check(L.sf_ode_step_fwd(ode.gru_c.packed().struct, ode.p_model.packed().struct, _lib.SOLVER[a.solver], 1,
                                             pr(s_in), pr(p_in), pr(coef), pr(e_in), pr(s_o), pr(p_o), Bs, h, w, pr(ws),
                                             ws.numel() * 4, runtime.stream_ptr(dev)), "ode_step")
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            L.sf_event_record(e0, runtime.stream_ptr(dev))
            for _ in range(reps):
                one()
            L.sf_event_record(e1, runtime.stream_ptr(dev))
            L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            t = ms.value / reps * 1e-3
            mult = {"euler": 1, "midpoint": 2, "rk4": 4}[a.solver]
            fl = mult * 728.0 * C * C * h * w * Bs
            nparam = sum(v.numel() for k, v in sd.items() if k.startswith(("gru_ode.gru_c.", "gru_ode.p_model."))
                         and "num_batches" not in k)
            by = (16.0 + 4.0 * S.DRAWS_PER_STEP[a.solver]) * C * h * w * Bs + 4.0 * nparam
            return {"batch": Bs, "latent": f"{h}x{w}x{C}", "us_per_step": t * 1e6, "steps_per_s": Bs / t,
                    "tflops": fl / t / 1e12, "mfma_frac": fl / t / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "algorithmic_bytes_per_step": by / Bs, "algorithmic_gbs": by / t / 1e9,
                    "hbm_frac": by / t / 1e9 / PEAK_HBM_GBS}
        step_only = {"single_sample": time_step(1, H // 4, W // 4, 50), "batch8": time_step(8, H // 4, W // 4, 20),
                     "stress_latent_200x200": time_step(1, H, W, 5),
                     # `tflops` / `mfma_frac` here divide the reference's (direct-form) FLOPs of a step by its time.  The 3x3 layers of a step
                     # run in Winograd form (2.25x fewer executed products) — from 14 000 pixels per launch on conv_wino.hip, and since
                     # round 6 on the single latent too (conv_sp.hip, SF_WINO_SP) — so these are direct-form-EQUIVALENT rates, not
                     # matrix-pipe utilisation
                     "flop_accounting": "algorithmic (direct-form) FLOPs / time; every case includes Winograd layers (SF_WINO_MIN_P = 14000 on the large tiles, "
                                        "SF_WINO_SP on the single latent): direct-form-equivalent rates"}
        if rank == 0 and world == 1 and not a.no_cpu_baseline:      # the same unit of work on the host cores (oracle, 16 threads)
            from oracle import ref_torch as R
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            s_c, p_c = torch.randn((1, C, H // 4, W // 4)) * 0.5, torch.randn((1, C, H // 4, W // 4)) * 0.5
            with torch.no_grad():
                R.ode_step(sd, "gru_ode", s_c, p_c, dt, a.solver, True, hashfill.HashedNoise(0))
                t0 = time.perf_counter()
                for _ in range(5):
                    R.ode_step(sd, "gru_ode", s_c, p_c, dt, a.solver, True, hashfill.HashedNoise(0))
                tc = (time.perf_counter() - t0) / 5
            step_only["cpu_baseline"] = {"value": 1.0 / tc, "unit": "ODE-steps/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                                         "sample": f"5 single-sample {a.solver} steps at latent 50x50x64, oracle/ref_torch.py:ode_step, {tc * 1e3:.1f} ms each"}
        pmc_step = os.path.join(ROOT, "profiles", "pmc_ode_step.json")
        pmc_step_data = None
        if os.path.exists(pmc_step):
            try:
                pmc_step_data = json.load(open(pmc_step))
                step_only["rocprof_fabric_traffic"] = pmc_step_data
            except Exception:
                pass

        # ---- roofline object of the ODE-step kernel group: ONE Euler step of one sample at latent 50x50x64
        # captured into a hipGraph, >= 20 replays timed one by one with hipEvents on the launch stream: median and p95
        STEPS_PER_GRAPH = 10      # one replay = 10 chained steps (state / input ping-pong): the 10-16 us host-side floor of a
                                  # graph replay (MI355X_MICROARCH.md, graph-replay-floor) is not kernel time

        def graph_step_times(h, w, n=25):
            bufs = [torch.randn((1, h, w, C), device=dev) * 0.5 for _ in range(4)]      # s_a, p_a, s_b, p_b
            e_in = torch.randn((S.DRAWS_PER_STEP[a.solver], 1, h, w, C), device=dev)
            coef = torch.from_numpy(S.Schedule(dts=[float(dt)]).coef_array()).to(dev)
            ws = torch.empty(L.sf_ode_step_ws_bytes(C, 1, h, w) // 4 + 1024, dtype=torch.float32, device=dev)
            pr = runtime.ptr

            def chain(sp):
                for i in range(STEPS_PER_GRAPH):
                    si, pi, so, po = (bufs[0], bufs[1], bufs[2], bufs[3]) if i % 2 == 0 else (bufs[2], bufs[3], bufs[0], bufs[1])
                    _lib.check(L.sf_ode_step_fwd(ode.gru_c.packed().struct, ode.p_model.packed().struct, _lib.SOLVER[a.solver], 1,
                                                 pr(si), pr(pi), pr(coef), pr(e_in), pr(so), pr(po), 1, h, w, pr(ws), ws.numel() * 4, sp), "ode_step")
            chain(runtime.stream_ptr(dev))
            torch.cuda.synchronize()
            cap = torch.cuda.Stream(device=dev)
            ex = ctypes.c_void_p()
            with torch.cuda.stream(cap):
                sp = runtime.stream_ptr(dev)
                _lib.check(L.sf_graph_begin(sp), "graph_begin")
                try:
                    chain(sp)
                finally:
                    _lib.check(L.sf_graph_end(sp, ctypes.byref(ex)), "graph_end")
            sp = runtime.stream_ptr(dev)
            for _ in range(3):
                L.sf_graph_launch(ex, sp)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                L.sf_event_record(e0, sp)
                L.sf_graph_launch(ex, sp)
                L.sf_event_record(e1, sp)
                L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                ts.append(ms.value * 1e3 / STEPS_PER_GRAPH)
            L.sf_graph_destroy(ex)
            ts.sort()
            return ts[len(ts) // 2], ts[min(len(ts) - 1, int(round(0.95 * (len(ts) - 1))))], n
        def rollout_step_times(h, w, n1=10, n2=30, n=25, Bs=1):
            """the same step inside a rollout (what FuturePredictionODE.forward runs): hipGraph replays of rollouts with one jump +
            N Euler steps for N = n1 and n2; per-step time = (t(n2) - median t(n1)) / (n2 - n1).  Inside a rollout branch 2 of the
            next cell and the state half of its gates ride in infer_state's launches (csrc/api.hip: Carry): 9 launches per step."""
            per = S.DRAWS_PER_STEP[a.solver]
            hx1 = torch.randn((1, Bs, h, w, C), device=dev) * 0.5
            out = {}
            for nn_ in (n1, n2):
                scn = S.Schedule(ops=[(_lib.OP_JUMP, 0)] + [(_lib.OP_STEP, i) for i in range(nn_)], dts=[float(dt)] * nn_, sel_nops=[nn_ + 1],
                                 n_draws=1 + per * nn_)
                en = torch.randn((scn.n_draws, Bs, h, w, C), device=dev)
                ode.use_graph = True
                for _ in range(3):
                    ode.rollout_nhwc(hx1, scn, en)
                torch.cuda.synchronize()
                ts = []
                sp = runtime.stream_ptr(dev)
                for _ in range(n):
                    L.sf_event_record(e0, sp)
                    ode.rollout_nhwc(hx1, scn, en)
                    L.sf_event_record(e1, sp)
                    L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                    ts.append(ms.value * 1e3)
                ode.use_graph = None
                ts.sort()
                out[nn_] = ts
            base = out[n1][len(out[n1]) // 2]
            per_step = sorted((t - base) / (n2 - n1) for t in out[n2])
            return per_step[len(per_step) // 2], per_step[min(len(per_step) - 1, int(round(0.95 * (len(per_step) - 1))))], n
        roof_step = None
        try:
            hh, ww = H // 4, W // 4
            alone_us, alone_p95, _ = graph_step_times(hh, ww)
            med_us, p95_us, nrep = rollout_step_times(hh, ww)
            mult = {"euler": 1, "midpoint": 2, "rk4": 4}[a.solver]
            fl = mult * 728.0 * C * C * hh * ww
            nparam = sum(v.numel() for k, v in sd.items() if k.startswith(("gru_ode.gru_c.", "gru_ode.p_model.")) and "num_batches" not in k)
            by = (16.0 + 4.0 * S.DRAWS_PER_STEP[a.solver]) * C * hh * ww + 4.0 * nparam
            tr = None
            if pmc_step_data:
                tr = pmc_step_data.get("cases", {}).get("1_50_50", {}).get("fabric_bytes_per_step_launch")
            roof_step = {"bound": "mfma", "achieved": fl / (med_us * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / (med_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": tr,
                         "traffic_source": {"file": "profiles/pmc_ode_step.json", "measured_in_this_run": False,
                                            "commit": (pmc_step_data or {}).get("commit")},
                         "kernel": f"ode_step kernel group ({a.solver}, one sample, latent {hh}x{ww}x{C}) inside sf_nnfo_rollout_fwd: hipGraph replays of "
                                   f"one jump + N steps, (t(N=30) - t(N=10)) / 20 (steady state: 9 launches per step, branch 2 of the next cell beside infer_state)",
                         "us_per_step_median": med_us, "us_per_step_p95": p95_us, "graph_replays_timed": nrep, "launches_per_step": 9 if a.solver == "euler" else None,
                         "standalone_call": {"what": f"{STEPS_PER_GRAPH} chained sf_ode_step_fwd calls per hipGraph replay, time / {STEPS_PER_GRAPH} (no cross-call pipelining: 10 launches per step; the round-1/2 figure)",
                                             "us_per_step_median": alone_us, "us_per_step_p95": alone_p95,
                                             "frac": fl / (alone_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS},
                         "flops_per_step": fl, "algorithmic_bytes_per_step": by,
                         "hbm_frac_if_bytes_bound": by / (med_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                         "steps_per_s": 1e6 / med_us}
            # the opt-in persistent flow kernel (sf_set_flow_mode / SF_PERSIST=1): the same rollouts as ONE resident launch ordered by
            # tile-level dataflow (north star: "one LDS-tiled kernel per step"; bitwise the same results; needs an otherwise idle device)
            try:
                sfa.set_persistent_flow(True)
                fmed, fp95, _ = rollout_step_times(hh, ww)
                c5f = None
                try:
                    c5, l5, t5, dt5 = cases.timeset("stream40")
                    tm5, _ = S.merge_observations(c5[0].tolist(), l5[0].tolist())
                    sc5 = S.build_schedule(tm5, t5[0].tolist(), dt5, True, a.solver)
                    hx5 = torch.randn((len(tm5), hh, ww, C), device=dev) * 0.5
                    eps5 = torch.randn((sc5.n_draws, hh, ww, C), device=dev)
                    ode.use_graph = True
                    for _ in range(2):
                        ode.rollout_nhwc(hx5, sc5, eps5)
                    torch.cuda.synchronize()
                    L.sf_event_record(e0, runtime.stream_ptr(dev))
                    for _ in range(10):
                        ode.rollout_nhwc(hx5, sc5, eps5)
                    L.sf_event_record(e1, runtime.stream_ptr(dev))
                    L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                    c5f = ms.value / 10
                finally:
                    ode.use_graph = None
                roof_step["persistent_flow_kernel"] = {
                    "what": "opt-in (sf_set_flow_mode(1) / SF_PERSIST=1): every launch group of the rollout a phase of one resident launch, tile-level "
                            "dependencies instead of kernel boundaries; same measurement as us_per_step_median",
                    "us_per_step_median": fmed, "us_per_step_p95": fp95, "frac": fl / (fmed * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "vs_launch_per_layer": med_us / fmed, "stream40_rollout_hipgraph_ms": c5f, "launches_per_rollout": 3}
            except Exception as ex:
                roof_step["persistent_flow_kernel"] = {"error": repr(ex)}
            finally:
                sfa.set_persistent_flow(None)
                ode.drop_graphs()
        except Exception as ex:      # secondary object: never lose the headline line over it
            roof_step = {"error": repr(ex)}
        try:      # the same two cases inside a rollout (pipelined stages where they pay: DESIGN 4.1)
            mult = {"euler": 1, "midpoint": 2, "rk4": 4}[a.solver]
            for key, (Bs, hh_, ww_, n1_, n2_) in {"batch8": (8, H // 4, W // 4, 6, 16), "stress_latent_200x200": (1, H, W, 3, 8)}.items():
                med_, p95_, _ = rollout_step_times(hh_, ww_, n1_, n2_, 9, Bs)
                fl_ = mult * 728.0 * C * C * hh_ * ww_ * Bs
                step_only[key]["in_rollout"] = {"us_per_step_median": med_, "us_per_step_p95": p95_, "tflops": fl_ / (med_ * 1e-6) / 1e12,
                                                "mfma_frac": fl_ / (med_ * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        except Exception as ex:
            step_only["in_rollout_error"] = repr(ex)

    # ---- roofline of the dominant kernel: per-launch hipEvents in a dedicated pass ---------------
    roof = None
    if not a.no_roofline:
        L.sf_prof_enable(1)
        for _ in range(2):
            forward()
        torch.cuda.synchronize()
        NK = _lib.SF_PROF_KEYS
        calls = (ctypes.c_int32 * NK)(); pms = (ctypes.c_double * NK)()
        pfl = (ctypes.c_double * NK)(); pby = (ctypes.c_double * NK)()
        L.sf_prof_collect(calls, pms, pfl, pby)
        L.sf_prof_enable(0)
        tot = sum(pms)
        k = max(range(NK), key=lambda i: pms[i])
        achieved = pfl[k] / (pms[k] * 1e-3) / 1e12
        traffic, traffic_commit, traffic_kernel = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic, traffic_commit, traffic_kernel = pj.get("hbm_bytes_per_launch"), pj.get("commit"), pj.get("kernel")
            except Exception:
                traffic = None
        roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                # PMC counters need their own rocprofv3 passes: the figure is the committed summary of the same command,
                # never a measurement of this run
                "traffic_source": {"file": "profiles/pmc_dominant.json", "measured_in_this_run": False, "commit": traffic_commit, "kernel": traffic_kernel},
                "kernel": _lib.KERNEL_NAMES.get(k, str(k)), "launches_per_forward": calls[k] // 2,
                # Winograd launches are priced at their EXECUTED FLOPs (16 products per 2x2 outputs and (cin, cout) pair; the direct
                # form the reference computes has 36): `achieved` / `frac` are what the matrix pipe did, the direct-form figure what
                # the layer is worth (savings are not credited as achieved FLOPs; ODE-steps/s is time-based)
                "flop_accounting": ("executed Winograd F(2x2,3x3) FLOPs; direct-form equivalent = x 2.25 = %.1f TFLOP/s" % (2.25 * achieved))
                                   if _lib.KERNEL_NAMES.get(k, "").startswith("conv_wino") else "algorithmic (direct-form) FLOPs = executed FLOPs",
                # numeric twins of flop_accounting: what the same launches are worth in the reference's (direct-form) FLOPs
                "direct_form_equivalent_tflops": (2.25 if _lib.KERNEL_NAMES.get(k, "").startswith("conv_wino") else 1.0) * achieved,
                "direct_form_equivalent_frac": (2.25 if _lib.KERNEL_NAMES.get(k, "").startswith("conv_wino") else 1.0) * achieved / PEAK_F32_MFMA_TFLOPS,
                "avg_launch_us": 1e3 * pms[k] / max(1, calls[k]),
                "flops_per_launch": pfl[k] / max(1, calls[k]),
                "algorithmic_bytes_per_launch": pby[k] / max(1, calls[k]),
                "hbm_frac_if_bytes_bound": (pby[k] / (pms[k] * 1e-3) / 1e9) / PEAK_HBM_GBS,
                "share_of_conv_time": pms[k] / tot if tot else None,
                "all_conv_tflops": sum(pfl) / (tot * 1e-3) / 1e12 if tot else None,
                # every launch family against ITS roof: arithmetic intensity (FLOPs / algorithmic bytes: each input pixel, weight and output once)
                # against the machine balance (fp32 MFMA peak / HBM peak = 19.7 FLOP/B) decides which
                "per_kernel": {_lib.KERNEL_NAMES.get(i, str(i)): _family(calls[i] // 2, pms[i] / 2, pfl[i], pby[i], pms[i]) for i in range(NK) if calls[i]}}

    # flat scalars of the GRU-ODE step inside `roofline` (the driver's record keeps the scalar members of that object)
    if roof is not None and isinstance(roof_step, dict) and "error" not in roof_step:
        roof["ode_step_us_median"] = roof_step["us_per_step_median"]
        roof["ode_step_us_p95"] = roof_step["us_per_step_p95"]
        roof["ode_step_frac"] = roof_step["frac"]
        roof["ode_step_launches"] = roof_step.get("launches_per_step")
        roof["ode_step_traffic_ratio"] = (roof_step["traffic"] / roof_step["algorithmic_bytes_per_step"]) if roof_step.get("traffic") else None
        pf = roof_step.get("persistent_flow_kernel") or {}
        roof["ode_step_flow_kernel_us_median"] = pf.get("us_per_step_median")      # opt-in persistent flow kernel, same measurement
        roof["ode_step_flow_kernel_frac"] = pf.get("frac")
    if roof is not None:
        roof["batch1_forward_ms"] = single_ms

    # ---- camera lift-splat voxel pooling feeding the BEV tensor (HBM-bound gather) ----
    lift = None
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:     # secondary figures: single-GPU runs only (no rank may lag behind the others at N > 1)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import liftbench
            lift = liftbench.run(reps=10, cpu=(world == 1 and not a.no_cpu_baseline), dev=dev)
            pl = os.path.join(ROOT, "profiles", "pmc_lift_pool.json")
            if os.path.exists(pl):
                lift["roofline"]["traffic"] = json.load(open(pl)).get("hbm_bytes_per_launch")
        except Exception as ex:      # secondary figure: never lose the headline line over it
            lift = {"error": repr(ex)}

    vox = None
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:
        try:
            import voxelbench
            vox = voxelbench.run(reps=10, cpu=(world == 1 and not a.no_cpu_baseline), dev=dev)
            import sparsebench
            vox["sparse_encoder"] = sparsebench.run(reps=3, cpu=(world == 1 and not a.no_cpu_baseline), dev=dev)
        except Exception as ex:
            vox = {"error": repr(ex)}

    dec = None
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:
        try:
            import decoderbench
            dec = decoderbench.run(reps=5, cpu=(world == 1 and not a.no_cpu_baseline), dev=dev)
            import temporalbench
            dec["temporal_model"] = temporalbench.run(reps=5, cpu=(world == 1 and not a.no_cpu_baseline), dev=dev)
            import e2ebench
            dec["end_to_end_after_image_backbone"] = e2ebench.run(reps=2, dev=dev)
        except Exception as ex:
            dec = {"error": repr(ex)}

    # ---- the same forward with every 3x3 layer in direct form (set_winograd(False)): same box, same inputs --------------------------
    # what the Winograd kernel (csrc/conv_wino.hip) is worth on the headline, and how far its results are from the direct form's
    direct = None
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:
        try:
            net.gru_ode.in_kernel_noise = False    # the two forwards compared below see the same eps (torch.randn under one seed)
            torch.manual_seed(4321)
            yw, _ = forward()
            yw = yw.clone()
            sfa.set_winograd(False)                # re-packs every module without Winograd weights
            torch.manual_seed(4321)
            yd, _ = forward()
            errd = float((yd - yw).abs().max())
            net.gru_ode.in_kernel_noise = None
            forward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nd = max(3, a.steps // 4)
            for _ in range(nd):
                forward()
            torch.cuda.synchronize()
            msd = 1e3 * (time.perf_counter() - t0) / nd
            direct = {"what": "streamingflow_amd.set_winograd(False) / SF_WINO=0: every 3x3 layer and the cells' 7x7 in direct form on the implicit-GEMM tiles (rounds 1-3)",
                      "ms_per_step": msd, "ode_steps_per_s": n_ode * B / (msd * 1e-3), "headline_over_this": msd / ms_per_step,
                      "max_abs_winograd_vs_direct_same_forward": errd, "absmax_of_output": float(yd.abs().max())}
        except Exception as ex:
            direct = {"error": repr(ex)}
        finally:
            sfa.set_winograd(True)
            net.gru_ode.in_kernel_noise = None

    # ---- opt-in math mode "bf16x3" (VERDICT r2 item 4): a SEPARATE object, never the headline -------------------------
    # operands split into two bf16 pieces, three v_mfma_f32_16x16x32_bf16 products, fp32 accumulators; `value` above is exact fp32
    b3 = None
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:
        try:
            net.gru_ode.in_kernel_noise = False    # the two forwards compared below see the same eps (torch.randn under one seed)
            torch.manual_seed(1234)
            y32, _ = forward()
            y32 = y32.clone()
            sfa.set_math_mode("bf16x3")
            torch.manual_seed(1234)
            y3, _ = forward()                      # re-packs every module with split-bf16 weights
            err = float((y3 - y32).abs().max())
            net.gru_ode.in_kernel_noise = None
            for _ in range(2):
                forward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb3 = max(3, a.steps // 4)
            for _ in range(nb3):
                forward()
            torch.cuda.synchronize()
            ms3 = 1e3 * (time.perf_counter() - t0) / nb3
            forward1(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                forward1()
            torch.cuda.synchronize()
            ms3_1 = 1e3 * (time.perf_counter() - t0) / 3
            step3 = roll3 = None
            try:
                med3, p953, _ = rollout_step_times(H // 4, W // 4)
                step3 = {"us_per_step_median": med3, "us_per_step_p95": p953, "fp32_mfma_peak_equivalent_frac": 728.0 * C * C * (H // 4) * (W // 4) / (med3 * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
                ode.use_graph = True
                for _ in range(2):
                    ode.rollout_nhwc(hx, sc, eps)
                torch.cuda.synchronize()
                L.sf_event_record(e0, runtime.stream_ptr(dev))
                for _ in range(10):
                    ode.rollout_nhwc(hx, sc, eps)
                L.sf_event_record(e1, runtime.stream_ptr(dev))
                L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                ode.use_graph = None
                roll3 = ms.value / 10
            except Exception as ex:
                step3 = {"error": repr(ex)}
            e2e3 = None
            try:
                import e2ebench
                e2e3 = e2ebench.run(reps=2, dev=dev).get("ms_per_sample")      # the module-level mode is bf16x3 here
            except Exception as ex:
                e2e3 = repr(ex)
            b3 = {"mode": "bf16x3", "opt_in": "streamingflow_amd.set_math_mode('bf16x3'); the default and every figure outside this object is exact fp32",
                  "what": "operands split into two bf16 pieces (weights once by sf_pack_conv, activations in registers after the LDS read), "
                          "hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulators; LDS-DMA kernel family and the small-P kernel",
                  "max_abs_vs_fp32_same_forward": err, "absmax_of_output": float(y32.abs().max()), "tolerance_north_star": 1e-3,
                  "max_abs_vs_oracle": "tests/test_gpu_bf16x3.py (4.4e-5 on the full-size forward), profiles/r03_bf16x3_accuracy_study.json (<= 1.1e-4 on configs 1, 2, 4, 5)",
                  "ms_per_step": ms3, "ode_steps_per_s": n_ode * B / (ms3 * 1e-3), "speedup_vs_fp32_headline": ms_per_step / ms3,
                  "batch1_forward_ms": ms3_1, "ode_step_in_rollout": step3, "rollout_18op_hipgraph_ms": roll3,
                  "end_to_end_after_image_backbone_ms_per_sample": e2e3}
        except Exception as ex:
            b3 = {"error": repr(ex)}
        finally:
            sfa.set_math_mode("fp32")

    # ---- BASELINE configs 4 and 5 as whole forwards (evaluate.py:43 --future-frames 16; evaluate_streaming.py:119-126): batch 1 and a batch ----
    other_cfgs = {}
    if rank == 0 and world == 1 and not a.no_extras and not a.headline_only:
        for key, ts_name, Bc, cpu_targets in (("config4_future16", "future16", 8, None), ("config5_stream40", "stream40", 4, 13)):
            try:
                c_, l_, t_, dt_ = cases.timeset(ts_name)
                net_c = sfa.FuturePredictionODE(C, C, 4, cfg, n_gru_blocks=2, n_res_layers=1, delta_t=dt_).eval()
                net_c.load_state_dict(sd)
                net_c = net_c.to(dev)
                tm_, _ = S.merge_observations(c_[0].tolist(), l_[0].tolist())
                sc_ = S.build_schedule(tm_, t_[0].tolist(), dt_, True, a.solver)

                def run_c(nb, reps):
                    cam_c, lid_c = cam_d[:nb], lid_d[:nb]
                    x_c = cases.present_input(cam_c, lid_c)
                    args = (x_c, cam_c, lid_c, c_.repeat(nb, 1), l_.repeat(nb, 1), t_.repeat(nb, 1))
                    for _ in range(2):
                        yc, _ = net_c(*args)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        yc, _ = net_c(*args)
                    torch.cuda.synchronize()
                    assert yc.shape[1] == t_.shape[1]
                    del yc
                    return 1e3 * (time.perf_counter() - t0) / reps
                m1, mb = run_c(1, 5), run_c(Bc, 3)
                obj = {"workload": f"timeset '{ts_name}': {len(tm_)} observations, {t_.shape[1]} decoded frames, {sc_.n_steps} ODE steps + {sc_.n_jumps} jumps per sample "
                                   f"(variable-step {a.solver}), C=64, BEV 200x200; whole FuturePredictionODE.forward",
                       "batch1": {"ms_per_forward": m1, "ode_steps_per_s": sc_.n_steps / (m1 * 1e-3), "frames_per_s": t_.shape[1] / (m1 * 1e-3)},
                       "batched": {"samples_per_forward": Bc, "ms_per_forward": mb, "ms_per_sample": mb / Bc, "ode_steps_per_s": sc_.n_steps * Bc / (mb * 1e-3),
                                   "frames_per_s": t_.shape[1] * Bc / (mb * 1e-3)}}
                if not a.no_cpu_baseline:
                    from oracle import ref_torch as R
                    cores = min(os.cpu_count() or 1, 16)
                    torch.set_num_threads(cores)
                    t_cpu = t_ if cpu_targets is None else t_[:, :cpu_targets]      # bounded sample: the first targets only
                    sc_cpu = S.build_schedule(tm_, t_cpu[0].tolist(), dt_, True, a.solver)
                    with torch.no_grad():
                        t0 = time.perf_counter()
                        R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, c_, l_, t_cpu, dt_, 2, a.solver, True, True, hashfill.HashedNoise(0))
                        tc = time.perf_counter() - t0
                    obj["cpu_baseline"] = {"value": sc_cpu.n_steps / tc, "unit": "ODE-steps/s", "cores": cores, "kind": "port",
                                           "sample": f"ONE single-sample forward (no warm-up pass) of the oracle, oracle/ref_torch.py, on the first {t_cpu.shape[1]} of the {t_.shape[1]} target "
                                                     f"frames ({sc_cpu.n_steps} ODE steps + {sc_cpu.n_jumps} jumps, {t_cpu.shape[1]} decoded frames at 200x200x64): {tc:.1f} s",
                                           "frames_per_s": t_cpu.shape[1] / tc}
                other_cfgs[key] = obj
                del net_c
                torch.cuda.empty_cache()
            except Exception as ex:
                other_cfgs[key] = {"error": repr(ex)}

    # ---- CPU baseline: the oracle (torch-CPU port of the reference path) on the host cores --------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.headline_only:
        from oracle import ref_torch as R
        cores = min(os.cpu_count() or 1, 16)   # more threads only thrash on these small convs
        torch.set_num_threads(cores)
        tcs = []
        with torch.no_grad():
            for _ in range(4):      # ~10 s of CPU work; the first pass also warms the thread pool and the allocator
                t0 = time.perf_counter()
                R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2,
                                                a.solver, True, True, hashfill.HashedNoise(0))
                tcs.append(time.perf_counter() - t0)
        tc = sorted(tcs[1:])[1]     # median of the three warm passes
        cpu = {"value": n_ode / tc, "unit": "ODE-steps/s", "cores": cores, "kind": "port",
               "sample": f"median of 3 warm single-sample forwards of the same workload ({n_ode} ODE steps + {sc.n_jumps} jumps, 8+7 frames at "
                         f"200x200x64), oracle/ref_torch.py on torch {torch.__version__} CPU, {tc:.1f} s"}

    if rank == 0:
        out = {"metric": "ODE-steps/s on 200x200x64 BEV (GRU-ODE future-state path, full FuturePredictionODE.forward)",
               "value": value, "unit": "ODE-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"BASELINE config 2: C=64, BEV 200x200 (latent 50x50), timeset '{a.timeset}' "
                                      f"({len(times)} observations, {tts.shape[1]} targets), variable-step {a.solver}: "
                                      f"{n_ode} ODE steps + {sc.n_jumps} jumps per sample, {B} sample(s) per forward per GPU",
                          "batch_per_gpu": B,
                          "parallelism": f"replicas x{world} (sample sharding; " + ((("RCCL" if backend == "nccl" else "gloo, host-staged") + " all-gather of one BEV grid per rank per forward on a side stream)") if do_gather else "no data-path collective)")},
               "samples_per_s": B * a.steps * world / el, "batch_per_gpu": B, "ms_per_sample": ms_per_step / B,
               "rollout_mode": ("eager launches (the batched rollout is not launch-bound; auto-graph applies to single-latent rollouts only), "
                                "eps drawn in the sampling epilogue (Philox4x32-10, module default when no noise source is injected)"),
               "batch1_forward": None if single_ms is None else {
                   "what": "the same FuturePredictionODE.forward at the reference's own batch size (evaluate.py:46: one sample per call)",
                   "ms_per_forward": single_ms, "ode_steps_per_s": n_ode / (single_ms * 1e-3), "samples_per_s": 1e3 / single_ms,
                   "mode": "module defaults: rollout replayed from a captured hipGraph, eps drawn in the sampling epilogue (Philox4x32-10)",
                   "ms_per_forward_eager_randn": single_eager_ms,
                   "mode_eager_randn": "use_graph=False, in_kernel_noise=False: rollout enqueued launch by launch, eps from one torch.randn (the form rounds 1-3 timed)"},
               "single_sample_forward_ms": single_ms,
               "single_sample_ode_steps_per_s": None if single_ms is None else n_ode / (single_ms * 1e-3),
               "roofline_ode_step": roof_step, "multi_gpu": multi,
               "config4_future16": other_cfgs.get("config4_future16"), "config5_stream40": other_cfgs.get("config5_stream40"),
               "ode_rollout_only": rollout, "ode_step_only": step_only, "lift_splat": lift, "lidar_voxelize": vox, "bev_decoder": dec, "direct_form_3x3": direct, "bf16x3_mode": b3, "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
