"""Wall time of the single-sample FuturePredictionODE.forward against the GPU-busy time (sum of kernel durations from a
rocprofv3 kernel trace of this script): how much of the batch-1 latency is host-side launch overhead.
Usage: rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/r02/fwd1_busy.py ; then sum the trace."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import streamingflow_amd as sfa   # noqa: E402
from oracle import cases, refimport   # noqa: E402

C, H, W = 64, 200, 200
cts, lts, tts, dt = cases.timeset("shipped")
cfg = refimport.make_cfg(C, impute=True, solver="euler", variable=True)
net = sfa.FuturePredictionODE(C, C, 4, cfg, n_gru_blocks=2, n_res_layers=1, delta_t=dt).eval()
net.load_state_dict(cases.fpode_state_dict(net.state_dict()))
net = net.to("cuda:0")
cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1], seed=0)
cam, lid = cam.cuda(), lid.cuda()
x_in = cases.present_input(cam, lid)
for _ in range(3):
    net(x_in, cam, lid, cts, lts, tts)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    net(x_in, cam, lid, cts, lts, tts)
torch.cuda.synchronize()
print("wall ms per forward", 1e3 * (time.perf_counter() - t0) / N, "forwards", N + 3)
