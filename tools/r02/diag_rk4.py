import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from util import build_pair, cases, hashfill
from streamingflow_amd import schedule as S
solver = sys.argv[1]
C, h, w = 64, 50, 50
cts, lts, tts, dt = cases.timeset("stream40")
net, _ = build_pair(C, solver, True, True, dt)
ode = net.gru_ode
times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
sc = S.build_schedule(times, tts[0].tolist(), dt, True, solver)
for k in range(3):
    hx = (hashfill.normal(f"s40hx{k}", (8, h, w, C), 61) * 0.5).cuda()
    eps = hashfill.normal(f"s40eps{k}", (sc.n_draws, h, w, C), 62).cuda()
    ode.use_graph = False
    a, fa = ode.rollout_nhwc(hx, sc, eps); a = a.clone()
    a2, _ = ode.rollout_nhwc(hx, sc, eps); a2 = a2.clone()
    ode.use_graph = True
    b, fb = ode.rollout_nhwc(hx, sc, eps); b = b.clone()
    b2, _ = ode.rollout_nhwc(hx, sc, eps); b2 = b2.clone()
    d = (a - b).abs()
    first = int((d.flatten(1).max(1)[0] > 0).nonzero()[0]) if d.max() > 0 else -1
    print(k, "eager==eager", torch.equal(a, a2), "graph==graph", torch.equal(b, b2), "eager==graph", torch.equal(a, b), "maxdiff", float(d.max()), "first differing target", first, flush=True)
