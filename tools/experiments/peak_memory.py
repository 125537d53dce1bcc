"""Peak device memory of the bench workload (32 samples per forward): 55 GB allocated / 58 GB reserved in round 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair, cases
C, H, W, B = 64, 200, 200, 32
cts, lts, tts, dt = cases.timeset("shipped")
net, _ = build_pair(C, "euler", True, True, dt)
cam, lid = cases.bev_inputs(C, H, W, 3, 5)
cam, lid = cam.cuda(), lid.cuda()
pres = cases.present_input(cam, lid)
rep = lambda t: t.expand(B, *t.shape[1:]).contiguous()
a = (rep(pres), rep(cam), rep(lid), cts.expand(B, -1).contiguous(), lts.expand(B, -1).contiguous(), tts.expand(B, -1).contiguous())
for _ in range(2): net(*a)
torch.cuda.synchronize()
print("peak allocated GB", torch.cuda.max_memory_allocated() / 1e9, "reserved GB", torch.cuda.max_memory_reserved() / 1e9)
