"""forward time and max-abs difference of the opt-in bf16x3 math mode against exact fp32.  Usage: python3 tools/r03/b3_quick.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import streamingflow_amd as sfa  # noqa: E402
from util import build_pair, cases, hashfill  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    C, H, W = 64, 200, 200
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", True, True, dt)
    cams, lids = zip(*[cases.bev_inputs(C, H, W, 3, 5, seed=i) for i in range(B)])
    cam, lid = torch.cat(cams, 0).cuda(), torch.cat(lids, 0).cuda()
    x = cases.present_input(cam, lid)
    args = (x, cam, lid, cts.repeat(B, 1), lts.repeat(B, 1), tts.repeat(B, 1))
    outs = {}
    modes = (sys.argv[2],) if len(sys.argv) > 2 else ("fp32", "bf16x3", "fp32")
    for mode in modes:
        sfa.set_math_mode(mode)
        torch.manual_seed(5)
        y, _ = net(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            torch.manual_seed(5)
            y, _ = net(*args)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 3
        outs[mode] = y.clone()
        print(f"mode {mode}: batch {B}: {ms:.2f} ms per forward ({ms / B:.2f} per sample)", flush=True)
    if len(modes) > 1:
        print("max-abs bf16x3 vs fp32:", float((outs["bf16x3"] - outs["fp32"]).abs().max()), " absmax", float(outs["fp32"].abs().max()))


if __name__ == "__main__":
    main()
