"""Does running the batch as k concurrent sub-batches on k HIP streams fill launch tails / overlap the output-write-bound
layers of one sub-batch with the MFMA-bound layers of another?  Usage (GPU box): python tools/experiments/two_stream.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair, cases  # noqa: E402


def main():
    C, H, W, B = 64, 200, 200, 32
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", True, True, dt)
    cam, lid = cases.bev_inputs(C, H, W, 3, 5)
    cam, lid = cam.cuda(), lid.cuda()
    pres = cases.present_input(cam, lid)
    rep = lambda t, n: t.expand(n, *t.shape[1:]).contiguous()
    for k in (1, 2, 4):
        n = B // k
        args = [(rep(pres, n), rep(cam, n), rep(lid, n), cts.expand(n, -1).contiguous(), lts.expand(n, -1).contiguous(), tts.expand(n, -1).contiguous())
                for _ in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]

        def run():
            for s, a in zip(streams, args):
                with torch.cuda.stream(s):
                    net(*a)

        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"{k} stream(s) x {n} samples: {ms:.1f} ms per {B} samples = {B * 10 / ms * 1e3:.0f} ODE-steps/s")


if __name__ == "__main__":
    main()
