"""Accuracy study for VERDICT r2 item 4 (split-bf16 operands on v_mfma_f32_*_bf16) — CPU, test infrastructure only.

Every dense convolution / linear layer of the oracle (oracle/ref_torch.py) is re-evaluated with its two operands split into
bf16 pieces and the cross products accumulated in fp32, which is what an MFMA bf16 kernel with fp32 accumulators computes
(products of bf16 values are exact in fp32; only the summation order differs):
    x3: a = a_hi + a_lo,        a*b ~ hi*hi + hi*lo + lo*hi                      (3 MFMA products)
    x6: a = a_hi + a_mid + a_lo, a*b ~ hh + hm + mh + hl + lh + mm               (6 MFMA products)
    x1: plain bf16 operands (1 product), for scale.
The output is compared with the exact-fp32 oracle on the same inputs / weights / noise: max-abs over the BEV logits.
Usage: python3 tools/r03/bf16x3_study.py [quick]"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases, ref_torch as R  # noqa: E402
from workloads import hashfill  # noqa: E402
from util import build_pair  # noqa: E402

MODE = {"n": 0}
_conv2d, _convT, _linear = F.conv2d, F.conv_transpose2d, F.linear


def pieces(a, n):
    out, r = [], a
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def products(n):
    if n == 1:
        return [(0, 0)]
    if n == 2:
        return [(0, 0), (0, 1), (1, 0)]
    return [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]


def split_apply(fn, x, w, n):
    xs, ws = pieces(x, min(n, 3) if n > 1 else 1), pieces(w, min(n, 3) if n > 1 else 1)
    y = None
    for i, j in reversed(products(n)):       # small terms first
        t = fn(xs[i], ws[j])
        y = t if y is None else y + t
    return y


class Shim:
    """stands in for torch.nn.functional inside oracle.ref_torch"""
    def __getattr__(self, k):
        return getattr(F, k)

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        n = MODE["n"]
        if n == 0 or groups != 1:      # depthwise layers are not MFMA work
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        y = split_apply(lambda a, c: _conv2d(a, c, None, stride, padding, dilation, 1), x, w, n)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def conv_transpose2d(self, x, w, b=None, stride=1, padding=0, *a, **k):
        n = MODE["n"]
        if n == 0:
            return _convT(x, w, b, stride, padding, *a, **k)
        y = split_apply(lambda p, q: _convT(p, q, None, stride, padding, *a, **k), x, w, n)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def linear(self, x, w, b=None):
        n = MODE["n"]
        if n == 0:
            return _linear(x, w, b)
        y = split_apply(lambda p, q: _linear(p, q), x, w, n)
        return y if b is None else y + b


R.F = Shim()


def forward(C, H, W, ts, solver):
    cts, lts, tts, dt = cases.timeset(ts)
    _, sd = build_pair(C, solver, True, True, dt, device="cpu")
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    with torch.no_grad():
        y, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2, solver, True, True,
                                                hashfill.HashedNoise(cases.EPS_SEED))
    return y


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    torch.set_num_threads(8)
    runs = [("golden-size C=8 16x16 shipped euler", 8, 16, 16, "shipped", "euler"),
            ("golden-size C=16 24x24 stream40 (46 steps) euler", 16, 24, 24, "stream40", "euler"),
            ("golden-size C=16 24x24 stream40 (46 steps) rk4", 16, 24, 24, "stream40", "rk4"),
            ("C=64 latent 50x50 (BEV 200x200) config 2 shipped euler", 64, 200, 200, "shipped", "euler")]
    if not quick:
        runs += [("C=64 BEV 200x200 config 4 future16 euler", 64, 200, 200, "future16", "euler"),
                 ("C=64 BEV 200x200 config 5 stream40 (46 steps) euler", 64, 200, 200, "stream40", "euler"),
                 ("C=64 BEV 200x200 config 5 stream40 (46 steps) midpoint", 64, 200, 200, "stream40", "midpoint"),
                 ("C=32 BEV 200x200 config 1 euler", 32, 200, 200, "config1", "euler")]
    rows = []
    for name, C, H, W, ts, solver in runs:
        t0 = time.time()
        MODE["n"] = 0
        ref = forward(C, H, W, ts, solver)
        row = {"case": name, "absmax_of_output": float(ref.abs().max())}
        for tag, n in (("bf16x1", 1), ("bf16x3", 2), ("bf16x6", 3)):
            MODE["n"] = n
            y = forward(C, H, W, ts, solver)
            row[tag + "_maxabs"] = float((y - ref).abs().max())
            row[tag + "_frames_maxabs"] = [float(v) for v in (y - ref).abs().flatten(2).max(2)[0][0]][:: max(1, y.shape[1] // 8)]
        row["seconds"] = time.time() - t0
        rows.append(row)
        print(json.dumps(row), flush=True)
    json.dump({"what": __doc__.split("Usage")[0], "rows": rows}, open(os.path.join(ROOT, "profiles", "r03_bf16x3_accuracy_study.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
