"""Accuracy study for VERDICT r3 item 4 (Winograd F(2x2, 3x3) for the 3x3 layers) — CPU, test infrastructure only.

Every 3x3 / stride-1 convolution of the oracle (oracle/ref_torch.py; dilated ones through their polyphase components,
ConvTranspose2d(k3, s1, p1) as the equivalent convolution) is re-evaluated as
    U = G g G^T (per cout, cin),  V = B^T d B (per 4x4 input tile, stride 2),  M = sum_cin U (.) V,  Y = A^T M A
in fp32 — what a Winograd kernel with fp32 MFMA accumulators computes: 2.25x fewer multiplies than the direct form, different
rounding (the transforms add / subtract before the products).  The output is compared with the exact direct-convolution
oracle on the same inputs / weights / noise: max-abs over the BEV output.  Two scopes:
    head: only layers at >= 100x100 pixels per image (encoder / decoder / SpatialGRU / DeepLab head — where a Winograd
          kernel would be used: the large-tile launches of the batched forward)
    all:  every 3x3 stride-1 layer, the 50x50 GRU-ODE step included (46 chained steps in the streaming configs)
Usage: python3 tools/r04/winograd_study.py [quick]"""
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

MODE = {"scope": None, "count": 0, "skipped": 0}
_conv2d, _convT = F.conv2d, F.conv_transpose2d

BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def winograd_3x3(x, w):
    """conv2d(x, w, padding=1) for 3x3 w, stride 1, via F(2x2, 3x3) in fp32.  x [N, C, H, W], w [O, C, 3, 3]."""
    N, C, H, W = x.shape
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [N, C, Th, Tw, 4, 4]
    V = torch.einsum("ai,nctuij,bj->nctuab", BT, d, BT)          # B^T d B
    U = torch.einsum("ai,ocij,bj->ocab", G, w, G)                # G g G^T
    Th, Tw = V.shape[2], V.shape[3]
    # 16 independent [O, C] x [C, N Th Tw] products
    Vm = V.permute(4, 5, 1, 0, 2, 3).reshape(16, C, N * Th * Tw)
    Um = U.permute(2, 3, 0, 1).reshape(16, w.shape[0], C)
    M = torch.bmm(Um, Vm).reshape(4, 4, w.shape[0], N, Th, Tw)
    Y = torch.einsum("ai,ijontu,bj->notuab", AT, M, AT)          # [N, O, Th, Tw, 2, 2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], 2 * Th, 2 * Tw)
    return Y[:, :, :H, :W].contiguous()


def conv3x3_any_dilation(x, w, dil):
    if dil == 1:
        return winograd_3x3(x, w)
    N, C, H, W = x.shape
    y = x.new_empty((N, w.shape[0], H, W))
    for a in range(dil):
        for b in range(dil):
            sub = x[:, :, a::dil, b::dil]
            if sub.numel():
                y[:, :, a::dil, b::dil] = winograd_3x3(sub.contiguous(), w)
    return y


def eligible(x, w, stride, padding, dilation, groups):
    s = stride if isinstance(stride, int) else stride[0]
    p = padding if isinstance(padding, int) else padding[0]
    d = dilation if isinstance(dilation, int) else dilation[0]
    if groups != 1 or tuple(w.shape[2:]) != (3, 3) or s != 1 or p != d:
        return 0
    if MODE["scope"] == "head" and x.shape[2] * x.shape[3] < 100 * 100:
        return 0
    return d


class Shim:
    """stands in for torch.nn.functional inside oracle.ref_torch"""
    def __getattr__(self, k):
        return getattr(F, k)

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        d = eligible(x, w, stride, padding, dilation, groups) if MODE["scope"] else 0
        if not d:
            MODE["skipped"] += 1
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        MODE["count"] += 1
        y = conv3x3_any_dilation(x, w, d)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def conv_transpose2d(self, x, w, b=None, stride=1, padding=0, *a, **k):
        ok = MODE["scope"] and tuple(w.shape[2:]) == (3, 3) and stride == 1 and padding == 1 and not a and not k
        if ok and MODE["scope"] == "head" and x.shape[2] * x.shape[3] < 100 * 100:
            ok = False
        if not ok:
            MODE["skipped"] += 1
            return _convT(x, w, b, stride, padding, *a, **k)
        MODE["count"] += 1
        y = winograd_3x3(x, w.flip(2, 3).transpose(0, 1).contiguous())       # res_models.py:19-20: the equivalent convolution
        return y if b is None else y + b.view(1, -1, 1, 1)


R.F = Shim()


def forward(C, H, W, ts, solver):
    cts, lts, tts, dt = cases.timeset(ts)
    _, sd = build_pair(C, solver, True, True, dt, device="cpu")
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    with torch.no_grad():
        y, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2, solver, True, True,
                                                hashfill.HashedNoise(cases.EPS_SEED))
    return y


def selfcheck():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 11, 14, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    for d in (1, 2, 3):
        e = (conv3x3_any_dilation(x, w, d) - _conv2d(x, w, None, 1, d, d)).abs().max()
        assert e < 2e-5, (d, float(e))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    torch.set_num_threads(8)
    selfcheck()
    runs = [("golden-size C=8 16x16 shipped euler", 8, 16, 16, "shipped", "euler"),
            ("golden-size C=16 24x24 stream40 (46 steps) euler", 16, 24, 24, "stream40", "euler"),
            ("golden-size C=16 24x24 stream40 (46 steps) rk4", 16, 24, 24, "stream40", "rk4"),
            ("C=64 latent 50x50 (BEV 200x200) config 2 shipped euler", 64, 200, 200, "shipped", "euler")]
    if not quick:
        runs += [("C=32 BEV 200x200 config 1 euler", 32, 200, 200, "config1", "euler"),
                 ("C=64 BEV 200x200 config 4 future16 euler", 64, 200, 200, "future16", "euler"),
                 ("C=64 BEV 200x200 config 5 stream40 (46 steps) euler", 64, 200, 200, "stream40", "euler"),
                 ("C=64 BEV 200x200 config 5 stream40 (46 steps) midpoint", 64, 200, 200, "stream40", "midpoint")]
    rows = []
    out_path = os.path.join(ROOT, "profiles", "r04_winograd_accuracy_study.json")
    for name, C, H, W, ts, solver in runs:
        t0 = time.time()
        MODE["scope"] = None
        ref = forward(C, H, W, ts, solver)
        row = {"case": name, "absmax_of_output": float(ref.abs().max())}
        for scope in ("head", "all"):
            if scope == "head" and H * W < 100 * 100:
                continue
            MODE.update(scope=scope, count=0, skipped=0)
            y = forward(C, H, W, ts, solver)
            row[scope + "_maxabs"] = float((y - ref).abs().max())
            row[scope + "_layers_winograd"] = MODE["count"]
            row[scope + "_layers_direct"] = MODE["skipped"]
            row[scope + "_frames_maxabs"] = [float(v) for v in (y - ref).abs().flatten(2).max(2)[0][0]][:: max(1, y.shape[1] // 8)]
        row["seconds"] = time.time() - t0
        rows.append(row)
        print(json.dumps(row), flush=True)
        json.dump({"what": __doc__.split("Usage")[0], "gate": "<= 1e-4 max-abs on the BEV output (north-star tolerance 1e-3)", "rows": rows},
                  open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
