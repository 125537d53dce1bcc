"""In-kernel time stamps of the persistent flow kernel (SF_PERSIST=1) on a diagnostic build (-DSF_STAMP):
per phase of the last two steady-state Euler steps of a 1-jump + N-step rollout of one 50x50x64 latent, over the workgroups that
had an item: dependency wait, acquire + barrier, first chunk, K loop, hand-off, epilogue, drain (10-ns clock, s_memrealtime),
and the spans / overlaps between consecutive phases.
Usage: SF_PERSIST=1 SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r04/flow_stamps.py [n_steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from util import build_pair  # noqa: E402
from streamingflow_amd import _lib  # noqa: E402
import chainbench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    C, h, w = 64, 50, 50
    net, _ = build_pair(C, "euler", True, True, 0.05)
    ode = net.gru_ode
    ode.use_graph = False
    sc = chainbench.chain_schedule(n, "euler")
    hx = torch.randn(1, 1, h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, 1, h, w, C, device="cuda")
    for _ in range(3):
        ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    st = torch.zeros((64, 4096, 16), dtype=torch.int64, device="cuda")
    L = _lib.lib()
    _lib.check(L.sf_debug_stamps(st.data_ptr()), "stamps")
    ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(None), "stamps")
    a = st.cpu().numpy().astype(np.float64) * 0.01      # us
    used = [k for k in range(64) if (a[k][:, 7] > 0).any()]
    nph = max(used) + 1                    # the last phase is the tail of the last step (nobody reads the infer_state behind it)
    assert nph < 64, "slots wrap at 64 phases"
    names = ["gates1(x)", "cand1+dec2", "7x7+proj+1x1", "tail 3x3", "pm conv1+proj+gates2", "pm conv2+cand2+gates1(s)", "pm rb1.conv1 (SE)", "pm rb1.conv2", "pm last (SE, sample)"]
    prev_end = prev_fin = None
    t_first = None
    for k in range(max(0, nph - 18), nph):
        t = a[k]
        m = t[:, 7] > 0
        t = t[m]
        if not len(t):
            print(f"phase {k}: no stamps")
            continue
        fin = t[:, 5] > 0                      # workgroups that ran an epilogue
        start = t[:, 7]
        end_all = np.where(t[:, 6] > 0, t[:, 6], np.where(t[:, 3] > 0, t[:, 3], t[:, 7]))
        t0 = start.min()
        if t_first is None:
            t_first = t0
        med = lambda x: float(np.median(x)) if len(x) else float("nan")
        seg = {
            "poll": med(t[:, 9] - t[:, 7]), "acq+bar": med(t[:, 10] - t[:, 9]), "1st chunk": med(t[:, 2] - t[:, 10]),
            "K loop": med(t[:, 3] - t[:, 2]), "hand-off": med(t[fin, 4] - t[fin, 3]), "epilogue": med(t[fin, 5] - t[fin, 4]),
            "drain": med(t[fin, 6] - t[fin, 5]),
        }
        last_fin = t[fin, 6].max() if fin.any() else float("nan")
        first_fin = t[fin, 6].min() if fin.any() else float("nan")
        line = (f"phase {k:2d} {names[(3 - (nph - 1 - k)) % 9]:26s} {len(t):3d} items ({int(fin.sum()):3d} finish) | first start {t0 - t_first:7.2f}  start skew {start.max() - t0:5.2f}"
                f" | " + "  ".join(f"{n_} {v:5.2f}" for n_, v in seg.items()) +
                f" | tiles finish {first_fin - t_first:7.2f} .. {last_fin - t_first:7.2f}")
        if prev_fin is not None:
            line += f" | K loops start {np.median(t[:, 2]) - prev_fin:+5.2f} us after the previous phase's last tile"
        print(line)
        prev_end, prev_fin = end_all.max(), last_fin
    print(f"{prev_fin - t_first:.2f} us between the first item start of phase {max(0, nph - 18)} and the last finished tile of phase {nph - 1} = 18 phases = two steps -> {(prev_fin - t_first) / 2:.2f} us per step (stamp build)")


if __name__ == "__main__":
    main()
