"""Which Winograd launches make up the batch-32 forward, and what each costs (SF_WINO_LIST=1: every launch timed by itself).
Usage: SF_WINO_LIST=1 python3 tools/r05/wino_layers.py [batch] 2> list.txt ; python3 tools/r05/wino_layers.py --summarise list.txt"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def summarise(path):
    agg = collections.OrderedDict()
    for line in open(path):
        if not line.startswith("[sf-wino]"):
            continue
        kv = dict(re.findall(r"(\w+)=(\S+)", line))
        key = " ".join(f"{k}={kv[k]}" for k in ("n", "c", "epi", "act", "mode", "add", "add_scale", "out2", "in_scale", "bias_img", "clamp", "dil", "up", "var")) + " " + line.split()[2]
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(kv["us"]); a[2] += float(kv["gflop"])
    tot = sum(a[1] for a in agg.values())
    print(f"total {tot / 1e3:.2f} ms in {sum(a[0] for a in agg.values())} launches")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[1] / 1e3:8.2f} ms {100 * a[1] / tot:5.1f}%  x{a[0]:3d}  {a[2] / a[1] * 1e3 / 157.3:5.3f} of peak  {k}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2])
    import torch
    from util import build_pair
    from workloads import synthetic as cases
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    C, H, W = 64, 200, 200
    cts, lts, tts, dt = cases.timeset("shipped")
    net, sd = build_pair(C)
    cams, lids = zip(*[cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1], seed=i) for i in range(B)])
    cam, lid = torch.cat(cams).cuda(), torch.cat(lids).cuda()
    x = cases.present_input(cam, lid)
    rep = lambda t: t.repeat(B, 1)
    with torch.no_grad():
        os.environ.pop("SF_WINO_LIST_ON", None)
        net(x, cam, lid, rep(cts), rep(lts), rep(tts))
        torch.cuda.synchronize()
        sys.stderr.write("[sf-wino-begin]\n")
        net(x, cam, lid, rep(cts), rep(lts), rep(tts))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
