import os, sys, json, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_configs as T
from util import cases
tag = "config4_future16"
C, H, W, ts, solver, impute, variable = cases.BIG_CASES[tag]
y, sd, _ = T._forward(C, H, W, ts, solver, impute, variable)
st = json.load(open(os.path.join(T.GOLD, "big_stats.json")))["cases"][tag]["out"]
flat = y.reshape(-1).double().cpu()
print("PIPE", os.environ.get("SF_PIPE"), "samples maxdiff", float((flat[torch.tensor(st["sample_idx"])] - torch.tensor(st["samples"])).abs().max()),
      "mean", flat.mean().item(), "ref mean", st["mean"], "absmax", flat.abs().max().item(), st["absmax"])

