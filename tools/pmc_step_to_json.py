"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step.sh -> profiles/pmc_ode_step.json.
KiB counters, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); all kernels launched by
the N steps are summed (conv + aux), torch's own fill kernels of the set-up excluded by name."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {"commit": __import__("os").environ.get("SF_COMMIT"), "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/stepbench.py B h w %d; "
                 "sum over the sf:: kernels / steps; FETCH_SIZE x2 (gfx950), KiB.  These are L2 <-> fabric bytes (TCC_EA requests; Infinity-Cache hits are counted, MI355X_MICROARCH.md:297): an upper bound on HBM bytes, not HBM bytes" % N, "cases": {}}
_prev = os.path.join(ROOT, "profiles", "pmc_ode_step.json")
if os.path.exists(_prev):      # cases not re-measured in this run keep their committed figures (and say which commit they are from)
    try:
        _pj = json.load(open(_prev))
        for _k, _v in _pj.get("cases", {}).items():
            _v.setdefault("measured_at_commit", _pj.get("commit"))
            out["cases"][_k] = _v
    except Exception:
        pass
for tag in ("8_50_50", "1_200_200"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcs_{c}_{tag}", "*", "*counter_collection.csv")))
        if not fs:
            continue
        v, dur = 0.0, 0.0
        for r in csv.DictReader(open(fs[-1])):
            if "sf::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"])
        tot[c] = v
        kt = fs[-1].replace("counter_collection", "kernel_trace")
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                if "sf::" in r["Kernel_Name"]:
                    dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            tot["kernel_ns_" + c] = dur
    if len(tot) >= 2:
        B = int(tag.split("_")[0])
        by = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / N
        ns = tot.get("kernel_ns_WRITE_SIZE", 0.0) / N
        out["cases"][tag] = {"batch": B, "fabric_bytes_per_step_launch": by, "fabric_bytes_per_sample_step": by / B,
                             "fetch_bytes": 2.0 * tot["FETCH_SIZE"] * 1024.0 / N, "write_bytes": tot["WRITE_SIZE"] * 1024.0 / N,
                             "kernel_us_per_step_under_pmc": ns / 1e3,
                             "fabric_gbs": by / ns if ns else None, "fabric_gbs_over_hbm_peak_8TBs": by / ns / 8000.0 if ns else None}
# the single latent inside a rollout (pipelined stages: 9 launches per step): chains of 10 and 30 steps, 5 repetitions each
# (tools/chainrun.py); per-step figure = difference of the two sums / (20 steps * 5 repetitions)
def _chain(c, n, prefix="pmcc"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{prefix}_{c}_{n}", "*", "*counter_collection.csv")))
    if not fs:
        return None, None
    v = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(fs[-1])) if "sf::" in r["Kernel_Name"] and r["Counter_Name"] == c)
    kt = fs[-1].replace("counter_collection", "kernel_trace")
    ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "sf::" in r["Kernel_Name"]) if os.path.exists(kt) else 0
    return v, ns
ch = {c: (_chain(c, 10), _chain(c, 30)) for c in ("FETCH_SIZE", "WRITE_SIZE")}
if all(ch[c][0][0] is not None and ch[c][1][0] is not None for c in ch):
    REPS, DN = 5, 20
    f = (ch["FETCH_SIZE"][1][0] - ch["FETCH_SIZE"][0][0]) * 2.0 * 1024.0 / (REPS * DN)
    w = (ch["WRITE_SIZE"][1][0] - ch["WRITE_SIZE"][0][0]) * 1024.0 / (REPS * DN)
    ns = (ch["WRITE_SIZE"][1][1] - ch["WRITE_SIZE"][0][1]) / (REPS * DN)
    out["cases"]["1_50_50"] = {"batch": 1, "what": "steady-state Euler step of one 50x50x64 latent inside sf_nnfo_rollout_fwd (pipelined stages), (chain of 30 - chain of 10) / 20",
                               "fabric_bytes_per_step_launch": f + w, "fabric_bytes_per_sample_step": f + w, "fetch_bytes": f, "write_bytes": w,
                               "kernel_us_per_step_under_pmc": ns / 1e3, "fabric_gbs": (f + w) / ns if ns else None,
                               "fabric_gbs_over_hbm_peak_8TBs": (f + w) / ns / 8000.0 if ns else None}
# the same rollouts on the opt-in persistent flow kernel (SF_PERSIST=1; tools/r04/final_profile.sh: pmcf_* passes)
chf = {c: (_chain(c, 10, "pmcf"), _chain(c, 30, "pmcf")) for c in ("FETCH_SIZE", "WRITE_SIZE")}
if all(chf[c][0][0] is not None and chf[c][1][0] is not None for c in chf):
    REPS, DN = 5, 20
    f = (chf["FETCH_SIZE"][1][0] - chf["FETCH_SIZE"][0][0]) * 2.0 * 1024.0 / (REPS * DN)
    w = (chf["WRITE_SIZE"][1][0] - chf["WRITE_SIZE"][0][0]) * 1024.0 / (REPS * DN)
    ns = (chf["WRITE_SIZE"][1][1] - chf["WRITE_SIZE"][0][1]) / (REPS * DN)
    out["cases"]["1_50_50_flow_kernel"] = {"batch": 1, "what": "the same step on the persistent flow kernel (SF_PERSIST=1), (chain of 30 - chain of 10) / 20",
                                           "fabric_bytes_per_step_launch": f + w, "fetch_bytes": f, "write_bytes": w, "kernel_us_per_step_under_pmc": ns / 1e3,
                                           "fabric_gbs": (f + w) / ns if ns else None}
    out["cases"]["1_50_50_flow_kernel"].pop("measured_at_commit", None)
for _k in ("1_50_50",):
    if all(ch[c][0][0] is not None and ch[c][1][0] is not None for c in ch):
        out["cases"][_k].pop("measured_at_commit", None)
for d in ("profiles", "gpurun_out"):
    json.dump(out, open(os.path.join(ROOT, d, "pmc_ode_step.json"), "w"), indent=1)
print(json.dumps(out["cases"], indent=1))
