"""Per-launch timeline of one fused GRU-ODE Euler step from a rocprofv3 kernel trace of tools/stepbench.py.
Usage: python3 tools/step_trace.py <kernel_trace.csv> <launches per step>"""
import csv
import sys


def main():
    path, per = sys.argv[1], int(sys.argv[2])
    rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith(("void sf::", "sf::"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // per
    if len(sys.argv) > 3 and sys.argv[3] == "tail":        # a rollout trace: the last 2 steps' worth of launches before the final copies
        rows = rows[-(2 * per + 4):]
    else:
        rows = rows[(n - 1) * per:n * per] if n > 1 else rows   # the last full step (warm)
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += b - a
        name = r["Kernel_Name"].replace("void sf::", "").replace("sf::", "")[:58]
        print(f"{(a - t0) / 1e3:8.1f} us  +gap {(a - prev_end) / 1e3:5.1f}  dur {(b - a) / 1e3:6.1f}  grid {r.get('Grid_Size_X', '?'):>7}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')} wg {r.get('Workgroup_Size_X', '?'):>4}  {name}")
        prev_end = b
    print(f"step wall {(prev_end - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, {len(rows)} launches")


if __name__ == "__main__":
    main()
