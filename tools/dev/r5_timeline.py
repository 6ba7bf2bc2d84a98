"""dev: one training step out of a rocprofv3 kernel trace: kernels in order with durations, gaps, and totals by group."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with clip_step_kernel; take the last complete one
ends = [i for i, r in enumerate(rows) if "clip_step_kernel" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]
tot = collections.OrderedDict(); gaps = 0.0
verbose = len(sys.argv) > 2
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); k = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3; gaps += max(gap, 0)
    if verbose: print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f}  {k}  grid {r['Grid_Size_X']}")
    d = tot.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3; prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
print(f"step span {span:.0f} us, kernels {len(step)}, sum of gaps {gaps:.0f} us")
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print(f"{d:9.1f} us {n:4d} x {d / n:7.1f}  {k}")
