"""Per-step busy time and gaps from a rocprofv3 kernel trace of tools/dev/decode_bench.py: python tools/dev/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by beam_update_kernel (beam) or by the copy_logits kernel (one per step)
marks = [i for i, e in enumerate(ev) if "copy_logits" in e[2]]
print("steps", len(marks))
import statistics
per = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = ev[a:b]
    busy = sum(e[1] - e[0] for e in seg)
    span = ev[b][0] - ev[a][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    per.append((span, busy, len(seg), max(gaps) if gaps else 0))
per = per[len(per) // 2:]          # second (timed) run
print("median span us %.1f  busy us %.1f  kernels %d  largest gap us %.1f" % (
    statistics.median(p[0] for p in per) / 1e3, statistics.median(p[1] for p in per) / 1e3, statistics.median(p[2] for p in per),
    statistics.median(p[3] for p in per) / 1e3))
a, b = marks[-20], marks[-19]
seg = ev[a:b + 1]
for i in range(len(seg) - 1):
    g = (seg[i + 1][0] - seg[i][1]) / 1e3
    if g > 8: print("  gap %.1f us after %s" % (g, seg[i][2][:60]))
