#!/bin/bash
# dev (GPU box): kernel-trace statistics of the bench step -> gpurun_out/bench_prof/ + a per-step summary
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/bench_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/bench_prof/*kernel_stats.csv") + glob.glob("gpurun_out/bench_prof/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
# steps traced: warmup 2 + 1 count + 5 timed + 1 extra + untrimmed (2 + 5) = 16 (10 trimmed-equivalent ...): normalise by the qkv GEMM count instead
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def name(r): return re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:70]
att = [r for r in rows if "attn_bwd_kernel<0, 11" in r["Name"]]
steps = int(att[0]["Calls"]) / 12 if att else 1
print(f"steps traced ~{steps:.0f}; kernel time {tot/1e6/steps:.2f} ms per step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{float(r["TotalDurationNs"])/1e6/steps:7.3f} ms/step {int(r["Calls"])/steps:6.1f} calls/step  avg {float(r["AverageNs"])/1e3:7.1f} us  {name(r)}')
PY
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
