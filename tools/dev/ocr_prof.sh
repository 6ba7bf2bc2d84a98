#!/bin/bash
# dev (GPU box): kernel-trace statistics of the OCR optimisation step (tools/dev/ocr_bench.py) -> gpurun_out/ocr_prof/
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ocr_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ocr -- python3 $ROOT/tools/dev/ocr_bench.py --steps 10 --cpu-steps 0 "$@" > $OUT/run.log 2>&1
cd $ROOT
tail -1 $OUT/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ocr_prof/*kernel_stats.csv") + glob.glob("gpurun_out/ocr_prof/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time {tot/1e6/13:.2f} ms per step, {calls/13:.0f} launches per step (13 steps traced)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6/13:7.3f} ms/step {int(r["Calls"])/13:6.1f} calls/step  avg {float(r["AverageNs"])/1e3:7.1f} us  {r["Name"][:90]}')
PY
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
