#!/bin/bash
# dev: per-kernel times of the generation bench (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dprof
DEC_GRAPHS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dprof -- python3 $GRAFT_REPO_ROOT/tools/dev/decode_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r3/decode_prof_run.txt 2>&1
f=$(find /tmp/dprof -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3/decode_kernel_stats.csv
grep graph= $GRAFT_REPO_ROOT/gpurun_out/r3/decode_prof_run.txt
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
