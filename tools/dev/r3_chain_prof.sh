#!/bin/bash
mkdir -p gpurun_out/r3
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -o c -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/r3/chain_prof_run.txt 2>&1
cp $(find /tmp/cprof -name 'c_kernel_stats.csv' | head -1) $ROOT/gpurun_out/r3/chain_kernel_stats.csv
python3 - $ROOT/gpurun_out/r3/chain_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernels per step ms', tot / 8 / 1e6)
for r in rows[:40]:
    n = int(r['Calls']) / 8
    print(f"{r['Name'][:84]:84s} /step {n:6.1f} avg {float(r['AverageNs'])/1e3:8.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
