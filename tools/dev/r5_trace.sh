#!/bin/bash
# dev (GPU box): kernel trace of a few TRIMMED bench steps -> gpurun_out/r5_trace/ (timeline of one step by tools/dev/r5_timeline.py)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KZV_BENCH_NO_UNTRIMMED=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
ls -la $OUT
