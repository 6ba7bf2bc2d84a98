#!/bin/bash
# dev (GPU box): gemm_tn_group durations in a traced step for several workgroup budgets (KZV_TN_BLOCKS)
ROOT=$(pwd)
for b in 512 768 1024 1536 2048; do
  OUT=$ROOT/gpurun_out/r5_trace; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && KZV_TN_BLOCKS=$b KZV_BENCH_NO_UNTRIMMED=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1)
  f=$(find $OUT -name "t_kernel_trace.csv" | head -1)
  echo "KZV_TN_BLOCKS=$b: $(python tools/dev/r5_timeline.py $f v | grep -E "gemm_tn_group_kernel  grid" | awk '{printf "%s us (grid %s) ", $2, $NF}')"
  rm -rf $OUT
done
