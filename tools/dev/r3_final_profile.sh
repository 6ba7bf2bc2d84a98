#!/bin/bash
# round 3 profile set (one box, one call): tools/profile_round.sh r03 + the generation step's kernel stats
mkdir -p gpurun_out/r3
bash tools/profile_round.sh r03 > gpurun_out/r3/profile_round.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r3/profile_round.log
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r03/decode -o d -- python3 $ROOT/tools/dev/decode_bench.py > $ROOT/gpurun_out/prof_r03/decode.log 2>&1; echo "decode prof rc=$?"
cd $ROOT
cp $(find gpurun_out/prof_r03/decode -name d_kernel_stats.csv | head -1) gpurun_out/prof_r03/decode_kernel_stats.csv
find gpurun_out/prof_r03/decode -name "d_kernel_trace.csv" -delete
tail -5 gpurun_out/prof_r03/decode.log
python -c "import json; d=json.load(open('gpurun_out/prof_r03/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'], d['cpu_baseline'])"
