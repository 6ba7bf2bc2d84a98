#!/bin/bash
# Round profile set (run on the GPU box through gpurun, from the repo root): bench line, kernel-trace stats of the same
# command, the two HBM-traffic PMC passes and the MfmaUtil pass (each PMC pass on its own, with --kernel-trace only:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Outputs land in gpurun_out/prof_rNN/; copy the summaries into profiles/.
#   usage: bash tools/profile_round.sh r02
set -e
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
cd $ROOT
python tools/hbm_traffic.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv > $OUT/hbm_traffic.json
python tools/mfma_util.py $OUT/pmc_mfma/m_counter_collection.csv > $OUT/mfma_util.json
cp $OUT/stats/b_kernel_stats.csv $OUT/bench_kernel_stats.csv
# the bench line last, with this run's traffic summary in place (bench.py quotes roofline.traffic from profiles/ when the source hash matches)
cp $OUT/hbm_traffic.json $ROOT/profiles/${R}_hbm_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/stats/b_kernel_trace.csv   # the raw traces are large; the summaries stay
ls -la $OUT
