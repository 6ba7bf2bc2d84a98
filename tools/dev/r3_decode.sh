#!/bin/bash
# dev: the one-launch decoder step -- tests that generate, then the generation bench with and without it
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_trained_gpu.py tests/test_beam_gpu.py tests/test_model_gpu.py tests/test_buckets_gpu.py -x -q -m gpu > gpurun_out/r3/decode_tests.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -15 gpurun_out/r3/decode_tests.txt
[ $rc -ne 0 ] && exit $rc
for one in 0 1; do
  echo "KZV_DECODE_ONE_LAUNCH=$one"
  KZV_DECODE_ONE_LAUNCH=$one DEC_GRAPHS=1 timeout -k 10 300 python tools/dev/decode_bench.py 2>&1 | grep graph= || exit 1
done | tee gpurun_out/r3/decode_bench.txt
