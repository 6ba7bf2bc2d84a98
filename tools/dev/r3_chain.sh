#!/bin/bash
# dev: the decoder chains -- A/B test, the parity tests at the 256-wide decoder, bench with and without
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_decoder_chain_gpu.py -x -q -m gpu -s 2>&1 | grep -E "loss|passed|failed|Error|error" | tee gpurun_out/r3/chain_tests.txt
grep -q failed gpurun_out/r3/chain_tests.txt && exit 1
grep -q passed gpurun_out/r3/chain_tests.txt || exit 1
for i in 1 2; do
for c in 0 1; do
  echo "KZV_DEC_CHAIN=$c"
  KZV_DEC_CHAIN=$c timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" || exit 1
done
done | tee gpurun_out/r3/chain_bench.txt
