#!/bin/bash
# dev (round 3): attention parity tests, per-launch times, the bench line, then the in-kernel timeline (-DKZV_STAMPS)
set -e
mkdir -p gpurun_out/r3
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r3/attn_tests.txt 2>&1 || { tail -40 gpurun_out/r3/attn_tests.txt; exit 1; }
tail -3 gpurun_out/r3/attn_tests.txt
python tools/dev/attn_bench.py > gpurun_out/r3/attn_bench.txt 2>&1
cat gpurun_out/r3/attn_bench.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err
python -c "import json; d=json.load(open('gpurun_out/r3/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other_kernels'])"
cd kuzushiji-vision_amd/csrc && touch attention.hip && make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DKZV_STAMPS" > /dev/null 2>&1 && cd ../..
python tools/dev/stamps_attn.py > gpurun_out/r3/stamps_attn.txt 2>&1
cat gpurun_out/r3/stamps_attn.txt
