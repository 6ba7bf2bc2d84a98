#!/bin/bash
# dev: whole GPU suite + bench line
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3/gpu_tests.txt
python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err
python -c "import json; d=json.load(open('gpurun_out/r3/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac']); print(d['roofline']['other_kernels'])"
