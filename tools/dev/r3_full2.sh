#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3/gpu_tests.txt
bash tools/profile_round.sh r03 > gpurun_out/r3/profile_round.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/r3/profile_round.log
cat gpurun_out/prof_r03/bench_line.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'], d['cpu_baseline'])"
