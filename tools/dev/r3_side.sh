#!/bin/bash
# dev (round 3): side measurements quoted in DESIGN.md -- the one-rank RCCL rehearsal of the DP branch, configs[3] in both decoder
# variants, generation times
mkdir -p gpurun_out/r3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dp-rehearsal > gpurun_out/r3/bench_dp_rehearsal.json 2> gpurun_out/r3/bench_dp_rehearsal.err; echo "dp rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r3/bench_dp_rehearsal.json')); print(d['value'], d['ms_per_step']); print(d['dp_rehearsal'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --encoder vit_l --dec-layers 12 > gpurun_out/r3/bench_vitl.json 2> gpurun_out/r3/bench_vitl.err; echo "vitl rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r3/bench_vitl.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_TFLOP/s'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --encoder vit_l --dec-layers 12 --wide-decoder > gpurun_out/r3/bench_vitl_wide.json 2> gpurun_out/r3/bench_vitl_wide.err; echo "vitl wide rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r3/bench_vitl_wide.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_TFLOP/s'], d['config']['workload'][:160])"
python tools/dev/decode_bench.py > gpurun_out/r3/decode_bench.log 2>&1; tail -4 gpurun_out/r3/decode_bench.log
