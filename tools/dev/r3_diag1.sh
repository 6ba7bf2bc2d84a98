#!/bin/bash
# dev (round 3, first GPU call): VALU issue-cost microbench, attention baseline, bench line, attention stamps
set -e
mkdir -p gpurun_out/r3
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/dev/valu_rate.hip -o /tmp/valu_rate
/tmp/valu_rate > gpurun_out/r3/valu_rate.txt 2>&1
python tools/dev/attn_bench.py > gpurun_out/r3/attn_bench_base.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench_base.json 2> gpurun_out/r3/bench_base.err
cd kuzushiji-vision_amd/csrc && touch attention.hip && make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DKZV_STAMPS" > /dev/null 2>&1 && cd ../..
python tools/dev/stamps_attn.py > gpurun_out/r3/stamps_attn.txt 2>&1
cat gpurun_out/r3/valu_rate.txt gpurun_out/r3/attn_bench_base.txt gpurun_out/r3/bench_base.json gpurun_out/r3/stamps_attn.txt
