#!/bin/bash
mkdir -p gpurun_out/r3
for i in 1 2 3; do
for c in 0 1; do
  echo -n "KZV_DEC_CHAIN=$c: "
  KZV_DEC_CHAIN=$c timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline'].get('whole_step_frac',0),4))" || exit 1
done
done | tee gpurun_out/r3/chain_ab.txt
