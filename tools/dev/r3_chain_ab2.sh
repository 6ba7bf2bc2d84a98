#!/bin/bash
for s in 1 3 4; do
for c in 0 1; do
  echo -n "stride=$s KZV_DEC_CHAIN=$c: "
  KZV_BENCH_EVENT_STRIDE=$s KZV_DEC_CHAIN=$c timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), r['launches_per_step'], round(r['avg_launch_us'],2), round(r['kernel_ms_per_step'],3), r['events'])" || exit 1
done
done
