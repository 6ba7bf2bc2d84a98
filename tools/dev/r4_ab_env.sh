#!/bin/bash
# dev (GPU box): bench.py under two environments, alternating: bash tools/dev/r4_ab_env.sh "A=1" "KZV_PAIR=0" [rounds]
run() { echo -n "$1: "; env $1 KZV_BENCH_NO_UNTRIMMED=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"; }
for i in $(seq 1 ${3:-3}); do run "$1"; run "$2"; done
