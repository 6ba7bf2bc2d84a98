#!/bin/bash
# dev (GPU box): does running the weight gradients on the side stream pay once gemm_nt is NOT the static persistent kernel?
run() { echo -n "$1: "; env $2 KZV_BENCH_NO_UNTRIMMED=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
run "persistent, one stream      " "A=1"
run "persistent, side stream 1   " "KZV_SIDE_STREAM=1"
run "one tile per WG, one stream " "KZV_NT256P=0"
run "one tile per WG, side 1     " "KZV_NT256P=0 KZV_SIDE_STREAM=1"
run "one tile per WG, side 2     " "KZV_NT256P=0 KZV_SIDE_STREAM=2"
done
