#!/bin/bash
# dev: same-box A/B of the shipped library against tools/dev/_ab/libkzv_prev.so (the previous commit's build; not tracked)
for i in 1 2 3; do
  echo -n "prev: "; KZV_LIB=$(pwd)/tools/dev/_ab/libkzv_prev.so python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" || exit 1
  echo -n "new:  "; python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" || exit 1
done
