#!/bin/bash
# dev: A/B of an environment setting on the bench line, same box, alternating.  usage: ab_env2.sh "VAR=value [VAR2=value2]"
for i in 1 2; do
  echo "== A (default)"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v['ms_per_step'],3) for k,v in d['roofline']['other_kernels'].items()})"
  echo "== B ($1)"; env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v['ms_per_step'],3) for k,v in d['roofline']['other_kernels'].items()})"
done
