#!/bin/bash
# A/B of one environment knob on one box: tools/dev/ab_env.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python - "$var" "$v" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/ab_{sys.argv[2]}.json')); r=d['roofline']
o=r['other_kernels'].get('gemm_nt_fp8_kernel')
print(sys.argv[1], sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'nt', round(r['achieved']), round(r['avg_launch_us'],1), 'fp8', (round(o['ms_per_step'],2), round(o['TFLOP/s'])) if o else None)
PY
done
