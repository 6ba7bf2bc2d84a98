#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_model_gpu.py tests/test_parity_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r3/misc_tests.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/r3/misc_tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/stats2 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3/stats2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3/stats2/**/b_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("total ms/step", sum(float(r['TotalDurationNs']) for r in rows) / 8e6)
for r in rows:
    if any(k in r['Name'] for k in ('ce_', 'embed_', 'attn_', 'tn256', 'gemm_tn_kernel')):
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
rm -rf gpurun_out/r3/stats2/*/b_kernel_trace.csv gpurun_out/r3/stats2/b_kernel_trace.csv
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
