#!/bin/bash
# dev: the batched LayerNorm folds and the three-launch embed_assemble_bwd -- gradient parity tests, then the step's kernel stats
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_parity_gpu.py tests/test_buckets_gpu.py tests/test_ops_gpu.py tests/test_decoder_chain_gpu.py -x -q -m gpu > gpurun_out/r3/small_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r3/small_tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/dev/r3_chain_prof.sh 2>&1 | grep -E "kernels per step|embed_assemble|ln_bwd_reduce|ln_bwd_fast_kernel<3"
for i in 1 2; do python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['roofline']['events'])"; done
