#!/bin/bash
set -e
mkdir -p gpurun_out/r3
python tools/dev/attn_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3/attn_probe.txt
cat gpurun_out/r3/attn_probe.txt
python -m pytest tests/test_model_gpu.py -x -q -k "wide_decoder or vit_large" > gpurun_out/r3/wide.txt 2>&1 || { tail -40 gpurun_out/r3/wide.txt; exit 1; }
tail -3 gpurun_out/r3/wide.txt
