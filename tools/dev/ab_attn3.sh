#!/bin/bash
# dev: A/B of the shipped attention.hip against an older copy placed at tools/dev/_attention_prev.hip (e.g. `git show HEAD~1:kuzushiji-vision_amd/csrc/attention.hip > tools/dev/_attention_prev.hip`), same box
set -e
mkdir -p /tmp/kzv_b2/x && rm -rf /tmp/kzv_b2/x/csrc /tmp/kzv_b2/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b2/x/csrc && mkdir -p /tmp/kzv_b2/x/kzv /tmp/kzv_b2/include && cp include/kzv.h /tmp/kzv_b2/include/
cp tools/dev/_attention_prev.hip /tmp/kzv_b2/x/csrc/attention.hip
make -C /tmp/kzv_b2/x/csrc -j16 > /tmp/kzv_b2/build.log 2>&1 || { tail -20 /tmp/kzv_b2/build.log; exit 1; }
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -2
for i in 1 2; do
  echo "== A (shipped)"; python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
  echo "== B (previous)"; KZV_LIB=/tmp/kzv_b2/x/kzv/libkzv.so python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
done
