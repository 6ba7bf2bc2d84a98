#!/bin/bash
# dev: A/B of attention.hip built with extra flags against the shipped build, same box (attn_bench.py)
set -e
EXTRA="$1"
mkdir -p /tmp/kzv_b2/x && rm -rf /tmp/kzv_b2/x/csrc /tmp/kzv_b2/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b2/x/csrc && mkdir -p /tmp/kzv_b2/x/kzv /tmp/kzv_b2/include && cp include/kzv.h /tmp/kzv_b2/include/
touch /tmp/kzv_b2/x/csrc/attention.hip
make -C /tmp/kzv_b2/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $EXTRA" > /tmp/kzv_b2/build.log 2>&1 || { tail -20 /tmp/kzv_b2/build.log; exit 1; }
for i in 1 2; do
  echo "== A (shipped)"; python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
  echo "== B ($EXTRA)"; KZV_LIB=/tmp/kzv_b2/x/kzv/libkzv.so python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
done
