#!/bin/bash
# dev: A/B of the shipped build against a variant built with extra flags, same box, alternating runs.
#   usage (on the GPU box): bash tools/dev/ab_build.sh "-DKZV_NT_PLAIN_STORES" [bench args]
set -e
EXTRA="$1"; shift || true
mkdir -p /tmp/kzv_b && rm -rf /tmp/kzv_b/* && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b/csrc && mkdir -p /tmp/kzv_b/kzv && mkdir -p /tmp/kzv_b/include && cp include/kzv.h /tmp/kzv_b/include/
# the copied tree keeps csrc's relative include of ../../include/kzv.h: mirror the depth
mkdir -p /tmp/kzv_b2/x && rm -rf /tmp/kzv_b2/x/csrc /tmp/kzv_b2/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b2/x/csrc && mkdir -p /tmp/kzv_b2/x/kzv /tmp/kzv_b2/include && cp include/kzv.h /tmp/kzv_b2/include/
rm -rf /tmp/kzv_b2/x/csrc/build
make -C /tmp/kzv_b2/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $EXTRA" > /tmp/kzv_b2/build.log 2>&1 || { tail -20 /tmp/kzv_b2/build.log; exit 1; }
for i in $(seq 1 ${AB_ROUNDS:-2}); do
  echo "== A (shipped)"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v['ms_per_step'],3) for k,v in d['roofline']['other_kernels'].items()})"
  echo "== B ($EXTRA)"; KZV_LIB=/tmp/kzv_b2/x/kzv/libkzv.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v['ms_per_step'],3) for k,v in d['roofline']['other_kernels'].items()})"
done
