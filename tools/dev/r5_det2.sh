#!/bin/bash
set -e
echo "shipped:"; R5_LOGITS_ONLY=1 python tools/dev/r5_det2.py 2>&1 | grep -v amdgpu.ids
for f in "$@"; do
  rm -rf /tmp/kzv_s && mkdir -p /tmp/kzv_s/x /tmp/kzv_s/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_s/x/csrc && mkdir -p /tmp/kzv_s/x/kzv && cp include/kzv.h /tmp/kzv_s/include/
  rm -rf /tmp/kzv_s/x/csrc/build
  make -C /tmp/kzv_s/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -D$f" > /tmp/kzv_s/build.log 2>&1 || { tail -20 /tmp/kzv_s/build.log; exit 1; }
  echo "$f:"; R5_LOGITS_ONLY=1 KZV_LIB=/tmp/kzv_s/x/kzv/libkzv.so python tools/dev/r5_det2.py 2>&1 | grep -v amdgpu.ids
done
