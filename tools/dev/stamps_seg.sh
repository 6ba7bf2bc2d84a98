#!/bin/bash
# dev (GPU box): build a -DKZV_STAMPS variant and print the segment timeline for the four instances
set -e
for inst in "8 1" "24 1" "24 3"; do
  set -- $inst
  rm -rf /tmp/kzv_s && mkdir -p /tmp/kzv_s/x /tmp/kzv_s/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_s/x/csrc && mkdir -p /tmp/kzv_s/x/kzv && cp include/kzv.h /tmp/kzv_s/include/
  rm -rf /tmp/kzv_s/x/csrc/build
  make -C /tmp/kzv_s/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DKZV_STAMPS -DKZV_STAMP_SEG_KS1=$1 -DKZV_STAMP_SEG_NP2=$2" > /tmp/kzv_s/build.log 2>&1 || { tail -20 /tmp/kzv_s/build.log; exit 1; }
  echo "<$1, $2>"; KZV_LIB=/tmp/kzv_s/x/kzv/libkzv.so python tools/dev/stamps_seg.py 2>&1 | grep -v amdgpu.ids
done
