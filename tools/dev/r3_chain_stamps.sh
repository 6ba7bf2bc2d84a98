#!/bin/bash
rm -rf /tmp/kzv_b2 && mkdir -p /tmp/kzv_b2/x/kzv /tmp/kzv_b2/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b2/x/csrc && cp include/kzv.h /tmp/kzv_b2/include/ && rm -rf /tmp/kzv_b2/x/csrc/build
make -C /tmp/kzv_b2/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DKZV_STAMPS" > /tmp/kzv_b2/build.log 2>&1 || { tail -20 /tmp/kzv_b2/build.log; exit 1; }
mkdir -p gpurun_out/r3
KZV_LIB=/tmp/kzv_b2/x/kzv/libkzv.so timeout -k 10 300 python tools/dev/stamps_chain.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/chain_stamps.txt
