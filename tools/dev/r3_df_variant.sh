#!/bin/bash
# dev: the decode tests against a variant build (extra flags in $1)
rm -rf /tmp/kzv_b2 && mkdir -p /tmp/kzv_b2/x/kzv /tmp/kzv_b2/include && cp -r kuzushiji-vision_amd/csrc /tmp/kzv_b2/x/csrc && cp include/kzv.h /tmp/kzv_b2/include/ && rm -rf /tmp/kzv_b2/x/csrc/build
make -C /tmp/kzv_b2/x/csrc -j16 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $1" > /tmp/kzv_b2/build.log 2>&1 || { tail -20 /tmp/kzv_b2/build.log; exit 1; }
KZV_LIB=/tmp/kzv_b2/x/kzv/libkzv.so timeout -k 10 600 python -m pytest tests/test_decode_fused_gpu.py -q -m gpu -s 2>&1 | grep -E "rows per image|passed|failed|agreement|Error"
