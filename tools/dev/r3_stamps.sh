#!/bin/bash
# dev: attention in-kernel timelines (-DKZV_STAMPS build of attention.hip only)
set -e
mkdir -p gpurun_out/r3
cd kuzushiji-vision_amd/csrc && touch attention.hip && make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DKZV_STAMPS" > /dev/null 2>&1 && cd ../..
python tools/dev/stamps_attn.py > gpurun_out/r3/stamps_attn2.txt 2>&1
cat gpurun_out/r3/stamps_attn2.txt
