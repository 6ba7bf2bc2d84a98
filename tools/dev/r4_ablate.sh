#!/bin/bash
# dev (GPU box): the shipped build (bit-equality + timing), then each ablation variant of tools/dev/_ab (timing only)
set -e
mkdir -p gpurun_out
echo "== shipped" ; timeout -k 10 300 python tools/dev/r4_free.py ${1:-all}
for v in tools/dev/_ab/libkzv_*.so; do
  echo "== $v"; KZV_LIB=$PWD/$v timeout -k 10 200 python tools/dev/r4_free.py time2
done
