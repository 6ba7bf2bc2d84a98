#!/bin/bash
# dev (build container): variant builds of ONE source file linked against the shipped objects -> tools/dev/_ab/libkzv_<tag>.so
#   usage: tools/dev/r4_variants.sh gemm_nt256f.hip tag1 "-DFLAG1" tag2 "-DFLAG2 -DFLAG3" ...
set -e
cd "$(dirname "$0")/../../kuzushiji-vision_amd/csrc"
src=$1; shift
obj=build/${src%.*}.o
mkdir -p ../../tools/dev/_ab
while [ $# -gt 0 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $flags -c $src -o /tmp/rp/var_$tag.o
  others=$(ls build/*.o | grep -v "^$obj$")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/dev/_ab/libkzv_$tag.so $others /tmp/rp/var_$tag.o
  echo "built _ab/libkzv_$tag.so ($flags)"
done
