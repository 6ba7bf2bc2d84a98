#!/bin/bash
# dev (GPU box): build the csrc snapshots under tmp_ab/<commit>/ and run r5_segdiff.py on each
for d in tmp_ab/*/; do
  rm -rf $d/x/csrc/build
  make -C $d/x/csrc -j16 > $d/build.log 2>&1 || { tail -5 $d/build.log; continue; }
  echo "== $d"; KZV_LIB=$(pwd)/$d/x/kzv/libkzv.so R5_LD=12 R5_B=5 timeout -k 10 200 python tools/dev/r5_segdiff.py 2>&1 | grep -v amdgpu | grep "launches vs" | cut -c1-330
done
