#!/bin/bash
# dev (GPU box): A/B of store policies of the GEMM epilogues, each against the shipped build
for f in KZV_DGELU_PLAIN KZV_GELU_ACT_PLAIN KZV_BF16_PLAIN; do
  AB_ROUNDS=3 KZV_BENCH_NO_UNTRIMMED=1 bash tools/dev/ab_build.sh "-D$f" 2>&1 | grep -v amdgpu | cut -c1-34 | paste - - | awk -v f=$f '{print f, $0}'
done
