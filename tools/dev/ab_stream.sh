set -e
cd kuzushiji-vision_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast"
for v in plain stream plain stream; do
  touch gemm.hip gemm_nt256.hip gemm_nt256p.hip
  if [ $v = stream ]; then make FLAGS="$F -DKZV_NT_STREAM_STORES" -j8 > /dev/null 2>&1; else make FLAGS="$F" -j8 > /dev/null 2>&1; fi
  cd ../..
  echo "=== $v nt256"; python tools/dev/gemm_bench.py nt 2>&1 | grep -v amdgpu.ids | grep "41216\|weighted"
  echo "=== $v nt256p"; KZV_NT256P=1 python tools/dev/gemm_bench.py nt 2>&1 | grep -v amdgpu.ids | grep "41216\|weighted"
  cd kuzushiji-vision_amd/csrc
done
