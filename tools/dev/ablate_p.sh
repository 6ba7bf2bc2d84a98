#!/bin/bash
# dev: what bounds the persistent GEMM's K loop?  Rebuilds gemm_nt256p.hip on the GPU box with one ingredient removed at a time
# (results are garbage; only the in-kernel timeline of tools/dev/stamps_p.py is read) and prints the per-tile K-loop time.
#   A0 baseline | A1 no MFMA (fragments still read) | A2 no LDS reads | A3 no LDS-DMA | A4 no barriers | A5 MFMA only
set -e
SRC=kuzushiji-vision_amd/csrc/gemm_nt256p.hip
cp $SRC /tmp/p_orig.hip
# whatever happens below, leave the tree as it was: the pristine source back and libkzv.so rebuilt from it
trap 'cp /tmp/p_orig.hip $SRC; touch $SRC; make -C kuzushiji-vision_amd/csrc -j8 > /dev/null 2>&1' EXIT
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=fast -DKZV_STAMPS"
run() {
  make -C kuzushiji-vision_amd/csrc -j8 FLAGS="$FL" > /dev/null 2>&1
  echo "== $1"; KZV_NT256P=1 python tools/dev/stamps_p.py 2>/dev/null | grep -A1 "hot  41216x2304x768" | tail -1
}
patch_nomfma() { python - <<'PY'
import re
p='kuzushiji-vision_amd/csrc/gemm_nt256p.hip'; s=open(p).read()
s=s.replace("acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].k[kh], fa[i].k[kh], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);",
            'asm volatile("" :: "v"(fb[j].k[kh]), "v"(fa[i].k[kh]));')
open(p,'w').write(s)
PY
}
patch_noreads() { python - <<'PY'
p='kuzushiji-vision_amd/csrc/gemm_nt256p.hip'; s=open(p).read()
s=s.replace("f.k[0] = *(const bf16x8*)(q + slot0); f.k[1] = *(const bf16x8*)(q + slot1);", 'asm volatile("" : "=v"(f.k[0]), "=v"(f.k[1]));')
open(p,'w').write(s)
PY
}
patch_nodma() { python - <<'PY'
p='kuzushiji-vision_amd/csrc/gemm_nt256p.hip'; s=open(p).read()
s=s.replace('asm volatile("s_mov_b32 m0, %2\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");', 'asm volatile("" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");')
open(p,'w').write(s)
PY
}
patch_nobar() { sed -i '0,/^namespace {/s//#define __builtin_amdgcn_s_barrier() ((void)0)\nnamespace {/' $SRC; }
touch $SRC; run "A0 baseline"
cp /tmp/p_orig.hip $SRC; patch_nomfma; run "A1 no MFMA"
cp /tmp/p_orig.hip $SRC; patch_noreads; run "A2 no LDS reads"
cp /tmp/p_orig.hip $SRC; patch_nodma; run "A3 no LDS-DMA"
cp /tmp/p_orig.hip $SRC; patch_nobar; run "A4 no barriers"
cp /tmp/p_orig.hip $SRC; patch_noreads; patch_nodma; run "A5 MFMA + barriers only"
cp /tmp/p_orig.hip $SRC; patch_noreads; patch_nodma; patch_nobar; run "A6 MFMA only"
cp /tmp/p_orig.hip $SRC
