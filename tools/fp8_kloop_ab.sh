#!/bin/bash
# what a K-tile of the 4-wave MXFP8 kernel costs without its memory side: stamped timing-only rebuilds ON THE GPU BOX
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
for V in ${ABLS:-0 64 128 192 256 448 512 1024 2048}; do
  (cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F -DKEDS_FQ_ABL=$V -DKEDS_FQ_STAMP" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log)
  echo "KEDS_FQ_ABL=$V (64: no DMA pieces, 128: no fragment reads, 256: no wait + barrier, 512: wait only, 1024: lgkmcnt + barrier, 2048: barrier only)"; N=3072 timeout 120 python tools/fp8_stamp.py 2>&1 | grep -E "K-loop|tile "
done | tee $O/fp8_kloop_ablation.txt
(cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F" > /tmp/mk.log 2>&1)
