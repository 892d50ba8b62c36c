#!/bin/bash
# K-loop against epilogue in the 4-wave MXFP8 kernel: timing-only rebuilds ON THE GPU BOX (KEDS_FQ_ABL, gemm_fp8.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
for V in ${ABLS:-0 1 2 3}; do
  (cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F -DKEDS_FQ_ABL=$V" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log)
  echo "KEDS_FQ_ABL=$V"; ROUNDS=2 timeout 300 python tools/fp8_forms.py 2>&1 | grep -v amdgpu.ids | grep -E "${SHAPES:-.}" | sed 's/8 waves:.*4 waves/4 waves/'
done | tee $O/fp8_ablate.txt
(cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F" > /tmp/mk.log 2>&1)
