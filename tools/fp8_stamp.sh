#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${ROUND:-r04}; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
(cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F -DKEDS_FQ_STAMP" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log)
for n in 3072 4096; do echo "N = $n"; N=$n timeout 300 python tools/fp8_stamp.py 2>&1 | grep -v amdgpu.ids; done | tee $O/fp8_stamps.txt
(cd keds_amd/csrc; rm -f build/gemm_fp8.o; make -j8 CXXFLAGS="$F" > /tmp/mk.log 2>&1)
