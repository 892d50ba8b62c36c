#!/bin/bash
# Same-box A/B of compile-time variants: bench, rebuild the library on the GPU box with -D<MACRO>, bench, ... , base again.
# Usage (on the box, from the repo root): [BENCH_ARGS='--precision fp8'] bash tools/ab_nt.sh MACRO1 MACRO2 ...   (round 2: the non-temporal store / load hints)
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline $BENCH_ARGS"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result"
run() { for i in 1 2; do $B 2>&1 | tail -1 | cut -c60-180; done; }
echo "base"; run
for V in "$@"; do
  (cd keds_amd/csrc; rm -f build/gemm.o build/gemm_fp8.o build/attention.o build/search.o; make CXXFLAGS="$F -D$V" > /tmp/mk.log 2>&1)
  echo "+ $V"; run
done
(cd keds_amd/csrc; rm -f build/gemm.o build/gemm_fp8.o build/attention.o build/search.o; make CXXFLAGS="$F" > /tmp/mk.log 2>&1)
echo "base again"; run
