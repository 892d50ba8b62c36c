#!/bin/bash
# Same-box A/B of compile-time variants: bench, rebuild the library ON THE GPU BOX with extra flags, bench, ..., base again.
# Usage (on the box, from the repo root): [BENCH_ARGS='--precision fp8'] [RUNS=2] bash tools/ab_build.sh "-DA=1" "-DA=1 -DB=2" ...
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline $BENCH_ARGS"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
OBJS="build/gemm.o build/gemm_fp8.o build/attention.o build/search.o"
run() { for i in $(seq ${RUNS:-2}); do $B 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   %.0f img/s  %.3f ms/step  gemm %.3f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['gemm'], d['roofline']['frac']))"; done; }
echo "base"; run
for V in "$@"; do
  (cd keds_amd/csrc; rm -f $OBJS; make -j8 CXXFLAGS="$F $V" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log)
  echo "+ $V"; run
done
(cd keds_amd/csrc; rm -f $OBJS; make -j8 CXXFLAGS="$F" > /tmp/mk.log 2>&1)
echo "base again"; run
