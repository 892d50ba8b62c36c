#!/bin/bash
# Round 5: the two-accumulator-set kernel with its in-loop output stores under each cache policy (make EXTRA="-DKEDS_ST_DUO=n":
# 0 plain, 1 nt, 3 sc1 nt = the LayerNorm-epilogue policy of the other kernels), same box; correctness first.
set -u
export KEDS_GEMM_DUO=1
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="${FORMS:-4 waves, persistent;dispatcher}" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | grep -v "no deferred"; }
echo "### product (KEDS_ST_DUO = KEDS_ST_LN = 3: sc1 nt)"; python tools/duo_debug.py 2>&1 | grep -E "^M " ; run
for V in "-DKEDS_ST_DUO=0" "-DKEDS_ST_DUO=1" "-DKEDS_DUO_DBG=8"; do
  if build "$V"; then echo "### $V"; run; fi
done
restore
trap - EXIT
