#!/bin/bash
# Round 5: what an output store costs inside the K-loop of the two-accumulator-set kernel, and why (timing-only builds).
set -u
export KEDS_GEMM_DUO=1
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="dispatcher" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | cut -c1-16,77-140; }
for V in "" "-DKEDS_DUO_DBG=8" "-DKEDS_DUO_DBG=64" "-DKEDS_DUO_DBG=32" "-DKEDS_DUO_DBG=16 -DKEDS_ST_DUO=0" "-DKEDS_DUO_DBG=16" "-DKEDS_DUO_DBG=32 -DKEDS_ST_DUO=0"; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
