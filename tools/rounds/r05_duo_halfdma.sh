#!/bin/bash
# Round 5: is the cost of the in-loop output stores a queueing effect?  The two-accumulator-set kernel with every other DMA piece
# dropped (timing only: half the operand bytes, results wrong), with and without its stores, same box.
set -u
export KEDS_GEMM_DUO=1
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="dispatcher" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | cut -c1-16,77-140; }
for V in ${VARIANTS:-"" "-DKEDS_DUO_DBG=8" "-DKEDS_DUO_DBG=128" "-DKEDS_DUO_DBG=136"}; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
