#!/bin/bash
# Round 5: is the cost of the in-loop output stores the in-order vmcnt?  Timing-only builds whose K-tile waits leave 16 more
# operations in flight (the DMA data may not have landed: results wrong), with and without the stores, same box.
set -u
export KEDS_GEMM_DUO=1
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="dispatcher" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | cut -c1-16,77-140; }
for V in "" "-DKEDS_DUO_DBG=8" "-DKEDS_DUO_DBG=256" "-DKEDS_DUO_DBG=264"; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
