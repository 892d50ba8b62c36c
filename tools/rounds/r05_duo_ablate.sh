#!/bin/bash
# Round 5: what the K-loop of the two-accumulator-set kernel (gemm_duo.hip) costs by itself, and what each ingredient adds.
# Timing-only builds (make EXTRA="-DKEDS_DUO_DBG=n"; results wrong), same box:
#   0 product | 1 no sub-slices (K-loop alone) | 9.. | 3 no sub-slices, no DMA pieces | 5 no sub-slices, no fragment reads | 8 sub-slices without stores
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="${FORMS:-4 waves, persistent;dispatcher}" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " ; }
echo "### product"; run
for V in "-DKEDS_DUO_DBG=1" "-DKEDS_DUO_DBG=3" "-DKEDS_DUO_DBG=5" "-DKEDS_DUO_DBG=7" "-DKEDS_DUO_DBG=8"; do
  if build "$V"; then echo "### $V"; run; fi
done
restore
trap - EXIT
