#!/bin/bash
# Round 5: what would a DMA piece without an address VGPR buy?  The persistent 4-wave kernel with its LDS-DMA pieces addressed by
# the descriptor's ADD_TID_ENABLE (lane l reads base + offset + 16 l: 1 KiB contiguous, the traffic shape of a tiled operand
# layout) instead of a per-lane offset register -- timing only (operands wrong), same box, with and without the epilogue.
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="4 waves, persistent" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | grep -v "no deferred" | cut -c1-16,57-140; }
for V in "" "-DKEDS_QUAD_TIDDMA=1" "-DKEDS_QUAD_NOEPI=1" "-DKEDS_QUAD_NOEPI=1 -DKEDS_QUAD_TIDDMA=1"; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
