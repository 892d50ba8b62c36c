#!/bin/bash
# Round 5, second pass of tools/rounds/r05_tid_dma.sh: WHICH operand's pieces pay when they need no address VGPR?  Bit 1 = the A
# operand (a tiled activation format: every producing epilogue changes), bit 2 = W (a tiled weight copy made at packing time:
# no other kernel changes).  Timing only (operands wrong, same bytes), same box, all four block GEMMs.
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="4 waves, persistent" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  |^out|^proj  " | grep -v "no deferred" | cut -c1-16,57-140; }
for V in "" "-DKEDS_QUAD_TIDDMA=2" "-DKEDS_QUAD_TIDDMA=1" "-DKEDS_QUAD_TIDDMA=3" ""; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
