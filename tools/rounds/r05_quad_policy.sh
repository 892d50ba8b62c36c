#!/bin/bash
# Round 5: cache policy of the persistent 4-wave kernel's DMA pieces per operand.  PMC says a qkv / c_fc launch fetches 354 MB
# where 75 MB are algorithmic: every supertile brings its 4 MB of A panels and its 2 MB of W panels in from beyond L2, although
# an XCD's consecutive supertiles use the SAME W panels.  A pieces non-temporal (evict first) should keep W resident.
# Bit-identical results; same box, the kernels alone (tools/ab_quad.py), then the step.
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
run() { FORMS="4 waves, persistent" ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 | grep -E "^qkv|^fc  " | grep -v "no deferred" | cut -c1-16,57-140; }
for V in "" "-DKEDS_QUAD_AUX_X=2" "-DKEDS_QUAD_AUX_W=2" "-DKEDS_QUAD_AUX_X=2 -DKEDS_QUAD_AUX_W=2" "-DKEDS_QUAD_AUX_X=18" "-DKEDS_QUAD_AUX_X=1" ""; do
  if build "$V"; then echo "### ${V:-product}"; run; fi
done
restore
trap - EXIT
