#!/bin/bash
# Round 5, item 1 of the review: the bound for "the epilogue under the MFMAs" on the bf16 / fp16 256 x 256 GEMMs, as round 4
# measured it for fp8.  Same box, the library rebuilt per variant (timing-only builds: results are wrong):
#   base | no epilogue at all (K-loop + prologue only) | 1 / 2 plain vector instructions in EVERY K-loop gap | 1 plain + 1
#   transcendental per gap (what a QuickGELU epilogue spread over the next half-tile's K-loop would add)
# Run on the GPU box from the repo root: bash tools/rounds/r05_noepi_bound.sh > gpurun_out/r05_noepi_bound.txt 2>&1
set -u
build() {   # $1 = extra flags; a failed build is reported and SKIPPED (the previous library must not run under the new label)
  (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }
}
restore() { build "" || true; }
trap restore EXIT
run() { ROUNDS=${ROUNDS:-5} ITERS=${ITERS:-20} timeout 600 python tools/ab_quad.py 2>&1 ; }
echo "### base"; run
for V in "-DKEDS_QUAD_NOEPI=1" "-DKEDS_QUAD_FILL=1" "-DKEDS_QUAD_FILL=2" "-DKEDS_QUAD_FILL=1 -DKEDS_QUAD_FILLX=1" "-DKEDS_QUAD_FILL=2 -DKEDS_QUAD_FILLX=1"; do
  if build "$V"; then echo "### $V"; run; fi
done
restore
trap - EXIT
echo "### base again"; run
