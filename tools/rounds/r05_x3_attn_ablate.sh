#!/bin/bash
# Round 5: where the split-operand attention's time goes -- timing-only builds (KEDS_AX_DBG: 1 staging alone, 2 no staging,
# 4 no exponentials, 8 no MFMAs), same box.
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
for V in "" "-DKEDS_AX_DBG=1" "-DKEDS_AX_DBG=2" "-DKEDS_AX_DBG=6" "-DKEDS_AX_DBG=10" "-DKEDS_AX_DBG=14" ""; do
  if build "$V"; then echo "### ${V:-product}"; timeout 300 python tools/ab_attn_x3.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; fi
done
restore
trap - EXIT
