#!/bin/bash
# Round 5 iteration helper for csrc/attention_x3.hip: builds with EXTRA variants on the box, times the kernel alone.
set -u
build() { (cd keds_amd/csrc && make -j8 EXTRA="$1" > /tmp/mk.log 2>&1) || { echo "BUILD FAILED: $1"; tail -5 /tmp/mk.log; return 1; }; }
restore() { build "" || true; }
trap restore EXIT
for V in "$@"; do
  if build "$V"; then echo "### ${V:-product}"; timeout 300 python tools/ab_attn_x3.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; fi
done
restore
trap - EXIT
