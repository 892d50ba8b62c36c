#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
for S in qkv proj; do for Q in 0 1; do SHAPE=$S QUAD=$Q timeout 300 python tools/stamp_gemm.py > $O/stamp_${S}_q$Q.txt 2>&1; echo "== $S quad=$Q"; grep -A9 "^== product epilogue:" $O/stamp_${S}_q$Q.txt | head -12; grep "unstamped" $O/stamp_${S}_q$Q.txt; done; done
