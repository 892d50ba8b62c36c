#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### attention: L2 prefetch of the next round's rows (every variant must reproduce the embedding checksum of the base)"
  BENCH_ARGS="--prof-all" RUNS=2 bash tools/ab_build.sh "-DKEDS_ATTN_PREFETCH=512" "-DKEDS_ATTN_PREFETCH=256" "-DKEDS_ATTN_PREFETCH=1024"; } > $O/r04_attn_prefetch_ab.txt 2>&1
grep -v "amdgpu.ids" $O/r04_attn_prefetch_ab.txt
