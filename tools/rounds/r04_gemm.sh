#!/bin/bash
# round-4 GEMM experiment run (GPU box, repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### tools/gemm_instep.py"; python tools/gemm_instep.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_gemm_instep2.txt 2>&1
cat $O/r04_gemm_instep2.txt
