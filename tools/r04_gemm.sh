#!/bin/bash
# round-4 GEMM experiment run (GPU box, repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "quad_kernel_bit_identical" 2>&1 | tail -3
{ echo "### tools/gemm_cold_matrix.py"; python tools/gemm_cold_matrix.py 2>&1 | grep -v amdgpu.ids
  echo; echo "### tools/gemm_instep.py"; python tools/gemm_instep.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_gemm_cold.txt 2>&1
cat $O/r04_gemm_cold.txt
python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | cut -c1-150
