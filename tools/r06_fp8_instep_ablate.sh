#!/bin/bash
# round 6: where the fp8 residual GEMMs' in-step penalty goes (out-proj 91 us in the step against 58 alone, c_proj 166 against 122):
# the epilogue ablations of gemm_fp8.hip (timing-only builds, results are wrong) measured IN THE STEP instead of alone
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
BENCH_ARGS='--precision fp8 --no-verify --no-legs' RUNS=2 bash tools/ab_build.sh "-DKEDS_FQ_ABL=4" "-DKEDS_FQ_ABL=8" "-DKEDS_FQ_ABL=16" "-DKEDS_FQ_ABL=32" "-DKEDS_FQ_ABL=60" "-DKEDS_FQ_ABL=1" 2>&1 | grep -v amdgpu.ids | tee $O/fp8_instep_ablation.txt
