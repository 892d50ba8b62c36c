#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### runtime switch: out-proj on the 4-wave kernel (KEDS_RESID_QUAD_K=1024)"
  for i in 1 2; do for k in 2048 1024; do echo -n "KEDS_RESID_QUAD_K=$k "; KEDS_RESID_QUAD_K=$k python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-140; done; done
  echo "### load policies / tile walk (base = sc1 nt stores of the LayerNorm-epilogue outputs)"
  RUNS=2 bash tools/ab_build.sh "-DKEDS_LD_RESID_AUX=2" "-DKEDS_LD_RESID_AUX=18" "-DKEDS_LD_RESID_AUX=16" "-DKEDS_LD_A3_AUX=2" "-DKEDS_LD_ATTN_AUX=2" "-DKEDS_SUPER_M_LOG2=2" "-DKEDS_SUPER_M_LOG2=4"; } > $O/r04_load_policy_ab.txt 2>&1
cat $O/r04_load_policy_ab.txt
