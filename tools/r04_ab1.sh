#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### store policies, second matrix (LN = 3 is the new reference)"; RUNS=2 bash tools/ab_build.sh "-DKEDS_ST_LN=3" "-DKEDS_ST_LN=4" "-DKEDS_ST_LN=5" "-DKEDS_ST_LN=3 -DKEDS_ST_RESID=3" "-DKEDS_ST_LN=3 -DKEDS_ST_RESID=4" "-DKEDS_ST_LN=3 -DKEDS_ST_RESID=1" "-DKEDS_ST_LN=3 -DKEDS_ST_ATTN=3" "-DKEDS_ST_LN=3 -DKEDS_ST_ATTN=4" "-DKEDS_ST_LN=3 -DKEDS_ST_ATTN=1"; } > $O/r04_store_policy_ab2.txt 2>&1
cat $O/r04_store_policy_ab2.txt
