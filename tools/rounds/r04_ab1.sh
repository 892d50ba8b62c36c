#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### is it the cache policy or the placement of the stores?  (base = builtin buffer store sc1 nt)"; RUNS=2 bash tools/ab_build.sh "-DKEDS_ST_LN=13" "-DKEDS_ST_LN=10" "-DKEDS_ST_LN=11" "-DKEDS_ST_LN=23" "-DKEDS_ST_LN=21" "-DKEDS_ST_LN=1"; } > $O/r04_store_policy_ab3.txt 2>&1
cat $O/r04_store_policy_ab3.txt
