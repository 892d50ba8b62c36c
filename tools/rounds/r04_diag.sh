#!/bin/bash
# round-4 diagnostic run (GPU box, repo root): where the tower GEMMs lose time between a stand-alone loop and the step.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
python bench.py --steps 40 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; echo "bench rc=$?"; cut -c1-200 $O/bench_base.json
{ echo "### tools/gemm_instep.py"; python tools/gemm_instep.py 2>&1 | grep -v amdgpu.ids
  echo; echo "### tools/cold_gemm.py (hot = back to back, cold = behind a 1 GiB fill)"; python tools/cold_gemm.py 2>&1 | grep -v amdgpu.ids
  echo; echo "### rocprofv3 kernel stats of bench.py, side lane ON"; bash tools/kstats.sh 2>&1 | grep -v amdgpu.ids
  echo; echo "### rocprofv3 kernel stats of bench.py, side lane OFF (KEDS_SIDE_STREAM=0)"; KEDS_SIDE_STREAM=0 bash tools/kstats.sh 2>&1 | grep -v amdgpu.ids
} > $O/r04_gemm_hot_cold.txt 2>&1
tail -60 $O/r04_gemm_hot_cold.txt
