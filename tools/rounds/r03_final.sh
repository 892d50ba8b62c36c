#!/bin/bash
# round-3 evidence run (GPU box, repo root): full parity suite, the bench lines of every workload, rocprofv3 kernel stats of
# the bench command + the two PMC passes, search chain, GEMM A/B and stamps.  Everything lands under gpurun_out/r03/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/r03_bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python bench.py --workload dual --steps 40 > $O/r03_bench_dual_final.json 2>/dev/null; echo "dual rc=$?"
python bench.py --precision fp8 --db-rows 2000000 --steps 40 --no-cpu-baseline > $O/r03_bench_fp8_2m_final.json 2>/dev/null; echo "fp8 rc=$?"
python bench.py --precision fp8 --steps 40 --no-cpu-baseline > $O/r03_bench_fp8_final.json 2>/dev/null; echo "fp8 0.5M rc=$?"
KEDS_BENCH_FORCE_DIST=1 python bench.py --steps 60 --no-cpu-baseline > $O/r03_bench_dist1.json 2>/dev/null; echo "dist1 rc=$?"
KEDS_BENCH_FORCE_DIST=1 python bench.py --workload dual --steps 30 --no-cpu-baseline > $O/r03_bench_dual_dist1.json 2>/dev/null; echo "dual dist1 rc=$?"
timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/r03_bench_gpus2_on_one_gpu.out 2> $O/r03_bench_gpus2_on_one_gpu.err; echo "gpus2 rc=$? (expected: fails inside device enumeration)"; grep -i "invalid device\|ordinal" $O/r03_bench_gpus2_on_one_gpu.err | head -2
python tools/bench_train.py > $O/r03_bench_train.json 2>/dev/null
bash tools/profile_round.sh r03_final > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log
cp gpurun_out/r03_final_* $O/ 2>/dev/null
python tools/ab_quad.py > $O/r03_gemm_kernels_ab.txt 2>&1
{ for S in qkv proj; do echo "### SHAPE=$S, 4-wave kernel: K-loop ablations (cycles)"; SHAPE=$S python tools/stamp_quad.py 2>&1 | grep -v amdgpu.ids; done; for S in qkv out proj; do echo "### SHAPE=$S, 8-wave kernel timeline"; SHAPE=$S QUAD=0 python tools/stamp_gemm.py 2>&1 | grep -A8 "^== product epilogue:" | head -10; done; } > $O/r03_gemm_stamps.txt 2>&1
python tools/micro/power_clock_probe.py > $O/r03_power_clock_probe.txt 2>&1
{ for C in one shard; do CONFIG=$C NO_AB=1 python tools/search_profile.py 2>&1 | grep -v amdgpu.ids; done; CONFIG=one NO_AB=1 ITERS=10 bash tools/kstats_cmd.sh tools/search_profile.py 2>&1 | grep -v amdgpu.ids; } > $O/r03_search_chain.txt 2>&1
ls $O
