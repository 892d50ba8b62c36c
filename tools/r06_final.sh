#!/bin/bash
# round-6 evidence run (GPU box, repo root): full parity suite, the bench lines of every workload / operating point, rocprofv3
# kernel stats of the bench command + the two PMC passes, per-workload kernel tables.  Everything lands under gpurun_out/r06/.
# Every stage runs under its own timeout: a wedged stage must not take the rest of the budget with it.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
rm -f gpurun_out/gpu_test_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cp gpurun_out/gpu_test_metrics.jsonl $O/gpu_test_metrics.jsonl 2>/dev/null
timeout 600 python bench.py > $O/r06_bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_form.json 2>/dev/null; echo "bench (driver's arguments) rc=$?"
timeout 300 python bench.py --workload dual --steps 40 > $O/r06_bench_dual_final.json 2>/dev/null; echo "dual rc=$?"
timeout 300 python bench.py --precision fp8 --db-rows 2000000 --steps 40 --no-cpu-baseline --no-legs > $O/r06_bench_fp8_2m_final.json 2>/dev/null; echo "fp8 2M rc=$?"
timeout 300 python bench.py --precision fp8 --steps 40 --no-cpu-baseline --no-legs > $O/r06_bench_fp8_final.json 2>/dev/null; echo "fp8 rc=$?"
timeout 300 python bench.py --precision fp32 --steps 6 --warmup 1 --no-cpu-baseline --no-legs > $O/r06_bench_fp32_final.json 2>/dev/null; echo "fp32 rc=$?"
timeout 300 python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline --no-legs > $O/r06_bench_fp32x3_final.json 2>/dev/null; echo "fp32x3 rc=$?"
KEDS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-legs > $O/r06_bench_dist1.json 2>/dev/null; echo "dist1 rc=$?"
KEDS_BENCH_SHARED_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/r06_bench_gpus2_shared_gpu.json 2>/dev/null; echo "shared-gpu 2 ranks rc=$?"
KEDS_BENCH_INJECT_FAULT=2 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs > $O/r06_bench_injected_fault.json 2>/dev/null; echo "injected fault (a corrupted result list) rc=$? (3 expected)"
timeout 300 python tools/bench_train.py > $O/r06_bench_train.json 2>/dev/null; echo "train rc=$?"
timeout 900 bash tools/profile_round.sh r06_final > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -4 $O/profile_round.log
cp gpurun_out/r06_final_* $O/ 2>/dev/null
timeout 600 bash tools/pmc_l2.sh r06 > $O/pmc_l2.log 2>&1; echo "pmc l2 rc=$?"; cp gpurun_out/r06_pmc_l2.json $O/ 2>/dev/null
timeout 300 bash tools/kstats_cmd.sh bench.py --workload dual --steps 8 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" > $O/r06_dual_kstats.txt
timeout 300 bash tools/kstats_cmd.sh bench.py --precision fp8 --steps 8 --warmup 2 --no-cpu-baseline --no-verify --no-legs 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" > $O/r06_fp8_kstats.txt
timeout 300 bash tools/kstats_cmd.sh bench.py --precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-legs 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" > $O/r06_x3_kstats.txt
{ for C in one shard; do CONFIG=$C NO_AB=1 timeout 200 python tools/search_profile.py 2>&1 | grep -v amdgpu.ids; done; } > $O/r06_search_chain.txt 2>&1
ls $O
