#!/bin/bash
# round-5 evidence run (GPU box, repo root): full parity suite, the bench lines of every workload / operating point, rocprofv3
# kernel stats of the bench command + the two PMC passes, search chain, GEMM kernels against the vendor's.  Everything lands
# under gpurun_out/r05/.  Every stage runs under its own timeout: a wedged stage must not take the rest of the budget with it.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05; mkdir -p $O
rm -f gpurun_out/gpu_test_metrics.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 400 python bench.py > $O/r05_bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --workload dual --steps 40 --no-cpu-baseline > $O/r05_bench_dual_final.json 2>/dev/null; echo "dual rc=$?"
timeout 300 python bench.py --precision fp8 --db-rows 2000000 --steps 40 --no-cpu-baseline > $O/r05_bench_fp8_2m_final.json 2>/dev/null; echo "fp8 2M rc=$?"
timeout 300 python bench.py --precision fp8 --steps 40 --no-cpu-baseline > $O/r05_bench_fp8_final.json 2>/dev/null; echo "fp8 rc=$?"
timeout 300 python bench.py --precision fp32 --steps 6 --warmup 1 --no-cpu-baseline > $O/r05_bench_fp32_final.json 2>/dev/null; echo "fp32 rc=$?"
timeout 300 python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline > $O/r05_bench_fp32x3_final.json 2>/dev/null; echo "fp32x3 rc=$?"
timeout 300 python bench.py --precision fp32x3 --workload dual --steps 10 --warmup 2 --no-cpu-baseline > $O/r05_bench_fp32x3_dual_final.json 2>/dev/null; echo "fp32x3 dual rc=$?"
KEDS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 60 --no-cpu-baseline > $O/r05_bench_dist1.json 2>/dev/null; echo "dist1 rc=$?"
KEDS_BENCH_SHARED_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/r05_bench_gpus2_shared_gpu.json 2>/dev/null; echo "shared-gpu 2 ranks rc=$?"
KEDS_BENCH_INJECT_FAULT=2 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r05_bench_injected_fault.json 2>/dev/null; echo "injected fault (a corrupted result list) rc=$? (3 expected)"
KEDS_BENCH_INJECT_FAULT=1 timeout 400 python bench.py --steps 5 --warmup 1 > $O/r05_bench_injected_fault_encoder.json 2>/dev/null; echo "injected fault (a corrupted encoder; needs the oracle's embeddings, i.e. the cpu_baseline leg) rc=$? (3 expected)"
timeout 300 python tools/bench_train.py > $O/r05_bench_train.json 2>/dev/null; echo "train rc=$?"
timeout 900 bash tools/profile_round.sh r05_final > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -4 $O/profile_round.log
cp gpurun_out/r05_final_* $O/ 2>/dev/null
{ for C in one shard; do CONFIG=$C NO_AB=1 timeout 200 python tools/search_profile.py 2>&1 | grep -v amdgpu.ids; done; CONFIG=one NO_AB=1 ITERS=10 timeout 300 bash tools/kstats_cmd.sh tools/search_profile.py 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026"; } > $O/r05_search_chain.txt 2>&1
{ timeout 300 python tools/micro/vendor_gemm.py 2>&1 | grep -v amdgpu.ids; FORMS="4 waves, persistent" timeout 300 python tools/ab_quad.py 2>&1 | grep -v "^$\|amdgpu.ids"; } > $O/r05_gemm_vs_vendor.txt 2>&1
timeout 300 bash tools/kstats_cmd.sh bench.py --precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" > $O/r05_x3_kstats_final.txt
timeout 200 python tools/ab_attn_x3.py 2>&1 | grep -v amdgpu.ids > $O/r05_x3_attention_alone.txt
ls $O
