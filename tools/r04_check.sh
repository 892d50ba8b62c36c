#!/bin/bash
# round-4 check run (GPU box, repo root): the full parity suite + the fp32 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
rm -f gpurun_out/gpu_test_metrics.jsonl
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python bench.py --precision fp32 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"; cut -c1-300 $O/bench_fp32.json
