#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_exchange2.py -x -q 2>&1 | tail -2
KEDS_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/r04_bench_gpus2_shared_gpu.json 2> $O/shared.err; echo "rc=$?"; tail -1 $O/r04_bench_gpus2_shared_gpu.json | cut -c1-600; tail -3 $O/shared.err
KEDS_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --workload dual --steps 6 --warmup 2 > $O/r04_bench_dual_gpus2_shared_gpu.json 2> $O/shared2.err; echo "rc=$?"; tail -1 $O/r04_bench_dual_gpus2_shared_gpu.json | cut -c1-400; tail -3 $O/shared2.err
