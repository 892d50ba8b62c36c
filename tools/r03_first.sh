#!/bin/bash
# round-3 first GPU pass: parity suite + the bench entry points the driver uses
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r03a/pytest.log
python bench.py --steps 40 --warmup 5 > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r03a/bench_default.json | head -c 600; echo
KEDS_BENCH_FORCE_DIST=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_dist1_auto.json 2> gpurun_out/r03a/bench_dist1_auto.err; echo "dist1 rc=$?"
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r03a/bench_gpus2.out 2> gpurun_out/r03a/bench_gpus2.err; echo "gpus2 rc=$?"
tail -5 gpurun_out/r03a/bench_gpus2.err
python bench.py --workload dual --steps 20 --warmup 3 > gpurun_out/r03a/bench_dual.json 2> gpurun_out/r03a/bench_dual.err; echo "dual rc=$?"
KEDS_BENCH_FORCE_DIST=1 python bench.py --workload dual --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03a/bench_dual_dist1.json 2> gpurun_out/r03a/bench_dual_dist1.err; echo "dual dist1 rc=$?"
