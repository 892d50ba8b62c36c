#!/bin/bash
# Round 5: the per-launch event pairs as the launches' own start / stop events (KEDS_LAUNCH) and the towers' fork as the attention
# launch's stop event, against no events at all (KEDS_BENCH_NO_EVENTS=1: the floor) and the recorded fork (KEDS_FORK_EXT=0).
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_search.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
  echo -n "product                 "; python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 tools/ab_line.py
  echo -n "KEDS_BENCH_NO_EVENTS=1  "; KEDS_BENCH_NO_EVENTS=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 tools/ab_line.py
  echo -n "KEDS_FORK_EXT=0         "; KEDS_FORK_EXT=0 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 tools/ab_line.py
done
echo -n "fp8                     "; python bench.py --precision fp8 --steps 40 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 tools/ab_line.py
echo -n "fp8 NO_EVENTS           "; KEDS_BENCH_NO_EVENTS=1 python bench.py --precision fp8 --steps 40 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 tools/ab_line.py
