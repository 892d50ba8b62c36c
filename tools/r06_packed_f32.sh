#!/bin/bash
# round 6: packed rows in the fp32 / fp32x3 text towers -- parity, then the fp32x3 dual line with and without (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fp32.py -x -q -k "packed or dual or text" 2>&1 | tail -5
for i in 1 2; do
  echo "fp32x3 dual, packed rows"; timeout 300 python bench.py --workload dual --precision fp32x3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py
  echo "fp32x3 dual, rectangular"; KEDS_BENCH_TEXT_RECT=1 timeout 300 python bench.py --workload dual --precision fp32x3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py
done
} 2>&1 | tee $O/text_packed_fp32x3_ab.txt
