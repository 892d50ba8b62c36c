#!/bin/bash
# round 6: the text tower on packed rows -- parity tests, then the dual line with and without (same box, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "packed or readout_row or text_tower_does" 2>&1 | tail -5
for i in 1 2; do
  echo "packed rows"; timeout 300 python bench.py --workload dual --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py
  echo "rectangular (KEDS_BENCH_TEXT_RECT=1)"; KEDS_BENCH_TEXT_RECT=1 timeout 300 python bench.py --workload dual --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py
done
timeout 300 bash tools/kstats_cmd.sh bench.py --workload dual --steps 8 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026"
} 2>&1 | tee $O/text_packed_ab.txt
