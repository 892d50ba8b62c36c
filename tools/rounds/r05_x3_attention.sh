#!/bin/bash
# Round 5: the split-operand attention and the remainder-row lane of the fp32x3 mode -- parity tests, the bench line (twice, and
# with the lane off), per-kernel times of the step.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_fp32.py -x -q 2>&1 | tail -5
for i in 1 2; do
python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_fp32x3_v3.json
python3 tools/ab_line.py < $O/r05_bench_fp32x3_v3.json
echo -n "KEDS_SIDE_STREAM=0 "; KEDS_SIDE_STREAM=0 python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py
done
bash tools/kstats_cmd.sh bench.py --precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" | tee $O/r05_x3_kstats_v3.txt
