#!/bin/bash
# Round 5: the fp32x3 flow's LayerNorm -> planes pass as a stream (KEDS_LN_STREAM=1, default) against one row per wave (=0):
# parity tests, then the bench line, interleaved, same box.
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_fp32.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in 1 0; do echo -n "KEDS_LN_STREAM=$v "; KEDS_LN_STREAM=$v python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py; done
done
bash tools/kstats_cmd.sh bench.py --precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | grep -i "layernorm"
