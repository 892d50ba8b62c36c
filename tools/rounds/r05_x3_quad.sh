#!/bin/bash
# Round 5: the split-operand GEMMs on the 4-wave kernel (AGPR accumulators, persistent walk) against the 8-wave kernel
# (KEDS_X3_QUAD=0): parity tests, then the bench line, interleaved, same box.
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_fp32.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in 1 0; do echo -n "KEDS_X3_QUAD=$v "; KEDS_X3_QUAD=$v python bench.py --precision fp32x3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py; done
done
