#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -k "side_lane or recall or vitl14 or tiny" 2>&1 | tail -3
{ echo "### tail samples' attention on the side lane (KEDS_TAIL_ATTN=1, default) vs the join in front of every attention launch (=0)"
  for i in 1 2 3 4; do for k in 0 1; do echo -n "KEDS_TAIL_ATTN=$k "; KEDS_TAIL_ATTN=$k timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 tools/ab_line.py; done; done
  echo "--- dual workload"
  for i in 1 2; do for k in 0 1; do echo -n "KEDS_TAIL_ATTN=$k "; KEDS_TAIL_ATTN=$k timeout 200 python bench.py --workload dual --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-180; done; done
} > $O/r04_tail_attention_ab.txt 2>&1
cat $O/r04_tail_attention_ab.txt
