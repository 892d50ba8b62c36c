#!/bin/bash
# Round 5: the remainder-row chain beside the ATTENTION launch (KEDS_TAIL_ATTN=1: tail samples' attention on the side lane, the
# chain's GEMMs in their 64 KiB-LDS form so that they fit next to a resident attention workgroup) against the default placement
# (beside the main GEMMs), same box, alternating.
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-verify"
for i in 1 2 3; do
  for v in 0 1; do echo -n "KEDS_TAIL_ATTN=$v "; KEDS_TAIL_ATTN=$v $B 2>/dev/null | tail -1 | python3 tools/ab_line.py; done
done
echo -n "KEDS_SIDE_STREAM=0 (no lane) "; KEDS_SIDE_STREAM=0 $B 2>/dev/null | tail -1 | python3 tools/ab_line.py
