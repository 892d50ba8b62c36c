#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
{ echo "### search tail: fused (default) vs three launches (KEDS_SEARCH_FUSED=0/1; rounds 4's run of this script set KEDS_SEARCH_UNFUSED, which nothing reads: both arms were the unfused default), tools/search_profile.py"
  for i in 1 2 3; do for u in 0 1; do echo -n "FUSED=$u  "; KEDS_SEARCH_FUSED=$u CONFIG=one NO_AB=1 python tools/search_profile.py 2>&1 | grep "us per search"; done; done
  for u in 0 1; do echo -n "shard FUSED=$u  "; KEDS_SEARCH_FUSED=$u CONFIG=shard NO_AB=1 python tools/search_profile.py 2>&1 | grep "us per search"; done
  echo "### out-proj on the 4-wave kernel with the three-deep A ring (KEDS_RESID_QUAD_K=1024) vs the 8-wave kernel (2048)"
  for i in 1 2 3 4; do for k in 2048 1024; do echo -n "KEDS_RESID_QUAD_K=$k "; KEDS_RESID_QUAD_K=$k python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f img/s  gemm %.3f ms  whole search %.1f us' % (d['value'], d['stage_ms_per_step']['gemm'], d['roofline_scan']['whole_search']['ms']*1e3))"; done; done
} > $O/r04_ab3.txt 2>&1
cat $O/r04_ab3.txt
