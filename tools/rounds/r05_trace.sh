#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; W=/tmp/kt; rm -rf $W; mkdir -p $W gpurun_out/r05
for lane in 1 0; do
  rm -rf $W/t$lane
  KEDS_SIDE_STREAM=$lane timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W/t$lane -o b -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify > $W/b$lane.log 2>&1
  f=$(find $W/t$lane -name "*kernel_trace.csv" | head -1)
  echo "### KEDS_SIDE_STREAM=$lane"; head -1 "$f" | cut -c1-300; python3 tools/trace_gaps.py "$f"
done > gpurun_out/r05/r05_trace_gaps.txt 2>&1
cat gpurun_out/r05/r05_trace_gaps.txt
