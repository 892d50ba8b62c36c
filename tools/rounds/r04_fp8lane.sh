#!/bin/bash
# fp8 bench with the side lane on / off: in-step durations of the main kernels from a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
for lane in 1 0; do
export KEDS_SIDE_STREAM=$lane
timeout 600 python bench.py --precision fp8 --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lane $lane: %.1f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']), d['stage_ms_per_step'])"
rm -rf /tmp/fp8tr; timeout 900 rocprofv3 --kernel-trace -d /tmp/fp8tr -o fp8 --output-format csv -- python3 bench.py --precision fp8 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/fp8tr -name "*kernel_trace.csv" | head -1)
python3 tools/trace_block.py "$f" | grep -A12 "mean duration" | tee $O/fp8_trace_lane$lane.txt
done
