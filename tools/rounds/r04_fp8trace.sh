#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/fp8tr; timeout 900 rocprofv3 --kernel-trace -d /tmp/fp8tr -o fp8 --output-format csv -- python3 bench.py --precision fp8 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/fp8tr -name "*kernel_trace.csv" | head -1)
python3 tools/trace_block.py "$f" | tee $O/fp8_trace_block.txt
