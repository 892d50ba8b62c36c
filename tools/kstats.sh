#!/bin/bash
# kernel-time table of the bench command (rocprofv3 --kernel-trace --stats), top rows
export TMPDIR=/tmp; W=/tmp/kp; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $W/bench.log 2>&1
f=$(find $W/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 14: break
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = name.split("(")[0][-60:]
    print(f'{name:62s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}  {r["Percentage"]}%')
PY
