#!/bin/bash
# kernel-time table of any python script (rocprofv3 --kernel-trace --stats): tools/kstats_cmd.sh script.py  [env via export]
export TMPDIR=/tmp; W=/tmp/kp_$$; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o run -- python3 "$@" > $W/run.log 2>&1
tail -5 $W/run.log
f=$(find $W/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 16: break
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = name.split("(")[0][-60:]
    print(f'{name:62s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}  {r["Percentage"]}%')
PY
