export TMPDIR=/tmp; W=/tmp/kpd; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o b -- python3 tools/bench_dual.py > $W/bench.log 2>&1
f=$(find $W/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for i, r in enumerate(rows):
    if i >= 22: break
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][-56:]
    print(f'{name:58s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {r["Percentage"]}%')
PY
tail -1 $W/bench.log | cut -c1-200
