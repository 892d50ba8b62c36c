#!/bin/bash
# kernel stats of the dual-stream bench command (GPU box, repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/dprof; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/dprof -o d --output-format csv -- python3 bench.py --workload dual --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dual_under_rocprof.json 2>/dev/null
f=$(find /tmp/dprof -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_dual_kernel_stats.csv
python3 - <<'P'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r04/bench_dual_kernel_stats.csv')))
for r in rows[:22]:
    n=re.sub(r"\(anonymous namespace\)::","",r['Name'])
    print('%-90s calls %5s avg %9.1f us  %5.2f%%' % (n[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
P
