#!/bin/bash
# kernel stats of the fp8 bench command (GPU box, repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/fp8prof; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/fp8prof -o fp8 --output-format csv -- python3 bench.py --precision fp8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fp8_under_rocprof.json 2>/dev/null
f=$(find /tmp/fp8prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_fp8_kernel_stats.csv
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04/bench_fp8_kernel_stats.csv')))
for r in rows[:16]:
    print('%-110s calls %5s avg %9.1f us  %5.2f%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
P
