#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
python bench.py --steps 60 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python bench.py --precision fp8 --db-rows 2000000 --steps 30 --warmup 4 --no-cpu-baseline > $O/bench_fp8_2m.json 2> $O/bench_fp8_2m.err; echo "fp8 rc=$?"
python bench.py --workload dual --steps 30 --warmup 4 > $O/bench_dual.json 2> $O/bench_dual.err; echo "dual rc=$?"
KEDS_BENCH_FORCE_DIST=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
python tools/bench_train.py > $O/bench_train.txt 2>&1; echo "train rc=$?"; tail -2 $O/bench_train.txt
for f in bench_default bench_fp8_2m bench_dual bench_dist1; do python -c "
import json;d=json.loads(open('$O/$f.json').read().splitlines()[-1]);print('$f', round(d['value'],1), round(d['ms_per_step'],3), 'gemm frac', round(d['roofline']['frac'],4), 'scan', d['roofline_scan']['whole_search'], d.get('cpu_baseline',{}).get('value'))"; done
