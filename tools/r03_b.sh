#!/bin/bash
# round-3 GPU pass b: residual-as-accumulator A/B, kernel tests, benches, vendor kernel names
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q > $O/pytest_kernels.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_kernels.log
python tools/ab_resid_prologue.py > $O/ab_resid_prologue.txt 2>&1; cat $O/ab_resid_prologue.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_rp.json 2> $O/bench_rp.err; echo "bench rc=$?"; python -c "import json;d=json.loads(open('$O/bench_rp.json').read().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['stage_ms_per_step'])"
KEDS_BENCH_FORCE_DIST=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_dist1_auto.json 2> $O/bench_dist1_auto.err; echo "dist1 rc=$?"; python -c "import json;d=json.loads(open('$O/bench_dist1_auto.json').read().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['search_overlap'],d['search_overlap_pilot'],d['per_rank_search'])"
KEDS_BENCH_FORCE_DIST=1 python bench.py --workload dual --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dual_dist1.json 2> $O/bench_dual_dist1.err; echo "dual dist1 rc=$?"; python -c "import json;d=json.loads(open('$O/bench_dual_dist1.json').read().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/vendor -o vendor -- python3 $GRAFT_REPO_ROOT/tools/micro/vendor_gemm.py > $GRAFT_REPO_ROOT/$O/vendor_gemm.txt 2>&1
cd $GRAFT_REPO_ROOT; cat $O/vendor_gemm.txt | grep TF; find $O/vendor -name "*kernel_stats*" | head -3
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r03b/vendor/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:200], r['Calls'], r['AverageNs'])
PY
