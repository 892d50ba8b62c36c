#!/bin/bash
# round 6: the full GPU suite + the default bench line on the current build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
rm -f gpurun_out/gpu_test_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
cp gpurun_out/gpu_test_metrics.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_check.json 2> $O/bench_check.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_check.json').read().strip().splitlines()[-1])
print('value',round(d['value']),'frac',round(d['roofline']['frac'],4),'verification',d['verification']['ok'])
for k in ('safe_point','fp8_point','dual_point','fp32_point','fp32x3_point'):
    p=d.get(k) or {}; print(k, round(p.get('value',0)), p.get('ok'), (p.get('roofline') or {}).get('frac'))
r=d['recall_parity_measured']; print('recall ok',r['ok'], {k:(v['outcomes_flipped_of_1280'],v['query_features_vs_reference']['rel_l2']) for k,v in r['points'].items()})
print('x3 vs fp32', d['fp32x3_point']['vs_fp32_embeddings'])
PY
