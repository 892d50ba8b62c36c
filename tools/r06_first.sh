#!/bin/bash
# round 6, first call: the driver line with the new legs (safe / fp8 / dual / measured recall parity) + the fp8 line as the
# baseline of this round's fp8 GEMM work
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_legs_v1.json 2> $O/bench_legs_v1.err; echo "bench rc=$?"
tail -c 6000 $O/bench_legs_v1.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'])
for k in ('safe_point','fp8_point','dual_point','fp32_point','fp32x3_point'):
    p=d.get(k); print(k, p and {a:p[a] for a in p if a in ('value','ms_per_step','ok')}, (p or {}).get('roofline',{}).get('frac'))
print('recall', json.dumps(d.get('recall_parity_measured'))[:1500])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['warmup_s_per_image'], d['cpu_baseline']['runs_s_per_image'], d['cpu_baseline']['spread_over_median'])
print('dual ver', d['dual_point']['verification'])
"
tail -5 $O/bench_legs_v1.err
timeout 300 python bench.py --precision fp8 --steps 40 --no-cpu-baseline --no-legs > $O/bench_fp8_base.json 2>/dev/null; echo "fp8 rc=$?"
python3 tools/ab_line.py < $O/bench_fp8_base.json
