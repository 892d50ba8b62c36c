#!/bin/bash
# copy the files of the last tools/r06_final.sh call from gpurun_out/r06 into profiles/ and refresh the parity record (run HERE, repo root)
O=gpurun_out/r06
for f in r06_bench_final r06_bench_driver_form r06_bench_dual_final r06_bench_fp8_final r06_bench_fp8_2m_final r06_bench_fp32_final r06_bench_fp32x3_final r06_bench_dist1 r06_bench_gpus2_shared_gpu r06_bench_injected_fault r06_bench_train; do cp $O/$f.json profiles/$f.json; done
cp $O/r06_final_bench.json profiles/r06_bench_final_under_rocprof.json
cp $O/r06_final_kernel_stats.csv profiles/r06_bench_kernel_stats_final.csv
cp $O/r06_final_pmc_traffic.json profiles/r06_pmc_traffic.json
cp $O/r06_pmc_l2.json profiles/r06_pmc_l2.json
for f in r06_dual_kstats r06_fp8_kstats r06_x3_kstats r06_search_chain; do cp $O/$f.txt profiles/$f.txt; done
cp $O/pytest_gpu.log profiles/r06_pytest_gpu.log
cp $O/gpu_test_metrics.jsonl gpurun_out/gpu_test_metrics.jsonl
ROUND=r06 python tools/update_parity_baseline.py 2>&1 | tail -1
python - <<'PY'
import json
from keds_amd import _lib
print("sources", _lib.source_digest(), "| pmc", json.load(open('profiles/r06_pmc_traffic.json')).get('csrc_sha16'), "| l2", json.load(open('profiles/r06_pmc_l2.json')).get('csrc_sha16'), "| parity", json.load(open('profiles/r06_parity.json')).get('csrc_sha16'))
for f in ['r06_bench_final','r06_bench_driver_form','r06_bench_dual_final','r06_bench_fp8_final','r06_bench_fp8_2m_final','r06_bench_fp32_final','r06_bench_fp32x3_final','r06_bench_dist1','r06_bench_gpus2_shared_gpu','r06_bench_injected_fault','r06_bench_train']:
    try:
        d=json.loads(open('profiles/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), 'ms/step', round(d.get('ms_per_step',0),3), 'frac', d.get('roofline',{}).get('frac'), 'verification', (d.get('verification') or {}).get('ok'))
    except Exception as e:
        print(f, 'UNREADABLE', e)
d=json.loads(open('profiles/r06_bench_final.json').read().strip().splitlines()[-1])
for k in ('safe_point','fp8_point','dual_point','fp32_point','fp32x3_point'):
    p=d.get(k) or {}; print(' ', k, round(p.get('value',0)), p.get('ok'), (p.get('roofline') or {}).get('frac'))
c=d['cpu_baseline']; print('  cpu', round(c['value'],2), 'spread', c['spread_over_median'], c.get('spread_all_runs_over_median'), c['runs_s_per_image'])
print('  traffic', d['roofline']['traffic'], '| scan', d['roofline_scan']['traffic'], d['roofline_scan']['frac'], d['roofline_scan']['whole_search'])
r=d['recall_parity_measured']; print('  recall', r['ok'], {k:(v['outcomes_flipped_of_1280'], v['query_features_vs_reference']['rel_l2']) for k,v in r['points'].items()})
PY
