#!/bin/bash
# copy the files of the last tools/rounds/r05_final.sh call from gpurun_out/r05 into profiles/ and refresh the parity record (run HERE, repo root)
O=gpurun_out/r05
for f in r05_bench_final r05_bench_dual_final r05_bench_fp8_final r05_bench_fp8_2m_final r05_bench_fp32_final r05_bench_fp32x3_final r05_bench_fp32x3_dual_final r05_bench_dist1 r05_bench_gpus2_shared_gpu r05_bench_injected_fault r05_bench_injected_fault_encoder r05_bench_train; do cp $O/$f.json profiles/$f.json; done
cp $O/r05_final_bench.json profiles/r05_bench_final_under_rocprof.json
cp $O/r05_final_kernel_stats.csv profiles/r05_bench_kernel_stats_final.csv
cp $O/r05_final_pmc_traffic.json profiles/r05_pmc_traffic.json
cp $O/r05_search_chain.txt profiles/r05_search_chain.txt
cp $O/r05_gemm_vs_vendor.txt profiles/r05_gemm_vs_vendor.txt
cp $O/r05_x3_kstats_final.txt profiles/r05_x3_kstats_final.txt
cp $O/r05_x3_attention_alone.txt profiles/r05_x3_attention_alone.txt
cp $O/pytest_gpu.log profiles/r05_pytest_gpu.log
ROUND=r05 python tools/update_parity_baseline.py 2>&1 | tail -1
python - <<'PY'
import json
from keds_amd import _lib
print("sources", _lib.source_digest(), "| pmc", json.load(open('profiles/r05_pmc_traffic.json')).get('csrc_sha16'), "| parity", json.load(open('profiles/r05_parity.json')).get('csrc_sha16'))
for f in ['r05_bench_final','r05_bench_dual_final','r05_bench_fp8_final','r05_bench_fp8_2m_final','r05_bench_fp32_final','r05_bench_fp32x3_final','r05_bench_fp32x3_dual_final','r05_bench_dist1','r05_bench_gpus2_shared_gpu','r05_bench_injected_fault','r05_bench_train']:
    try:
        d=json.loads(open('profiles/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), 'ms/step', round(d.get('ms_per_step',0),3), 'frac', d.get('roofline',{}).get('frac'), 'verification', (d.get('verification') or {}).get('ok'))
    except Exception as e:
        print(f, 'UNREADABLE', e)
PY
head -3 profiles/r05_search_chain.txt
