#!/bin/bash
# copy the files of the last tools/rounds/r04_final.sh call from gpurun_out/r04 into profiles/ and refresh the parity record (run HERE, repo root)
O=gpurun_out/r04
for f in r04_bench_final r04_bench_dual_final r04_bench_fp8_final r04_bench_fp8_2m_final r04_bench_fp32_final r04_bench_dist1 r04_bench_gpus2_shared_gpu r04_bench_train; do cp $O/$f.json profiles/$f.json; done
cp $O/r04_final_bench.json profiles/r04_bench_final_under_rocprof.json
cp $O/r04_final_kernel_stats.csv profiles/r04_bench_kernel_stats_final.csv
cp $O/r04_final_pmc_traffic.json profiles/r04_pmc_traffic.json
cp $O/r04_search_chain.txt profiles/r04_search_chain.txt
cp $O/r04_gemm_hot_cold_final.txt profiles/r04_gemm_hot_cold.txt
cp $O/r04_gemm_kernels_ab.txt profiles/r04_gemm_kernels_ab.txt
ROUND=r04 python tools/update_parity_baseline.py 2>&1 | tail -1
python - <<'PY'
import json
from keds_amd import _lib
print("sources", _lib.source_digest(), "| pmc", json.load(open('profiles/r04_pmc_traffic.json')).get('csrc_sha16'), "| parity", json.load(open('profiles/r04_parity.json')).get('csrc_sha16'))
for f in ['r04_bench_final','r04_bench_dual_final','r04_bench_fp8_final','r04_bench_fp8_2m_final','r04_bench_fp32_final','r04_bench_dist1','r04_bench_gpus2_shared_gpu','r04_bench_train']:
    d=json.loads(open('profiles/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value'],1), 'ms/step', round(d.get('ms_per_step',0),3), 'frac', d.get('roofline',{}).get('frac'))
PY
head -3 profiles/r04_search_chain.txt
