#!/bin/bash
# the MXFP8 GEMM forms (GPU box, repo root): tests, per-shape timing 8-wave vs 4-wave, cold-operand matrix, the fp8 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q > $O/pytest_fp8.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fp8.log
timeout 600 python tools/fp8_forms.py 2>&1 | grep -v amdgpu.ids | tee $O/fp8_forms.txt
[ -n "$COLD" ] && timeout 600 python tools/fp8_cold_matrix.py 2>&1 | grep -v amdgpu.ids | tee $O/fp8_cold_matrix.txt
for q in 0 1; do
KEDS_FP8_QUAD=$q timeout 600 python bench.py --precision fp8 --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fp8_quad$q.json
python3 -c "
import json
d=json.load(open('$O/bench_fp8_quad$q.json')); print('KEDS_FP8_QUAD=$q: %.1f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']), d['stage_ms_per_step'], 'frac %.4f' % d['roofline']['frac'])"
done
