#!/bin/bash
# quick correctness pass after a kernel change (GPU box, repo root): the tower / kernel / search tests, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_search.py tests/test_gpu_search_cert.py -x -q > $O/pytest_quick.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_quick.log | tail -5
timeout 300 python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f img/s  %.3f ms/step  gemm %.3f ms  frac %.4f  guard %s  whole search %s' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['gemm'], d['roofline']['frac'], d['numerics_guard']['tripped'], d['roofline_scan']['whole_search']))"
