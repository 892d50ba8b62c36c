#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_fp8.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline"
for V in auto 0 auto 0; do if [ $V = auto ]; then unset KEDS_GEMM_QUAD; else export KEDS_GEMM_QUAD=$V; fi; echo "quad=$V $($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['stage_ms_per_step']['gemm'])")"; done
