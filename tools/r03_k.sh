#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_session.py tests/test_gpu_fp8.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline"
for V in 1 0 1 0; do export KEDS_TOWER_FILL=$V; echo "fill=$V $($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['stage_ms_per_step']['gemm'], d['side_lane_rows'])")"; done
unset KEDS_TOWER_FILL
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --prof-all --prof-every 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prof-all', d['ms_per_step'], d['stage_ms_per_step'])"
