#!/bin/bash
# the fp32 operating point after a change of csrc/f32path.hip (GPU box, repo root): its tests, then its bench line with the stage split
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp32.py -x -q > $O/pytest_fp32.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_fp32.log
timeout 600 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --prof-all --prof-every 1 2>/dev/null | tail -1 > $O/bench_fp32.json
python3 -c "
import json
d=json.load(open('$O/bench_fp32.json')); print('%.1f img/s  %.1f ms/step' % (d['value'], d['ms_per_step']), d['stage_ms_per_step'], d['roofline'])"
