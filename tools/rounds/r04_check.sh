#!/bin/bash
# round-4 check run (GPU box, repo root): the full parity suite + kernel stats / PMC traffic of the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
rm -f gpurun_out/gpu_test_metrics.jsonl
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
bash tools/profile_round.sh r04_mid > $O/profile_round.log 2>&1; tail -6 $O/profile_round.log
python3 - <<'PY'
import csv, re
for i, r in enumerate(csv.DictReader(open("gpurun_out/r04_mid_kernel_stats.csv"))):
    if i >= 12: break
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][-60:]
    print(f'{name:62s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]}%')
PY
