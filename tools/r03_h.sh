#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for C in one shard; do CONFIG=$C NO_AB=1 python tools/search_profile.py 2>&1 | grep "us per search"; done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import keds_amd
from keds_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2002)
db = torch.nn.functional.normalize(torch.randn(500000, 768, generator=gen, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn(128, 768, generator=gen, device=dev), dim=1)
idx = keds_amd.FlatIndex(768, "l2", device=dev); idx.add(db)
def run(iters=30):
    for _ in range(3): idx.search_device(q, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): idx.search_device(q, 10)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rnd in range(3):
    for code, name in ((1 << 7, "depth 1"), (4 << 7, "depth 4")):
        lib.keds_scan_debug(code)
        print(f"threshold pass {name}: {run():7.1f} us per search", flush=True)
lib.keds_scan_debug(0)
PY
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.loads(open('$O/bench.json').read().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_scan']['whole_search'],d['numerics_guard'])"
python bench.py --workload dual --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dual.json 2> $O/bench_dual.err; python -c "import json;d=json.loads(open('$O/bench_dual.json').read().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
