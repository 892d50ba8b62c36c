#!/bin/bash
# L2 hit rate and memory-side read requests of the GEMM / scan / attention kernels of two bench-identical steps (tools/pmc_step.py):
# what stands behind the 2.1x read ratio of the 256 x 256 GEMM class (FETCH_SIZE counts what leaves L2, Infinity-Cache hits included).
# Counter passes on their own (no trace domains beside --pmc).  Run on the GPU box from the repo root: bash tools/pmc_l2.sh <tag>
tag=${1:-r06}
export TMPDIR=/tmp
W=/tmp/keds_pmc_l2_$tag; rm -rf $W; mkdir -p $W gpurun_out
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $W/a -o p -- python3 tools/pmc_step.py > $W/a.log 2>&1; echo "pmc hit/miss rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $W/b -o p -- python3 tools/pmc_step.py > $W/b.log 2>&1; echo "pmc rdreq rc=$?"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $W/c -o p -- python3 tools/pmc_step.py > $W/c.log 2>&1; echo "pmc wrreq rc=$?"
python3 - "$W" "gpurun_out/${tag}_pmc_l2.json" <<'PY'
import collections, csv, glob, json, sys, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
keys = ("gemm_bt_pair_kernel", "gemm_bt_quad_kernel", "gemm_bt_quad3_kernel", "gemm_bt_kernel", "scan_topk_kernel<768, 16", "attention_s257_kernel")
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in out.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if e.get("TCC_EA0_RDREQ_sum"):
        e["read_requests_to_local_dram_space_frac"] = e.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / e["TCC_EA0_RDREQ_sum"]
    res[k] = e
sys.path.insert(0, os.getcwd())
from keds_amd import _lib
res["csrc_sha16"] = _lib.source_digest()
res["what"] = ("per-launch means over two bench-identical steps; TCC_HIT / TCC_MISS: L2 (all eight XCDs); TCC_EA0_RDREQ / WRREQ: requests L2 sends to the "
               "memory side (Infinity Cache in front of HBM: its hits are not separable here), _DRAM: of those, to this device's own memory")
json.dump(res, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
