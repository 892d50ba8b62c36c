#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel -> JSON.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads,
so hbm_read_bytes = 2 * FETCH_SIZE * 1024 (the counter is in KiB); WRITE_SIZE * 1024 is exact."""
import collections, csv, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("gemm_bt_pair_kernel", "gemm_bt_quad_kernel", "gemm_bt_quad3_kernel", "gemm_bt_kernel", "scan_topk_kernel<768, 16", "scan_topk_kernel<768, 4", "attention_s257_kernel", "attention_tail1_kernel", "attention_kernel", "layernorm_kernel", "merge_pairs_kernel"):
            if key in name:
                out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in out.items():
    e = {"launches": max(len(v) for v in d.values())}
    if "FETCH_SIZE" in d:
        e["FETCH_SIZE_KiB_mean"] = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        e["hbm_read_bytes_per_launch"] = 2 * 1024 * e["FETCH_SIZE_KiB_mean"]
    if "WRITE_SIZE" in d:
        e["WRITE_SIZE_KiB_mean"] = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        e["hbm_write_bytes_per_launch"] = 1024 * e["WRITE_SIZE_KiB_mean"]
    res[k] = e
# the 256 x 256 GEMM class as one row (8-wave, 4-wave and 4-wave three-deep-ring kernels: launch-weighted mean)
big = [res[k] for k in ("gemm_bt_pair_kernel", "gemm_bt_quad_kernel", "gemm_bt_quad3_kernel") if k in res]
if big:
    n = sum(e["launches"] for e in big)
    res["gemm_256x256_all"] = {"launches": n}
    for f in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch"):
        if all(f in e for e in big):
            res["gemm_256x256_all"][f] = sum(e[f] * e["launches"] for e in big) / n
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib      # noqa: E402  (the digest of the kernel sources these passes ran on; bench.py checks it)
res["csrc_sha16"] = _lib.source_digest()
json.dump(res, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
