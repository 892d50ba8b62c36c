#!/usr/bin/env python3
"""Whole-search timing (hipEvents around FlatIndex.search_device) for the two shapes that matter:
CONFIG=one   : 128 queries x 0.5 M x 768 (the 1-GPU headline search)
CONFIG=shard : 1024 queries x 62.5 k x 768 (one rank of the 8-GPU form: every rank's queries against its shard)
Interleaves the default and the non-temporal key stream (keds_scan_debug bit 6).  Under
`rocprofv3 --kernel-trace --stats` the per-kernel split comes out of the same run.  GPU only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keds_amd
from keds_amd import _lib
lib = _lib.load()
cfg = os.environ.get("CONFIG", "one")
nq, n = (128, 500000) if cfg == "one" else (1024, 62500)
k = int(os.environ.get("K", "10"))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2002)
db = torch.nn.functional.normalize(torch.randn(n, 768, generator=gen, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn(nq, 768, generator=gen, device=dev), dim=1)
idx = keds_amd.FlatIndex(768, "l2", device=dev)
idx.add(db)
def run(iters):
    for _ in range(3):
        idx.search_device(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        idx.search_device(q, k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = [(0, "nt key stream")] if os.environ.get("NO_AB") else [(0, "nt key stream"), (64, "default policy"), (0, "nt key stream"), (64, "default policy")]
for code, name in modes:
    lib.keds_scan_debug(code)
    us = run(int(os.environ.get("ITERS", "30")))
    print(f"{cfg}: {nq} q x {n} rows, k={k}, {name:14s}: {us:7.1f} us per search = {n * 768 * 2 / us / 1e6:6.2f} TB/s of one bf16 pass "
          f"({nq * n * 768 * 2 / us / 1e6:7.1f} TFLOP/s)", flush=True)
lib.keds_scan_debug(0)
print("certificate counts (certified, exact pass):", idx.certificate_counts())
