#!/usr/bin/env python3
"""Scan rate vs database size (fixed cost vs per-byte cost of scan_topk_kernel): 128 queries, k = 10, N = 0.25 M .. 2 M rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keds_amd
from keds_amd import _lib

dev = torch.device("cuda", 0)
D = 768
q = torch.nn.functional.normalize(torch.randn(128, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3)), dim=1)
for N in (250_000, 500_000, 1_000_000, 2_000_000):
    gen = torch.Generator(device=dev).manual_seed(2002)
    idx = keds_amd.FlatIndex(D, "l2", device=dev)
    for s in range(0, N, 250_000):
        idx.add(torch.nn.functional.normalize(torch.randn(250_000, D, generator=gen, device=dev), dim=1))
    for _ in range(3):
        idx.search_device(q, 10)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        idx.search_device(q, 10)
    e1.record()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms, n = _lib.prof_read(_lib.PROF_SCAN)
    other_ms, _ = _lib.prof_read(_lib.PROF_OTHER)
    print(f"N={N:8d}: scan launches {ms / reps * 1e3:7.1f} us/search ({n // reps} launches) -> {N * D * 2 / (ms / reps * 1e-3) / 1e12:5.2f} TB/s as charged; "
          f"whole search {e0.elapsed_time(e1) / reps * 1e3:7.1f} us (events on)", flush=True)
    del idx
