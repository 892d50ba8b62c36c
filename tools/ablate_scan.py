#!/usr/bin/env python3
"""Timing of the search pipeline stages and timing-only ablations of the scan kernel (0.5 M x 768, B=128)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keds_amd
from keds_amd import _lib
lib = _lib.load()
N, D, B = int(os.environ.get("N", "500000")), 768, int(os.environ.get("B", "128"))
db = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=1)
idx = keds_amd.FlatIndex(D)
idx.add(db)
q = torch.nn.functional.normalize(torch.randn(B, D, device="cuda"), dim=1)
for code, name in [(0, "product"), (1, "no list update")]:
    lib.keds_scan_debug(code)
    for _ in range(3):
        idx.search_device(q, 10)
    _lib.prof_reset(); _lib.prof_enable(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx.search_device(q, 10)
    e1.record(); torch.cuda.synchronize()
    _lib.prof_enable(False)
    sms, sn = _lib.prof_read(_lib.PROF_SCAN)
    oms, on = _lib.prof_read(_lib.PROF_OTHER)
    print(f"{name:24s} search total {e0.elapsed_time(e1)/20*1e3:7.1f} us | scan launches {sn/20:.0f} x {sms/sn*1e3:7.1f} us | merge+rerank+select {oms/20*1e3:6.1f} us", flush=True)
lib.keds_scan_debug(0)
