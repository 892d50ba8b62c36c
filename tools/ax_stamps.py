#!/usr/bin/env python3
"""Phase stamps of attention_x3_kernel (build with EXTRA=-DKEDS_AX_DBG=16): wave 0 of the first 256 workgroups.  GPU only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
lib = _lib.load()
B, S, H = 128, 257, 16
d = 64 * H
qkv = torch.randn(B * S, 3 * d, device="cuda")
out = torch.zeros(B * S, d, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
pair = torch.zeros(2, B * S * d, dtype=torch.float16, device="cuda")   # the real output; `out` only takes the stamps in this build
for _ in range(3):
    _lib.check(lib.keds_attention_x3(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(pair), B * S * d, B, S, H, 0, 0, _lib.ptr(flag), _lib.stream()), "x3")
torch.cuda.synchronize()
t = out.reshape(-1).view(torch.int64)[:256 * 8].reshape(256, 8).cpu().numpy().astype(np.float64)
names = ["staging (loads, split, LDS writes)", "barrier", "own q load + split", "own tile: 9 key tiles", "own output", "lone q + key tile(s)", "lone barrier + combine"]
dt = np.diff(t, axis=1) / 100.0          # s_memtime: 100 MHz -> us
for i, n in enumerate(names):
    print(f"{n:40s} median {np.median(dt[:, i]):7.2f} us   min {dt[:, i].min():7.2f}  max {dt[:, i].max():7.2f}")
print(f"{'workgroup lifetime (wave 0)':40s} median {np.median(t[:, 7] - t[:, 0]) / 100.0:7.2f} us")
