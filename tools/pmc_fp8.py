#!/usr/bin/env python3
"""One MXFP8 GEMM shape (proj: 32768 x 1024 x 4096) a few times, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
lib = _lib.load()
M, N, K = 32768, 1024, 4096
def quant(x):
    rows, k = x.shape
    q = torch.zeros((rows, k), dtype=torch.uint8, device="cuda"); s = torch.full((k // 128, rows, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(x), 0, rows, k, rows, _lib.ptr(q), _lib.ptr(s), _lib.stream()), "q")
    return q, s
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
aq, as_ = quant(a); wq, ws = quant(w)
bias = torch.randn(N, device="cuda"); out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
for _ in range(6):
    _lib.check(lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bias), _lib.ptr(out), M, N, K, _lib.stream()), "g")
torch.cuda.synchronize()
