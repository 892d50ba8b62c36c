#!/usr/bin/env python3
"""MXFP8 GEMM vs the bf16 256^2 kernel on the ViT-L/14 shapes (M = 32768 rows: the full 256-row tiles)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()

def quant(x):
    rows, K = x.shape
    q = torch.zeros((rows, K), dtype=torch.uint8, device="cuda")
    s = torch.full((K // 128, rows, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(x), 0, rows, K, rows, _lib.ptr(q), _lib.ptr(s), _lib.stream()), "q")
    return q, s

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K, tag in [(32768, 3072, 1024, "qkv"), (32768, 4096, 1024, "fc"), (32768, 1024, 4096, "proj"), (32768, 1024, 1024, "out"), (8192, 8192, 8192, "square 8k")]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
    bias = torch.randn(N, device="cuda")
    aq, as_ = quant(a); wq, ws = quant(w)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
    t8 = timeit(lambda: _lib.check(lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bias), _lib.ptr(out), M, N, K, _lib.stream()), "g"))
    t16 = timeit(lambda: ops.gemm_bt(ab, wb, bias, _lib.EPI_BIAS_BF16, out=out, m=M))
    tq = timeit(lambda: quant(a))
    abl = []
    for code in (1, 2, 4):
        lib.keds_mxfp8_debug(code)
        abl.append(timeit(lambda: _lib.check(lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bias), _lib.ptr(out), M, N, K, _lib.stream()), "g")))
    lib.keds_mxfp8_debug(3)            # in-kernel clock: core ticks / 100 MHz ticks of the K-loop, after a sustained run
    for _ in range(200):
        lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bias), _lib.ptr(out), M, N, K, _lib.stream())
    torch.cuda.synchronize()
    stamps = out.view(torch.int64).reshape(M, N // 4)[::256, ::64][:, :, None]
    raw = out.view(torch.int64).reshape(M, N // 4)
    core = raw[::256, 0::64].flatten().float(); real = raw[::256, 1::64].flatten().float()
    ghz = (core / real * 0.1).median().item()
    lib.keds_mxfp8_debug(0)
    print(f"{tag:10s} mxfp8 {t8:7.1f} us {2.0*M*N*K/t8/1e6:7.1f} TF   bf16 {t16:7.1f} us {2.0*M*N*K/t16/1e6:7.1f} TF   speedup {t16/t8:4.2f}   (quantise A: {tq:6.1f} us)  [DMA+barriers only {abl[0]:6.1f} us, no MFMA {abl[1]:6.1f} us, MFMA only {abl[2]:6.1f} us, in-loop clock {ghz:4.2f} GHz]", flush=True)
