#!/usr/bin/env python3
"""Epilogue / remainder-row cost of the GEMM path per ViT-L/14 shape (timing only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()
E = {"bf16": _lib.EPI_BIAS_BF16, "qgelu": _lib.EPI_BIAS_QGELU_BF16, "resid": _lib.EPI_BIAS_RESID_F32, "f32": _lib.EPI_BIAS_F32}
def run(M, N, K, epi, code, iters=20):
    a = torch.randn((M + 255) // 256 * 256, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    f32 = epi in ("resid", "f32")
    out = torch.zeros(a.shape[0], N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    lib.keds_gemm_force_small(code)
    best = 1e9
    for rnd in range(3):
        for _ in range(3):
            ops.gemm_bt(a, w, bias, E[epi], out=out, m=M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm_bt(a, w, bias, E[epi], out=out, m=M)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    lib.keds_gemm_force_small(0)
    return best
for (M, N, K, name) in [(32896, 3072, 1024, "qkv"), (32896, 1024, 1024, "out"), (32896, 4096, 1024, "fc"), (32896, 1024, 4096, "proj")]:
    print(f"--- {name}: M={M} N={N} K={K}")
    for epi in ("bf16", "qgelu", "resid", "f32"):
        full = run(M, N, K, epi, 0x000)
        notail = run(M, N, K, epi, 0x100)
        tail_alone = run(128, N, K, epi, 0x000)
        print(f"   {epi:6s} product {full:7.1f} us  ({2.0*M*N*K/full/1e6:6.0f} TF)   main only {notail:7.1f} us   "
              f"remainder kernel alone (M=128) {tail_alone:6.1f} us", flush=True)
