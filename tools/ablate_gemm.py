#!/usr/bin/env python3
"""Timing-only ablations of the 256x256 GEMM kernel (results of variants != 0 are wrong by design)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops  # noqa: E402

lib = _lib.load()
VARIANTS = [(0x000, "product (pair + tail)"), (0x100, "pair kernel only (no tail)"), (0x170, "ring kernel only (no tail)"),
            (0x001, "128^2 kernel")]
if os.environ.get("ABLATE_RING"):
    VARIANTS += [(0x110, "ring: no s_barrier"), (0x120, "ring: no LDS-DMA in loop"), (0x130, "ring: no fragment reads"),
                 (0x140, "ring: no MFMA"), (0x150, "ring: no MFMA, full-line DMA"), (0x160, "ring: MFMA + full-line DMA")]
for M, N, K in [(32896, 3072, 1024), (32896, 1024, 4096), (32896, 1024, 1024)]:
    a = torch.randn((M + 255) // 256 * 256, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(a.shape[0], N, device="cuda", dtype=torch.bfloat16)
    print(f"--- M={M} N={N} K={K}")
    for rnd in range(2):
        for code, name in VARIANTS:
            lib.keds_gemm_force_small(code)
            for _ in range(3):
                ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            if rnd == 1:
                print(f"  {name:24s} {us:8.1f} us   {2.0 * M * N * K / us / 1e6:7.1f} TF-equivalent", flush=True)
lib.keds_gemm_force_small(0)
