#!/usr/bin/env python3
"""In-kernel timeline of the 4-wave 256^2 GEMM (gemm_bt_quad_kernel) and timing-only ablations of its K-loop (stamped
diagnostic builds): which of DMA issue / fragment reads / wait + barrier the 2.4 k cycles per K-tile go to.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
from keds_amd._lib import ptr, check, stream
lib = _lib.load()
SHAPE = os.environ.get("SHAPE", "qkv")
M, N, K = {"qkv": (32768, 3072, 1024), "proj": (32768, 1024, 4096)}[SHAPE]
RESID = SHAPE != "qkv"
if RESID:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.randn(M, N, device="cuda").half()
    stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64); EPI = _lib.EPI_RESID_STATS_F16
else:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(2 * N, device="cuda"); out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64); stats[:, 1] = (1 << 28) * K; EPI = _lib.EPI_LN_BIAS_BF16_H
tiles = (M // 256) * (N // 256)
buf = torch.zeros(tiles * 8 * 8 + tiles * 8, device="cuda", dtype=torch.int64)
_lib.ensure_gemm_workspace(torch.device("cuda"))
def gemm(aux2):
    if RESID:
        out.normal_()
    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, EPI, ptr(stats), 0, ptr(aux2), stream()), "gemm")
names = {1: "product loop", 2: "no DMA pieces", 3: "no fragment reads", 4: "no DMA, no reads", 5: "no wait + barrier",
         6: "no DMA, no wait + barrier", 7: "MFMAs only"}
for variant in (1, 2, 3, 4, 5, 6, 7, 1):
    lib.keds_gemm_force_small((variant << 13) | (1 << 11))
    for _ in range(20):
        gemm(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gemm(buf); e1.record(); torch.cuda.synchronize()
    lib.keds_gemm_force_small(0)
    t = buf[:tiles * 64].view(tiles, 8, 8).double().cpu()[:, :4]
    print(f"{names[variant]:28s} launch {e0.elapsed_time(e1) * 1e3:7.1f} us   prologue {t[:, :, 0].mean():7.0f}  K-loop {t[:, :, 1].mean():8.0f} "
          f"= {t[:, :, 1].mean() / (K // 64):6.0f} per K-tile   epilogue {t[:, :, 2].mean():7.0f}  drain {t[:, :, 5].mean():6.0f}", flush=True)
