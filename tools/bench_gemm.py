#!/usr/bin/env python3
"""GEMM microbench on the ViT-L/14 and text-tower shapes (random bf16 data; TFLOP/s per shape,
256x256 kernel vs 128x128 kernel interleaved in one process).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops  # noqa: E402

SHAPES = [  # (M, N, K, epilogue, tag)
    (32896, 3072, 1024, _lib.EPI_BIAS_BF16, "vit.qkv"),
    (32896, 1024, 1024, _lib.EPI_BIAS_RESID_F32, "vit.out"),
    (32896, 4096, 1024, _lib.EPI_BIAS_QGELU_BF16, "vit.fc"),
    (32896, 1024, 4096, _lib.EPI_BIAS_RESID_F32, "vit.proj"),
    (9856, 2304, 768, _lib.EPI_BIAS_BF16, "text.qkv"),
    (9856, 768, 768, _lib.EPI_BIAS_RESID_F32, "text.out"),
    (9856, 3072, 768, _lib.EPI_BIAS_QGELU_BF16, "text.fc"),
    (9856, 768, 3072, _lib.EPI_BIAS_RESID_F32, "text.proj"),
    (19712, 2304, 768, _lib.EPI_BIAS_BF16, "text2.qkv"),
    (19712, 768, 768, _lib.EPI_BIAS_RESID_F32, "text2.out"),
    (19712, 3072, 768, _lib.EPI_BIAS_QGELU_BF16, "text2.fc"),
    (19712, 768, 3072, _lib.EPI_BIAS_RESID_F32, "text2.proj"),
]


def main():
    lib = _lib.load()
    iters = int(os.environ.get("ITERS", "20"))
    for M, N, K, epi, tag in SHAPES:
        Mp = (M + 255) // 256 * 256
        a = (torch.randn(Mp, K, device="cuda")).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        f32 = epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_BIAS_F32)
        out = torch.zeros(Mp, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        res = {}
        for rnd in range(3):
            for small in (0, 1):
                lib.keds_gemm_force_small(small)
                for _ in range(3):
                    ops.gemm_bt(a, w, bias, epi, out=out, m=M)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.gemm_bt(a, w, bias, epi, out=out, m=M)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                res.setdefault(small, []).append(2.0 * M * N * K / ms / 1e9)
        lib.keds_gemm_force_small(0)
        print(f"{tag:10s} M={M} N={N} K={K}: big {max(res[0]):7.1f} TF (min {min(res[0]):7.1f})   "
              f"128^2 {max(res[1]):7.1f} TF   [{2.0 * M * N * K / max(res[0]) / 1e9 * 1e3:.0f} us]", flush=True)


if __name__ == "__main__":
    main()
