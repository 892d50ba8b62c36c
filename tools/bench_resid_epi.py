#!/usr/bin/env python3
"""Cost of the residual epilogues on the ViT-L/14 out-proj / proj shapes: fp32 residual + bf16 copy (EPI 8, 10 B/element),
fp16 residual (EPI 9, 4 B/element), plain bf16 store (EPI 0, 2 B/element: the floor).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402


def main():
    lib = _lib.load()
    iters = int(os.environ.get("ITERS", "20"))
    M = 32768
    for N, K, tag in ((1024, 1024, "out"), (1024, 4096, "proj")):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        x32 = torch.zeros(M, N, device="cuda")
        x16 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        xb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
        _lib.ensure_gemm_workspace(a.device)
        cases = {"f32+bf16 copy (8)": (_lib.EPI_RESID_STATS_F32, x32, stats, xb),
                 "f16 (9)": (_lib.EPI_RESID_STATS_F16, x16, stats, None),
                 "bf16 store (0)": (_lib.EPI_BIAS_BF16, xb, None, None)}
        for rnd in range(2):
            for name, (epi, out, aux, aux2) in cases.items():
                def run():
                    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, epi, ptr(aux), 0, ptr(aux2),
                                               stream()), "gemm")
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / iters * 1e3
                print(f"{tag:5s} {name:20s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
