#!/usr/bin/env python3
"""Phases of one tile of the 4-wave MXFP8 GEMM, from a library built with -DKEDS_FQ_STAMP (tools/fp8_stamp.sh): the qkv shape
(LN epilogue), ticks of s_memtime per wave (the counter runs at the core clock on this part: a 19.5 us tile reads 38.7 k ticks), median
over workgroups."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402

lib = _lib.load()
M, N, K = 32768, int(os.environ.get("N", "3072")), 1024


def main():
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5 * 0.5
    b = torch.randn(N, device="cuda") * 0.02
    aq = torch.zeros((M, K), dtype=torch.uint8, device="cuda")
    as_ = torch.full((K // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(a), 0, M, K, M, _lib.ptr(aq), _lib.ptr(as_), _lib.stream()), "q")
    wq = torch.zeros((N, K), dtype=torch.uint8, device="cuda")
    ws = torch.full((K // 128, N, 4), 127, dtype=torch.uint8, device="cuda")
    bc = torch.zeros(2 * N, device="cuda")
    _lib.check(lib.keds_fold_layernorm_mxfp8(_lib.ptr(w), _lib.ptr(b), _lib.ptr(torch.ones(K, device="cuda")), _lib.ptr(torch.zeros(K, device="cuda")),
                                             N, K, N, _lib.ptr(wq), _lib.ptr(ws), _lib.ptr(bc), _lib.stream()), "fold")
    stats = (torch.stack([a.sum(1), (a * a).sum(1)], dim=1).double() * 2.0 ** 28).round().to(torch.int64).contiguous()
    other = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    stamp = torch.zeros((256 * 4, 8), dtype=torch.int64, device="cuda")
    for _ in range(12):
        _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bc), _lib.ptr(out), M, N, K,
                                          _lib.FP8_EPI_LN_BIAS_BF16, _lib.ptr(stats), _lib.ptr(other), _lib.ptr(stamp), None, 0, _lib.stream()), "gemm")
    torch.cuda.synchronize()
    s = stamp.cpu().numpy()
    names = ("wait + barrier", "K-tile 0 fragment reads", "K-loop", "barrier + next tile's requests", "epilogue")
    tot = 0.0
    for i, n in enumerate(names):
        v = statistics.median(s[:, i].tolist())
        tot += v
        print(f"  {n:34s} {v:8.0f} cycles   (p10 {sorted(s[:, i].tolist())[len(s) // 10]:6.0f}, p90 {sorted(s[:, i].tolist())[len(s) * 9 // 10]:6.0f})")
    print(f"  {'tile':34s} {tot:8.0f} cycles;  {N // 256 * (M // 256) / 256:.1f} tiles per workgroup")


if __name__ == "__main__":
    main()
