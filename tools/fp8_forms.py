#!/usr/bin/env python3
"""The four MXFP8 GEMMs of a ViT-L/14 block at B = 128 (32,768 full-tile rows) with the epilogues the tower uses, 8-wave kernel
(keds_mxfp8_debug(16)) against the 4-wave persistent kernel, interleaved rounds, medians of per-launch event pairs."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402

lib = _lib.load()
M = 32768
ITERS, ROUNDS = int(os.environ.get("ITERS", "20")), int(os.environ.get("ROUNDS", "3"))


def quant(x):
    rows, K = x.shape
    q = torch.zeros((rows, K), dtype=torch.uint8, device="cuda")
    s = torch.full((K // 128, rows, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(x), 0, rows, K, rows, _lib.ptr(q), _lib.ptr(s), _lib.stream()), "q")
    return q, s


def fold(w, b, gamma, beta):
    N, K = w.shape
    wq = torch.zeros((N, K), dtype=torch.uint8, device="cuda")
    ws = torch.full((K // 128, N, 4), 127, dtype=torch.uint8, device="cuda")
    bc = torch.zeros(2 * N, device="cuda")
    _lib.check(lib.keds_fold_layernorm_mxfp8(_lib.ptr(w), _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), N, K, N, _lib.ptr(wq), _lib.ptr(ws),
                                             _lib.ptr(bc), _lib.stream()), "fold")
    return wq, ws, bc


def main():
    for tag, N, K, epi in (("qkv ", 3072, 1024, "ln"), ("out ", 1024, 1024, "resid"), ("c_fc", 4096, 1024, "gelu"), ("proj", 1024, 4096, "resid")):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5 * 0.5
        b = torch.randn(N, device="cuda") * 0.02
        aq, as_ = quant(a)
        if epi == "resid":
            wq, ws = quant(w)
            x = (torch.randn(M, N, device="cuda") * 0.5).half()
            stats = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
            q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
            qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
            call = lambda: lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(b), _lib.ptr(x), M, N, K,
                                                  _lib.FP8_EPI_RESID_STATS_MX_H, _lib.ptr(stats), None, _lib.ptr(q), _lib.ptr(qs), M, _lib.stream())
        else:
            wq, ws, bc = fold(w, b, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
            stats = (torch.stack([a.sum(1), (a * a).sum(1)], dim=1).double() * 2.0 ** 28).round().to(torch.int64).contiguous()
            other = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
            out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
            q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
            qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
            code = _lib.FP8_EPI_LN_QGELU_MX if epi == "gelu" else _lib.FP8_EPI_LN_BIAS_BF16
            call = lambda: lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bc),
                                                  None if epi == "gelu" else _lib.ptr(out), M, N, K, code, _lib.ptr(stats), _lib.ptr(other),
                                                  _lib.ptr(q), _lib.ptr(qs), M, _lib.stream())
        res = {}
        for rnd in range(ROUNDS):
            for form, dbg in (("8 waves", 16), ("4 waves, persistent", 0)):
                lib.keds_mxfp8_debug(dbg)
                ev = []
                for it in range(ITERS):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.check(call(), "gemm")
                    e1.record()
                    ev.append((e0, e1))
                torch.cuda.synchronize()
                res.setdefault(form, []).append(statistics.median(x.elapsed_time(y) * 1e3 for x, y in ev[4:]))
                if epi == "resid":
                    x.mul_(0.25)
        lib.keds_mxfp8_debug(0)
        fl = 2.0 * M * N * K
        print(f"{tag} (N {N}, K {K}, {epi}):  " + "   ".join(f"{f}: {statistics.median(v):7.1f} us = {fl / statistics.median(v) / 1e6:6.0f} TFLOP/s"
                                                          for f, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
