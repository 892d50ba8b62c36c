#!/usr/bin/env python3
"""Which operand's coldness costs the MXFP8 residual GEMMs their in-step time (out-proj 68 -> 92 us, c_proj 129 -> 166 us in the
fp8 encoder step)?  As tools/gemm_cold_matrix.py: the GEMM runs back to back while ONE streamed operand rotates through buffers
that together exceed the 256 MiB memory-side cache (every launch finds it in HBM) and the others stay on one buffer (warm):
A (MXFP8 activations + scales), the fp16 residual tile the epilogue read-modify-writes, the MXFP8 copy it writes."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402

lib = _lib.load()
M = 32768
ITERS, ROUNDS = int(os.environ.get("ITERS", "18")), int(os.environ.get("ROUNDS", "3"))


def quant(x):
    rows, K = x.shape
    q = torch.zeros((rows, K), dtype=torch.uint8, device="cuda")
    s = torch.full((K // 128, rows, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(x), 0, rows, K, rows, _lib.ptr(q), _lib.ptr(s), _lib.stream()), "q")
    return q, s


def main():
    for tag, N, K in (("out ", 1024, 1024), ("proj", 1024, 4096)):
        nb_a = 10 if K == 1024 else 3                                      # 10 x 33.5 MB / 3 x 134 MB
        nb_h = 6                                                           # 6 x 67 MB
        a = [quant(torch.randn(M, K, device="cuda")) for _ in range(nb_a)]
        h = [(torch.randn(M, N, device="cuda") * 0.5).half() for _ in range(nb_h)]
        qo = [torch.zeros((M, N), dtype=torch.uint8, device="cuda") for _ in range(nb_h)]
        qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
        wq, ws = quant(torch.randn(N, K, device="cuda") * K ** -0.5 * 0.1)
        bias = torch.randn(N, device="cuda") * 0.02
        st = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
        res = {}
        for rnd in range(ROUNDS):
            for form, dbg in (("8 waves", 16), ("4 waves, persistent", 0)):
                for mode, ca, ch, cq in (("none", 0, 0, 0), ("A cold", 1, 0, 0), ("resid cold", 0, 1, 0), ("copy cold", 0, 0, 1), ("all", 1, 1, 1)):
                    lib.keds_mxfp8_debug(dbg)
                    ev = []
                    for it in range(ITERS):
                        aq, as_ = a[it % nb_a] if ca else a[0]
                        hi = h[it % nb_h] if ch else h[0]
                        qi = qo[it % nb_h] if cq else qo[0]
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), M, _lib.ptr(wq), _lib.ptr(ws), N, _lib.ptr(bias), _lib.ptr(hi),
                                                          M, N, K, _lib.FP8_EPI_RESID_STATS_MX_H, _lib.ptr(st), None, _lib.ptr(qi), _lib.ptr(qs), M,
                                                          _lib.stream()), "gemm")
                        e1.record()
                        ev.append((e0, e1))
                    torch.cuda.synchronize()
                    res.setdefault((form, mode), []).append(statistics.median(x.elapsed_time(y) * 1e3 for x, y in ev[10:]))
                    for t in h:
                        t.mul_(0.5)
        lib.keds_mxfp8_debug(0)
        modes = ("none", "A cold", "resid cold", "copy cold", "all")
        print(f"{tag} (M {M}, N {N}, K {K}), us per launch:     " + "   ".join(f"{m:>10s}" for m in modes))
        for form in ("8 waves", "4 waves, persistent"):
            print(f"  {form:24s} " + "   ".join(f"{statistics.median(res[(form, m)]):10.1f}" for m in modes), flush=True)
        del a, h, qo


if __name__ == "__main__":
    main()
