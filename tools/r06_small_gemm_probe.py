#!/usr/bin/env python3
"""Where a mid-size GEMM launch's time goes: [M, K] x [N, K]^T with the residual epilogue at the text tower's shapes, K swept
(fixed part = prologue + epilogue, slope = time per K-tile of 64), every kernel form the dispatcher can be forced to."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402

lib = _lib.load()
FORMS = (("dispatcher", 0), ("8 waves 256^2", 3 << 11), ("4 waves 256^2", 1 << 11), ("128^2", 1), ("128^2 no split", 1 | (1 << 9)))


def run(M, N, K, epi, forms=FORMS, iters=20):
    ln = epi != _lib.EPI_RESID_STATS_F16
    Mp = (M + 255) // 256 * 256
    a = torch.randn(Mp, K, device="cuda")
    a = a.half() if ln else a.to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    w = w.half() if ln else w.to(torch.bfloat16)
    bias = torch.randn(2 * N, device="cuda")
    stats = torch.zeros(Mp, 2, device="cuda", dtype=torch.int64)
    stats[:, 0] = int(0.1 * K * 2 ** 28)
    stats[:, 1] = int(1.0 * K * 2 ** 28)
    other = torch.zeros(Mp, 2, device="cuda", dtype=torch.int64)
    out = torch.randn(Mp, N, device="cuda").half() if not ln else torch.zeros(Mp, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for name, flag in forms:
            lib.keds_gemm_force_small(flag)
            ev = []
            for it in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, epi, ptr(stats), 0, ptr(other) if ln else None,
                                           stream()), "gemm")
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            res.setdefault(name, []).append(statistics.median(x.elapsed_time(y) * 1e3 for x, y in ev[4:]))
            if not ln:
                out.normal_()
    lib.keds_gemm_force_small(0)
    return {n: statistics.median(v) for n, v in res.items()}


def main():
    _lib.ensure_gemm_workspace("cuda")
    for M in (int(x) for x in os.environ.get("MS", "11008,5504").split(",")):
        for N, tag, epi in ((768, "resid", _lib.EPI_RESID_STATS_F16), (2304, "ln   ", _lib.EPI_LN_BIAS_BF16_H), (3072, "gelu ", _lib.EPI_LN_QGELU_BF16_H)):
            for K in (128, 256, 768, 1536, 3072):
                r = run(M, N, K, epi)
                fl = 2.0 * M * N * K
                print(f"M {M:6d} N {N:5d} {tag} K {K:5d}:  " + "   ".join(f"{n}: {v:6.1f} us ({fl / v / 1e6:5.0f} TF)" for n, v in r.items()), flush=True)


if __name__ == "__main__":
    main()
