#!/usr/bin/env python3
"""The four GEMMs of a text-tower block (width 768, B = 128 x 77 tokens = 9,856 rows = 38.5 row tiles of 256) with the tower's
epilogues, every kernel form the dispatcher can be forced to (keds_gemm_force_small), alone; and the same shapes at M = 9,728 /
10,240 (38 / 40 whole row tiles) to price the ragged tile.  Interleaved rounds, medians of per-launch event pairs."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402

lib = _lib.load()
ITERS, ROUNDS = int(os.environ.get("ITERS", "16")), int(os.environ.get("ROUNDS", "3"))
FORMS = (("dispatcher", 0), ("8 waves", 3 << 11), ("4 waves", 1 << 11), ("4 waves, persistent", 2 << 11), ("128 x 128 tiles", 1))


def main():
    _lib.ensure_gemm_workspace("cuda")
    W_ = int(os.environ.get("WIDTH", "768"))
    for M in (int(x) for x in os.environ.get("MS", "9856,9728,10240").split(",")):
        for N, K, tag, epi in ((3 * W_, W_, "qkv ", _lib.EPI_LN_BIAS_BF16_H), (W_, W_, "out ", _lib.EPI_RESID_STATS_F16),
                               (4 * W_, W_, "fc  ", _lib.EPI_LN_QGELU_BF16_H), (W_, 4 * W_, "proj", _lib.EPI_RESID_STATS_F16)):
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
            for rnd in range(ROUNDS):
                for name, flag in FORMS:
                    lib.keds_gemm_force_small(flag)
                    ev = []
                    for it in range(ITERS):
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
            fl = 2.0 * M * N * K
            print(f"M {M:6d} {tag} (N {N}, K {K}):  " + "   ".join(f"{n}: {statistics.median(v):6.1f} us ({fl / statistics.median(v) / 1e6:5.0f} TF)" for n, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
