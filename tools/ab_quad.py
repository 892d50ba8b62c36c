#!/usr/bin/env python3
"""Same-process A/B of the 256 x 256 GEMM on the ViT-L/14 shapes (B = 128: 32,768 full-tile rows) with the epilogues the
towers run: the 8-wave kernel (round 1-2) against the 4-wave kernel with AGPR accumulators (round 3,
keds_gemm_force_small bit 11).  Interleaved rounds, median / min per variant (cdna_hip_programming.md rule 24)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402


def main():
    lib = _lib.load()
    iters, rounds = int(os.environ.get("ITERS", "20")), int(os.environ.get("ROUNDS", "5"))
    M = 32768
    shapes = ((3072, 1024, "qkv ", _lib.EPI_LN_BIAS_BF16_H), (4096, 1024, "fc  ", _lib.EPI_LN_QGELU_BF16_H),
              (1024, 1024, "out ", _lib.EPI_RESID_STATS_F16), (1024, 4096, "proj", _lib.EPI_RESID_STATS_F16),
              (4096, 1024, "fc plain bias", _lib.EPI_BIAS_BF16), (1024, 4096, "proj plain bias", _lib.EPI_BIAS_BF16))
    _lib.ensure_gemm_workspace("cuda")
    for N, K, tag, epi in shapes:
        ln = epi in (_lib.EPI_LN_BIAS_BF16_H, _lib.EPI_LN_QGELU_BF16_H)
        a = torch.randn(M, K, device="cuda")
        a = a.half() if ln else a.to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        w = w.half() if ln else w.to(torch.bfloat16)
        bias = torch.randn(2 * N, device="cuda")
        stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
        stats[:, 0] = int(0.1 * K * 2 ** 28)
        stats[:, 1] = int(1.0 * K * 2 ** 28)
        other = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
        out = torch.randn(M, N, device="cuda").half() if epi == _lib.EPI_RESID_STATS_F16 else torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        forms = (("8 waves", 3 << 11), ("4 waves", 1 << 11), ("4 waves, persistent", 2 << 11), ("4 waves, 2-deep A", (1 << 11) | (1 << 16)),
                 ("4 waves, persistent, no deferred stores", (2 << 11) | (1 << 17)),
                 ("dispatcher (round 5: two accumulator sets where it applies)", 0))
        if os.environ.get("FORMS"):                                      # e.g. FORMS="4 waves, persistent;dispatcher"
            forms = tuple(f for f in forms if any(f[0].startswith(k) for k in os.environ["FORMS"].split(";")))
        res = {name: [] for name, _ in forms}
        for rnd in range(rounds):
            for name, flag in forms:
                lib.keds_gemm_force_small(flag)

                def run():
                    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, epi,
                                               ptr(stats) if epi != _lib.EPI_BIAS_BF16 else None, 0, ptr(other) if ln else None,
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
                res[name].append(e0.elapsed_time(e1) / iters * 1e3)
                if epi == _lib.EPI_RESID_STATS_F16:
                    out.normal_()
        lib.keds_gemm_force_small(0)
        for name, v in res.items():
            med, mn = statistics.median(v), min(v)
            print(f"{tag:16s} {name:40s} median {med:7.1f} us ({2.0 * M * N * K / med / 1e6:7.1f} TF)   min {mn:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
