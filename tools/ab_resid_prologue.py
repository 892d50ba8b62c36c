#!/usr/bin/env python3
"""Same-process A/B of the fp16-residual GEMM (out-proj / c_proj shapes of ViT-L/14 at B = 128): residual + bias as the
accumulators' initial value (round 3, default) against the residual loaded in the epilogue (round 2; keds_gemm_force_small
bit 10).  Interleaved rounds, median and min per variant (cdna_hip_programming.md rule 24).  GPU only."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402


def main():
    lib = _lib.load()
    iters, rounds = int(os.environ.get("ITERS", "20")), int(os.environ.get("ROUNDS", "7"))
    M = 32768
    for N, K, tag in ((1024, 1024, "out"), (1024, 4096, "proj")):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        x16 = torch.randn(M, N, device="cuda").half()
        stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
        _lib.ensure_gemm_workspace(a.device)
        res = {"acc-init (r03)": [], "epilogue load (r02)": []}
        for rnd in range(rounds):
            for name, flag in (("acc-init (r03)", (1 << 10) | (3 << 11)), ("epilogue load (r02)", 3 << 11)):
                lib.keds_gemm_force_small(flag)

                def run():
                    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(x16), N, M, N, K, _lib.EPI_RESID_STATS_F16,
                                               ptr(stats), 0, None, stream()), "gemm")
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
                x16.normal_()
        lib.keds_gemm_force_small(0)
        for name, v in res.items():
            med, mn = statistics.median(v), min(v)
            print(f"{tag:5s} {name:22s} median {med:7.1f} us ({2.0 * M * N * K / med / 1e6:7.1f} TF)   min {mn:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
