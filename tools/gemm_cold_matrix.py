#!/usr/bin/env python3
"""Which operand's coldness costs a residual tower GEMM its time, and which kernel form tolerates it?

tools/gemm_instep.py shows out-proj at 68 us back to back on one buffer set and 83 us in the block's launch order: in the step
its operands were last touched 470 MB of other traffic ago, more than the 256 MiB memory-side cache holds.  Here the GEMM
runs back to back while ONE of its streamed operands rotates through 6 buffers (6 x 67 MB = 400 MB > the cache: every
launch finds that operand in HBM) and the other stays on one buffer (warm):

    A cold      the MFMA A operand (attention output / MLP hidden) comes from HBM: felt by the K-loop's LDS-DMA
    resid cold  the fp16 residual tile the epilogue reads-modifies-writes comes from HBM
    both / none

for every kernel form of the 256 x 256 tile (keds_gemm_force_small bits 11-12, 16).  Interleaved rounds, medians."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402

lib = _lib.load()
M = 32768
NBUF = 6
ITERS, ROUNDS = int(os.environ.get("ITERS", "18")), int(os.environ.get("ROUNDS", "3"))
FORMS = (("8 waves", 3 << 11), ("4 waves, 3-deep A", 1 << 11), ("4 waves, 2-deep A", (1 << 11) | (1 << 16)), ("4 waves, persistent", 2 << 11))
if os.environ.get("FORMS"):
    FORMS = tuple(f for f in FORMS if f[0] in os.environ["FORMS"].split(";"))


def main():
    dev = "cuda"
    _lib.ensure_gemm_workspace(dev)
    for tag, N, K in (("out ", 1024, 1024), ("proj", 1024, 4096)):
        nb_a = NBUF if K == 1024 else 2                                   # (the 268 MB MLP hidden matrix is colder than the cache by itself)
        a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nb_a)]
        h = [(torch.randn(M, N, device=dev) * 0.5).half() for _ in range(NBUF)]
        w = (torch.randn(N, K, device=dev) * K ** -0.5 * 0.1).bfloat16()
        bias = torch.randn(N, device=dev) * 0.02
        st = torch.zeros(M, 2, device=dev, dtype=torch.int64)
        res = {}
        for rnd in range(ROUNDS):
            for form, flag in FORMS:
                for mode, ca, ch in (("none", 0, 0), ("A cold", 1, 0), ("resid cold", 0, 1), ("both", 1, 1)):
                    lib.keds_gemm_force_small(flag)
                    ev = []
                    for it in range(ITERS):
                        ai = a[it % nb_a] if ca else a[0]
                        hi = h[it % NBUF] if ch else h[0]
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        check(lib.keds_gemm_bt_ex2(ptr(ai), K, ptr(w), ptr(bias), ptr(hi), N, M, N, K, _lib.EPI_RESID_STATS_F16,
                                                   ptr(st), 0, None, stream()), "gemm")
                        e1.record()
                        ev.append((e0, e1))
                    torch.cuda.synchronize()
                    res.setdefault((form, mode), []).append(statistics.median(x.elapsed_time(y) * 1e3 for x, y in ev[NBUF:]))
                    for t in h:                                           # keep the fp16 stream bounded
                        t.mul_(0.5)
        lib.keds_gemm_force_small(0)
        print(f"{tag} (M {M}, N {N}, K {K}), us per launch:   " + "   ".join(f"{m:>10s}" for m in ("none", "A cold", "resid cold", "both")))
        for form, _ in FORMS:
            print(f"  {form:34s} " + "   ".join(f"{statistics.median(res[(form, m)]):10.1f}" for m in ("none", "A cold", "resid cold", "both")), flush=True)
        del a, h


if __name__ == "__main__":
    main()
