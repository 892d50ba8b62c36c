#!/usr/bin/env python3
"""Where the two-accumulator-set kernel (gemm_duo.hip) differs from the 4-wave persistent kernel: per shape and epilogue, the
count of differing elements, the tiles they sit in, and inside the first bad tile the 16 x 16 blocks (row group mi x column
block) that differ.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402


def main():
    lib = _lib.load()
    _lib.ensure_gemm_workspace("cuda")
    shapes = [(256, 256, 1024), (256, 1024, 1024), (1024, 256, 1024), (4096, 4096, 1024), (2048, 3072, 1024), (4096, 3072, 1024),
              (33024, 3072, 1024), (4096, 4096, 2048)]
    for M, N, K in shapes:
        for epi, tag in ((_lib.EPI_LN_BIAS_BF16_H, "ln  "), (_lib.EPI_LN_QGELU_BF16_H, "gelu")):
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            a = (torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.2).half()
            w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
            bias = torch.randn(2 * N, generator=g, device="cuda")
            stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
            af = a.float()
            stats[:, 0] = (af.sum(1) * 2 ** 28).long()
            stats[:, 1] = ((af * af).sum(1) * 2 ** 28).long()
            outs = []
            for duo in (0, 1):
                lib.keds_gemm_duo_enable(duo)
                lib.keds_gemm_force_small(0 if duo else (2 << 11))
                o = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                other = torch.full((M, 2), 7, device="cuda", dtype=torch.int64)
                check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(o), N, M, N, K, epi, ptr(stats), 0, ptr(other), stream()), "gemm")
                torch.cuda.synchronize()
                outs.append((o, other))
            lib.keds_gemm_force_small(0)
            lib.keds_gemm_duo_enable(-1)
            (o0, z0), (o1, z1) = outs
            bad = (o0.view(torch.int16) != o1.view(torch.int16))
            nb = int(bad.sum())
            line = f"M {M:6d} N {N:5d} K {K:5d} {tag}: {nb:9d} of {M * N} differ; other-stats equal {bool(torch.equal(z0, z1))}"
            if nb:
                tiles = bad.view(M // 256, 256, N // 256, 256).permute(0, 2, 1, 3).reshape(M // 256, N // 256, -1).any(-1)
                line += f"; bad tiles {int(tiles.sum())} of {tiles.numel()}"
                tm, tn = [int(x) for x in tiles.nonzero()[0]]
                t = bad[tm * 256:(tm + 1) * 256, tn * 256:(tn + 1) * 256]
                blk = t.view(16, 16, 16, 16).permute(0, 2, 1, 3).reshape(16, 16, -1).float().mean(-1)
                line += f"\n   first bad tile ({tm},{tn}); share of differing elements per 16-row group (rows) x 16-column block (cols):\n"
                line += "\n".join("   " + " ".join(f"{float(v):4.2f}" for v in row) for row in blk)
                d = (o0.float() - o1.float())[tm * 256:(tm + 1) * 256, tn * 256:(tn + 1) * 256]
                line += f"\n   max |diff| {float(d.abs().max()):.4g}, duo finite {bool(torch.isfinite(o1.float()).all())}, duo zero share {float((o1 == 0).float().mean()):.3f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
