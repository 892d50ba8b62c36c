#!/usr/bin/env python3
"""One launch shape of the 256^2 GEMM a few times, for rocprofv3 --pmc passes (tools/pmc_gemm.sh).
SHAPE=qkv (32768 x 3072 x 1024, LN-folded fp16 operands, the most frequent tower GEMM) or proj (32768 x 1024 x 4096, fp16 residual)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
from keds_amd._lib import ptr, check, stream
lib = _lib.load()
M = 32768
shape = os.environ.get("SHAPE", "qkv")
N, K, epi = (3072, 1024, _lib.EPI_LN_BIAS_BF16_H) if shape == "qkv" else (1024, 4096, _lib.EPI_RESID_STATS_F16)
f16 = shape == "qkv"
a = torch.randn(M, K, device="cuda").to(torch.float16 if f16 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.float16 if f16 else torch.bfloat16)
bias = torch.randn(2 * N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if f16 else torch.float16)
stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
stats[:, 1] = (1 << 28) * K
other = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
_lib.ensure_gemm_workspace(torch.device("cuda"))
for _ in range(6):
    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, epi, ptr(stats), 0, ptr(other) if f16 else None,
                               stream()), "gemm")
torch.cuda.synchronize()
