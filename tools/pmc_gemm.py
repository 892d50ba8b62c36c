#!/usr/bin/env python3
"""A few launches of the vit.qkv GEMM (product path) for rocprofv3 --pmc runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()
M, N, K = 32896, 3072, 1024
a = torch.randn((M + 255) // 256 * 256, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros(a.shape[0], N, device="cuda", dtype=torch.bfloat16)
lib.keds_gemm_force_small(int(os.environ.get("VARIANT", "0x100"), 16))
for _ in range(5):
    ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M)
torch.cuda.synchronize()
