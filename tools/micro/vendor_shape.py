"""Reference point only (NOT part of the product): run the vendor BLAS GEMM and ours once each on the c_fc / c_proj / qkv shapes so
that a rocprofv3 --kernel-trace of this script shows each kernel's launch shape (workgroup, grid, LDS, VGPR / AGPR counts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keds_amd import _lib, ops
lib = _lib.load()
for M, N, K, tag in [(32768, 4096, 1024, "fc"), (32768, 1024, 4096, "proj"), (32768, 3072, 1024, "qkv")]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.t())
        ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M)
    torch.cuda.synchronize()
