"""Reference point only (NOT part of the product): what torch.matmul (hipBLASLt / rocBLAS) reaches on the ViT-L/14
GEMM shapes on this GPU, next to our kernels.  bf16 in, bf16 out, random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keds_amd import _lib, ops
lib = _lib.load()
for M, N, K, tag in [(32768, 3072, 1024, "qkv"), (32768, 4096, 1024, "fc"), (32768, 1024, 4096, "proj"), (32768, 1024, 1024, "out"), (8192, 8192, 8192, "square 8k")]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for name, fn in [("torch.matmul(a, w.T)", lambda: torch.matmul(a, w.t())), ("torch F.linear+bias", lambda: torch.nn.functional.linear(a, w, bias.to(torch.bfloat16))),
                     ("keds 256^2 kernel", lambda: ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M))]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[name] = 2.0 * M * N * K / us / 1e6
    print(f"{tag:10s} " + "   ".join(f"{k}: {v:7.1f} TF" for k, v in res.items()), flush=True)
