#!/usr/bin/env python3
"""How much of a tower GEMM's time in the step is its operands coming from HBM instead of the Infinity Cache?  Each launch is
timed alone (hipEvents), once back to back on the same buffers (hot: the 256 MB memory-side cache holds them) and once behind
a 1 GB fill that evicts everything (cold: what the step sees for tensors last touched a layer ago)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
from keds_amd._lib import ptr, check, stream
lib = _lib.load()
M = 32768
_lib.ensure_gemm_workspace("cuda")
junk = torch.empty(1 << 28, device="cuda", dtype=torch.float32)        # 1 GiB
for N, K, tag, epi in ((3072, 1024, "qkv ", _lib.EPI_LN_BIAS_BF16_H), (4096, 1024, "fc  ", _lib.EPI_LN_QGELU_BF16_H),
                       (1024, 1024, "out ", _lib.EPI_RESID_STATS_F16), (1024, 4096, "proj", _lib.EPI_RESID_STATS_F16)):
    ln = epi != _lib.EPI_RESID_STATS_F16
    a = torch.randn(M, K, device="cuda"); a = a.half() if ln else a.bfloat16()
    w = torch.randn(N, K, device="cuda") * K ** -0.5; w = w.half() if ln else w.bfloat16()
    bias = torch.randn(2 * N, device="cuda")
    stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64); stats[:, 0] = int(0.1 * K * 2 ** 28); stats[:, 1] = int(1.0 * K * 2 ** 28)
    other = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
    out = torch.randn(M, N, device="cuda").half() if not ln else torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, epi, ptr(stats), 0, ptr(other) if ln else None, stream()), "gemm")
    res = {}
    for mode in ("hot", "cold", "hot", "cold"):
        ts = []
        for it in range(12):
            if mode == "cold":
                junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            if it >= 2:
                ts.append(e0.elapsed_time(e1) * 1e3)
            if not ln and it % 4 == 3:
                out.normal_()
        res.setdefault(mode, []).append(statistics.median(ts))
    print(f"{tag}: hot {min(res['hot']):6.1f} us   cold {min(res['cold']):6.1f} us", flush=True)
