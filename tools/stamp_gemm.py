#!/usr/bin/env python3
"""In-kernel timeline of the 256^2 GEMM (qkv shape, 32768 x 3072 x 1024): s_memtime stamps of the stamped diagnostic
build (keds_gemm_force_small bit 12): prologue / K-loop / epilogue cycles per wave and the cycles every wave spends in the
K-loop's `s_waitcnt vmcnt(0)` and `s_barrier`.  GPU only; not a product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
from keds_amd._lib import ptr, check, stream
lib = _lib.load()
QUADBITS = {"0": 3, "1": 1, "2": 2}[os.environ.get("QUAD", "0")]     # 8-wave kernel | 4-wave | 4-wave, early DMA
SHAPE = os.environ.get("SHAPE", "qkv")          # qkv (LN-folded epilogue) | out | proj (fp16-residual epilogue)
M, N, K = {"qkv": (32768, 3072, 1024), "out": (32768, 1024, 1024), "proj": (32768, 1024, 4096)}[SHAPE]
RESID = SHAPE != "qkv"
if RESID:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda").half()
    stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64)
    EPI = _lib.EPI_RESID_STATS_F16
else:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(2 * N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(M, 2, device="cuda", dtype=torch.int64); stats[:, 1] = (1 << 28) * K
    EPI = _lib.EPI_LN_BIAS_BF16_H
tiles = (M // 256) * (N // 256)
buf = torch.zeros(tiles * 8 * 8 + tiles * 8, device="cuda", dtype=torch.int64)
_lib.ensure_gemm_workspace(torch.device("cuda"))
def run(aux2):
    if RESID:
        out.normal_()           # keep the fp16 stream bounded over repeated accumulation
        stats.zero_()
    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, EPI, ptr(stats), 0, ptr(aux2) if aux2 is not None else None, stream()), "gemm")
other = None if RESID else torch.zeros(M, 2, device="cuda", dtype=torch.int64)
for _ in range(20):
    run(other)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
DELAY = 0
def gemm(aux2):
    check(lib.keds_gemm_bt_ex2(ptr(a), K, ptr(w), ptr(bias), ptr(out), N, M, N, K, EPI, ptr(stats), DELAY, ptr(aux2) if aux2 is not None else None, stream()), "gemm")
lib.keds_gemm_force_small(QUADBITS << 11)
run(other); torch.cuda.synchronize()
e0.record(); gemm(other); e1.record(); torch.cuda.synchronize()
print(f"== SHAPE={SHAPE} {M}x{N}x{K}: product launch (unstamped) {e0.elapsed_time(e1) * 1e3:.1f} us")
VARIANTS = (((1, "product epilogue", 0), (3, "no stores", 0), (4, "no statistics atomics", 0), (5, "no residual loads / stores / atomics", 0),
             (1, "product epilogue, half of the CUs 25k cycles late", 25000), (1, "product epilogue again", 0))
            if RESID else ((1, "product epilogue", 0), (2, "no statistics loads", 0), (3, "no stores", 0),
                           (1, "product epilogue, half of the CUs 25k cycles late", 25000), (1, "product epilogue again", 0)))
for variant, what, DELAY in VARIANTS:
    lib.keds_gemm_force_small((variant << 13) | (QUADBITS << 11))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        run(buf)
    if RESID:
        out.normal_(); stats.zero_()
    gemm(buf)
    torch.cuda.synchronize()
    e0.record(); gemm(buf); e1.record()
    torch.cuda.synchronize()
    lib.keds_gemm_force_small(0)
    t = buf[:tiles * 64].view(tiles, 8, 8).double().cpu()
    hw = buf[tiles * 64:].view(tiles, 8).cpu()
    print(f"== {what}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {tiles} tiles")
    names = ["prologue", "K-loop", "epilogue issue", "vm_wait(sum)", "barrier_wait(sum)", "store drain"]
    for i, n in enumerate(names):
        v = t[:, :, i]
        print(f"{n:18s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f} cycles;  per wave index: " +
              " ".join(f"{v[:, wv].mean():7.0f}" for wv in range(8)))
    print(f"K-loop per K-tile {t[:, :, 1].mean() / (K // 64):.0f} cycles (MFMA alone: 2048)")
    # gaps between consecutive workgroups on one CU: wave 0 of each workgroup, CU = (xcc, se, sh, cu) of HW_ID
    raw = buf[:tiles * 64].view(tiles, 8, 8).cpu()
    ids = hw[:, 0]
    cu_key = ((ids >> 32) & 0xF) * 4096 + ((ids >> 8) & 0xFF) * 1    # xcc | se/sh/cu bits [15:8] of HW_ID
    ent, end = raw[:, :, 6].min(dim=1).values, raw[:, :, 7].max(dim=1).values
    gaps, lifes = [], []
    for key in cu_key.unique().tolist():
        sel = (cu_key == key).nonzero().flatten()
        order = sel[ent[sel].argsort()]
        for a_, b_ in zip(order[:-1].tolist(), order[1:].tolist()):
            gaps.append(int(ent[b_] - end[a_]))
        lifes += [int(end[i] - ent[i]) for i in order.tolist()]
    g = torch.tensor(gaps, dtype=torch.float64); l = torch.tensor(lifes, dtype=torch.float64)
    print(f"CUs seen {len(cu_key.unique())}; workgroup lifetime (first wave in -> last wave out) mean {l.mean():.0f}; gap to the next workgroup on the same CU: "
          f"mean {g.mean():.0f} median {g.median():.0f} min {g.min():.0f} max {g.max():.0f} cycles")
