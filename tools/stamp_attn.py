#!/usr/bin/env python3
"""Where a workgroup of the 8-wave S = 257 attention kernel spends its cycles (in-kernel s_memtime stamps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()
B, S, H = 128, 257, 16
qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
buf = torch.zeros(B * H * 64, dtype=torch.int64, device="cuda")
lib.keds_attention_stamp_buffer(_lib.ptr(buf))
lib.keds_attention_debug(64 + 8)
for _ in range(3):
    ops.attention(qkv, B, S, H, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attention(qkv, B, S, H, False); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
lib.keds_attention_debug(0)
st = buf.cpu().reshape(B * H, 8, 8).double()
span = (st[:, :, 5].max() - st[:, :, 4].min()).item()
print(f"launch {us:.1f} us; first entry -> last exit {span:.0f} cycles => {span / us / 1e3:.2f} GHz")
names = ["staging (Q issued, K/V DMA landed, barrier)", "key tiles 0-3 + wait for keys 128-255 + barrier", "last-query partial + key tiles 4-7", "last key + normalise + stores issued"]
life = (st[:, :, 5] - st[:, :, 4])
print(f"wave lifetime mean {life.mean():.0f} cycles (max {life.max():.0f})")
for i, n in enumerate(names):
    v = st[:, :, i]
    print(f"  {n:46s} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}   ({100 * v.mean() / life.mean():4.1f} % of a wave's life)")
v = st[:, :, 7]
print(f"  {'store drain':46s} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}")
# workgroups per CU over time: HW_ID -> (xcc, se, cu)
raw = buf.cpu().reshape(B * H, 8, 8)
hw = raw[:, 0, 6] & 0xFFFFFFFF
xcc = (raw[:, 0, 6] >> 32) & 0xF
print("distinct HW_ID CU fields:", len(set(((hw >> 8) & 0xF).tolist())), "x SE", len(set(((hw >> 13) & 0x7).tolist())))

# per-CU residency: key = (xcc, se, cu); a workgroup is resident from its first wave's entry to its last wave's exit
import collections
key = (xcc * 64 + ((hw >> 13) & 0x7) * 16 + ((hw >> 8) & 0xF)).tolist()
t_in = raw[:, :, 4].min(dim=1).values.tolist()
t_out = raw[:, :, 5].max(dim=1).values.tolist()
cus = collections.defaultdict(list)
for k, a, b_ in zip(key, t_in, t_out):
    cus[k].append((a, b_))
print("CUs seen:", len(cus), " workgroups per CU min/max:", min(len(v) for v in cus.values()), max(len(v) for v in cus.values()))
occ, spans, gaps, idle1 = [], [], [], []
for k, v in cus.items():
    v.sort()
    span = max(b_ for _, b_ in v) - v[0][0]
    resident = sum(b_ - a for a, b_ in v)
    occ.append(resident / span); spans.append(span)
    # time with fewer than two workgroups resident
    ev = sorted([(a, 1) for a, _ in v] + [(b_, -1) for _, b_ in v])
    n, last, lt2 = 0, ev[0][0], 0
    for t, d in ev:
        if n < 2: lt2 += t - last
        n += d; last = t
    idle1.append(lt2 / span)
    # gap between an exit and the next entry on this CU
    outs = sorted(b_ for _, b_ in v)
    ins = sorted(a for a, _ in v)[2:]
    for o_, i_ in zip(outs, ins):
        gaps.append(i_ - o_)
import statistics as stt
print(f"per-CU span mean {stt.mean(spans):.0f} cycles (max {max(spans):.0f}); mean resident workgroups {stt.mean(occ):.2f} of 2; "
      f"time with < 2 resident {100 * stt.mean(idle1):.1f} %")
if gaps:
    print(f"exit -> next entry on the same CU: mean {stt.mean(gaps):.0f} cycles, median {stt.median(gaps):.0f}, max {max(gaps):.0f}")
