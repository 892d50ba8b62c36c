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
hw = st[:, 0, 6].long()
print("distinct HW_ID CU fields:", len(set(((hw >> 8) & 0xF).tolist())), "x SE", len(set(((hw >> 13) & 0x7).tolist())))
