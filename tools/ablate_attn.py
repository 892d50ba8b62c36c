#!/usr/bin/env python3
"""Timing-only ablations of the ViT attention kernel (B=128, S=257, 16 heads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()
B, S, H = 128, 257, 16
qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
for code, name in [(0, "product (8 waves, 32-key tiles)"), (32, "4-wave tail kernel"), (16, "generic kernel (padded to 288 keys)"),
                   (0, "product again"), (32, "4-wave tail again"), (16, "generic again"),
                   (65, "8w: no K/V staging"), (66, "8w: no QK^T"), (67, "8w: no exp"), (68, "8w: no PV"), (69, "8w: staging only"), (73, "8w: no last-query row"), (0, "product again"),
                   (1, "no K/V staging"), (2, "no QK^T"), (3, "no exp"), (4, "no PV"), (5, "staging only (no q loop)")]:
    lib.keds_attention_debug(code)
    for _ in range(3):
        ops.attention(qkv, B, S, H, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, B, S, H, False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:36s} {us:7.1f} us  ({4.0*B*H*S*S*64/us/1e6:6.0f} TF-equivalent)", flush=True)
lib.keds_attention_debug(0)
