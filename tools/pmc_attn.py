#!/usr/bin/env python3
"""A few launches of the ViT attention (B=128, S=257, 16 heads) for counter collection (tools/pmc_attn.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib, ops
lib = _lib.load()
B, S, H = 128, 257, 16
qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
for _ in range(4):
    ops.attention(qkv, B, S, H, False)
torch.cuda.synchronize()
