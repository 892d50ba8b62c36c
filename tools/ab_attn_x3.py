#!/usr/bin/env python3
"""Time keds_attention_x3 / keds_attention_f32 alone at the ViT-L/14 (B = 128, S = 257, 16 heads) and text (S = 43 / 77, 12 heads) shapes.
GPU only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib
lib = _lib.load()
def run(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, S, H, causal in ((128, 257, 16, 0), (128, 77, 12, 1), (128, 43, 12, 1)):
    d = 64 * H
    qkv = torch.randn(B * S, 3 * d, device="cuda")
    out = torch.zeros(B * S, d, device="cuda")
    plane = B * S * d
    pair = torch.zeros(2, plane, dtype=torch.float16, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    x3 = run(lambda: _lib.check(lib.keds_attention_x3(_lib.ptr(qkv), None, _lib.ptr(pair), plane, B, S, H, causal, 0, _lib.ptr(flag), _lib.stream()), "x3"))
    f32 = run(lambda: _lib.check(lib.keds_attention_f32(_lib.ptr(qkv), _lib.ptr(out), B, S, H, causal, 0, _lib.stream()), "f32"))
    fl = 4.0 * B * H * S * S * 64 / (2 if causal else 1)
    print(f"B {B} S {S} heads {H} causal {causal}: split-fp16 {x3:7.1f} us ({fl / x3 / 1e6:6.1f} TF)   f32-input MFMA {f32:7.1f} us ({fl / f32 / 1e6:6.1f} TF)", flush=True)
