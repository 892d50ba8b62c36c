#!/usr/bin/env python3
"""Soak test of the two-lane tower pass: the same 128 images through ViT-L/14 N times (default 300) per precision; every
output must be bit-identical to the first (a missing fork / join between the lanes would show up as a rare mismatch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(os.environ.get("N", "300"))
dev = torch.device("cuda", 0)
model = bench.random_clip(dev)
img = torch.randn(128, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
other = torch.randn(128, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
for prec in ("bf16", "fp8"):
    model.set_precision(prec)
    ref = model.encode_image(img).clone()
    bad = 0
    for i in range(n):
        if i % 7 == 3:
            model.encode_image(other)              # interleave different data
        out = model.encode_image(img)
        bad += int(not torch.equal(out, ref))
    torch.cuda.synchronize()
    print(f"{prec}: {n} passes, mismatching passes: {bad}, finite: {bool(torch.isfinite(ref).all())}", flush=True)
    assert bad == 0
