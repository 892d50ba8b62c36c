#!/usr/bin/env python3
"""Two bench-identical steps (B=128 ViT-L/14 encode + 0.5M x 768 top-10) for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, keds_amd
dev = torch.device("cuda", 0)
model = bench.random_clip(dev)
gen = torch.Generator(device=dev).manual_seed(2002)
db = torch.nn.functional.normalize(torch.randn(500000, 768, generator=gen, device=dev), dim=1)
index = keds_amd.FlatIndex(768, "l2", device=dev)
index.add(db)
images = torch.randn(128, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(1001), device=dev)
qr = torch.nn.functional.normalize(torch.randn(128, 768, generator=torch.Generator(device=dev).manual_seed(3003), device=dev), dim=1)
for _ in range(int(os.environ.get("STEPS", "2"))):
    q = model.encode_image(images, normalize=True)
    index.search_device(q, 10)
    index.search_device(qr, 10)          # independent random queries: the data-robust case for the scan
torch.cuda.synchronize()
