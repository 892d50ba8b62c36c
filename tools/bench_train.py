#!/usr/bin/env python3
"""Training-step throughput on one GPU: B = 128 precomputed image features, top-16 retrieval over two
0.5 M x 768 databases, IM2TEXT + 2 x CrossFormer forward / backward through the frozen ViT-L/14-size text tower, loss,
AdamW.  Random-init weights, synthetic inputs.  Prints samples/s and ms per step."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, keds_amd
from keds_amd.train import KnowledgeTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, N, D = 128, int(os.environ.get("N", "500000")), 768
model = bench.random_clip(dev)
gen = torch.Generator(device=dev).manual_seed(2002)
ib = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device=dev), dim=1)
tb = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device=dev), dim=1)
database = keds_amd.build_database(ib, tb, None, device=dev)
del ib, tb
torch.manual_seed(1)
a, b, c = keds_amd.make_stream_modules(model, middle_dim=512, n_layer=2, device=dev)
tr = KnowledgeTrainer(model, a, b, c, lr=1e-4, wd=0.1, dropout=0.1)
feats = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device=dev), dim=1)
prompt = torch.zeros(77, dtype=torch.int64)
prompt[:6] = torch.tensor([49406, 320, 1125, 539, 265, 49407])           # "<sot> a photo of * <eot>"
steps = int(os.environ.get("STEPS", "10"))
losses = []
for _ in range(3):
    tr.step(feats, database, prompt, 265)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    losses.append(tr.step(feats, database, prompt, 265))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"metric": "training samples/sec (1 GPU, B=128, text-tower backward + 3 modules + AdamW)", "value": B / dt,
                  "ms_per_step": dt * 1e3, "loss_first": float(losses[0]), "loss_last": float(losses[-1])}))
