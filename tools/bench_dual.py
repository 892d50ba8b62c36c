#!/usr/bin/env python3
"""Dual-stream composed-query throughput on one GPU (BASELINE config 4 at 1 GPU): per 128-query batch
ViT-L/14 encode_image + top-16 with row gather over TWO 0.5 M x 768 databases + 2 knowledge streams
(IM2TEXT + 2 x CrossFormer) + 2 text-tower passes with pseudo-token splice + normalise/mixture.
Random-init weights, synthetic inputs.  Prints queries/s and the hipEvent stage breakdown."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, keds_amd
from keds_amd import _lib
from oracle import keds_oracle as O

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, N, D = 128, int(os.environ.get("N", "500000")), 768
model = bench.random_clip(dev)
gen = torch.Generator(device=dev).manual_seed(2002)
ib = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device=dev), dim=1)
tb = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device=dev), dim=1)
database = keds_amd.build_database(ib, tb, None, device=dev)
del ib, tb
def stream(seed):
    torch.manual_seed(seed)
    a, b, c = keds_amd.make_stream_modules(model, middle_dim=512, n_layer=2, device=dev)
    return keds_amd.KnowledgeStream(a, b, c)
s_img, s_txt = stream(1), stream(2)
images = torch.randn(B, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(1001), device=dev)
tokens = O.synth_tokens(B).to(dev)
steps, warm = int(os.environ.get("STEPS", "10")), 3
def step():
    return keds_amd.compose_query_features(model, s_img, s_txt, images, tokens, database, id_split=265, verify=False)
for _ in range(warm):
    step()
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
t0 = time.perf_counter()
for _ in range(steps):
    out = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.prof_enable(False)
names = ["gemm", "attention", "scan", "layernorm", "other"]
stages = {n: _lib.prof_read(i)[0] / steps for i, n in enumerate(names)}
launches = {n: _lib.prof_read(i)[1] / steps for i, n in enumerate(names)}
print(json.dumps({"metric": "composed dual-stream queries/sec (1 GPU)", "value": B * steps / dt, "ms_per_batch": dt / steps * 1e3,
                  "stage_ms": stages, "launches_per_batch": launches,
                  "finite": bool(torch.isfinite(out["mixture"]).all())}))
