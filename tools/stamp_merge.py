#!/usr/bin/env python3
"""Phases of merge_pairs_kernel in shader cycles (in-kernel s_memtime stamps), for the 1-GPU headline search."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keds_amd
from keds_amd import _lib
lib = _lib.load()
nq, n = 128, 500000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2002)
db = torch.nn.functional.normalize(torch.randn(n, 768, generator=gen, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn(nq, 768, generator=gen, device=dev), dim=1)
idx = keds_amd.FlatIndex(768, "l2", device=dev)
idx.add(db)
for _ in range(3):
    idx.search_device(q, 10)
buf = torch.zeros(nq * 8, dtype=torch.int64, device=dev)
lib.keds_merge_stamp_buffer(_lib.ptr(buf))
idx.search_device(q, 10)          # the buffer holds the LAST merge of the search (the candidate-pass merge)
torch.cuda.synchronize()
lib.keds_merge_stamp_buffer(None)
st = buf.cpu().reshape(nq, 8).double()
names = ["loads issued + landed, counts, wave scans", "first barrier", "pack into LDS + barrier", "radix selection + barriers       ", "collect", "ties + fillers"]
print("valid pairs per query: mean %.0f  max %.0f" % (st[:, 7].mean(), st[:, 7].max()))
for i, nme in enumerate(names):
    v = st[:, i + 1] - st[:, i]
    print(f"  {nme:46s} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f} cycles")
print(f"  {'block lifetime':46s} mean {(st[:, 6] - st[:, 0]).mean():8.0f}")
print(f"  first entry -> last exit: {(st[:, 6].max() - st[:, 0].min()):.0f} cycles")
