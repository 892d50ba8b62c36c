#!/usr/bin/env python3
"""Timing-only ablations of the candidate scan (keds_scan_debug CODE = 1: no list update, 2: no MFMA, 3: no LDS fragment reads) at the
shard shape (CONFIG=shard: 1,024 queries x 62.5 k rows) or the headline shape (CONFIG=one).  Run under tools/kstats_cmd.sh and read the
scan_topk_kernel<768, 16, CODE> line: the ablated scans produce wrong lists, so every certificate fails and the whole search takes
the exact pass -- only the kernel's own duration means anything."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keds_amd
from keds_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = os.environ.get("CONFIG", "shard")
nq, n = (1024, 62500) if cfg == "shard" else (128, 500000)
gen = torch.Generator(device=dev).manual_seed(2002)
db = torch.nn.functional.normalize(torch.randn(n, 768, generator=gen, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn(nq, 768, generator=gen, device=dev), dim=1)
idx = keds_amd.FlatIndex(768, "l2", device=dev)
idx.add(db)
lib.keds_scan_debug(int(os.environ.get("CODE", "0")))
for _ in range(int(os.environ.get("ITERS", "6"))):
    idx.search_device(q, 10)
torch.cuda.synchronize()
lib.keds_scan_debug(0)
