#!/usr/bin/env python3
"""Experiment: one ViT-L/14 encode_image pass over B=128 on one stream vs the same 128 images as S sub-batches on S
streams (the sub-batches are independent; their kernels interleave on the CUs, so the store bursts, prologues and launch
boundaries of one stream run beside the K-loops of the other).  Prints images/s per arrangement.  GPU only."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from keds_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load()
model = bench.random_clip(dev)
B = int(os.environ.get("B", "128"))
images = torch.randn(B, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(1001), device=dev)
steps = int(os.environ.get("STEPS", "10"))


def clones(n):
    out = [model]
    for _ in range(n - 1):
        m = copy.copy(model)
        m._ws = _lib.Workspace()
        out.append(m)
    return out


def run(nstreams, side):
    lib.keds_side_lane_enable(1 if side else 0)
    ms = clones(nstreams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    per = B // nstreams
    parts = [images[i * per:(i + 1) * per].contiguous() for i in range(nstreams)]

    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for m, s, p in zip(ms, streams, parts):
            if nstreams == 1:
                outs.append(m.encode_image(p))
                continue
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(m.encode_image(p))
        if nstreams > 1:
            for s in streams:
                cur.wait_stream(s)
        return outs

    for _ in range(3):
        o = step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best, torch.cat(o)


ref = None
for ns, side in ((1, 1), (1, 0), (2, 0), (2, 1), (4, 0), (1, 1)):
    dt, out = run(ns, side)
    if ref is None:
        ref = out
    same = bool(torch.equal(out, ref))
    print(f"streams={ns} side_lane={side}: {dt * 1e3:7.2f} ms per {B} images = {B / dt:8.1f} img/s   bit-identical to 1-stream: {same}",
          flush=True)
