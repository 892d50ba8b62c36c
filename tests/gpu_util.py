"""Helpers for the -m gpu parity tests: error metrics + a metrics log under gpurun_out/."""
import json
import os

import numpy as np
import torch

from tests.conftest import ROOT

LOG = os.path.join(ROOT, "gpurun_out", "gpu_test_metrics.jsonl")


def report(name, **metrics):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    rec = {"test": name}
    rec.update({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in metrics.items()})
    with open(LOG, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("[metric]", json.dumps(rec))


def to_np(x):
    return x.detach().float().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def rel_l2(a, b):
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_abs(a, b):
    return float(np.abs(to_np(a).astype(np.float64) - to_np(b).astype(np.float64)).max())


def min_cosine(a, b):
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    a = a.reshape(-1, a.shape[-1])
    b = b.reshape(-1, b.shape[-1])
    c = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-30)
    return float(c.min())


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)
