#!/usr/bin/env python3
"""Clock and power of the chip while one GEMM kernel runs back to back (rocm-smi polled from a thread): the vendor BLAS
(reference point only) against our 8-wave and 4-wave 256^2 kernels on the c_proj / c_fc shapes.  Says whether the kernels
sit at the power cap and what shader clock each holds there."""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keds_amd import _lib, ops
lib = _lib.load()

def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            out.append(d)
        except Exception as e:
            out.append({"err": str(e)})
        time.sleep(0.25)

def summarize(samples):
    sclk, pw = [], []
    for s in samples:
        for card, v in s.items():
            if not isinstance(v, dict):
                continue
            for k, x in v.items():
                if "sclk" in k.lower():
                    try: sclk.append(float(str(x).strip("()Mhz ")))
                    except Exception: pass
                if "power" in k.lower() and "(w)" in k.lower():
                    try: pw.append(float(x))
                    except Exception: pass
    return sclk, pw

M = 32768
for N, K, tag in [(1024, 4096, "proj"), (4096, 1024, "fc")]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    cases = [("vendor", lambda: torch.matmul(a, w.t()), 0), ("8 waves", lambda: ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M), 0),
             ("4 waves", lambda: ops.gemm_bt(a, w, bias, _lib.EPI_BIAS_BF16, out=out, m=M), 1 << 11)]
    for name, fn, flag in cases:
        lib.keds_gemm_force_small(flag)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=poll, args=(stop, samples)); th.start()
        t0 = time.perf_counter(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < 4.0:
            for _ in range(50): fn()
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        us = e0.elapsed_time(e1) / n * 1e3
        sclk, pw = summarize(samples)
        raw = samples[len(samples) // 2] if samples else None
        print(f"{tag:5s} {name:8s} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF  sclk MHz {sorted(sclk)[len(sclk)//2] if sclk else None} (n={len(sclk)})  power W {sorted(pw)[len(pw)//2] if pw else None}", flush=True)
        if name == "vendor" and tag == "proj":
            print("   sample:", json.dumps(raw)[:600], flush=True)
    lib.keds_gemm_force_small(0)
