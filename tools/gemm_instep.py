#!/usr/bin/env python3
"""Where do the tower GEMMs lose time between a stand-alone loop and the encoder step?

The four 256 x 256 GEMMs of a ViT-L/14 block (B = 128: 32,768 full-tile rows) and the attention kernel are launched on the
tower's own buffer set (h fp16 | qkv | att | hid | two statistics buffers), with a hipEvent pair around every launch, in
these arrangements (interleaved rounds in ONE process, medians):

  alone      each GEMM back to back with itself (what tools/ab_quad.py measures: operands warm in the memory-side cache,
             nothing else between two launches)
  chain      the block's launch order qkv -> attention -> out -> fc -> proj, 24 times (every GEMM finds its A operand as the
             previous kernel left it and its weights cold: what the step does, without the side lane and without events of
             other classes)
  chain-na   the same chain without the attention launch (what the attention kernel's 270 MB of traffic displaces)
  chain+w    the chain with 24 distinct weight sets (as in the tower: weights are read once per step, 25 MB per block)
  chain:q3      out and proj both on the 4-wave kernel with the three-deep A ring (default: out on the 8-wave kernel)
  chain:nores   timing only: out and proj with a plain bias epilogue into a scratch buffer (no residual read: what the cold
                residual tile costs them)
  chain:warmh   a 67 MB read of the residual stream in front of out and proj (its own launch, not inside their event pairs)

Prints per-kernel medians for every arrangement; the side-lane share comes from rocprofv3 runs of bench.py with
KEDS_SIDE_STREAM=0/1 (tools/kstats.sh)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keds_amd import _lib  # noqa: E402
from keds_amd._lib import ptr, check, stream  # noqa: E402

lib = _lib.load()
M, MP, W_, B, S, H = 32768, 32896, 1024, 128, 257, 16   # GEMMs on the 128 full row tiles; the buffers and the attention cover all 128 x 257 rows
LAYERS = int(os.environ.get("LAYERS", "24"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))


def main():
    dev = "cuda"
    _lib.ensure_gemm_workspace(dev)
    h = (torch.randn(MP, W_, device=dev) * 0.5).half()
    qkv = torch.zeros(MP, 3 * W_, device=dev, dtype=torch.bfloat16)
    att = torch.zeros(MP, W_, device=dev, dtype=torch.bfloat16)
    hid = torch.zeros(MP, 4 * W_, device=dev, dtype=torch.bfloat16)
    st1 = torch.zeros(MP, 2, device=dev, dtype=torch.int64)
    st2 = torch.zeros(MP, 2, device=dev, dtype=torch.int64)

    def weights(n):
        out = []
        for _ in range(n):
            out.append(dict(
                qkv=(torch.randn(3 * W_, W_, device=dev) * W_ ** -0.5).half(), qkv_b=torch.randn(2 * 3 * W_, device=dev) * 0.02,
                out=(torch.randn(W_, W_, device=dev) * W_ ** -0.5 * 0.15).bfloat16(), out_b=torch.randn(W_, device=dev) * 0.02,
                fc=(torch.randn(4 * W_, W_, device=dev) * (2 * W_) ** -0.5).half(), fc_b=torch.randn(2 * 4 * W_, device=dev) * 0.02,
                proj=(torch.randn(W_, 4 * W_, device=dev) * W_ ** -0.5 * 0.15).bfloat16(), proj_b=torch.randn(W_, device=dev) * 0.02))
        return out
    wsets = weights(LAYERS)

    def reset():
        h.copy_((torch.randn(MP, W_, device=dev) * 0.5).half())
        x = h.float()
        st1[:, 0] = (x.sum(1) * 2 ** 28).long()
        st1[:, 1] = ((x * x).sum(1) * 2 ** 28).long()
        st2.copy_(st1)

    def k_qkv(w):
        check(lib.keds_gemm_bt_ex2(ptr(h), W_, ptr(w["qkv"]), ptr(w["qkv_b"]), ptr(qkv), 3 * W_, M, 3 * W_, W_,
                                   _lib.EPI_LN_BIAS_BF16_H, ptr(st1), 0, ptr(st2), stream()), "qkv")

    def k_att(w):
        check(lib.keds_attention(ptr(qkv), ptr(att), B, S, H, 0, stream()), "attention")

    resid_flag = {"v": 0}                   # keds_gemm_force_small argument for the two residual GEMMs (A/B of their kernel forms)

    def k_out(w):
        if mode["noresid"]:      # timing only: the same product with a plain bias epilogue into a scratch buffer (no residual read)
            check(lib.keds_gemm_bt_ex2(ptr(att), W_, ptr(w["out"]), ptr(w["out_b"]), ptr(dummy), W_, M, W_, W_, _lib.EPI_BIAS_BF16, None, 0,
                                       None, stream()), "out")
            return
        lib.keds_gemm_force_small(resid_flag["v"])
        check(lib.keds_gemm_bt_ex2(ptr(att), W_, ptr(w["out"]), ptr(w["out_b"]), ptr(h), W_, M, W_, W_,
                                   _lib.EPI_RESID_STATS_F16, ptr(st2), 0, None, stream()), "out")
        lib.keds_gemm_force_small(0)

    def k_fc(w):
        check(lib.keds_gemm_bt_ex2(ptr(h), W_, ptr(w["fc"]), ptr(w["fc_b"]), ptr(hid), 4 * W_, M, 4 * W_, W_,
                                   _lib.EPI_LN_QGELU_BF16_H, ptr(st2), 0, ptr(st1), stream()), "fc")

    def k_proj(w):
        if mode["noresid"]:
            check(lib.keds_gemm_bt_ex2(ptr(hid), 4 * W_, ptr(w["proj"]), ptr(w["proj_b"]), ptr(dummy), W_, M, W_, 4 * W_, _lib.EPI_BIAS_BF16,
                                       None, 0, None, stream()), "proj")
            return
        lib.keds_gemm_force_small(resid_flag["v"])
        check(lib.keds_gemm_bt_ex2(ptr(hid), 4 * W_, ptr(w["proj"]), ptr(w["proj_b"]), ptr(h), W_, M, W_, 4 * W_,
                                   _lib.EPI_RESID_STATS_F16, ptr(st1), 0, None, stream()), "proj")
        lib.keds_gemm_force_small(0)
    kern = {"qkv": k_qkv, "att": k_att, "out": k_out, "fc": k_fc, "proj": k_proj}
    order = ("qkv", "att", "out", "fc", "proj")

    sink = torch.zeros(1, device=dev)

    def timed(fn, w):
        if mode["warm"] and fn in (k_out, k_proj):     # a 67 MB read of the residual stream just before the launch: what a warm h is worth
            sink.add_(h.view(torch.int32).max().float() * 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(w)
        e1.record()
        return e0, e1

    def run_alone():
        res = {}
        for name in order:
            reset()
            ev = [timed(kern[name], wsets[0]) for _ in range(LAYERS)]
            torch.cuda.synchronize()
            res[name] = statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev[2:])
        return res

    def run_chain(skip_att, many_w):
        reset()
        ev = {n: [] for n in order}
        for l in range(LAYERS):
            w = wsets[l if many_w else 0]
            for name in order:
                if skip_att and name == "att":
                    continue
                ev[name].append(timed(kern[name], w))
        torch.cuda.synchronize()
        return {n: statistics.median(a.elapsed_time(b) * 1e3 for a, b in v[2:]) for n, v in ev.items() if v}

    def with_flag(flag, fn):
        def run():
            resid_flag["v"] = flag
            try:
                return fn()
            finally:
                resid_flag["v"] = 0
        return run
    Q3 = 1 << 11                                   # residual GEMMs on the 4-wave kernel with the three-deep A ring
    mode = {"noresid": False, "warm": False}
    dummy = torch.zeros(MP, W_, device=dev, dtype=torch.bfloat16)

    def with_mode(key, fn):
        def run():
            mode[key] = True
            try:
                return fn()
            finally:
                mode[key] = False
        return run
    arrangements = (("alone", run_alone), ("chain", lambda: run_chain(False, False)), ("chain-na", lambda: run_chain(True, False)),
                    ("chain+w", lambda: run_chain(False, True)),
                    ("chain:q3", with_flag(Q3, lambda: run_chain(False, True))),
                    ("chain:nores", with_mode("noresid", lambda: run_chain(False, True))),
                    ("chain:warmh", with_mode("warm", lambda: run_chain(False, True))))
    res = {a: {n: [] for n in order} for a, _ in arrangements}
    for a, fn in arrangements:          # warm-up of every arrangement
        fn()
    for _ in range(ROUNDS):
        for a, fn in arrangements:
            for n, v in fn().items():
                res[a][n].append(v)
    print(f"{'kernel':6s} " + " ".join(f"{a:>10s}" for a, _ in arrangements) + "    (us per launch, median of per-round medians)")
    tot = {a: 0.0 for a, _ in arrangements}
    for n in order:
        row = []
        for a, _ in arrangements:
            v = res[a][n]
            m = statistics.median(v) if v else float("nan")
            row.append(m)
            if n != "att" and v:
                tot[a] += m
        print(f"{n:6s} " + " ".join(f"{m:10.1f}" for m in row))
    print(f"{'GEMMs':6s} " + " ".join(f"{tot[a]:10.1f}" for a, _ in arrangements) + "    (sum of the four, per block)")


if __name__ == "__main__":
    main()
