#!/usr/bin/env python3
"""Headline benchmark: query-images/sec for (CLIP ViT-L/14 encode_image of B=128 synthetic 224x224
images) + (top-10 over a synthetic 0.5 M x 768 database), BASELINE.json `metric`, `configs[1]`.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch: every rank encodes its own
128 images (data parallel, weights replicated), the query embeddings are all-gathered, every rank scans
ITS shard of the database rows for all 128*N queries, and the [B,k] partial results are all-gathered
and merged keyed on (distance, id) (weak scaling: per-GPU encoder work is fixed and every rank streams
one full database's worth of bytes per step).  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VITL = dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
            context_length=77, vocab_size=49408, transformer_width=768, transformer_heads=12, transformer_layers=12)

# GEMM work EXECUTED per ViT-L/14 image (SURVEY.md App. A shapes): patch embed + 24 x qkv + 23 x (out, fc, proj) on all
# 257 tokens + the last block's (out, fc, proj) on the CLS token only (nothing else is read after the last block,
# model.py:412; the reference computes those 256 unused rows) + read-out.  The reference-equivalent count
# (all 24 blocks on 257 tokens) is 77.77 GMAC; executed: 75.35 GMAC.
_PER_TOKEN_TAIL = 1024 * 1024 + 2 * 1024 * 4096                     # out-proj + fc + proj MACs per token
GEMM_MAC_PER_IMAGE = (256 * 588 * 1024 + 24 * 257 * 1024 * 3072 + 23 * 257 * _PER_TOKEN_TAIL + 1 * _PER_TOKEN_TAIL
                      + 1024 * 768)
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0      # dense MX-fp8 MFMA (--precision fp8 only; the headline is bf16 / fp16)
PEAK_FP8_MEASURED_TFLOPS = 3400.0    # register-only v_mfma_scale_f32_16x16x128_f8f6f4 loop (profiles/r01_microbench.txt)
PEAK_BF16_MEASURED_TFLOPS = 2060.0   # tools/micro/mfma_peak.hip on the gpurun MI355X
PEAK_HBM_MEASURED_GBPS = 7150.0      # tools/micro/hbm_stream.hip (read-only)
PEAK_HBM_GBPS = 8000.0        # HBM3E spec, MI355X_MICROARCH.md
PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def pmc_traffic(kernel_key, batch, rows, world):
    """HBM bytes per launch of `kernel_key` from the committed PMC passes (tools/pmc_step.py + tools/parse_pmc.py:
    separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same workload, FETCH_SIZE doubled per the gfx950
    rule of MI355X_MICROARCH.md).  Only valid for the configuration they were taken on; otherwise None."""
    if batch != 128 or rows != 500000 or world != 1 or not os.path.exists(PMC_FILE):
        return None
    e = json.load(open(PMC_FILE)).get(kernel_key)
    if not e:
        return None
    return e.get("hbm_read_bytes_per_launch", 0.0) + e.get("hbm_write_bytes_per_launch", 0.0)


def random_clip(device):
    """Random-init ViT-L/14 CLIP of the reference architecture (there are no checkpoints offline)."""
    import keds_amd
    torch.manual_seed(1234)
    with torch.device(device):
        model = keds_amd.CLIP(**VITL)
    with torch.no_grad():
        for blk in model.visual.transformer.resblocks:        # visual tower: same stds as the text tower init
            w, L = 1024, 24
            blk.attn.in_proj_weight.normal_(std=w ** -0.5)
            blk.attn.out_proj.weight.normal_(std=w ** -0.5 * (2 * L) ** -0.5)
            blk.mlp.c_fc.weight.normal_(std=(2 * w) ** -0.5)
            blk.mlp.c_proj.weight.normal_(std=w ** -0.5 * (2 * L) ** -0.5)
    return model.eval()


def cpu_baseline(model, n_db, dim, k):
    """Oracle (CPU restatement, fp32 torch) timed on this host's cores on a bounded sample (about 20-30 s):
    2 images through ViT-L/14 and 128 queries against a 65,536-row slice, scaled to one
    query-image = 1 encode + 1 top-k over n_db rows.  The thread count is the fastest of a short pilot
    (one residual block) because oversubscribing a many-core host makes torch's CPU GEMMs slower."""
    from oracle import keds_oracle as O
    ncpu = os.cpu_count() or 1
    sd = {k_: v.detach().float().cpu() for k_, v in model.state_dict().items() if k_.startswith("visual.")}
    for k_ in ("text_projection", "positional_embedding", "token_embedding.weight", "ln_final.weight"):
        sd[k_] = model.state_dict()[k_].detach().float().cpu()   # arch inference reads their shapes only
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        x = torch.randn(2, 257, 1024)
        best_t, threads = None, 1
        for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)}):
            torch.set_num_threads(th)
            O.residual_block(x, sd, "visual.transformer.resblocks.0.", 16, False)
            t0 = time.perf_counter()
            O.residual_block(x, sd, "visual.transformer.resblocks.0.", 16, False)
            dt = time.perf_counter() - t0
            if best_t is None or dt < best_t:
                best_t, threads = dt, th
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        O.encode_image(sd, img)
        t_img = (time.perf_counter() - t0) / img.shape[0]
        rows = 65536
        db = torch.nn.functional.normalize(torch.randn(rows, dim, generator=torch.Generator().manual_seed(2)), dim=1)
        q = torch.nn.functional.normalize(torch.randn(128, dim, generator=torch.Generator().manual_seed(3)), dim=1)
        O.flat_l2_search_f32(db, q, k)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            O.flat_l2_search_f32(db, q, k)
        t_q = (time.perf_counter() - t0) / reps / q.shape[0] * (n_db / rows)
    return {"value": 1.0 / (t_img + t_q), "unit": "query-images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle fp32, {threads} of {ncpu} host threads: 2 images through ViT-L/14 ({t_img:.2f} s/image) "
                      f"+ 128 queries x {rows}-row slice scaled to {n_db} rows ({t_q * 1e3:.2f} ms/query)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--db-rows", type=int, default=500000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["bf16", "fp8"], default="bf16",
                    help="fp8 = BASELINE config 5 (MXFP8 GEMM operands in the image tower); the headline metric is bf16")
    ap.add_argument("--prof-every", type=int, default=4, help="record the per-launch hipEvents on every N-th timed step")
    ap.add_argument("--prof-all", action="store_true", help="hipEvent pairs around every kernel class (default: only the "
                    "dominant GEMM class and the scan; the full breakdown costs ~1-2 %% of the step)")
    args = ap.parse_args()
    args.prof_every = max(1, args.prof_every)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("KEDS_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank test of the RCCL path
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)         # "nccl" is RCCL on ROCm

    import keds_amd
    from keds_amd import _lib
    from keds_amd.index import PackedExchange, shard_bounds
    _lib.load()                                                # fail loudly if the HIP library is missing

    B, N, D, k = args.batch, args.db_rows, 768, args.k
    model = random_clip(dev)
    if args.precision == "fp8":
        model.set_precision("fp8")
    # synthetic database: seeded unit-norm rows; this rank keeps rows [lo, hi)
    lo, hi = shard_bounds(N, world, rank)
    gen = torch.Generator(device=dev).manual_seed(2002)
    index = keds_amd.FlatIndex(D, "l2", device=dev, row0=lo)
    chunk = 65536
    parts = []
    for s in range(0, N, chunk):                               # same global stream on every rank, keep own rows
        blk = torch.randn(min(chunk, N - s), D, generator=gen, device=dev)
        a, b = max(lo, s), min(hi, s + blk.shape[0])
        if a < b:
            parts.append(torch.nn.functional.normalize(blk[a - s:b - s], dim=1))
    index.add(torch.cat(parts))
    del parts
    images = torch.randn(B, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(1001 + rank), device=dev)

    # The search of batch i runs on its own stream behind batch i's encoder pass, beside batch i+1's: its two KB-sized,
    # latency-bound collectives (one packed all-gather of the queries, one packed all-to-all of the partial lists;
    # preallocated buffers, keds_amd.index.PackedExchange) and its short merge / re-rank launches hide under the next
    # encoder pass instead of standing between two of them.
    # (With one GPU there is no collective to hide, and a scan that shares the CUs with the next encoder pass reads at a lower
    # rate for no gain: the search then stays on the encoder's stream.)
    xchg = PackedExchange() if use_dist else None
    search_stream = torch.cuda.Stream(device=dev) if use_dist else torch.cuda.current_stream()
    comm_events = []                                            # (gather, scan, return) hipEvent quadruples of profiled steps
    search_events = []                                          # hipEvent pairs around the whole local search of profiled steps

    def step(timed=False):
        q = model.encode_image(images, normalize=True)          # [B,768] on device
        if use_dist:
            search_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(search_stream):
            if use_dist:
                q.record_stream(search_stream)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if (timed and use_dist) else None
            if ev:
                ev[0].record()
            allq = xchg.gather_queries(q) if use_dist else q
            if ev:
                ev[1].record()
            sev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if timed else None
            if sev:
                sev[0].record()
            Dk, Ik, _ = index.search_device(allq, k)
            if sev:
                sev[1].record()
                search_events.append(sev)
            if ev:
                ev[2].record()
            if use_dist:
                Dk, Ik = xchg.return_partials(Dk, Ik, index.metric)
            if ev:
                ev[3].record()
                comm_events.append(ev)
        return Dk, Ik

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.prof_reset()
    # hipEvent pair around every launch of the dominant kernel class (GEMM) and of the scan, on the launch stream, inside
    # the timed region -- on every `--prof-every`-th timed step (default 4: steps 0, 4, 8, ...).  An event pair costs ~3.5 us
    # of stream time; around all 100 launches of every step that is 2.7 % of the step, sampled it is 0.7 %.
    classes = None if args.prof_all else (_lib.PROF_GEMM, _lib.PROF_SCAN)
    events = os.environ.get("KEDS_BENCH_NO_EVENTS") != "1"      # A/B of the event overhead only
    prof_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = events and i % args.prof_every == 0
        if on:
            _lib.prof_enable(True, classes)
            prof_steps += 1
        Dk, Ik = step(timed=on)
        if on:
            _lib.prof_enable(False)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    per_rank = None
    if use_dist:                                                # what each rank's search costs, so a scaling run explains itself
        mine = [sum(e[i].elapsed_time(e[i + 1]) for e in comm_events) / max(len(comm_events), 1) * 1e3 for i in range(3)]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = {"query_allgather_us": [round(g[0], 1) for g in gathered],
                    "local_search_us": [round(g[1], 1) for g in gathered],
                    "partials_alltoall_merge_us": [round(g[2], 1) for g in gathered]}
    gemm_ms, gemm_n = _lib.prof_read(_lib.PROF_GEMM)
    scan_ms, scan_n = _lib.prof_read(_lib.PROF_SCAN)
    attn_ms, attn_n = _lib.prof_read(_lib.PROF_ATTN)
    ln_ms, ln_n = _lib.prof_read(_lib.PROF_LN)
    other_ms, other_n = _lib.prof_read(_lib.PROF_OTHER)

    if rank == 0:
        steps = args.steps
        psteps = max(prof_steps, 1)                              # timed steps whose launches carried event pairs
        gemm_flops = 2.0 * GEMM_MAC_PER_IMAGE * B * psteps     # this rank's GEMM launches on those steps
        # rows the towers run on the side lane (B*257 mod 256 = 128 of 32,896): those launches overlap the main stream and
        # carry no event pairs, so their flops leave the numerator as well
        side_rows = _lib.load().keds_tower_side_rows(VITL["vision_width"], 257, B, int(args.precision == "fp8"))
        if side_rows:
            tower_mac = 24 * 257 * 1024 * 3072 + 23 * 257 * _PER_TOKEN_TAIL
            gemm_flops -= 2.0 * tower_mac * B * psteps * side_rows / (B * 257.0)
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        fp8 = args.precision == "fp8"
        peak_tf = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS              # dense MFMA peak of the operand type
        peak_meas_tf = PEAK_FP8_MEASURED_TFLOPS if fp8 else PEAK_BF16_MEASURED_TFLOPS
        # algorithmic bytes of one search = one pass over this rank's bf16 rows (N_local*D*2 B); a search issues two
        # scan launches (threshold pass over the first 1/16 of the rows + the full candidate pass): both are charged
        # to the time, only the single pass to the bytes
        n_search = psteps * world                                # query blocks of 128 searched by this rank (profiled steps)
        scan_bytes = (hi - lo) * D * 2.0 * n_search
        scan_ach = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        # the WHOLE local search (query preparation, both scan passes, merges, re-rank, certificate + select, the exact-pass
        # launches) against the same algorithmic bytes: what a caller of index.search pays per query block
        whole_ms = sum(a.elapsed_time(b2) for a, b2 in search_events) / max(len(search_events), 1)
        whole_ach = (hi - lo) * D * 2.0 * world / (whole_ms * 1e-3) / 1e9 if whole_ms > 0 else 0.0
        out = {
            "metric": "query-images/sec (encode+0.5M top-10) ViT-L/14",
            "value": world * B * steps / elapsed,
            "unit": "query-images/sec",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16/fp16 operands, fp32 accumulate" if args.precision == "bf16" else "fp8 (MXFP8 e4m3 operands, fp32 accumulate; config 5)", "data": "synthetic",
            "config": {"workload": "ViT-L/14 encode_image (224x224 synthetic, random-init weights) + exact top-10 "
                                   "over a synthetic unit-norm 0.5M x 768 database",
                       "batch_per_gpu": B, "global_batch": B * world, "db_rows": N, "dim": D, "k": k,
                       "db_shards": world, "parallelism": f"dp{world} encoders + {world}-way row-sharded scan"},
            "roofline": {"kernel": ("gemm_mxfp8_kernel" if fp8 else "gemm_bt_pair_kernel") + " (256x256 tiles; all main-lane ViT GEMM launches of the step incl. the few 128x128-tile ones)", "bound": "mfma",
                         "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf,
                         "traffic": None if fp8 else pmc_traffic("gemm_bt_pair_kernel", B, N, world), "traffic_unit": "bytes/launch (PMC, mean "
                         "over the 256x256 GEMM launches)", "launches": gemm_n, "avg_launch_ms": gemm_ms / max(gemm_n, 1),
                         # practical ceiling of this chip, measured (profiles/r01_microbench.txt): a register-only
                         # v_mfma_f32_16x16x32_bf16 loop sustains 2.06 PFLOP/s (clock ~2.0 GHz under MFMA load)
                         "peak_measured": peak_meas_tf, "frac_of_measured": ach / peak_meas_tf},
            "roofline_scan": {"kernel": "scan_topk_kernel", "bound": "hbm", "achieved": scan_ach,
                              "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": scan_ach / PEAK_HBM_GBPS,
                              "traffic": pmc_traffic("scan_topk_kernel<768, 16", B, N, world),
                              "algorithmic_bytes_per_search": (hi - lo) * D * 2.0,
                              "launches": scan_n, "searches": n_search, "ms_per_search": scan_ms / max(n_search, 1),
                              "whole_search": {"ms": whole_ms, "query_blocks": world, "achieved": whole_ach,
                                               "frac": whole_ach / PEAK_HBM_GBPS},
                              # read-only stream over 4 GiB on this box: 7.15 TB/s (profiles/r01_microbench.txt)
                              "peak_measured": PEAK_HBM_MEASURED_GBPS, "frac_of_measured": scan_ach / PEAK_HBM_MEASURED_GBPS},
            "profiled_steps": prof_steps, "side_lane_rows": side_rows,
            "search_overlap": ("search of batch i on a second stream beside the encoder pass of batch i+1" if use_dist
                               else "none (one GPU: no collective to hide)"),
            "per_rank_search": per_rank,
            "stage_ms_per_step": {"gemm": gemm_ms / psteps, "attention": attn_ms / psteps, "layernorm": ln_ms / psteps,
                                  "scan": scan_ms / psteps, "other": other_ms / psteps},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, N, D, k)
        try:                                   # RCCL writes its version banner through C stdio, which flushes at exit:
            import ctypes                      # push it out now so that the JSON line is the last line on stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
