"""Thin torch-tensor wrappers over the primitive entry points of libkeds_hip.so.

Used by the parity tests (one kernel at a time) and by the facade classes.  Every
function launches on torch's current stream and returns device tensors; nothing
here computes on the CPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check, load, ptr, stream


def _pad128(m: int) -> int:
    return (m + 127) // 128 * 128


def cast_bf16(x: torch.Tensor, rows_padded: Optional[int] = None) -> torch.Tensor:
    """fp32 [..., d] -> bf16 copy made by the HIP cast kernel; optional zero row padding of a 2-D input."""
    x = x.contiguous().float()
    if rows_padded is not None:
        out = torch.zeros((rows_padded, x.shape[-1]), dtype=torch.bfloat16, device=x.device)
    else:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(load().keds_cast_bf16(ptr(x), ptr(out), x.numel(), stream()), "keds_cast_bf16")
    return out


def gemm_bt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int,
            out: Optional[torch.Tensor] = None, m: Optional[int] = None,
            aux: Optional[torch.Tensor] = None, aux_i: int = 0) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  `a` must be bf16 with rows padded to 128."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    M = a.shape[0] if m is None else m
    N, K = w.shape
    if a.shape[0] < _pad128(M):
        raise ValueError("A must have its rows padded to a multiple of 128")
    if out is None:
        odt = torch.float32 if epilogue in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_BIAS_F32, _lib.EPI_PATCH_F32) \
            else torch.bfloat16
        out = torch.zeros((_pad128(M), N), dtype=odt, device=a.device)
    _lib.ensure_gemm_workspace(a.device)
    check(load().keds_gemm_bt(ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K, epilogue, ptr(aux), aux_i, stream()),
          "keds_gemm_bt")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_f32: bool = False) -> torch.Tensor:
    rows, dim = x.shape
    out = torch.empty((rows, dim), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    check(load().keds_layernorm(ptr(x), dim, ptr(gamma), ptr(beta), ptr(out), 1 if out_f32 else 0, rows, dim, stream()),
          "keds_layernorm")
    return out


def attention(qkv: torch.Tensor, B: int, S: int, heads: int, causal: bool) -> torch.Tensor:
    """qkv bf16 [B*S, 3*heads*64] -> bf16 [B*S, heads*64]"""
    assert qkv.dtype == torch.bfloat16
    out = torch.empty((B * S, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    check(load().keds_attention(ptr(qkv), ptr(out), B, S, heads, 1 if causal else 0, stream()), "keds_attention")
    return out


def im2col(image: torch.Tensor, patch: int, kpad: int) -> torch.Tensor:
    B, _, R, _ = image.shape
    g = R // patch
    out = torch.zeros((_pad128(B * g * g), kpad), dtype=torch.bfloat16, device=image.device)
    image = image.contiguous().float()           # a named reference: a temporary would be freed (and its block reused) before launch
    check(load().keds_im2col(ptr(image), ptr(out), B, R, patch, kpad, stream()), "keds_im2col")
    return out


_PIL_COEFFS = {}


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """PIL's bicubic resampling weights for one axis, as PIL computes and quantises them (Pillow src/libImaging/Resample.c:
    precompute_coeffs with the bicubic filter a = -0.5, antialiasing support 2 * max(scale, 1), weights normalised to 1 in
    float64, then normalize_coeffs_8bpc: round-half-away to 22 fractional bits).  Returns (bounds int32 [out, 2] =
    {first source index, taps}, kk int32 [out, ksize]).  Pure host integer / float64 work (cached per size pair); pinned to
    PIL bit for bit in tests/test_clip_host.py."""
    import math
    key = (int(in_size), int(out_size))
    hit = _PIL_COEFFS.get(key)
    if hit is not None:
        return hit
    import numpy as np
    a = -0.5

    def bicubic(x: float) -> float:
        if x < 0.0:
            x = -x
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            w = bicubic((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + k[x] * (1 << 22)) if k[x] < 0 else int(0.5 + k[x] * (1 << 22))
    _PIL_COEFFS[key] = (bounds, kk)
    return bounds, kk


def resize_crop_geometry(H: int, W: int, n_px: int):
    """torchvision Resize(int) + CenterCrop(int) as the reference's `_transform` applies them (src/model/clip.py:107-123):
    the shorter side becomes n_px, the other int(n_px * long / short); crop offsets int(round((size - n_px) / 2.0))."""
    if W <= H:
        RW, RH = n_px, int(n_px * H / W)
    else:
        RW, RH = int(n_px * W / H), n_px
    if RW < n_px or RH < n_px:
        raise ValueError("resized image smaller than the crop")
    return RW, RH, int(round((RW - n_px) / 2.0)), int(round((RH - n_px) / 2.0))


def preprocess(images_u8: torch.Tensor, n_px: int, mean=(0.48145466, 0.4578275, 0.40821073),
               std=(0.26862954, 0.26130258, 0.27577711), return_u8: bool = False):
    """uint8 [B,H,W,3] (one size per batch, device) -> fp32 [B,3,n_px,n_px]: the eval `_transform` of the reference
    (bicubic resize of the shorter side, centre crop, ToTensor, Normalize) in one kernel, BIT-EXACT with the PIL pipeline:
    PIL's own 22-bit integer filter weights (host, float64, cached per image size), integer accumulation, the horizontal
    pass rounded to uint8 before the vertical one.  return_u8: also the uint8 image [B,n_px,n_px,3] PIL hands to ToTensor."""
    import ctypes as C
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise ValueError("expected uint8 images [B,H,W,3]")
    images_u8 = images_u8.contiguous()
    B, H, W, _ = images_u8.shape
    dev = images_u8.device
    RW, RH, left, top = resize_crop_geometry(H, W, n_px)
    need_h, need_v = RW != W, RH != H                     # PIL skips a pass that keeps the size
    tabs = {}
    for name, need, in_size, out_size, off in (("x", need_h, W, RW, left), ("y", need_v, H, RH, top)):
        if need:
            b, k = pil_bicubic_coeffs(in_size, out_size)
            tabs[name] = (torch.from_numpy(b[off:off + n_px].copy()).to(dev), torch.from_numpy(k[off:off + n_px].copy()).to(dev),
                          k.shape[1])
        else:
            tabs[name] = (None, None, 0)
    out = torch.empty((B, 3, n_px, n_px), dtype=torch.float32, device=dev)
    u8 = torch.empty((B, n_px, n_px, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    (xb, xk, ksx), (yb, yk, ksy) = tabs["x"], tabs["y"]
    check(load().keds_preprocess_pil(ptr(images_u8), B, H, W, n_px, int(need_h), int(need_v), left, top, ptr(xb), ptr(xk), ksx,
                                     ptr(yb), ptr(yk), ksy, (C.c_float * 3)(*mean), (C.c_float * 3)(*std), ptr(out), ptr(u8),
                                     stream()), "keds_preprocess_pil")
    return (out, u8) if return_u8 else out


def embed_tokens(tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor,
                 img_tokens: Optional[torch.Tensor] = None, insert_col: int = 0) -> torch.Tensor:
    B, L = tokens.shape
    d = table.shape[1]
    x = torch.empty((B, L, d), dtype=torch.float32, device=table.device)
    n_tok = 0 if img_tokens is None else img_tokens.shape[1]
    tok32 = tokens.to(torch.int32).contiguous()
    it32 = None if img_tokens is None else img_tokens.contiguous().float()
    check(load().keds_embed_tokens(ptr(tok32), ptr(table), ptr(pos), ptr(it32),
                                   n_tok, insert_col, ptr(x), B, L, d, stream()), "keds_embed_tokens")
    return x


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous().float()
    out = torch.empty_like(x)
    check(load().keds_l2_normalize(ptr(x), ptr(out), x.shape[0], x.shape[1], stream()), "keds_l2_normalize")
    return out


def mix_normalize(a: torch.Tensor, b: torch.Tensor, wa: float = 0.5, wb: float = 0.5):
    """(normalize(a), normalize(b), normalize(wa*normalize(a) + wb*normalize(b)))  -- eval_utils.py:704-710"""
    a = a.contiguous().float()
    b = b.contiguous().float()
    an, bn, mix = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    check(load().keds_mix_normalize(ptr(a), ptr(b), wa, wb, ptr(an), ptr(bn), ptr(mix), a.shape[0], a.shape[1],
                                    stream()), "keds_mix_normalize")
    return an, bn, mix


def gather_rows(db: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    flat = idx.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty((flat.numel(), db.shape[1]), dtype=torch.float32, device=db.device)
    check(load().keds_gather_rows(ptr(db), db.shape[1], ptr(flat), flat.numel(), ptr(out), stream()), "keds_gather_rows")
    return out.reshape(*idx.shape, db.shape[1])


def rank_gallery(ref: torch.Tensor, gallery: torch.Tensor) -> torch.Tensor:
    """order[q,:] = stable argsort(1 - ref[q] @ gallery.T) as int32 [Q,G]  (eval_utils.py:1042-1043)"""
    if not ref.is_cuda or not gallery.is_cuda:          # the eval drivers sometimes hand over features they moved to the CPU
        _lib.require_gpu()
        dev = ref.device if ref.is_cuda else (gallery.device if gallery.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        ref, gallery = ref.to(dev), gallery.to(dev)
    ref = ref.contiguous().float()
    gallery = gallery.contiguous().float()
    nq, ng = ref.shape[0], gallery.shape[0]
    lib = load()
    nbytes = lib.keds_rank_gallery_workspace_bytes(nq, ng)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
    order = torch.empty((nq, ng), dtype=torch.int32, device=ref.device)
    check(lib.keds_rank_gallery(ptr(ref), nq, ptr(gallery), ng, ref.shape[1], ptr(order), ptr(ws), nbytes, stream()),
          "keds_rank_gallery")
    return order


def cirr_target_rank(order: torch.Tensor, gallery_ids: torch.Tensor, ref_ids: torch.Tensor, target_ids: torch.Tensor):
    nq, ng = order.shape
    rank = torch.empty(nq, dtype=torch.int32, device=order.device)
    counts = torch.empty((nq, 2), dtype=torch.int32, device=order.device)
    # named references keep the converted copies alive until the launch is enqueued (a temporary passed straight to ptr()
    # is freed at once and the allocator may hand its block to the next conversion)
    gid, rid, tid = (t.to(order.device, torch.int32).contiguous() for t in (gallery_ids, ref_ids, target_ids))
    check(load().keds_cirr_target_rank(ptr(order), nq, ng, ptr(gid), ptr(rid), ptr(tid), ptr(rank), ptr(counts), stream()),
          "keds_cirr_target_rank")
    return rank, counts


def label_hits(order: torch.Tensor, gallery_labels: torch.Tensor, query_labels: torch.Tensor, ks):
    """(hits [Q,len(ks)], total [Q]) int32: same-label items among the first k of every ranking / in the gallery."""
    import ctypes as C
    nq, ng = order.shape
    hits = torch.empty((nq, len(ks)), dtype=torch.int32, device=order.device)
    total = torch.empty(nq, dtype=torch.int32, device=order.device)
    karr = (C.c_int * len(ks))(*[int(k) for k in ks])
    gl = gallery_labels.to(order.device, torch.int32).contiguous()       # named: see cirr_target_rank
    ql = query_labels.to(order.device, torch.int32).contiguous()
    check(load().keds_label_hits(ptr(order), nq, ng, ptr(gl), ptr(ql), karr, len(ks), ptr(hits),
                                 ptr(total), stream()), "keds_label_hits")
    return hits, total


def topk_merge_parts(D_parts: torch.Tensor, I_parts: torch.Tensor, metric: int):
    """[parts, nq, k] sorted partial results -> global (D [nq,k], I [nq,k]) keyed on (D, I)."""
    parts, nq, k = D_parts.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
    D_parts, I_parts = D_parts.contiguous().float(), I_parts.contiguous().to(torch.int64)
    check(load().keds_topk_merge_parts(ptr(D_parts), ptr(I_parts),
                                       parts, nq, k, metric, ptr(D), ptr(I), stream()), "keds_topk_merge_parts")
    return D, I


def exchange_pack(D_p: torch.Tensor, I_p: torch.Tensor, rows_p: Optional[torch.Tensor], world: int, send: torch.Tensor,
                  w_stride: int) -> None:
    """Partial lists (+ rows) of a sharded search -> one int32 message per peer (keds_hip.h, keds_exchange_pack).
    `send` may be a view into a buffer shared by several databases (w_stride = words between consecutive parts)."""
    n, k = D_p.shape
    dim = 0 if rows_p is None else rows_p.shape[2]
    check(load().keds_exchange_pack(ptr(D_p), ptr(I_p), ptr(rows_p), world, n // world, k, dim, int(w_stride), ptr(send),
                                    stream()), "keds_exchange_pack")


def exchange_merge(recv: torch.Tensor, world: int, B: int, k: int, dim: int, w_stride: int, metric: int, with_rows: bool):
    """Received parts -> (D [B,k], I [B,k], rows [B,k,dim] or None) keyed on (distance, id) (keds_exchange_merge)."""
    D = torch.empty((B, k), dtype=torch.float32, device=recv.device)
    I = torch.empty((B, k), dtype=torch.int64, device=recv.device)
    rows = torch.empty((B, k, dim), dtype=torch.float32, device=recv.device) if with_rows else None
    check(load().keds_exchange_merge(ptr(recv), world, B, k, dim if with_rows else 0, int(w_stride), metric, ptr(D), ptr(I),
                                     ptr(rows), stream()), "keds_exchange_merge")
    return D, I, rows
