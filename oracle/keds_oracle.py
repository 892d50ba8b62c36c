"""CPU oracle for the KEDs retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU fp32 restatement of the reference algorithm for
the hot path named in BASELINE.json (CLIP ViT-L/14 image/text encoders, the
dual-stream knowledge injection, the brute-force L2 top-k search and the CIRR
recall metric).  It exists so the HIP path can be checked; it is never the
product.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `keds_amd/` imports it.

Parity pin: every function here is checked in `tests/test_oracle_golden.py`
against golden vectors minted by running the reference's own Python
(`/root/reference/src/model/model.py`, `src/eval_utils.py`) in the build
container -- see `tools/mint_golden.py` and `tests/golden/`.  The one piece of
arithmetic that is NOT in the reference tree is Faiss `IndexFlatL2` (pinned only
as faiss-gpu=1.4.0 in src/third_party/open_clip/environment.yml:27); it is exact
brute force by definition, and the reference's own torch statement of the same
search (src/trainer.py:246-257) is what `flat_l2_search` is pinned against.

All functions take a flat ``state_dict``-style mapping of fp32 tensors using
the reference's key names; there are no nn.Modules here
on purpose, so this file cannot be mistaken for (or drift into) a copy of the
reference's module code.

Citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

LN_EPS = 1e-5  # torch.nn.LayerNorm default used by src/model/model.py:291-297


# ----------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """LayerNorm over the last dim, statistics in fp32 (src/model/model.py:291-297)."""
    x = x.float()
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * w.float() + b.float()


def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x)  (src/model/model.py:300-302)."""
    return x / (1.0 + torch.exp(-1.702 * x))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.float().t()
    return y if b is None else y + b.float()


def self_attention(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
                   heads: int, causal: bool) -> Tensor:
    """nn.MultiheadAttention(d, heads)(x, x, x, attn_mask) for batch-first x [B,S,d].

    Packed in-projection, scale 1/sqrt(dh), additive -inf causal mask for the
    text tower (src/model/model.py:309,319-321,543-549).
    """
    B, S, d = x.shape
    dh = d // heads
    qkv = linear(x, in_w, in_b)                                  # [B,S,3d]
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(B, S, heads, dh).transpose(1, 2)                # [B,H,S,dh]
    k = k.reshape(B, S, heads, dh).transpose(1, 2)
    v = v.reshape(B, S, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))         # [B,H,S,S]
    if causal:
        future = torch.ones(S, S, dtype=torch.bool).triu(1)
        s = s.masked_fill(future, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, S, d)
    return linear(o, out_w, out_b)


def residual_block(x: Tensor, sd: SD, pfx: str, heads: int, causal: bool) -> Tensor:
    """x += attn(ln_1(x)); x += c_proj(qgelu(c_fc(ln_2(x))))  (src/model/model.py:305-326)."""
    h = layer_norm(x, sd[pfx + "ln_1.weight"], sd[pfx + "ln_1.bias"])
    x = x + self_attention(h, sd[pfx + "attn.in_proj_weight"], sd[pfx + "attn.in_proj_bias"],
                           sd[pfx + "attn.out_proj.weight"], sd[pfx + "attn.out_proj.bias"],
                           heads, causal)
    h = layer_norm(x, sd[pfx + "ln_2.weight"], sd[pfx + "ln_2.bias"])
    h = quick_gelu(linear(h, sd[pfx + "mlp.c_fc.weight"], sd[pfx + "mlp.c_fc.bias"]))
    return x + linear(h, sd[pfx + "mlp.c_proj.weight"], sd[pfx + "mlp.c_proj.bias"])


def _count_blocks(sd: SD, pfx: str) -> int:
    n = 0
    while (pfx + f"resblocks.{n}.ln_1.weight") in sd:
        n += 1
    return n


def transformer(x: Tensor, sd: SD, pfx: str, heads: int, causal: bool,
                collect: Optional[list] = None) -> Tensor:
    """Plain branch of Transformer.forward (src/model/model.py:372-373)."""
    for i in range(_count_blocks(sd, pfx)):
        x = residual_block(x, sd, pfx + f"resblocks.{i}.", heads, causal)
        if collect is not None:
            collect.append(x)
    return x


# ----------------------------------------------------------------------------
# CLIP towers
# ----------------------------------------------------------------------------
def arch_from_state_dict(sd: SD) -> Dict[str, int]:
    """Shape inference of build_model (src/model/model.py:951-975), ViT branch only."""
    vw = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = sd["ln_final.weight"].shape[0]
    return dict(
        embed_dim=sd["text_projection"].shape[1],
        image_resolution=patch * grid,
        vision_layers=_count_blocks(sd, "visual.transformer."),
        vision_width=vw, vision_patch_size=patch, vision_heads=vw // 64,
        context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0],
        transformer_width=tw, transformer_heads=tw // 64,
        transformer_layers=_count_blocks(sd, "transformer."),
    )


def patch_embed(sd: SD, image: Tensor) -> Tensor:
    """conv1 (stride = kernel = patch, no bias) as an im2col matmul, CLS + pos-emb
    (src/model/model.py:381,394-398).  Returns [B, 1+grid^2, width]."""
    w = sd["visual.conv1.weight"].float()                  # [width,3,P,P]
    width, _, P, _ = w.shape
    B, C, H, W = image.shape
    gy, gx = H // P, W // P
    cols = image.float().reshape(B, C, gy, P, gx, P).permute(0, 2, 4, 1, 3, 5)   # [B,gy,gx,C,P,P]
    cols = cols.reshape(B, gy * gx, C * P * P)
    tok = cols @ w.reshape(width, -1).t()                                          # [B,G,width]
    cls = sd["visual.class_embedding"].float().expand(B, 1, width)
    return torch.cat([cls, tok], dim=1) + sd["visual.positional_embedding"].float()


def encode_image(sd: SD, image: Tensor, collect: Optional[list] = None) -> Tensor:
    """CLIP.encode_image -> VisualTransformer.forward (src/model/model.py:569-575,393-415)."""
    a = arch_from_state_dict(sd)
    x = patch_embed(sd, image)
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = transformer(x, sd, "visual.transformer.", a["vision_heads"], causal=False, collect=collect)
    cls = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return cls @ sd["visual.proj"].float()


def _eot_column(text: Tensor, end_id: int) -> Tensor:
    """Column of the (single) EOT token per row (src/model/model.py:587-588).  The
    reference indexes `nonzero()[:, 1]` with arange(B), so every row must hold
    exactly one EOT; anything else is an IndexError there and a ValueError here."""
    hits = (text == end_id)
    if not bool((hits.sum(dim=1) == 1).all()):
        raise ValueError("every token row must contain exactly one EOT token")
    return hits.float().argmax(dim=1)


def _text_tower(sd: SD, x: Tensor, readout: Tensor) -> Tensor:
    a = arch_from_state_dict(sd)
    x = x + sd["positional_embedding"].float()
    x = transformer(x, sd, "transformer.", a["transformer_heads"], causal=True)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    rows = x[torch.arange(x.shape[0]), readout]
    return rows @ sd["text_projection"].float()


def encode_text(sd: SD, text: Tensor) -> Tensor:
    """CLIP.encode_text (src/model/model.py:577-590)."""
    end_id = sd["token_embedding.weight"].shape[0] - 1            # model.py:499
    x = sd["token_embedding.weight"].float()[text]
    return _text_tower(sd, x, _eot_column(text, end_id))


def encode_text_img_retrieval(sd: SD, text: Tensor, img_tokens: Tensor,
                              split_ind: int = 4, repeat: bool = True) -> Tensor:
    """CLIP.encode_text_img_retrieval, tensor (non-tuple) img_tokens (src/model/model.py:808-851).

    The first occurrence of `split_ind` in ROW 0 of `text` gives the insertion
    column for every row (:820,828).  n = 2 or 3 pseudo tokens replace that one
    token, the tail is shifted right and the last n-1 columns fall off (:832,834);
    the read-out row is EOT column + n - 1 (:847,849).
    """
    B = img_tokens.shape[0]
    if repeat:
        text = text.repeat(B, 1)
    n_tok = img_tokens.shape[1]
    if n_tok not in (2, 3):
        # the reference's else-branch would build a sequence of the wrong length
        raise ValueError("img_tokens must carry 2 or 3 pseudo tokens per row")
    end_id = sd["token_embedding.weight"].shape[0] - 1
    eot = _eot_column(text, end_id)
    where = (text[0] == split_ind).nonzero()
    if where.numel() == 0:
        raise IndexError("split token not present in text[0]")
    ins = int(where[0])
    emb = sd["token_embedding.weight"].float()[text]              # [B,L,d]
    L = emb.shape[1]
    tail = emb[:, ins + 1: L - (n_tok - 1)]
    x = torch.cat([emb[:, :ins], img_tokens.float(), tail], dim=1)
    readout = eot + (n_tok - 1)
    if int(readout.max()) >= L:
        raise IndexError("read-out row beyond the context length")
    return _text_tower(sd, x, readout)


def encode_text_img_train(sd: SD, text: Tensor, img_tokens: Tensor, split_ind: int = 4,
                          repeat: bool = True) -> Tensor:
    """CLIP.encode_text_img_train (src/model/model.py:853-892), the splice evaluate_fashion calls
    (src/eval_utils.py:957,969): THREE token positions starting at the first `split_ind` of row 0 are
    overwritten by the pseudo tokens (:880), nothing shifts, read-out at the EOT column (:891).
    Any other token count makes the sequence length differ from the positional table, which fails in
    the reference too (:883).  `repeat` is accepted and ignored, as in the reference."""
    if img_tokens.shape[1] != 3:
        raise RuntimeError("encode_text_img_train needs exactly 3 pseudo tokens (sequence length would not be "
                           f"{text.shape[1]})")
    end_id = sd["token_embedding.weight"].shape[0] - 1
    eot = _eot_column(text, end_id)
    where = (text[0] == split_ind).nonzero()
    if where.numel() == 0:
        raise IndexError("split token not present in text[0]")
    ins = int(where[0])
    emb = sd["token_embedding.weight"].float()[text]
    x = torch.cat([emb[:, :ins], img_tokens.float(), emb[:, ins + 3:]], dim=1)
    return _text_tower(sd, x, eot)


# ----------------------------------------------------------------------------
# knowledge injection modules
# ----------------------------------------------------------------------------
def im2text(sd: SD, x: Tensor) -> Tensor:
    """IM2TEXT.forward in eval mode: (Linear, Dropout=id, ReLU) x n_layer, then fc_out
    (src/model/model.py:105-123)."""
    i = 0
    x = x.float()
    while f"layers.{i}.0.weight" in sd:
        x = torch.relu(linear(x, sd[f"layers.{i}.0.weight"], sd[f"layers.{i}.0.bias"]))
        i += 1
    return linear(x, sd["fc_out.weight"], sd["fc_out.bias"])


def cross_attention(sd: SD, pfx: str, q: Tensor, k: Tensor, v: Tensor, heads: int = 8) -> Tensor:
    """CrossAttention.forward (src/model/model.py:56-79): biased q/k/v projections to
    heads*64, softmax(q k^T / sqrt(64)) v, biased out projection.  No residual, no norm."""
    B, nq, _ = q.shape
    nk = k.shape[1]
    Q = linear(q.float(), sd[pfx + "to_q.weight"], sd[pfx + "to_q.bias"])
    K = linear(k.float(), sd[pfx + "to_k.weight"], sd[pfx + "to_k.bias"])
    V = linear(v.float(), sd[pfx + "to_v.weight"], sd[pfx + "to_v.bias"])
    dh = Q.shape[-1] // heads
    Q = Q.reshape(B, nq, heads, dh).transpose(1, 2)
    K = K.reshape(B, nk, heads, dh).transpose(1, 2)
    V = V.reshape(B, nk, heads, dh).transpose(1, 2)
    p = torch.softmax((Q @ K.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    o = (p @ V).transpose(1, 2).reshape(B, nq, heads * dh)
    return linear(o, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])


def crossformer(sd: SD, q: Tensor, k: Tensor, v: Tensor, heads: int = 8) -> Tensor:
    """CrossFormer.forward: q chained through the layers, k and v fixed (src/model/model.py:98-101)."""
    i = 0
    while f"cross_layers.{i}.to_q.weight" in sd:
        q = cross_attention(sd, f"cross_layers.{i}.", q, k, v, heads)
        i += 1
    return q


# ----------------------------------------------------------------------------
# search (Faiss IndexFlatL2 stand-in) and knowledge retrieval
# ----------------------------------------------------------------------------
def l2_normalize(x: Tensor) -> Tensor:
    return x / x.norm(dim=-1, keepdim=True)


def flat_l2_search(db: Tensor, q: Tensor, k: int, chunk: int = 65536) -> Tuple[Tensor, Tensor]:
    """Exact squared-L2 brute force: returns (D [B,k] ascending fp32, I [B,k] int64).

    What `faiss.IndexFlatL2(768).search(q, k)` returns (src/eval_retrieval.py:291-296,
    src/eval_utils.py:169,177).  Distances are evaluated directly as sum((q-d)^2) in
    fp64 then rounded to fp32, so the oracle itself has no cancellation error; ties
    go to the lower index.
    """
    db = db.float()
    q64 = q.double()
    B = q.shape[0]
    best_d = torch.full((B, 0), 0.0, dtype=torch.float64)
    best_i = torch.zeros((B, 0), dtype=torch.int64)
    for s in range(0, db.shape[0], chunk):
        blk = db[s:s + chunk].double()
        d = (q64 * q64).sum(1, keepdim=True) - 2.0 * (q64 @ blk.t()) + (blk * blk).sum(1)[None, :]
        # exact re-evaluation is not needed in fp64: cancellation error ~1e-16
        idx = torch.arange(s, s + blk.shape[0], dtype=torch.int64).expand(B, -1)
        best_d = torch.cat([best_d, d], dim=1)
        best_i = torch.cat([best_i, idx], dim=1)
        kk = min(k, best_d.shape[1])
        # stable sort => lower index first among equal distances
        order = torch.sort(best_d, dim=1, stable=True).indices[:, :kk]
        best_d = torch.gather(best_d, 1, order)
        best_i = torch.gather(best_i, 1, order)
    return best_d.clamp_min(0).float(), best_i


def flat_l2_search_f32(db: Tensor, q: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """Same search evaluated the way a CPU library does it (fp32 GEMM expansion
    ||q||^2 - 2 q.x + ||x||^2, then top-k): the form timed as `cpu_baseline` in bench.py.
    Equal to flat_l2_search up to fp32 cancellation error (not used as the parity checker)."""
    db = db.float()
    q = q.float()
    d = (q * q).sum(1, keepdim=True) - 2.0 * (q @ db.t()) + (db * db).sum(1)[None, :]
    v, i = torch.topk(d, min(k, db.shape[0]), dim=1, largest=False, sorted=True)
    return v.clamp_min(0), i


def flat_ip_search(db: Tensor, q: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """The reference's own brute-force statement: `feature @ base.t()` then `.topk`
    (src/trainer.py:246-257).  Returns (scores desc, indices)."""
    s = q.double() @ db.double().t()
    order = torch.sort(-s, dim=1, stable=True).indices[:, :k]
    return torch.gather(s, 1, order).float(), order


def get_retrieved_features(feature: Tensor, image_base: Tensor, text_base: Tensor,
                           topk: int = 16) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """get_retrieved_features (src/eval_utils.py:153-186) without the randperm of the
    neighbour axis (a numerical no-op: softmax attention over keys has no position,
    SURVEY.md App. B).  Returns (topk_image [B,k,D], topk_text [B,k,D], I_img, I_txt)."""
    q = l2_normalize(feature.float())
    _, ii = flat_l2_search(image_base, q, topk)
    _, it = flat_l2_search(text_base, q, topk)
    B = q.shape[0]
    ti = image_base.float()[ii.reshape(-1)].reshape(B, topk, -1)
    tt = text_base.float()[it.reshape(-1)].reshape(B, topk, -1)
    return ti, tt, ii, it


def knowledge_tokens(sd_i2t: SD, sd_fuse: SD, sd_cond: SD, q_feat: Tensor,
                     topk_image: Tensor, topk_text: Tensor) -> Tensor:
    """One stream of the knowledge injection (src/eval_utils.py:661-672): returns the
    three pseudo tokens [fused, text_conditioned, mapped] as [B,3,D]."""
    mapped = im2text(sd_i2t, q_feat)
    ti = im2text(sd_i2t, topk_image)
    tt = im2text(sd_i2t, topk_text)
    fused = crossformer(sd_fuse, mapped[:, None, :], ti, ti)
    cond = crossformer(sd_cond, mapped[:, None, :], tt, tt)
    return torch.cat([fused, cond, mapped[:, None, :]], dim=1)


def compose_query(sd_clip: SD, stream_img: Sequence[SD], stream_txt: Sequence[SD],
                  ref_images: Tensor, text_with_blank: Tensor,
                  image_base: Tensor, text_base: Tensor, split_ind: int = 265,
                  topk: int = 16, repeat: bool = False, w_text_stream: float = 0.5) -> Dict[str, Tensor]:
    """Per-batch body of evaluate_cirr (src/eval_utils.py:652-714).  The other drivers run the same body with
    a different prompt broadcast and mixture weight: evaluate_coco (:511-548, repeat=False, w = 0.05 j) and
    evaluate_imgnet_retrieval (:372-415, one prompt row repeated over the batch, w = 0.1 j).

    stream_* = (img2text, retrieval_fuse, text_condition) state dicts of the image
    stream and of the text stream ('_tb').  Returns the three normalised feature sets
    under the reference's dict names (src/eval_utils.py:728-732): 'composed' is the
    image-stream feature, 'image' the text-stream one, 'mixture' their mean."""
    q_feat = encode_image(sd_clip, ref_images)
    ti, tt, ii, it = get_retrieved_features(q_feat, image_base, text_base, topk)
    tok_a = knowledge_tokens(*stream_img, q_feat, ti, tt)
    comp_a = encode_text_img_retrieval(sd_clip, text_with_blank, tok_a, split_ind=split_ind, repeat=repeat)
    tok_b = knowledge_tokens(*stream_txt, q_feat, ti, tt)
    comp_b = encode_text_img_retrieval(sd_clip, text_with_blank, tok_b, split_ind=split_ind, repeat=repeat)
    a = l2_normalize(comp_a)
    b = l2_normalize(comp_b)
    mix = l2_normalize(w_text_stream * b + (1.0 - w_text_stream) * a)
    return {"composed": a, "image": b, "mixture": mix, "query_image_features": q_feat,
            "tokens_image_stream": tok_a, "tokens_text_stream": tok_b,
            "topk_image_indices": ii, "topk_text_indices": it}


# ----------------------------------------------------------------------------
# metric
# ----------------------------------------------------------------------------
def get_metrics_cirr(image_features: Tensor, ref_features: Tensor, reference_names: Sequence[str],
                     index_names: Sequence[str], target_names: Sequence[str]) -> Dict[str, float]:
    """get_metrics_cirr (src/eval_utils.py:1040-1067): rank the gallery by 1 - cosine,
    drop the reference image from each ranking, Recall@{1,5,10,50,100} in percent.
    Names are compared by basename (:1046-1048)."""
    dist = 1.0 - ref_features.float() @ image_features.float().t()
    order = torch.sort(dist, dim=1, stable=True).indices.numpy()        # [Q,G]
    base = np.array([os.path.basename(n) for n in index_names])
    ranked = base[order]                                                 # [Q,G] names
    ref = np.array(list(reference_names))[:, None]
    keep = ranked != ref
    if not (keep.sum(1) == ranked.shape[1] - 1).all():
        raise AssertionError("each reference image must appear exactly once in the gallery")
    ranked = ranked[keep].reshape(ranked.shape[0], ranked.shape[1] - 1)
    hit = ranked == np.array(list(target_names))[:, None]
    if not (hit.sum(1) == 1).all():                                      # :1063
        raise AssertionError("each target must appear exactly once in the ranking")
    out = {}
    for k in (1, 5, 10, 50, 100):
        out[f"recall_R@{k}"] = float(hit[:, :k].sum()) / hit.shape[0] * 100.0
    return out


def get_metrics_fashion(image_features: Tensor, ref_features: Tensor, target_names: Sequence[str],
                        answer_names: Sequence[str]) -> Dict[str, float]:
    """get_metrics_fashion (src/eval_utils.py:1025-1037): rank the gallery by 1 - cosine, Recall@k in
    percent of the answer image (names compared whole, no reference removal)."""
    dist = 1.0 - ref_features.float() @ image_features.float().t()
    order = torch.sort(dist, dim=1, stable=True).indices.numpy()
    ranked = np.array(list(target_names))[order]
    hit = ranked == np.array(list(answer_names))[:, None]
    if not (hit.sum(1) == 1).all():                                      # :1033
        raise AssertionError("each answer must appear exactly once in the gallery")
    return {f"R@{k}": float(hit[:, :k].sum()) / hit.shape[0] * 100.0 for k in (1, 5, 10, 50, 100)}


def get_metrics_coco(image_features: Tensor, ref_features: Tensor, logit_scale) -> Dict[str, float]:
    """get_metrics_coco (src/eval_utils.py:1008-1022): paired features, both directions; rank of the
    pair partner (0-based `preds`), mean/median rank (1-based) and R@k as fractions."""
    logits = (float(logit_scale) * image_features.float() @ ref_features.float().t())
    out: Dict[str, float] = {}
    gt = torch.arange(len(ref_features)).view(-1, 1)
    for name, lg in (("image_to_ref", logits), ("ref_to_image", logits.t())):
        ranking = torch.sort(lg, dim=1, descending=True, stable=True).indices
        preds = torch.where(ranking == gt)[1].numpy()
        out[f"{name}_mean_rank"] = float(preds.mean() + 1)
        out[f"{name}_median_rank"] = float(np.floor(np.median(preds)) + 1)
        for k in (1, 5, 10, 50, 100):
            out[f"{name}_R@{k}"] = float(np.mean(preds < k))
    return out


def get_metrics_imgnet(query_features: Tensor, image_features: Tensor, query_labels: Tensor,
                       target_labels: Tensor) -> Dict[str, float]:
    """get_metrics_imgnet (src/eval_utils.py:1090-1134): multi-positive retrieval.  For k in
    {1,5,10,50,100,200}: hits = same-label items among the top-k by dot product; recall = hits /
    (#same-label targets + 1e-5), precision = hits / k, both averaged over the queries (the
    reference's per-100 batching is a weighted mean of batch means = the plain mean)."""
    sim = query_features.float() @ image_features.float().t()
    ranking = torch.sort(sim, dim=1, descending=True, stable=True).indices
    same = (target_labels[ranking] == query_labels[:, None])             # [Q,T] in rank order
    total = same.sum(1).float()
    out: Dict[str, float] = {}
    for k in (1, 5, 10, 50, 100, 200):
        hits = same[:, :k].sum(1).float()
        out[f"Real2Sketch_R@{k}"] = float((hits / (total + 1e-5)).mean())
        out[f"Real2Sketch_P@{k}"] = float((hits / float(min(k, same.shape[1]))).mean())
    return out


def get_cirr_testoutput(image_features: Tensor, ref_features: Tensor, reference_names: Sequence[str],
                        index_names: Sequence[str], id_names) -> Dict[str, object]:
    """get_cirr_testoutput (src/eval_utils.py:1070-1087): the CIRR test-server submission: per pair id
    the top-50 gallery names (reference image removed, '.png' stripped)."""
    dist = 1.0 - ref_features.float() @ image_features.float().t()
    order = torch.sort(dist, dim=1, stable=True).indices.numpy()
    ranked = np.array(list(index_names))[order]
    keep = ranked != np.array(list(reference_names))[:, None]
    ranked = ranked[keep].reshape(ranked.shape[0], ranked.shape[1] - 1)
    out: Dict[str, object] = {"version": "rc2", "metric": "recall"}
    for i in range(len(id_names)):
        out[str(int(id_names[i]))] = [str(n).replace(".png", "") for n in ranked[i][:50]]
    return out


# ----------------------------------------------------------------------------
# synthetic, regenerable inputs (shared by tests, smoke and bench)
# ----------------------------------------------------------------------------
def _key_seed(key: str, seed: int) -> int:
    h = 2166136261
    for ch in (key + f"#{seed}").encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def synth_tensor(key: str, shape: Sequence[int], std: float, seed: int = 0) -> Tensor:
    rs = np.random.RandomState(_key_seed(key, seed))
    return torch.from_numpy((rs.standard_normal(tuple(shape)) * std).astype(np.float32))


def synth_clip_state_dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024,
                          vision_patch_size=14, context_length=77, vocab_size=49408,
                          transformer_width=768, transformer_layers=12, seed: int = 0,
                          visual_only: bool = False) -> Dict[str, Tensor]:
    """Seeded random-init weights with the reference's init stds
    (src/model/model.py:383-391,511-541).  LayerNorm gains get a small perturbation
    around 1 so a gain/bias mix-up cannot hide.  Every tensor has its own key-derived seed, so
    `visual_only=True` (text-tower tensors left zero: right shapes, no random draw; a third of the
    generation time) yields the SAME visual tower as the full dictionary."""
    sd: Dict[str, Tensor] = {}
    g = image_resolution // vision_patch_size

    def t(key, shape, std):
        if visual_only and not key.startswith("visual."):
            sd[key] = torch.zeros(tuple(shape), dtype=torch.float32)
            return
        sd[key] = synth_tensor(key, shape, std, seed)

    def ln(key, d):
        if visual_only and not key.startswith("visual."):
            sd[key + ".weight"], sd[key + ".bias"] = torch.ones(d), torch.zeros(d)
            return
        sd[key + ".weight"] = 1.0 + synth_tensor(key + ".weight", [d], 0.05, seed)
        sd[key + ".bias"] = synth_tensor(key + ".bias", [d], 0.05, seed)

    def tower(pfx, width, layers):
        proj_std = width ** -0.5 * (2 * layers) ** -0.5
        for i in range(layers):
            p = pfx + f"resblocks.{i}."
            t(p + "attn.in_proj_weight", [3 * width, width], width ** -0.5)
            t(p + "attn.in_proj_bias", [3 * width], 0.02)
            t(p + "attn.out_proj.weight", [width, width], proj_std)
            t(p + "attn.out_proj.bias", [width], 0.02)
            ln(p + "ln_1", width)
            t(p + "mlp.c_fc.weight", [4 * width, width], (2 * width) ** -0.5)
            t(p + "mlp.c_fc.bias", [4 * width], 0.02)
            t(p + "mlp.c_proj.weight", [width, 4 * width], proj_std)
            t(p + "mlp.c_proj.bias", [width], 0.02)
            ln(p + "ln_2", width)

    vs = vision_width ** -0.5
    t("visual.class_embedding", [vision_width], vs)
    t("visual.positional_embedding", [g * g + 1, vision_width], vs)
    t("visual.proj", [vision_width, embed_dim], vs)
    t("visual.conv1.weight", [vision_width, 3, vision_patch_size, vision_patch_size],
      (3 * vision_patch_size ** 2) ** -0.5)
    ln("visual.ln_pre", vision_width)
    ln("visual.ln_post", vision_width)
    tower("visual.transformer.", vision_width, vision_layers)
    t("positional_embedding", [context_length, transformer_width], 0.01)
    t("text_projection", [transformer_width, embed_dim], transformer_width ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    t("token_embedding.weight", [vocab_size, transformer_width], 0.02)
    ln("ln_final", transformer_width)
    tower("transformer.", transformer_width, transformer_layers)
    return sd


def synth_im2text_state_dict(embed_dim=768, middle_dim=512, output_dim=768, n_layer=2,
                             seed: int = 0, tag: str = "i2t") -> Dict[str, Tensor]:
    sd: Dict[str, Tensor] = {}
    d = embed_dim
    for i in range(n_layer):
        sd[f"layers.{i}.0.weight"] = synth_tensor(f"{tag}.l{i}.w", [middle_dim, d], d ** -0.5, seed)
        sd[f"layers.{i}.0.bias"] = synth_tensor(f"{tag}.l{i}.b", [middle_dim], 0.02, seed)
        d = middle_dim
    sd["fc_out.weight"] = synth_tensor(f"{tag}.o.w", [output_dim, middle_dim], middle_dim ** -0.5, seed)
    sd["fc_out.bias"] = synth_tensor(f"{tag}.o.b", [output_dim], 0.02, seed)
    return sd


def synth_crossformer_state_dict(dim=768, num_layers=3, heads=8, dim_head=64,
                                 seed: int = 0, tag: str = "xf") -> Dict[str, Tensor]:
    sd: Dict[str, Tensor] = {}
    inner = heads * dim_head
    for i in range(num_layers):
        p = f"cross_layers.{i}."
        for nm in ("to_q", "to_k", "to_v"):
            sd[p + nm + ".weight"] = synth_tensor(f"{tag}.{i}.{nm}.w", [inner, dim], dim ** -0.5, seed)
            sd[p + nm + ".bias"] = synth_tensor(f"{tag}.{i}.{nm}.b", [inner], 0.02, seed)
        sd[p + "to_out.0.weight"] = synth_tensor(f"{tag}.{i}.o.w", [dim, inner], inner ** -0.5, seed)
        sd[p + "to_out.0.bias"] = synth_tensor(f"{tag}.{i}.o.b", [dim], 0.02, seed)
    return sd


def synth_database(n: int, dim: int = 768, seed: int = 2002, clustered: bool = False,
                   n_centroids: int = 4096, sigma: float = 0.15) -> Tensor:
    """Seeded unit-norm database rows: iid Gaussian, or clustered
    (centroid + sigma * noise) for realistic neighbour gaps."""
    rs = np.random.RandomState(seed)
    if clustered:
        cent = rs.standard_normal((n_centroids, dim)).astype(np.float32)
        cent /= np.linalg.norm(cent, axis=1, keepdims=True)
        which = rs.randint(0, n_centroids, size=n)
        # per-element noise std sigma/sqrt(dim): the noise vector has norm ~ sigma
        x = cent[which] + (sigma / math.sqrt(dim)) * rs.standard_normal((n, dim)).astype(np.float32)
    else:
        x = rs.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return torch.from_numpy(x.astype(np.float32))


def synth_tokens(batch: int, context_length: int = 77, seed: int = 4004,
                 sot: int = 49406, eot: int = 49407, star: int = 265) -> Tensor:
    """'a photo of * , <filler>' token rows: SOT 320 1125 539 * 267 filler... EOT 0...
    with the EOT column varying per row in [8, 40] (ids from
    src/third_party/open_clip/simple_tokenizer.py:73-74 and src/data.py:295)."""
    rs = np.random.RandomState(seed)
    out = np.zeros((batch, context_length), dtype=np.int64)
    for b in range(batch):
        e = 8 + int(rs.randint(0, 33))
        row = [sot, 320, 1125, 539, star, 267] + list(rs.randint(300, 40000, size=e - 6)) + [eot]
        out[b, :len(row)] = row
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------
# weight / input variants for the full-size parity fixtures (tools/mint_golden.py, tests/test_gpu_fullsize.py)
# ----------------------------------------------------------------------------
def sharpen_clip(sd: Mapping[str, Tensor], qk: float = 1.5, resid: float = 2.0) -> Dict[str, Tensor]:
    """Random-init towers average their tokens almost uniformly, so every input lands on nearly the same embedding
    (cosine 0.995 between unrelated images) and a ranking over such features is decided by rounding noise.  Scaling the
    q/k projections (peakier attention) and the residual-branch output projections of the VISUAL tower spreads the
    embeddings: cosine ~0.9 between unrelated images at (1.5, 2.0).  Stronger settings turn the random network chaotic --
    at (3, 4) rounding the WEIGHTS to bf16 alone moves the fp32 embedding to cosine 0.90, at (2.5, 3) to 0.997, at (1.5, 2)
    to 0.999993 (measured with this oracle) -- which would test the network's conditioning, not the kernels."""
    out = dict(sd)
    w = sd["visual.conv1.weight"].shape[0]
    i = 0
    while f"visual.transformer.resblocks.{i}.attn.in_proj_weight" in sd:
        p = f"visual.transformer.resblocks.{i}."
        a = sd[p + "attn.in_proj_weight"].clone()
        a[:2 * w] *= qk
        out[p + "attn.in_proj_weight"] = a
        out[p + "attn.out_proj.weight"] = sd[p + "attn.out_proj.weight"] * resid
        out[p + "mlp.c_proj.weight"] = sd[p + "mlp.c_proj.weight"] * resid
        i += 1
    return out


def make_heavy_tailed(sd: Mapping[str, Tensor]) -> Dict[str, Tensor]:
    """Massive activations as real CLIP checkpoints have them: from block 3 (visual) / block 2 (text) on, three channels of
    the residual stream carry offsets of +120 / -80 / +50 (about 50-100 x the std of the other channels) that vary per
    token, and later LayerNorm gains damp (x0.3) or amplify (x2) exactly those channels."""
    out = dict(sd)
    for pfx, blk, chans in (("visual.transformer.", 3, (17, 389, 700)), ("transformer.", 2, (5, 300, 611))):
        p = pfx + f"resblocks.{blk}."
        b = sd[p + "mlp.c_proj.bias"].clone()
        wgt = sd[p + "mlp.c_proj.weight"].clone()
        for c, v in zip(chans, (120.0, -80.0, 50.0)):
            b[c] += v
            wgt[c] *= 20.0
        out[p + "mlp.c_proj.bias"], out[p + "mlp.c_proj.weight"] = b, wgt
        i = blk + 1
        while pfx + f"resblocks.{i}.ln_1.weight" in sd:
            for ln in ("ln_1", "ln_2"):
                g = sd[pfx + f"resblocks.{i}.{ln}.weight"].clone()
                g[chans[0]] *= 0.3
                g[chans[1]] *= 2.0
                g[chans[2]] *= 0.3
                out[pfx + f"resblocks.{i}.{ln}.weight"] = g
            i += 1
    return out


def synth_gallery_images(n: int, start: int = 0, res: int = 224) -> Tensor:
    """Image i of the synthetic gallery: RandomState(50000 + i) (chunkable: any slice is reproducible on its own)."""
    out = np.empty((n, 3, res, res), dtype=np.float32)
    for i in range(n):
        out[i] = np.random.RandomState(50000 + start + i).standard_normal((3, res, res)).astype(np.float32)
    return torch.from_numpy(out)


def synth_recall_plan(n_gallery: int, n_query: int, seed: int = 6006):
    """(target index, reference index, noise level) of every query: noise levels graded geometrically 0.1 .. 3."""
    rs = np.random.RandomState(seed)
    tgt = rs.randint(0, n_gallery, size=n_query)
    ref = (tgt + 1 + rs.randint(0, n_gallery - 1, size=n_query)) % n_gallery
    sigma = (0.1 * 30.0 ** (np.arange(n_query) / max(n_query - 1, 1))).astype(np.float32)
    return tgt.astype(np.int64), ref.astype(np.int64), sigma


def synth_recall_queries(tgt: np.ndarray, sigma: np.ndarray, start: int = 0, count: Optional[int] = None,
                         res: int = 224) -> Tensor:
    """Query j = gallery image tgt[j] + sigma[j] * RandomState(70000 + j) noise (chunkable)."""
    count = len(tgt) - start if count is None else count
    out = np.empty((count, 3, res, res), dtype=np.float32)
    for j in range(start, start + count):
        g = np.random.RandomState(50000 + int(tgt[j])).standard_normal((3, res, res)).astype(np.float32)
        n = np.random.RandomState(70000 + j).standard_normal((3, res, res)).astype(np.float32)
        out[j - start] = g + float(sigma[j]) * n
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------
# training step (src/trainer.py:44-165 get_loss_img2text_image), differentiable: torch autograd on this restatement is the
# reference the HIP backward kernels are checked against (tests/test_gpu_train.py)
# ----------------------------------------------------------------------------
def im2text_train(sd: SD, x: Tensor, masks: Optional[Sequence[Tensor]], p_drop: float) -> Tensor:
    """IM2TEXT.forward in TRAINING mode (src/model/model.py:112-116,120-123): Linear -> Dropout(p) -> ReLU per hidden
    layer with the given keep-masks (1 = keep, scaled by 1/(1-p)), then fc_out."""
    i = 0
    x = x.float()
    while f"layers.{i}.0.weight" in sd:
        z = linear(x, sd[f"layers.{i}.0.weight"], sd[f"layers.{i}.0.bias"])
        if masks is not None and p_drop > 0:
            z = z * masks[i].to(z.dtype) / (1.0 - p_drop)
        x = torch.relu(z)
        i += 1
    return linear(x, sd["fc_out.weight"], sd["fc_out.bias"])


def training_loss(sd_clip: SD, sd_i2t: SD, sd_fuse: SD, sd_cond: SD, image_features: Tensor, nbr_img: Tensor,
                  nbr_txt: Tensor, text: Tensor, split_ind: int, masks: Optional[Sequence[Tensor]] = None,
                  p_drop: float = 0.1, other_img_n: Optional[Tensor] = None, other_txt_n: Optional[Tensor] = None) -> Tensor:
    """src/trainer.py:44-127.  image_features [B,D] (precomputed, no gradient: :48-52), neighbours [B,K,D] as
    get_retrieved_features returns them (:55-57); mapped = img2text(...) of the query and of both neighbour sets (:60-62,
    one stacked call here so that one mask covers the rows [q; I; T]); fused / text_conditioned (:65-66); tokens
    [fused, text_conditioned, mapped] (:70); text features through the FROZEN text tower -- with the evaluation splice
    (model.py:808-851), see keds_amd/train.py for why; normalise (:80-81); logits = exp(logit_scale) * I . T^T with this
    rank's rows first and the other ranks' (no gradient) behind (:88-112); loss = (CE_rows + CE_cols) / 2 (:113-116,148)."""
    B, K, D = nbr_img.shape
    rows = torch.cat([image_features.float(), nbr_img.reshape(B * K, D).float(), nbr_txt.reshape(B * K, D).float()])
    mapped_all = im2text_train(sd_i2t, rows, masks, p_drop)
    mapped, mi, mt = mapped_all[:B], mapped_all[B:B + B * K].reshape(B, K, -1), mapped_all[B + B * K:].reshape(B, K, -1)
    fused = crossformer(sd_fuse, mapped.unsqueeze(1), mi, mi)
    cond = crossformer(sd_cond, mapped.unsqueeze(1), mt, mt)
    tokens = torch.cat([fused, cond, mapped.unsqueeze(1)], dim=1)
    if text.dim() == 1:
        text = text[None, :].repeat(B, 1)
    tf = encode_text_img_retrieval(sd_clip, text, tokens, split_ind=split_ind, repeat=False)
    img_n = image_features.float() / image_features.float().norm(dim=-1, keepdim=True)
    txt_n = tf / tf.norm(dim=-1, keepdim=True)
    if other_img_n is not None:
        img_n = torch.cat([img_n, other_img_n.float()])
        txt_n = torch.cat([txt_n, other_txt_n.float()])
    logits = sd_clip["logit_scale"].float().exp() * img_n @ txt_n.t()
    gt = torch.arange(logits.shape[0])
    ce = torch.nn.functional.cross_entropy
    return (ce(logits, gt) + ce(logits.t(), gt)) / 2
