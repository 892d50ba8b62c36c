"""Host-side mirror of the reference's retrieval glue (src/eval_utils.py) on top of the HIP path.

  get_retrieved_features   eval_utils.py:153-186   (normalise, top-16 over both DBs, gather rows)
  compose_query_features   eval_utils.py:652-714   (per-batch body of evaluate_cirr)
  get_metrics_cirr         eval_utils.py:1040-1067 (gallery ranking, reference removal, Recall@k)
  get_metrics_fashion / _coco / _imgnet, get_cirr_testoutput   eval_utils.py:1008-1134 (the other drivers' metrics)
  build_database           eval_retrieval.py:281-298 (DB tensors + two flat indices)

Everything stays on the device: no .cpu().numpy() round trip, no CPU fancy-index gather.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .index import FlatIndex, ShardedFlatIndex
from .model import CLIP, CrossFormer, IM2TEXT, KnowledgeStream


def build_database(image_bases: torch.Tensor, text_bases: torch.Tensor, basenames: Optional[Sequence[str]] = None,
                   device=None) -> list:
    """database = [image_bases, text_bases, basenames, image_index, text_index]
    (eval_retrieval.py:287,297-298).  The fp32 rows live on the device inside the indices."""
    d = image_bases.shape[1]
    ii, ti = FlatIndex(d, "l2", device=device), FlatIndex(d, "l2", device=device)
    ii.add(image_bases)
    ti.add(text_bases)
    return [ii.rows, ti.rows, list(basenames) if basenames is not None else None, ii, ti]


@torch.no_grad()
def extract_feature_database(model: CLIP, image_batches, text_batches, out_dir: Optional[str] = None, device=None):
    """Build the bi-modal database on the device (the reference ships only the result,
    README.md:8 / eval_retrieval.py:253-286): encode every image batch and every caption batch, L2-normalise, and return
    (image_bases, text_bases) as float32 [N, D] device tensors -- what `build_database` takes.  With `out_dir` the two
    matrices are also written as `cc_image_databases.pt` / `cc_text_databases.pt` (plain float32 tensors: the reference's
    own on-disk format) and the ready-to-search indices as `cc_image_index.pt` / `cc_text_index.pt` (FlatIndex.save)."""
    # Streaming: `image_batches` / `text_batches` may be DataLoaders or generators over the whole dataset (0.5 M images are
    # 300 GB of fp32 pixels) -- only VERIFY_CHUNK input batches are held at a time.  No row enters the database from an
    # unverified pass of the numerics guard: the guard is synchronised behind every chunk (CLIP.numerics_checked) and a chunk
    # during which it tripped late is encoded again -- on the safe flow by then -- from the inputs still held.
    import itertools

    def encode_stream(batches, enc):
        feats, it = [], iter(batches)
        while True:
            held = list(itertools.islice(it, VERIFY_CHUNK))
            if not held:
                break
            feats += model.numerics_checked(lambda: [enc(b.to(device) if device is not None else b, normalize=True).float() for b in held])
            held = None                                           # released before the next chunk is drawn
        if not feats:
            raise RuntimeError("extract_feature_database: no batches")
        return torch.cat(feats)
    image_bases = encode_stream(image_batches, model.encode_image)
    text_bases = encode_stream(text_batches, model.encode_text)
    if image_bases.shape != text_bases.shape:
        raise RuntimeError(f"image / text databases differ in shape: {tuple(image_bases.shape)} vs {tuple(text_bases.shape)}")
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        torch.save(image_bases.cpu(), os.path.join(out_dir, "cc_image_databases.pt"))
        torch.save(text_bases.cpu(), os.path.join(out_dir, "cc_text_databases.pt"))
        for name, base in (("cc_image_index.pt", image_bases), ("cc_text_index.pt", text_bases)):
            idx = FlatIndex(base.shape[1], "l2", device=base.device)
            idx.add(base)
            idx.save(os.path.join(out_dir, name))
    return image_bases, text_bases


VERIFY_CHUNK = 8          # input batches held (and re-encoded after a late trip of the numerics guard) at a time
SHARD_MANIFEST = "cc_database_shards.json"


@torch.no_grad()
def extract_feature_database_sharded(model: CLIP, n_rows: int, image_rows, text_rows, out_dir: str, rank: int = 0,
                                     world: int = 1, batch: int = 128, device=None, encode_image=None, encode_text=None):
    """Per-rank build of the row-sharded bi-modal database: rank r encodes ONLY the dataset rows
    `shard_bounds(n_rows, world, r)` -- `image_rows(lo, hi)` / `text_rows(lo, hi)` return the preprocessed images
    [hi-lo,3,R,R] / token rows [hi-lo,77] of that range -- normalises them and writes its two shard files
    `cc_{image,text}_index.shard{r}-of-{world}.pt` (FlatIndex.save: fp32 rows + the packed bf16 scan image + the shard's
    first global row id), exactly what `load_database_shard` puts back on the device.  A pure partition: no collective, no
    rank ever holds another rank's rows.  Rank 0 also writes the manifest.  Returns this rank's (image_index, text_index)."""
    import json
    from .index import shard_bounds
    lo, hi = shard_bounds(n_rows, world, rank)
    enc_i = encode_image or (lambda x: model.encode_image(x, normalize=True))
    enc_t = encode_text or (lambda x: model.encode_text(x, normalize=True))
    os.makedirs(out_dir, exist_ok=True)
    out = []
    for name, rows_fn, enc in (("image", image_rows, enc_i), ("text", text_rows, enc_t)):
        def encode_shard():
            idx = None
            for a in range(lo, hi, batch):
                b = min(hi, a + batch)
                x = rows_fn(a, b)
                f = enc(x.to(device) if device is not None else x).float()
                if idx is None:
                    idx = FlatIndex(f.shape[1], "l2", device=f.device, row0=lo)
                idx.add(f)                                        # chunked add: packs only the new stages
            return idx
        # (re-encoded on the safe flow if the guard tripped late; a caller that brings its own encoders passes model = None)
        idx = model.numerics_checked(encode_shard) if model is not None else encode_shard()
        if idx is None:
            raise RuntimeError(f"rank {rank} of {world} owns no rows of a {n_rows}-row database")
        idx.save(os.path.join(out_dir, f"cc_{name}_index.shard{rank}-of-{world}.pt"))
        out.append(idx)
    if rank == 0:
        with open(os.path.join(out_dir, SHARD_MANIFEST), "w") as f:
            json.dump({"n_rows": int(n_rows), "world": int(world), "dim": int(out[0].d),
                       "bounds": [list(shard_bounds(n_rows, world, r)) for r in range(world)]}, f)
    return out[0], out[1]


def load_database_shard(out_dir: str, rank: int = 0, world: int = 1, device=None) -> list:
    """database = [image rows, text rows, None, image_index, text_index] of THIS rank's shard, from the files
    `extract_feature_database_sharded` wrote (same list shape as `build_database`, eval_retrieval.py:287,297-298); the
    indices carry the shard's first global row id, so `ShardedFlatIndex` / `PackedExchange` merge them directly."""
    import json
    from .index import shard_bounds
    with open(os.path.join(out_dir, SHARD_MANIFEST)) as f:
        man = json.load(f)
    if man["world"] != world:
        raise RuntimeError(f"database was sharded {man['world']} ways, this job has {world} ranks: rebuild or re-shard")
    ii = FlatIndex.load(os.path.join(out_dir, f"cc_image_index.shard{rank}-of-{world}.pt"), device=device)
    ti = FlatIndex.load(os.path.join(out_dir, f"cc_text_index.shard{rank}-of-{world}.pt"), device=device)
    lo, hi = shard_bounds(man["n_rows"], world, rank)
    if ii.row0 != lo or ii.ntotal != hi - lo or ti.row0 != lo or ti.ntotal != hi - lo:
        raise RuntimeError("shard file does not match the manifest")
    return [ii.rows, ti.rows, None, ii, ti]


def get_retrieved_features(feature: torch.Tensor, database, args=None, topk: int = 16, use_faiss: bool = True):
    """eval_utils.py:153-186.  feature [B,D] (any norm) -> (topk_image [B,k,D], topk_text [B,k,D]).
    The reference shuffles the image neighbours along K (a numerical no-op for attention over keys);
    here they stay in rank order."""
    image_index, text_index = database[3], database[4]
    if isinstance(image_index, ShardedFlatIndex) and isinstance(text_index, ShardedFlatIndex):
        # row-sharded databases: one query all-gather and one packed all-to-all for BOTH databases
        (_, _, ti), (_, _, tt) = ShardedFlatIndex.search_gather_many([image_index, text_index], feature, topk, normalize=True)
        return ti, tt
    _, _, ti = image_index.search_gather(feature, topk, normalize=True)
    _, _, tt = text_index.search_gather(feature, topk, normalize=True)
    return ti, tt


def compose_query_features(model: CLIP, stream_image: KnowledgeStream, stream_text: KnowledgeStream,
                           ref_images: torch.Tensor, text_with_blank: torch.Tensor, database,
                           id_split: int = 265, topk: int = 16, repeat: bool = False,
                           w_text_stream: float = 0.5, verify: bool = True) -> Dict[str, torch.Tensor]:
    """Per-batch body of evaluate_cirr (eval_utils.py:652-714).  evaluate_coco (:511-548) is the same body with
    mixture weight w = 0.05 j for the text stream, evaluate_imgnet_retrieval (:372-415) passes ONE prompt row with
    repeat=True and w = 0.1 j.

    Returns the reference's three feature sets under its dict names (eval_utils.py:728-732):
    'composed' = image-stream feature, 'image' = text-stream feature, 'mixture' = their normalised mean.

    verify (default): the batch's three encoder passes are verified by the numerics guard before the features are
    returned, and re-run on the fp32-stream flow if it tripped (`CLIP.numerics_checked`: one host wait per batch, which
    the reference's own `.cpu()` in get_retrieved_features pays too).  A caller that keeps the device busy across
    batches passes verify=False and calls `model.numerics_sync()` itself before it uses the features (bench.py).
    """
    if verify:
        return model.numerics_checked(lambda: compose_query_features(
            model, stream_image, stream_text, ref_images, text_with_blank, database, id_split=id_split, topk=topk,
            repeat=repeat, w_text_stream=w_text_stream, verify=False))
    q = model.encode_image(ref_images).float()
    topk_image, topk_text = get_retrieved_features(q, database, None, topk=topk)
    prec = "fp32" if getattr(model, "precision", "bf16") in ("fp32", "fp32x3") else "bf16"   # fp32: no operand is rounded anywhere
    tok_a = stream_image(q, topk_image, topk_text, precision=prec)                   # [B,3,D]
    tok_b = stream_text(q, topk_image, topk_text, precision=prec)
    if not repeat and text_with_blank.shape[0] == tok_a.shape[0] == tok_b.shape[0]:
        # the two text-tower passes share the captions and differ in the spliced tokens only: one pass over 2B rows
        # (rows are independent; B = 128 -> 19,712 rows = 77 full 256-row GEMM tiles instead of two ragged 9,856-row passes)
        B = tok_a.shape[0]
        both = model.encode_text_img_retrieval(torch.cat([text_with_blank, text_with_blank]), torch.cat([tok_a, tok_b]),
                                               split_ind=id_split, repeat=False)
        comp_a, comp_b = both[:B], both[B:]
    else:
        comp_a = model.encode_text_img_retrieval(text_with_blank, tok_a, split_ind=id_split, repeat=repeat)
        comp_b = model.encode_text_img_retrieval(text_with_blank, tok_b, split_ind=id_split, repeat=repeat)
    b_n, a_n, mix = ops.mix_normalize(comp_b.float(), comp_a.float(), float(w_text_stream), 1.0 - float(w_text_stream))
    return {"composed": a_n, "image": b_n, "mixture": mix, "query_image_features": q,
            "tokens_image_stream": tok_a, "tokens_text_stream": tok_b}


def all_gather_features(features: torch.Tensor, group=None) -> torch.Tensor:
    """Multi-GPU evaluation glue (gallery ranking): every rank encoded its slice of the gallery (or of the
    queries); returns the row-wise concatenation in rank order on every rank, ragged slices allowed.  The metric functions
    then run on the full matrices exactly as on one GPU -- galleries are 1e3..1e5 rows, the exchange is one small
    collective (the 0.5 M-row databases stay sharded: `ShardedFlatIndex`)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return features
    world = dist.get_world_size(group)
    n = torch.tensor([features.shape[0]], dtype=torch.int64, device=features.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    pad = max(counts)
    buf = features.new_zeros((pad,) + tuple(features.shape[1:]))
    buf[:features.shape[0]] = features
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def _intern(names: Sequence[str], table: Dict[str, int]) -> np.ndarray:
    out = np.empty(len(names), dtype=np.int32)
    for i, n in enumerate(names):
        b = os.path.basename(str(n))
        out[i] = table.setdefault(b, len(table))
    return out


def get_metrics_cirr(image_features: torch.Tensor, ref_features: torch.Tensor, reference_names, index_names,
                     target_names) -> Dict[str, float]:
    """eval_utils.py:1040-1067 with the ranking, the reference-image removal and the target look-up on
    device.  Names are compared by basename, interned to integers once (O(Q+G) host work instead of the
    reference's O(Q*G) Python loop)."""
    table: Dict[str, int] = {}
    gal = _intern(index_names, table)
    ref = _intern(reference_names, table)
    tgt = _intern(target_names, table)
    order = ops.rank_gallery(ref_features, image_features)
    dev = order.device
    rank, counts = ops.cirr_target_rank(order, torch.from_numpy(gal).to(dev), torch.from_numpy(ref).to(dev),
                                        torch.from_numpy(tgt).to(dev))
    counts = counts.cpu()
    if not bool((counts[:, 0] == 1).all()):
        raise AssertionError("each reference image must appear exactly once in the gallery")
    if not bool((counts[:, 1] == 1).all()):       # eval_utils.py:1063
        raise AssertionError("each target must appear exactly once in the ranking")
    rank = rank.cpu()
    n = rank.shape[0]
    return {f"recall_R@{k}": float((rank < k).sum().item()) / n * 100.0 for k in (1, 5, 10, 50, 100)}


def get_metrics_cirr_topk(gallery_index, ref_features: torch.Tensor, reference_names, index_names, target_names,
                          depth: int = 101) -> Dict[str, float]:
    """get_metrics_cirr (eval_utils.py:1040-1067) from the best `depth` gallery rows only -- the sharded form SURVEY 8e
    prescribes ("gallery sharded like the DB, K = 101"): `gallery_index` is an inner-product FlatIndex / ShardedFlatIndex
    over the L2-normalised gallery features (rank r owns its rows; the partial top-101 lists are merged keyed on
    (score, id), so the ranking is the reference's stable ascending sort of 1 - ref.gallery^T truncated at 101).
    Removing the reference image can move the target up by one place, so Recall@100 needs the best 101.
    The full [Q, G] distance matrix and its sort are never built."""
    table: Dict[str, int] = {}
    gal = _intern(index_names, table)
    ref = _intern(reference_names, table)
    tgt = _intern(target_names, table)
    if len(set(gal.tolist())) != len(gal):
        raise AssertionError("gallery names must be unique")            # eval_utils.py:1063 (exactly one hit per row)
    n = gallery_index.ntotal if hasattr(gallery_index, "ntotal") else gallery_index.n_global
    depth = min(int(depth), int(n))
    got = gallery_index.search(ref_features, depth)
    I = got[1]
    dev = I.device
    names = torch.from_numpy(gal).to(dev)[I.clamp(min=0)]              # [Q, depth] interned gallery names
    names = torch.where(I >= 0, names, torch.full_like(names, -1))
    is_ref = names == torch.from_numpy(ref).to(dev)[:, None]
    is_tgt = names == torch.from_numpy(tgt).to(dev)[:, None]
    slots = torch.arange(depth, device=dev)[None, :]
    pos = torch.where(is_tgt, slots, torch.full_like(slots, depth)).amin(dim=1)          # first (only) hit or `depth`
    rank = pos - (is_ref & (slots < pos[:, None])).sum(dim=1)
    rank = torch.where(pos < depth, rank, torch.full_like(rank, 1 << 30)).cpu()
    q = rank.shape[0]
    return {f"recall_R@{k}": float((rank < k).sum().item()) / q * 100.0 for k in (1, 5, 10, 50, 100)}


def _intern_whole(names: Sequence[str], table: Dict[str, int]) -> np.ndarray:
    return np.fromiter((table.setdefault(str(n), len(table)) for n in names), dtype=np.int32, count=len(names))


def get_metrics_fashion(image_features: torch.Tensor, ref_features: torch.Tensor, target_names, answer_names
                        ) -> Dict[str, float]:
    """eval_utils.py:1025-1037: Recall@k (percent) of the answer image in the gallery ranking, on device."""
    table: Dict[str, int] = {}
    gal = _intern_whole(target_names, table)
    ans = _intern_whole(answer_names, table)
    order = ops.rank_gallery(ref_features, image_features)
    dev = order.device
    none = torch.full((len(ans),), -1, dtype=torch.int32, device=dev)          # nothing is removed from the ranking
    rank, counts = ops.cirr_target_rank(order, torch.from_numpy(gal).to(dev), none, torch.from_numpy(ans).to(dev))
    if not bool((counts[:, 1] == 1).all()):                                    # eval_utils.py:1033
        raise AssertionError("each answer must appear exactly once in the gallery")
    rank = rank.cpu()
    n = rank.shape[0]
    return {f"R@{k}": float((rank < k).sum().item()) / n * 100.0 for k in (1, 5, 10, 50, 100)}


def get_metrics_coco(image_features: torch.Tensor, ref_features: torch.Tensor, logit_scale=None) -> Dict[str, float]:
    """eval_utils.py:1008-1022: paired (image, composed) features, both retrieval directions; the positive
    logit_scale does not change a ranking, so it is accepted and unused."""
    n = ref_features.shape[0]
    if image_features.shape[0] != n:
        raise RuntimeError("get_metrics_coco needs paired features")
    out: Dict[str, float] = {}
    for name, a, b in (("image_to_ref", image_features, ref_features), ("ref_to_image", ref_features, image_features)):
        order = ops.rank_gallery(a, b)                 # ascending 1 - a.b == descending logits
        dev = order.device
        ids = torch.arange(n, dtype=torch.int32, device=dev)
        none = torch.full((n,), -1, dtype=torch.int32, device=dev)
        rank, _ = ops.cirr_target_rank(order, ids, none, ids)
        preds = rank.cpu().numpy()
        out[f"{name}_mean_rank"] = float(preds.mean() + 1)
        out[f"{name}_median_rank"] = float(np.floor(np.median(preds)) + 1)
        for k in (1, 5, 10, 50, 100):
            out[f"{name}_R@{k}"] = float(np.mean(preds < k))
    return out


def get_metrics_imgnet(query_features: torch.Tensor, image_features: torch.Tensor, query_labels, target_labels
                       ) -> Dict[str, float]:
    """eval_utils.py:1090-1134: multi-positive P@k / R@k for k in {1,5,10,50,100,200}; ranking and label counting
    on device (one-hot label matrices of the reference are never built)."""
    ks = (1, 5, 10, 50, 100, 200)
    order = ops.rank_gallery(query_features, image_features)
    hits, total = ops.label_hits(order, torch.as_tensor(target_labels), torch.as_tensor(query_labels), ks)
    hits, total = hits.cpu().float(), total.cpu().float()
    ng = image_features.shape[0]
    out: Dict[str, float] = {}
    for j, k in enumerate(ks):
        out[f"Real2Sketch_R@{k}"] = float((hits[:, j] / (total + 1e-5)).mean())
        out[f"Real2Sketch_P@{k}"] = float((hits[:, j] / float(min(k, ng))).mean())
    return out


def get_cirr_testoutput(image_features: torch.Tensor, ref_features: torch.Tensor, reference_names, index_names,
                        id_names) -> Dict[str, object]:
    """eval_utils.py:1070-1087: CIRR test-server submission: per pair id the 50 best gallery names with the
    reference image removed and '.png' stripped.  Ranking on device, 51 ids per query come back to the host."""
    order = ops.rank_gallery(ref_features, image_features)[:, :51].cpu().numpy()
    names = [str(n) for n in index_names]
    out: Dict[str, object] = {"version": "rc2", "metric": "recall"}
    for i in range(len(id_names)):
        ref = str(reference_names[i])
        ranked = [names[j] for j in order[i] if names[j] != ref][:50]
        out[str(int(id_names[i]))] = [n.replace(".png", "") for n in ranked]
    return out


# ---------------------------------------------------------------------------------------------------
# checkpoint plumbing (main.py:330-341, eval_retrieval.py:171-189, eval_utils.py:59-86)
# ---------------------------------------------------------------------------------------------------
def _strip_module(sd):
    if sd and next(iter(sd)).startswith("module."):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


def load_checkpoint(checkpoint: dict, model: Optional[CLIP], img2text: IM2TEXT, retrieval_fuse: CrossFormer,
                    text_condition: CrossFormer) -> None:
    """Load the reference's 4-part checkpoint dict {state_dict, state_dict_img2text,
    state_dict_retrieval_fuse, state_dict_text_condition} (optional 'module.' prefixes)."""
    if model is not None and "state_dict" in checkpoint:
        model.load_state_dict(_strip_module(checkpoint["state_dict"]), strict=False)
    img2text.load_state_dict(_strip_module(checkpoint["state_dict_img2text"]))
    retrieval_fuse.load_state_dict(_strip_module(checkpoint["state_dict_retrieval_fuse"]))
    text_condition.load_state_dict(_strip_module(checkpoint["state_dict_text_condition"]))


def make_stream_modules(model: CLIP, middle_dim: int = 512, n_layer: int = 2, device=None):
    """IM2TEXT + 2 x CrossFormer(num_layers=3) as instantiated at eval_retrieval.py:96-101."""
    d = model.token_embedding.weight.shape[1]
    a = IM2TEXT(embed_dim=model.embed_dim, middle_dim=middle_dim, output_dim=d, n_layer=n_layer).eval()
    b = CrossFormer(q_dim=d, k_dim=d, v_dim=d, num_layers=3).eval()
    c = CrossFormer(q_dim=d, k_dim=d, v_dim=d, num_layers=3).eval()
    if device is not None:
        a, b, c = a.to(device), b.to(device), c.to(device)
    return a, b, c
