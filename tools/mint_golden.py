#!/usr/bin/env python3
"""Mint golden vectors by RUNNING THE REFERENCE (build container only).

    python tools/mint_golden.py            # writes tests/golden/*.npz

Imports /root/reference/src/model/model.py directly (torch + einops only) and pulls
the pure functions `get_retrieved_features` / `get_metrics_cirr` out of
src/eval_utils.py and the brute-force branch of src/trainer.py with `ast`, so the
import-time file IO of those modules (eval_utils.py:46-57, data.py:56-76) never
runs.  Weights are NOT stored: they come from the seeded generator
`oracle.keds_oracle.synth_*_state_dict` (numpy legacy RandomState stream, frozen),
loaded into the reference modules with `load_state_dict`; a float64 checksum of the
weights is stored so generator drift is detected.  Only inputs + reference outputs
are written, as small .npz files.  The reference itself never travels.
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
sys.path.insert(0, REF)

from model.model import CLIP, IM2TEXT, CrossFormer  # noqa: E402  (the reference)
from oracle import keds_oracle as O  # noqa: E402  (only for the seeded generators)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_grad_enabled(False)


def extract_function(path, name, namespace):
    """exec one top-level function of a reference file in `namespace`."""
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, namespace)
            return namespace[name]
    raise KeyError(name)


def checksum(sd):
    return float(sum(v.double().sum().item() for v in sd.values()))


class FakeFlatL2:
    """Exact L2 brute force with the Faiss call shape (add / search -> D, I).
    Stands in for faiss.IndexFlatL2 (not installed; exact by definition)."""

    def __init__(self):
        self.x = None

    def add(self, x):
        self.x = np.asarray(x, dtype=np.float64)

    def search(self, q, k):
        q = np.asarray(q, dtype=np.float64)
        d = (q * q).sum(1)[:, None] - 2.0 * q @ self.x.T + (self.x * self.x).sum(1)[None, :]
        idx = np.argsort(d, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(d, idx, 1).astype(np.float32), idx.astype(np.int64)


def tiny_tokens(batch, vocab, star, rs):
    sot, eot = vocab - 2, vocab - 1
    out = np.zeros((batch, 77), dtype=np.int64)
    for b in range(batch):
        e = 8 + int(rs.randint(0, 33))
        row = [sot, 20, 21, 22, star, 23] + list(rs.randint(30, 200, size=e - 6)) + [eot]
        out[b, :len(row)] = row
    return out


def mint_clip(tag, cfg, batch, star):
    sd = O.synth_clip_state_dict(**cfg, seed=7)
    heads = cfg["transformer_width"] // 64
    model = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"],
                 cfg["vision_patch_size"], cfg["context_length"], cfg["vocab_size"],
                 cfg["transformer_width"], heads, cfg["transformer_layers"]).eval().float()
    missing = model.load_state_dict(sd, strict=True)
    rs = np.random.RandomState(1001)
    res = cfg["image_resolution"]
    image = rs.standard_normal((batch, 3, res, res)).astype(np.float32)
    if cfg["vocab_size"] == 49408:
        text = O.synth_tokens(batch, seed=4004).numpy()
    else:
        text = tiny_tokens(batch, cfg["vocab_size"], star, rs)
    d = cfg["transformer_width"]
    tok3 = (rs.standard_normal((batch, 3, d)) * 0.05).astype(np.float32)
    tok2 = (rs.standard_normal((batch, 2, d)) * 0.05).astype(np.float32)
    timg, ttxt = torch.from_numpy(image), torch.from_numpy(text)
    out = {"image": image, "text": text, "tok3": tok3, "tok2": tok2, "star": np.int64(star),
           "weights_checksum": np.float64(checksum(sd))}
    feat, mids = model.encode_image(timg, mid_feature=True)
    out["encode_image"] = feat.numpy()
    out["block_cls"] = np.stack([m[:, 0, :].numpy() for m in mids])        # [layers,B,width]
    if cfg["vision_layers"] <= 2:
        out["block_tokens"] = np.stack([m.numpy() for m in mids])
    out["encode_text"] = model.encode_text(ttxt).numpy()
    out["eti3"] = model.encode_text_img_retrieval(ttxt, torch.from_numpy(tok3), split_ind=star, repeat=False).numpy()
    out["eti2"] = model.encode_text_img_retrieval(ttxt, torch.from_numpy(tok2), split_ind=star, repeat=False).numpy()
    out["eti3_repeat"] = model.encode_text_img_retrieval(ttxt[:1], torch.from_numpy(tok3), split_ind=star,
                                                         repeat=True).numpy()
    i_n, t_n, scale = model(timg, ttxt)
    out["forward_image"], out["forward_text"], out["forward_scale"] = i_n.numpy(), t_n.numpy(), scale.numpy()
    np.savez_compressed(os.path.join(OUT, f"clip_{tag}.npz"), **out)
    print(f"clip_{tag}: ok", {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})
    return model, sd, timg, ttxt


def mint_knowledge(dim, middle, batch=5, k=16):
    sd_i = O.synth_im2text_state_dict(dim, middle, dim, 2, seed=11, tag="i2t")
    sd_f = O.synth_crossformer_state_dict(dim, 3, seed=12, tag="fuse")
    i2t = IM2TEXT(embed_dim=dim, middle_dim=middle, output_dim=dim, n_layer=2).eval()
    i2t.load_state_dict(sd_i, strict=True)
    xf = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).eval()
    xf.load_state_dict(sd_f, strict=True)
    rs = np.random.RandomState(55)
    x = rs.standard_normal((batch, dim)).astype(np.float32)
    nb = rs.standard_normal((batch, k, dim)).astype(np.float32)
    y = i2t(torch.from_numpy(x))
    ynb = i2t(torch.from_numpy(nb))
    z = xf(y.unsqueeze(1), ynb, ynb)
    np.savez_compressed(os.path.join(OUT, f"knowledge_d{dim}.npz"), x=x, nb=nb, im2text_x=y.numpy(),
                        im2text_nb=ynb.numpy(), crossformer=z.numpy(),
                        weights_checksum=np.float64(checksum(sd_i) + checksum(sd_f)))
    print("knowledge: ok")


def mint_cirr_batch(model, sd_clip, timg, ttxt, star, dim, middle):
    """Per-batch body of evaluate_cirr (eval_utils.py:652-714) driven through the
    reference's own modules and its own get_retrieved_features."""
    ns = {"torch": torch, "np": np}
    grf = extract_function(os.path.join(REF, "eval_utils.py"), "get_retrieved_features", ns)
    n_db = 2048
    image_base = O.synth_database(n_db, dim, seed=2002)
    text_base = O.synth_database(n_db, dim, seed=2003)
    ii, ti = FakeFlatL2(), FakeFlatL2()
    ii.add(image_base.numpy())
    ti.add(text_base.numpy())
    database = [image_base, text_base, [str(i) for i in range(n_db)], ii, ti]

    def stream(seed):
        sds = (O.synth_im2text_state_dict(dim, middle, dim, 2, seed=seed, tag="i2t"),
               O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="fuse"),
               O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="cond"))
        a = IM2TEXT(embed_dim=dim, middle_dim=middle, output_dim=dim, n_layer=2).eval()
        b = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).eval()
        c = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).eval()
        a.load_state_dict(sds[0]); b.load_state_dict(sds[1]); c.load_state_dict(sds[2])
        return a, b, c

    img2text, retrieval_fuse, text_condition = stream(21)
    img2text_tb, retrieval_fuse_tb, text_condition_tb = stream(22)
    m = model
    # ---- eval_utils.py:654-695, statement for statement, on the reference objects ----
    query_image_features = m.encode_image(timg)
    topk_image, topk_text = grf(query_image_features, database, None)
    mapped_features = img2text(query_image_features)
    topk_image_features = img2text(topk_image)
    topk_text_features = img2text(topk_text)
    fused_features = retrieval_fuse(mapped_features.unsqueeze(1), topk_image_features, topk_image_features)
    text_conditioned = text_condition(mapped_features.unsqueeze(1), topk_text_features, topk_text_features)
    fused_features = torch.cat([fused_features, text_conditioned, mapped_features.unsqueeze(1)], dim=1)
    composed_feature = m.encode_text_img_retrieval(ttxt, fused_features, split_ind=star, repeat=False)
    mapped_features_tb = img2text_tb(query_image_features)
    topk_image_features_tb = img2text_tb(topk_image)
    topk_text_features_tb = img2text_tb(topk_text)
    fused_features_tb = retrieval_fuse_tb(mapped_features_tb.unsqueeze(1), topk_image_features_tb, topk_image_features_tb)
    text_conditioned_tb = text_condition_tb(mapped_features_tb.unsqueeze(1), topk_text_features_tb, topk_text_features_tb)
    fused_features_tb = torch.cat([fused_features_tb, text_conditioned_tb, mapped_features_tb.unsqueeze(1)], dim=1)
    composed_feature_tb = m.encode_text_img_retrieval(ttxt, fused_features_tb, split_ind=star, repeat=False)
    # ---- eval_utils.py:698-710 ----
    query_image_features = composed_feature_tb
    query_image_features = query_image_features / query_image_features.norm(dim=-1, keepdim=True)
    composed_feature = composed_feature / composed_feature.norm(dim=-1, keepdim=True)
    mixture_features = 0.5 * query_image_features + 0.5 * composed_feature
    mixture_features = mixture_features / mixture_features.norm(dim=-1, keepdim=True)
    np.savez_compressed(os.path.join(OUT, "cirr_batch_tiny.npz"),
                        composed=composed_feature.numpy(), image=query_image_features.numpy(),
                        mixture=mixture_features.numpy(),
                        tokens_image_stream=fused_features.numpy(), tokens_text_stream=fused_features_tb.numpy(),
                        topk_image_sorted=np.sort(topk_image.numpy(), axis=1),
                        topk_text=topk_text.numpy(), n_db=np.int64(n_db))
    print("cirr_batch: ok")


def mint_search():
    """The reference's own brute-force statement (trainer.py:232-257, use_faiss=False)."""
    ns = {"torch": torch, "np": np}
    grf = extract_function(os.path.join(REF, "trainer.py"), "get_retrieved_features", ns)
    dim, n = 64, 3000
    image_base = O.synth_database(n, dim, seed=31)
    text_base = O.synth_database(n, dim, seed=32, clustered=True, n_centroids=64)
    q = O.synth_database(9, dim, seed=33)
    ti, tt = grf(q, [image_base, text_base], None, topk=16, use_faiss=False)
    fa, fb = FakeFlatL2(), FakeFlatL2()
    fa.add(image_base.numpy()); fb.add(text_base.numpy())
    Da, Ia = fa.search(q.numpy(), 16)
    Db, Ib = fb.search(q.numpy(), 16)
    # unit-norm rows: inner-product order == L2 order; check the fake index against the reference here
    assert np.array_equal(ti.numpy(), image_base.numpy()[Ia.reshape(-1)].reshape(9, 16, dim))
    assert np.array_equal(tt.numpy(), text_base.numpy()[Ib.reshape(-1)].reshape(9, 16, dim))
    np.savez_compressed(os.path.join(OUT, "search_small.npz"), q=q.numpy(), topk_image=ti.numpy(),
                        topk_text=tt.numpy(), I_image=Ia, I_text=Ib, D_image=Da, D_text=Db)
    print("search: ok")


def mint_metrics():
    ns = {"torch": torch, "np": np, "os": os}
    gm = extract_function(os.path.join(REF, "eval_utils.py"), "get_metrics_cirr", ns)
    rs = np.random.RandomState(77)
    G, Q, dim = 300, 40, 32
    gallery = O.l2_normalize(torch.from_numpy(rs.standard_normal((G, dim)).astype(np.float32)))
    index_names = [f"/data/cirr/dev/img_{i:05d}.png" for i in range(G)]
    ref_idx = rs.randint(0, G, size=Q)
    tgt_idx = (ref_idx + 1 + rs.randint(0, G - 1, size=Q)) % G
    ref_feats = O.l2_normalize(gallery[tgt_idx] + 0.9 * torch.from_numpy(rs.standard_normal((Q, dim)).astype(np.float32)) / np.sqrt(dim) * 3)
    reference_names = [os.path.basename(index_names[i]) for i in ref_idx]
    target_names = [os.path.basename(index_names[i]) for i in tgt_idx]
    metrics = gm(gallery, ref_feats, np.array(reference_names), np.array(index_names), np.array(target_names))
    np.savez_compressed(os.path.join(OUT, "metrics_cirr.npz"), gallery=gallery.numpy(), ref=ref_feats.numpy(),
                        ref_idx=ref_idx, tgt_idx=tgt_idx,
                        **{k.replace("@", "_at_"): np.float64(v) for k, v in metrics.items()})
    print("metrics:", metrics)


def mint_eval_glue():
    """(f) rank 1: the other eval drivers' glue — encode_text_img_train (model.py:853-892, evaluate_fashion's splice),
    get_metrics_fashion / get_metrics_coco / get_metrics_imgnet / get_cirr_testoutput (eval_utils.py:1008-1134)."""
    import torch.nn.functional as F
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    model = CLIP(TINY["embed_dim"], TINY["image_resolution"], TINY["vision_layers"], TINY["vision_width"],
                 TINY["vision_patch_size"], TINY["context_length"], TINY["vocab_size"], TINY["transformer_width"],
                 TINY["transformer_width"] // 64, TINY["transformer_layers"]).eval().float()
    model.load_state_dict(sd, strict=True)
    rs = np.random.RandomState(4242)
    star = 265
    text = tiny_tokens(5, TINY["vocab_size"], star, rs)
    tok3 = (rs.standard_normal((5, 3, 128)) * 0.05).astype(np.float32)
    out = {"text": text, "tok3": tok3, "star": np.int64(star)}
    out["eti_train3"] = model.encode_text_img_train(torch.from_numpy(text), torch.from_numpy(tok3), split_ind=star,
                                                    repeat=False).numpy()
    ns = {"torch": torch, "np": np, "os": os, "F": F}
    path = os.path.join(REF, "eval_utils.py")
    G, Q, dim = 300, 40, 32
    gallery = O.l2_normalize(torch.from_numpy(rs.standard_normal((G, dim)).astype(np.float32)))
    names = [f"dress/img_{i:05d}.jpg" for i in range(G)]
    ans = rs.randint(0, G, size=Q)
    noise = torch.from_numpy(rs.standard_normal((Q, dim)).astype(np.float32)) * (2.7 / np.sqrt(dim))
    ref = O.l2_normalize(gallery[ans] + noise)
    out.update(gallery=gallery.numpy(), ref=ref.numpy(), answer_idx=ans)
    m = extract_function(path, "get_metrics_fashion", ns)(gallery, ref, names, [names[i] for i in ans])
    out.update({"fashion_" + k.replace("@", "_at_"): np.float64(v) for k, v in m.items()})
    # coco: Q paired (image, composed) features
    img = O.l2_normalize(torch.from_numpy(rs.standard_normal((60, dim)).astype(np.float32)))
    comp = O.l2_normalize(img + torch.from_numpy(rs.standard_normal((60, dim)).astype(np.float32)) * (2.0 / np.sqrt(dim)))
    m = extract_function(path, "get_metrics_coco", ns)(img, comp, torch.tensor(100.0))
    out.update(coco_image=img.numpy(), coco_ref=comp.numpy())
    out.update({"coco_" + k.replace("@", "_at_"): np.float64(v) for k, v in m.items()})
    # imgnet: 250 queries (3 reference batches of 100/100/50), 700 targets, 20 classes
    tl = rs.randint(0, 20, size=700)
    ql = rs.randint(0, 20, size=250)
    cent = rs.standard_normal((20, dim)).astype(np.float32)
    tf = O.l2_normalize(torch.from_numpy(cent[tl] + 1.2 * rs.standard_normal((700, dim)).astype(np.float32)))
    qf = O.l2_normalize(torch.from_numpy(cent[ql] + 1.2 * rs.standard_normal((250, dim)).astype(np.float32)))
    m = extract_function(path, "get_metrics_imgnet", ns)(qf, tf, torch.from_numpy(ql), torch.from_numpy(tl))
    out.update(imgnet_q=qf.numpy(), imgnet_t=tf.numpy(), imgnet_ql=ql, imgnet_tl=tl)
    out.update({"imgnet_" + k.replace("@", "_at_"): np.float64(float(v)) for k, v in m.items()})
    # cirr test submission: top-50 names with the reference image removed
    tnames = [f"test1-{i}-img0.png" for i in range(G)]
    refi = rs.randint(0, G, size=Q)
    res = extract_function(path, "get_cirr_testoutput", ns)(gallery, ref, np.array([tnames[i] for i in refi]),
                                                             np.array(tnames), torch.arange(1000, 1000 + Q))
    out["cirr_test_ref_idx"] = refi
    out["cirr_test_top50"] = np.array([[int(n.split("-")[1]) for n in res[str(1000 + i)]] for i in range(Q)], dtype=np.int64)
    assert res["version"] == "rc2" and res["metric"] == "recall"
    np.savez_compressed(os.path.join(OUT, "eval_glue.npz"), **out)
    print("eval_glue:", {k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()
                         if not k.startswith(("gallery", "ref", "coco_image", "coco_ref", "imgnet_q", "imgnet_t", "text", "tok3"))})


def mint_tokenizer():
    """(string -> ids) pairs from the reference's own tokenizer (src/third_party/open_clip/simple_tokenizer.py + clip.py
    tokenize), with an identity stand-in for the missing `ftfy` (a no-op on these clean strings)."""
    import importlib.util, json, types
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda t: t))
    base = os.path.join(os.path.dirname(REF), "src", "third_party", "open_clip") if not REF.endswith("src") else os.path.join(REF, "third_party", "open_clip")
    spec = importlib.util.spec_from_file_location("ref_simple_tokenizer", os.path.join(base, "simple_tokenizer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tk = mod.SimpleTokenizer()
    texts = ["*", "a photo of * , in the style of sketch", "A photo of a cat.", "remove the dog and add two red   balloons!",
             "it's the zebra's 3rd birthday -- isn't it?", "caf\u00e9 na\u00efve r\u00e9sum\u00e9 \u00fcber", "price: $1,234.50 (50% off) &amp; more",
             "the quick brown fox jumps over the lazy dog " * 12, "\u65e5\u672c\u8a9e\u306e\u30c6\u30b9\u30c8 emoji \U0001F600 ok", ""]
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    rows = []
    for t in texts:
        ids = [sot] + tk.encode(t) + [eot]
        if len(ids) > 77:
            ids = ids[:77]
            ids[-1] = eot
        rows.append(ids + [0] * (77 - len(ids)))
    json.dump({"texts": texts, "tokens": rows, "decoded": [tk.decode(r[1:r.index(eot)]) for r in rows]},
              open(os.path.join(OUT, "tokenizer.json"), "w"), ensure_ascii=True, indent=0)
    print("tokenizer:", len(texts), "strings; '*' ->", rows[0][:3])


def mint_keys():
    """state_dict key -> shape lists of the reference modules (the checkpoint contract)."""
    import json
    out = {}
    for tag, cfg in (("tiny", TINY), ("vitl14", VITL)):
        with torch.device("meta"):
            m = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"],
                     cfg["vision_patch_size"], cfg["context_length"], cfg["vocab_size"], cfg["transformer_width"],
                     cfg["transformer_width"] // 64, cfg["transformer_layers"])
        out[f"clip_{tag}"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    out["im2text"] = {k: list(v.shape) for k, v in IM2TEXT(768, 512, 768, 2).state_dict().items()}
    out["crossformer"] = {k: list(v.shape) for k, v in CrossFormer(768, 768, 768, num_layers=3).state_dict().items()}
    json.dump(out, open(os.path.join(OUT, "state_dict_keys.json"), "w"), indent=0, sort_keys=True)
    print("keys:", {k: len(v) for k, v in out.items()})



def _load_vitl(sd):
    model = CLIP(VITL["embed_dim"], VITL["image_resolution"], VITL["vision_layers"], VITL["vision_width"],
                 VITL["vision_patch_size"], VITL["context_length"], VITL["vocab_size"], VITL["transformer_width"],
                 VITL["transformer_width"] // 64, VITL["transformer_layers"]).eval().float()
    model.load_state_dict(sd, strict=True)
    return model


def mint_dual_full(batch=8, n_db=500000):
    """BASELINE config 4 at FULL size: ViT-L/14, d = 768, two seeded 0.5 M x 768 databases, two stream checkpoints;
    the per-batch body of evaluate_cirr (eval_utils.py:652-714) statement for statement on the reference's objects,
    with the reference's own get_retrieved_features (eval_utils.py:153-186) over an exact float64 flat index."""
    dim, middle, star = 768, 512, 265
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = _load_vitl(sd)
    ns = {"torch": torch, "np": np}
    grf = extract_function(os.path.join(REF, "eval_utils.py"), "get_retrieved_features", ns)
    image_base = O.synth_database(n_db, dim, seed=2002)
    text_base = O.synth_database(n_db, dim, seed=2003, clustered=True)
    ii, ti = FakeFlatL2(), FakeFlatL2()
    ii.add(image_base.numpy())
    ti.add(text_base.numpy())
    database = [image_base, text_base, None, ii, ti]

    def stream(seed):
        sds = (O.synth_im2text_state_dict(dim, middle, dim, 2, seed=seed, tag="i2t"),
               O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="fuse"),
               O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="cond"))
        a = IM2TEXT(embed_dim=dim, middle_dim=middle, output_dim=dim, n_layer=2).eval()
        b = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).eval()
        c = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).eval()
        a.load_state_dict(sds[0]); b.load_state_dict(sds[1]); c.load_state_dict(sds[2])
        return a, b, c

    img2text, retrieval_fuse, text_condition = stream(21)
    img2text_tb, retrieval_fuse_tb, text_condition_tb = stream(22)
    rs = np.random.RandomState(1001)
    timg = torch.from_numpy(rs.standard_normal((batch, 3, 224, 224)).astype(np.float32))
    ttxt = O.synth_tokens(batch, seed=4004)
    # ---- eval_utils.py:654-695 ----
    query_image_features = m.encode_image(timg)
    qfeat = query_image_features.clone()
    torch.manual_seed(1234)                      # the reference shuffles the K axis with randperm (eval_utils.py:174)
    topk_image, topk_text = grf(query_image_features, database, None)
    mapped_features = img2text(query_image_features)
    topk_image_features = img2text(topk_image)
    topk_text_features = img2text(topk_text)
    fused_features = retrieval_fuse(mapped_features.unsqueeze(1), topk_image_features, topk_image_features)
    text_conditioned = text_condition(mapped_features.unsqueeze(1), topk_text_features, topk_text_features)
    fused_features = torch.cat([fused_features, text_conditioned, mapped_features.unsqueeze(1)], dim=1)
    composed_feature = m.encode_text_img_retrieval(ttxt, fused_features, split_ind=star, repeat=False)
    mapped_features_tb = img2text_tb(query_image_features)
    topk_image_features_tb = img2text_tb(topk_image)
    topk_text_features_tb = img2text_tb(topk_text)
    fused_features_tb = retrieval_fuse_tb(mapped_features_tb.unsqueeze(1), topk_image_features_tb, topk_image_features_tb)
    text_conditioned_tb = text_condition_tb(mapped_features_tb.unsqueeze(1), topk_text_features_tb, topk_text_features_tb)
    fused_features_tb = torch.cat([fused_features_tb, text_conditioned_tb, mapped_features_tb.unsqueeze(1)], dim=1)
    composed_feature_tb = m.encode_text_img_retrieval(ttxt, fused_features_tb, split_ind=star, repeat=False)
    # ---- eval_utils.py:698-710 ----
    query_image_features = composed_feature_tb
    query_image_features = query_image_features / query_image_features.norm(dim=-1, keepdim=True)
    composed_feature = composed_feature / composed_feature.norm(dim=-1, keepdim=True)
    mixture_features = 0.5 * query_image_features + 0.5 * composed_feature
    mixture_features = mixture_features / mixture_features.norm(dim=-1, keepdim=True)
    qn = (qfeat / qfeat.norm(dim=-1, keepdim=True)).numpy()
    Di, Ii = ii.search(qn, 17)
    Dt, It = ti.search(qn, 17)
    np.savez_compressed(os.path.join(OUT, "dual_vitl14_full.npz"),
                        composed=composed_feature.numpy(), image=query_image_features.numpy(),
                        mixture=mixture_features.numpy(), query_image_features=qfeat.numpy(),
                        tokens_image_stream=fused_features.numpy(), tokens_text_stream=fused_features_tb.numpy(),
                        I_image=Ii, I_text=It, D_image=Di, D_text=Dt, n_db=np.int64(n_db), batch=np.int64(batch),
                        weights_checksum=np.float64(checksum(sd)))
    print("dual_vitl14_full: ok; top-16/17 gaps image", (Di[:, 16] - Di[:, 15]).min(), "text", (Dt[:, 16] - Dt[:, 15]).min())


def mint_train_step(tag, cfg, dim, middle, batch, n_db=512):
    """One training step of the knowledge modules through the reference's OWN loss function (src/trainer.py:44-165,
    `get_loss_img2text_image`, extracted with `ast`) and its own `get_retrieved_features` (:198-259), under a 1-rank gloo
    group with args.distributed = args.aggregate = True (the all_gather branch, :84-127), on the reference modules in
    training mode with dropout 0.  The one documented deviation: the reference's `get_text_features` builds a 78-token
    sequence and fails, so the name is bound to the reference model's evaluation splice
    `encode_text_img_retrieval("a photo of *", tokens)`.  Stored: inputs, the loss, and torch-autograd gradients of the
    reference modules' 54 parameters (whole tensors when small; sum / abs-sum / leading 256 elements otherwise)."""
    import torch.distributed as dist
    import torch.nn as nn
    sd_clip = O.synth_clip_state_dict(**cfg, seed=7)
    model = CLIP(**{**cfg, "transformer_heads": cfg["transformer_width"] // 64}).eval()
    model.load_state_dict(sd_clip, strict=True)
    sds = (O.synth_im2text_state_dict(dim, middle, dim, 2, seed=31, tag="i2t"),
           O.synth_crossformer_state_dict(dim, 3, seed=31, tag="fuse"),
           O.synth_crossformer_state_dict(dim, 3, seed=31, tag="cond"))
    img2text = IM2TEXT(embed_dim=dim, middle_dim=middle, output_dim=dim, n_layer=2, dropout=0.0).train()
    retrieval_fuse = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).train()
    text_condition = CrossFormer(q_dim=dim, k_dim=dim, v_dim=dim, num_layers=3).train()
    img2text.load_state_dict(sds[0]); retrieval_fuse.load_state_dict(sds[1]); text_condition.load_state_dict(sds[2])
    image_base = O.synth_database(n_db, dim, seed=2002)
    text_base = O.synth_database(n_db, dim, seed=2003)
    ii, ti = FakeFlatL2(), FakeFlatL2()
    ii.add(image_base.numpy()); ti.add(text_base.numpy())
    database = [image_base, text_base, [str(i) for i in range(n_db)], ii, ti]
    rs = np.random.RandomState(77)
    image_features = torch.from_numpy((image_base[rs.randint(0, n_db, size=batch)].numpy()
                                       + 0.3 * rs.standard_normal((batch, dim)) / np.sqrt(dim)).astype(np.float32) * 3.0)
    caps = torch.zeros(batch, dim)                                   # ori_cap_feature: read, never used in this branch
    star = 265
    text = torch.zeros(77, dtype=torch.long)
    text[:6] = torch.tensor([49406 if cfg["vocab_size"] > 49406 else cfg["vocab_size"] - 2, 320, 1125, 539, star,
                             cfg["vocab_size"] - 1])             # <sot> a photo of * <eot>
    text[1:4] = text[1:4] % (cfg["vocab_size"] - 2)

    def get_text_features(model_, token_features, args_):            # the documented deviation (evaluation splice)
        return model_.encode_text_img_retrieval(text[None, :].repeat(token_features.size(0), 1), token_features,
                                                split_ind=star, repeat=False)

    ns = {"torch": torch, "np": np, "dist": dist, "get_text_features": get_text_features}
    ns["get_retrieved_features"] = extract_function(os.path.join(REF, "trainer.py"), "get_retrieved_features", ns)
    loss_fn = extract_function(os.path.join(REF, "trainer.py"), "get_loss_img2text_image", ns)

    class Args:
        distributed, aggregate, gpu = True, True, None
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                    # trainer.py:56-57 move tensors to args.gpu unconditionally
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 1000), rank=0, world_size=1)
    try:
        torch.manual_seed(5)                                          # (the randperm of the neighbour axis, :216-217: a numerical no-op)
        with torch.enable_grad():
            loss = loss_fn(model, img2text, retrieval_fuse, text_condition, image_features, (caps, None, None),
                           nn.CrossEntropyLoss(), nn.CrossEntropyLoss(), None, Args(), database)
            params = [("i2t." + k, v) for k, v in img2text.named_parameters()] + \
                     [("fuse." + k, v) for k, v in retrieval_fuse.named_parameters()] + \
                     [("cond." + k, v) for k, v in text_condition.named_parameters()]
            grads = torch.autograd.grad(loss, [v for _, v in params])
    finally:
        dist.destroy_process_group()
        torch.Tensor.cuda = real_cuda
    out = {"image_features": image_features.numpy(), "text": text.numpy(), "n_db": np.int64(n_db), "star": np.int64(star),
           "loss": np.float64(loss.item()), "weights_checksum": np.float64(checksum(sd_clip) + sum(checksum(x) for x in sds)),
           "names": np.array([k for k, _ in params])}
    for (k, _), g in zip(params, grads):
        g = g.detach()
        if g.numel() <= 4096:
            out["g." + k] = g.numpy()
        else:
            out["gs." + k] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
            out["gh." + k] = g.reshape(-1)[:256].numpy()
    np.savez_compressed(os.path.join(OUT, f"train_step_{tag}.npz"), **out)
    print(f"train_step_{tag}: loss {loss.item():.6f}, {len(params)} gradients")


def mint_heavy_tail(batch=2):
    """ViT-L/14 + text tower with heavy-tailed activations (massive channels of 50-200 sigma in the residual stream,
    as real CLIP checkpoints have): the reference's fp32 outputs and per-block CLS rows."""
    sd = O.make_heavy_tailed(O.synth_clip_state_dict(**VITL, seed=7))
    m = _load_vitl(sd)
    rs = np.random.RandomState(1001)
    image = rs.standard_normal((batch, 3, 224, 224)).astype(np.float32)
    text = O.synth_tokens(batch, seed=4004).numpy()
    feat, mids = m.encode_image(torch.from_numpy(image), mid_feature=True)
    mids = [x.float() for x in mids]
    stats = np.stack([np.array([float((x.mean(-1).abs() / x.std(-1)).max()), float(x.abs().max()),
                                float((x.abs().amax(-1) / x.std(-1)).max())]) for x in mids])
    tfeat = m.encode_text(torch.from_numpy(text))
    np.savez_compressed(os.path.join(OUT, "clip_vitl14_heavy.npz"), encode_image=feat.numpy(),
                        block_cls=np.stack([x[:, 0, :].numpy() for x in mids]), encode_text=tfeat.numpy(),
                        block_stats=stats, weights_checksum=np.float64(checksum(sd)))
    print("heavy tail: per-block [max |mean|/std, max |x|, max |x|/std]:\n", np.round(stats, 2))


def mint_recall_vitl(n_gallery=1000, n_query=256, chunk=8):
    """BASELINE config 1 at its stated size: ViT-L/14 features of a 1 k-image gallery and 256 queries (noisy copies of
    gallery images at graded noise levels), ranked by the reference's own get_metrics_cirr (eval_utils.py:1040-1067).
    Weights: the seeded ViT-L/14 with sharpened attention / residual branches (oracle.sharpen_clip: unrelated images at
    cosine ~0.9 instead of the 0.995 of plain random init, while the network stays well conditioned)."""
    sd = O.sharpen_clip(O.synth_clip_state_dict(**VITL, seed=7))
    m = _load_vitl(sd)
    tgt_idx, ref_idx, sigma = O.synth_recall_plan(n_gallery, n_query)

    def enc(make, n):
        out = []
        for i in range(0, n, chunk):
            out.append(m.encode_image(make(i, min(chunk, n - i))))
            if (i // chunk) % 8 == 0:
                print("  encoded", i + chunk, "/", n, flush=True)
        f = torch.cat(out)
        return f / f.norm(dim=-1, keepdim=True)

    gal = enc(lambda i, c: O.synth_gallery_images(c, start=i), n_gallery)
    qf = enc(lambda i, c: O.synth_recall_queries(tgt_idx, sigma, start=i, count=c), n_query)
    ns = {"torch": torch, "np": np, "os": os}
    gm = extract_function(os.path.join(REF, "eval_utils.py"), "get_metrics_cirr", ns)
    index_names = [f"/data/cirr/dev/img_{i:05d}.png" for i in range(n_gallery)]
    reference_names = [os.path.basename(index_names[i]) for i in ref_idx]
    target_names = [os.path.basename(index_names[i]) for i in tgt_idx]
    metrics = gm(gal, qf, np.array(reference_names), np.array(index_names), np.array(target_names))
    np.savez_compressed(os.path.join(OUT, "recall_vitl14.npz"), gallery=gal.numpy().astype(np.float32),
                        query=qf.numpy().astype(np.float32), ref_idx=ref_idx, tgt_idx=tgt_idx, sigma=sigma,
                        weights_checksum=np.float64(checksum(sd)),
                        **{k.replace("@", "_at_"): np.float64(v) for k, v in metrics.items()})
    print("recall_vitl14:", metrics)


TINY = dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
            context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2)
VITL = dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
            context_length=77, vocab_size=49408, transformer_width=768, transformer_layers=12)

if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "knowledge", "cirr", "search", "metrics", "glue", "tokenizer", "keys", "vitl"]
    if "glue" in which:
        mint_eval_glue()
    if "tokenizer" in which:
        mint_tokenizer()
    if "keys" in which:
        mint_keys()
    model = None
    if "tiny" in which or "cirr" in which:
        model, sd, timg, ttxt = mint_clip("tiny", TINY, batch=4, star=265)
    if "knowledge" in which:
        mint_knowledge(128, 128)
        mint_knowledge(768, 512, batch=3)
    if "cirr" in which:
        mint_cirr_batch(model, sd, timg, ttxt, 265, 128, 128)
    if "search" in which:
        mint_search()
    if "metrics" in which:
        mint_metrics()
    if "vitl" in which:
        mint_clip("vitl14", VITL, batch=2, star=265)
    if "train" in which:
        torch.set_grad_enabled(True)
        mint_train_step("tiny", TINY, 128, 128, batch=6)
        # text tower of ViT-L/14 (12 x 768) with a one-block stand-in for the visual tower, which the loss never runs
        mint_train_step("vitl_text", {**VITL, "image_resolution": 56, "vision_layers": 1, "vision_width": 128}, 768, 512, batch=4)
        torch.set_grad_enabled(False)
    # full-size fixtures (minutes of CPU each; not in the default list)
    if "heavy" in which:
        mint_heavy_tail()
    if "dual_full" in which:
        mint_dual_full()
    if "recall_vitl" in which:
        mint_recall_vitl()
