"""GPU parity of the whole towers / knowledge path behind the reference's model API, against the golden
vectors minted from the reference (tests/golden, tools/mint_golden.py) and against the oracle.

Tolerance (north_star: "within a stated fp tolerance of the reference CPU path"): the HIP path rounds
GEMM operands to bf16 (8 significant bits) and keeps fp32 accumulators / residual stream / LayerNorm /
softmax statistics.  Stated bar on the final embeddings: cosine >= 0.9999 per row and rel-L2 <= 1.5e-2
(measured on MI355X: cosine >= 0.99995, rel-L2 <= 9e-3)
against the fp32 reference; Recall@k on synthetic retrieval problems identical (test_gpu_search,
test_recall_equal_to_cpu_reference).
"""
import os

import numpy as np
import pytest
import torch

import keds_amd
from oracle import keds_oracle as O
from tests.conftest import golden_path
from tests.gpu_util import assert_parity, max_abs, min_cosine, rel_l2, report

pytestmark = pytest.mark.gpu

TINY = dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
            context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2)
VITL = dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
            context_length=77, vocab_size=49408, transformer_width=768, transformer_layers=12)
COS_MIN, REL_MAX = 0.9999, 1.5e-2


def _assert_close(name, got, want, cos_min=None, rel_max=None):
    """Class ceiling (image 6e-3 / text 1.1e-2 / composed 1.3e-2 rel-L2, cosine >= 0.99994-0.99995) AND at most twice the
    error measured on the committed build (tests/golden/parity_baseline.json)."""
    assert_parity(name, got, want, cos_min, rel_max)


@pytest.fixture(scope="module")
def tiny_model():
    g = dict(np.load(golden_path("clip_tiny.npz")))
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    m = keds_amd.build_model(dict(sd), fp16=False).cuda()
    return g, sd, m


def test_tiny_encode_image(tiny_model):
    g, sd, m = tiny_model
    out = m.encode_image(torch.from_numpy(g["image"]).cuda())
    assert out.shape == (4, 128) and out.dtype == torch.float32
    _assert_close("tiny.encode_image", out, g["encode_image"])
    n = m.encode_image(torch.from_numpy(g["image"]).cuda(), normalize=True)
    _assert_close("tiny.encode_image.normalized", n, g["forward_image"])


def test_tiny_tower_blocks(tiny_model):
    """Residual stream after each block against the reference's mid features (model.py:337-342)."""
    import ctypes as C
    from keds_amd import _lib, ops
    g, sd, m = tiny_model
    eng = m._engine()
    img = torch.from_numpy(g["image"]).cuda()
    B, S, w = 4, 17, 128
    col = ops.im2col(img, 14, eng.kpad)
    x = torch.zeros((128, w), device="cuda")
    ops.gemm_bt(col, eng.keep[0]["conv_w"], None, _lib.EPI_PATCH_F32, out=x, m=B * 16, aux=eng.keep[0]["pos_emb"], aux_i=16)
    x = x[: B * S].reshape(B, S, w)
    x[:, 0] = eng.keep[0]["class_emb"] + eng.keep[0]["pos_emb"][0]
    ref0 = O.patch_embed(sd, torch.from_numpy(g["image"]))
    _assert_close("tiny.patch_embed", x, ref0, rel_max=1e-2)
    xl = ops.layernorm(x.reshape(B * S, w).contiguous(), eng.keep[0]["ln_pre_g"], eng.keep[0]["ln_pre_b"], out_f32=True)
    xp = torch.zeros((128, w), device="cuda")
    xp[: B * S] = xl
    lib = _lib.load()
    for layer in range(2):
        one = _lib.TowerParams(w, 1, 2, S, 0, C.cast(C.byref(eng.vit.tower.blocks[layer]), C.POINTER(_lib.BlockParams)))
        nbytes = lib.keds_tower_workspace_bytes(w, S, B)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        _lib.check(lib.keds_tower_forward(C.byref(one), _lib.ptr(xp), B, _lib.ptr(ws), nbytes, _lib.stream()), "tower")
        _assert_close(f"tiny.block{layer}", xp[: B * S].reshape(B, S, w), g["block_tokens"][layer], rel_max=2e-2)


def test_tiny_text_paths(tiny_model):
    g, sd, m = tiny_model
    text = torch.from_numpy(g["text"]).cuda()
    star = int(g["star"])
    _assert_close("tiny.encode_text", m.encode_text(text), g["encode_text"])
    _assert_close("tiny.eti3", m.encode_text_img_retrieval(text, torch.from_numpy(g["tok3"]).cuda(), split_ind=star,
                                                            repeat=False), g["eti3"])
    _assert_close("tiny.eti2", m.encode_text_img_retrieval(text, torch.from_numpy(g["tok2"]).cuda(), split_ind=star,
                                                            repeat=False), g["eti2"])
    _assert_close("tiny.eti3_repeat", m.encode_text_img_retrieval(text[:1], torch.from_numpy(g["tok3"]).cuda(),
                                                                   split_ind=star, repeat=True), g["eti3_repeat"])
    i_n, t_n, scale = m(torch.from_numpy(g["image"]).cuda(), text)
    _assert_close("tiny.forward.text", t_n, g["forward_text"])
    assert abs(float(scale.detach()) - float(g["forward_scale"])) < 1e-3
    # CPU token tensors are accepted as well (the reference's loaders hand over CPU tensors)
    _assert_close("tiny.encode_text.cpu_tokens", m.encode_text(torch.from_numpy(g["text"])), g["encode_text"])


def test_tiny_ragged_batches(tiny_model):
    """Batches that do not fill a 128-row GEMM tile, and one that spans tiles."""
    g, sd, m = tiny_model
    rs = np.random.RandomState(5)
    for B in (1, 3, 9):
        img = torch.from_numpy(rs.standard_normal((B, 3, 56, 56)).astype(np.float32))
        _assert_close(f"tiny.encode_image.B{B}", m.encode_image(img.cuda()), O.encode_image(sd, img))
    text = torch.from_numpy(g["text"])[:3]
    _assert_close("tiny.encode_text.B3", m.encode_text(text.cuda()), O.encode_text(sd, text))


@pytest.mark.parametrize("dim,middle", [(128, 128), (768, 512)])
def test_knowledge_modules(dim, middle):
    g = dict(np.load(golden_path(f"knowledge_d{dim}.npz")))
    i2t = keds_amd.IM2TEXT(dim, middle, dim, 2).eval()
    i2t.load_state_dict(O.synth_im2text_state_dict(dim, middle, dim, 2, seed=11, tag="i2t"))
    xf = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    xf.load_state_dict(O.synth_crossformer_state_dict(dim, 3, seed=12, tag="fuse"))
    i2t, xf = i2t.cuda(), xf.cuda()
    y = i2t(torch.from_numpy(g["x"]).cuda())
    ynb = i2t(torch.from_numpy(g["nb"]).cuda())                    # [B,16,dim]: leading dims are flattened
    _assert_close(f"im2text.d{dim}", y, g["im2text_x"], rel_max=1e-2)
    _assert_close(f"im2text_nb.d{dim}", ynb, g["im2text_nb"], rel_max=1e-2)
    z = xf(torch.from_numpy(g["im2text_x"]).cuda().unsqueeze(1), torch.from_numpy(g["im2text_nb"]).cuda(),
           torch.from_numpy(g["im2text_nb"]).cuda())
    assert z.shape == (g["x"].shape[0], 1, dim)
    _assert_close(f"crossformer.d{dim}", z, g["crossformer"], rel_max=2e-2)


def _streams(dim, middle, seed):
    a = keds_amd.IM2TEXT(dim, middle, dim, 2).eval()
    b = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    c = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    a.load_state_dict(O.synth_im2text_state_dict(dim, middle, dim, 2, seed=seed, tag="i2t"))
    b.load_state_dict(O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="fuse"))
    c.load_state_dict(O.synth_crossformer_state_dict(dim, 3, seed=seed, tag="cond"))
    return keds_amd.KnowledgeStream(a.cuda(), b.cuda(), c.cuda())


def test_cirr_batch_composition_tiny(tiny_model):
    """evaluate_cirr per-batch body end to end (eval_utils.py:652-714) against the reference's outputs."""
    gt, sd, m = tiny_model
    g = dict(np.load(golden_path("cirr_batch_tiny.npz")))
    n_db = int(g["n_db"])
    image_base = O.synth_database(n_db, 128, seed=2002)
    text_base = O.synth_database(n_db, 128, seed=2003)
    database = keds_amd.build_database(image_base, text_base, [str(i) for i in range(n_db)])
    out = keds_amd.compose_query_features(m, _streams(128, 128, 21), _streams(128, 128, 22),
                                          torch.from_numpy(gt["image"]).cuda(), torch.from_numpy(gt["text"]).cuda(),
                                          database, id_split=265)
    _assert_close("cirr.tokens_image_stream", out["tokens_image_stream"], g["tokens_image_stream"], rel_max=5e-2)
    _assert_close("cirr.tokens_text_stream", out["tokens_text_stream"], g["tokens_text_stream"], rel_max=5e-2)
    _assert_close("cirr.composed", out["composed"], g["composed"])
    _assert_close("cirr.image", out["image"], g["image"])
    _assert_close("cirr.mixture", out["mixture"], g["mixture"])
    # retrieved neighbours: same rows as the reference retrieved (sorted along K: the reference shuffles)
    ti, tt = keds_amd.get_retrieved_features(out["query_image_features"], database)
    want_i = g["topk_image_sorted"]
    got_i = np.sort(ti.cpu().numpy(), axis=1)
    # bf16 encoder noise may swap a near-tie at the list boundary; require >= 15 of 16 rows identical
    same = (got_i == want_i).all(axis=2).sum(axis=1)
    report("cirr.neighbour_rows_identical", min_rows=int(same.min()))
    assert int(same.min()) >= 15


def test_vitl14_full_size_against_reference_golden():
    """ViT-L/14 (24 x 1024) + 12-layer text tower at B=2 against the reference's own fp32 outputs."""
    g = dict(np.load(golden_path("clip_vitl14.npz")))
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda()
    del sd
    img = torch.from_numpy(g["image"]).cuda()
    text = torch.from_numpy(g["text"]).cuda()
    _assert_close("vitl.encode_image", m.encode_image(img), g["encode_image"])
    _assert_close("vitl.encode_text", m.encode_text(text), g["encode_text"])
    _assert_close("vitl.eti3", m.encode_text_img_retrieval(text, torch.from_numpy(g["tok3"]).cuda(), split_ind=265,
                                                            repeat=False), g["eti3"])
    _assert_close("vitl.eti2", m.encode_text_img_retrieval(text, torch.from_numpy(g["tok2"]).cuda(), split_ind=265,
                                                            repeat=False), g["eti2"])
    # batch-size independence: B=2 rows must reappear inside a B=130 batch (spans two GEMM row tiles per 128)
    big = torch.cat([img, torch.from_numpy(O.synth_tensor("imgs", [128, 3, 224, 224], 1.0).numpy()).cuda()])
    out = m.encode_image(big)
    _assert_close("vitl.encode_image.in_B130", out[:2], g["encode_image"])
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_side_lane_for_remainder_rows_changes_no_bit(precision):
    """At B = 128 the ViT-L/14 tower has 128 full 256-row tiles + 128 remainder rows; the remainder chain runs on a second
    stream beside the full tiles (towers.hip, RowLanes).  Same kernels on the same rows: the embeddings must be the very
    same bits with the lane on and off, run after run (a missing fork / join would show up here as a race)."""
    from keds_amd import _lib
    lib = _lib.load()
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda()
    m.set_precision(precision)
    del sd
    img = torch.from_numpy(O.synth_tensor("imgs", [128, 3, 224, 224], 1.0).numpy()).cuda()
    try:
        lib.keds_tower_fill_enable(0)      # (the filler-row experiment of round 3, off by default: next test)
        lib.keds_side_lane_enable(1)
        assert lib.keds_tower_side_rows(1024, 257, 128, int(precision == "fp8")) == 128
        on = [m.encode_image(img).clone() for _ in range(3)]
        lib.keds_side_lane_enable(0)
        assert lib.keds_tower_side_rows(1024, 257, 128, int(precision == "fp8")) == 0
        off = m.encode_image(img).clone()
    finally:
        lib.keds_side_lane_enable(1)
    assert torch.isfinite(off).all()
    for o in on:
        assert torch.equal(o, off)


def test_ragged_row_tile_on_filler_rows_matches_the_two_lane_tower():
    """Round 3 experiment (KEDS_TOWER_FILL=1; off by default because every GEMM launch then pays a nearly empty extra round of
    workgroups): at B = 128 the 128 remainder rows of a bf16 ViT-L/14 tower run as a 129th FULL row tile on 128 filler rows
    (copies of the first rows; never read by the attention or the read-out) instead of as a chain of small GEMMs on the side
    lane.  Rows are independent outside the attention, so every real row must come out as before -- the same MFMA chains in
    another kernel: equal bits are expected and reported, closeness is asserted -- and run after run the same."""
    from keds_amd import _lib
    lib = _lib.load()
    if "KEDS_EXPERIMENTS" not in _lib.build_flags():
        assert lib.keds_tower_fill_enable(1) != 0            # the product library says so instead of silently ignoring the request
        pytest.skip("the filler-row tower lost its A/B and is not in the product library (experiment build only)")
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda()
    del sd
    base = torch.from_numpy(O.synth_tensor("imgs", [200, 3, 224, 224], 1.0).numpy()).cuda()
    try:
        for B in (128, 129, 200, 64):
            lib.keds_tower_fill_enable(1)
            filled = lib.keds_tower_side_rows(1024, 257, B, 0) == 0 and (B * 257) % 256 != 0
            a = m.encode_image(base[:B]).clone()
            a2 = m.encode_image(base[:B]).clone()
            lib.keds_tower_fill_enable(0)
            b = m.encode_image(base[:B]).clone()
            assert torch.equal(a, a2) and torch.isfinite(a).all()
            cos = float(torch.nn.functional.cosine_similarity(a, b).min().item())
            report("tower_fill_vs_two_lanes", B=B, filler_rows_used=bool(filled), bit_equal=bool(torch.equal(a, b)), min_cosine=cos)
            assert cos >= 0.99999, (B, cos)
    finally:
        lib.keds_tower_fill_enable(0)


def test_batch_size_sweep_lane_split_and_dispatch_boundaries():
    """ViT-L/14 at batch sizes that land on every dispatch rule of the tower (128^2 / 256^2 tiles, rows split over the two
    lanes or not, fp8 main rows + fp16 remainder rows): lane on == lane off bit for bit, finite, and image 0's embedding
    independent of the batch it travels in (different kernels per batch size: close, not bit-equal)."""
    from keds_amd import _lib
    lib = _lib.load()
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda()
    del sd
    base = torch.from_numpy(O.synth_tensor("imgs", [200, 3, 224, 224], 1.0).numpy()).cuda()
    try:
        lib.keds_tower_fill_enable(0)       # the two-lane dispatch rules are what this sweep is about
        for precision, cos_min in (("bf16", 0.9999), ("fp8", 0.995)):
            m.set_precision(precision)
            first, splits = None, set()
            for B in (1, 2, 33, 64, 127, 128, 129, 200):
                lib.keds_side_lane_enable(1)
                on = m.encode_image(base[:B]).clone()
                splits.add(lib.keds_tower_side_rows(1024, 257, B, int(precision == "fp8")) > 0)
                lib.keds_side_lane_enable(0)
                off = m.encode_image(base[:B]).clone()
                assert torch.equal(on, off) and torch.isfinite(on).all(), (precision, B)
                first = on[:1].clone() if first is None else first
                c = float(torch.nn.functional.cosine_similarity(on[:1], first).item())
                assert c >= cos_min, (precision, B, c)
            assert splits == {True, False} or precision == "fp8"
    finally:
        lib.keds_side_lane_enable(1)


def test_side_lane_under_foreign_stream_and_graph_capture():
    """The two-lane tower pass forks from / joins to whatever stream the caller is on: same bits on a non-default torch
    stream, and the whole encode_image can be captured into a graph (the side stream joins the capture through its fork
    event) and replayed on new inputs."""
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda()
    del sd
    img = torch.from_numpy(O.synth_tensor("imgs", [128, 3, 224, 224], 1.0).numpy()).cuda()
    ref = m.encode_image(img).clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        other = m.encode_image(img).clone()
    s.synchronize()
    assert torch.equal(other, ref)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.encode_image(img)
    img2 = torch.from_numpy(O.synth_tensor("imgs2", [128, 3, 224, 224], 1.0).numpy()).cuda()
    img.copy_(img2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, m.encode_image(img2))


def test_recall_equal_to_cpu_reference():
    """Recall@1/5/10 of a synthetic retrieval problem: GPU features vs oracle features, same ranking metric.
    Queries are encoded images; the gallery holds the oracle's embeddings of the same images plus
    distractors, so ground truth is known and both feature sets must reach the same recall."""
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    m = keds_amd.build_model(dict(sd), fp16=False).cuda()
    rs = np.random.RandomState(9)
    base = rs.standard_normal((24, 3, 56, 56)).astype(np.float32)
    gallery_img = torch.from_numpy(np.concatenate([base, rs.standard_normal((40, 3, 56, 56)).astype(np.float32)]))
    query_img = torch.from_numpy(base + 0.3 * rs.standard_normal(base.shape).astype(np.float32))
    names = [f"g/{i}.png" for i in range(64)]
    tgt = [f"{i}.png" for i in range(24)]
    ref = [f"{40 + i}.png" for i in range(24)]                      # a distractor plays the reference image
    go, qo = O.l2_normalize(O.encode_image(sd, gallery_img)), O.l2_normalize(O.encode_image(sd, query_img))
    gg = m.encode_image(gallery_img.cuda(), normalize=True)
    qg = m.encode_image(query_img.cuda(), normalize=True)
    mo = O.get_metrics_cirr(go, qo, ref, names, tgt)
    mg = keds_amd.get_metrics_cirr(gg, qg, ref, names, tgt)
    report("recall_equal", **{k.replace("@", "_at_"): v for k, v in mg.items()})
    assert mo == mg


# ---- the other eval drivers' glue on the same kernels ---------------------------------
def test_encode_text_img_train_fashion_splice(tiny_model):
    """model.py:853-892 against the reference's own output (tests/golden/eval_glue.npz)."""
    gt, sd, m = tiny_model
    g = dict(np.load(golden_path("eval_glue.npz")))
    text, tok3, star = torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["tok3"]).cuda(), int(g["star"])
    _assert_close("glue.eti_train3", m.encode_text_img_train(text, tok3, split_ind=star, repeat=False), g["eti_train3"])
    with pytest.raises(RuntimeError):                 # 2 tokens: the reference fails with a size mismatch (:883)
        m.encode_text_img_train(text, tok3[:, :2], split_ind=star)
    with pytest.raises(IndexError):
        m.encode_text_img_train(text, tok3, split_ind=499)


def test_metrics_of_the_other_drivers_match_reference_values():
    """get_metrics_fashion / _coco / _imgnet and get_cirr_testoutput (eval_utils.py:1008-1134) with ranking and
    counting on the GPU: the reference's own numbers for the same inputs."""
    g = dict(np.load(golden_path("eval_glue.npz")))
    gallery, ref = torch.from_numpy(g["gallery"]).cuda(), torch.from_numpy(g["ref"]).cuda()
    names = [f"dress/img_{i:05d}.jpg" for i in range(gallery.shape[0])]
    m = keds_amd.get_metrics_fashion(gallery, ref, names, [names[i] for i in g["answer_idx"]])
    for k in (1, 5, 10, 50, 100):
        assert abs(m[f"R@{k}"] - float(g[f"fashion_R_at_{k}"])) < 1e-4
    with pytest.raises(AssertionError):
        keds_amd.get_metrics_fashion(gallery, ref, names, ["missing.jpg"] * ref.shape[0])
    m = keds_amd.get_metrics_coco(torch.from_numpy(g["coco_image"]).cuda(), torch.from_numpy(g["coco_ref"]).cuda(),
                                  torch.tensor(100.0))
    assert len(m) == 14
    for key, v in m.items():
        assert abs(v - float(g["coco_" + key.replace("@", "_at_")])) < 1e-6, key
    m = keds_amd.get_metrics_imgnet(torch.from_numpy(g["imgnet_q"]).cuda(), torch.from_numpy(g["imgnet_t"]).cuda(),
                                    torch.from_numpy(g["imgnet_ql"]), torch.from_numpy(g["imgnet_tl"]))
    assert len(m) == 12
    for key, v in m.items():
        assert abs(v - float(g["imgnet_" + key.replace("@", "_at_")])) < 2e-6, key
    tnames = [f"test1-{i}-img0.png" for i in range(gallery.shape[0])]
    res = keds_amd.get_cirr_testoutput(gallery, ref, [tnames[i] for i in g["cirr_test_ref_idx"]], tnames,
                                       torch.arange(1000, 1000 + ref.shape[0]))
    assert res["version"] == "rc2" and res["metric"] == "recall" and len(res) == 2 + ref.shape[0]
    for i in range(ref.shape[0]):
        assert [int(n.split("-")[1]) for n in res[str(1000 + i)]] == g["cirr_test_top50"][i].tolist()


def test_imgnet_and_coco_batch_bodies_against_oracle(tiny_model):
    """evaluate_imgnet_retrieval (one prompt row, repeat=True, w = 0.1 j) and evaluate_coco (w = 0.05 j) run the CIRR
    body with other arguments (eval_utils.py:372-415, 511-548)."""
    gt, sd, m = tiny_model
    n_db = 4096
    image_base, text_base = O.synth_database(n_db, 128, seed=2002), O.synth_database(n_db, 128, seed=2003)
    database = keds_amd.build_database(image_base, text_base, [str(i) for i in range(n_db)])

    def sds(seed):
        return (O.synth_im2text_state_dict(128, 128, 128, 2, seed=seed, tag="i2t"),
                O.synth_crossformer_state_dict(128, 3, seed=seed, tag="fuse"),
                O.synth_crossformer_state_dict(128, 3, seed=seed, tag="cond"))
    img, text = torch.from_numpy(gt["image"]), torch.from_numpy(gt["text"])
    for name, kw, txt in (("imgnet", dict(repeat=True, w_text_stream=0.3), text[:1]),
                          ("coco", dict(repeat=False, w_text_stream=0.15), text)):
        want = O.compose_query(sd, sds(21), sds(22), img, txt, image_base, text_base, split_ind=265, **kw)
        got = keds_amd.compose_query_features(m, _streams(128, 128, 21), _streams(128, 128, 22), img.cuda(), txt.cuda(),
                                              database, id_split=265, **kw)
        for key in ("composed", "image", "mixture"):
            _assert_close(f"glue.{name}.{key}", got[key], want[key])


def test_deterministic_switch_gives_bitwise_reproducible_embeddings(monkeypatch):
    """KEDS_DETERMINISTIC=1 keeps the separate LayerNorm kernels (no fp32 atomics): two runs are the same bits, and the
    result agrees with the default (LayerNorm folded into the GEMMs) path within the usual tolerance."""
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    g = dict(np.load(golden_path("clip_tiny.npz")))
    img = torch.from_numpy(np.random.RandomState(3).standard_normal((40, 3, 56, 56)).astype(np.float32)).cuda()
    fast = keds_amd.build_model(dict(sd), fp16=False).cuda().encode_image(img)
    monkeypatch.setenv("KEDS_DETERMINISTIC", "1")
    m = keds_amd.build_model(dict(sd), fp16=False).cuda()
    a, b = m.encode_image(img), m.encode_image(img)
    assert torch.equal(a, b)
    assert m._engine().vit.tower.blocks[0].qkv_wf is None
    _assert_close("deterministic.vs_folded", a, fast, cos_min=0.99995, rel_max=1e-2)
    _assert_close("deterministic.golden", m.encode_image(torch.from_numpy(g["image"]).cuda()), g["encode_image"])


def test_user_style_inputs_cpu_features_and_other_neighbour_counts():
    """Call shapes a reference user produces: metric functions fed CPU feature tensors (the drivers .cpu() their features,
    eval_utils.py:549-553), CrossFormer with neighbour counts other than 16, half-precision images."""
    g = dict(np.load(golden_path("metrics_cirr.npz")))
    G = g["gallery"].shape[0]
    index_names = [f"/data/cirr/dev/img_{i:05d}.png" for i in range(G)]
    ref_names = [os.path.basename(index_names[i]) for i in g["ref_idx"]]
    tgt_names = [os.path.basename(index_names[i]) for i in g["tgt_idx"]]
    m = keds_amd.get_metrics_cirr(torch.from_numpy(g["gallery"]), torch.from_numpy(g["ref"]), ref_names, index_names, tgt_names)
    for k in (1, 5, 10, 50, 100):
        assert abs(m[f"recall_R@{k}"] - float(g[f"recall_R_at_{k}"])) < 1e-4
    e = dict(np.load(golden_path("eval_glue.npz")))
    mc = keds_amd.get_metrics_coco(torch.from_numpy(e["coco_image"]), torch.from_numpy(e["coco_ref"]), torch.tensor(100.0))
    assert abs(mc["image_to_ref_R@1"] - float(e["coco_image_to_ref_R_at_1"])) < 1e-6
    dim = 128
    fuse = O.synth_crossformer_state_dict(dim, 3, seed=5, tag="fuse")
    xf = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    xf.load_state_dict(fuse)
    xf = xf.cuda()
    for K in (1, 5, 32):
        q = O.synth_database(4, dim, seed=K)
        kv = O.synth_database(4 * K, dim, seed=K + 1).reshape(4, K, dim)
        got = xf(q.cuda().unsqueeze(1), kv.cuda(), kv.cuda())
        want = O.crossformer(fuse, q.unsqueeze(1), kv, kv)
        _assert_close(f"crossformer.K{K}", got, want, rel_max=2e-2)
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    mm = keds_amd.build_model(dict(sd), fp16=False).cuda()
    img = torch.from_numpy(dict(np.load(golden_path("clip_tiny.npz")))["image"])
    _assert_close("tiny.encode_image.half_input", mm.encode_image(img.cuda().half()), O.encode_image(sd, img.half().float()))


def test_default_fp16_checkpoint_model_and_empty_batches():
    """build_model(state_dict) with the reference's default fp16 conversion (model.py:988): half outputs, results within the
    tolerance of the fp16-rounded weights; empty batches return empty tensors like the reference."""
    g = dict(np.load(golden_path("clip_tiny.npz")))
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    m = keds_amd.build_model(dict(sd)).cuda()                      # fp16=True default
    assert m.dtype == torch.float16
    img = torch.from_numpy(g["image"]).cuda()
    out = m.encode_image(img.half())
    assert out.dtype == torch.float16
    _assert_close("tiny.fp16_model.encode_image", out, g["encode_image"], cos_min=0.9995, rel_max=4e-2)
    txt = m.encode_text(torch.from_numpy(g["text"]).cuda())
    assert txt.dtype == torch.float16
    _assert_close("tiny.fp16_model.encode_text", txt, g["encode_text"], cos_min=0.9995, rel_max=4e-2)
    assert tuple(m.encode_image(img[:0]).shape) == (0, 128)
    assert tuple(m.encode_text(torch.from_numpy(g["text"][:0]).cuda()).shape) == (0, 128)


def _ragged_tokens(B, L, eots, end_id, star, vocab, seed=11):
    """[B, L] token rows: start token, random ids, the split token `star` at column 3, ONE EOT at column eots[b], zeros behind."""
    rs = np.random.RandomState(seed)
    t = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        e = int(eots[b % len(eots)])
        t[b, :e] = rs.randint(1, min(vocab, 40000) - 2, size=e)
        t[b, t[b] == star] = star + 1
        t[b, 0] = end_id - 1
        if e > 3:
            t[b, 3] = star
        t[b, e] = end_id
    return torch.from_numpy(t)


@pytest.mark.parametrize("size", ["tiny", "vitl"])
def test_text_tower_on_packed_rows_equals_the_rectangular_layout(size, tiny_model):
    """Round 6: captions end at different columns, and under the causal mask (model.py:543-549) sample b needs its columns
    [0, read-out column] only.  keds_text_run_packed gives every sample exactly those rows (sum of the lengths instead of
    B x the longest); the GEMMs and LayerNorm statistics are row-wise, the attention takes per-sample offsets.  Same arithmetic per
    row as the rectangular layout (rows only land in other tiles): equal bits on the tiny model, the batch-size sweep's class at
    ViT-L/14 width (other tile shapes / split-K); the fp32-grade flows within their own tolerance.  Every text entry point, ragged
    read-out columns from 6 to 73; the fp32-stream flow (set_numerics("safe")) packs as well."""
    import keds_amd.model as M
    if size == "tiny":
        _, sd, m = tiny_model
        d = 128
    else:
        sd = O.synth_clip_state_dict(**VITL, seed=7)
        m = keds_amd.build_model(sd, fp16=False).cuda()
        del sd
        d = 768
    L, star = 77, 7
    rs = np.random.RandomState(5)

    def both_layouts(fn):
        M.TEXT_PACKED = True
        a = fn().clone()
        assert torch.equal(a, fn()) and torch.isfinite(a).all()
        M.TEXT_PACKED = False
        return a, fn().clone()
    try:
        for precision in ("bf16", "fp32x3", "fp32"):
            m.set_precision(precision)
            for tag, B, eots in (("mixed", 128, [9, 40, 12, 30, 41, 8]), ("wide", 37, [6, 73, 20, 33]), ("two", 2, [10, 50])):
                if precision != "bf16" and (tag != "wide" or (size == "vitl" and precision == "fp32")):
                    continue                    # the fp32-grade flows: one ragged case (fp32x3 at both sizes, fp32 on the tiny model)
                text = _ragged_tokens(B, L, eots, m.end_id, star, m.vocab_size)
                tok3 = torch.from_numpy(rs.standard_normal((B, 3, d)).astype(np.float32) * 0.05).cuda()
                calls = {
                    "encode_text": lambda: m.encode_text(text.cuda()),
                    "eti3": lambda: m.encode_text_img_retrieval(text.cuda(), tok3, split_ind=star, repeat=False),
                    "eti2": lambda: m.encode_text_img_retrieval(text, tok3[:, :2].contiguous(), split_ind=star, repeat=False),
                    "eti_train3": lambda: m.encode_text_img_train(text.cuda(), tok3, split_ind=star),
                }
                for name, fn in calls.items():
                    a, b = both_layouts(fn)
                    c, r = min_cosine(a, b), rel_l2(a, b)
                    report("text_packed_vs_rectangular", size=size, precision=precision, case=tag, call=name, B=B,
                           bit_equal=bool(torch.equal(a, b)), min_cosine=c, rel_l2=r)
                    assert m.precision == precision         # (no fp32x3 range trip on the zero rows behind the last sample)
                    if precision != "bf16":
                        assert r <= (2e-6 if precision == "fp32" else 1e-5), (size, precision, tag, name, r)
                    elif size == "tiny":
                        assert torch.equal(a, b), (size, tag, name, c, r)
                    else:
                        assert c >= 0.99998 and r <= 6e-3, (size, tag, name, c, r)
        m.set_precision("bf16")
        if size == "tiny":
            m.set_numerics("safe")
            text = _ragged_tokens(40, L, [9, 40, 12, 30, 41, 8], m.end_id, star, m.vocab_size)
            tok3 = torch.from_numpy(rs.standard_normal((40, 3, d)).astype(np.float32) * 0.05).cuda()
            a, b = both_layouts(lambda: m.encode_text_img_retrieval(text.cuda(), tok3, split_ind=star, repeat=False))
            assert torch.equal(a, b)
    finally:
        M.TEXT_PACKED = True
        m.set_numerics("auto")
        m.set_precision("bf16")


def test_text_readout_row_outside_the_declared_cut_comes_out_as_nan(tiny_model):
    """keds_text_run_ex trusts the host's `seq_used` (max read-out column + 1).  A device read-out row at or beyond it would read
    another sample's token out of the cut [B, seq_used, w] layout: the library returns NaN for such a row (every flow: the column
    cut's gather and the plain read-out), never a neighbour's embedding; rows inside the cut are untouched."""
    import ctypes as C
    from keds_amd import _lib
    lib = _lib.load()
    _, sd, m = tiny_model
    m.set_precision("bf16")
    B, L = 6, 77
    text = _ragged_tokens(B, L, [9, 12, 30], m.end_id, 7, m.vocab_size)
    good = m.encode_text(text.cuda()).float()
    eng = m._engine()
    tok = text.to("cuda", dtype=torch.int32).contiguous()
    ro = m._eot_columns(text).to(torch.int32).cuda().contiguous()
    seq_used = int(ro.max()) + 1
    bad = ro.clone()
    bad[2] = seq_used + 3                                        # one row lies about its read-out column
    nbytes = lib.keds_text_workspace_bytes(C.byref(eng.text), B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    try:
        for mode in (1, 2, 0):                                   # cut + tail (the default), the column cut only, no trimming
            lib.keds_text_trim_enable(mode)
            out = torch.empty((B, m.embed_dim), dtype=torch.float32, device="cuda")
            _lib.check(lib.keds_text_run_ex(C.byref(eng.text), _lib.ptr(tok), _lib.ptr(bad), None, 0, 0, B, seq_used if mode else 0,
                                            _lib.ptr(out), 0, _lib.ptr(ws), ws.numel(), _lib.stream()), "keds_text_run_ex")
            torch.cuda.synchronize()
            if mode:
                assert bool(torch.isnan(out[2]).all()), mode
            keep = [0, 1, 3, 4, 5]
            assert bool(torch.isfinite(out[keep]).all()), mode
            assert float((out[keep] - good[keep]).abs().max()) <= 2e-2 * float(good.abs().max()), mode
    finally:
        lib.keds_text_trim_enable(-1)


@pytest.mark.parametrize("size", ["tiny", "vitl"])
def test_text_tower_does_only_the_work_that_reaches_the_readout(size, tiny_model):
    """Round 5: keds_text_run_ex cuts the sequence behind the last read-out column (causal mask, model.py:543-549) and runs
    the last block's out-proj / ln_2 / MLP on the B read-out rows only (model.py:587-589, 847-849).  Against the all-columns /
    all-rows flow (keds_text_trim_enable(0) = the round-4 flow):
      * the COLUMN CUT alone (mode 2) changes no arithmetic -- the same MFMA chains on the same rows, which only land in other
        tiles / other kernels (256 x 256 vs 128 x 128 vs split-K): equal bits are reported, closeness far inside the operand
        rounding is asserted;
      * the read-out-row TAIL (mode 1 = cut + tail) runs the last block's second half in the fp32-stream form (stand-alone
        LayerNorm, fp32 residual: what the ViT's CLS tail does) where the all-rows flow uses the folded form on the fp16
        stream: two roundings of the same fp32 function, asserted inside the operand-rounding class;
      * in fp32 mode (no operand rounding) all flows agree to 2e-6 (measured: equal bits); in fp8 mode the cut changes which rows
        run on MXFP8 operands (full 256-row tiles) and which on bf16 (remainder rows): the fp8 tower's own error class.
    Ragged EOT sweep: EOT at column 6 (with the splice: read-out at 8), 40, 73 (the last column a 3-token splice allows), mixed."""
    from keds_amd import _lib
    lib = _lib.load()
    if size == "tiny":
        _, sd, m = tiny_model
        d = 128
    else:
        sd = O.synth_clip_state_dict(**VITL, seed=7)
        m = keds_amd.build_model(sd, fp16=False).cuda()
        del sd
        d = 768
    L, star = 77, 7
    rs = np.random.RandomState(3)
    try:
        for precision in ("bf16", "fp32", "fp8") if size == "vitl" else ("bf16",):
            m.set_precision(precision)
            for tag, B, eots in (("eot6", 5, [6]), ("eot40", 128, [40]), ("eot73", 33, [73]), ("mixed", 128, [9, 40, 12, 30, 41, 8])):
                if size == "vitl" and precision == "fp32" and B > 33:
                    B = 33
                text = _ragged_tokens(B, L, eots, m.end_id, star, m.vocab_size)
                tok3 = torch.from_numpy(rs.standard_normal((B, 3, d)).astype(np.float32) * 0.05).cuda()
                tok2 = tok3[:, :2].contiguous()
                calls = {
                    "encode_text": lambda: m.encode_text(text.cuda()),
                    "eti3": lambda: m.encode_text_img_retrieval(text.cuda(), tok3, split_ind=star, repeat=False),
                    "eti2": lambda: m.encode_text_img_retrieval(text, tok2, split_ind=star, repeat=False),
                    "eti3_repeat": lambda: m.encode_text_img_retrieval(text[:1], tok3, split_ind=star, repeat=True),
                    "eti_train3": lambda: m.encode_text_img_train(text.cuda(), tok3, split_ind=star),
                }
                for name, fn in calls.items():
                    out = {}
                    for mode in (1, 2, 0):
                        lib.keds_text_trim_enable(mode)
                        out[mode] = fn().clone()
                        assert torch.equal(out[mode], fn()) and torch.isfinite(out[mode]).all(), (size, precision, tag, name, mode)
                    for mode, what in ((2, "cut"), (1, "cut+tail")):
                        c, r = min_cosine(out[mode], out[0]), rel_l2(out[mode], out[0])
                        report("text_trim_vs_full", size=size, precision=precision, case=tag, call=name, B=B, flow=what,
                               bit_equal=bool(torch.equal(out[mode], out[0])), min_cosine=c, rel_l2=r)
                        if precision == "fp32":
                            assert r <= 2e-6, (size, tag, name, what, r)
                        elif precision == "fp8":      # which rows sit in full 256-row tiles (MXFP8) and which in the remainder (bf16)
                            assert c >= 0.99 and r <= 0.15, (size, precision, tag, name, what, c, r)    # changes with the cut: fp8's own class
                        elif mode == 2 and size == "tiny":
                            assert torch.equal(out[mode], out[0]), (size, precision, tag, name, what, c, r)     # measured: equal bits
                        else:               # (ViT-L/14 width: small-M launches split K, another summation order -- the batch-size sweep's class)
                            assert c >= 0.99998 and r <= 6e-3, (size, precision, tag, name, what, c, r)
    finally:
        lib.keds_text_trim_enable(-1)
        m.set_precision("bf16")
