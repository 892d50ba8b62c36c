"""CPU: pin the oracle (oracle/keds_oracle.py) to the golden vectors minted from the
reference's own Python (tools/mint_golden.py).  fp32 both sides; atol 1e-5 on
O(1)-scaled outputs (summation order differs between the reference's fused torch
ops and the oracle's explicit ones)."""
import os

import numpy as np
import pytest
import torch

from oracle import keds_oracle as O
from tests.conftest import golden_path

TINY = dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
            context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2)
VITL = dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
            context_length=77, vocab_size=49408, transformer_width=768, transformer_layers=12)


def _checksum(*sds):
    return float(sum(v.double().sum().item() for sd in sds for v in sd.values()))


def _close(a, b, atol=1e-5, rtol=1e-5):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def tiny():
    g = dict(np.load(golden_path("clip_tiny.npz")))
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    assert abs(_checksum(sd) - float(g["weights_checksum"])) < 1e-6, "seeded weight generator drifted"
    return g, sd


def test_arch_inference(tiny):
    _, sd = tiny
    a = O.arch_from_state_dict(sd)
    assert a["vision_layers"] == 2 and a["vision_width"] == 128 and a["image_resolution"] == 56
    assert a["transformer_layers"] == 2 and a["context_length"] == 77 and a["vocab_size"] == 512


def test_encode_image_tiny(tiny):
    g, sd = tiny
    mids = []
    out = O.encode_image(sd, torch.from_numpy(g["image"]), collect=mids)
    _close(out, g["encode_image"])
    for i, m in enumerate(mids):
        _close(m, g["block_tokens"][i], atol=2e-5)


def test_encode_text_tiny(tiny):
    g, sd = tiny
    _close(O.encode_text(sd, torch.from_numpy(g["text"])), g["encode_text"])


def test_encode_text_img_retrieval_tiny(tiny):
    g, sd = tiny
    text = torch.from_numpy(g["text"])
    star = int(g["star"])
    _close(O.encode_text_img_retrieval(sd, text, torch.from_numpy(g["tok3"]), split_ind=star, repeat=False), g["eti3"])
    _close(O.encode_text_img_retrieval(sd, text, torch.from_numpy(g["tok2"]), split_ind=star, repeat=False), g["eti2"])
    _close(O.encode_text_img_retrieval(sd, text[:1], torch.from_numpy(g["tok3"]), split_ind=star, repeat=True),
           g["eti3_repeat"])


def test_forward_normalised_tiny(tiny):
    g, sd = tiny
    i = O.l2_normalize(O.encode_image(sd, torch.from_numpy(g["image"])))
    t = O.l2_normalize(O.encode_text(sd, torch.from_numpy(g["text"])))
    _close(i, g["forward_image"])
    _close(t, g["forward_text"])
    assert abs(float(sd["logit_scale"].exp()) - float(g["forward_scale"])) < 1e-4


def test_text_edge_cases(tiny):
    g, sd = tiny
    text = torch.from_numpy(g["text"]).clone()
    bad = text.clone()
    bad[0, 70] = TINY["vocab_size"] - 1                       # second EOT in a row
    with pytest.raises(ValueError):
        O.encode_text(sd, bad)
    with pytest.raises(ValueError):                            # 1 or 4 tokens: wrong length in the reference
        O.encode_text_img_retrieval(sd, text, torch.zeros(4, 4, 128), split_ind=265, repeat=False)
    nostar = text.clone()
    nostar[0][nostar[0] == 265] = 24
    with pytest.raises(IndexError):
        O.encode_text_img_retrieval(sd, nostar, torch.zeros(4, 3, 128), split_ind=265, repeat=False)
    late = torch.zeros(1, 77, dtype=torch.int64)
    late[0, :6] = torch.tensor([510, 20, 21, 22, 265, 23])
    late[0, 6:76] = 40
    late[0, 76] = 511                                          # EOT in the last column: read-out overflows
    with pytest.raises(IndexError):
        O.encode_text_img_retrieval(sd, late, torch.zeros(1, 3, 128), split_ind=265, repeat=False)


@pytest.mark.parametrize("dim,middle", [(128, 128), (768, 512)])
def test_knowledge_modules(dim, middle):
    g = dict(np.load(golden_path(f"knowledge_d{dim}.npz")))
    sd_i = O.synth_im2text_state_dict(dim, middle, dim, 2, seed=11, tag="i2t")
    sd_f = O.synth_crossformer_state_dict(dim, 3, seed=12, tag="fuse")
    assert abs(_checksum(sd_i, sd_f) - float(g["weights_checksum"])) < 1e-6
    y = O.im2text(sd_i, torch.from_numpy(g["x"]))
    ynb = O.im2text(sd_i, torch.from_numpy(g["nb"]))
    _close(y, g["im2text_x"])
    _close(ynb, g["im2text_nb"])
    z = O.crossformer(sd_f, y[:, None, :], ynb, ynb)
    _close(z, g["crossformer"], atol=2e-5)


def test_cirr_batch_composition(tiny):
    """evaluate_cirr per-batch body incl. get_retrieved_features (eval_utils.py:652-714)."""
    gt, sd = tiny
    g = dict(np.load(golden_path("cirr_batch_tiny.npz")))
    n_db = int(g["n_db"])
    image_base = O.synth_database(n_db, 128, seed=2002)
    text_base = O.synth_database(n_db, 128, seed=2003)

    def stream(seed):
        return (O.synth_im2text_state_dict(128, 128, 128, 2, seed=seed, tag="i2t"),
                O.synth_crossformer_state_dict(128, 3, seed=seed, tag="fuse"),
                O.synth_crossformer_state_dict(128, 3, seed=seed, tag="cond"))

    out = O.compose_query(sd, stream(21), stream(22), torch.from_numpy(gt["image"]),
                          torch.from_numpy(gt["text"]), image_base, text_base, split_ind=265)
    _close(out["composed"], g["composed"], atol=2e-5)
    _close(out["image"], g["image"], atol=2e-5)
    _close(out["mixture"], g["mixture"], atol=2e-5)
    _close(out["tokens_image_stream"], g["tokens_image_stream"], atol=2e-5)
    _close(out["tokens_text_stream"], g["tokens_text_stream"], atol=2e-5)
    # retrieved neighbours: the reference shuffles the image neighbours along K, so compare sorted
    ti = image_base[out["topk_image_indices"].reshape(-1)].reshape(4, 16, 128).numpy()
    np.testing.assert_array_equal(np.sort(ti, axis=1), g["topk_image_sorted"])
    tt = text_base[out["topk_text_indices"].reshape(-1)].reshape(4, 16, 128).numpy()
    np.testing.assert_array_equal(tt, g["topk_text"])


def test_search_against_reference_bruteforce():
    """flat_l2_search vs the reference's own torch brute force (trainer.py:246-257)."""
    g = dict(np.load(golden_path("search_small.npz")))
    image_base = O.synth_database(3000, 64, seed=31)
    text_base = O.synth_database(3000, 64, seed=32, clustered=True, n_centroids=64)
    q = torch.from_numpy(g["q"])
    D, I = O.flat_l2_search(image_base, q, 16, chunk=1000)
    np.testing.assert_array_equal(I.numpy(), g["I_image"])
    np.testing.assert_allclose(D.numpy(), g["D_image"], atol=1e-6)
    np.testing.assert_array_equal(image_base[I.reshape(-1)].reshape(9, 16, 64).numpy(), g["topk_image"])
    D, I = O.flat_l2_search(text_base, q, 16)
    np.testing.assert_array_equal(I.numpy(), g["I_text"])
    np.testing.assert_array_equal(text_base[I.reshape(-1)].reshape(9, 16, 64).numpy(), g["topk_text"])
    # unit-norm DB: inner-product ranking == L2 ranking
    _, Iip = O.flat_ip_search(image_base, q, 16)
    np.testing.assert_array_equal(Iip.numpy(), g["I_image"])


def test_search_edge_cases():
    db = O.synth_database(50, 16, seed=5)
    q = O.synth_database(3, 16, seed=6)
    D, I = O.flat_l2_search(db, q, 50)                       # k == N
    assert I.shape == (3, 50) and (np.sort(I.numpy(), axis=1) == np.arange(50)).all()
    assert (np.diff(D.numpy(), axis=1) >= 0).all()
    dup = torch.cat([db, db[:5]])                             # exact duplicates: lower index first
    D, I = O.flat_l2_search(dup, db[:5], 2)
    np.testing.assert_array_equal(I.numpy()[:, 0], np.arange(5))
    np.testing.assert_array_equal(I.numpy()[:, 1], np.arange(50, 55))
    assert np.allclose(D.numpy(), 0, atol=1e-6)


def test_metrics_cirr():
    g = dict(np.load(golden_path("metrics_cirr.npz")))
    index_names = [f"/data/cirr/dev/img_{i:05d}.png" for i in range(g["gallery"].shape[0])]
    ref_names = [os.path.basename(index_names[i]) for i in g["ref_idx"]]
    tgt_names = [os.path.basename(index_names[i]) for i in g["tgt_idx"]]
    m = O.get_metrics_cirr(torch.from_numpy(g["gallery"]), torch.from_numpy(g["ref"]), ref_names, index_names, tgt_names)
    for k in (1, 5, 10, 50, 100):
        assert abs(m[f"recall_R@{k}"] - float(g[f"recall_R_at_{k}"])) < 1e-4
    with pytest.raises(AssertionError):                        # target absent from the gallery (eval_utils.py:1063)
        O.get_metrics_cirr(torch.from_numpy(g["gallery"]), torch.from_numpy(g["ref"]), ref_names, index_names,
                           ["nope.png"] * len(tgt_names))


def test_eval_glue_other_drivers():
    """encode_text_img_train + the fashion / coco / imgnet metrics + the CIRR test-server output,
    against vectors minted from the reference functions themselves (tools/mint_golden.py::mint_eval_glue)."""
    g = dict(np.load(golden_path("eval_glue.npz")))
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    text, tok3, star = torch.from_numpy(g["text"]), torch.from_numpy(g["tok3"]), int(g["star"])
    _close(O.encode_text_img_train(sd, text, tok3, split_ind=star, repeat=False), g["eti_train3"], atol=2e-5, rtol=1e-4)
    with pytest.raises(RuntimeError):                              # 2 tokens: 76 != 77 rows (model.py:880-883)
        O.encode_text_img_train(sd, text, tok3[:, :2], split_ind=star)
    gallery, ref = torch.from_numpy(g["gallery"]), torch.from_numpy(g["ref"])
    names = [f"dress/img_{i:05d}.jpg" for i in range(gallery.shape[0])]
    m = O.get_metrics_fashion(gallery, ref, names, [names[i] for i in g["answer_idx"]])
    for k in (1, 5, 10, 50, 100):
        assert abs(m[f"R@{k}"] - float(g[f"fashion_R_at_{k}"])) < 1e-4
    m = O.get_metrics_coco(torch.from_numpy(g["coco_image"]), torch.from_numpy(g["coco_ref"]), torch.tensor(100.0))
    assert len(m) == 14
    for key, v in m.items():
        assert abs(v - float(g["coco_" + key.replace("@", "_at_")])) < 1e-6, key
    m = O.get_metrics_imgnet(torch.from_numpy(g["imgnet_q"]), torch.from_numpy(g["imgnet_t"]),
                             torch.from_numpy(g["imgnet_ql"]), torch.from_numpy(g["imgnet_tl"]))
    assert len(m) == 12
    for key, v in m.items():
        assert abs(v - float(g["imgnet_" + key.replace("@", "_at_")])) < 2e-6, key
    tnames = [f"test1-{i}-img0.png" for i in range(gallery.shape[0])]
    res = O.get_cirr_testoutput(gallery, ref, [tnames[i] for i in g["cirr_test_ref_idx"]], tnames,
                                torch.arange(1000, 1000 + ref.shape[0]))
    assert res["version"] == "rc2" and res["metric"] == "recall"
    for i in range(ref.shape[0]):
        assert [int(n.split("-")[1]) for n in res[str(1000 + i)]] == g["cirr_test_top50"][i].tolist()
        assert all(not n.endswith(".png") for n in res[str(1000 + i)])


@pytest.mark.skipif(not os.path.exists(golden_path("clip_vitl14.npz")), reason="ViT-L/14 golden not minted")
def test_vitl14_full_size():
    """Full ViT-L/14 + 12-layer text tower at B=2 (about 20 s of CPU)."""
    g = dict(np.load(golden_path("clip_vitl14.npz")))
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    assert abs(_checksum(sd) - float(g["weights_checksum"])) < 1e-3
    mids = []
    out = O.encode_image(sd, torch.from_numpy(g["image"]), collect=mids)
    _close(out, g["encode_image"], atol=5e-5, rtol=1e-4)
    for i in (0, 11, 23):
        _close(mids[i][:, 0, :], g["block_cls"][i], atol=2e-4, rtol=1e-4)
    text = torch.from_numpy(g["text"])
    _close(O.encode_text(sd, text), g["encode_text"], atol=5e-5, rtol=1e-4)
    _close(O.encode_text_img_retrieval(sd, text, torch.from_numpy(g["tok3"]), split_ind=265, repeat=False),
           g["eti3"], atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("tag,cfg,dim,middle", [
    ("tiny", dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
                  context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2), 128, 128),
    ("vitl_text", dict(embed_dim=768, image_resolution=56, vision_layers=1, vision_width=128, vision_patch_size=14,
                       context_length=77, vocab_size=49408, transformer_width=768, transformer_layers=12), 768, 512)])
def test_training_loss_and_gradients_match_the_reference_loss_function(tag, cfg, dim, middle):
    """`oracle.training_loss` (restatement of src/trainer.py:44-127) pinned to the reference's OWN
    `get_loss_img2text_image` + `get_retrieved_features` run under a 1-rank gloo group with distributed = aggregate = True
    and dropout 0 (tools/mint_golden.py train): the loss and torch-autograd gradients of all 54 module parameters.  The
    oracle's gradients are in turn the reference of the HIP backward kernels (tests/test_gpu_train.py)."""
    g = np.load(golden_path(f"train_step_{tag}.npz"))
    sd_clip = O.synth_clip_state_dict(**cfg, seed=7)
    sds = (O.synth_im2text_state_dict(dim, middle, dim, 2, seed=31, tag="i2t"),
           O.synth_crossformer_state_dict(dim, 3, seed=31, tag="fuse"),
           O.synth_crossformer_state_dict(dim, 3, seed=31, tag="cond"))
    chk = sum(float(sum(v.double().sum().item() for v in sd.values())) for sd in (sd_clip,) + sds)
    assert abs(chk - float(g["weights_checksum"])) <= 1e-6 * max(1.0, abs(chk))       # generator drift would show here
    n_db = int(g["n_db"])
    image_base, text_base = O.synth_database(n_db, dim, seed=2002), O.synth_database(n_db, dim, seed=2003)
    feat = torch.from_numpy(g["image_features"])
    ti, tt, _, _ = O.get_retrieved_features(feat, image_base, text_base, 16)
    leaves = [{k: v.clone().requires_grad_(True) for k, v in sd.items()} for sd in sds]
    with torch.enable_grad():
        loss = O.training_loss(sd_clip, leaves[0], leaves[1], leaves[2], feat, ti, tt, torch.from_numpy(g["text"]),
                               int(g["star"]), masks=None, p_drop=0.0)
        names, tensors = [], []
        for pfx, leaf in zip(("i2t.", "fuse.", "cond."), leaves):
            for k, v in leaf.items():
                names.append(pfx + k)
                tensors.append(v)
        grads = dict(zip(names, torch.autograd.grad(loss, tensors)))
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    assert sorted(names) == sorted(str(n) for n in g["names"])
    for k in names:
        mine = grads[k].detach()
        if "g." + k in g.files:
            want = torch.from_numpy(g["g." + k])
            tol = 1e-4 * max(float(want.abs().max()), 1e-3)      # fp32 summation order through up to 12 blocks of backward
            assert float((mine - want).abs().max()) <= tol, k
        else:
            s_, a_ = g["gs." + k]
            assert abs(mine.double().sum().item() - s_) <= 1e-4 * max(a_, 1e-6), k
            assert abs(mine.double().abs().sum().item() - a_) <= 1e-4 * max(a_, 1e-6), k
            head = torch.from_numpy(g["gh." + k])
            assert float((mine.reshape(-1)[:256] - head).abs().max()) <= 1e-4 * max(float(head.abs().max()), 1e-3), k
