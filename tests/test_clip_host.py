"""CPU: the host-side loaders in front of the encoders (keds_amd/clip.py): tokenizer against ids minted from the
reference's own tokenizer, image preprocessing semantics, local checkpoint loading."""
import json
import os

import numpy as np
import pytest
import torch

from keds_amd import clip as kclip
from oracle import keds_oracle as O
from tests.conftest import golden_path

# the merge table is reference data that is not copied into this repository; the test runs where a checkout has it
BPE = os.environ.get("KEDS_BPE_VOCAB", "/root/reference/src/third_party/open_clip/bpe_simple_vocab_16e6.txt.gz")


@pytest.mark.skipif(not os.path.isfile(BPE), reason="BPE merge table (bpe_simple_vocab_16e6.txt.gz) not available")
def test_tokenizer_matches_reference_ids(monkeypatch):
    monkeypatch.setenv("KEDS_TOKENIZER", "python")                         # the pure-Python SimpleTokenizer (the checker)
    g = json.load(open(golden_path("tokenizer.json")))
    got = kclip.tokenize(g["texts"], bpe_path=BPE)
    assert got.dtype == torch.int32 and tuple(got.shape) == (len(g["texts"]), 77)
    assert got.tolist() == g["tokens"]
    assert int(kclip.tokenize(["*"], bpe_path=BPE)[0][1]) == 265            # id_split of every eval driver (eval_utils.py:656)
    tk = kclip.SimpleTokenizer(BPE)
    for row, want in zip(g["tokens"], g["decoded"]):
        assert tk.decode(row[1:row.index(49407)]) == want
    with pytest.raises(RuntimeError):
        kclip.tokenize(["word " * 100], truncate=False, bpe_path=BPE)
    assert int(kclip.tokenize(["word " * 100], bpe_path=BPE)[0, 76]) == 49407


def _nasty_texts(n, seed):
    import random
    rnd = random.Random(seed)
    pools = ["abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789",
             " \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f\x85\xa0\u1680\u2000\u2009\u2028\u2029\u202f\u205f\u3000",
             "'`\".,;:!?-_()[]{}<>|*&#@/\\^~+=%$", "ΣσςΆέήίόύώΑΒΓabcİıǅǆßẞÀÉÎÕÜçñſ", "ʰʱ\u0345\u00ad\u0301\u0307’·\u200d",
             "中文日本語한국어العربيةעבריתहिन्दीไทย", "😀🎉🚀👍🏽\U0001F1FA\U0001F1F8①②③ⅣⅤ½¾", "٠١०१１２①²³"]
    ents = ["&amp;", "&lt;", "&gt", "&quot;", "&#39;", "&#x27;", "&#X41", "&#65", "&#0;", "&#128;", "&#x80;", "&#xD800;", "&#x110000;",
            "&#99999999999999999999;", "&notit;", "&notin;", "&amp;amp;", "&ampx", "&nbsp;", "&copy", "&copyright", "&zzz;", "&#", "&#x",
            "&#xg", "&;", "& ", "&&amp;", "&AMP;", "&Aacute", "&acE;", "&NotEqualTilde;", "<|startoftext|>", "<|endoftext|>",
            "<|STARTOFTEXT|>", "<|ſtartoftext|>", "'s", "'ſ", "'T", "'re", "'VE", "'ll", "'d", "'m", "&ampΣ", "&amé;"]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(rnd.randint(0, 14)):
            k = rnd.random()
            if k < 0.2:
                parts.append(rnd.choice(ents))
            elif k < 0.25:
                parts.append(chr(rnd.choice([rnd.randint(1, 0xD7FF), rnd.randint(0xE000, 0x2FFFF)])))
            else:
                pool = rnd.choice(pools)
                parts.append("".join(rnd.choice(pool) for _ in range(rnd.randint(1, 6))))
        out.append("".join(parts))
    return out


def _synthetic_merges(path, n_merges, seed):
    """A merge table in the file format of bpe_simple_vocab_16e6.txt (header line, then `a b` per line) over the byte
    stand-ins: random, but a valid input for both tokenizers (ids are positions, so any table works)."""
    import random
    rnd = random.Random(seed)
    sym = kclip._byte_symbols()
    units = [sym[b] for b in range(256)]
    units += [u + "</w>" for u in units]
    lines, seen = ["#version: synthetic"], set()
    while len(lines) <= n_merges:
        a, b = rnd.choice(units), rnd.choice(units)
        if a.endswith("</w>") or (a, b) in seen:
            continue
        seen.add((a, b))
        lines.append(f"{a} {b}")
        units.append(a + b)
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


def test_native_tokenizer_equals_python_tokenizer_on_nasty_text(tmp_path):
    """keds_tokenize (C++: tokenizer.cpp + the generated Unicode / HTML tables) against SimpleTokenizer, id for id, on
    text built to hit every rule: HTML references (valid, invalid, prefixes, double escaping), every kind of white space,
    Final_Sigma, multi-code-point lower-casing, combining marks, the IGNORECASE corner cases (U+0345, U+017F), astral
    characters, the special-token literals.  Synthetic merge table: runs anywhere."""
    bpe = tmp_path / "merges.txt"
    _synthetic_merges(str(bpe), 3000, seed=1)
    py, nat = kclip.SimpleTokenizer(str(bpe)), kclip.NativeTokenizer(str(bpe))
    assert (nat.sot, nat.eot) == (py.encoder["<|startoftext|>"], py.encoder["<|endoftext|>"]) == (512 + 3001, 512 + 3002)
    texts = [t for t in _nasty_texts(4000, seed=5) if "\x00" not in t] + ["", " ", "a", "word " * 100]
    got = nat(texts)
    for t, row in zip(texts, got.tolist()):
        ids = [nat.sot] + py.encode(t) + [nat.eot]
        if len(ids) > 77:
            ids = ids[:77]
            ids[-1] = nat.eot
        assert row == ids + [0] * (77 - len(ids)), repr(t)
    with pytest.raises(RuntimeError, match="too long"):
        nat(["word " * 100], truncate=False)
    assert tuple(nat([]).shape) == (0, 77)


@pytest.mark.skipif(not os.path.isfile(BPE), reason="BPE merge table (bpe_simple_vocab_16e6.txt.gz) not available")
def test_native_tokenizer_matches_reference_ids(monkeypatch):
    g = json.load(open(golden_path("tokenizer.json")))
    monkeypatch.setenv("KEDS_TOKENIZER", "native")
    assert kclip.tokenize(g["texts"], bpe_path=BPE).tolist() == g["tokens"]        # the default path of tokenize()
    nat, py = kclip.NativeTokenizer(BPE), kclip.SimpleTokenizer(BPE)
    texts = [t for t in _nasty_texts(2000, seed=9) if "\x00" not in t]
    for t, row in zip(texts, nat(texts).tolist()):
        ids = ([nat.sot] + py.encode(t) + [nat.eot])
        if len(ids) > 77:
            ids = ids[:77]
            ids[-1] = nat.eot
        assert row == ids + [0] * (77 - len(ids)), repr(t)


def test_tokenizer_needs_the_merge_table(monkeypatch):
    monkeypatch.delenv("KEDS_BPE_VOCAB", raising=False)
    kclip._tokenizer = None
    with pytest.raises(FileNotFoundError):
        kclip.tokenize(["a photo"], bpe_path="/nonexistent/bpe.txt.gz")


def test_eval_transform_semantics():
    from PIL import Image
    tf = kclip._transform(224, is_train=False)
    # a constant image normalises to (c/255 - mean) / std everywhere, whatever its size
    img = Image.new("RGB", (640, 480), (128, 64, 255))
    t = tf(img)
    assert tuple(t.shape) == (3, 224, 224) and t.dtype == torch.float32
    want = (np.array([128, 64, 255], np.float32) / 255 - np.array(kclip.CLIP_MEAN, np.float32)) / np.array(kclip.CLIP_STD, np.float32)
    assert np.allclose(t.mean(dim=(1, 2)).numpy(), want, atol=1e-5) and float(t.std(dim=(1, 2)).max()) < 1e-5
    # shorter side -> 224 keeping the aspect ratio, then the CENTRAL 224 columns: a left/right split image stays split
    arr = np.zeros((300, 600, 3), np.uint8)
    arr[:, 300:] = 255
    t = tf(Image.fromarray(arr))
    raw = t * torch.tensor(kclip.CLIP_STD)[:, None, None] + torch.tensor(kclip.CLIP_MEAN)[:, None, None]
    assert float(raw[:, :, :100].max()) < 0.02 and float(raw[:, :, 124:].min()) > 0.98
    # already-sized input is only normalised (no resampling)
    rs = np.random.RandomState(0).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    t = tf(Image.fromarray(rs))
    assert np.allclose((t.permute(1, 2, 0).numpy() * np.array(kclip.CLIP_STD) + np.array(kclip.CLIP_MEAN)) * 255, rs, atol=1e-3)
    # grayscale input is converted to RGB
    assert tuple(tf(Image.new("L", (300, 250), 77)).shape) == (3, 224, 224)
    tr = kclip._transform(224, is_train=True)(Image.fromarray(arr))
    assert tuple(tr.shape) == (3, 224, 224)


def test_load_local_checkpoints(tmp_path):
    sd = O.synth_clip_state_dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
                                 context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2, seed=7)
    plain = tmp_path / "plain.pt"
    torch.save(sd, plain)
    wrapped = tmp_path / "ckpt.pt"
    torch.save({"epoch": 3, "state_dict": {"module." + k: v for k, v in sd.items()}}, wrapped)
    for path in (plain, wrapped):
        model, ptrain, pval = kclip.load(str(path), device="cpu", jit=False)
        assert model.visual.input_resolution == 56 and model.embed_dim == 128
        got = model.state_dict()
        # build_model casts the matmul weights to fp16 like the reference (model.py:927-948,988): equal up to that rounding
        assert all(torch.allclose(got[k].float(), v.float(), rtol=1e-3, atol=1e-6) for k, v in sd.items())
        assert callable(ptrain) and callable(pval)
    with pytest.raises(RuntimeError, match="not found"):
        kclip.load("ViT-L/14", device="cpu")
    assert kclip.available_models() == []


def test_pil_bicubic_coefficients_reproduce_pil_resize_bit_for_bit():
    """The host half of the bit-exact image preprocessing (reference: src/model/clip.py:107-123 runs
    PIL's ImagingResample): `ops.pil_bicubic_coeffs` must give PIL's own 22-bit integer weights.  Pinned by running the
    integer two-pass resampler on those tables in numpy and comparing with PIL's resize byte for byte (down- and
    up-scaling, odd sizes)."""
    from PIL import Image
    from keds_amd import ops

    def resample(img, out_size, axis):
        if axis == 0:
            img = img.transpose(1, 0, 2)
        b, kk = ops.pil_bicubic_coeffs(img.shape[1], out_size)
        im = img.astype(np.int64)
        out = np.zeros((img.shape[0], out_size, 3), np.uint8)
        for xx in range(out_size):
            x0, n = b[xx]
            acc = (1 << 21) + (im[:, x0:x0 + n, :] * kk[xx, :n][None, :, None]).sum(1)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        return out.transpose(1, 0, 2) if axis == 0 else out

    rs = np.random.RandomState(0)
    for H, W, oh, ow in ((480, 640, 224, 298), (640, 427, 335, 224), (231, 1000, 224, 969), (300, 300, 224, 224),
                         (100, 80, 280, 224), (97, 211, 224, 487)):
        arr = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))
        got = arr
        if ow != W:
            got = resample(got, ow, 1)                       # PIL: horizontal pass first, uint8 intermediate
        if oh != H:
            got = resample(got, oh, 0)
        assert np.array_equal(got, want), (H, W, oh, ow)
    assert ops.resize_crop_geometry(480, 640, 224) == (298, 224, 37, 0)
    assert ops.resize_crop_geometry(640, 427, 224) == (224, 335, 0, 56)
