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
def test_tokenizer_matches_reference_ids():
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
