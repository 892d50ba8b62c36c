"""Host-side loaders that sit in front of the encoders: `load`, `_transform`, `tokenize`.

Mirrors the call shapes of the reference (src/model/clip.py:107-123 `_transform`, :132-188 `load`;
src/third_party/open_clip/clip.py:191-227 `tokenize`, simple_tokenizer.py:62-132) so the eval drivers'
`load(args.model, jit=False)`, `preprocess_val(img)` and `tokenize(["*"])[0][1]` lines keep working.
Nothing here touches the GPU path: it is PIL / numpy / pure-Python preprocessing (SURVEY.md 8f rank 3 keeps the
GPU versions for later).

Differences that are deliberate:
* no downloads (no network): `load` takes a local file -- a state dict, a training checkpoint
  ({"state_dict": ...} with an optional "module." prefix) or a TorchScript archive;
* the BPE merge table is not shipped with this package: point `KEDS_BPE_VOCAB` (or the `bpe_path` argument) at the
  `bpe_simple_vocab_16e6.txt.gz` of an OpenAI-CLIP / open_clip checkout (the reference keeps it under
  src/third_party/open_clip/);
* `ftfy.fix_text` is applied only when ftfy is installed (it is a no-op on clean text).
"""
from __future__ import annotations

import gzip
import html
import os
import random
from functools import lru_cache
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .model import build_model

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # src/model/clip.py:108
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ---------------------------------------------------------------------------------------------------
# image preprocessing (torchvision semantics restated on PIL + numpy; torchvision is not a dependency)
# ---------------------------------------------------------------------------------------------------
def _to_normalized_tensor(img) -> torch.Tensor:
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0             # HWC in [0,1]   (ToTensor)
    a = (a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def _resize_shorter_side(img, n_px: int):
    """torchvision Resize(int): the shorter side becomes n_px, the other int(n_px * long / short), PIL bicubic."""
    from PIL import Image
    w, h = img.size
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        return img
    if w < h:
        return img.resize((n_px, int(n_px * h / w)), Image.BICUBIC)
    return img.resize((int(n_px * w / h), n_px), Image.BICUBIC)


def _center_crop(img, n_px: int):
    w, h = img.size
    left, top = int(round((w - n_px) / 2.0)), int(round((h - n_px) / 2.0))
    return img.crop((left, top, left + n_px, top + n_px))


def _transform(n_px: int, is_train: bool = False) -> Callable:
    """src/model/clip.py:107-123.  Eval: Resize(n_px, bicubic) -> CenterCrop -> RGB -> [0,1] -> normalise.
    Train: RandomResizedCrop(n_px, scale=(0.9, 1.0), ratio=(3/4, 4/3), bicubic) in place of resize + crop."""
    from PIL import Image

    def eval_tf(img):
        return _to_normalized_tensor(_center_crop(_resize_shorter_side(img, n_px), n_px))

    def train_tf(img):
        w, h = img.size
        area = w * h
        for _ in range(10):
            target = area * random.uniform(0.9, 1.0)
            log_r = random.uniform(np.log(3.0 / 4.0), np.log(4.0 / 3.0))
            ar = float(np.exp(log_r))
            cw, ch = int(round(np.sqrt(target * ar))), int(round(np.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                top, left = random.randint(0, h - ch), random.randint(0, w - cw)
                break
        else:                                                                  # fallback: central crop at a valid ratio
            ratio = w / h
            if ratio < 3.0 / 4.0:
                cw, ch = w, int(round(w / (3.0 / 4.0)))
            elif ratio > 4.0 / 3.0:
                ch, cw = h, int(round(h * (4.0 / 3.0)))
            else:
                cw, ch = w, h
            top, left = (h - ch) // 2, (w - cw) // 2
        return _to_normalized_tensor(img.crop((left, top, left + cw, top + ch)).resize((n_px, n_px), Image.BICUBIC))

    return train_tf if is_train else eval_tf


# ---------------------------------------------------------------------------------------------------
# model loading
# ---------------------------------------------------------------------------------------------------
def available_models() -> List[str]:
    """The reference lists downloadable names (src/model/clip.py:125-127); this package never downloads."""
    return []


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, is_train: bool = False,
         pretrained: bool = True):
    """`model, preprocess_train, preprocess_val = load(path, device, jit=False)` (src/model/clip.py:132-188).

    `name` must be a local file: a state dict, a checkpoint dict with a "state_dict" entry (keys may carry the
    "module." prefix of DistributedDataParallel, :176-178) or a TorchScript archive whose `state_dict()` is used.
    `jit=True` has no meaning here (the forward is the HIP path) and is ignored like `pretrained`."""
    if not os.path.isfile(name):
        raise RuntimeError(f"Model {name} not found; this build loads local checkpoint files only "
                           f"(available models = {available_models()})")
    try:
        state_dict = torch.jit.load(name, map_location="cpu").eval().state_dict()
    except RuntimeError:
        state_dict = torch.load(name, map_location="cpu")
    if isinstance(state_dict, dict) and "state_dict" in state_dict and "visual.conv1.weight" not in state_dict:
        state_dict = state_dict["state_dict"]
    if next(iter(state_dict)).startswith("module."):
        state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
    model = build_model(state_dict).to(device)
    if str(device) == "cpu":
        model.float()                                   # src/model/clip.py:183-184
    res = model.visual.input_resolution
    return model, _transform(res, is_train=True), _transform(res, is_train=False)


# ---------------------------------------------------------------------------------------------------
# byte-level BPE tokenizer (the published CLIP scheme, written from its description)
# ---------------------------------------------------------------------------------------------------
@lru_cache()
def _byte_symbols() -> dict:
    """Every byte gets a printable unicode stand-in: printable latin-1 bytes map to themselves, the remaining 68 bytes
    to code points 256, 257, ... in increasing byte order."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    """Lower-cased, whitespace-normalised text -> regex word pieces -> byte symbols -> greedy lowest-rank merges.

    Vocabulary order: the 256 byte symbols (printable bytes first), the same with the end-of-word mark, one entry per
    merge, then <|startoftext|> = 49406 and <|endoftext|> = 49407."""

    END = "</w>"

    def __init__(self, bpe_path: Optional[str] = None):
        bpe_path = bpe_path or os.environ.get("KEDS_BPE_VOCAB", "")
        if not bpe_path or not os.path.isfile(bpe_path):
            raise FileNotFoundError("BPE merge table not found: pass bpe_path or set KEDS_BPE_VOCAB to "
                                    "bpe_simple_vocab_16e6.txt.gz (src/third_party/open_clip/ in the reference)")
        import regex
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        n_merges = 49152 - 256 - 2
        merges = [tuple(line.split()) for line in lines[1:1 + n_merges]]
        sym = _byte_symbols()
        base = [sym[b] for b in sorted(sym, key=lambda b: (sym[b] != chr(b), b if sym[b] == chr(b) else ord(sym[b])))]
        vocab = base + [s + self.END for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sym = sym
        self.unsym = {v: k for k, v in sym.items()}
        self._memo = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}
        self.pat = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                 regex.IGNORECASE)
        self._ws = regex.compile(r"\s+")

    def _merge(self, piece: str) -> List[str]:
        if piece in self._memo:
            return self._memo[piece]
        parts = list(piece[:-1]) + [piece[-1] + self.END]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best_rank is None:
                break
            a, b = parts[best], parts[best + 1]
            out, i = [], 0
            while i < len(parts):                       # merge EVERY occurrence of the winning pair, left to right
                if i < len(parts) - 1 and parts[i] == a and parts[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            parts = out
        self._memo[piece] = parts
        return parts

    @staticmethod
    def _clean(text: str) -> str:
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        return html.unescape(html.unescape(text)).strip()

    def encode(self, text: str) -> List[int]:
        text = self._ws.sub(" ", self._clean(text)).strip().lower()
        ids: List[int] = []
        for word in self.pat.findall(text):
            piece = "".join(self.sym[b] for b in word.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._merge(piece))
        return ids

    def decode(self, tokens: Sequence[int]) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.unsym[c] for c in text).decode("utf-8", errors="replace").replace(self.END, " ")


class NativeTokenizer:
    """The C++ tokenizer of libkeds_hip.so (include/keds_session.h, keds_tokenize): same ids as SimpleTokenizer.encode,
    whole batches per call."""

    def __init__(self, bpe_path: Optional[str] = None):
        import ctypes as C
        from . import _lib
        bpe_path = bpe_path or os.environ.get("KEDS_BPE_VOCAB", "")
        if not bpe_path or not os.path.isfile(bpe_path):
            raise FileNotFoundError("BPE merge table not found: pass bpe_path or set KEDS_BPE_VOCAB to "
                                    "bpe_simple_vocab_16e6.txt.gz (src/third_party/open_clip/ in the reference)")
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.keds_tokenizer_create(bpe_path.encode(), C.byref(h)), "keds_tokenizer_create")
        self._h = h
        sot, eot = C.c_int32(), C.c_int32()
        _lib.check(self._lib.keds_tokenizer_special(self._h, C.byref(sot), C.byref(eot)), "keds_tokenizer_special")
        self.sot, self.eot = sot.value, eot.value

    def __call__(self, texts: List[str], context_length: int = 77, truncate: bool = True) -> torch.Tensor:
        import ctypes as C
        from . import _lib
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        if not texts:
            return out
        enc = [t.encode("utf-8") for t in texts]
        if any(b"\x00" in e for e in enc):
            raise ValueError("text contains a NUL character")
        arr = (C.c_char_p * len(enc))(*enc)
        rc = self._lib.keds_tokenize(self._h, arr, len(enc), context_length, 1 if truncate else 0, out.data_ptr())
        _lib.check(rc, "keds_tokenize")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.keds_tokenizer_destroy(self._h)
                self._h = None
        except Exception:
            pass


_tokenizer: Optional[SimpleTokenizer] = None
_native: Optional[NativeTokenizer] = None


def _get_tokenizer(bpe_path: Optional[str] = None) -> SimpleTokenizer:
    global _tokenizer
    if _tokenizer is None or bpe_path:
        _tokenizer = SimpleTokenizer(bpe_path)
    return _tokenizer


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = True,
             bpe_path: Optional[str] = None) -> torch.Tensor:
    """[n, context_length] int32: <|startoftext|> ids <|endoftext|> 0 0 ...  (open_clip/clip.py:191-227).
    Over-long rows are cut to the context length with the last id forced to <|endoftext|>, or raise with
    truncate=False.  Runs in the C++ tokenizer of libkeds_hip.so (keds_tokenize); KEDS_TOKENIZER=python selects the
    pure-Python SimpleTokenizer (the checker of the tests; also provides decode())."""
    if isinstance(texts, str):
        texts = [texts]
    if os.environ.get("KEDS_TOKENIZER", "native") != "python":      # the C++ tokenizer is the product path
        global _native
        if _native is None or bpe_path:
            _native = NativeTokenizer(bpe_path)
        return _native(list(texts), context_length, truncate)
    tk = _get_tokenizer(bpe_path)
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    out = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, t in enumerate(texts):
        ids = [sot] + tk.encode(t) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int)
    return out
