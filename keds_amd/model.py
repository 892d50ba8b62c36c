"""Drop-in model API of the KEDs retrieval path on MI355X.

Same class names, constructor arguments, method signatures and ``state_dict`` keys as the
reference's ``src/model/model.py`` (CLIP :431-911, IM2TEXT :105-123, CrossFormer :81-101,
build_model :951, convert_weights :927), so reference checkpoints load unchanged
(``main.py:330-341`` layout) and reference call sites (``eval_utils.py:610,652-695``,
``eval_retrieval.py:96-101``) run as written.  The modules here only HOLD parameters
(torch.nn containers); every forward runs hand-written gfx950 kernels through
``libkeds_hip.so`` -- bf16 / fp16 MFMA GEMMs with fp32 accumulation, the residual stream of a tower kept in
fp16 with LayerNorm folded into the GEMMs (fp32 stream + stand-alone LayerNorm in numerics mode "safe",
selected automatically by the numerics guard), fp32 LayerNorm / softmax statistics.  There is no PyTorch compute path and no CPU fallback:
calling a forward without the library or without a GPU raises RuntimeError.

Out of scope: ModifiedResNet towers, `mid_feature`,
`encode_text_img_vis`, the training-only splice variants and `forward(extra=True)` (undefined
in the reference itself, App. B).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check, load, ptr, stream


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.bfloat16).contiguous()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------------
# parameter containers (key names are the checkpoint contract)
# ---------------------------------------------------------------------------------------------------
class LayerNorm(nn.LayerNorm):
    """Parameter holder; the statistics run in fp32 inside the HIP kernels (model.py:291-297)."""


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x), fused into the c_fc GEMM epilogue (model.py:300-302)."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask: Optional[torch.Tensor] = None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)     # holds in_proj_weight/bias, out_proj.*
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.causal = attn_mask is not None


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask: Optional[torch.Tensor] = None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])


class VisualTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


def _pack_tower(tr: Transformer, seq: int, causal: bool, keep: list, cls_only: bool = False,
                fp8: bool = False, folded: bool = True, f32: bool = False) -> _lib.TowerParams:
    blocks = (_lib.BlockParams * tr.layers)()
    # f32 = 1: the fp32-accurate flow (csrc/f32path.hip) multiplies the weights as stored; f32 = 2 ("fp32x3"): the four block
    # weights as PAIRS of fp16 planes [2, N, K] of w 2^e (hi + lo to 22 bits; keds_split_f16_weight picks the exact per-matrix
    # power of two that keeps the low plane of small weights out of the fp16 subnormals) for the split-operand GEMMs
    x3_exps = []

    def _planes(w):
        w32 = _f32(w)
        n, k = w32.shape
        out = torch.empty((2, n, k), dtype=torch.float16, device=w32.device)
        e = C.c_int32(0)
        check(load().keds_split_f16_weight(ptr(w32), n, k, ptr(out), n * k, C.byref(e), stream()), "keds_split_f16_weight")
        x3_exps.append(int(e.value))
        return out
    wcast = _planes if f32 == 2 else _f32 if f32 else _bf16
    folded = folded and not f32
    for i, blk in enumerate(tr.resblocks):
        del x3_exps[:]                                          # (filled in the order qkv, out, fc, proj below)
        t = dict(
            ln1_g=_f32(blk.ln_1.weight), ln1_b=_f32(blk.ln_1.bias), ln2_g=_f32(blk.ln_2.weight), ln2_b=_f32(blk.ln_2.bias),
            qkv_w=wcast(blk.attn.in_proj_weight), out_w=wcast(blk.attn.out_proj.weight),
            fc_w=wcast(blk.mlp.c_fc.weight), proj_w=wcast(blk.mlp.c_proj.weight),
            qkv_b=_f32(blk.attn.in_proj_bias), out_b=_f32(blk.attn.out_proj.bias),
            fc_b=_f32(blk.mlp.c_fc.bias), proj_b=_f32(blk.mlp.c_proj.bias))
        # ln_1 folded into in_proj, ln_2 into c_fc (keds_fold_layernorm): the tower then runs without LayerNorm passes.
        # The row statistics cross workgroups as 64-bit fixed-point integer atomics (order independent: reproducible bits);
        # KEDS_DETERMINISTIC=1 keeps the separate LayerNorm kernels as an A/B reference.
        lib = load()
        folds = (("qkv", blk.attn.in_proj_weight, t["qkv_b"], ("ln1_g", "ln1_b")),
                 ("fc", blk.mlp.c_fc.weight, t["fc_b"], ("ln2_g", "ln2_b")))
        if not folded:
            folds = ()
        for name, lin_w, lin_b, ln in folds:
            w32 = _f32(lin_w)
            n, k = w32.shape
            wf = torch.empty((n, k), dtype=torch.float16, device=w32.device)     # fp16: multiplies the fp16 residual stream
            bc = torch.empty(2 * n, dtype=torch.float32, device=w32.device)
            check(lib.keds_fold_layernorm_ex(ptr(w32), ptr(lin_b), ptr(t[ln[0]]), ptr(t[ln[1]]), n, k, ptr(wf), 1, ptr(bc),
                                             stream()), "keds_fold_layernorm")
            t[name + "_wf"], t[name + "_bc"] = wf, bc
        if fp8 and folds:
            # BASELINE config 5: MXFP8 copies of the four weights (in_proj / c_fc with their LayerNorm folded in)
            for name, lin_w, lin_b, ln in (("qkv", blk.attn.in_proj_weight, t["qkv_b"], ("ln1_g", "ln1_b")),
                                           ("out", blk.attn.out_proj.weight, t["out_b"], None),
                                           ("fc", blk.mlp.c_fc.weight, t["fc_b"], ("ln2_g", "ln2_b")),
                                           ("proj", blk.mlp.c_proj.weight, t["proj_b"], None)):
                w32 = _f32(lin_w)
                n, k = w32.shape
                q8 = torch.empty((n, k), dtype=torch.uint8, device=w32.device)
                s8 = torch.empty((k // 128, n, 4), dtype=torch.uint8, device=w32.device)
                bc8 = torch.empty(2 * n, dtype=torch.float32, device=w32.device)
                check(lib.keds_fold_layernorm_mxfp8(ptr(w32), ptr(lin_b), ptr(t[ln[0]]) if ln else None,
                                                    ptr(t[ln[1]]) if ln else None, n, k, n, ptr(q8), ptr(s8), ptr(bc8),
                                                    stream()), "keds_fold_layernorm_mxfp8")
                t[name + "_q8"], t[name + "_s8"] = q8, s8
                if ln:
                    t[name + "_bc8"] = bc8
        torch.cuda.current_stream().synchronize()          # the fp32 temporaries die here
        for k, v in t.items():
            setattr(blocks[i], k, ptr(v))
        if f32 == 2:
            blocks[i].x3_exp = (C.c_int32 * 4)(*x3_exps)
        keep.append(t)
    keep.append(blocks)
    return _lib.TowerParams(tr.width, tr.layers, tr.heads, seq, 1 if causal else 0, blocks, 1 if fp8 else 0,
                            1 if cls_only else 0, int(f32))


class _Packed:
    """bf16/fp32 device copies of the weights in the layout the kernels want + the ABI structs."""

    def __init__(self, clip: "CLIP", folded: bool = True):
        self.keep: list = []
        self.folded = folded
        v = clip.visual
        width = v.conv1.weight.shape[0]
        P = v.patch_size
        kreal = 3 * P * P
        self.kpad = (kreal + 63) // 64 * 64
        prec = getattr(clip, "precision", "bf16")
        f32 = 2 if prec == "fp32x3" else 1 if prec == "fp32" else 0      # (2: the block GEMMs on split fp16 operands, the rest as 1)
        wdt = torch.float32 if f32 else torch.bfloat16
        wcast = _f32 if f32 else _bf16
        self.f32 = f32
        conv = torch.zeros((width, self.kpad), dtype=wdt, device=v.conv1.weight.device)
        conv[:, :kreal] = v.conv1.weight.detach().reshape(width, kreal).to(wdt)
        g = v.input_resolution // P
        t = dict(conv_w=conv, class_emb=_f32(v.class_embedding), pos_emb=_f32(v.positional_embedding),
                 ln_pre_g=_f32(v.ln_pre.weight), ln_pre_b=_f32(v.ln_pre.bias), ln_post_g=_f32(v.ln_post.weight),
                 ln_post_b=_f32(v.ln_post.bias), proj_t=wcast(v.proj.detach().t()))
        self.keep.append(t)
        fp8 = getattr(clip, "precision", "bf16") == "fp8"
        self.vit = _lib.VitParams(_pack_tower(v.transformer, g * g + 1, False, self.keep, cls_only=True, fp8=fp8, folded=folded,
                                              f32=f32),
                                  v.input_resolution, P,
                                  self.kpad, v.output_dim, *[ptr(t[k]) for k in (
                                      "conv_w", "class_emb", "pos_emb", "ln_pre_g", "ln_pre_b", "ln_post_g",
                                      "ln_post_b", "proj_t")])
        tt = dict(token_emb=_f32(clip.token_embedding.weight), pos_emb=_f32(clip.positional_embedding),
                  ln_final_g=_f32(clip.ln_final.weight), ln_final_b=_f32(clip.ln_final.bias),
                  proj_t=wcast(clip.text_projection.detach().t()))
        self.keep.append(tt)
        self.text = _lib.TextParams(_pack_tower(clip.transformer, clip.context_length, True, self.keep,
                                                fp8=fp8 and clip.transformer.width % 256 == 0, folded=folded, f32=f32),
                                    clip.vocab_size, clip.embed_dim,
                                    *[ptr(tt[k]) for k in ("token_emb", "pos_emb", "ln_final_g", "ln_final_b", "proj_t")])
        self.device = conv.device


TEXT_PACKED = True          # the text tower on packed rows when captions end at different columns (tests switch it off for an A/B)
GUARD_EAGER_PASSES = 8      # numerics = "auto": tower passes whose guard flag is read back at once (see CLIP.__init__)


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: Union[Tuple[int, int, int, int], int],
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int,
                 extra_transformer_layers: int = 0, share_projection_layer: bool = True):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet visual towers are out of scope (ViT only)")
        if extra_transformer_layers:
            raise NotImplementedError("extra_transformer_layers is unused by the retrieval path")
        self.embed_dim, self.context_length = embed_dim, context_length
        self.share_projection_layer, self.has_extra = share_projection_layer, False
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer_width = transformer_width
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.end_id = vocab_size - 1
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.initialize_parameters()
        self._packed: Optional[_Packed] = None
        self._ws = _lib.Workspace()
        self.precision = os.environ.get("KEDS_PRECISION", "bf16")
        # "auto": the fast tower flow (fp16 residual stream, LayerNorm folded into the GEMMs) guarded by a device flag --
        # a row with |mean|/std > 32 or non-finite statistics re-runs the pass on the fp32-stream flow with stand-alone
        # LayerNorm and keeps the model there (keds_hip.h, keds_numerics_guard_set); "fast": no guard; "safe": always the
        # fp32-stream flow (what KEDS_DETERMINISTIC=1 selected before).
        self.numerics = "safe" if os.environ.get("KEDS_DETERMINISTIC", "0") == "1" else "auto"      # (set_numerics() changes it)
        self.numerics_tripped = False
        self._guard: Optional[torch.Tensor] = None
        # "auto" reads the guard flag back EAGERLY (one host-device sync per pass, the pass is re-run on the safe flow when it
        # tripped) for the first GUARD_EAGER_PASSES passes after the weights were packed -- what trips the guard is a property
        # of the weights far more than of one batch -- and LAZILY afterwards: the flag travels to pinned host memory behind
        # the pass, is looked at when the next pass is enqueued (or by numerics_sync()), and a late trip warns, switches the
        # model to the safe flow for every later pass and sets `numerics_late_trip` (the passes since the last check ran on
        # the fast flow: re-run them if their accuracy matters).  The host then never waits for a tower: the search of batch
        # i and the tower of batch i+1 are enqueued while batch i is still running (round-2 advisor finding).
        self.numerics_late_trip = False
        self._guard_eager_left = GUARD_EAGER_PASSES
        self._guard_host: Optional[torch.Tensor] = None
        self._guard_event = None

    # ---- init (same distributions as model.py:511-541) ------------------------------------------
    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, L = self.transformer.width, self.transformer.layers
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=w ** -0.5)
            nn.init.normal_(blk.attn.out_proj.weight, std=w ** -0.5 * (2 * L) ** -0.5)
            nn.init.normal_(blk.mlp.c_fc.weight, std=(2 * w) ** -0.5)
            nn.init.normal_(blk.mlp.c_proj.weight, std=w ** -0.5 * (2 * L) ** -0.5)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    def build_attention_mask(self):
        m = torch.full((self.context_length, self.context_length), float("-inf"))
        return m.triu_(1)           # kept for API parity; the kernel applies the causal mask itself

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    # ---- weight packing -------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def repack(self):
        """Re-read the parameters (call after mutating weights in place)."""
        self._packed = None

    def set_precision(self, precision: str = "bf16"):
        """"bf16" (default: bf16 / fp16 GEMM operands, fp32 accumulate), "fp8" (BASELINE config 5: the image tower's GEMMs on
        MXFP8 operands -- OCP e4m3 with an e8m0 scale per 32 elements; needs vision width % 256 == 0; the text tower follows
        when its width is a multiple of 256 too, e.g. 768), or "fp32": the reference's own evaluation arithmetic
        (eval_retrieval.py:108-109, `--precision` params.py:227-232) -- no operand is rounded, every product runs on the
        f32-input matrix instruction, residual stream / LayerNorm / attention stay fp32 (csrc/f32path.hip).  About a tenth
        of the default flow's throughput; embeddings agree with the fp32 reference to ~1e-6 and Recall@k is equal.
        "fp32x3" (round 5): the same fp32 flow with the four GEMMs of every block on SPLIT fp16 operands -- x = hi + lo
        (22 significant bits), hi.hi + hi.lo + lo.hi on the fp16 matrix instruction, fp32 accumulate: fp32-grade embeddings
        (Recall@k equal) at more than twice "fp32"'s throughput; values beyond the fp16 range (|x| >= 65504) send the model
        back to "fp32" by themselves."""
        if precision not in ("bf16", "fp8", "fp32", "fp32x3"):
            raise ValueError("precision must be 'bf16', 'fp8', 'fp32' or 'fp32x3'")
        if precision == "fp8" and (self.visual.transformer.width % 256 != 0 or self.numerics == "safe"):
            raise ValueError("fp8 needs a vision width that is a multiple of 256 and the folded LayerNorm path")
        self.precision = precision
        self._packed = None
        return self

    def _engine(self) -> _Packed:
        _lib.require_gpu()
        load()
        if not self.visual.conv1.weight.is_cuda:
            raise RuntimeError("keds_amd.CLIP: move the model to the GPU first (model.cuda()); no CPU path exists")
        folded = self.numerics != "safe" and not self.numerics_tripped
        if self._packed is None or self._packed.folded != folded:
            self._packed = _Packed(self, folded=folded)
            self._guard_eager_left = GUARD_EAGER_PASSES
        _lib.ensure_gemm_workspace(self._packed.device)
        return self._packed

    def set_numerics(self, mode: str = "auto"):
        if mode not in ("auto", "fast", "safe"):
            raise ValueError("numerics must be 'auto', 'fast' or 'safe'")
        if mode == "safe" and self.precision == "fp8":
            raise ValueError("fp8 needs the folded LayerNorm path")
        self._guard_poll(wait=True)
        self.numerics, self.numerics_tripped, self.numerics_late_trip = mode, False, False
        self._guard_eager_left = GUARD_EAGER_PASSES
        return self

    def _guard_trip(self, late: bool):
        import warnings
        warnings.warn("keds_amd.CLIP: activations left the range the fast tower flow is accurate in (|row mean|/std > 32 or a "
                      "non-finite fp16 residual); " + ("the passes since the last check ran on the fast flow (numerics_late_trip); "
                      "later passes run" if late else "re-running") + " on the fp32-stream flow and staying there", RuntimeWarning)
        self._guard.zero_()
        self.numerics_tripped = True
        self.numerics_late_trip = self.numerics_late_trip or late

    def _guard_poll(self, wait: bool) -> None:
        """Look at the flag copy that travelled behind the last lazily checked pass (blocking only when `wait`)."""
        ev = self._guard_event
        if ev is None or not (wait or ev.query()):
            return
        if wait:
            ev.synchronize()
        self._guard_event = None
        if int(self._guard_host[0]) != 0 and not self.numerics_tripped:
            self._guard_trip(late=True)

    def numerics_sync(self) -> bool:
        """Block until every lazily checked pass has been verified; True when the guard tripped at any point (the model is
        then on the safe flow; `numerics_late_trip` says whether passes had already been returned from the fast flow)."""
        self._guard_poll(wait=True)
        return self.numerics_tripped

    def numerics_checked(self, fn):
        """Run `fn()` (any number of encoder passes whose outputs the caller KEEPS: a feature-extraction or evaluation loop)
        and return its result only after every pass in it has been verified: the lazily checked guard is synchronised
        behind the loop, and when it tripped while passes of this loop had already been returned from the fast flow
        (`numerics_late_trip`), the loop is run again -- the model is on the fp32-stream flow by then, so the second run is
        clean.  Loops that collect features must not end on an unverified pass (retrieval.py uses this everywhere)."""
        before = self.numerics_late_trip
        out = fn()
        self.numerics_sync()
        if self.numerics_late_trip and not before:
            out = fn()
            self.numerics_sync()
        return out

    def _guarded(self, run):
        """Run one tower pass (`run(engine)` enqueues it and returns the output tensor) under the numerics guard."""
        # (a stream that is being captured into a hipGraph can be neither synchronised nor queried -- an event query inside a
        # global-mode capture invalidates it: no flag read there; call numerics_sync() before the capture, or capture a
        # model on set_numerics("safe"))
        if torch.cuda.is_current_stream_capturing():
            return run(self._engine())
        self._guard_poll(wait=False)
        eng = self._engine()
        if self.precision == "fp32x3":
            # split fp16 operands hold |x| < 65504 only: the split / LayerNorm kernels raise the guard flag beyond that, and a
            # non-finite output says the MLP hidden layer went there -- either way this pass and all later ones run on the
            # f32-input matrix instruction instead ("fp32": slower, no range limit).  Checked at once: a pass takes ~80 ms.
            if self._guard is None or self._guard.device != eng.device:
                self._guard = torch.zeros(1, dtype=torch.int32, device=eng.device)
            lib = load()
            check(lib.keds_numerics_guard_set(ptr(self._guard)), "keds_numerics_guard_set")
            try:
                out = run(eng)
            finally:
                check(lib.keds_numerics_guard_set(None), "keds_numerics_guard_set")
            if int(self._guard.item()) == 0 and bool(torch.isfinite(out).all()):
                return out
            self._guard.zero_()
            self.x3_range_trips = getattr(self, "x3_range_trips", 0) + 1
            self.set_precision("fp32")
            return run(self._engine())
        if self.numerics != "auto" or not eng.folded or self.precision in ("fp8", "fp32"):
            return run(eng)
        if self._guard is None or self._guard.device != eng.device:
            self._guard = torch.zeros(1, dtype=torch.int32, device=eng.device)
        lib = load()
        check(lib.keds_numerics_guard_set(ptr(self._guard)), "keds_numerics_guard_set")
        try:
            out = run(eng)
        finally:
            check(lib.keds_numerics_guard_set(None), "keds_numerics_guard_set")
        if self._guard_eager_left <= 0:                      # lazy: the flag follows the pass to pinned host memory
            if self._guard_host is None:
                self._guard_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._guard_event_obj = torch.cuda.Event()
            if self._guard_event is not None:                # at most one copy in flight: the flag is sticky, nothing is lost
                self._guard_poll(wait=True)
            self._guard_host.copy_(self._guard, non_blocking=True)
            self._guard_event_obj.record()
            self._guard_event = self._guard_event_obj
            return out
        self._guard_eager_left -= 1
        if int(self._guard.item()) == 0:
            return out
        self._guard_trip(late=False)
        return run(self._engine())

    # ---- encoders ------------------------------------------------------------------------------------
    def encode_image(self, image, mid_feature=False, mask_token=False, normalize: bool = False):
        """model.py:569-575 -> VisualTransformer.forward :393-415.  image [B,3,R,R] -> [B, embed_dim]."""
        if mid_feature:
            raise NotImplementedError("mid_feature is a training/ablation path (out of scope)")
        eng = self._engine()
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != self.visual.input_resolution \
                or image.shape[3] != self.visual.input_resolution:
            raise RuntimeError(f"expected images [B,3,{self.visual.input_resolution},{self.visual.input_resolution}], "
                               f"got {tuple(image.shape)}")
        img = image.to(eng.device, dtype=torch.float32).contiguous()
        B = img.shape[0]
        if B == 0:                                    # an empty last batch of a loader: the reference returns [0, embed_dim]
            return torch.empty((0, self.embed_dim), dtype=self.dtype, device=eng.device)
        if self.precision == "fp32x3":
            # the split-operand GEMMs address their planes with 32-bit byte offsets (keds_gemm_x3): the MLP hidden planes of one call
            # must stay below 2 GiB -- rows * 4 * width * 2 bytes.  Larger batches run in chunks (rows are independent between samples).
            rows_max = (1 << 31) // (8 * self.visual.transformer.width) - 512
            chunk = max(1, rows_max // ((self.visual.input_resolution // self.visual.conv1.kernel_size[0]) ** 2 + 1))
            if B > chunk:
                return torch.cat([self.encode_image(img[i:i + chunk], normalize=normalize) for i in range(0, B, chunk)])
        lib = load()

        def run(eng):
            nbytes = lib.keds_vit_workspace_bytes(C.byref(eng.vit), B)
            ws = self._ws.get(nbytes, eng.device)
            out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=eng.device)
            check(lib.keds_vit_run(C.byref(eng.vit), ptr(img), B, ptr(out), 1 if normalize else 0, ptr(ws), ws.numel(),
                                   stream()), "keds_vit_run")
            return out
        return self._guarded(run).to(self.dtype)

    def _host_tokens(self, text: torch.Tensor) -> torch.Tensor:
        """The token rows on the HOST for the argument checks the reference's own indexing performs (one EOT per row, ids
        inside the table, the split token, the read-out column): one device-to-host copy per call when the caller passed
        device tokens (none for the CPU tokens a DataLoader yields) instead of five blocking reductions on the device."""
        th = text.cpu() if text.is_cuda else text
        # nn.Embedding raises IndexError on an id outside the table (model.py:579); the gather kernel does not check
        if th.numel() and (int(th.min()) < 0 or int(th.max()) >= self.vocab_size):
            raise IndexError(f"token id outside [0, {self.vocab_size})")
        return th

    def _eot_columns(self, text: torch.Tensor) -> torch.Tensor:
        hits = text == self.end_id
        if not bool((hits.sum(dim=1) == 1).all()):
            # the reference indexes nonzero()[:,1] with arange(B): anything but one EOT per row fails there
            raise IndexError("every token row must contain exactly one EOT token")
        return hits.to(torch.int32).argmax(dim=1)

    def _run_text(self, text, readout, img_tokens, insert_col, normalize):
        eng = self._engine()
        lib = load()
        B = text.shape[0]
        if B == 0:
            return torch.empty((0, self.embed_dim), dtype=self.dtype, device=eng.device)
        tok = text.to(eng.device, dtype=torch.int32).contiguous()           # (ids were range-checked on the host: _host_tokens)
        ro = readout.to(torch.int32).to(eng.device, non_blocking=True).contiguous()
        it = None if img_tokens is None else img_tokens.to(eng.device, dtype=torch.float32).contiguous()
        n_tok = 0 if it is None else it.shape[1]
        # columns to the right of the last read-out column cannot reach any read-out under the causal mask
        # (model.py:543-549): the host knows the read-out columns (they were derived from the host copy of the tokens), so
        # the library cuts the sequence there (keds_text_run_ex)
        seq_used = int(readout.max()) + 1
        # ... and columns to the right of a caption's OWN read-out column cannot reach that caption's read-out: when the captions of
        # a batch end at different columns the tower runs on PACKED rows, sample b owning len_b = read-out column + 1 of them
        # (keds_text_run_packed, round 6) -- sum(len_b) rows instead of B * max(len_b).  Worth it from an eighth fewer rows; the
        # MXFP8 tower and KEDS_TEXT_TRIM=0 keep the rectangular layout.
        if self.precision == "fp32x3":                       # (see encode_image: the planes of one call stay below 2 GiB)
            chunk = max(1, ((1 << 31) // (8 * self.transformer.width) - 512) // self.context_length)
            if B > chunk:
                return torch.cat([self._run_text(text[i:i + chunk], readout[i:i + chunk], None if img_tokens is None else img_tokens[i:i + chunk],
                                                 insert_col, normalize) for i in range(0, B, chunk)])
        lens = readout.to(torch.int64).cpu() + 1
        rows_total = int(lens.sum())
        packed = None
        if self.precision != "fp8" and TEXT_PACKED and lib.keds_text_trim_mode() == 1 and rows_total * 8 <= B * seq_used * 7:
            off = torch.zeros(B + 1, dtype=torch.int64)
            off[1:] = torch.cumsum(lens, 0)
            both = torch.cat([off, off[:-1] + readout.to(torch.int64).cpu()]).to(torch.int32)
            both = both.to(eng.device, non_blocking=True)
            packed = (both[:B + 1], both[B + 1:])

        def run(eng):
            nbytes = lib.keds_text_workspace_bytes(C.byref(eng.text), B)
            ws = self._ws.get(nbytes, eng.device)
            out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=eng.device)
            if packed is not None and self.precision != "fp8":
                check(lib.keds_text_run_packed(C.byref(eng.text), ptr(tok), ptr(packed[0]), ptr(packed[1]), rows_total, seq_used,
                                               ptr(it), n_tok, int(insert_col), B, ptr(out), 1 if normalize else 0, ptr(ws),
                                               ws.numel(), stream()), "keds_text_run_packed")
                return out
            check(lib.keds_text_run_ex(C.byref(eng.text), ptr(tok), ptr(ro), ptr(it), n_tok, int(insert_col), B, seq_used,
                                       ptr(out), 1 if normalize else 0, ptr(ws), ws.numel(), stream()), "keds_text_run")
            return out
        return self._guarded(run).to(self.dtype)

    def encode_text(self, text, normalize: bool = False):
        """model.py:577-590.  text int [B, L] -> [B, embed_dim]; read-out at the EOT column."""
        if text.dim() != 2 or text.shape[1] != self.context_length:
            raise RuntimeError(f"expected tokens [B,{self.context_length}], got {tuple(text.shape)}")
        if text.shape[0] == 0:
            return self._run_text(text, text.new_zeros(0), None, 0, normalize)
        return self._run_text(text, self._eot_columns(self._host_tokens(text)), None, 0, normalize)

    def encode_text_img_retrieval(self, text, img_tokens, split_ind=4, repeat=True, normalize: bool = False):
        """model.py:808-851.  The first `split_ind` token of ROW 0 is replaced by the 2 or 3 pseudo tokens of
        every row, the tail shifts right, read-out at EOT column + n_tok - 1."""
        if isinstance(img_tokens, tuple):
            raise NotImplementedError("tuple img_tokens (multi-insert) is not used by the retrieval path")
        b_size = img_tokens.shape[0]
        if repeat:
            text = text.repeat(b_size, 1)
        if text.shape[0] != b_size or text.shape[1] != self.context_length:
            raise RuntimeError(f"token rows {tuple(text.shape)} do not match {b_size} pseudo-token rows")
        n_tok = img_tokens.shape[1]
        if n_tok not in (2, 3):
            raise RuntimeError("img_tokens must carry 2 or 3 pseudo tokens per row (sequence length would not be "
                               f"{self.context_length})")
        if img_tokens.shape[2] != self.transformer_width:
            raise RuntimeError("pseudo-token width does not match the text transformer")
        th = self._host_tokens(text)
        where = (th[0] == int(split_ind)).nonzero()
        if where.numel() == 0:
            raise IndexError("split token not present in text[0]")
        ins = int(where[0])
        readout = self._eot_columns(th) + (n_tok - 1)
        if int(readout.max()) >= self.context_length:
            raise IndexError("read-out row beyond the context length")
        return self._run_text(text, readout, img_tokens, ins, normalize)

    def encode_text_img_train(self, text, img_tokens, split_ind=4, repeat=True, normalize: bool = False):
        """model.py:853-892 (the splice evaluate_fashion calls, eval_utils.py:957,969): the THREE token positions
        starting at the first `split_ind` of row 0 are overwritten by the pseudo tokens, nothing shifts, read-out at
        the EOT column.  Identical to the retrieval splice of a token row with positions ins+1, ins+2 deleted, which
        is how it runs here (integer column shuffling on the host side, same kernels).  `repeat` is ignored, as in
        the reference."""
        if img_tokens.dim() != 3 or img_tokens.shape[1] != 3:
            raise RuntimeError("encode_text_img_train needs exactly 3 pseudo tokens per row (sequence length would not "
                               f"be {self.context_length})")               # reference: size mismatch at model.py:883
        if text.dim() != 2 or text.shape[0] != img_tokens.shape[0] or text.shape[1] != self.context_length:
            raise RuntimeError(f"token rows {tuple(text.shape)} do not match {img_tokens.shape[0]} pseudo-token rows")
        th = self._host_tokens(text)
        where = (th[0] == int(split_ind)).nonzero()
        if where.numel() == 0:
            raise IndexError("split token not present in text[0]")
        ins = int(where[0])
        if ins + 3 > self.context_length:
            raise RuntimeError("no room for 3 pseudo tokens after the split token")
        eot = self._eot_columns(th)
        if bool(((eot >= ins) & (eot < ins + 3)).any()):
            raise IndexError("the EOT token lies inside the overwritten span")
        L = self.context_length
        squeezed = torch.zeros_like(text)
        squeezed[:, :ins + 1] = text[:, :ins + 1]
        squeezed[:, ins + 1:L - 2] = text[:, ins + 3:]
        return self._run_text(squeezed, eot, img_tokens, ins, normalize)

    def get_text_tokens(self, text):
        raise NotImplementedError("get_text_tokens is not on the retrieval path")

    def forward(self, image, text, extra=False):
        """model.py:894-911 (extra=True is broken in the reference: undefined encode_text_extra)."""
        if extra:
            raise NotImplementedError("extra text tower is not part of the retrieval path")
        if image is None:
            return self.encode_text(text)
        if text is None:
            return self.encode_image(image)
        return self.encode_image(image, normalize=True), self.encode_text(text, normalize=True), self.logit_scale.exp()


# ---------------------------------------------------------------------------------------------------
# knowledge-injection modules
# ---------------------------------------------------------------------------------------------------
class IM2TEXT(nn.Module):
    """model.py:105-123: (Linear, Dropout, ReLU) x n_layer then fc_out; eval-mode forward on HIP."""

    def __init__(self, embed_dim=512, middle_dim=512, output_dim=512, n_layer=2, dropout=0.1):
        super().__init__()
        if not 1 <= n_layer <= 4:
            raise ValueError("n_layer must be in [1,4]")
        self.fc_out = nn.Linear(middle_dim, output_dim)
        layers, dim = [], embed_dim
        for _ in range(n_layer):
            layers.append(nn.Sequential(nn.Linear(dim, middle_dim), nn.Dropout(dropout), nn.ReLU()))
            dim = middle_dim
        self.layers = nn.Sequential(*layers)
        self.embed_dim, self.middle_dim, self.output_dim, self.n_layer = embed_dim, middle_dim, output_dim, n_layer
        self._packed = None
        self._ws = _lib.Workspace()

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def params(self, f32: bool = False) -> _lib.Im2TextParams:
        """f32: the weights as fp32 arrays (the fp32-accurate flow, keds_knowledge_run_f32); default bf16 GEMM operands."""
        _lib.require_gpu()
        if not self.fc_out.weight.is_cuda:
            raise RuntimeError("keds_amd.IM2TEXT: move the module to the GPU first; no CPU path exists")
        if self._packed is None:
            self._packed = {}
        if f32 not in self._packed:
            keep = []
            wcast = _f32 if f32 else _bf16
            p = _lib.Im2TextParams()
            p.dim_in, p.middle, p.dim_out, p.n_layer = self.embed_dim, self.middle_dim, self.output_dim, self.n_layer
            for i, blk in enumerate(self.layers):
                w, b = wcast(blk[0].weight), _f32(blk[0].bias)
                keep += [w, b]
                p.w[i], p.b[i] = ptr(w), ptr(b)
            ow, ob = wcast(self.fc_out.weight), _f32(self.fc_out.bias)
            keep += [ow, ob]
            p.out_w, p.out_b = ptr(ow), ptr(ob)
            self._packed[f32] = (p, keep)
        _lib.ensure_gemm_workspace(self.fc_out.weight.device)
        return self._packed[f32][0]

    def forward(self, x: torch.Tensor):
        if self.training:
            raise RuntimeError("keds_amd.IM2TEXT runs the eval-mode forward only (dropout = identity); call .eval()")
        p = self.params()
        lib = load()
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).to(dtype=torch.float32).contiguous()
        rows = x2.shape[0]
        nbytes = lib.keds_im2text_workspace_bytes(C.byref(p), rows)
        ws = self._ws.get(nbytes, x2.device)
        out = torch.empty((rows, self.output_dim), dtype=torch.float32, device=x2.device)
        check(lib.keds_im2text_forward(C.byref(p), ptr(x2), rows, ptr(out), ptr(ws), ws.numel(), stream()),
              "keds_im2text_forward")
        return out.reshape(*shape[:-1], self.output_dim).to(x.dtype)


class CrossAttention(nn.Module):
    """Parameter holder for one layer (model.py:37-54)."""

    def __init__(self, q_dim, k_dim, v_dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        if dim_head != 64:
            raise ValueError("the HIP cross-attention core is built for dim_head = 64")
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(q_dim, inner, bias=True)
        self.to_k = nn.Linear(k_dim, inner, bias=True)
        self.to_v = nn.Linear(v_dim, inner, bias=True)
        self.to_out = nn.Sequential(nn.Linear(inner, q_dim), nn.Dropout(dropout))


class CrossFormer(nn.Module):
    """model.py:81-101: q chained through `num_layers` CrossAttention layers, k and v fixed."""

    def __init__(self, q_dim, k_dim, v_dim, num_layers=1, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        if not (q_dim == k_dim == v_dim):
            raise ValueError("the HIP path needs q_dim == k_dim == v_dim")
        self._num_layers, self.dim, self.heads = num_layers, q_dim, heads
        self.cross_layers = nn.ModuleList([CrossAttention(q_dim, k_dim, v_dim, heads, dim_head, dropout)
                                           for _ in range(num_layers)])
        self._packed = self._packed32 = None
        self._ws = _lib.Workspace()

    def _apply(self, fn, *a, **k):
        self._packed = self._packed32 = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._packed = self._packed32 = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def params(self, f32: bool = False) -> _lib.CrossFormerParams:
        """f32: the per-layer weights as fp32 arrays, nothing fused (keds_knowledge_run_f32); default bf16 GEMM operands."""
        _lib.require_gpu()
        if not self.cross_layers[0].to_q.weight.is_cuda:
            raise RuntimeError("keds_amd.CrossFormer: move the module to the GPU first; no CPU path exists")
        if f32:
            if getattr(self, "_packed32", None) is None:
                keep = []
                arr = (_lib.CrossLayerParams * self._num_layers)()
                for i, l in enumerate(self.cross_layers):
                    t = dict(wq=_f32(l.to_q.weight), wk=_f32(l.to_k.weight), wv=_f32(l.to_v.weight), wo=_f32(l.to_out[0].weight),
                             bq=_f32(l.to_q.bias), bk=_f32(l.to_k.bias), bv=_f32(l.to_v.bias), bo=_f32(l.to_out[0].bias))
                    for k, v in t.items():
                        setattr(arr[i], k, ptr(v))
                    keep.append(t)
                self._packed32 = (_lib.CrossFormerParams(self.dim, self.heads, self._num_layers, arr, None), keep, arr)
            return self._packed32[0]
        if self._packed is None:
            keep = []
            arr = (_lib.CrossLayerParams * self._num_layers)()
            for i, l in enumerate(self.cross_layers):
                t = dict(wq=_bf16(l.to_q.weight), wk=_bf16(l.to_k.weight), wv=_bf16(l.to_v.weight),
                         wo=_bf16(l.to_out[0].weight), bq=_f32(l.to_q.bias), bk=_f32(l.to_k.bias),
                         bv=_f32(l.to_v.bias), bo=_f32(l.to_out[0].bias))
                for k, v in t.items():
                    setattr(arr[i], k, ptr(v))
                keep.append(t)
            p = _lib.CrossFormerParams(self.dim, self.heads, self._num_layers, arr, None)
            # launch-saving re-arrangement (keds_hip.h, keds_crossformer_fused): one k/v GEMM for all layers, and every
            # later layer's query projection folded with the previous output projection
            lib = load()
            if self._num_layers <= 8:
                dev = self.cross_layers[0].to_q.weight.device
                buf = torch.empty(int(lib.keds_crossformer_fused_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
                fused = _lib.CrossFormerFused()
                check(lib.keds_crossformer_fuse(C.byref(p), ptr(buf), buf.numel(), C.byref(fused), stream()),
                      "keds_crossformer_fuse")
                torch.cuda.current_stream().synchronize()
                p.fused = C.pointer(fused)
                keep += [buf, fused]
            self._packed = (p, keep, arr)
        _lib.ensure_gemm_workspace(self.cross_layers[0].to_q.weight.device)
        return self._packed[0]

    def forward(self, q, k, v):
        """q [B,1,D], k [B,K,D], v [B,K,D] -> [B,1,D]"""
        p = self.params()
        if q.dim() != 3 or q.shape[1] != 1:
            raise RuntimeError("the HIP CrossFormer handles single-query attention: q must be [B,1,D]")
        if k.shape != v.shape or k.shape[0] != q.shape[0] or k.shape[2] != self.dim:
            raise RuntimeError("k and v must both be [B,K,D]")
        B, K = k.shape[0], k.shape[1]
        lib = load()
        qf = q.reshape(B, self.dim).to(dtype=torch.float32).contiguous()
        kf = k.to(dtype=torch.float32).contiguous()
        vf = kf if v is k else v.to(dtype=torch.float32).contiguous()
        nbytes = lib.keds_crossformer_workspace_bytes(C.byref(p), B, K)
        ws = self._ws.get(nbytes, qf.device)
        out = torch.empty((B, self.dim), dtype=torch.float32, device=qf.device)
        check(lib.keds_crossformer_forward(C.byref(p), ptr(qf), ptr(kf), ptr(vf), B, K, ptr(out), ptr(ws), ws.numel(),
                                           stream()), "keds_crossformer_forward")
        return out.reshape(B, 1, self.dim).to(q.dtype)


class KnowledgeStream:
    """One stream (img2text + retrieval_fuse + text_condition) fused into a single library call
    (eval_utils.py:661-672): tokens = [fuse(m,I,I), cond(m,T,T), m]."""

    def __init__(self, img2text: IM2TEXT, retrieval_fuse: CrossFormer, text_condition: CrossFormer):
        self.img2text, self.retrieval_fuse, self.text_condition = img2text, retrieval_fuse, text_condition
        self._ws = _lib.Workspace()

    def __call__(self, q: torch.Tensor, nbr_img: torch.Tensor, nbr_txt: torch.Tensor, precision: str = "bf16") -> torch.Tensor:
        """precision "fp32": the fp32-accurate flow (no operand rounding, keds_knowledge_run_f32) -- what
        compose_query_features passes when the CLIP model is on set_precision("fp32")."""
        f32 = precision in ("fp32", "fp32x3")
        kp = _lib.KnowledgeParams(self.img2text.params(f32), self.retrieval_fuse.params(f32), self.text_condition.params(f32))
        lib = load()
        B, K, dim = nbr_img.shape
        qf = q.to(dtype=torch.float32).contiguous()
        ni = nbr_img.to(dtype=torch.float32).contiguous()
        nt = nbr_txt.to(dtype=torch.float32).contiguous()
        if f32:
            nbytes = lib.keds_knowledge_f32_workspace_bytes(C.byref(kp), B, K)
            ws = self._ws.get(nbytes, qf.device)
            out = torch.empty((B, 3, dim), dtype=torch.float32, device=qf.device)
            check(lib.keds_knowledge_run_f32(C.byref(kp), ptr(qf), ptr(ni), ptr(nt), B, K, ptr(out), ptr(ws), ws.numel(), stream()),
                  "keds_knowledge_run_f32")
            return out
        nbytes = lib.keds_knowledge_workspace_bytes(C.byref(kp), B, K)
        ws = self._ws.get(nbytes, qf.device)
        out = torch.empty((B, 3, dim), dtype=torch.float32, device=qf.device)
        check(lib.keds_knowledge_run(C.byref(kp), ptr(qf), ptr(ni), ptr(nt), B, K, ptr(out), ptr(ws), ws.numel(),
                                         stream()), "keds_knowledge_run")
        return out


# ---------------------------------------------------------------------------------------------------
# builders (model.py:927-991)
# ---------------------------------------------------------------------------------------------------
def convert_weights(model: nn.Module):
    """fp16 cast of Linear/Conv/MHA/projection weights (model.py:927-948).  Kept for API parity: it
    changes the checkpoint dtype (`model.dtype`), the kernels always compute bf16 x bf16 -> fp32."""
    def _cast(l):
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            l.weight.data = l.weight.data.half()
            if l.bias is not None:
                l.bias.data = l.bias.data.half()
        if isinstance(l, nn.MultiheadAttention):
            for attr in ["in_proj_weight", "q_proj_weight", "k_proj_weight", "v_proj_weight", "in_proj_bias",
                         "bias_k", "bias_v"]:
                t = getattr(l, attr, None)
                if t is not None:
                    t.data = t.data.half()
        for name in ("text_projection", "proj"):
            t = getattr(l, name, None)
            if isinstance(t, torch.Tensor):
                t.data = t.data.half()
    model.apply(_cast)
    if hasattr(model, "repack"):
        model.repack()


def clip_config_from_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, int]:
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only ViT visual towers are supported")
    vw = state_dict["visual.conv1.weight"].shape[0]
    patch = state_dict["visual.conv1.weight"].shape[-1]
    grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = state_dict["ln_final.weight"].shape[0]
    return dict(
        embed_dim=state_dict["text_projection"].shape[1], image_resolution=patch * grid,
        vision_layers=len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")]),
        vision_width=vw, vision_patch_size=patch, context_length=state_dict["positional_embedding"].shape[0],
        vocab_size=state_dict["token_embedding.weight"].shape[0], transformer_width=tw, transformer_heads=tw // 64,
        transformer_layers=len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks"))))


def build_model(state_dict: dict, fp16: bool = True) -> CLIP:
    """Infer the architecture from checkpoint shapes and load it (model.py:951-991)."""
    cfg = clip_config_from_state_dict(state_dict)
    model = CLIP(**cfg)
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    if fp16:
        convert_weights(model)
    model.load_state_dict(sd)
    return model.eval()


def convert_models_to_fp32(model: nn.Module):
    """utils.py:44-48 (`--precision amp|fp32`)."""
    for p in model.parameters():
        p.data = p.data.float()
    if hasattr(model, "repack"):
        model.repack()
