"""Training step of the knowledge-injection modules on the HIP path.

Mirrors the reference's training hot path -- `get_loss_img2text_image` (src/trainer.py:44-165: device-side retrieval,
IM2TEXT + retrieval_fuse + text_condition, pseudo tokens through the FROZEN text tower, symmetric contrastive loss with
the features of all ranks as negatives, trainer.py:100-101) and the optimizer of src/main.py:215-237 (AdamW, no weight
decay on biases) -- with every forward AND backward kernel in libkeds_hip.so (keds_hip.h section 9): there is no autograd
and no torch compute here; torch holds the parameters (fp32 masters), device buffers and the process group.

Deviation from the reference, on purpose: as committed the reference splices the three pseudo tokens with
`encode_text_img`, which builds a 78-token sequence and fails; here the splice is the one its own
evaluation uses -- `encode_text_img_retrieval` on the prompt "a photo of *" (model.py:808-851): tokens at the `*`, tail
shifted, read-out at EOT + 2.

Numerics: GEMM operands (activations, weights, the gradients that feed a GEMM) are bf16 with fp32 accumulation; the
residual stream and its gradient, LayerNorm, softmax, loss and AdamW are fp32.  Parity bar (tests/test_gpu_train.py):
loss within 2e-3 relative and every parameter gradient within rel-L2 3e-2 / cosine 0.999 of torch autograd on the fp32
oracle with the same dropout masks (IM2TEXT hidden layers 6e-2 / 0.998: ReLU gates on bf16 pre-activations).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _lib, ops
from ._lib import check, load, ptr, stream
from .model import CLIP, CrossFormer, IM2TEXT

BF16, F32 = torch.bfloat16, torch.float32


def _pad(m: int) -> int:
    return (m + 127) // 128 * 128


# ---- thin wrappers over the section-9 entry points ------------------------------------------------------------------
def _gemm(a, w, bias, epi, M, out=None):
    """out[M, N] = epi(a[M, K] . w[N, K]^T + bias); a / out rows padded to 128."""
    N, K = w.shape
    if out is None:
        out = torch.zeros((_pad(M), N), dtype=F32 if epi in (_lib.EPI_BIAS_F32, _lib.EPI_BIAS_RESID_F32) else BF16, device=a.device)
    check(load().keds_gemm_bt(ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K, epi, None, 0, stream()), "keds_gemm_bt")
    return out


def _transpose(x, rows, cols, ld_out=None):
    """bf16 [cols, ld_out] = x[:rows, :cols]^T, zero padded (x fp32 or bf16, row-major with stride x.shape[1])."""
    ld_out = _pad(rows) if ld_out is None else ld_out
    out = torch.empty((cols, ld_out), dtype=BF16, device=x.device)
    check(load().keds_transpose_to_bf16(ptr(x), 1 if x.dtype == F32 else 0, x.shape[1], rows, cols, ptr(out), ld_out, stream()),
          "keds_transpose_to_bf16")
    return out


def _colsum(x, rows, cols):
    out = torch.empty(cols, dtype=F32, device=x.device)
    check(load().keds_colsum(ptr(x), 1 if x.dtype == F32 else 0, x.shape[1], rows, cols, ptr(out), 0, stream()), "keds_colsum")
    return out


def _cast(x, rows=None):
    return ops.cast_bf16(x, rows_padded=_pad(x.shape[0] if rows is None else rows))


def _grad_weight(dy, x, M):
    """dW [N, K] fp32 = dy[:M]^T . x[:M]   (dy [Mp, N], x [Mp, K]; both become K-contiguous operands by transposition)."""
    dyT = _transpose(dy, M, dy.shape[1])                 # [N, Mp]
    xT = _transpose(x, M, x.shape[1])                    # [K, Mp]
    out = _gemm(dyT, xT, None, _lib.EPI_BIAS_F32, dy.shape[1])
    return out[:dy.shape[1]]


class _Linear:
    """One trainable nn.Linear: fp32 masters + this step's bf16 operand copies (W for the forward, W^T for dX)."""

    def __init__(self, name: str, lin: torch.nn.Linear):
        self.name, self.lin = name, lin
        self.N, self.K = lin.weight.shape

    def refresh(self):
        w = self.lin.weight.detach().float().contiguous()
        self.w = ops.cast_bf16(w)                                        # [N, K]
        self.wT = _transpose(w, self.N, self.K, ld_out=self.N)           # [K, N]
        self.b = self.lin.bias.detach().float().contiguous()


class KnowledgeTrainer:
    """loss + gradients + AdamW for (img2text, retrieval_fuse, text_condition) against a frozen CLIP text tower.

        trainer = KnowledgeTrainer(model, img2text, retrieval_fuse, text_condition, lr=1e-4, wd=0.1)
        loss = trainer.step(image_features, database, prompt_tokens, id_split)      # one optimizer step

    `image_features` [B, D] are the (frozen) CLIP image embeddings of the batch, as in the reference, where the dataset
    yields precomputed features (trainer.py:44-52); `database` is `keds_amd.build_database(...)` (or a shard of it).
    With torch.distributed initialised (and aggregate=True) the normalised features of all ranks are the negatives
    (trainer.py:78-99: own rows first, remote rows carry no gradient) and the parameter gradients are averaged over the
    ranks (the reference wraps the three modules in DistributedDataParallel)."""

    def __init__(self, model: CLIP, img2text: IM2TEXT, retrieval_fuse: CrossFormer, text_condition: CrossFormer,
                 lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, wd: float = 0.1,
                 dropout: float = 0.1, topk: int = 16, aggregate: bool = True, group=None, seed: int = 0):
        _lib.require_gpu()
        self.model, self.i2t, self.fuse, self.cond = model, img2text, retrieval_fuse, text_condition
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, wd
        self.p_drop, self.topk, self.aggregate, self.group = float(dropout), topk, aggregate, group
        self.seed, self.steps = int(seed), 0
        self.lin: Dict[str, _Linear] = {}
        for i, blk in enumerate(img2text.layers):
            self.lin[f"i2t.layers.{i}.0"] = _Linear(f"i2t.layers.{i}.0", blk[0])
        self.lin["i2t.fc_out"] = _Linear("i2t.fc_out", img2text.fc_out)
        for tag, xf in (("fuse", retrieval_fuse), ("cond", text_condition)):
            for l, layer in enumerate(xf.cross_layers):
                for nm, lin in (("to_q", layer.to_q), ("to_k", layer.to_k), ("to_v", layer.to_v), ("to_out.0", layer.to_out[0])):
                    self.lin[f"{tag}.cross_layers.{l}.{nm}"] = _Linear(f"{tag}.cross_layers.{l}.{nm}", lin)
        self._opt: Dict[str, torch.Tensor] = {}
        self._frozen = None

    # ---- the frozen text tower: bf16 operands for the forward and their transposes for dX, made once -----------------
    def _tower(self):
        if self._frozen is None:
            m = self.model
            blocks = []
            for blk in m.transformer.resblocks:
                t = {}
                for nm, lin_w, lin_b in (("qkv", blk.attn.in_proj_weight, blk.attn.in_proj_bias),
                                         ("out", blk.attn.out_proj.weight, blk.attn.out_proj.bias),
                                         ("fc", blk.mlp.c_fc.weight, blk.mlp.c_fc.bias),
                                         ("proj", blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)):
                    w = lin_w.detach().float().contiguous()
                    t[nm + "_w"] = ops.cast_bf16(w)
                    t[nm + "_wT"] = _transpose(w, w.shape[0], w.shape[1], ld_out=w.shape[0])
                    t[nm + "_b"] = lin_b.detach().float().contiguous()
                for nm, ln in (("ln1", blk.ln_1), ("ln2", blk.ln_2)):
                    t[nm + "_g"], t[nm + "_b"] = ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()
                blocks.append(t)
            P = m.text_projection.detach().float().contiguous()              # [w, embed]
            self._frozen = dict(blocks=blocks, proj_t=_transpose(P, P.shape[0], P.shape[1], ld_out=P.shape[0]),   # [embed, w]
                                proj=ops.cast_bf16(P),                         # [w, embed]: the dX operand of the read-out
                                lnf_g=m.ln_final.weight.detach().float().contiguous(),
                                lnf_b=m.ln_final.bias.detach().float().contiguous(),
                                tok=m.token_embedding.weight.detach().float().contiguous(),
                                pos=m.positional_embedding.detach().float().contiguous())
        return self._frozen

    # ---- forward / backward of one CrossFormer (training: layer by layer, model.py:98-101) -----------------------------
    def _xf_forward(self, tag, xf: CrossFormer, q_bf, kv_bf, B, BK, K):
        lib, saved = load(), []
        heads = xf.heads
        qn = None
        for l in range(len(xf.cross_layers)):
            L = lambda nm: self.lin[f"{tag}.cross_layers.{l}.{nm}"]
            Q = _gemm(q_bf, L("to_q").w, L("to_q").b, _lib.EPI_BIAS_BF16, B)
            Kp = _gemm(kv_bf, L("to_k").w, L("to_k").b, _lib.EPI_BIAS_BF16, BK)
            Vp = _gemm(kv_bf, L("to_v").w, L("to_v").b, _lib.EPI_BIAS_BF16, BK)
            att = torch.zeros_like(Q)
            check(lib.keds_cross_core_fwd(ptr(Q), ptr(Kp), ptr(Vp), ptr(att), B, K, heads, stream()), "keds_cross_core_fwd")
            qn = _gemm(att, L("to_out.0").w, L("to_out.0").b, _lib.EPI_BIAS_F32, B)
            saved.append((q_bf, Q, Kp, Vp, att))
            q_bf = _cast(qn[:B])
        return qn, saved

    def _xf_backward(self, tag, xf: CrossFormer, saved, dq, kv_bf, B, BK, K, grads, dq_out, dkv_out):
        """dq fp32 [Bp, d]: gradient of the CrossFormer output.  ADDS the gradient of its query input to dq_out [>= B
        rows, d] and the gradient of its key / value rows to dkv_out [>= BK rows, d] (views of the IM2TEXT-output
        gradient)."""
        lib, heads = load(), xf.heads
        for l in reversed(range(len(xf.cross_layers))):
            L = lambda nm: self.lin[f"{tag}.cross_layers.{l}.{nm}"]
            q_in, Q, Kp, Vp, att = saved[l]
            dy = _cast(dq[:B])
            grads[L("to_out.0").name + ".weight"] = _grad_weight(dy, att, B)
            grads[L("to_out.0").name + ".bias"] = _colsum(dq, B, dq.shape[1])
            datt = _gemm(dy, L("to_out.0").wT, None, _lib.EPI_BIAS_BF16, B)
            dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(Kp), torch.zeros_like(Vp)
            check(lib.keds_cross_core_bwd(ptr(Q), ptr(Kp), ptr(Vp), ptr(datt), ptr(dQ), ptr(dK), ptr(dV), B, K, heads, stream()),
                  "keds_cross_core_bwd")
            for nm, g, x, M in (("to_q", dQ, q_in, B), ("to_k", dK, kv_bf, BK), ("to_v", dV, kv_bf, BK)):
                grads[L(nm).name + ".weight"] = _grad_weight(g, x, M)
                grads[L(nm).name + ".bias"] = _colsum(g, M, g.shape[1])
            # the gradient of the key / value rows accumulates over the layers in the GEMM epilogue (out += acc); the
            # first layer's query gradient lands on the IM2TEXT rows of the query the same way
            _gemm(dK, L("to_k").wT, None, _lib.EPI_BIAS_RESID_F32, BK, out=dkv_out)
            _gemm(dV, L("to_v").wT, None, _lib.EPI_BIAS_RESID_F32, BK, out=dkv_out)
            if l > 0:
                dq = _gemm(dQ, L("to_q").wT, None, _lib.EPI_BIAS_F32, B)
            else:
                _gemm(dQ, L("to_q").wT, None, _lib.EPI_BIAS_RESID_F32, B, out=dq_out)

    # ---- the frozen text tower with saved activations (model.py:305-326, 808-851) and its backward to the tokens --------
    def _text_forward(self, tokens, img_tokens, ins, readout_rows):
        fz, lib = self._tower(), load()
        m = self.model
        B, Lc = tokens.shape
        w, heads = m.transformer.width, m.transformer.heads
        M = B * Lc
        x = ops.embed_tokens(tokens.to(torch.int32), fz["tok"], fz["pos"], img_tokens, ins).reshape(M, w)
        xin = torch.zeros((_pad(M), w), dtype=F32, device=x.device)
        xin[:M] = x
        saved = []
        for t in fz["blocks"]:
            st1 = torch.empty((M, 2), dtype=F32, device=x.device)
            ln1 = torch.zeros((_pad(M), w), dtype=BF16, device=x.device)
            check(lib.keds_ln_fwd_stats(ptr(xin), w, None, ptr(t["ln1_g"]), ptr(t["ln1_b"]), ptr(ln1), ptr(st1), M, w, stream()), "ln")
            qkv = _gemm(ln1, t["qkv_w"], t["qkv_b"], _lib.EPI_BIAS_BF16, M)
            att = torch.zeros((_pad(M), w), dtype=BF16, device=x.device)
            check(lib.keds_attention(ptr(qkv), ptr(att), B, Lc, heads, 1, stream()), "keds_attention")
            xmid = xin.clone()
            _gemm(att, t["out_w"], t["out_b"], _lib.EPI_BIAS_RESID_F32, M, out=xmid)
            st2 = torch.empty((M, 2), dtype=F32, device=x.device)
            ln2 = torch.zeros((_pad(M), w), dtype=BF16, device=x.device)
            check(lib.keds_ln_fwd_stats(ptr(xmid), w, None, ptr(t["ln2_g"]), ptr(t["ln2_b"]), ptr(ln2), ptr(st2), M, w, stream()), "ln")
            u = _gemm(ln2, t["fc_w"], t["fc_b"], _lib.EPI_BIAS_BF16, M)
            hid = torch.zeros_like(u)
            check(lib.keds_qgelu_fwd(ptr(u), ptr(hid), u.numel(), stream()), "keds_qgelu_fwd")
            xout = xmid.clone()
            _gemm(hid, t["proj_w"], t["proj_b"], _lib.EPI_BIAS_RESID_F32, M, out=xout)
            saved.append((xin, st1, qkv, xmid, st2, u))
            xin = xout
        stf = torch.empty((B, 2), dtype=F32, device=x.device)
        lnf = torch.zeros((_pad(B), w), dtype=BF16, device=x.device)
        check(lib.keds_ln_fwd_stats(ptr(xin), w, ptr(readout_rows), ptr(fz["lnf_g"]), ptr(fz["lnf_b"]), ptr(lnf), ptr(stf), B, w,
                                    stream()), "ln_final")
        feat = _gemm(lnf, fz["proj_t"], None, _lib.EPI_BIAS_F32, B)
        return feat, (saved, xin, stf, M, B, Lc)

    def _text_backward(self, dfeat, ctx, readout_rows, token_rows):
        fz, lib = self._tower(), load()
        saved, xlast, stf, M, B, Lc = ctx
        w, heads = self.model.transformer.width, self.model.transformer.heads
        dev = dfeat.device
        dlnf = _gemm(_cast(dfeat[:B]), fz["proj"], None, _lib.EPI_BIAS_F32, B)              # [Bp, w]
        dx = torch.zeros((_pad(M), w), dtype=F32, device=dev)
        check(lib.keds_ln_bwd(ptr(dlnf), ptr(xlast), w, ptr(readout_rows), ptr(stf), ptr(fz["lnf_g"]), ptr(dx), None, B, w, stream()),
              "keds_ln_bwd")
        dx_bf = ops.cast_bf16(dx)
        for t, (xin, st1, qkv, xmid, st2, u) in zip(reversed(fz["blocks"]), reversed(saved)):
            dhid = _gemm(dx_bf, t["proj_wT"], None, _lib.EPI_BIAS_BF16, M)
            du = torch.zeros_like(dhid)
            check(lib.keds_qgelu_bwd(ptr(dhid), ptr(u), ptr(du), du.numel(), stream()), "keds_qgelu_bwd")
            dln2 = _gemm(du, t["fc_wT"], None, _lib.EPI_BIAS_F32, M)
            check(lib.keds_ln_bwd(ptr(dln2), ptr(xmid), w, None, ptr(st2), ptr(t["ln2_g"]), ptr(dx), ptr(dx_bf), M, w, stream()),
                  "keds_ln_bwd")
            datt = _gemm(dx_bf, t["out_wT"], None, _lib.EPI_BIAS_BF16, M)
            dqkv = torch.zeros_like(qkv)
            check(lib.keds_attention_bwd(ptr(qkv), ptr(datt), ptr(dqkv), B, Lc, heads, 1, stream()), "keds_attention_bwd")
            dln1 = _gemm(dqkv, t["qkv_wT"], None, _lib.EPI_BIAS_F32, M)
            check(lib.keds_ln_bwd(ptr(dln1), ptr(xin), w, None, ptr(st1), ptr(t["ln1_g"]), ptr(dx), ptr(dx_bf), M, w, stream()),
                  "keds_ln_bwd")
        n_tok = token_rows.numel() // B
        dtok = torch.empty((B * n_tok, w), dtype=F32, device=dev)
        check(lib.keds_rows_gather(ptr(dx), w, ptr(token_rows), ptr(dtok), B * n_tok, w, stream()), "keds_rows_gather")
        return dtok.reshape(B, n_tok, w)

    # ---- loss and gradients ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def loss_and_grads(self, image_features: torch.Tensor, nbr_img: torch.Tensor, nbr_txt: torch.Tensor,
                       prompt_tokens: torch.Tensor, id_split: int, masks: Optional[List[torch.Tensor]] = None):
        """One forward + backward.  image_features [B, D] fp32, nbr_img / nbr_txt [B, K, D] the retrieved neighbour rows,
        prompt_tokens [77] or [B, 77] with one `id_split` token.  masks (optional): one uint8 keep-mask [B(1+2K), middle]
        per IM2TEXT hidden layer (parity tests pass torch's); by default drawn on the device from (seed, step).
        Returns (loss device scalar, {parameter name: fp32 gradient})."""
        lib, m = load(), self.model
        dev = image_features.device
        for L in self.lin.values():
            L.refresh()
        B, D = image_features.shape
        K = nbr_img.shape[1]
        BK, R = B * K, B * (1 + 2 * K)
        feats = image_features.float().contiguous()
        rows = torch.zeros((_pad(R), D), dtype=F32, device=dev)
        rows[:B], rows[B:B + BK], rows[B + BK:R] = feats, nbr_img.reshape(BK, D), nbr_txt.reshape(BK, D)
        rows_bf = ops.cast_bf16(rows)
        # IM2TEXT (training mode: Linear -> Dropout -> ReLU, model.py:112-116)
        scale = 1.0 / (1.0 - self.p_drop) if self.p_drop > 0 else 1.0
        cur, i2t_saved = rows_bf, []
        for i in range(len(self.i2t.layers)):
            L = self.lin[f"i2t.layers.{i}.0"]
            z = _gemm(cur, L.w, L.b, _lib.EPI_BIAS_BF16, R)
            mask = None
            if self.p_drop > 0:
                if masks is not None:
                    mask = torch.zeros((z.shape[0], z.shape[1]), dtype=torch.uint8, device=dev)
                    mask[:R] = masks[i].to(dev, dtype=torch.uint8)
                else:
                    mask = torch.empty(z.shape, dtype=torch.uint8, device=dev)
                    check(lib.keds_dropout_mask(ptr(mask), mask.numel(), (self.seed << 32) + self.steps * 16 + i, self.p_drop, stream()),
                          "keds_dropout_mask")
            y = torch.zeros_like(z)
            check(lib.keds_dropout_relu_fwd(ptr(z), ptr(mask), scale, ptr(y), z.numel(), stream()), "keds_dropout_relu_fwd")
            i2t_saved.append((cur, z, mask))
            cur = y
        Lo = self.lin["i2t.fc_out"]
        mapped = _gemm(cur, Lo.w, Lo.b, _lib.EPI_BIAS_F32, R)                          # [Rp, d] fp32
        y_last = cur
        d = mapped.shape[1]
        map_bf = ops.cast_bf16(mapped)
        q_bf = _cast(mapped[:B])
        kv = {"fuse": _cast(mapped[B:B + BK]), "cond": _cast(mapped[B + BK:R])}
        outs, xf_saved = {}, {}
        for tag, xf in (("fuse", self.fuse), ("cond", self.cond)):
            outs[tag], xf_saved[tag] = self._xf_forward(tag, xf, q_bf, kv[tag], B, BK, K)
        tokens = torch.stack([outs["fuse"][:B], outs["cond"][:B], mapped[:B]], dim=1).contiguous()       # [B, 3, d]
        # frozen text tower on "a photo of *" with the three tokens at the * (model.py:808-851)
        text = prompt_tokens.to(dev)
        if text.dim() == 1:
            text = text[None, :].repeat(B, 1)
        where = (text[0] == int(id_split)).nonzero()
        if where.numel() == 0:
            raise IndexError("split token not present in the prompt")
        ins = int(where[0])
        Lc = text.shape[1]
        eot = (text == m.end_id).to(torch.int32).argmax(dim=1)
        if int(eot.max()) + 2 >= Lc:
            raise IndexError("read-out row beyond the context length")
        ar = torch.arange(B, device=dev, dtype=torch.int32)
        readout_rows = (ar * Lc + eot.to(torch.int32) + 2).contiguous()
        token_rows = (ar[:, None] * Lc + ins + torch.arange(3, device=dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        feat, ctx = self._text_forward(text, tokens, ins, readout_rows)
        txt_n = ops.l2_normalize(feat[:B].contiguous())
        img_n = ops.l2_normalize(feats)
        all_img, all_txt = self._gather_negatives(img_n, txt_n)
        N = all_img.shape[0]
        ws = torch.empty(int(lib.keds_clip_loss_workspace_bytes(N)), dtype=torch.uint8, device=dev)
        loss = torch.zeros(1, dtype=F32, device=dev)
        dtxt_n = torch.empty((B, txt_n.shape[1]), dtype=F32, device=dev)
        logit_scale = float(m.logit_scale.detach().exp())
        check(lib.keds_clip_loss(ptr(all_img), ptr(all_txt), N, B, all_img.shape[1], logit_scale, ptr(loss), ptr(dtxt_n), ptr(ws),
                                 ws.numel(), stream()), "keds_clip_loss")
        dfeat = torch.empty_like(dtxt_n)
        check(lib.keds_l2norm_bwd(ptr(feat), ptr(dtxt_n), ptr(dfeat), B, feat.shape[1], stream()), "keds_l2norm_bwd")
        dtok = self._text_backward(dfeat, ctx, readout_rows, token_rows)                # [B, 3, d]
        # ---- backward of the three modules
        grads: Dict[str, torch.Tensor] = {}
        dmap = torch.zeros((_pad(R), d), dtype=F32, device=dev)                        # gradient of the IM2TEXT output rows
        dmap[:B] = dtok[:, 2]
        for slot, (tag, xf) in enumerate((("fuse", self.fuse), ("cond", self.cond))):
            dq = torch.zeros((_pad(B), d), dtype=F32, device=dev)
            dq[:B] = dtok[:, slot]
            lo = B + slot * BK
            self._xf_backward(tag, xf, xf_saved[tag], dq, kv[tag], B, BK, K, grads, dmap, dmap[lo:])
        dy = ops.cast_bf16(dmap)
        grads["i2t.fc_out.weight"] = _grad_weight(dy, y_last, R)
        grads["i2t.fc_out.bias"] = _colsum(dmap, R, d)
        dcur = _gemm(dy, Lo.wT, None, _lib.EPI_BIAS_BF16, R)
        for i in reversed(range(len(self.i2t.layers))):
            L = self.lin[f"i2t.layers.{i}.0"]
            x_in, z, mask = i2t_saved[i]
            dz = torch.zeros_like(z)
            check(lib.keds_dropout_relu_bwd(ptr(dcur), 0, ptr(z), ptr(mask), scale, ptr(dz), z.numel(), stream()), "keds_dropout_relu_bwd")
            grads[L.name + ".weight"] = _grad_weight(dz, x_in, R)
            grads[L.name + ".bias"] = _colsum(dz, R, dz.shape[1])
            if i > 0:
                dcur = _gemm(dz, L.wT, None, _lib.EPI_BIAS_BF16, R)
        return loss, grads

    def _gather_negatives(self, img_n, txt_n):
        """trainer.py:78-99: with torch.distributed, the features of every rank, this rank's first."""
        import torch.distributed as dist
        if not (self.aggregate and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return img_n.contiguous(), txt_n.contiguous()
        return gather_own_first(img_n, self.group), gather_own_first(txt_n, self.group)

    # ---- one optimizer step (main.py:215-237: AdamW; biases without weight decay) -------------------------------------------
    @torch.no_grad()
    def step(self, image_features, database, prompt_tokens, id_split: int, masks=None):
        from .retrieval import get_retrieved_features
        nbr_img, nbr_txt = get_retrieved_features(image_features.float(), database, None, topk=self.topk)
        loss, grads = self.loss_and_grads(image_features, nbr_img, nbr_txt, prompt_tokens, id_split, masks)
        self.apply_gradients(grads)
        return loss

    @torch.no_grad()
    def apply_gradients(self, grads: Dict[str, torch.Tensor]):
        import torch.distributed as dist
        lib = load()
        names = sorted(grads)
        gscale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            flat = torch.cat([grads[n].reshape(-1) for n in names])             # one bucket: 5.8 M floats, one collective
            dist.all_reduce(flat, group=self.group)                             # SUM; the mean is taken inside the AdamW kernel
            gscale = 1.0 / dist.get_world_size(self.group)
            off = 0
            for n in names:
                k = grads[n].numel()
                grads[n] = flat[off:off + k].reshape(grads[n].shape)
                off += k
        self.steps += 1
        for n in names:
            lin = self.lin[n.rsplit(".", 1)[0]].lin
            p = lin.weight if n.endswith(".weight") else lin.bias
            g = grads[n].contiguous()
            if p.dtype != F32 or not p.is_contiguous():
                raise RuntimeError("trainable parameters must be contiguous fp32 tensors")
            if n not in self._opt:
                self._opt[n] = torch.zeros((2,) + tuple(p.shape), dtype=F32, device=p.device)
            mv = self._opt[n]
            wd = 0.0 if n.endswith(".bias") else self.wd                  # main.py:215-222: no decay on gains / biases
            check(lib.keds_adamw_step(ptr(p.data), ptr(g), ptr(mv[0]), ptr(mv[1]), p.numel(), self.lr, self.b1, self.b2, self.eps, wd,
                                      self.steps, gscale, stream()), "keds_adamw_step")
        for mod in (self.i2t, self.fuse, self.cond):                      # the inference packs are stale now
            mod._packed = None
            if hasattr(mod, "_packed32"):
                mod._packed32 = None


def gather_own_first(x: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of [B, d] rows with THIS rank's rows first and the others in rank order (trainer.py:78-99); one
    all_gather_into_tensor.  Works on CPU tensors too (gloo tests of the ordering)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    parts = out.reshape((world,) + tuple(x.shape))
    order = [rank] + [r for r in range(world) if r != rank]
    return parts[order].reshape(out.shape).contiguous()
