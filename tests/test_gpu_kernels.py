"""GPU parity of the encoder primitives, one kernel at a time, against plain fp32 torch on the CPU.

Tolerances (stated per test): the kernels take bf16 operands and accumulate in fp32, so each test
feeds the fp32 reference the SAME bf16-rounded operands; what remains is accumulation order and the
bf16 rounding of outputs (<= 2^-8 relative), i.e. rel-L2 <= 5e-3 for bf16 outputs and <= 1e-5 for fp32
outputs.
"""
import math

import numpy as np
import pytest
import torch

import keds_amd
from keds_amd import _lib, ops
from oracle import keds_oracle as O
from tests.gpu_util import bf16_round, max_abs, rel_l2, report

pytestmark = pytest.mark.gpu


def _rand(shape, seed, std=1.0):
    return O.synth_tensor(f"t{seed}", list(shape), std, seed)


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (128, 128, 64), (1000, 384, 128), (257 * 3, 1024, 1024),
                                   (77 * 5, 768, 3072)])
@pytest.mark.parametrize("epi", ["bias_bf16", "qgelu", "relu", "resid", "bias_f32", "nobias"])
def test_gemm_epilogues(M, N, K, epi):
    a = bf16_round(_rand((M, K), 1))
    w = bf16_round(_rand((N, K), 2, std=K ** -0.5))
    bias = _rand((N,), 3, std=0.1)
    ref = a @ w.t() + (0 if epi == "nobias" else bias)
    a_d = ops.cast_bf16(a.cuda(), rows_padded=(M + 127) // 128 * 128)
    w_d = w.cuda().to(torch.bfloat16)
    b_d = None if epi == "nobias" else bias.cuda()
    if epi in ("bias_bf16", "nobias"):
        out = ops.gemm_bt(a_d, w_d, b_d, _lib.EPI_BIAS_BF16, m=M)[:M]
        tol = 5e-3
    elif epi == "qgelu":
        ref = O.quick_gelu(ref)
        out = ops.gemm_bt(a_d, w_d, b_d, _lib.EPI_BIAS_QGELU_BF16, m=M)[:M]
        tol = 5e-3
    elif epi == "relu":
        ref = torch.relu(ref)
        out = ops.gemm_bt(a_d, w_d, b_d, _lib.EPI_BIAS_RELU_BF16, m=M)[:M]
        tol = 5e-3
    elif epi == "resid":
        x0 = _rand((M, N), 4)
        ref = x0 + ref
        xd = torch.zeros(((M + 127) // 128 * 128, N), device="cuda")
        xd[:M] = x0.cuda()
        out = ops.gemm_bt(a_d, w_d, b_d, _lib.EPI_BIAS_RESID_F32, out=xd, m=M)[:M]
        tol = 1e-5
    else:
        out = ops.gemm_bt(a_d, w_d, b_d, _lib.EPI_BIAS_F32, m=M)[:M]
        tol = 1e-5
    err = rel_l2(out, ref)
    report("gemm", M=M, N=N, K=K, epi=epi, rel_l2=err, max_abs=max_abs(out, ref))
    assert err <= tol


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed or permuted C-write exactly (integers in bf16)."""
    N = K = 128
    a = torch.eye(128)
    w = ((torch.arange(N)[:, None] + 3 * torch.arange(K)[None, :]) % 251).float()  # asymmetric, exact in bf16
    out = ops.gemm_bt(ops.cast_bf16(a.cuda(), 128), w.cuda().to(torch.bfloat16), None, _lib.EPI_BIAS_F32, m=128)
    assert torch.equal(out.cpu(), w.t().contiguous())


def test_gemm_strided_rows_and_query_limited_attention():
    """keds_gemm_bt_ex (row strides; A rows past M never read) and keds_attention_ex (first q rows only): the pieces of
    the CLS-row-only last ViT layer."""
    from keds_amd._lib import check, load, ptr, stream
    lib = load()
    B, S, w = 5, 17, 128
    att = bf16_round(_rand((B * S, w), 61))
    W = bf16_round(_rand((w, w), 62, std=w ** -0.5))
    bias = _rand((w,), 63, 0.1)
    x0 = _rand((B * S, w), 64)
    a_d = att.cuda().to(torch.bfloat16)                     # exactly B*S rows: no padding available
    x_d = x0.clone().cuda()
    _lib.ensure_gemm_workspace(a_d.device)
    check(lib.keds_gemm_bt_ex(ptr(a_d), S * w, ptr(W.cuda().to(torch.bfloat16)), ptr(bias.cuda()), ptr(x_d), S * w, B, w, w,
                              _lib.EPI_BIAS_RESID_F32, None, 0, stream()), "keds_gemm_bt_ex")
    ref = x0.clone()
    ref[::S] += att[::S] @ W.t() + bias
    assert rel_l2(x_d, ref) <= 1e-5                         # CLS rows updated, every other row untouched
    assert torch.equal(x_d.cpu().reshape(B, S, w)[:, 1:], x0.reshape(B, S, w)[:, 1:])
    heads = 2
    qkv = bf16_round(_rand((B * S, 3 * heads * 64), 65, std=1.5))
    full = ops.attention(qkv.cuda().to(torch.bfloat16), B, S, heads, False)
    out = torch.full((B * S, heads * 64), 7.0, dtype=torch.bfloat16, device="cuda")
    check(lib.keds_attention_ex(ptr(qkv.cuda().to(torch.bfloat16)), ptr(out), B, S, heads, 0, 1, stream()), "keds_attention_ex")
    o = out.float().cpu().reshape(B, S, -1)
    assert torch.equal(o[:, 0], full.float().cpu().reshape(B, S, -1)[:, 0])      # query row 0 identical
    assert bool((o[:, 1:] == 7.0).all())                                           # other rows not written


def test_gemm_patch_epilogue():
    B, G, N, K = 3, 16, 128, 640
    a = bf16_round(_rand((B * G, K), 5))
    w = bf16_round(_rand((N, K), 6, std=K ** -0.5))
    pos = _rand((G + 1, N), 7)
    out = torch.zeros((B * (G + 1), N), device="cuda")
    ops.gemm_bt(ops.cast_bf16(a.cuda(), 128), w.cuda().to(torch.bfloat16), None, _lib.EPI_PATCH_F32, out=out, m=B * G,
                aux=pos.cuda(), aux_i=G)
    ref = torch.zeros(B, G + 1, N)
    ref[:, 1:] = (a @ w.t()).reshape(B, G, N) + pos[1:]
    err = rel_l2(out.reshape(B, G + 1, N)[:, 1:], ref[:, 1:])
    assert err <= 1e-5 and float(out.reshape(B, G + 1, N)[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("rows,dim", [(5, 128), (300, 768), (257 * 2, 1024), (7, 2048), (64, 256)])
def test_layernorm(rows, dim):
    x = _rand((rows, dim), 11) * 3 + 0.5
    g, b = 1 + _rand((dim,), 12, 0.1), _rand((dim,), 13, 0.1)
    ref = O.layer_norm(x, g, b)
    out32 = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), out_f32=True)
    out16 = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), out_f32=False)
    report("layernorm", rows=rows, dim=dim, rel_l2_f32=rel_l2(out32, ref), rel_l2_bf16=rel_l2(out16, ref))
    assert rel_l2(out32, ref) <= 1e-5         # fp32 statistics, fp32 out
    assert rel_l2(out16, ref) <= 5e-3         # bf16 rounding of the output only


def _attn_ref(qkv, B, S, heads, causal):
    d = heads * 64
    q, k, v = qkv.reshape(B, S, 3 * d).split(d, dim=-1)
    q = q.reshape(B, S, heads, 64).transpose(1, 2)
    k = k.reshape(B, S, heads, 64).transpose(1, 2)
    v = v.reshape(B, S, heads, 64).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s.masked_fill(torch.ones(S, S, dtype=torch.bool).triu(1), float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, d)


@pytest.mark.parametrize("B,S,heads,causal", [(2, 257, 16, False), (3, 77, 12, True), (2, 17, 2, False),
                                              (1, 5, 2, True), (2, 96, 2, True), (1, 288, 2, False), (2, 33, 2, False)])
def test_attention(B, S, heads, causal):
    """P is rounded to bf16 before PV (flash-attention practice): tolerance rel-L2 1e-2."""
    qkv = bf16_round(_rand((B * S, 3 * heads * 64), 21, std=1.5))
    ref = _attn_ref(qkv, B, S, heads, causal)
    out = ops.attention(qkv.cuda().to(torch.bfloat16), B, S, heads, causal)
    err = rel_l2(out, ref)
    report("attention", B=B, S=S, heads=heads, causal=causal, rel_l2=err, max_abs=max_abs(out, ref))
    assert err <= 1e-2


def test_attention_257_tail_kernel_against_generic_and_reference():
    """S = 257 runs as 16 MFMA key tiles + a rank-1 update for the last key, and the last query as its own VALU row:
    same tolerance against the fp32 reference as the generic (padded) kernel, close to it, and the q_limit (CLS-only) form
    of the last tower block agrees on the rows it computes."""
    lib = _lib.load()
    B, S, heads = 3, 257, 4
    qkv = bf16_round(_rand((B * S, 3 * heads * 64), 23, std=1.5))
    ref = _attn_ref(qkv, B, S, heads, False)
    dev = qkv.cuda().to(torch.bfloat16)
    tail = ops.attention(dev, B, S, heads, False).float().cpu()
    try:
        lib.keds_attention_debug(16)
        gen = ops.attention(dev, B, S, heads, False).float().cpu()
    finally:
        lib.keds_attention_debug(0)
    last = torch.arange(B) * S + 256
    report("attention.tail257", rel_l2=rel_l2(tail, ref), generic_rel_l2=rel_l2(gen, ref), vs_generic=rel_l2(tail, gen),
           last_query_rel_l2=rel_l2(tail[last], ref[last]), first_query_rel_l2=rel_l2(tail[last - 256], ref[last - 256]))
    assert rel_l2(tail, ref) <= 1e-2 and rel_l2(tail[last], ref[last]) <= 1e-2
    assert rel_l2(tail, gen) <= 1e-2
    out1 = torch.zeros(B * S, heads * 64, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.keds_attention_ex(_lib.ptr(dev), _lib.ptr(out1), B, S, heads, 0, 1, _lib.stream()), "attention q_limit")
    first = torch.arange(B) * S
    assert torch.equal(out1[first.cuda()].float().cpu(), tail[first])


def test_attention_257_eight_wave_kernel():
    """The product kernel for S = 257 (eight waves per (batch, head), 32-key tiles, probabilities relative to the first
    tile's row maximum): against the fp32 reference no worse than the 4-wave kernels, q_limit forms (CLS only, a partial
    block, all MFMA blocks without the last query; the dominant key's probability is 2^x rounded to bf16 here where a kernel
    that subtracts the exact row maximum has an exact 1.0, hence the factor 1.3), and the recompute path -- rows whose maximum lies far above the first
    tile's (here: > 64 in log2 units) are recomputed with the true maximum and still meet the tolerance."""
    lib = _lib.load()
    B, S, heads = 3, 257, 4
    qkv = bf16_round(_rand((B * S, 3 * heads * 64), 29, std=1.5))
    ref = _attn_ref(qkv, B, S, heads, False)
    dev = qkv.cuda().to(torch.bfloat16)
    new = ops.attention(dev, B, S, heads, False).float().cpu()
    errs = {}
    try:
        for code, name in ((32, "tail4"), (16, "generic")):
            lib.keds_attention_debug(code)
            errs[name] = rel_l2(ops.attention(dev, B, S, heads, False).float().cpu(), ref)
    finally:
        lib.keds_attention_debug(0)
    last = torch.arange(B) * S + 256
    report("attention.s257_8wave", rel_l2=rel_l2(new, ref), last_query_rel_l2=rel_l2(new[last], ref[last]), **errs)
    assert rel_l2(new, ref) <= 1.3 * max(errs.values()) and rel_l2(new[last], ref[last]) <= 1e-2
    for ql in (1, 100, 256):
        out = torch.full((B * S, heads * 64), 7.0, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.keds_attention_ex(_lib.ptr(dev), _lib.ptr(out), B, S, heads, 0, ql, _lib.stream()), "attention q_limit")
        out = out.float().cpu().reshape(B, S, -1)
        assert torch.equal(out[:, :ql], new.reshape(B, S, -1)[:, :ql])
        assert bool((out[:, ql:] == 7.0).all())                # rows past q_limit are not written
    # recompute path: keys 32.. carry 40x larger K rows for head 1 of sample 0 -> row maxima ~500 raw score units above tile 0's
    big = qkv.clone().reshape(B, S, 3, heads, 64)
    big[0, 32:, 1, 1] *= 40.0
    big[1, 256, 1, 2] *= 60.0                                  # only the LAST key is far above (rank-1 path decides)
    big = bf16_round(big.reshape(B * S, -1))
    ref2 = _attn_ref(big, B, S, heads, False)
    out2 = ops.attention(big.cuda().to(torch.bfloat16), B, S, heads, False).float().cpu()
    assert torch.isfinite(out2).all()
    report("attention.s257_8wave.recompute", rel_l2=rel_l2(out2, ref2))
    assert rel_l2(out2, ref2) <= 1e-2


def test_attention_peaked_rows():
    """One dominant key per query (softmax ~ one-hot) and large logits: exercises the max subtraction."""
    B, S, heads = 1, 257, 2
    qkv = bf16_round(_rand((B * S, 3 * heads * 64), 22, std=6.0))
    ref = _attn_ref(qkv, B, S, heads, False)
    out = ops.attention(qkv.cuda().to(torch.bfloat16), B, S, heads, False)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) <= 2e-2


def test_im2col_matches_conv():
    B, R, P, width = 2, 56, 14, 128
    img = _rand((B, 3, R, R), 31)
    w = _rand((width, 3, P, P), 32, std=0.05)
    kpad = 640
    col = ops.im2col(img.cuda(), P, kpad)[: B * 16].float().cpu()
    wf = torch.zeros(width, kpad)
    wf[:, : 3 * P * P] = w.reshape(width, -1)
    ref = torch.nn.functional.conv2d(bf16_round(img), w, stride=P).reshape(B, width, -1).permute(0, 2, 1).reshape(B * 16, width)
    assert rel_l2(col @ wf.t(), ref) <= 1e-5
    assert float(col[:, 3 * P * P:].abs().max()) == 0.0


def test_embed_tokens_splice():
    B, L, d, vocab = 3, 77, 128, 512
    table, pos = _rand((vocab, d), 41, 0.02), _rand((L, d), 42, 0.01)
    rs = np.random.RandomState(0)
    tokens = torch.from_numpy(rs.randint(0, vocab, size=(B, L)).astype(np.int64))
    x = ops.embed_tokens(tokens.cuda(), table.cuda(), pos.cuda()).cpu()
    assert torch.equal(x, table[tokens] + pos)
    for n_tok in (2, 3):
        it = _rand((B, n_tok, d), 43)
        ins = 4
        emb = table[tokens]
        ref = torch.cat([emb[:, :ins], it, emb[:, ins + 1: L - (n_tok - 1)]], dim=1) + pos     # model.py:832-837
        x = ops.embed_tokens(tokens.cuda(), table.cuda(), pos.cuda(), it.cuda(), ins).cpu()
        assert torch.equal(x, ref)


def test_normalise_and_mixture():
    a, b = _rand((9, 768), 51), _rand((9, 768), 52)
    an, bn, mix = ops.mix_normalize(a.cuda(), b.cuda())
    ra, rb = O.l2_normalize(a), O.l2_normalize(b)
    assert rel_l2(an, ra) <= 1e-6 and rel_l2(bn, rb) <= 1e-6
    assert rel_l2(mix, O.l2_normalize(0.5 * ra + 0.5 * rb)) <= 1e-6            # eval_utils.py:704-710
    assert rel_l2(ops.l2_normalize(a.cuda()), ra) <= 1e-6


def test_gallery_ranking_and_recall_metric():
    import keds_amd
    g = dict(np.load(__import__("tests.conftest", fromlist=["golden_path"]).golden_path("metrics_cirr.npz")))
    G = g["gallery"].shape[0]
    index_names = [f"/data/cirr/dev/img_{i:05d}.png" for i in range(G)]
    ref_names = [f"img_{i:05d}.png" for i in g["ref_idx"]]
    tgt_names = [f"img_{i:05d}.png" for i in g["tgt_idx"]]
    gal, ref = torch.from_numpy(g["gallery"]).cuda(), torch.from_numpy(g["ref"]).cuda()
    order = ops.rank_gallery(ref, gal).cpu().long()
    dist = 1 - torch.from_numpy(g["ref"]).double() @ torch.from_numpy(g["gallery"]).double().t()
    assert torch.equal(torch.sort(order, dim=1).values, torch.arange(G).expand(order.shape[0], G))   # a permutation
    along = torch.gather(dist, 1, order)
    assert float((along[:, :-1] - along[:, 1:]).max()) <= 1e-6          # ascending up to fp32 evaluation error
    m = keds_amd.get_metrics_cirr(gal, ref, ref_names, index_names, tgt_names)
    for k in (1, 5, 10, 50, 100):
        assert abs(m[f"recall_R@{k}"] - float(g[f"recall_R_at_{k}"])) < 1e-4     # reference's own numbers
    with pytest.raises(AssertionError):
        keds_amd.get_metrics_cirr(gal, ref, ref_names, index_names, ["nope.png"] * len(tgt_names))
    # a larger ragged case against the oracle
    rs = np.random.RandomState(3)
    G2, Q2 = 2297, 300
    gal2 = O.l2_normalize(torch.from_numpy(rs.standard_normal((G2, 768)).astype(np.float32)))
    ri, ti = rs.randint(0, G2, Q2), None
    ti = (ri + 1 + rs.randint(0, G2 - 1, Q2)) % G2
    ref2 = O.l2_normalize(gal2[ti] + 0.05 * torch.from_numpy(rs.standard_normal((Q2, 768)).astype(np.float32)))
    names = [f"d/{i}.png" for i in range(G2)]
    mo = O.get_metrics_cirr(gal2, ref2, [f"{i}.png" for i in ri], names, [f"{i}.png" for i in ti])
    mg = keds_amd.get_metrics_cirr(gal2.cuda(), ref2.cuda(), [f"{i}.png" for i in ri], names, [f"{i}.png" for i in ti])
    assert mo == mg


# ---- LayerNorm folded into the GEMMs (KEDS_EPI_LN_* / KEDS_EPI_RESID_STATS_F32) ------------------------------
def _fold(w, b, gamma, beta, dt=torch.bfloat16):
    lib = _lib.load()
    n, k = w.shape
    wf = torch.empty((n, k), dtype=dt, device="cuda")
    bc = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    if dt == torch.bfloat16:
        _lib.check(lib.keds_fold_layernorm(_lib.ptr(w), _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), n, k, _lib.ptr(wf),
                                           _lib.ptr(bc), _lib.stream()), "fold")
    else:
        _lib.check(lib.keds_fold_layernorm_ex(_lib.ptr(w), _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), n, k, _lib.ptr(wf), 1,
                                              _lib.ptr(bc), _lib.stream()), "fold f16")
    # the fp16 build rounds the exact product once (v_fma_mixlo_f16) where torch rounds to fp32 first: a 1-ulp difference
    # on the few elements whose fp32 product is a tie
    ref = (w * gamma).to(dt)
    differ = wf != ref
    assert float(differ.float().mean()) <= 1e-3
    assert float((wf.float() - ref.float()).abs().max()) <= float(ref.abs().max()) * 2.0 ** -10
    return wf, bc


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,epi", [(700, 384, 256, "ln"), (33024 + 128, 3072, 1024, "ln"), (1500, 1024, 256, "gelu"),
                                       (4224, 4096, 1024, "gelu")])
def test_gemm_layernorm_epilogues_equal_layernorm_then_linear(M, N, K, epi, dt):
    """rstd (x W'^T - mean colsum W') + (b + W beta) == Linear(LayerNorm(x)): against torch fp32 on the GPU's own
    rounded operands, and the second statistics buffer is cleared for exactly the rows of the launch.  bf16 operands
    (KEDS_EPI_LN_*) and fp16 operands (KEDS_EPI_LN_*_H: what the towers run on their fp16 residual stream)."""
    lib = _lib.load()
    f16 = dt == torch.float16
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, generator=g, device="cuda") * 1.7 + 0.3 * torch.randn(M, 1, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    gamma = 1 + 0.2 * torch.randn(K, generator=g, device="cuda")
    beta = 0.1 * torch.randn(K, generator=g, device="cuda")
    wf, bc = _fold(w, b, gamma, beta, dt)
    Mp = (M + 127) // 128 * 128
    xb = torch.zeros((Mp, K), dtype=dt, device="cuda")
    stats = torch.zeros((Mp, 2), dtype=torch.int64, device="cuda")      # {sum, sum sq} as 64-bit fixed point (* 2^28)
    _lib.check(lib.keds_rowstats_cast_ex(_lib.ptr(x), _lib.ptr(xb), int(f16), _lib.ptr(stats), M, K, _lib.stream()), "rowstats")
    assert torch.equal(xb[:M], x.to(dt))
    sf = stats.double() / 2.0 ** 28
    assert torch.allclose(sf[:M, 0].float(), x.sum(1), rtol=1e-5, atol=1e-3) and torch.allclose(sf[:M, 1].float(), (x * x).sum(1), rtol=1e-5)
    other = torch.full((Mp + 8, 2), 7, dtype=torch.int64, device="cuda")
    out = torch.zeros((Mp, N), dtype=torch.bfloat16, device="cuda")
    code = _lib.EPI_LN_BIAS_BF16 if epi == "ln" else _lib.EPI_LN_QGELU_BF16
    if f16:
        code = _lib.EPI_LN_BIAS_BF16_H if epi == "ln" else _lib.EPI_LN_QGELU_BF16_H
    _lib.ensure_gemm_workspace("cuda")
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(xb), K, _lib.ptr(wf), _lib.ptr(bc), _lib.ptr(out), N, M, N, K, code,
                                    _lib.ptr(stats), 0, _lib.ptr(other), _lib.stream()), "gemm ln")
    assert bool((other[:M] == 0).all()) and bool((other[M:] == 7).all())
    # reference on the rounded operands the kernel saw
    xr = xb[:M].float()
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    want = rstd * (xr @ wf.float().t() - mean * wf.float().sum(1)[None, :]) + (b + w @ beta)[None, :]
    if epi == "gelu":
        want = want * torch.sigmoid(1.702 * want)
    report(f"gemm_ln_{epi}", M=M, N=N, K=K, operands=str(dt), rel_l2=rel_l2(out[:M], want))
    assert rel_l2(out[:M], want) <= 4e-3                       # bf16 output rounding
    # and it IS LayerNorm + Linear (fp32 torch), up to the bf16 operand rounding
    full = torch.nn.functional.layer_norm(x, (K,), gamma, beta) @ w.t() + b
    if epi == "gelu":
        full = full * torch.sigmoid(1.702 * full)
    assert rel_l2(out[:M], full) <= (5e-3 if f16 else 1.2e-2)       # fp16 operands carry 3 more mantissa bits


@pytest.mark.parametrize("M,N,K", [(300, 128, 256), (33024 + 128, 1024, 1024), (128, 1024, 4096), (2000, 768, 3072),
                                   (16384 + 64, 1024, 4096)])
def test_gemm_residual_stats_epilogue_fp16_stream(M, N, K):
    """KEDS_EPI_RESID_STATS_F16: x (fp16, in place) = round(x + a W^T + b) with the sum in fp32, and the per-row
    {sum, sum sq} of the fp32 sums; statistics may be NULL (last block)."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(K + M)
    Mp = (M + 127) // 128 * 128
    a = torch.zeros((Mp, K), dtype=torch.bfloat16, device="cuda")
    a[:M] = (torch.randn(M, K, generator=g, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    x = torch.zeros((Mp, N), dtype=torch.float16, device="cuda")
    x[:M] = (3 * torch.randn(M, N, generator=g, device="cuda")).half()
    want = x[:M].float() + a[:M].float() @ w.float().t() + b             # fp32 sum
    x0 = x.clone()
    stats = torch.zeros((Mp, 2), dtype=torch.int64, device="cuda")
    _lib.ensure_gemm_workspace("cuda")
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(x), N, M, N, K,
                                    _lib.EPI_RESID_STATS_F16, _lib.ptr(stats), 0, None, _lib.stream()), "gemm resid f16")
    # one fp16 rounding of the fp32 sum: within 1 ulp of the rounded reference (accumulation order differs from torch's)
    assert max_abs(x[:M].float(), want.half().float()) <= float(want.abs().max()) * 2.0 ** -10
    assert rel_l2(x[:M].float(), want) <= 4e-4
    sf = (stats.double() / 2.0 ** 28).float()
    assert torch.allclose(sf[:M, 0], want.sum(1), rtol=1e-4, atol=5e-3)
    assert torch.allclose(sf[:M, 1], (want * want).sum(1), rtol=1e-4)
    assert bool((stats[M:] == 0).all()) and bool((x[M:] == 0).all())
    x2 = x0.clone()                                                      # reproducible bits; NULL statistics accepted
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(x2), N, M, N, K,
                                    _lib.EPI_RESID_STATS_F16, None, 0, None, _lib.stream()), "gemm resid f16 2")
    assert torch.equal(x2, x)
    # A/B variant of the 256^2 kernel (bit 10): x + b fed in as the accumulators' initial value instead of being loaded in
    # the epilogue -- only the order of the fp32 additions differs: same reference, same tolerance, and a missing bias is
    # zeros in both
    lib.keds_gemm_force_small((1 << 10) | (3 << 11))       # (3 << 11: on the 8-wave kernel, whatever the shape rule says)
    try:
        x3 = x0.clone()
        _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(x3), N, M, N, K,
                                        _lib.EPI_RESID_STATS_F16, None, 0, None, _lib.stream()), "gemm resid f16 3")
    finally:
        lib.keds_gemm_force_small(0)
    assert max_abs(x3[:M].float(), want.half().float()) <= float(want.abs().max()) * 2.0 ** -10
    assert rel_l2(x3[:M].float(), x[:M].float()) <= 4e-4
    x4 = x0.clone()
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), None, _lib.ptr(x4), N, M, N, K,
                                    _lib.EPI_RESID_STATS_F16, None, 0, None, _lib.stream()), "gemm resid f16 4")
    assert rel_l2(x4[:M].float(), want - b) <= 4e-4


@pytest.mark.parametrize("M,N,K", [(300, 128, 256), (33024 + 128, 1024, 1024), (128, 1024, 4096), (2000, 768, 3072)])
def test_gemm_residual_stats_epilogue(M, N, K):
    """x += a W^T + b in fp32, plus the bf16 copy and the per-row {sum, sum sq} the next folded LayerNorm needs
    (256^2 kernel, 128^2 kernel, remainder rows and the split-K reduce all produce them)."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(K + M)
    Mp = (M + 127) // 128 * 128
    a = torch.zeros((Mp, K), dtype=torch.bfloat16, device="cuda")
    a[:M] = (torch.randn(M, K, generator=g, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    x = torch.zeros((Mp, N), device="cuda")
    x[:M] = torch.randn(M, N, generator=g, device="cuda")
    want = x[:M] + a[:M].float() @ w.float().t() + b
    x0 = x.clone()
    xb = torch.zeros((Mp, N), dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros((Mp, 2), dtype=torch.int64, device="cuda")
    _lib.ensure_gemm_workspace("cuda")
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(x), N, M, N, K,
                                    _lib.EPI_RESID_STATS_F32, _lib.ptr(stats), 0, _lib.ptr(xb), _lib.stream()), "gemm resid")
    assert max_abs(x[:M], want) <= 2e-4
    assert torch.equal(xb[:M], x[:M].to(torch.bfloat16))
    sf = (stats.double() / 2.0 ** 28).float()
    assert torch.allclose(sf[:M, 0], x[:M].sum(1), rtol=1e-4, atol=2e-3)
    assert torch.allclose(sf[:M, 1], (x[:M] * x[:M]).sum(1), rtol=1e-4)
    assert bool((stats[M:] == 0).all()) and bool((xb[M:] == 0).all())
    # integer accumulation: a second run from the same inputs reproduces the statistics bit for bit
    x2 = x0.clone()
    stats2 = torch.zeros((Mp, 2), dtype=torch.int64, device="cuda")
    _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(a), K, _lib.ptr(w), _lib.ptr(b), _lib.ptr(x2), N, M, N, K,
                                    _lib.EPI_RESID_STATS_F32, _lib.ptr(stats2), 0, _lib.ptr(xb), _lib.stream()), "gemm resid 2")
    assert torch.equal(stats2, stats) and torch.equal(x2, x)


def test_gemm_random_shapes_all_epilogues():
    """Seeded sweep over shapes that straddle every dispatch boundary (128^2 / 256^2 kernels, remainder rows, split-K,
    ring depth) for every epilogue, against torch fp32 on the rounded operands."""
    lib = _lib.load()
    rs = np.random.RandomState(123)
    _lib.ensure_gemm_workspace("cuda")
    cases = 0
    for _ in range(28):
        M = int(rs.choice([1, 7, 128, 129, 255, 256, 257, 1000, 4100, 33000, 66000]))
        N = int(rs.choice([128, 256, 384, 768, 1024, 2304]))
        K = int(rs.choice([64, 128, 320, 768, 1024, 2048, 4096]))
        if M * N * K > 3.5e11:
            continue
        epi = int(rs.choice([_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16, _lib.EPI_BIAS_RELU_BF16, _lib.EPI_BIAS_RESID_F32,
                             _lib.EPI_BIAS_F32]))
        g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
        Mp = (M + 127) // 128 * 128
        a = torch.zeros((Mp, K), dtype=torch.bfloat16, device="cuda")
        a[:M] = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device="cuda") * 0.3
        f32 = epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_BIAS_F32)
        out = torch.randn((Mp, N), generator=g, device="cuda") if f32 else torch.zeros((Mp, N), dtype=torch.bfloat16, device="cuda")
        ref = a[:M].float() @ w.float().t() + b
        if epi == _lib.EPI_BIAS_QGELU_BF16:
            ref = ref * torch.sigmoid(1.702 * ref)
        elif epi == _lib.EPI_BIAS_RELU_BF16:
            ref = torch.relu(ref)
        elif epi == _lib.EPI_BIAS_RESID_F32:
            ref = ref + out[:M]
        guard = out[M:].clone()
        ops.gemm_bt(a, w, b, epi, out=out, m=M)
        err = rel_l2(out[:M], ref)
        assert err <= (2e-6 if f32 else 4e-3), (M, N, K, epi, err)
        assert torch.equal(out[M:], guard), "rows past M were written"
        cases += 1
    assert cases >= 20


@pytest.mark.parametrize("H,W", [(480, 640), (640, 427), (224, 224), (500, 224), (231, 1000), (150, 180), (224, 300)])
def test_gpu_preprocess_equals_pil_transform_bit_for_bit(H, W):
    """keds_preprocess_pil vs the PIL pipeline of keds_amd.clip._transform (the reference's eval `_transform`,
    src/model/clip.py:107-123): byte work, so the uint8 image after resize + crop must be EQUAL to PIL's, and the normalised
    float tensor equal to ToTensor + Normalize on it (same fp32 operations).  Down- and up-scaling, a pass that PIL skips
    (one side already n_px), both orientations."""
    from PIL import Image
    from keds_amd import clip as kclip
    rs = np.random.RandomState(H + W)
    base = rs.randint(0, 256, (H // 8 + 1, W // 8 + 1, 3)).astype(np.uint8)           # smooth-ish content + noise
    arr = np.asarray(Image.fromarray(base).resize((W, H), Image.BILINEAR), dtype=np.int16) + rs.randint(-20, 21, (H, W, 3))
    arr = np.clip(arr, 0, 255).astype(np.uint8)
    pil = kclip._center_crop(kclip._resize_shorter_side(Image.fromarray(arr), 224), 224)
    want_u8 = torch.from_numpy(np.asarray(pil))
    want = kclip._transform(224, is_train=False)(Image.fromarray(arr))
    got, got_u8 = ops.preprocess(torch.from_numpy(arr)[None].cuda(), 224, return_u8=True)
    assert torch.equal(got_u8[0].cpu(), want_u8), "uint8 image after resize + crop differs from PIL"
    assert torch.equal(got[0].cpu(), want), "normalised tensor differs from ToTensor + Normalize"
    two = ops.preprocess(torch.from_numpy(np.stack([arr, arr[::-1].copy()])).cuda(), 224)
    assert torch.equal(two[0].cpu(), got[0].cpu())


@pytest.mark.parametrize("nq,ng", [(37, 8193), (20, 17000), (5, 40000)])
def test_large_gallery_ranking_and_imgnet_metric(nq, ng):
    """Galleries beyond one LDS sort (chunk sort + run merges): the full ordering equals torch's stable argsort, and the
    ImageNet-style multi-positive metric on top of it equals the oracle's."""
    dim = 64
    g = torch.Generator().manual_seed(ng)
    gal = torch.nn.functional.normalize(torch.randn(ng, dim, generator=g), dim=1)
    ref = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g), dim=1)
    gal[5] = gal[3]                                               # exact ties: lower index first
    gal[ng - 1] = gal[ng - 2]
    order = ops.rank_gallery(ref.cuda(), gal.cuda()).cpu()
    dist = 1.0 - ref.cuda() @ gal.cuda().t()                       # same fp32 evaluation order is not guaranteed: compare by keys
    want = torch.sort(dist, dim=1, stable=True).indices.cpu()
    same = (order.long() == want)
    if not bool(same.all()):                                       # tolerate swaps of keys that differ only by fp32 summation order
        d = dist.cpu()
        got_d = torch.gather(d, 1, order.long())
        assert bool((got_d[:, 1:] >= got_d[:, :-1] - 2e-6).all()), "not sorted"
        assert float(1.0 - same.float().mean()) < 1e-2
    assert bool((torch.sort(order.long(), dim=1).values == torch.arange(ng)[None, :]).all()), "not a permutation"
    tl = torch.randint(0, 30, (ng,), generator=g)
    ql = torch.randint(0, 30, (nq,), generator=g)
    mg = keds_amd.get_metrics_imgnet(ref.cuda(), gal.cuda(), ql, tl)
    same_label = tl[order.long()] == ql[:, None]                   # the metric's definition evaluated on this very ordering
    total = same_label.sum(1).float()
    for k in (1, 5, 10, 50, 100, 200):
        hits = same_label[:, :k].sum(1).float()
        assert abs(float((hits / (total + 1e-5)).mean()) - mg[f"Real2Sketch_R@{k}"]) < 1e-6
        assert abs(float((hits / k).mean()) - mg[f"Real2Sketch_P@{k}"]) < 1e-6
    mo = O.get_metrics_imgnet(ref, gal, ql, tl)                    # and close to the oracle (near-tie flips move single hits)
    for k, v in mo.items():
        assert abs(v - mg[k]) <= 1.0 / nq + 1e-6, (k, v, mg[k])


@pytest.mark.parametrize("M,K", [(4096, 256), (4096, 1024), (8192 + 1024, 256), (12288, 512), (11008, 1024), (11008, 256), (4096, 2048)])
def test_gemm_quad_kernel_bit_identical_to_the_eight_wave_kernel(M, K):
    """The 4-wave 256 x 256 kernel (round 3: one wave per SIMD, 128 x 128 outputs per wave, accumulators in fixed AGPRs) runs
    the same MFMA chains in the same k order and the same epilogues as the 8-wave kernel: every epilogue the towers use must
    give the same bits (plain bias -> bf16, LayerNorm-folded bias / QuickGELU on fp16 operands, fp16 residual + statistics).
    Its persistent form (one workgroup per CU walks the tiles; M = 12288 / 11008 here: 768 / 688 tiles = three rounds, the last
    one ragged -- fewer tiles than 85 % of whole rounds go to the 128 x 128 kernel and would not test it)
    must as well -- including the LayerNorm epilogue's deferred stores (K >= 512: 18 of a lane's 32 stores of a tile are issued
    from inside the next tile's K-loop; K = 256 is too short for the trickle and stores everything in the epilogue).
    What the dispatcher picks by shape (`default`) must give the same bits too, with the other statistics buffer cleared for
    exactly the rows of the launch.  (The two-accumulator-set kernel of round 5 is not in the product library: tools/experiments/.)"""
    lib = _lib.load()
    N = 4096                                                     # >= 256 tiles of 256 x 256: the big-tile path
    g = torch.Generator(device="cuda").manual_seed(K)
    x = torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.2
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    gamma = 1 + 0.2 * torch.randn(K, generator=g, device="cuda")
    beta = 0.1 * torch.randn(K, generator=g, device="cuda")
    wf, bc = _fold(w, b, gamma, beta, torch.float16)
    xh = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    stats = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
    _lib.check(lib.keds_rowstats_cast_ex(_lib.ptr(x), _lib.ptr(xh), 1, _lib.ptr(stats), M, K, _lib.stream()), "rowstats")
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    resid0 = (2 * torch.randn(M, N, generator=g, device="cuda")).half()
    _lib.ensure_gemm_workspace("cuda")

    def run(flag):
        lib.keds_gemm_force_small(flag)
        try:
            outs = []
            o = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(xb), K, _lib.ptr(wb), _lib.ptr(b), _lib.ptr(o), N, M, N, K, _lib.EPI_BIAS_BF16,
                                            None, 0, None, _lib.stream()), "plain")
            outs.append(o)
            for code in (_lib.EPI_LN_BIAS_BF16_H, _lib.EPI_LN_QGELU_BF16_H):
                o = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
                other = torch.full((M, 2), 7, dtype=torch.int64, device="cuda")
                _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(xh), K, _lib.ptr(wf), _lib.ptr(bc), _lib.ptr(o), N, M, N, K, code,
                                                _lib.ptr(stats), 0, _lib.ptr(other), _lib.stream()), "ln")
                outs += [o, other]
            r = resid0.clone()
            st = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
            _lib.check(lib.keds_gemm_bt_ex2(_lib.ptr(xb), K, _lib.ptr(wb), _lib.ptr(b), _lib.ptr(r), N, M, N, K,
                                            _lib.EPI_RESID_STATS_F16, _lib.ptr(st), 0, None, _lib.stream()), "resid")
            outs += [r, st]
            torch.cuda.synchronize()
            return outs
        finally:
            lib.keds_gemm_force_small(0)
    eight, four, deep = run(3 << 11), run(1 << 11), run(2 << 11)       # (2 << 11: the 4-wave kernel's persistent form)
    default = run(0)                                                   # what the dispatcher picks by shape (quad_by_shape)
    want = xb.float() @ wb.float().t() + b
    assert rel_l2(four[0], want) <= 4e-3
    for a, c, d, e in zip(eight, four, deep, default):
        assert torch.equal(a, c) and torch.equal(a, d) and torch.equal(a, e)
