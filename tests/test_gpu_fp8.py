"""GPU parity of the MXFP8 building blocks (BASELINE config 5: fp8 encoders): the OCP-MX quantiser and the block-scaled
fp8 GEMM (v_mfma_scale_f32_16x16x128_f8f6f4).

Bars: the quantiser is bit-exact against a torch restatement of the OCP MX rule (shared exponent floor(log2 amax) - 8,
round-to-nearest-even to e4m3, saturation at 448); the GEMM equals the fp32 product of the DEQUANTISED operands up to
the bf16 rounding of the output (the block scales are powers of two, so the hardware path is exact up to fp32
accumulation order), and is within fp8 tolerance (rel-L2 <= 8e-2: two e4m3 operands, 3 mantissa bits each) of the unquantised fp32 product."""
import numpy as np
import pytest
import torch

from keds_amd import _lib
from tests.gpu_util import rel_l2, report

pytestmark = pytest.mark.gpu


def _quantize(x, rows_pad=None):
    lib = _lib.load()
    rows, K = x.shape
    rows_pad = rows_pad or (rows + 255) // 256 * 256
    q = torch.zeros((rows, K), dtype=torch.uint8, device="cuda")
    s = torch.full((K // 128, rows_pad, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_quantize_mxfp8(_lib.ptr(x), 1 if x.dtype == torch.bfloat16 else 0, rows, K, rows_pad, _lib.ptr(q),
                                       _lib.ptr(s), _lib.stream()), "quantize")
    return q, s


def _dequantize(q, s):
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).float().reshape(rows, K // 32, 32)
    e = s[:, :rows, :].permute(1, 0, 2).reshape(rows, K // 32).float() - 127.0           # [rows, K/32]
    return (v * torch.exp2(e)[:, :, None]).reshape(rows, K)


def _torch_mx(x):
    rows, K = x.shape
    b = x.float().reshape(rows, K // 32, 32)
    amax = b.abs().amax(dim=2)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax)) - 8, torch.full_like(amax, -127.0)).clamp(-127, 127)
    scaled = (b * torch.exp2(-e)[:, :, None]).clamp(-448, 448)
    scaled = torch.where(amax[:, :, None] > 0, scaled, torch.zeros_like(scaled))
    return scaled.to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, K), (e + 127).to(torch.uint8)


@pytest.mark.parametrize("rows,K,dtype", [(300, 256, torch.float32), (1024, 1024, torch.bfloat16), (257, 4096, torch.float32)])
def test_quantizer_matches_ocp_mx_rule(rows, K, dtype):
    g = torch.Generator(device="cuda").manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g, device="cuda") * torch.exp(2.0 * torch.randn(rows, 1, generator=g, device="cuda"))
    x[3, 64:96] = 0                                               # an all-zero block
    x[5, 0] = 3.0e4                                               # an outlier: its block saturates nothing else
    x = x.to(dtype)
    q, s = _quantize(x)
    qt, st = _torch_mx(x)
    assert torch.equal(s[:, :rows, :].permute(1, 0, 2).reshape(rows, K // 32), st)
    assert torch.equal(q, qt)
    err = rel_l2(_dequantize(q, s), x.float())
    report("mxfp8_quantizer", rows=rows, K=K, rel_l2=err)
    assert err <= 6e-2


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 1024), (2048, 1024, 4096)])
def test_mxfp8_gemm_equals_product_of_dequantised_operands(M, N, K):
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device="cuda") * torch.exp(torch.randn(M, 1, generator=g, device="cuda"))
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    bias = torch.randn(N, generator=g, device="cuda") * 0.1
    aq, as_ = _quantize(a)
    wq, ws = _quantize(w)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                   _lib.ptr(bias), _lib.ptr(out), M, N, K, _lib.stream()), "gemm_mxfp8")
    want = _dequantize(aq, as_) @ _dequantize(wq, ws).t() + bias
    exact = rel_l2(out, want)
    full = rel_l2(out, a @ w.t() + bias)
    report("mxfp8_gemm", M=M, N=N, K=K, rel_l2_vs_dequantised=exact, rel_l2_vs_fp32=full)
    assert exact <= 4e-3                                          # bf16 rounding of the output only
    assert full <= 8e-2


def test_mxfp8_gemm_argument_errors():
    lib = _lib.load()
    assert lib.keds_gemm_mxfp8(1, 1, 256, 1, 1, 256, None, 1, 200, 256, 256, None) == -1 and "256" in _lib.last_error()
    assert lib.keds_gemm_mxfp8(1, 1, 256, 1, 1, 256, None, 1, 256, 256, 128, None) == -1
    assert lib.keds_mxfp8_scale_bytes(256, 1024) == 8 * 256 * 4 and lib.keds_mxfp8_scale_bytes(256, 100) == 0


# ---- fused epilogues of the fp8 tower ---------------------------------------------------------------------------------
def _fold_fp8(w, b, gamma, beta):
    lib = _lib.load()
    n, k = w.shape
    n_pad = (n + 255) // 256 * 256
    wq = torch.zeros((n, k), dtype=torch.uint8, device="cuda")
    ws = torch.full((k // 128, n_pad, 4), 127, dtype=torch.uint8, device="cuda")
    bc = torch.zeros(2 * n, device="cuda")
    _lib.check(lib.keds_fold_layernorm_mxfp8(_lib.ptr(w), _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), n, k, n_pad, _lib.ptr(wq),
                                             _lib.ptr(ws), _lib.ptr(bc), _lib.stream()), "fold fp8")
    return wq, ws, bc


def test_fold_layernorm_mxfp8_weight_preparation():
    g = torch.Generator(device="cuda").manual_seed(5)
    N, K = 768, 1024
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    gamma = 1 + 0.2 * torch.randn(K, generator=g, device="cuda")
    beta = 0.1 * torch.randn(K, generator=g, device="cuda")
    wq, ws, bc = _fold_fp8(w, b, gamma, beta)
    qt, st = _torch_mx(w * gamma)
    assert torch.equal(wq, qt) and torch.equal(ws[:, :N, :].permute(1, 0, 2).reshape(N, K // 32), st)
    deq = _dequantize(wq, ws)
    assert torch.allclose(bc[N:], deq.sum(1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(bc[:N], b + w @ beta, rtol=1e-5, atol=1e-5)
    wq2, ws2, bc2 = _fold_fp8(w, b, None, None)                   # plain quantisation
    qt2, _ = _torch_mx(w)
    assert torch.equal(wq2, qt2) and torch.allclose(bc2[:N], b)


@pytest.mark.parametrize("M,N,K,gelu", [(512, 768, 1024, False), (1024, 1024, 256, True), (2048, 4096, 1024, True)])
def test_mxfp8_layernorm_epilogues(M, N, K, gelu):
    """LayerNorm folded into the fp8 GEMM: equals Linear(LayerNorm(x)) at fp8 tolerance and, exactly (up to the output
    rounding), the formula on the dequantised operands; the QuickGELU variant emits a valid MXFP8 tensor."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, generator=g, device="cuda") * 1.5 + 0.2 * torch.randn(M, 1, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    gamma = 1 + 0.2 * torch.randn(K, generator=g, device="cuda")
    beta = 0.1 * torch.randn(K, generator=g, device="cuda")
    wq, ws, bc = _fold_fp8(w, b, gamma, beta)
    xq, xs = _quantize(x)
    stats = (torch.stack([x.sum(1), (x * x).sum(1)], dim=1).double() * 2.0 ** 28).round().to(torch.int64).contiguous()
    other = torch.full((M + 8, 2), 7, dtype=torch.int64, device="cuda")           # statistics are 64-bit fixed point (* 2^28)
    mean, rstd = x.mean(1, keepdim=True), torch.rsqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    want = rstd * (_dequantize(xq, xs) @ _dequantize(wq, ws).t() - mean * bc[N:][None, :]) + bc[:N][None, :]
    full = torch.nn.functional.layer_norm(x, (K,), gamma, beta) @ w.t() + b
    if gelu:
        want, full = want * torch.sigmoid(1.702 * want), full * torch.sigmoid(1.702 * full)
        q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
        qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
        _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(xq), _lib.ptr(xs), xs.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                          _lib.ptr(bc), None, M, N, K, _lib.FP8_EPI_LN_QGELU_MX, _lib.ptr(stats), _lib.ptr(other),
                                          _lib.ptr(q), _lib.ptr(qs), M, _lib.stream()), "gemm fp8 ln gelu")
        got = _dequantize(q, qs)
        qt, st = _torch_mx(got)                                    # idempotent: re-quantising the decoded tensor reproduces it
        assert torch.equal(qt, q)
        tol_exact = 5e-2                                           # the output itself is e4m3 now
    else:
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(xq), _lib.ptr(xs), xs.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                          _lib.ptr(bc), _lib.ptr(out), M, N, K, _lib.FP8_EPI_LN_BIAS_BF16, _lib.ptr(stats),
                                          _lib.ptr(other), None, None, 0, _lib.stream()), "gemm fp8 ln")
        got = out.float()
        tol_exact = 4e-3
    assert bool((other[:M] == 0).all()) and bool((other[M:] == 7).all())
    report("mxfp8_ln_epilogue", M=M, N=N, K=K, gelu=gelu, rel_l2_formula=rel_l2(got, want), rel_l2_fp32=rel_l2(got, full))
    assert rel_l2(got, want) <= tol_exact
    assert rel_l2(got, full) <= 9e-2


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1024, 1024, 4096)])
def test_mxfp8_residual_stats_epilogue_fp16_stream(M, N, K):
    """KEDS_FP8_EPI_RESID_STATS_MX_H: the residual stream in fp16 (what the fp8 towers run); the MXFP8 copy and the
    statistics come from the fp32 sum, the stored row is its fp16 rounding."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn(M, K, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    x = (3 * torch.randn(M, N, generator=g, device="cuda")).half()
    aq, as_ = _quantize(a)
    wq, ws = _quantize(w)
    want = x.float() + _dequantize(aq, as_) @ _dequantize(wq, ws).t() + b
    stats = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
    q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
    qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1], _lib.ptr(b),
                                      _lib.ptr(x), M, N, K, _lib.FP8_EPI_RESID_STATS_MX_H, _lib.ptr(stats), None, _lib.ptr(q),
                                      _lib.ptr(qs), M, _lib.stream()), "gemm fp8 resid f16")
    assert rel_l2(x.float(), want) <= 4e-4
    assert float((x.float() - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -10
    qt, st = _torch_mx(want)                                       # MXFP8 copy of the fp32 sums (up to accumulation order)
    same = (q == qt).float().mean().item()
    assert same >= 0.999, same
    sf = (stats.double() / 2.0 ** 28).float()
    assert torch.allclose(sf[:, 0], want.sum(1), rtol=1e-4, atol=5e-3) and torch.allclose(sf[:, 1], (want * want).sum(1), rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1024, 1024, 4096)])
def test_mxfp8_residual_stats_epilogue(M, N, K):
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn(M, K, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    x = torch.randn(M, N, generator=g, device="cuda")
    aq, as_ = _quantize(a)
    wq, ws = _quantize(w)
    want = x + _dequantize(aq, as_) @ _dequantize(wq, ws).t() + b
    stats = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
    q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
    qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1], _lib.ptr(b),
                                      _lib.ptr(x), M, N, K, _lib.FP8_EPI_RESID_STATS_MX, _lib.ptr(stats), None, _lib.ptr(q),
                                      _lib.ptr(qs), M, _lib.stream()), "gemm fp8 resid")
    assert float((x - want).abs().max()) <= 5e-4
    qt, st = _torch_mx(x)                                          # the MXFP8 copy is the quantisation of the NEW rows
    assert torch.equal(q, qt) and torch.equal(qs.permute(1, 0, 2).reshape(M, N // 32), st)
    sf = (stats.double() / 2.0 ** 28).float()
    assert torch.allclose(sf[:, 0], x.sum(1), rtol=1e-4, atol=2e-3) and torch.allclose(sf[:, 1], (x * x).sum(1), rtol=1e-4)


def test_vitl14_fp8_encoder_against_reference_golden():
    """BASELINE config 5: ViT-L/14 with MXFP8 GEMM operands vs the reference's fp32 outputs.  Stated tolerance for the fp8
    path: cosine >= 0.995 per row, rel-L2 <= 0.1 (two e4m3 operands per GEMM, 48 GEMMs deep; bf16 path: 0.9999 / 1.5e-2).
    B = 2 gives 514 rows = two full 256-row MXFP8 tiles + 2 remainder rows on the bf16 kernels; B = 5 adds a ragged case."""
    import keds_amd
    from oracle import keds_oracle as O
    from tests.conftest import golden_path
    from tests.gpu_util import min_cosine
    from tests.test_gpu_model import VITL
    g = dict(np.load(golden_path("clip_vitl14.npz")))
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda().set_precision("fp8")
    img = torch.from_numpy(g["image"]).cuda()
    out = m.encode_image(img)
    c, r = min_cosine(out, g["encode_image"]), rel_l2(out, g["encode_image"])
    report("vitl14_fp8.encode_image", min_cosine=c, rel_l2=r)
    assert torch.isfinite(out).all() and c >= 0.995 and r <= 0.1
    assert torch.equal(out, m.encode_image(img))                  # integer statistics atomics: the same bits every run
    ref = m.set_precision("bf16").encode_image(img)
    assert min_cosine(ref, g["encode_image"]) >= 0.9999            # switching back restores the bf16 path
    assert torch.equal(ref, m.encode_image(img))
    rs = np.random.RandomState(3)
    img5 = torch.from_numpy(rs.standard_normal((5, 3, 224, 224)).astype(np.float32)).cuda()
    a = m.set_precision("fp8").encode_image(img5)
    b = m.set_precision("bf16").encode_image(img5)
    c5 = min_cosine(a, b)
    report("vitl14_fp8.vs_bf16_B5", min_cosine=c5, rel_l2=rel_l2(a, b))
    assert c5 >= 0.995
    with pytest.raises(ValueError):
        m.set_precision("int4")
    # fewer than 256 rows (one prompt = 77 rows): no full tile exists, the bf16 kernels serve the call
    one = torch.from_numpy(g["text"][:1]).cuda()
    assert torch.equal(m.set_precision("fp8").encode_text(one), m.set_precision("bf16").encode_text(one))


def test_attention_mxfp8_output_equals_quantised_bf16_path():
    """keds_attention_mx: rows below q8_rows leave the attention kernel as MXFP8, the rest as bf16.  Quantising the fp32
    result before (here) or after (reference: bf16 output, then the quantiser) the bf16 rounding differs by one bf16 ulp
    of the input at most, so the decoded tensors agree to fp8 resolution and the bf16 rows are bit-identical."""
    from keds_amd import ops
    lib = _lib.load()
    B, S, H = 3, 257, 16
    d = H * 64
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = (torch.randn(B * S, 3 * d, generator=g, device="cuda") * 1.2).to(torch.bfloat16)
    ref = ops.attention(qkv, B, S, H, False)          # bf16 everywhere, from the same kernel (its MX form differs per row only)
    rows8 = 512
    out = torch.zeros((B * S, d), dtype=torch.bfloat16, device="cuda")
    q8 = torch.zeros((rows8, d), dtype=torch.uint8, device="cuda")
    s8 = torch.full((d // 128, rows8, 4), 127, dtype=torch.uint8, device="cuda")
    _lib.check(lib.keds_attention_mx(_lib.ptr(qkv), _lib.ptr(out), B, S, H, 0, S, _lib.ptr(q8), _lib.ptr(s8), rows8,
                                     _lib.stream()), "attention_mx")
    assert torch.equal(out[rows8:], ref[rows8:]) and bool((out[:rows8] == 0).all())
    got = _dequantize(q8, s8)
    assert rel_l2(got, ref[:rows8].float()) <= 4e-2
    qt, st = _torch_mx(got)                                                      # a valid MX tensor: re-quantising reproduces it
    assert torch.equal(qt, q8)


def test_text_tower_and_session_handles_in_fp8():
    """The text tower (width 768 = 3 x 256) follows set_precision('fp8'); the handle ABI takes KEDS_FP8 as compute type
    and gives the same bits as the torch-hosted path."""
    import keds_amd
    from keds_amd import session
    from oracle import keds_oracle as O
    from tests.conftest import golden_path
    from tests.gpu_util import min_cosine
    from tests.test_gpu_model import VITL
    g = dict(np.load(golden_path("clip_vitl14.npz")))
    sd = O.synth_clip_state_dict(**VITL, seed=7)
    m = keds_amd.build_model(sd, fp16=False).cuda().set_precision("fp8")
    text = torch.from_numpy(g["text"]).cuda()
    rs = np.random.RandomState(11)
    many = torch.from_numpy(np.tile(g["text"], (5, 1))).cuda()                  # 10 rows x 77 = 770 rows: three full tiles + 2 rows
    t8 = m.encode_text(many)
    c = min_cosine(t8[:2], g["encode_text"])
    report("vitl14_fp8.encode_text", min_cosine=c, rel_l2=rel_l2(t8[:2], g["encode_text"]))
    assert c >= 0.99            # the 12-block text tower is more sensitive to e4m3 operands than the image tower (measured 0.9935)
    tok3 = torch.from_numpy(np.tile(g["tok3"], (5, 1, 1))).cuda()
    e8 = m.encode_text_img_retrieval(many, tok3, split_ind=265, repeat=False)
    assert min_cosine(e8[:2], g["eti3"]) >= 0.99
    ctx = session.Context(0)
    try:
        vit = session.Vit(ctx, {k: v.numpy() for k, v in sd.items()}, compute=_lib.DT_FP8)
        img = torch.from_numpy(g["image"]).cuda()
        # same kernels and order-independent (integer) statistics: the same bits
        assert torch.equal(vit.forward(img), m.encode_image(img))
        txt = session.Text(ctx, {k: v.numpy() for k, v in sd.items()}, compute=_lib.DT_FP8)
        eot = (many == 49407).int().argmax(dim=1)
        assert torch.equal(txt.forward(many, eot), t8)
        vit.close()
        txt.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("epi,M,N,K", [("bias", 8448, 1024, 1024), ("ln", 8448, 3072, 1024), ("ln_gelu", 8192, 4096, 1024),
                                       ("resid_h", 8448, 1024, 4096), ("resid_h", 8448, 1024, 1024), ("resid", 4352, 1024, 512),
                                       ("ln", 512, 768, 768), ("resid_h", 1280, 1024, 512), ("ln_gelu", 768, 512, 512), ("bias", 65792, 256, 3072)])
def test_mxfp8_four_wave_kernel_is_bit_identical_to_the_eight_wave_kernel(epi, M, N, K):
    """The 4-wave persistent kernel (gemm_mxfp8_quad_kernel: 128 x 128 wave tiles, accumulators in fixed AGPRs, W fragments refilled
    in place, the next tile's K-tiles requested before the epilogue) against the 8-wave kernel (keds_mxfp8_debug(16)) on the same
    operands: every output byte and scale byte equal (the row statistics up to the order of three fp32 additions) -- per accumulator
    both run the same chain of block-scaled MFMAs in K order and share the epilogue code.  Shapes: more tiles than CUs (the persistent walk, ragged 33-row-tile counts), K = 512 (the
    shortest K the 4-wave form takes) ... 4096."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device="cuda") * torch.exp(0.5 * torch.randn(M, 1, generator=g, device="cuda"))
    w = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    aq, as_ = _quantize(a)
    res = {}
    for form in (16, 0):
        lib.keds_mxfp8_debug(form)
        try:
            q = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
            qs = torch.full((N // 128, M, 4), 127, dtype=torch.uint8, device="cuda")
            other = torch.full((M + 8, 2), 7, dtype=torch.int64, device="cuda")
            if epi in ("ln", "ln_gelu"):
                gamma = 1 + 0.2 * torch.randn(K, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda")
                beta = 0.1 * torch.randn(K, generator=torch.Generator(device="cuda").manual_seed(2), device="cuda")
                wq, ws, bc = _fold_fp8(w, b, gamma, beta)
                stats = (torch.stack([a.sum(1), (a * a).sum(1)], dim=1).double() * 2.0 ** 28).round().to(torch.int64).contiguous()
                out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
                code = _lib.FP8_EPI_LN_QGELU_MX if epi == "ln_gelu" else _lib.FP8_EPI_LN_BIAS_BF16
                _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                                  _lib.ptr(bc), None if epi == "ln_gelu" else _lib.ptr(out), M, N, K, code, _lib.ptr(stats),
                                                  _lib.ptr(other), _lib.ptr(q), _lib.ptr(qs), M, _lib.stream()), "gemm fp8 ln")
                res[form] = (out, q, qs, other)
            elif epi == "bias":
                wq, ws = _quantize(w)
                out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
                _lib.check(lib.keds_gemm_mxfp8(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                               _lib.ptr(b), _lib.ptr(out), M, N, K, _lib.stream()), "gemm_mxfp8")
                res[form] = (out,)
            else:
                wq, ws = _quantize(w)
                x = 3 * torch.randn(M, N, generator=torch.Generator(device="cuda").manual_seed(3), device="cuda")
                x = x.half() if epi == "resid_h" else x
                stats = torch.zeros((M, 2), dtype=torch.int64, device="cuda")
                code = _lib.FP8_EPI_RESID_STATS_MX_H if epi == "resid_h" else _lib.FP8_EPI_RESID_STATS_MX
                _lib.check(lib.keds_gemm_mxfp8_ex(_lib.ptr(aq), _lib.ptr(as_), as_.shape[1], _lib.ptr(wq), _lib.ptr(ws), ws.shape[1],
                                                  _lib.ptr(b), _lib.ptr(x), M, N, K, code, _lib.ptr(stats), None, _lib.ptr(q),
                                                  _lib.ptr(qs), M, _lib.stream()), "gemm fp8 resid")
                res[form] = (x, q, qs, stats)
            torch.cuda.synchronize()
        finally:
            lib.keds_mxfp8_debug(0)
    for k, (t8, t4) in enumerate(zip(res[16], res[0])):
        if epi.startswith("resid") and k == 3:
            # row statistics: the 8-wave kernel adds four fixed-point conversions per row and tile (one per wave column), the 4-wave
            # kernel adds the four fp32 partials in LDS and converts once: equal up to the rounding of those three fp32 additions
            a, b = t8.double() / 2.0 ** 28, t4.double() / 2.0 ** 28
            assert float(((a[:, 1] - b[:, 1]).abs() / a[:, 1]).max()) <= 2e-6                        # sum of squares: relative
            assert float(((a[:, 0] - b[:, 0]).abs() / ((N * a[:, 1]).sqrt() + 1.0)).max()) <= 1e-6   # sum: against sum |x| <= sqrt(N ss)
        else:
            assert torch.equal(t8.view(torch.uint8), t4.view(torch.uint8))
    assert float(res[0][1 if epi == "ln_gelu" else 0].float().abs().max()) > 0.1     # (not two all-zero tensors)
