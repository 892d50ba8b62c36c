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
