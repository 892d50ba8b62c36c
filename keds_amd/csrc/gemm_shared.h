// Pieces shared by the GEMM translation units (gemm.hip, gemm_duo.hip): the LDS image's swizzle and W-row permutation, the
// epilogue classes, the LayerNorm row coefficients and the numerics guard.  Everything here is internal to the library.
#pragma once
#include "keds_common.h"
#include <math.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
// LDS row r (128 bytes = 8 chunks of 16 B): chunk c is stored at slot c ^ f(r)
__device__ __forceinline__ int swz_f(int row) { return (row >> 1) & 7; }

// LDS row R (0..127) of the W tile holds W row n0 + perm_w(R): with i = R&15 (MFMA row), g = i>>2,
// r = i&3, tile t = R>>4: n = 64*(t>>2) + 32*((t>>1)&1) + 8*g + 4*(t&1) + r, so the accumulator
// registers of tiles (2p, 2p+1) of one lane are 8 consecutive output columns.
__device__ __forceinline__ int perm_w(int R) {
    const int t = R >> 4, i = R & 15;
    return 64 * (t >> 2) + 32 * ((t >> 1) & 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
}

// KEDS_EPI_LN_*_H: the same epilogues with fp16 operands (A = the fp16 residual stream, W' folded to fp16)
// KEDS_EPI_X3_*: split-operand GEMMs (keds_hip.h): fp16 MFMA, three K segments (hi.hi, hi.lo, lo.hi) over two operand planes
constexpr bool epi_x3(int e) { return e == KEDS_EPI_X3_BIAS_F32 || e == KEDS_EPI_X3_RESID_F32 || e == KEDS_EPI_X3_QGELU_PAIR; }
constexpr bool epi_ln_h(int e) { return e == KEDS_EPI_LN_BIAS_BF16_H || e == KEDS_EPI_LN_QGELU_BF16_H; }
constexpr bool epi_f16(int e) { return epi_ln_h(e) || epi_x3(e); }          // fp16 (not bf16) MFMA operands
constexpr bool epi_is_ln(int e) { return e == KEDS_EPI_LN_BIAS_BF16 || e == KEDS_EPI_LN_QGELU_BF16 || epi_ln_h(e); }
constexpr int epi_base(int e) {
    return (e == KEDS_EPI_LN_BIAS_BF16 || e == KEDS_EPI_LN_BIAS_BF16_H)     ? KEDS_EPI_BIAS_BF16
           : (e == KEDS_EPI_LN_QGELU_BF16 || e == KEDS_EPI_LN_QGELU_BF16_H) ? KEDS_EPI_BIAS_QGELU_BF16
           : e == KEDS_EPI_X3_BIAS_F32                                      ? KEDS_EPI_BIAS_F32
           : e == KEDS_EPI_X3_RESID_F32                                     ? KEDS_EPI_BIAS_RESID_F32
                                                                            : e;
}
// K-tile p (of 3 * np1) of a split-operand GEMM: byte offset along K inside a plane, and which plane of A / W it reads
struct X3Seg {
    unsigned koff;      // bytes: (p mod np1) * 128
    bool a_lo, w_lo;
};
__device__ __forceinline__ X3Seg x3_seg(int p, int np1) {
    const int seg = (p >= np1) + (p >= 2 * np1);
    return X3Seg{(unsigned)(p - seg * np1) * 128u, seg == 2, seg == 1};
}

// KEDS_EPI_X3_*: the W planes hold W * 2^e (keds_split_f16_weight), the epilogue multiplies the accumulators by 2^-e before the
// bias.  Inside the library e travels in bits 40-47 of the kernels' `w_plane` argument (the plane stride itself is < 2^30 and every
// use of it truncates to 32 bits), so that no kernel signature and no launch site changes.
inline long long x3_pack_wplane(long long w_plane, int w_exp) { return w_plane | ((long long)(w_exp & 0xFF) << 40); }
__device__ __forceinline__ float x3_wscale(long long w_plane) {
    const int e = (int)(signed char)((w_plane >> 40) & 0xFF);
    return __uint_as_float((unsigned)(127 - e) << 23);             // 2^-e, |e| <= 100
}

constexpr float LN_EPS = 1e-5f;

// Numerics guard of the folded-LayerNorm flow (keds_hip.h, keds_numerics_guard): the GEMM multiplies UN-centred rows, so
// operand rounding is amplified by |row mean| / row std = |nmr| (rel-L2 4.8e-3 at 20, 1.8e-2 at 100),
// and an fp16 residual stream that overflowed shows up as non-finite statistics.  Either raises the caller's flag; the
// host then re-runs the pass on the fp32-stream flow with stand-alone LayerNorm.
constexpr float GUARD_MAX_MEAN_OVER_STD = 32.0f;
__device__ __forceinline__ void guard_check(int* __restrict__ guard, float nmr) {
    if (guard && !(fabsf(nmr) <= GUARD_MAX_MEAN_OVER_STD)) *guard = 1;       // NaN / inf fail the comparison too
}

__device__ __forceinline__ void ln_coeff_from(keds_stat_t s_fixed, keds_stat_t ss_fixed, float invk, float& rstd, float& nmr,
                                              int* __restrict__ guard = nullptr) {
    const float mean = keds_stat_value(s_fixed) * invk;
    const float var = fmaxf(keds_stat_value(ss_fixed) * invk - mean * mean, 0.f);
    rstd = rsqrtf(var + LN_EPS);
    nmr = -mean * rstd;
    guard_check(guard, nmr);
}

}  // namespace
