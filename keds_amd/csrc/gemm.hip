// out[M,N] = epilogue(X[M,K] . W[N,K]^T + bias)  -- every nn.Linear / projection of the path
// (reference: src/model/model.py:309-326 MHA in/out projections and the MLP, :105-123 IM2TEXT,
//  :46-53 CrossAttention projections, :381 conv1 as im2col GEMM).
//
// gfx950 design: 128x128x64 tile, 4 waves (2x2), each wave 64(n) x 64(m) as 4x4 MFMA
// 16x16x32 bf16 tiles with fp32 accumulators.  Both operands are K-contiguous, so W is the MFMA
// A operand and X the B operand: the accumulator then holds, per lane, ONE output row m and runs
// of consecutive n -> 16-byte epilogue stores with no transpose.  Tiles are staged HBM -> LDS by
// LDS-DMA (global_load_lds_dwordx4, double buffered, one barrier per K tile); the LDS image is
// XOR-swizzled through the per-lane SOURCE address so the ds_read_b128 fragment reads are
// bank-conflict free, and W rows are permuted at staging time so a lane's two n-tiles are adjacent.
#include "keds_common.h"
#include <math.h>
#include <cstdlib>
#include "gemm_shared.h"
#include "gemm_quad_gen.h"
#ifdef KEDS_EXPERIMENTS
// tools/experiments/gemm_duo.hip (round 5, a measured negative: docs/findings_r05.md section 1): the two-accumulator-set kernel
// (LayerNorm-folded epilogues under the next unit's MFMAs); only the experiment build links it
bool keds_gemm_duo_ok(int epi, int M, int N, int K);
int keds_gemm_duo_launch(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                         void* aux2, hipStream_t st);
#endif

#ifndef KEDS_QUAD_NOEPI
#define KEDS_QUAD_NOEPI 0
#endif
// cache policy of the persistent 4-wave kernel's LDS-DMA pieces per operand (aux bits of the buffer load: 1 = sc0, 2 = nt, 16 = sc1).
// Per supertile (8 row panels x 4 column tiles on one XCD's 32 CUs) the A panels are 4 MB that nobody on this XCD reads again
// and the W panels 2 MB that the XCD's NEXT supertile reads again (same column group): round 5 A/B, tools/rounds/r05_quad_policy.sh
#ifndef KEDS_QUAD_AUX_X
#define KEDS_QUAD_AUX_X 0
#endif
#ifndef KEDS_QUAD_AUX_W
#define KEDS_QUAD_AUX_W 0
#endif
#ifndef KEDS_QUAD_TIDDMA
#define KEDS_QUAD_TIDDMA 0
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;       // X tile + W tile


// x * sigmoid(1.702 x) = x / (1 + 2^(-1.702*log2(e)*x)); v_exp_f32 + v_rcp_f32 (1 ulp-level, then rounded to bf16)
__device__ __forceinline__ float qgelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
}

// one lane's 8 consecutive output columns [n, n+8) of row m
template <int EPI>
__device__ __forceinline__ void epilogue_store(f32x4 v0, f32x4 v1, void* __restrict__ out, int m, int n, int N,
                                               const float* __restrict__ aux, int aux_i, long long ldc) {
    if constexpr (EPI == KEDS_EPI_BIAS_QGELU_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = qgelu(v0[j]);
            v1[j] = qgelu(v1[j]);
        }
    }
    if constexpr (EPI == KEDS_EPI_BIAS_RELU_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = fmaxf(v0[j], 0.f);
            v1[j] = fmaxf(v1[j], 0.f);
        }
    }
    if constexpr (EPI == KEDS_EPI_BIAS_BF16 || EPI == KEDS_EPI_BIAS_QGELU_BF16 || EPI == KEDS_EPI_BIAS_RELU_BF16 ||
                  EPI == KEDS_EPI_BIAS_BF16_HEADF32) {
        bf16x8 o = bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                          (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + (size_t)m * ldc + n) = o;
        if constexpr (EPI == KEDS_EPI_BIAS_BF16_HEADF32) {
            if (m < aux_i) {                      // the first aux_i rows also in fp32, row stride 3N (token slot of [B,3,N])
                float* t = const_cast<float*>(aux) + (size_t)m * 3 * N + n;
                *reinterpret_cast<f32x4*>(t) = v0;
                *reinterpret_cast<f32x4*>(t + 4) = v1;
            }
        }
    } else if constexpr (EPI == KEDS_EPI_X3_QGELU_PAIR) {
        // QuickGELU in full precision (model.py:300-302; f32path.hip's form), then the value as two fp16 planes: hi = fp16(v),
        // lo = fp16(v - hi) -- the A operand of the next split-operand GEMM (aux_i = elements between the planes)
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = j < 4 ? v0[j] : v1[j - 4];
            v = v / (1.0f + expf(-1.702f * v));
            hi[j] = (f16_t)v;
            lo[j] = (f16_t)(v - (float)hi[j]);
        }
        f16_t* o = reinterpret_cast<f16_t*>(out) + (size_t)m * ldc + n;
        *reinterpret_cast<f16x8*>(o) = hi;
        *reinterpret_cast<f16x8*>(o + (size_t)aux_i) = lo;
    } else if constexpr (EPI == KEDS_EPI_BIAS_RESID_F32) {
        float* o = reinterpret_cast<float*>(out) + (size_t)m * ldc + n;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(o);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(o + 4);
        *reinterpret_cast<f32x4*>(o) = r0 + v0;
        *reinterpret_cast<f32x4*>(o + 4) = r1 + v1;
    } else if constexpr (EPI == KEDS_EPI_BIAS_F32) {
        float* o = reinterpret_cast<float*>(out) + (size_t)m * ldc + n;
        *reinterpret_cast<f32x4*>(o) = v0;
        *reinterpret_cast<f32x4*>(o + 4) = v1;
    } else {  // KEDS_EPI_PATCH_F32: token row (m/G)*(G+1) + 1 + m%G, plus positional embedding
        const int G = aux_i;
        const int b = m / G, pidx = m - b * G;
        float* o = reinterpret_cast<float*>(out) + ((size_t)b * (G + 1) + 1 + pidx) * N + n;
        const float* pe = aux + (size_t)(1 + pidx) * N + n;
        *reinterpret_cast<f32x4*>(o) = v0 + *reinterpret_cast<const f32x4*>(pe);
        *reinterpret_cast<f32x4*>(o + 4) = v1 + *reinterpret_cast<const f32x4*>(pe + 4);
    }
}

// one 16x16x32 MFMA on fragments staged as raw 16-byte chunks: bf16 or fp16 operands, fp32 accumulate (same rate)
template <bool F16>
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LayerNorm statistics of row m from {sum, sum of squares}: returns (rstd, -mean * rstd)

__device__ __forceinline__ void ln_row_coeff(const float* __restrict__ stats_, int m, float invk, float& rstd, float& nmr,
                                             int* __restrict__ guard = nullptr) {
    const keds_stat_t* stats = reinterpret_cast<const keds_stat_t*>(stats_);
    const float s = keds_stat_value(stats[2 * (size_t)m]), ss = keds_stat_value(stats[2 * (size_t)m + 1]);
    const float mean = s * invk;
    const float var = fmaxf(ss * invk - mean * mean, 0.f);
    rstd = rsqrtf(var + LN_EPS);
    nmr = -mean * rstd;
    guard_check(guard, nmr);
}


__device__ __forceinline__ float sum8(f32x4 a, f32x4 b) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])); }

// Epilogue of one wave's accumulator tile, shared by the 128^2 and 256^2 kernels: lane (g, c) owns rows
// m_lane + 16*mi (mi < MI) and columns n_lane + 32*p + 0..7 (p = 0, 1) held in acc[2p][mi], acc[2p+1][mi].
template <int EPI, int MI, int DBG = 0>   // DBG (stamped diagnostic build only): 2 = no statistics loads, 3 = no stores
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[4][MI], const float* __restrict__ bias, void* __restrict__ out,
                                              int m_lane, int M, int n_lane, int N, int K, const float* __restrict__ aux,
                                              int aux_i, void* __restrict__ aux2, long long ldc, bool zero_lane,
                                              int* __restrict__ guard = nullptr, float x3_ws = 1.0f) {
    if constexpr (epi_is_ln(EPI)) {
        const float invk = 1.0f / (float)K;
        float rstd[MI], nmr[MI];
        keds_stat_t* zero = reinterpret_cast<keds_stat_t*>(aux2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m_lane + 16 * mi;
            if constexpr (DBG == 2) {
                rstd[mi] = 1.0f + invk;
                nmr[mi] = invk;
            } else
                ln_row_coeff(aux, m < M ? m : M - 1, invk, rstd[mi], nmr[mi], zero_lane ? guard : nullptr);
            if (zero && zero_lane && m < M) keds_stat_zero(zero + 2 * (size_t)m);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n_lane + 32 * p;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + n), b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(bias + N + n), c1 = *reinterpret_cast<const f32x4*>(bias + N + n + 4);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m_lane + 16 * mi;
                if (m >= M) continue;
                if constexpr (DBG == 1) {      // TIMING ONLY (wrong placement): every store instruction covers 8 rows x 128 B
                    const int c_ = m & 15, row = m - c_ + (c_ & 7) + 8 * p;
                    const int col = (n - 32 * p) - (n & 31) + ((n & 31) >> 3) * 8 + 32 * (c_ >> 3);
                    epilogue_store<epi_base(EPI)>(acc[2 * p][mi] * rstd[mi] + (c0 * nmr[mi] + b0),
                                                  acc[2 * p + 1][mi] * rstd[mi] + (c1 * nmr[mi] + b1), out, row, col, N, nullptr, 0, ldc);
                    continue;
                }
                if constexpr (DBG == 3) {      // keep the math alive, store (almost) nothing
                    const f32x4 v = acc[2 * p][mi] * rstd[mi] + (c0 * nmr[mi] + b0) + acc[2 * p + 1][mi] * rstd[mi] + (c1 * nmr[mi] + b1);
                    if (v[0] + v[1] + v[2] + v[3] == 12345.678f) reinterpret_cast<float*>(out)[0] = v[0];
                    continue;
                }
                epilogue_store<epi_base(EPI)>(acc[2 * p][mi] * rstd[mi] + (c0 * nmr[mi] + b0),
                                              acc[2 * p + 1][mi] * rstd[mi] + (c1 * nmr[mi] + b1), out, m, n, N, nullptr, 0, ldc);
            }
        }
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F32) {
        // (Measured: batching all of a tile's residual loads up front -- inline-asm loads, hand-counted vmcnt, no
        // per-row round trips -- leaves this epilogue's cost unchanged.  It is the burst of 10 B per element that all
        // CUs issue at the same moment, not the dependent chain, that takes the time.)
        keds_stat_t* stats = reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux));
        bf16_t* xb = reinterpret_cast<bf16_t*>(aux2);
        f32x4 b[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            b[p][0] = b[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias) {
                b[p][0] = *reinterpret_cast<const f32x4*>(bias + n_lane + 32 * p);
                b[p][1] = *reinterpret_cast<const f32x4*>(bias + n_lane + 32 * p + 4);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m_lane + 16 * mi;
            const bool valid = m < M;
            float s = 0.f, ss = 0.f;
            if (valid) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int n = n_lane + 32 * p;
                    float* o = reinterpret_cast<float*>(out) + (size_t)m * ldc + n;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(o) + (acc[2 * p][mi] + b[p][0]);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(o + 4) + (acc[2 * p + 1][mi] + b[p][1]);
                    *reinterpret_cast<f32x4*>(o) = v0;
                    *reinterpret_cast<f32x4*>(o + 4) = v1;
                    *reinterpret_cast<bf16x8*>(xb + (size_t)m * N + n) =
                        bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                               (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
                    s += sum8(v0, v1);
                    ss += sum8(v0 * v0, v1 * v1);
                }
            }
            // the four lanes (g = 0..3) that share row m hold this wave's 64 columns of it
            s = rows_sum(s);
            ss = rows_sum(ss);
            if (valid && zero_lane) keds_stat_add(stats + 2 * (size_t)m, s, ss);
        }
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
        // the residual stream kept in fp16 (the reference's own storage type, model.py:531-548 convert_weights): one
        // copy is both the residual and the next GEMM's operand, 4 B per element of traffic instead of 10
        keds_stat_t* stats = reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux));
        f32x4 b[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            b[p][0] = b[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias) {
                b[p][0] = *reinterpret_cast<const f32x4*>(bias + n_lane + 32 * p);
                b[p][1] = *reinterpret_cast<const f32x4*>(bias + n_lane + 32 * p + 4);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m_lane + 16 * mi;
            const bool valid = m < M;
            float s = 0.f, ss = 0.f;
            if (valid) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f16x8* o = reinterpret_cast<f16x8*>(reinterpret_cast<f16_t*>(out) + (size_t)m * ldc + n_lane + 32 * p);
                    const f16x8 r = *o;
                    const f32x4 v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]} + (acc[2 * p][mi] + b[p][0]);
                    const f32x4 v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]} + (acc[2 * p + 1][mi] + b[p][1]);
                    *o = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3],
                               (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
                    s += sum8(v0, v1);
                    ss += sum8(v0 * v0, v1 * v1);
                }
            }
            s = rows_sum(s);
            ss = rows_sum(ss);
            if (valid && zero_lane && stats) keds_stat_add(stats + 2 * (size_t)m, s, ss);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n_lane + 32 * p;
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (bias) {
                b0 = *reinterpret_cast<const f32x4*>(bias + n);
                b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m_lane + 16 * mi;
                if (m >= M) continue;
                if constexpr (epi_x3(EPI))      // the weight planes hold W * 2^e: the product comes back to scale before the bias (exact)
                    epilogue_store<epi_base(EPI)>(acc[2 * p][mi] * x3_ws + b0, acc[2 * p + 1][mi] * x3_ws + b1, out, m, n, N, aux, aux_i, ldc);
                else
                    epilogue_store<epi_base(EPI)>(acc[2 * p][mi] + b0, acc[2 * p + 1][mi] + b1, out, m, n, N, aux, aux_i, ldc);
            }
        }
    }
}

// LN epilogues of the 256^2 kernel.  The row coefficients {rstd, -mean rstd} and the tile's bias' / column-sum slices were
// computed once per workgroup in the prologue (their loads and the 64-bit fixed-point -> float conversions hide behind the
// wait for the first K-tile) and sit in the LDS side area; every row of the tile is valid (M % 256 == 0), and a lane's store
// address is a uniform tile base + a 32-bit offset.  DBG 3 (stamped diagnostic build): no stores.
// ND > 0 (persistent 4-wave kernel; H = the wave's 64-column half this call covers, PSEL = its 32-column quarter, -1: both):
// a lane's 32 stores of a tile are numbered s = 16 H + 8 p + mi; with `defer` the last ND of them are NOT issued but handed
// back in `pend` (pend[s - (32 - ND)]); the caller issues them between the first K-steps of its next tile
// (quad_flush_pending), where nothing competes with them -- see the note at the persistent kernel.
template <int EPI>
constexpr int quad_nd() { return 18; }       // (measured on qkv, same-process A/B: 12 stores -7.7 us, 18 -8.7, 22 -9.3 of 179)
template <int EPI, int DBG, int ND = 0, int H = 0, int PSEL = -1>
__device__ __forceinline__ void pair_ln_epilogue(f32x4 (&acc)[4][8], const char* __restrict__ side, void* __restrict__ out,
                                                 int m0, int n0, int N, int wm, int wn, int g, int c,
                                                 void* __restrict__ aux2, u32x4* __restrict__ pend = nullptr, bool defer = false) {
    float rstd[8], nmr[8];
    int r0 = 128 * wm + c;
    // opaque to the optimiser: inside the persistent kernel's tile loop the 16 lane-constant store offsets derived from r0
    // would otherwise be hoisted out of the loop and, with no register to spare, kept in scratch (measured: +48 % time)
    asm volatile("" : "+v"(r0));
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const f32x2 cf = *reinterpret_cast<const f32x2*>(side + (r0 + 16 * mi) * 8);
        rstd[mi] = cf[0];
        nmr[mi] = cf[1];
    }
    if (PSEL <= 0 && aux2 && n0 == 0 && wn == 0 && g == 0) {          // the one wave column that clears the other statistics buffer
        keds_stat_t* zero = reinterpret_cast<keds_stat_t*>(aux2) + 2 * (size_t)(m0 + r0);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) keds_stat_zero(zero + 32 * mi);
    }
    char* tile_out = reinterpret_cast<char*>(out) + ((size_t)m0 * N + n0) * 2;          // wave-uniform
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (PSEL >= 0 && p != PSEL) continue;
        const int nl = 64 * wn + 32 * p + 8 * g;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(side + 2048 + nl * 4), b1 = *reinterpret_cast<const f32x4*>(side + 2048 + nl * 4 + 16);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(side + 3072 + nl * 4), c1 = *reinterpret_cast<const f32x4*>(side + 3072 + nl * 4 + 16);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            f32x4 v0 = acc[2 * p][mi] * rstd[mi] + (c0 * nmr[mi] + b0);
            f32x4 v1 = acc[2 * p + 1][mi] * rstd[mi] + (c1 * nmr[mi] + b1);
            if constexpr (epi_base(EPI) == KEDS_EPI_BIAS_QGELU_BF16) {
                // qgelu() on whole vectors: the same operations in the same order, but the scale, the + 1 and the final product are
                // packed (v_pk_mul_f32 / v_pk_add_f32: two elements per issue slot) -- 6 of the 28 issue cycles per element
                f32x4 z0 = v0 * -2.4554669595930157f, z1 = v1 * -2.4554669595930157f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z0[j] = __builtin_amdgcn_exp2f(z0[j]);
                    z1[j] = __builtin_amdgcn_exp2f(z1[j]);
                }
                z0 = z0 + 1.0f;
                z1 = z1 + 1.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z0[j] = __builtin_amdgcn_rcpf(z0[j]);
                    z1[j] = __builtin_amdgcn_rcpf(z1[j]);
                }
                v0 = v0 * z0;
                v1 = v1 * z1;
            }
            if constexpr (DBG == 3) {
                const f32x4 v = v0 + v1;
                if (v[0] + v[1] + v[2] + v[3] == 12345.678f) reinterpret_cast<float*>(out)[0] = v[0];
                continue;
            }
            const unsigned off = ((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)nl) * 2u;
            const bf16x8 ov = bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                                     (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
            // non-temporal: the tile is read next by another kernel, after 200+ MB of other traffic; keeping it out of the way
            // of the operand panels in L2 is worth 0.55 ms of the 21.7 ms step (same-box A/B, round 2, tools/ab_nt.sh).  The same
            // hint on the A-panel DMA costs 2.5 ms (the four tiles of an XCD that share a panel stop sharing it), on the
            // fp16 residual stores it is neutral, on the attention kernel's 8-byte output stores it costs 2.2 ms, and on the bf16
            // output of the MXFP8 kernel (gemm_fp8.hip) it costs 0.2 ms of that mode's 17.2 ms step.
            if constexpr (ND > 0) {
                if (16 * H + p * 8 + mi >= 32 - ND && defer) {              // (wave-uniform)
                    pend[16 * H + p * 8 + mi - (32 - ND)] = __builtin_bit_cast(u32x4, ov);
                    continue;
                }
            }
            keds_store16<KEDS_ST_LN>(ov, tile_out, off);
        }
    }
}
// stores [i0, i1) of the ND deferred ones: the same addresses the epilogue would have used (tile base kept by the caller)
template <int ND>
__device__ __forceinline__ void quad_flush_pending(const u32x4* __restrict__ pend, char* __restrict__ tile_out, int N, int wm, int wn2,
                                                   int g, int c, int i0, int i1) {
    int r0 = 128 * wm + c;
    asm volatile("" : "+v"(r0));
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        if (i < i0 || i >= i1) continue;
        const int s_ = 32 - ND + i, h = s_ >> 4, p = (s_ >> 3) & 1, mi = s_ & 7;
        const unsigned off = ((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)(64 * (2 * wn2 + h) + 32 * p + 8 * g)) * 2u;
        keds_store16<KEDS_ST_LN>(pend[i], tile_out, off);
    }
}

// KEDS_EPI_RESID_STATS_F16 in the 256^2 kernel: the same simplifications (every row valid, uniform tile base + 32-bit lane
// offsets for the read-modify-write of the fp16 stream, bias slice from the LDS side area), all 16 loads of the lane issued
// before the first is used.
// REDUCE = false (4-wave kernel: a wave runs this once per 64-column half): leave the partial sums in `red`, the caller adds them
template <int DBG = 0, bool REDUCE = true>   // DBG, stamped diagnostic build only: 3 = no stores, 4 = no statistics atomics, 5 = neither (and no loads)
__device__ __forceinline__ void pair_resid_epilogue(f32x4 (&acc)[4][8], const char* __restrict__ side, void* __restrict__ out,
                                                    int m0, int n0, int N, int wm, int wn, int g, int c,
                                                    keds_stat_t* __restrict__ stats, char* __restrict__ red) {
    // `red`: 8 KiB of LDS nobody reads any more (the operand buffer of the last but one K-tile): the four waves that share
    // a row (wn = 0..3) leave their {sum, sum sq} of it there, and after one barrier threads 0-255 add ONE statistics pair
    // per row of the tile -- a quarter of the 64-bit atomics (they cost 5-7 k of this epilogue's ~21 k cycles).
    const int r0 = 128 * wm + c;
    char* tile = reinterpret_cast<char*>(out) + ((size_t)m0 * N + n0) * 2;               // wave-uniform
    const int nl = 64 * wn + 8 * g;
    f16x8 r[2][8];
#if KEDS_LD_RESID_AUX
    const auto trs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7FFFFFFF, 0x00020000);
#endif
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#if KEDS_LD_RESID_AUX
            r[p][mi] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                           trs, (int)(((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)(nl + 32 * p)) * 2u), 0, KEDS_LD_RESID_AUX));
#else
            r[p][mi] = DBG == 5 ? f16x8{0, 0, 0, 0, 0, 0, 0, 0}
                                : *reinterpret_cast<const f16x8*>(tile + ((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)(nl + 32 * p)) * 2u);
#endif
        }
    f32x4 b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        b[p][0] = *reinterpret_cast<const f32x4*>(side + 2048 + (nl + 32 * p) * 4);
        b[p][1] = *reinterpret_cast<const f32x4*>(side + 2048 + (nl + 32 * p) * 4 + 16);
    }
    f32x2* rw = reinterpret_cast<f32x2*>(red) + wn * 256 + r0;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        // {sum, sum of squares} of the lane's 16 values of the row: four vector partial sums first (packed adds / FMAs, two
        // elements per issue slot), one horizontal sum at the end -- the per-chunk horizontal sums were 448 scalar adds per tile
        f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, qv = sv;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f16x8 q = r[p][mi];
            const f32x4 v0 = f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]} + (acc[2 * p][mi] + b[p][0]);
            const f32x4 v1 = f32x4{(float)q[4], (float)q[5], (float)q[6], (float)q[7]} + (acc[2 * p + 1][mi] + b[p][1]);
            if constexpr (DBG == 3 || DBG == 5) {
                const f32x4 v = v0 + v1;
                if (v[0] + v[1] + v[2] + v[3] == 12345.678f) reinterpret_cast<float*>(out)[0] = v[0];
            } else {
                const f16x8 ov = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3], (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
                // (a non-temporal store here is neutral: the stream is re-read by the very next GEMM)
                keds_store16<KEDS_ST_RESID>(ov, tile, ((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)(nl + 32 * p)) * 2u);
            }
            sv = sv + (v0 + v1);
            qv = qv + (v0 * v0 + v1 * v1);
        }
        float s = (sv[0] + sv[1]) + (sv[2] + sv[3]), ss = (qv[0] + qv[1]) + (qv[2] + qv[3]);
        s = rows_sum(s);                 // the four lanes (g = 0..3) that share the row hold this wave's 64 columns of it
        ss = rows_sum(ss);
        if constexpr (DBG == 4 || DBG == 5) {
            if (s + ss == 12345.678f) reinterpret_cast<float*>(out)[1] = s;
        } else if (stats && g == 0)
            rw[16 * mi] = f32x2{s, ss};
    }
    if constexpr (DBG != 4 && DBG != 5 && REDUCE) {
        if (stats) {                                                    // kernel-uniform
            __syncthreads();
            const int t = threadIdx.x;
            if (t < 256) {
                const f32x2* rr = reinterpret_cast<const f32x2*>(red) + t;
                const f32x2 a = rr[0], b = rr[256], c2 = rr[512], d = rr[768];
                keds_stat_add(stats + 2 * (size_t)(m0 + t), (a[0] + b[0]) + (c2[0] + d[0]), (a[1] + b[1]) + (c2[1] + d[1]));
            }
        }
    }
}

// KEDS_EPI_RESID_STATS_F16 with the residual tile + bias fed in as the accumulators' INITIAL value (the kernel's prologue
// loads the lane's 16 residual chunks ahead of the DMA pieces in the in-order vmcnt queue and converts them while K-tile 0
// is in flight): the accumulator already holds x + b + sum, so the epilogue is round + store + statistics, with no load
// in it (the 16 dependent residual loads per lane were ~9 k of this epilogue's 14.8 k cycles, round-2 stamps).
__device__ __forceinline__ void pair_resid_epilogue_acc(f32x4 (&acc)[4][8], void* __restrict__ out, int m0, int n0, int N,
                                                        int wm, int wn, int g, int c, keds_stat_t* __restrict__ stats,
                                                        char* __restrict__ red) {
    const int r0 = 128 * wm + c;
    char* tile = reinterpret_cast<char*>(out) + ((size_t)m0 * N + n0) * 2;               // wave-uniform
    const int nl = 64 * wn + 8 * g;
    f32x2* rw = reinterpret_cast<f32x2*>(red) + wn * 256 + r0;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 v0 = acc[2 * p][mi], v1 = acc[2 * p + 1][mi];
            const f16x8 ov = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3], (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
            *reinterpret_cast<f16x8*>(tile + ((unsigned)(r0 + 16 * mi) * (unsigned)N + (unsigned)(nl + 32 * p)) * 2u) = ov;
            s += sum8(v0, v1);
            ss += sum8(v0 * v0, v1 * v1);
        }
        s = rows_sum(s);                 // the four lanes (g = 0..3) that share the row hold this wave's 64 columns of it
        ss = rows_sum(ss);
        if (stats && g == 0) rw[16 * mi] = f32x2{s, ss};
    }
    if (stats) {                                                        // kernel-uniform
        __syncthreads();
        const int t = threadIdx.x;
        if (t < 256) {
            const f32x2* rr = reinterpret_cast<const f32x2*>(red) + t;
            const f32x2 a = rr[0], b = rr[256], c2 = rr[512], d = rr[768];
            keds_stat_add(stats + 2 * (size_t)(m0 + t), (a[0] + b[0]) + (c2[0] + d[0]), (a[1] + b[1]) + (c2[1] + d[1]));
        }
    }
}

template <int N>
__device__ __forceinline__ void small_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int MAXAHEAD>
__device__ __forceinline__ void small_wait_stage(int ahead) {   // `ahead` stages (8 loads each) stay in flight
    if constexpr (MAXAHEAD == 0) {
        small_wait_barrier<0>();
    } else {
        if (ahead >= MAXAHEAD) small_wait_barrier<MAXAHEAD * 8>();
        else small_wait_stage<MAXAHEAD - 1>(ahead);
    }
}

// NST = LDS ring depth: 2 (64 KiB, two workgroups per CU: throughput regime, many tiles) or 4 (128 KiB, counted
// vmcnt keeps 2 K-tiles in flight: latency regime, a handful of workgroups such as remainder rows / M <= 128)
// Split-K (part != nullptr): workgroup (tile, ks) accumulates K range [ks*k_len, (ks+1)*k_len) and stores its raw fp32
// tile to part[ks][m][n]; gemm_splitk_reduce_kernel sums the slices and applies the epilogue.  Used when a launch has
// too few tiles to fill the chip (remainder rows, M <= 256): one CU streams only ~60 GB/s through LDS-DMA.
template <int EPI, int NST>
__global__ __launch_bounds__(256, 2) void gemm_bt_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ bias, void* __restrict__ out,
                                                         int M, int N, int K, int n_tiles,
                                                         const float* __restrict__ aux, int aux_i,
                                                         float* __restrict__ part, int k_len, int tiles, int m_pad,
                                                         long long lda, long long ldc, void* __restrict__ aux2,
                                                         int* __restrict__ guard, long long a_plane, long long w_plane) {
    // (a_plane / w_plane: KEDS_EPI_X3_* only -- elements between the hi and lo planes of A / W; K-tile p of 3 K / 64 reads
    // segment p / (K / 64): hi.hi, hi.lo, lo.hi; x3_seg, gemm_shared.h)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = part ? bid_all / tiles : 0;
    const int bid = part ? bid_all - ks * tiles : bid_all;
    const int k_begin = ks * k_len;
    const int tm = bid / n_tiles, tn = bid - tm * n_tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging addresses: wave w stages LDS rows [32w, 32w+32) of both tiles, 8 rows per DMA
    // (LDS-DMA in its buffer form, round 4: uniform tile bases in descriptors + 32-bit lane offsets.  With the FLAT-encoded
    // global_load_lds inside the K-loop the compiler's wait-count pass treated every LDS wait of the loop as out of order -- all of
    // them `lgkmcnt(0)`, each MFMA group waiting for every fragment read in flight; cf. search.hip / gemm_fp8.hip.)
    const int srow = lane >> 3, sslot = lane & 7;
    auto xrs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(X + (size_t)m0 * lda)), 0, 0x7FFFFFFF, 0x00020000);
    auto wrs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(W + (size_t)n0 * K)), 0, 0x7FFFFFFF, 0x00020000);
    int xsrc[4], wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = 32 * wave + 8 * i + srow;
        const int ch = sslot ^ swz_f(R);
        const int xr = m0 + R < M ? R : M - 1 - m0;      // rows past M re-read row M-1 (results discarded): A needs no padding
        xsrc[i] = (int)((long long)xr * lda * 2) + ch * 16;
        wsrc[i] = perm_w(R) * K * 2 + ch * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* xb = smem + buf * BUF_BYTES + (32 * wave) * 128;
        char* wb = xb + TILE_BYTES;
        unsigned koff = (k_begin + kt * BK) * 2, kx = 0, kw = 0;
        if constexpr (epi_x3(EPI)) {
            const X3Seg sg = x3_seg(kt, K / BK);
            koff = sg.koff;
            kx = sg.a_lo ? (unsigned)(a_plane * 2) : 0u;
            kw = sg.w_lo ? (unsigned)(w_plane * 2) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs_, (__attribute__((address_space(3))) void*)(xb + i * 1024), 16, xsrc[i], koff + kx, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs_, (__attribute__((address_space(3))) void*)(wb + i * 1024), 16, wsrc[i], koff + kw, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a tile), per kk = 0,1
    int xoff[2], woff[2];
    {
        const int xr = 64 * wm + c, wr = 64 * wn + c;   // + 16*mi / 16*ni: same swizzle (f depends on (row>>1)&7)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            xoff[kk] = xr * 128 + (((4 * kk + g) ^ swz_f(xr)) << 4);
            woff[kk] = TILE_BYTES + wr * 128 + (((4 * kk + g) ^ swz_f(wr)) << 4);
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline (same scheme as the 256^2 kernel): fragments are register double-buffered; the 8 reads
    // of K-step s+1 are issued between the 16 MFMAs of K-step s.  K-tile t occupies ring slot t % NST; it is
    // fully read into registers by the end of its kk=0 step, so at the kk=1 step (after one counted wait +
    // barrier that also retires K-tile t+1) its slot is refilled with K-tile t+NST.
    const int nk = (epi_x3(EPI) ? 3 : 1) * ((part ? k_len : K) / BK);
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < nk) stage(s, s);
    {
        int ahead = nk - 1;
        if (ahead > NST - 1) ahead = NST - 1;
        small_wait_stage<NST - 1>(ahead);                         // K-tile 0 landed
    }
    bf16x8 xa[4], wa[4], xb[4], wb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xa[i] = *reinterpret_cast<const bf16x8*>(smem + xoff[0] + i * 2048);
        wa[i] = *reinterpret_cast<const bf16x8*>(smem + woff[0] + i * 2048);
    }
#define KEDS_SMALL_MFMA(xc, wc, xn, wn_, nb, kkn, PREFETCH)                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                     \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                   \
                acc[ni][mi] = mma<epi_f16(EPI)>(wc[ni], xc[mi], acc[ni][mi]);   \
            if (PREFETCH) {                                                                                    \
                xn[mi] = *reinterpret_cast<const bf16x8*>((nb) + xoff[kkn] + mi * 2048);                       \
                wn_[mi] = *reinterpret_cast<const bf16x8*>((nb) + woff[kkn] + mi * 2048);                      \
            }                                                                                                  \
        }                                                                                                      \
        if (PREFETCH) {                                                                                        \
            KEDS_SM_G KEDS_SM_G KEDS_SM_G KEDS_SM_G                                                            \
        }                                                                                                      \
    }
#define KEDS_SM_G                                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* cb = smem + slot * BUF_BYTES;
        int nslot = slot + 1 == NST ? 0 : slot + 1;
        const char* ob = smem + nslot * BUF_BYTES;
        // K-step (kt, 0): prefetch (kt, 1) from the same slot; no synchronisation needed
        __builtin_amdgcn_sched_barrier(0);
        KEDS_SMALL_MFMA(xa, wa, xb, wb, cb, 1, true)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            // K-step (kt, 1): retire K-tile kt+1 (younger tiles stay in flight), free slot of kt, refill it
            int ahead = nk - 2 - kt;
            if (ahead > NST - 2) ahead = NST - 2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            small_wait_stage<NST - 2>(ahead);
            if (kt + NST < nk) stage(slot, kt + NST);
            __builtin_amdgcn_sched_barrier(0);
            KEDS_SMALL_MFMA(xb, wb, xa, wa, ob, 0, true)
        } else {
            KEDS_SMALL_MFMA(xb, wb, xa, wa, ob, 0, false)
        }
        slot = nslot;
    }
#undef KEDS_SMALL_MFMA
#undef KEDS_SM_G

    // ---- epilogue: lane (g,c) owns rows m = m0 + 64*wm + 16*mi + c, columns n0 + 64*wn + 32*p + 8*g + 0..7
    if (part) {   // split-K slice: raw fp32 accumulators
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n0 + 64 * wn + 32 * p + 8 * g;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + 64 * wm + 16 * mi + c;
                float* o = part + ((size_t)ks * m_pad + m) * N + n;
                *reinterpret_cast<f32x4*>(o) = acc[2 * p][mi];
                *reinterpret_cast<f32x4*>(o + 4) = acc[2 * p + 1][mi];
            }
        }
        return;
    }
    // zero_lane: LN epilogues: the one wave column that clears the other stats buffer; RESID_STATS: the lane that adds
    const bool zl = epi_is_ln(EPI) ? (n0 == 0 && wn == 0 && g == 0) : (g == 0);
    tile_epilogue<EPI, 4>(acc, bias, out, m0 + 64 * wm + c, M, n0 + 64 * wn + 8 * g, N, K, aux, aux_i, aux2, ldc, zl, guard,
                          epi_x3(EPI) ? x3_wscale(w_plane) : 1.0f);
}

// split-K reduce: add this thread's partial {sum, sum sq} of a row to its statistics.  Lanes that are known to sit in
// one row (per_row % 64 == 0: the wave, % 32: its halves) add once; per-thread atomics on one address serialise
// (measured 56 us for 128 x 1024 outputs).  Every lane of the wave must call this.
__device__ __forceinline__ void row_stats_add(keds_stat_t* row, float s, float ss, int per_row) {
    const int seg = (per_row & 63) == 0 ? 64 : (per_row & 31) == 0 ? 32 : 1;
    if (seg > 1) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        if (seg == 64) {
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 32, 64);
        }
    }
    if ((threadIdx.x & (seg - 1)) == 0) keds_stat_add(row, s, ss);
}

// sum the split-K slices, add bias, apply the epilogue; one thread per 8 consecutive outputs of one row
template <int EPI>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int splits, int m_pad,
                                                                 const float* __restrict__ bias, void* __restrict__ out,
                                                                 int M, int N, int K, const float* __restrict__ aux, int aux_i,
                                                                 long long ldc, void* __restrict__ aux2,
                                                                 int* __restrict__ guard) {
    const int per_row = N >> 3;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= M * per_row) return;
    const int m = id / per_row, n = (id - m * per_row) << 3;
    f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
    for (int s = 0; s < splits; ++s) {
        const float* p = part + ((size_t)s * m_pad + m) * N + n;
        v0 += *reinterpret_cast<const f32x4*>(p);
        v1 += *reinterpret_cast<const f32x4*>(p + 4);
    }
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (bias) {
        b0 = *reinterpret_cast<const f32x4*>(bias + n);
        b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
    }
    if constexpr (epi_is_ln(EPI)) {
        float rstd, nmr;
        ln_row_coeff(aux, m, 1.0f / (float)K, rstd, nmr, n == 0 ? guard : nullptr);
        keds_stat_t* zero = reinterpret_cast<keds_stat_t*>(aux2);
        if (zero && n == 0) keds_stat_zero(zero + 2 * (size_t)m);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bias + N + n), c1 = *reinterpret_cast<const f32x4*>(bias + N + n + 4);
        epilogue_store<epi_base(EPI)>(v0 * rstd + (c0 * nmr + b0), v1 * rstd + (c1 * nmr + b1), out, m, n, N, nullptr, 0, ldc);
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F32) {
        float* o = reinterpret_cast<float*>(out) + (size_t)m * ldc + n;
        v0 += *reinterpret_cast<const f32x4*>(o) + b0;
        v1 += *reinterpret_cast<const f32x4*>(o + 4) + b1;
        *reinterpret_cast<f32x4*>(o) = v0;
        *reinterpret_cast<f32x4*>(o + 4) = v1;
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(aux2) + (size_t)m * N + n) =
            bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                   (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        row_stats_add(reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux)) + 2 * (size_t)m, sum8(v0, v1),
                      sum8(v0 * v0, v1 * v1), per_row);
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
        f16x8* o = reinterpret_cast<f16x8*>(reinterpret_cast<f16_t*>(out) + (size_t)m * ldc + n);
        const f16x8 r = *o;
        v0 += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]} + b0;
        v1 += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]} + b1;
        *o = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3], (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
        keds_stat_t* stats = reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux));
        if (stats) row_stats_add(stats + 2 * (size_t)m, sum8(v0, v1), sum8(v0 * v0, v1 * v1), per_row);
    } else {
        epilogue_store<EPI>(v0 + b0, v1 + b1, out, m, n, N, aux, aux_i, ldc);
    }
}

// (A 256 x 256 x 32 kernel with a 4-stage LDS ring and counted vmcnt lived here in round 1.  Its 64-byte LDS rows make
// every DMA lane group fetch half cache lines, the texture-address path saturates, and the 256 x 256 x 64 kernel below
// replaced it: +16 % DMA rate from full-line fetches.  profiles/r01_gemm_pmc_ring_kernel.txt keeps its counters.)

// (split-operand GEMMs: the planes' strides of the keds_gemm_x3 call in progress on this thread)
thread_local long long g_x3_aplane = 0, g_x3_wplane = 0;

template <int EPI, int NST>
int launch_small_nst(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                     int aux_i, void* aux2, int splits, long long lda, long long ldc, hipStream_t st) {
    if (int rc = keds_func_lds_once((const void*)gemm_bt_kernel<EPI, NST>, NST * BUF_BYTES, "gemm_bt_kernel")) return rc;
    const int m_tiles = (M + BM - 1) / BM, n_tiles = N / BN;
    const int tiles = m_tiles * n_tiles;
    if (splits > 1) {
        float* g_ws = nullptr;
        size_t g_ws_bytes = 0;
        keds_splitk_scratch(&g_ws, &g_ws_bytes);
        const int m_pad = m_tiles * BM;
        KEDS_LAUNCH((gemm_bt_kernel<EPI, NST>), tiles * splits, 256, NST * BUF_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out,
                    M, N, K, n_tiles, aux, aux_i, g_ws, K / splits, tiles, m_pad, lda, ldc, aux2, (int*)nullptr, 0LL, 0LL);
        int rc = keds_check_launch("gemm_bt_kernel(split-K)");
        if (rc) return rc;
        const int threads = M * (N / 8);
        KEDS_LAUNCH((gemm_splitk_reduce_kernel<EPI>), (threads + 255) / 256, 256, 0, st, (const float*)g_ws, splits, m_pad, bias, out, M, N, K,
                    aux, aux_i, ldc, aux2, keds_numerics_guard());
        return keds_check_launch("gemm_splitk_reduce_kernel");
    }
    KEDS_LAUNCH((gemm_bt_kernel<EPI, NST>), tiles, 256, NST * BUF_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K,
                n_tiles, aux, aux_i, (float*)nullptr, 0, tiles, 0, lda, ldc, aux2, keds_numerics_guard(), g_x3_aplane, g_x3_wplane);
    return keds_check_launch("gemm_bt_kernel");
}

int g_no_split = 0;   // test hook
static bool nosplit_env() {     // KEDS_NO_SPLITK=1 in the environment (A/B): no split-K anywhere
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_NO_SPLITK");
        v = e && e[0] == '1';
    }
    return v != 0;
}
// thread-local request of the calling composite (towers.hip): small launches take the 64 KiB kernel form; KEDS_SMALL_NST=2 in the
// environment forces it everywhere (A/B)
thread_local int tl_small_lds = 0;
bool keds_small_lds_scope() {
    static int env = -1;
    if (env < 0) {
        const char* e = keds_exp_env("KEDS_SMALL_NST");
        env = e && e[0] == '2';
    }
    return env || tl_small_lds;
}

template <int EPI>
int launch_small(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                 int aux_i, void* aux2, long long lda, long long ldc, hipStream_t st) {
    const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
    // too few tiles to fill 256 CUs: split K so that ~128+ workgroups stream the weights in parallel
    float* g_ws = nullptr;
    size_t g_ws_bytes = 0;
    if (tiles <= 64 && K >= 2048 && !g_no_split && !nosplit_env() && !epi_x3(EPI)) keds_splitk_scratch(&g_ws, &g_ws_bytes);
    if (g_ws) {   // (tiles <= 64 && K >= 2048; at K = 1024 the second launch costs what the split saves)
        int splits = 1;
        while (splits < 16 && tiles * splits * 2 <= 256 && K % (splits * 2 * BK) == 0 && K / (splits * 2) >= 2 * BK) splits *= 2;
        const size_t need = (size_t)splits * ((M + BM - 1) / BM * BM) * N * sizeof(float);
        if (splits > 1 && need <= g_ws_bytes)
            return keds_small_lds_scope() ? launch_small_nst<EPI, 2>(A, W, bias, out, M, N, K, aux, aux_i, aux2, splits, lda, ldc, st)
                                          : launch_small_nst<EPI, 4>(A, W, bias, out, M, N, K, aux, aux_i, aux2, splits, lda, ldc, st);
    }
    // fewer workgroups than 2 per CU: nothing else hides the DMA latency, so use the deep ring -- unless the launch is meant to run
    // BESIDE another kernel's workgroups (the towers' remainder-row chain beside the attention launch, round 5): the deep ring's
    // 128 KiB of LDS needs an EMPTY CU, the two-deep ring's 64 KiB fits next to one resident attention workgroup (74 KiB)
    // (round 6: the deep ring only while ONE round of it holds the launch.  Its 128 KiB of LDS mean one workgroup per CU, 256 at a
    // time: 324 workgroups -- the packed text tower's c_proj -- ran two rounds, the second a quarter full, where the two-deep form's
    // 512 slots take them in one and the second resident workgroup hides the DMA latency the deep ring was there for)
    if (tiles <= 256 && !keds_small_lds_scope()) return launch_small_nst<EPI, 4>(A, W, bias, out, M, N, K, aux, aux_i, aux2, 1, lda, ldc, st);
    return launch_small_nst<EPI, 2>(A, W, bias, out, M, N, K, aux, aux_i, aux2, 1, lda, ldc, st);
}

namespace pr {
constexpr int TM = 256, TN = 256, TK = 64;
constexpr int OP_BYTES = 256 * 128;             // 32 KiB per operand per K-tile
constexpr int PBUF_BYTES = 2 * OP_BYTES;        // X | W
constexpr int SIDE_OFF = 2 * PBUF_BYTES;        // side area: float2 {rstd, -mean rstd}[256 rows] | bias'[256 cols] | colsum[256 cols]
constexpr int LDS_BYTES = SIDE_OFF + 4096;      // 132 KiB
}  // namespace pr

// NOTE (measured, round 1): a persistent variant of this kernel (one workgroup per CU walking its tiles, next tile's
// first two K-tiles issued during the last two K-steps so they land under the epilogue) was built, passed parity and was
// SLOWER (out-proj main part 59 -> 67 us).  vmcnt retires in order and counts stores on gfx950, so every s_waitcnt that
// guards a later K-tile also waits for the tile's epilogue store burst to drain (256 CUs x 128 KiB at once, ~8 us);
// a freshly dispatched workgroup does not inherit that dependency.  Kept per-tile.
// NOTE (measured, round 1, tools/micro/): what the K-loop is NOT bound by.  (a) DMA latency from beyond L2: with every
// K-tile re-reading K-tile 0 (L2-hot) the kernel is not faster; LDS-DMA itself streams 59 B/clk/CU from L2.  (b) LDS
// read bandwidth: dropping a third of the fragment reads changes nothing.  (c) The nominal matrix peak: a register-only
// MFMA loop sustains 2.06 PFLOP/s on this chip (clock ~2.0 GHz under load), so 1.25-1.3 PF in this loop is ~62 % of the
// practical peak; the vendor BLAS reaches 1.07 / 1.20 / 1.46 PF on the qkv / fc / proj shapes where this kernel does
// 1.08 / 1.11 / 1.21.  A 4-wave variant (2 x 2 waves, 128 x 128 per wave, 256 accumulators) was built three ways:
// inline-asm MFMAs with "+a" operands (clean loop, results WRONG: the compiler does not know the asm reads its sources
// over several cycles and lets the fragment prefetch overwrite them); the MFMA builtin alone (correct, but the register
// allocator moves ~30 accumulators between AGPRs and VGPRs per K-tile: 1.21 vs 1.32 PF at 8192^3); the builtin plus an
// empty `asm volatile("" : "+a"(acc[i][j]))` for every accumulator at each K-step boundary, which pins them in AGPRs
// (clean loop: 128 MFMA, 32 ds_read_b128, 16 DMA, no v_accvgpr traffic; correct).  That last form reaches 1.39 vs 1.29 PF
// at 8192^3 but LOSES on the shapes of this path (qkv -5 %, fc -9 %, out -3 %, proj -1 %): with one wave per SIMD the
// per-tile prologue / epilogue is exposed, and K = 1024 tiles are mostly prologue and epilogue.  Not used.
// NOTE (measured, round 1): where the 8 DMA pieces of a K-tile sit inside their K-step does not matter either.  All eight
// before the first MFMA group, or two per group in the first half of the step (so the last piece has 1.5 K-steps to land
// instead of one), against one piece per group: qkv 190 / 195 / 200 us, out 71 / 73 / 75, fc 264 / 268 / 275, proj
// 229 / 228 / 234 (spread / early pairs / all first).  The K-loop does not wait for the last DMA piece.
// STAMP (diagnostic build of one instantiation, tools/stamp_gemm.py): s_memtime stamps around the prologue, every
// K-tile's wait and barrier, the loop and the epilogue; aux2 then receives 8 counters per wave instead of its usual role.
// RP (KEDS_EPI_RESID_STATS_F16 only): residual tile + bias as the accumulators' initial value (pair_resid_epilogue_acc)
template <int EPI, int STAMP = 0, int RP = 0>
__global__ __launch_bounds__(512, 2) void gemm_bt_pair_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int M, int N, int K, int n_tiles,
                                                              const float* __restrict__ aux, int aux_i,
                                                              void* __restrict__ aux2, int* __restrict__ guard,
                                                              long long a_plane = 0, long long w_plane = 0) {
    // (a_plane / w_plane: KEDS_EPI_X3_* only, as in gemm_bt_kernel)
    using namespace pr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t_entry = 0, vm_wait = 0, bar_wait = 0;
    if constexpr (STAMP) {
        // desynchronisation experiment (stamped build only): every other group of 8 first-round workgroups starts
        // `aux_i` cycles late, so half of the CUs run half a tile out of phase with the other half
        if (aux_i > 0 && blockIdx.x < 256 && (blockIdx.x & 8)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)aux_i) __builtin_amdgcn_s_sleep(8);
        }
        t_entry = __builtin_amdgcn_s_memtime();
    }
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // The 32 workgroups that run together on one XCD take consecutive logical ids.  Map each run of 32 ids to a
    // block of 8 m-tiles x 4 n-tiles (12 distinct operand panels per K-tile in that XCD's L2 instead of up to 18
    // with a plain n-fastest order) when the tile grid allows it.
    int tm, tn;
    const int m_tiles = gridDim.x / n_tiles;
    // groups walk m fastest: the 32-tile groups that follow each other on an XCD share their W panels (the smaller
    // operand), not their A panels: -0.12 ms per step in a same-box A/B (round 2).  Group shapes 16 x 2 (+0.3 ms),
    // 4 x 8 (no change) and 32 x 1 (+1.6 ms) measured against this 8 x 4.
    quad_tile_coords(bid, m_tiles, n_tiles, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging: piece j (8 LDS rows) of an operand; this wave owns pieces wave + 8*i (rows +64*i)
    const int R0 = 8 * wave + (lane >> 3);                        // 0..63
    const int sch = (lane & 7) ^ swz_f(R0);                        // swz_f(R0 + 64 i) == swz_f(R0)
    // uniform tile bases (SGPR pairs) + 32-bit lane offsets: the DMA then takes the `global_load_lds v_off, s[base]` form --
    // one address VGPR per lane instead of a 64-bit pointer, and no 64-bit VALU adds in the K-loop (16 per K-tile before)
    const char* xt = reinterpret_cast<const char*>(X + (size_t)m0 * K);
    const char* wt = reinterpret_cast<const char*>(W + (size_t)n0 * K);
    const unsigned xoff = (unsigned)R0 * (unsigned)K * 2u + sch * 16;
    const unsigned woff = (unsigned)perm_w(R0) * (unsigned)K * 2u + sch * 16;
    const unsigned rstride = 64u * (unsigned)K * 2u;              // 64 rows further down
    // DMA piece q (0..7: X pieces 0..3 then W pieces 0..3) of K-tile p
#ifndef KEDS_PAIR_BUFFER_DMA
#define KEDS_PAIR_BUFFER_DMA 1
#endif
#if KEDS_PAIR_BUFFER_DMA
    // buffer form of the LDS-DMA (round 3): lane offset in ONE loop-invariant VGPR per operand, piece / K-tile offset in an
    // SGPR -- no vector add per piece (8 per wave and K-tile before) on an issue port that the K-loop saturates
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xt), 0, 0x7FFFFFFF, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wt), 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](int p, int q) {
        const int i = q & 3;
        char* dst = smem + (p & 1) * PBUF_BYTES + (q < 4 ? 0 : OP_BYTES) + (wave + 8 * i) * 1024;
        unsigned so = i * rstride + (unsigned)p * (TK * 2);
        if constexpr (epi_x3(EPI)) {                                // K-tile p of 3 K / 64: its segment's planes (uniform arithmetic)
            const X3Seg sg = x3_seg(p, K / TK);
            so = i * rstride + sg.koff + (q < 4 ? (sg.a_lo ? (unsigned)(a_plane * 2) : 0u) : (sg.w_lo ? (unsigned)(w_plane * 2) : 0u));
        }
        if (q < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, xoff, so, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, woff, so, 0, 0);
    };
#else
    auto issue = [&](int p, int q) {
        const int i = q & 3;
        const char* src = q < 4 ? xt + (xoff + i * rstride + (unsigned)p * (TK * 2)) : wt + (woff + i * rstride + (unsigned)p * (TK * 2));
        char* dst = smem + (p & 1) * PBUF_BYTES + (q < 4 ? 0 : OP_BYTES) + (wave + 8 * i) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
#endif
    // ---- fragment offsets inside a buffer for K-step kk (0/1) of the tile
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xrow = (128 * wm + c) * 128;                         // + mi * 2048
    const int wrow = OP_BYTES + (64 * wn + c) * 128;               // + ni * 2048

    f32x4 acc[4][8];
    constexpr bool RESID_IN = RP != 0 && EPI == KEDS_EPI_RESID_STATS_F16 && STAMP == 0;
    if constexpr (!RESID_IN) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // RESID_IN: this lane's 16 residual chunks (rows 128 wm + c + 16 mi, columns 64 wn + 8 g + 32 p + 0..7 of the tile) and
    // its 16 bias values, requested FIRST: vmcnt retires in order, so "everything older than the 16 DMA pieces" below is
    // exactly these, and their latency overlaps the first K-tile's
    [[maybe_unused]] u32x4 rres[2][8];
    [[maybe_unused]] f32x4 rbias[2][2];
    if constexpr (RESID_IN) {
        const char* rtile = reinterpret_cast<const char*>(out) + ((size_t)m0 * N + n0) * 2;     // wave-uniform
        const unsigned roff = ((unsigned)(128 * wm + c) * (unsigned)N + (unsigned)(64 * wn + 8 * g)) * 2u;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const unsigned o = roff + ((unsigned)(16 * mi) * (unsigned)N + 32u * p) * 2u;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rres[p][mi]) : "v"(o), "s"(rtile) : "memory");
            }
        if (bias) {                                                                           // kernel-uniform
            const float* btile = bias + n0;
            const unsigned boff = (unsigned)(64 * wn + 8 * g) * 4u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rbias[p][0]) : "v"(boff + 128u * p), "s"(btile) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rbias[p][1]) : "v"(boff + 128u * p + 16u), "s"(btile) : "memory");
            }
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) rbias[p][0] = rbias[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    const int np = (epi_x3(EPI) ? 3 : 1) * (K / TK);               // >= 2
    // LN epilogues: one row's statistics (threads 0-255) or one column's bias' / column sum (threads 256-511) per thread,
    // fetched BEFORE the DMA pieces (vmcnt retires in order) and turned into the side-area image while those are in flight
    [[maybe_unused]] u32x4 st_raw = u32x4{0, 0, 0, 0};
    [[maybe_unused]] float pb = 0.f, pc = 0.f;
    // (inline asm + a hand-counted s_waitcnt: for a plain load the compiler waits with vmcnt(0) at the first use, which
    // here would also wait for all 16 DMA pieces)
    if constexpr (epi_is_ln(EPI)) {
        if (tid < 256) {
            const keds_stat_t* sp = reinterpret_cast<const keds_stat_t*>(aux) + 2 * (size_t)(m0 + tid);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st_raw) : "v"(sp) : "memory");
        } else {
            const float* bp = bias + n0 + tid - 256;
            const float* cp = bp + N;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(pc) : "v"(cp) : "memory");
        }
    }
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16 && !RESID_IN) {
        if (tid >= 256 && bias) {
            const float* bp = bias + n0 + tid - 256;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
        }
    }
    // prologue: K-tiles 0 and 1 in flight, retire tile 0
#pragma unroll
    for (int q = 0; q < 8; ++q) issue(0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue(1, q);
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16 && !RESID_IN) asm volatile("s_waitcnt vmcnt(16)" : "+v"(pb)::"memory");
    if constexpr (RESID_IN) {
        // everything older than the 16 DMA pieces has landed: x + b becomes the accumulators' initial value while K-tile 0
        // (requested above) is still in flight
        asm volatile("s_waitcnt vmcnt(16)"
                     : "+v"(rres[0][0]), "+v"(rres[0][1]), "+v"(rres[0][2]), "+v"(rres[0][3]), "+v"(rres[0][4]), "+v"(rres[0][5]),
                       "+v"(rres[0][6]), "+v"(rres[0][7]), "+v"(rres[1][0]), "+v"(rres[1][1]), "+v"(rres[1][2]), "+v"(rres[1][3]),
                       "+v"(rres[1][4]), "+v"(rres[1][5]), "+v"(rres[1][6]), "+v"(rres[1][7]), "+v"(rbias[0][0]), "+v"(rbias[0][1]),
                       "+v"(rbias[1][0]), "+v"(rbias[1][1])
                     :
                     : "memory");
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const f16x8 q = __builtin_bit_cast(f16x8, rres[p][mi]);
                acc[2 * p][mi] = f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]} + rbias[p][0];
                acc[2 * p + 1][mi] = f32x4{(float)q[4], (float)q[5], (float)q[6], (float)q[7]} + rbias[p][1];
            }
    }
    if constexpr (epi_is_ln(EPI)) {
        // everything older than the 16 DMA pieces has landed
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(st_raw), "+v"(pb), "+v"(pc)::"memory");
        if (tid < 256) {
            float rs, nm;
            if constexpr (STAMP == 2) {
                rs = 1.0f + 1.0f / (float)K;
                nm = 1.0f / (float)K;
            } else {
                ln_coeff_from((keds_stat_t)(((unsigned long long)st_raw[1] << 32) | st_raw[0]),
                              (keds_stat_t)(((unsigned long long)st_raw[3] << 32) | st_raw[2]), 1.0f / (float)K, rs, nm,
                              n0 == 0 ? guard : nullptr);
            }
            *reinterpret_cast<f32x2*>(smem + SIDE_OFF + tid * 8) = f32x2{rs, nm};
        } else {
            *reinterpret_cast<float*>(smem + SIDE_OFF + 2048 + (tid - 256) * 4) = pb;
            *reinterpret_cast<float*>(smem + SIDE_OFF + 3072 + (tid - 256) * 4) = pc;
        }
    }
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16 && !RESID_IN) {   // bias slice of the tile (zeros without a bias) -> side area
        if (tid >= 256) *reinterpret_cast<float*>(smem + SIDE_OFF + 2048 + (tid - 256) * 4) = pb;
    }
    if constexpr (RESID_IN) __builtin_amdgcn_sched_barrier(0);    // the conversions above stay in front of the wait for K-tile 0
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned long long t_loop0 = 0;
    if constexpr (STAMP) t_loop0 = __builtin_amdgcn_s_memtime();
    bf16x8 xf[8], wa[4], wb[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wa[ni] = *reinterpret_cast<const bf16x8*>(smem + wrow + slot0 + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(smem + xrow + slot0 + mi * 2048);

    // One K-step: 32 MFMAs from (xc, wc); the 12 fragment reads of the NEXT K-step go to (xn, wn_) from
    // buffer `nb` at chunk offset `nslot`; with ISSUE one DMA piece of K-tile `ip` follows each MFMA group.
    // X fragments are refilled IN PLACE: xf[mi] is read by the four MFMAs of group mi only, so the next K-step's xf[mi] is
    // loaded right behind them (a full K-step before its use); W fragments (read by every group) stay double-buffered.
    // 80 fragment registers instead of 96.
#define KEDS_PAIR_STEP(wc, wn_, nb, nslot, SYNC, ISSUE, ip, PREFETCH)                                          \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr (SYNC && STAMP != 0) {                                                                    \
            const unsigned long long ta = __builtin_amdgcn_s_memtime();                                        \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
            const unsigned long long tb = __builtin_amdgcn_s_memtime();                                        \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                    \
            const unsigned long long tc = __builtin_amdgcn_s_memtime();                                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
            vm_wait += tb - ta;                                                                                \
            bar_wait += tc - tb;                                                                               \
        }                                                                                                      \
        if constexpr (SYNC && STAMP == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) {                                                     \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                   \
                acc[ni][mi] = mma<epi_f16(EPI)>(wc[ni], xf[mi], acc[ni][mi]);                                  \
            if constexpr (PREFETCH) {                                                                          \
                if (mi == 0) {                                                                                 \
                    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                           \
                        wn_[ni] = *reinterpret_cast<const bf16x8*>((nb) + wrow + (nslot) + ni * 2048);         \
                }                                                                                              \
                xf[mi] = *reinterpret_cast<const bf16x8*>((nb) + xrow + (nslot) + mi * 2048);                  \
            }                                                                                                  \
            if constexpr (ISSUE) issue((ip), mi);                                                              \
        }                                                                                                      \
        if constexpr (PREFETCH) {                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);                                                 \
            if constexpr (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                            \
            KEDS_PAIR_G(ISSUE) KEDS_PAIR_G(ISSUE) KEDS_PAIR_G(ISSUE) KEDS_PAIR_G(ISSUE) KEDS_PAIR_G(ISSUE)     \
            KEDS_PAIR_G(ISSUE) KEDS_PAIR_G(ISSUE)                                                              \
        }                                                                                                      \
    }
#define KEDS_PAIR_G(ISSUE)                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                         \
    if constexpr (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);

    int p = 0;
    for (; p + 2 < np; ++p) {                                      // steady state: tile p+2 exists
        const char* cb = smem + (p & 1) * PBUF_BYTES;               // buffer of tile p
        const char* ob = smem + ((p + 1) & 1) * PBUF_BYTES;         // buffer of tile p+1
        KEDS_PAIR_STEP(wa, wb, cb, slot1, false, false, 0, true)          // K-step 2p
        KEDS_PAIR_STEP(wb, wa, ob, slot0, true, true, p + 2, true)       // K-step 2p+1
    }
    {                                                              // tile np-2: nothing left to issue
        const char* cb = smem + (p & 1) * PBUF_BYTES;
        const char* ob = smem + ((p + 1) & 1) * PBUF_BYTES;
        KEDS_PAIR_STEP(wa, wb, cb, slot1, false, false, 0, true)
        KEDS_PAIR_STEP(wb, wa, ob, slot0, true, false, 0, true)
        KEDS_PAIR_STEP(wa, wb, ob, slot1, false, false, 0, true)          // tile np-1
        KEDS_PAIR_STEP(wb, wa, ob, slot0, false, false, 0, false)
    }
#undef KEDS_PAIR_STEP
#undef KEDS_PAIR_G

    // ---- epilogue (same ownership pattern as the 128^2 kernel)
    const bool zl = epi_is_ln(EPI) ? (n0 == 0 && wn == 0 && g == 0) : (g == 0);
    if constexpr (STAMP) {
        const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
        if constexpr (epi_is_ln(EPI))
            pair_ln_epilogue<EPI, STAMP>(acc, smem + SIDE_OFF, out, m0, n0, N, wm, wn, g, c, nullptr);
        else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16)
            pair_resid_epilogue<STAMP>(acc, smem + SIDE_OFF, out, m0, n0, N, wm, wn, g, c,
                                       reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux)), smem + (np & 1) * PBUF_BYTES);
        else
            tile_epilogue<EPI, 8, STAMP>(acc, bias, out, m0 + 128 * wm + c, M, n0 + 64 * wn + 8 * g, N, K, aux, aux_i, nullptr, N, zl);
        const unsigned long long t_issued = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(aux2) + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = t_loop0 - t_entry;   // prologue (address set-up, 16 DMA pieces, first K-tile landed)
            o[1] = t_loop1 - t_loop0;   // K-loop
            o[2] = t_issued - t_loop1;  // epilogue until the last store is issued
            o[3] = vm_wait;             // summed over the K-tiles: s_waitcnt vmcnt(0) lgkmcnt(0)
            o[4] = bar_wait;            // summed: s_barrier
            o[5] = t_end - t_issued;    // store drain after the last issue
            o[6] = t_entry;             // raw stamps + where the wave ran: gaps between consecutive workgroups of a CU
            o[7] = t_end;
            unsigned long long* o2 = reinterpret_cast<unsigned long long*>(aux2) + (size_t)gridDim.x * 64 + (size_t)blockIdx.x * 8 + wave;
            *o2 = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
        }
        return;
    }
    if constexpr (epi_is_ln(EPI))
        pair_ln_epilogue<EPI, 0>(acc, smem + SIDE_OFF, out, m0, n0, N, wm, wn, g, c, aux2);
    else if constexpr (RESID_IN)
        pair_resid_epilogue_acc(acc, out, m0, n0, N, wm, wn, g, c, reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux)),
                                smem + (np & 1) * PBUF_BYTES);
    else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16)
        pair_resid_epilogue(acc, smem + SIDE_OFF, out, m0, n0, N, wm, wn, g, c,
                            reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux)), smem + (np & 1) * PBUF_BYTES);
    else
        tile_epilogue<EPI, 8>(acc, bias, out, m0 + 128 * wm + c, M, n0 + 64 * wn + 8 * g, N, K, aux, aux_i, aux2, N, zl, nullptr,
                              epi_x3(EPI) ? x3_wscale(w_plane) : 1.0f);
}

// ---- 256 x 256 x 64 tiles on FOUR waves (round 3) ---------------------------------------------------------------------
// The vendor BLAS is 20 % ahead of the 8-wave kernel above on the c_fc / c_proj shapes with a plain epilogue (same box,
// random data, tools/micro/vendor_gemm.py: fc 1,174 vs 969, proj 1,438 vs 1,199, 8192^3 1,526 vs 1,390 TFLOP/s), and
// rocprofv3 names what it runs: a 256-thread workgroup per 256 x 256 x 64 tile -- one wave per SIMD, 128 x 128 outputs per
// wave, 256 accumulators in AGPRs, 16 + 16 fragment registers per K-step.  Per K-tile that is 32 ds_read_b128 per wave
// (128 KB of LDS reads per workgroup against 192 KB for eight 128 x 64 waves), no second wave competing for the SIMD's
// matrix pipe and issue port, and four instead of eight parties at the barrier.  Both kernels run AT THE BOARD'S POWER CAP
// (rocm-smi, tools/micro/power_clock_probe.py: 1,390-1,400 W for the vendor's kernel and for ours), the vendor's at a
// LOWER shader clock (1.4-1.9 GHz against 2.0-2.25) with more of its cycles in the matrix pipe: at the cap, throughput is
// bought with utilisation (fewer stall / prologue / epilogue cycles run at a lower voltage point), not with clock.
// Same HBM -> LDS image, staging (LDS-DMA, one counted wait + barrier per K-tile), tile walk and epilogues as the 8-wave
// kernel: wave (wm, wn2) owns rows [128 wm, +128) x columns [128 wn2, +128) = the 8-wave kernel's wave columns 2 wn2 and
// 2 wn2 + 1 ("halves" h = 0, 1 below).
// The 256 accumulators are NOT C++ values: left to the register allocator (MFMA builtin, "+a" pins, physical-register
// pins -- all three built) they are renamed around the loop's back edge at a cost of 590-1,000 v_accvgpr moves per K-tile,
// or spilled.  They live in fixed AGPRs that only the literal-register inline asm of gemm_quad_gen.h touches
// (tools/gen_gemm_quad.py): one asm statement per MFMA (fragments come in as ordinary "v" operands, so the compiler still
// places the s_waitcnt lgkmcnt for its own ds_reads; the first K-step of a tile starts the chains from the constant 0),
// read-back per 64-column half for the C++ epilogues.  Fragments are fully double-buffered (a ds_read never targets a
// register an MFMA issued less than a K-step ago reads), and program order IS issue order: a sched_barrier closes every gap.
//
// A K-step is 32 gaps (one behind every pair of MFMAs), each carrying at most ONE memory instruction: a 16x16x32 MFMA holds
// the SIMD's vector issue for 8 of its 16 cycles and issue costs add (MI355X_MICROARCH.md, cycle constants), so a gap with
// two LDS reads and a DMA piece overruns by their sum.  Measured with the stamped ablation builds (tools/stamp_quad.py,
// cycles per K-tile; 2,099 for the MFMAs alone): 16 reads in 16 consecutive gaps +53; a DMA piece wants MORE room -- 16
// pieces in 16 consecutive gaps +400, in every other gap (one per four MFMAs) +140; everything in clusters of two reads
// + one piece per four MFMAs (the first build) +313.  So:
//   a step WITHOUT DMA pieces: its 16 reads in gaps 0-15 (they read the buffer the next step's pieces overwrite: done early,
//     the wait before that step's barrier finds them finished);
//   a step WITH the 16 pieces: piece i in gap 2i, read i in gap 2i+1, the X fragments first and W fragment j (first used by
//     the next step's MFMA 8j) last.                                                      -> 2,230-2,250 cycles per K-tile
// group g = eight MFMAs = gaps 4g..4g+3.  KEDS_QUAD_ORDER 0: X fragment g against the eight W fragments (SrcB constant, SrcA
// cycling); 1: W fragment g against the eight X fragments (SrcA constant over eight MFMAs, the vendor kernel's order)
#ifndef KEDS_QUAD_ORDER
#define KEDS_QUAD_ORDER 1
#endif
// TIMING ONLY (tools/rounds/r05_noepi_bound.sh): KEDS_QUAD_FILL plain + KEDS_QUAD_FILLX transcendental vector instructions in EVERY gap
// of the K-loop -- how much epilogue arithmetic the gaps between the MFMA pairs can carry before the K-tile grows
#ifndef KEDS_QUAD_FILL
#define KEDS_QUAD_FILL 0
#endif
#ifndef KEDS_QUAD_FILLX
#define KEDS_QUAD_FILLX 0
#endif
#if KEDS_QUAD_FILL || KEDS_QUAD_FILLX
#define KEDS_QUAD_FILLER                                                                                       \
    _Pragma("unroll") for (int f_ = 0; f_ < KEDS_QUAD_FILL; ++f_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fill_v[f_ & 3])); \
    _Pragma("unroll") for (int f_ = 0; f_ < KEDS_QUAD_FILLX; ++f_) asm volatile("v_exp_f32 %0, %0" : "+v"(fill_v[4 + (f_ & 1)]));
#else
#define KEDS_QUAD_FILLER
#endif
#define KEDS_QMFMA(FIRST, j, mi, wc, xc)                                                                       \
    if constexpr (FIRST) {                                                                                     \
        if constexpr (epi_f16(EPI)) { KEDS_QUAD_MFMAZ_##j##_##mi("v_mfma_f32_16x16x32_f16", wc[j], xc[mi]) }   \
        else { KEDS_QUAD_MFMAZ_##j##_##mi("v_mfma_f32_16x16x32_bf16", wc[j], xc[mi]) }                         \
    } else {                                                                                                   \
        if constexpr (epi_f16(EPI)) { KEDS_QUAD_MFMA_##j##_##mi("v_mfma_f32_16x16x32_f16", wc[j], xc[mi]) }    \
        else { KEDS_QUAD_MFMA_##j##_##mi("v_mfma_f32_16x16x32_bf16", wc[j], xc[mi]) }                          \
    }
#if KEDS_QUAD_ORDER == 0
#define KEDS_QG2(FIRST, g_, a_, b_, wc, xc) KEDS_QMFMA(FIRST, a_, g_, wc, xc) KEDS_QMFMA(FIRST, b_, g_, wc, xc)
#else
#define KEDS_QG2(FIRST, g_, a_, b_, wc, xc) KEDS_QMFMA(FIRST, g_, a_, wc, xc) KEDS_QMFMA(FIRST, g_, b_, wc, xc)
#endif
// read #r of the next K-step's fragments: X fragments 0-7, then W fragments 0-7
#define KEDS_QRD(r_, xn, wn_, nb, nslot)                                                                       \
    if constexpr ((r_) < 8) xn[(r_) & 7] = *reinterpret_cast<const bf16x8*>((nb) + xrow + (nslot) + ((r_) & 7) * 2048); \
    else wn_[(r_) & 7] = *reinterpret_cast<const bf16x8*>((nb) + wrow + (nslot) + ((r_) & 7) * 2048);
// gap n (0..31) of a step
// (ISSUE: 0 / false = no DMA pieces in this step, 1 / true = all 16, 2 = pieces 0-7 only)
#define KEDS_QGAP(n_, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                                                 \
    if constexpr (ISSUE) {                                                                                     \
        if constexpr (((n_) & 1) == 0) {                                                                       \
            if constexpr (!(DBG & 1) && ((int)(ISSUE) == 1 || (n_) < 16)) issue((ip), (n_) >> 1);              \
        } else if constexpr (PREFETCH && !(DBG & 2)) {                                                         \
            KEDS_QRD((n_) >> 1, xn, wn_, nb, nslot)                                                            \
        }                                                                                                      \
    } else if constexpr (PREFETCH && (n_) < 16 && !(DBG & 2)) {                                                \
        KEDS_QRD(n_, xn, wn_, nb, nslot)                                                                       \
    }                                                                                                          \
    KEDS_QUAD_FILLER                                                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define KEDS_QUAD_GROUP(FIRST, mi, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                            \
    KEDS_QG2(FIRST, mi, 0, 1, wc, xc) KEDS_QGAP(4 * (mi), xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)             \
    KEDS_QG2(FIRST, mi, 2, 3, wc, xc) KEDS_QGAP(4 * (mi) + 1, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)         \
    KEDS_QG2(FIRST, mi, 4, 5, wc, xc) KEDS_QGAP(4 * (mi) + 2, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)         \
    KEDS_QG2(FIRST, mi, 6, 7, wc, xc) KEDS_QGAP(4 * (mi) + 3, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)
// One K-step: 64 MFMAs from (xc, wc); the 16 fragment reads of the NEXT K-step go to (xn, wn_) from buffer `nb` at chunk
// offset `nslot`; with ISSUE the 16 DMA pieces of K-tile `ip` go out; SYNC: K-tile landed + buffer free (wait + barrier;
// SYNC = n >= 2: the n youngest memory operations -- deferred output stores of the previous tile -- may stay in flight).
#define KEDS_QUAD_STEP(FIRST, xc, wc, xn, wn_, nb, nslot, SYNC, ISSUE, ip, PREFETCH)                           \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr ((int)(SYNC) >= 2 && !(DBG & 4))                                                          \
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((int)(SYNC) >= 2 ? (int)(SYNC) : 0) : "memory"); \
        else if constexpr (SYNC && !(DBG & 4)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        KEDS_QUAD_GROUP(FIRST, 0, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 1, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 2, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 3, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 4, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 5, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 6, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
        KEDS_QUAD_GROUP(FIRST, 7, xc, wc, xn, wn_, nb, nslot, ISSUE, ip, PREFETCH)                             \
    }

namespace qd {
constexpr int SIDE0 = pr::SIDE_OFF;                 // two side areas (tile i uses i & 1) ...
constexpr int RED_OFF = pr::SIDE_OFF + 2 * 4096;    // ... and 8 KiB for the statistics pre-reduction of the residual epilogue
constexpr int LDS_BYTES = RED_OFF + 8192;           // 144 KiB
}  // namespace qd

// PERSIST: gridDim.x workgroups (one per CU) walk the tiles id = blockIdx.x, + gridDim.x, ... (the ids the dispatcher would
// have dealt them, so the XCD grouping of the walk is unchanged).  Behind a tile's K-loop and one barrier the workgroup
// requests the NEXT tile's first two K-tiles (and has fetched its row statistics / bias slices during the K-loop), THEN runs
// this tile's epilogue: the 5 k cycles of prologue latency, the store drain and the gap between workgroups lie under the
// 10 k cycles of epilogue.  The 8-wave kernel could not do this (round 1-2: at 256 VGPRs the tile loop spilled); here the
// accumulators are outside the allocator's view and the wave has ~90 VGPRs to spare.
// NOTE (measured, round 3): all workgroups walk equal tiles in lockstep, so the whole chip stores 33 MB at once every ~50 k
// cycles while the matrix pipes idle (the epilogue's 10.7 k cycles are 3.3 k of issue -- 256 accvgpr reads, 256 v_pk_fma,
// 128 cvt_pk, 40 stores -- and the rest store back-pressure).  Starting XCDs 4-7 16 k / 32 k cycles late (8 / 17 us) costs
// only 0-3 / 3-8 us -- but NOT because the bursts stop colliding: a zero-idle form of the same phase shift (XCDs 4-7 split
// their first tile into two half-width units -- W fragments 0-3, i.e. MFMA groups 0-3 of every K-step, on columns
// [0, 64) u [128, 192) in front of their other tiles and, on a W panel moved by 64 columns, the rest behind them; bit-
// identical) gains nothing: qkv 186.5 vs 187.5 us, c_fc 251.4 vs 244.9.  The chip runs these kernels at its power cap; an
// idle half buys the working half clock, and a stall cycle costs little energy, so removing stall cycles -- rather than
// joules -- buys little time.  (Plain vs non-temporal output stores, any mix: no difference in the isolated GEMM either.)
// What the stores do cost (stamped build, epilogue without stores: 5.2 k cycles instead of 10.1-11 k + 1.2 k of drain, the
// launch 184 instead of 209 us) is their back-pressure on the wave that issues them.  Whole 128-byte lines per instruction
// (lanes c and c ^ 8 swap a chunk with two DPP moves per dword; the burst alone drains 27 % faster so:
// tools/micro/store_burst.hip) changes nothing in the kernel (qkv 172.5 vs 173.1 us, c_fc slower), but NOT ISSUING a part
// of them in the epilogue does: the LayerNorm epilogue of this kernel reads the accumulators back a 32-column quarter at a
// time (64 live registers instead of 128), keeps the last 18 of a lane's 32 packed 16-byte results in registers, and the NEXT
// tile's K-loop issues them three at a time behind the K-steps of its first three K-tiles, where the memory pipeline carries
// nothing but the DMA pieces.  vmcnt counts stores and retires in order, so those K-tile waits are counted (vmcnt(6): the
// stores younger than the DMA pieces stay in flight) -- with vmcnt(0) the trickle returns what it saved.  qkv 170 vs 179 us
// (same process, medians of 7 x 20); not for the QuickGELU form (c_fc: its stores already leave under 512 transcendentals
// per lane; deferring costs it 3 us).
template <int EPI, int STAMP = 0, int PERSIST = 0>
__global__ __launch_bounds__(256, 1) void gemm_bt_quad_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int M, int N, int K, int n_tiles,
                                                              const float* __restrict__ aux, int aux_i,
                                                              void* __restrict__ aux2, int* __restrict__ guard, int ntiles,
                                                              long long a_plane = 0, long long w_plane = 0) {
    // (a_plane / w_plane: KEDS_EPI_X3_* only, as in gemm_bt_pair_kernel: K-tile p of 3 K / 64 reads segment p / (K / 64))
    using namespace pr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    [[maybe_unused]] unsigned long long t_entry = 0, t_loop0 = 0;
    // stamped diagnostic builds 2..8 (timing only, results wrong): STAMP - 1 = bit mask of what the steady-state K-loop leaves
    // out -- 1: the DMA pieces, 2: the fragment reads, 4: the per-K-tile wait + barrier
    constexpr int DBG = STAMP > 1 ? STAMP - 1 : 0;
    if constexpr (STAMP) t_entry = __builtin_amdgcn_s_memtime();
    const int m_tiles = ntiles / n_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn2 = wave & 1, wm = wave >> 1;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging: piece j (8 LDS rows = 1 KiB) of an operand; this wave owns pieces wave + 4 i (rows + 32 i), i < 8
    const int R0 = 8 * wave + (lane >> 3);                        // 0..31
    const int sch = (lane & 7) ^ swz_f(R0);                        // swz_f(R0 + 32 i) == swz_f(R0)
    const unsigned xoff = (unsigned)R0 * (unsigned)K * 2u + sch * 16;
    const unsigned woff = (unsigned)perm_w(R0) * (unsigned)K * 2u + sch * 16;   // perm_w(R0 + 32 i) == perm_w(R0) + 32 i
    const unsigned rstride = 32u * (unsigned)K * 2u;              // 32 rows further down
    // LDS-DMA in its buffer form (`buffer_load_dwordx4 v_lane, s[rsrc], s_off offen lds`): the lane part of the address is
    // ONE loop-invariant VGPR per operand, the piece / K-tile part a scalar -- no vector add per piece
    // (a macro, not a lambda: a lambda RETURNING the descriptor type makes hipcc 7.2 drop the kernel's host-side stub without
    // a diagnostic -- every instantiation then links as an undefined symbol)
#if KEDS_QUAD_TIDDMA   // TIMING ONLY (tools/rounds/r05_tid_dma.sh): what the K-loop would cost if a DMA piece needed NO address VGPR -- the
    // descriptor adds the lane id itself (ADD_TID_ENABLE, stride 16: lane l reads base + offset + 16 l, a contiguous 1 KiB: the
    // traffic shape of a tiled operand layout in which a piece is contiguous in memory).  Wrong operands, same bytes.
#define make_rs_tid(base) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(base)), 16, 0x7FFFFFFF, 0x00800000)
#define make_rs_lin(base) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(base)), 0, 0x7FFFFFFF, 0x00020000)
#define make_rs(base) make_rs_lin(base)
#define make_rsx(base) ((KEDS_QUAD_TIDDMA & 1) ? make_rs_tid(base) : make_rs_lin(base))
#define make_rsw(base) ((KEDS_QUAD_TIDDMA & 2) ? make_rs_tid(base) : make_rs_lin(base))
#else
#define make_rs(base) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(base)), 0, 0x7FFFFFFF, 0x00020000)
#define make_rsx(base) make_rs(base)
#define make_rsw(base) make_rs(base)
#endif
    // (xrs / wrs: buffer descriptors of the CURRENT tile's operand panels; re-pointed at the next tile once this tile's last
    // piece has been requested)
    auto xrs = make_rsx(X), wrs = make_rsw(W);
    auto issue = [&](int p, int q) {                               // DMA piece q (0..15: X pieces 0..7, W pieces 0..7) of K-tile p
        const int i = q & 7;
        char* dst = smem + (p & 1) * PBUF_BYTES + (q < 8 ? 0 : OP_BYTES) + (wave + 4 * i) * 1024;
        unsigned so = i * rstride + (unsigned)p * (TK * 2);
        if constexpr (epi_x3(EPI)) {                                // K-tile p of 3 K / 64: its segment's planes (uniform arithmetic)
            const X3Seg sg = x3_seg(p, K / TK);
            so = i * rstride + sg.koff + (q < 8 ? (sg.a_lo ? (unsigned)(a_plane * 2) : 0u) : (sg.w_lo ? (unsigned)(w_plane * 2) : 0u));
        }
#if KEDS_QUAD_TIDDMA
        if (q < 8) {
            if (KEDS_QUAD_TIDDMA & 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, 0, so + (unsigned)wave * 8u * (unsigned)K * 2u, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, xoff, so, 0, 0);
        } else {
            if (KEDS_QUAD_TIDDMA & 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, 0, so + (unsigned)wave * 8u * (unsigned)K * 2u, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, woff, so, 0, 0);
        }
#else
        if (q < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, xoff, so, 0, KEDS_QUAD_AUX_X);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, woff, so, 0, KEDS_QUAD_AUX_W);
#endif
    };
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xrow = (128 * wm + c) * 128;                         // + mi * 2048
    const int wrow = OP_BYTES + (128 * wn2 + c) * 128;             // + j * 2048, j = 4 h + ni
    const int np = (epi_x3(EPI) ? 3 : 1) * (K / TK);               // >= 2
    const int step = PERSIST ? (int)gridDim.x : ntiles;            // (not persistent: one tile per workgroup)
    [[maybe_unused]] float fill_v[6] = {1.f + K, 2.f, 3.f, 4.f, 0.5f, 0.25f};   // (KEDS_QUAD_FILLER)

    // side data of a tile: row t's LayerNorm statistics and column t's bias' / column sum (LN epilogues), column t's bias
    // (residual epilogue), one element per thread, requested with inline-asm loads (the compiler would wait vmcnt(0) for a
    // plain load at its first use, i.e. also for every DMA piece in flight) and turned into the LDS side area later
    [[maybe_unused]] u32x4 st_raw = u32x4{0, 0, 0, 0};
    [[maybe_unused]] float pb = 0.f, pc = 0.f;
    auto side_request = [&](int m0_, int n0_) {
        if constexpr (epi_is_ln(EPI)) {
            const keds_stat_t* sp = reinterpret_cast<const keds_stat_t*>(aux) + 2 * (size_t)(m0_ + tid);
            const float* bp = bias + n0_ + tid;
            const float* cp = bp + N;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st_raw) : "v"(sp) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(pc) : "v"(cp) : "memory");
        }
        if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
            if (bias) {
                const float* bp = bias + n0_ + tid;
                asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
            }
        }
    };
    auto side_write = [&](char* side, int n0_) {                   // (the requested values have landed: caller waited)
        if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) *reinterpret_cast<float*>(side + 2048 + tid * 4) = pb;
        if constexpr (epi_is_ln(EPI)) {
            float rs, nm;
            ln_coeff_from((keds_stat_t)(((unsigned long long)st_raw[1] << 32) | st_raw[0]),
                          (keds_stat_t)(((unsigned long long)st_raw[3] << 32) | st_raw[2]), 1.0f / (float)K, rs, nm,
                          n0_ == 0 ? guard : nullptr);
            *reinterpret_cast<f32x2*>(side + tid * 8) = f32x2{rs, nm};
            *reinterpret_cast<float*>(side + 2048 + tid * 4) = pb;
            *reinterpret_cast<float*>(side + 3072 + tid * 4) = pc;
        }
    };

    int id = blockIdx.x;
    int tm, tn;
    quad_tile_coords(xcd_remap(id, ntiles), m_tiles, n_tiles, tm, tn);
    int m0 = tm * TM, n0 = tn * TN;
    xrs = make_rsx(X + (size_t)m0 * K);
    wrs = make_rsw(W + (size_t)n0 * K);
    side_request(m0, n0);
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(1, q);
    if constexpr (epi_is_ln(EPI)) asm volatile("s_waitcnt vmcnt(32)" : "+v"(st_raw), "+v"(pb), "+v"(pc)::"memory");
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) asm volatile("s_waitcnt vmcnt(32)" : "+v"(pb)::"memory");
    side_write(smem + qd::SIDE0, n0);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // deferred output stores of the previous tile (PERSIST, LayerNorm epilogues): QUAD_ND x 16 bytes per lane
    constexpr int QUAD_ND = quad_nd<EPI>();                          // deferred stores per lane
    constexpr int QUAD_PS = (QUAD_ND + 5) / 6;                        // ... issued per slot (six slots: behind the K-steps of three K-tiles)
    [[maybe_unused]] u32x4 pend[QUAD_ND];
    [[maybe_unused]] char* pend_base = nullptr;
    [[maybe_unused]] bool have_pend = false;
    // (not the QuickGELU form: its epilogue is 2 x 256 transcendentals per lane long and its stores leave under them; deferring
    // 12 of them costs c_fc 3 us -- 242.8 vs 239.8 -- even with the quarter-wise read-back that keeps the registers free)
    constexpr bool DEFER = PERSIST && epi_is_ln(EPI) && epi_base(EPI) != KEDS_EPI_BIAS_QGELU_BF16 && !STAMP;
    for (int it = 0;; ++it) {
        char* side = smem + qd::SIDE0 + (it & 1) * 4096;
        if constexpr (STAMP) t_loop0 = __builtin_amdgcn_s_memtime();
        // next tile of this workgroup (PERSIST)
        const int nid = id + step;
        const bool more = PERSIST && nid < ntiles;
        int nm0 = 0, nn0 = 0;
        if (more) {
            int ntm, ntn;
            quad_tile_coords(xcd_remap(nid, ntiles), m_tiles, n_tiles, ntm, ntn);
            nm0 = ntm * TM;
            nn0 = ntn * TN;
        }
        bf16x8 xa[8], wa[8], xb[8], wb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wa[j] = *reinterpret_cast<const bf16x8*>(smem + wrow + slot0 + j * 2048);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) xa[mi] = *reinterpret_cast<const bf16x8*>(smem + xrow + slot0 + mi * 2048);

        // K-steps (0,0) | [(p,1) (p+1,0)] for p < np-2 | (np-2,1) (np-1,0) (np-1,1)
        KEDS_QUAD_STEP(true, xa, wa, xb, wb, smem, slot1, false, false, 0, true)
        int p = 0;
        if constexpr (DEFER) {
            // the previous tile's deferred stores, QUAD_PS behind every K-step of the first three K-tiles (np >= 8 here).  vmcnt
            // counts stores and retires in order: the K-tile waits of these iterations and of the one behind them leave the
            // 2 QUAD_PS youngest operations -- stores, younger than the DMA pieces the wait is for -- in flight (a store's
            // acknowledgement takes longer than a K-step; waited for with vmcnt(0) the trickle gives back what it saved)
            if (have_pend) {
#define KEDS_QUAD_PEEL(pp, SY)                                                                                          \
    {                                                                                                                   \
        const char* ob = smem + (((pp) + 1) & 1) * PBUF_BYTES;                                                          \
        KEDS_QUAD_STEP(false, xb, wb, xa, wa, ob, slot0, SY, true, (pp) + 2, true)                                      \
        quad_flush_pending<QUAD_ND>(pend, pend_base, N, wm, wn2, g, c, 2 * QUAD_PS * (pp), 2 * QUAD_PS * (pp) + QUAD_PS); \
        KEDS_QUAD_STEP(false, xa, wa, xb, wb, ob, slot1, false, false, 0, true)                                         \
        quad_flush_pending<QUAD_ND>(pend, pend_base, N, wm, wn2, g, c, 2 * QUAD_PS * (pp) + QUAD_PS, 2 * QUAD_PS * (pp) + 2 * QUAD_PS); \
    }
                KEDS_QUAD_PEEL(0, 1)
                KEDS_QUAD_PEEL(1, 2 * QUAD_PS)
                KEDS_QUAD_PEEL(2, 2 * QUAD_PS)
                KEDS_QUAD_PEEL(3, 2 * QUAD_PS)                         // (nothing left to flush: the indices are past QUAD_ND)
#undef KEDS_QUAD_PEEL
                p = 4;
                have_pend = false;
            }
        }
        for (; p + 2 < np; ++p) {                                      // steady state: tile p+2 exists
            const char* ob = smem + ((p + 1) & 1) * PBUF_BYTES;         // buffer of K-tile p+1
            KEDS_QUAD_STEP(false, xb, wb, xa, wa, ob, slot0, true, true, p + 2, true)       // K-step (p, 1)
            KEDS_QUAD_STEP(false, xa, wa, xb, wb, ob, slot1, false, false, 0, true)         // K-step (p+1, 0)
        }
        {
            const char* ob = smem + ((p + 1) & 1) * PBUF_BYTES;
            KEDS_QUAD_STEP(false, xb, wb, xa, wa, ob, slot0, true, false, 0, true)          // K-step (np-2, 1)
            KEDS_QUAD_STEP(false, xa, wa, xb, wb, ob, slot1, false, false, 0, true)         // K-step (np-1, 0)
            KEDS_QUAD_STEP(false, xb, wb, xa, wa, ob, slot0, false, false, 0, false)        // K-step (np-1, 1)
        }

        [[maybe_unused]] unsigned long long t_loop1 = 0;
        if constexpr (STAMP) t_loop1 = __builtin_amdgcn_s_memtime();
        if (more) {
            // every wave has its last fragments in registers: both operand buffers are free for the next tile's K-tiles 0, 1
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the next tile's side data: PLAIN loads (the compiler tracks them, so a register it spills during the epilogue
            // is spilled after its data arrived -- an inline-asm load in flight across the epilogue would not be), used after
            // the epilogue
            if constexpr (epi_is_ln(EPI)) {
                st_raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const keds_stat_t*>(aux) + 2 * (size_t)(nm0 + tid));
                pb = bias[nn0 + tid];
                pc = bias[N + nn0 + tid];
            }
            if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) pb = bias ? bias[nn0 + tid] : 0.f;
            xrs = make_rsx(X + (size_t)nm0 * K);
            wrs = make_rsw(W + (size_t)nn0 * K);
#pragma unroll
            for (int q = 0; q < 16; ++q) issue(0, q);
#pragma unroll
            for (int q = 0; q < 16; ++q) issue(1, q);
        }

        // ---- epilogues: the 8-wave kernel's, once per 64-column half (wave column 2 wn2 + h) read back from its AGPRs
        // (LayerNorm epilogues do not use aux_i: the A/B switch; the trickle needs the four peeled K-tiles + two more)
        [[maybe_unused]] const bool defer_now = more && aux_i != 0 && np >= 8;
        [[maybe_unused]] void* stamp_out = aux2;
        void* aux2e = STAMP ? nullptr : aux2;                             // (stamped build: aux2 carries the stamp buffer)
#if KEDS_QUAD_NOEPI   // TIMING ONLY (-DKEDS_QUAD_NOEPI=1, tools/rounds/r05_noepi_bound.sh): no read-back, no epilogue arithmetic, no stores
        if (more) { id = nid; m0 = nm0; n0 = nn0; side_write(smem + qd::SIDE0 + ((it + 1) & 1) * 4096, n0);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); continue; }
        break;
#endif
        KEDS_QUAD_DRAIN
        f32x4 av[4][8];
        [[maybe_unused]] keds_stat_t* stats = reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux));
        [[maybe_unused]] char* red = smem + qd::RED_OFF;
#define KEDS_QUAD_EPI(h)                                                                                               \
    if constexpr (epi_is_ln(EPI))                                                                                      \
        pair_ln_epilogue<EPI, 0>(av, side, out, m0, n0, N, wm, 2 * wn2 + h, g, c, aux2e);                              \
    else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16)                                                                \
        pair_resid_epilogue<0, false>(av, side, out, m0, n0, N, wm, 2 * wn2 + h, g, c, stats, red);                    \
    else                                                                                                               \
        tile_epilogue<EPI, 8>(av, bias, out, m0 + 128 * wm + c, M, n0 + 64 * (2 * wn2 + h) + 8 * g, N, K, aux, aux_i, aux2e, N, g == 0, nullptr, \
                              epi_x3(EPI) ? x3_wscale(w_plane) : 1.0f);
        if constexpr (DEFER) {
            // one 32-column quarter at a time: 64 read-back registers live instead of 128 leave room for the deferred stores
#define KEDS_QUAD_EPQ(h, p)                                                                                             \
    KEDS_QUAD_READ_Q##h##p(av)                                                                                          \
    pair_ln_epilogue<EPI, 0, QUAD_ND, h, p>(av, side, out, m0, n0, N, wm, 2 * wn2 + h, g, c, aux2e, pend, defer_now);    \
    __builtin_amdgcn_sched_barrier(0);
            KEDS_QUAD_EPQ(0, 0)
            KEDS_QUAD_EPQ(0, 1)
            KEDS_QUAD_EPQ(1, 0)
            KEDS_QUAD_EPQ(1, 1)
#undef KEDS_QUAD_EPQ
        } else {
            KEDS_QUAD_READ_HALF0(av)
            KEDS_QUAD_EPI(0)
            __builtin_amdgcn_sched_barrier(0);
            KEDS_QUAD_READ_HALF1(av)
            KEDS_QUAD_EPI(1)
        }
#undef KEDS_QUAD_EPI
        if constexpr (DEFER) {
            have_pend = defer_now;
            pend_base = reinterpret_cast<char*>(out) + ((size_t)m0 * N + n0) * 2;
        }
        if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
            if (stats) {                                                    // kernel-uniform
                __syncthreads();
                const f32x2* rr = reinterpret_cast<const f32x2*>(red) + tid;
                const f32x2 a = rr[0], b = rr[256], c2 = rr[512], d = rr[768];
                keds_stat_add(stats + 2 * (size_t)(m0 + tid), (a[0] + b[0]) + (c2[0] + d[0]), (a[1] + b[1]) + (c2[1] + d[1]));
            }
        }
        if constexpr (STAMP) {                                           // same record layout as the 8-wave kernel's stamped build
            const unsigned long long t_issued = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(stamp_out) + ((size_t)blockIdx.x * 8 + wave) * 8;
                o[0] = t_loop0 - t_entry;
                o[1] = t_loop1 - t_loop0;
                o[2] = t_issued - t_loop1;
                o[3] = 0;
                o[4] = 0;
                o[5] = t_end - t_issued;
                o[6] = t_entry;
                o[7] = t_end;
                unsigned long long* o2 = reinterpret_cast<unsigned long long*>(stamp_out) + (size_t)ntiles * 64 + (size_t)blockIdx.x * 8 + wave;
                *o2 = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
            }
        }
        if (!more) break;
        // the next tile: its side data and its K-tile 0 are on the way since before the epilogue
        id = nid;
        m0 = nm0;
        n0 = nn0;
        side_write(smem + qd::SIDE0 + ((it + 1) & 1) * 4096, n0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
// ---- the 4-wave kernel with a THREE-deep ring for the A operand (residual epilogue, long K: c_proj) --------------------
// c_proj streams its A operand (the 268 MB MLP hidden matrix) from HBM exactly once; with two LDS buffers a K-tile's pieces are
// requested one K-tile (~1.3 us) before they are needed, less than an HBM round trip under load, and the K-loop waits 9-16 % of
// its time for them (stamps: 2,570-2,780 cycles per K-tile against 2,200 without the pieces; the 8-wave kernel's vm_wait is
// 15-22 % of its loop).  Here LDS holds A in three buffers and W (8 MB for all tiles: L2-resident) in two -- 5 x 32 KiB, all
// 160 KiB -- so A pieces go out TWO K-tiles ahead: step (p, 1) requests W of tile p+2 and A of tile p+3, and the wait of
// step (p+1, 1) is a counted vmcnt(8) that leaves those eight A pieces in flight.  Bias slice and statistics scratch alias the
// W ring after the K-loop.  One tile per workgroup.
// (Measured, round 3: rotating the K walk per row panel -- panel tm starts at K-tile (s * tm) mod np and wraps, the vendor
// kernels' "StaggerU" -- is SLOWER here: 234 us at s = 5, 8, 17 against 218, 223 at s = 32, 219 at s = 1; the workgroups share
// the W panel's K-slices in L2 because they walk K together.)
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bt_quad3_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                               const float* __restrict__ bias, void* __restrict__ out,
                                                               int M, int N, int K, int n_tiles,
                                                               const float* __restrict__ aux, int ntiles) {
    static_assert(EPI == KEDS_EPI_RESID_STATS_F16, "three-deep A ring: residual epilogue only");
    using namespace pr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DBG = 0;
    constexpr int WRING = 3 * OP_BYTES;                               // A buffers at 0, 32, 64 KiB; W buffers at 96, 128 KiB
    const int m_tiles = ntiles / n_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn2 = wave & 1, wm = wave >> 1;
    const int g = lane >> 4, c = lane & 15;
    const int R0 = 8 * wave + (lane >> 3);
    const int sch = (lane & 7) ^ swz_f(R0);
    const unsigned xoff = (unsigned)R0 * (unsigned)K * 2u + sch * 16;
    const unsigned woff = (unsigned)perm_w(R0) * (unsigned)K * 2u + sch * 16;
    const unsigned rstride = 32u * (unsigned)K * 2u;
    int tm, tn;
    quad_tile_coords(xcd_remap(blockIdx.x, ntiles), m_tiles, n_tiles, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    auto xrs = make_rs(X + (size_t)m0 * K);
    auto wrs = make_rs(W + (size_t)n0 * K);
    auto issue_a = [&](int kt, int buf, int i) {                       // A piece i (0..7) of K-tile kt into A buffer buf
        char* dst = smem + buf * OP_BYTES + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, xoff,
                                                 i * rstride + (unsigned)kt * (TK * 2), 0, KEDS_LD_A3_AUX);
    };
    auto issue_w = [&](int kt, int buf, int i) {
        char* dst = smem + WRING + buf * OP_BYTES + (wave + 4 * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, woff,
                                                 i * rstride + (unsigned)kt * (TK * 2), 0, 0);
    };
    // the K-step macros call issue(ip, q): here pieces 0-7 = W of K-tile ip into the W buffer tile p just left, pieces 8-15 = A
    // of K-tile ip + 1 into the A buffer tile p just left
    int wfree = 0, afree = 0;
    auto issue = [&](int ip, int q) {
        if (q < 8) issue_w(ip, wfree, q);
        else issue_a(ip + 1, afree, q - 8);
    };
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xrow = (128 * wm + c) * 128;                              // relative to an A buffer
    const int wrow0 = (128 * wn2 + c) * 128;                            // relative to a W buffer
    const int np = K / TK;                                              // >= 4 (the launcher checks)
    [[maybe_unused]] float fill_v[6] = {1.f + K, 2.f, 3.f, 4.f, 0.5f, 0.25f};   // (KEDS_QUAD_FILLER)
    float pb = 0.f;
    if (bias) {
        const float* bp = bias + n0 + tid;
        asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a(0, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_w(0, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a(1, 1, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_w(1, 1, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a(2, 2, i);
    asm volatile("s_waitcnt vmcnt(40)" : "+v"(pb)::"memory");
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    bf16x8 xa[8], wa[8], xb[8], wb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wa[j] = *reinterpret_cast<const bf16x8*>(smem + WRING + wrow0 + slot0 + j * 2048);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) xa[mi] = *reinterpret_cast<const bf16x8*>(smem + xrow + slot0 + mi * 2048);

    // tile p: A buffer ia, W buffer iw; the macros read X fragments at (nb) + xrow and W fragments at (nb) + wrow, so `wrow`
    // carries the distance from the A buffer in use to the W buffer in use
    int ia = 0, iw = 0, wrow = WRING + wrow0;
    {
        const char* ab = smem;
        KEDS_QUAD_STEP(true, xa, wa, xb, wb, ab, slot1, false, 0, 0, true)                       // K-step (0, 0)
    }
    int p = 0;
    for (; p + 3 < np; ++p) {                                           // steady state: K-tiles p+2 (W) and p+3 (A) exist
        const int ia1 = ia == 2 ? 0 : ia + 1, iw1 = iw ^ 1;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                  // tile p+1 landed, tile p's buffers free
        wfree = iw;
        afree = ia;
        {
            const char* ab = smem + ia1 * OP_BYTES;
            wrow = WRING + (iw1 - ia1) * OP_BYTES + wrow0;
            KEDS_QUAD_STEP(false, xb, wb, xa, wa, ab, slot0, false, 1, p + 2, true)               // K-step (p, 1)
            KEDS_QUAD_STEP(false, xa, wa, xb, wb, ab, slot1, false, 0, 0, true)                   // K-step (p+1, 0)
        }
        ia = ia1;
        iw = iw1;
    }
    {                                                                   // p = np-3: W of the last K-tile is still to request
        const int ia1 = ia == 2 ? 0 : ia + 1, iw1 = iw ^ 1;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        wfree = iw;
        const char* ab = smem + ia1 * OP_BYTES;
        wrow = WRING + (iw1 - ia1) * OP_BYTES + wrow0;
        KEDS_QUAD_STEP(false, xb, wb, xa, wa, ab, slot0, false, 2, p + 2, true)                   // K-step (np-3, 1)
        KEDS_QUAD_STEP(false, xa, wa, xb, wb, ab, slot1, false, 0, 0, true)                       // K-step (np-2, 0)
        ia = ia1;
        iw = iw1;
    }
    {                                                                   // p = np-2: nothing left to request
        const int ia1 = ia == 2 ? 0 : ia + 1, iw1 = iw ^ 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* ab = smem + ia1 * OP_BYTES;
        wrow = WRING + (iw1 - ia1) * OP_BYTES + wrow0;
        KEDS_QUAD_STEP(false, xb, wb, xa, wa, ab, slot0, false, 0, 0, true)                       // K-step (np-2, 1)
        KEDS_QUAD_STEP(false, xa, wa, xb, wb, ab, slot1, false, 0, 0, true)                       // K-step (np-1, 0)
        KEDS_QUAD_STEP(false, xb, wb, xa, wa, ab, slot0, false, 0, 0, false)                      // K-step (np-1, 1)
    }
    // every wave has read its last fragments: the W ring becomes bias slice (side + 2048) and statistics scratch
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if KEDS_QUAD_NOEPI   // TIMING ONLY: see gemm_bt_quad_kernel
    if (np > 0) return;
#endif
    char* side = smem + WRING;
    char* red = smem + WRING + 8192;
    *reinterpret_cast<float*>(side + 2048 + tid * 4) = pb;
    __syncthreads();
    KEDS_QUAD_DRAIN
    f32x4 av[4][8];
    keds_stat_t* stats = reinterpret_cast<keds_stat_t*>(const_cast<float*>(aux));
    KEDS_QUAD_READ_HALF0(av)
    pair_resid_epilogue<0, false>(av, side, out, m0, n0, N, wm, 2 * wn2 + 0, g, c, stats, red);
    __builtin_amdgcn_sched_barrier(0);
    KEDS_QUAD_READ_HALF1(av)
    pair_resid_epilogue<0, false>(av, side, out, m0, n0, N, wm, 2 * wn2 + 1, g, c, stats, red);
    if (stats) {                                                        // kernel-uniform
        __syncthreads();
        const f32x2* rr = reinterpret_cast<const f32x2*>(red) + tid;
        const f32x2 a = rr[0], b = rr[256], c2 = rr[512], d = rr[768];
        keds_stat_add(stats + 2 * (size_t)(m0 + tid), (a[0] + b[0]) + (c2[0] + d[0]), (a[1] + b[1]) + (c2[1] + d[1]));
    }
}

#undef make_rs
#undef KEDS_QUAD_STEP
#undef KEDS_QUAD_GROUP
#undef KEDS_QG2
#undef KEDS_QGAP
#undef KEDS_QRD
#undef KEDS_QMFMA
#undef KEDS_QUAD_FILLER

// NOTE (measured twice in round 1): a PERSISTENT form of this kernel does not pay.  Second attempt, with the fp16 residual
// stream and the LDS-staged LN epilogue in place: one workgroup per CU walks its tiles; before the LAST K-step of a tile
// (all LDS reads complete, one extra barrier) it issues the next tile's first two K-tiles, so the prologue (in-kernel
// stamps: 4-5k of a tile's ~55k cycles, plus ~0.8k between workgroups) lands under that step and the epilogue; the
// epilogue's stores are younger than those DMA pieces in the in-order vmcnt queue, so "K-tile 0 landed" is
// vmcnt(8 + stores), not a drain.  Bit-identical output, but at 256 VGPRs the tile loop has no register to spare: LICM
// hoists the lane-constant store / DMA offsets out of the loop into scratch (+48 % time until they were made opaque with
// empty asm), and what remains (8 K-loop offsets, 3 fragments: ~20 scratch round trips per tile, each a vmcnt(0)) eats the
// gain: qkv 205 vs 199 us, fc 302 vs 270 us.  A third build on top of the in-place X fragments (16 registers freed, 31 scratch
// operations left in the tile loop) was no better: qkv 206 vs 190 us, fc 282 vs 264 us.  Issuing the next tile's 16 DMA pieces
// costs the same ~1-1.5k cycles of vector issue wherever it sits, the extra barrier, the drain before the side-area staging
// and the static tile -> CU assignment take the rest.  Not kept.

// NOTE (measured, round 2, tools/stamp_gemm.py with SHAPE=out / proj): the fp16-residual epilogue cost 21-25 k cycles per
// tile against 7 k for the same epilogue without residual traffic: ~9 k the 16 dependent residual loads per lane, 5-7 k
// the 64-bit statistics atomics (2,048 per tile), stores ~0.  Kept: the four waves that share a row reduce their partial
// {sum, sum sq} through LDS and threads 0-255 add ONE pair per row (512 atomics per tile: epilogue 20.8 k -> 14.8 k cycles).
// Dead ends, each bit-identical and measured with in-kernel stamps: (a) starting half of the CUs half a tile late changes
// no epilogue or prologue time (the store / load bursts are not a lock-step effect); (b) an L2 prefetch of the residual
// tile (4-byte LDS-DMA per lane, one 128-byte line each, dumped into unused LDS): all 1,024 lines in the prologue stall it by
// 10 k cycles; spread over the K-loop (one wave-instruction per K-tile) it costs the K-loop what it saves the epilogue
// (K-tile 2,432 -> 2,572 cycles, or 2,832 with a counted vmcnt that leaves it in flight), and the loads still take ~8 k
// afterwards: 32 tiles x 128 KB is the whole 4 MiB L2 of the XCD, and 6 MB of operands stream through it per tile;
// (c) the same prefetch for the OPERANDS two K-tiles ahead (c_proj's A streams from HBM, 10 % of its K-loop is vmcnt wait):
// 5 % slower on every shape; (d) two half-batches on two streams, so that one's epilogues run beside the other's
// K-loops: 21.6 ms vs 21.1 ms for 128 images (four quarter-batches: 31.8 ms).
// NOTE (measured, round 1): TWO half-size workgroups per CU do not pay either.  256 threads (2 x 2 waves, 128 x 64 per wave),
// tile 256 x 128, one 48 KiB LDS operand buffer per K-tile with all 24 fragments of the K-tile in registers (reads of
// K-tile t -> barrier -> DMA of t+1 into the same buffer under the MFMAs of t), two such workgroups per CU (2 x 52 KiB LDS,
// 8 waves x 246 VGPRs) started half a tile apart so that one's prologue / epilogue / DMA wait runs under the other's
// MFMAs.  Bit-identical output; qkv 201 us vs 178 us, c_fc 271 vs 246 us (scheduled with sched_group_barrier like the
// 256^2 kernel; 15 % behind before that): 50 % more DMA bytes per flop and two barriers per K-tile cost more than the
// hidden fixed costs return.

// NOTE (measured, round 2): TWO tiles per workgroup, straight-line instead of a tile loop (tile A, then tile B = A + tiles/2
// of the same walk order; behind A's K-loop one barrier, B's statistics loads and first two K-tiles requested, THEN A's
// epilogue, so that B's whole prologue and the gap between workgroups lie under it; counted vmcnt(8 + 16) / (16 + 16) for
// "B's K-tile 0 / statistics have landed"; a second side area and a private 8 KiB for the statistics pre-reduction: 144 KiB
// of LDS).  With the lane constants of the DMA and of the fragment reads made opaque between the tiles (otherwise every
// 64-bit piece address of tile A is kept for tile B in scratch) it compiles to 255 VGPRs and no spill, is bit-identical
// -- and changes nothing: 20.54 / 20.58 ms per step against 20.55 / 20.57 (with the spills: +1.2 ms).  Hiding the 4-5 k
// cycle prologue returns no time.  Together with the earlier findings (zero-filled operands run 19 % faster at identical
// cycle counts; the only changes that paid this round removed memory traffic) this says the kernel is bound by what the
// chip can power, not by its schedule: idle cycles removed come back as clock.  Not kept.

int g_skip_tail = 0;      // timing-only: skip the remainder-row launch
static int quad_defer_env() {     // KEDS_QUAD_DEFER=0 in the environment: no deferred epilogue stores (whole-step A/B)
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_QUAD_DEFER");
        v = !(e && e[0] == '0');
    }
    return v;
}
int g_quad_defer = 1;     // persistent 4-wave kernel, LayerNorm epilogue: 12 of a tile's 32 stores per lane wait for the next K-loop (bit 17: off)
int g_quad3 = 1;          // 4-wave kernel, residual epilogue: three-deep A ring (bit 16 of keds_gemm_force_small's argument: off)
// 256^2 tiles on the 4-wave kernel: -1 = by shape (quad_by_shape: persistent form), 0 = never, 1 = always, one tile per
// workgroup, 2 = always, persistent (one workgroup per CU walks the tiles, next tile's first K-tiles under the epilogue)
// (bits 11-12 of keds_gemm_force_small's argument force 1 / 2, 3 = never; KEDS_GEMM_QUAD=0/1/2 in the environment overrides
// the default)
int g_quad = -1;
int quad_env() {
    static int v = -2;
    if (v == -2) {
        const char* e = keds_exp_env("KEDS_GEMM_QUAD");
        v = e && e[0] ? atoi(e) : -1;
    }
    return v;
}
// Same-process A/B on the ViT-L/14 shapes at B = 128 (tools/ab_quad.py, medians of 5 x 20 launches, round 3; 8 waves /
// 4 waves / 4 waves persistent, us): qkv 189.4 / 184.1 / 179.2, c_fc 243.6 / 239.5 / 236.3, c_proj 216.3 / 214.7 / (212.5),
// out-proj 66.3 / 67.7 / -- : the 4-wave kernel wins where the K-loop dominates the tile and its persistent form where the
// LayerNorm epilogues (no loads of their own) leave registers for the tile loop; out-proj (K = 1024, a tile that is mostly
// read-modify-write epilogue, which one wave per SIMD runs with nothing beside it) stays on the 8-wave kernel.
// residual GEMMs go to the 4-wave kernel (three-deep A ring) from this K on (KEDS_RESID_QUAD_K in the environment: A/B)
static int resid_quad_min_k() {
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_RESID_QUAD_K");
        v = e && e[0] ? atoi(e) : 1024;       // round 4: out-proj too (+0.35 % on the headline in four same-box pairs: its A operand,
                                              // the attention output, is cold in the step and the three-deep ring tolerates that)
    }
    return v;
}
// (KEDS_X3_QUAD=0 in the environment: the 8-wave kernel for the split-operand GEMMs, A/B)
static bool x3_quad_env() {
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_X3_QUAD");
        v = !(e && e[0] == '0');
    }
    return v != 0;
}
template <int EPI>
bool quad_by_shape(int N, int K) {
    if constexpr (epi_is_ln(EPI)) return K >= 512;
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) return K >= resid_quad_min_k();
    if constexpr (epi_x3(EPI)) return x3_quad_env();               // split-operand GEMMs: a K-loop of 3 K / 64 K-tiles, the form that wins where the K-loop dominates
    return false;
}
// fp16-residual GEMMs: residual + bias as the accumulators' initial value instead of 16 loads per lane in the epilogue (round 3,
// asked for by the round-2 review).  Built, bit-compatible within the fp32 addition order, and SLOWER in a same-process
// interleaved A/B (tools/ab_resid_prologue.py, 7 rounds x 20 launches, medians): out-proj 75.5 vs 73.0 us, c_proj 232.4 vs
// 229.0 us -- the 20 extra loads per lane in front of the first K-tile's DMA pieces delay the K-loop's start by more than the
// epilogue saves (the loads themselves were never the epilogue's cost: round-2 note above).  Off; bit 10 of
// keds_gemm_force_small's argument turns it on for an A/B.
int g_resid_prologue = 0;
int g_pair_stamp = 0;     // diagnostic: stamped build of the qkv instantiation (aux2 = stamp buffer)

template <int EPI>
int launch_big(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
               int aux_i, void* aux2, hipStream_t st) {
#ifdef KEDS_EXPERIMENTS
    if constexpr (EPI == KEDS_EPI_LN_BIAS_BF16_H || EPI == KEDS_EPI_LN_QGELU_BF16_H) {
        // (a forced kernel form -- keds_gemm_force_small bits 11-15, KEDS_GEMM_QUAD -- keeps the round-4 kernels: A/B tools)
        if (g_quad < 0 && quad_env() < 0 && !g_pair_stamp && keds_gemm_duo_ok(EPI, M, N, K))
            return keds_gemm_duo_launch(EPI, A, W, bias, out, M, N, K, aux, aux2, st);
    }
#endif
    if (int rc = keds_func_lds_once((const void*)gemm_bt_pair_kernel<EPI>, pr::LDS_BYTES, "gemm_bt_pair_kernel")) return rc;
    if constexpr (EPI == KEDS_EPI_LN_BIAS_BF16_H || EPI == KEDS_EPI_RESID_STATS_F16) {
        if (g_pair_stamp && g_quad > 0) {                            // stamped build of the 4-wave kernel
            const int ntiles = (M / pr::TM) * (N / pr::TN);
#define KEDS_QSTAMP(V)                                                                                              \
    {                                                                                                              \
        (void)keds_func_lds_once((const void*)gemm_bt_quad_kernel<EPI, V>, qd::LDS_BYTES, "gemm_bt_quad_kernel<stamp>"); \
        gemm_bt_quad_kernel<EPI, V><<<ntiles, 256, qd::LDS_BYTES, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K, \
                                                                        N / pr::TN, aux, aux_i, aux2, nullptr, ntiles); \
    }
            switch (g_pair_stamp) {
                case 2: KEDS_QSTAMP(2) break;
                case 3: KEDS_QSTAMP(3) break;
                case 4: KEDS_QSTAMP(4) break;
                case 5: KEDS_QSTAMP(5) break;
                case 6: KEDS_QSTAMP(6) break;
                case 7: KEDS_QSTAMP(8) break;
                default: KEDS_QSTAMP(1) break;
            }
#undef KEDS_QSTAMP
            return keds_check_launch("gemm_bt_quad_kernel<stamp>");
        }
        if (g_pair_stamp) {
            const dim3 grid((M / pr::TM) * (N / pr::TN));
#define KEDS_STAMP_LAUNCH(V)                                                                                       \
    {                                                                                                             \
        (void)keds_func_lds_once((const void*)gemm_bt_pair_kernel<EPI, V>, pr::LDS_BYTES, "gemm_bt_pair_kernel<stamp>"); \
        gemm_bt_pair_kernel<EPI, V><<<grid, 512, pr::LDS_BYTES, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K, \
                                                                       N / pr::TN, aux, aux_i, aux2, nullptr);      \
    }
            switch (g_pair_stamp) {
                case 2: KEDS_STAMP_LAUNCH(2) break;
                case 3: KEDS_STAMP_LAUNCH(3) break;
                case 4: KEDS_STAMP_LAUNCH(4) break;
                case 5: KEDS_STAMP_LAUNCH(5) break;
                default: KEDS_STAMP_LAUNCH(1) break;
            }
#undef KEDS_STAMP_LAUNCH
            return keds_check_launch("gemm_bt_pair_kernel<stamp>");
        }
    }
    const int m_tiles = M / pr::TM, n_tiles = N / pr::TN;         // M is a multiple of 256 here
    int quad = g_quad >= 0 ? g_quad : quad_env();
    if (quad < 0) quad = quad_by_shape<EPI>(N, K) ? 2 : 0;
    if (quad) {                                   // 1: one tile per workgroup, 2: persistent (one workgroup per CU walks the tiles)
        const int ntiles = m_tiles * n_tiles;
        int cus = keds_device_cus();
        if (cus > 256) cus = 256;
        cus &= ~7;                                // whole XCD groups: workgroup b and tile ids b, b + grid, ... share an XCD label
        // (the residual epilogue holds 16 residual chunks per lane beside the read-back accumulators: in the tile loop it spills,
        // out-proj 94 vs 70 us -- that epilogue keeps one tile per workgroup.  Late in round 3, with the quarter-wise read-back
        // the tile loop no longer spills (209 VGPRs), and still does not pay: out-proj 66.7 vs 66.3 us, c_proj 216 vs 203.5 on
        // its three-deep ring -- two tiles per workgroup leave one prologue to hide, and the epilogue's residual loads queue
        // behind the 32 DMA pieces of the next tile in the in-order vmcnt)
        if (quad == 2 && ntiles > cus && cus >= 8 && EPI != KEDS_EPI_RESID_STATS_F16 && EPI != KEDS_EPI_X3_RESID_F32) {
            if (int rc = keds_func_lds_once((const void*)gemm_bt_quad_kernel<EPI, 0, 1>, qd::LDS_BYTES, "gemm_bt_quad_kernel")) return rc;
            KEDS_LAUNCH((gemm_bt_quad_kernel<EPI, 0, 1>), cus, 256, qd::LDS_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K,
                        n_tiles, aux, (int)(epi_is_ln(EPI) ? (g_quad_defer && quad_defer_env()) : aux_i), aux2,
                        keds_numerics_guard(), ntiles, g_x3_aplane, g_x3_wplane);
            return keds_check_launch("gemm_bt_quad_kernel<persistent>");
        }
        if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
            if (g_quad3 && K >= 1024 && K / pr::TK >= 4) {               // long K: A operand through a three-deep ring
                if (int rc = keds_func_lds_once((const void*)gemm_bt_quad3_kernel<EPI>, 5 * pr::OP_BYTES, "gemm_bt_quad3_kernel")) return rc;
                KEDS_LAUNCH((gemm_bt_quad3_kernel<EPI>), ntiles, 256, 5 * pr::OP_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K,
                            n_tiles, aux, ntiles);
                return keds_check_launch("gemm_bt_quad3_kernel");
            }
        }
        if (int rc = keds_func_lds_once((const void*)gemm_bt_quad_kernel<EPI>, qd::LDS_BYTES, "gemm_bt_quad_kernel")) return rc;
        KEDS_LAUNCH((gemm_bt_quad_kernel<EPI>), ntiles, 256, qd::LDS_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K, n_tiles,
                    aux, aux_i, aux2, keds_numerics_guard(), ntiles, g_x3_aplane, g_x3_wplane);
        return keds_check_launch("gemm_bt_quad_kernel");
    }
    if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
        if (g_resid_prologue) {
            if (int rc = keds_func_lds_once((const void*)gemm_bt_pair_kernel<EPI, 0, 1>, pr::LDS_BYTES, "gemm_bt_pair_kernel")) return rc;
            KEDS_LAUNCH((gemm_bt_pair_kernel<EPI, 0, 1>), m_tiles * n_tiles, 512, pr::LDS_BYTES, st,
                        (const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K, n_tiles, aux, aux_i, aux2, keds_numerics_guard(), 0LL, 0LL);
            return keds_check_launch("gemm_bt_pair_kernel");
        }
    }
    KEDS_LAUNCH((gemm_bt_pair_kernel<EPI>), m_tiles * n_tiles, 512, pr::LDS_BYTES, st, (const bf16_t*)A, (const bf16_t*)W, bias, out,
                M, N, K, n_tiles, aux, aux_i, aux2, keds_numerics_guard(), g_x3_aplane, g_x3_wplane);
    return keds_check_launch("gemm_bt_pair_kernel");
}

int g_force_small = 0;   // test hook: route everything through the 128^2 kernel

static int big_tiles_pct() {          // KEDS_BIG_TILES_PCT in the environment (A/B): the fill a 256^2 launch needs, default 85
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_BIG_TILES_PCT");
        v = e && e[0] ? atoi(e) : 85;
    }
    return v;
}
// (round 5) A launch whose rows are whole 256-row tiles (no remainder launch behind it) and whose K-loop is short needs only HALF
// of its last round filled: the 128^2 kernel's alternative is four times the workgroups on 512 slots, and 11,008 x 768 x 768 (the
// dual workload's 2B-row text pass at 43 columns: 516 workgroups, four more than fit at once) pays a whole second round for
// them -- 60 us against 30 on 129 tiles of 256^2; in_proj 61 -> 43, c_fc 82 -> 70.  Not for long K (c_proj, K = 3072: 76 us on the
// one-tile-per-workgroup kernel against 60).  profiles/r05_text_big_tiles_ab.txt
bool big_tiles_ok(int M, int N, int K) {
    const long bt = (long)(M / pr::TM) * (N / pr::TN);
    const long rounds = (bt + 255) / 256;
    int pct = big_tiles_pct();
    if (pct == 85 && M % pr::TM == 0 && K <= 1024) pct = 50;
    return !g_force_small && N % pr::TN == 0 && K % 64 == 0 && K >= 128 && bt > 0 && bt * 100 >= rounds * 256 * pct;
}

template <int EPI>
int launch_gemm(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                int aux_i, void* aux2, long long lda, long long ldc, hipStream_t st) {
    KedsProfScope prof(KEDS_PROF_GEMM, st, /*lazy: the launches bind the event pair (KEDS_LAUNCH)*/ true);
    prof.work(2.0 * M * N * K);
    // Large problems: full 256-row tiles go to the 256^2 kernel, the remainder rows (< 256) to the 128^2 one.
    // (ViT-L/14 at B=128: M = 32896 = 128*256 + 128, so 512..2048 big tiles = whole rounds on 256 CUs.)
    // the 256^2 kernel runs one workgroup per CU: use it when its full tiles keep >= 85% of the CU-rounds busy (a single
    // round counts: 19,712 x 768 x 3072 runs at 1.13 PF on 231 tiles vs 0.96 on 924 tiles of 128^2); otherwise the
    // 128^2 kernel's finer tiles quantise better
    const bool big_ok = big_tiles_ok(M, N, K) && lda == K && ldc == N && (EPI != KEDS_EPI_PATCH_F32 || M % pr::TM == 0) &&
                        EPI != KEDS_EPI_BIAS_BF16_HEADF32;      // (its fp32 head rows are numbered from row 0 of the launch)
    if (!big_ok) return launch_small<EPI>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
    const int m_main = M / pr::TM * pr::TM;
    int rc = launch_big<EPI>(A, W, bias, out, m_main, N, K, aux, aux_i, aux2, st);
    if (rc || m_main == M || g_skip_tail) return rc;
    const size_t esz = (EPI == KEDS_EPI_BIAS_RESID_F32 || EPI == KEDS_EPI_BIAS_F32 || EPI == KEDS_EPI_RESID_STATS_F32 ||
                        EPI == KEDS_EPI_X3_BIAS_F32 || EPI == KEDS_EPI_X3_RESID_F32) ? 4 : 2;
    // the remainder launch numbers its rows from 0: move the per-row side buffers along
    const float* aux_t = aux;
    void* aux2_t = aux2;
    if constexpr (epi_is_ln(EPI)) {
        aux_t = (const float*)((const keds_stat_t*)aux + 2 * (size_t)m_main);
        if (aux2) aux2_t = (keds_stat_t*)aux2 + 2 * (size_t)m_main;
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F32) {
        aux_t = (const float*)((const keds_stat_t*)aux + 2 * (size_t)m_main);
        aux2_t = (char*)aux2 + (size_t)m_main * N * 2;
    } else if constexpr (EPI == KEDS_EPI_RESID_STATS_F16) {
        if (aux) aux_t = (const float*)((const keds_stat_t*)aux + 2 * (size_t)m_main);
    }
    return launch_small<EPI>((const char*)A + (size_t)m_main * K * 2, W, bias, (char*)out + (size_t)m_main * N * esz,
                             M - m_main, N, K, aux_t, aux_i, aux2_t, lda, ldc, st);
}

}  // namespace

// true when a dense [M,K] x [N,K]^T problem sends its full 256-row tiles to the 256^2 kernel (and M % 256 rows to a
// second, small launch): the towers then run those remainder rows as their own chain on the side lane
bool keds_gemm_splits_rows(int M, int N, int K) { return big_tiles_ok(M, N, K) && M % pr::TM != 0; }
// (towers.hip) small GEMM launches of the calling thread take the 64 KiB-LDS kernel form while `on`
void keds_gemm_small_lds(int on) { tl_small_lds = on; }

extern "C" int keds_gemm_force_small(int on) {
    g_force_small = on & 1;
    g_no_split = (on >> 9) & 1;         // bit 9: disable split-K (A/B tests)
    g_skip_tail = (on >> 8) & 1;        // bit 8: timing-only, skip remainder rows
    g_quad3 = !((on >> 16) & 1);        // bit 16: no three-deep A ring in the 4-wave residual GEMM (A/B)
    g_quad_defer = !((on >> 17) & 1);   // bit 17: no deferred epilogue stores in the persistent 4-wave kernel (A/B)
    g_quad = (on >> 11) & 3;            // bits 11-12: 256^2 tiles on the 4-wave kernel (1), its persistent form (2), 3 = never
    if (g_quad == 0) g_quad = -1;       // (0 = the default: by shape)
    if (g_quad == 3) g_quad = 0;
    g_resid_prologue = (on >> 10) & 1;  // bit 10: fp16-residual GEMMs take residual + bias as the accumulators' initial value (A/B)
    g_pair_stamp = (on >> 13) & 7;      // bits 13-15: stamped diagnostic build of the qkv / residual GEMMs (2: no statistics loads, 3: no stores, 4: no atomics, 5: no residual traffic at all)
    return KEDS_OK;
}

extern "C" int keds_gemm_bt_ex2(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                                int M, int N, int K, int epilogue, const float* aux, int aux_i, void* aux2, void* stream) {
    KEDS_REQUIRE(A && W && out, "keds_gemm_bt: null pointer");
    KEDS_REQUIRE(M > 0 && N > 0 && K > 0, "keds_gemm_bt: empty problem");
    KEDS_REQUIRE(N % BN == 0, "keds_gemm_bt: N=%d must be a multiple of %d", N, BN);
    KEDS_REQUIRE(K % BK == 0, "keds_gemm_bt: K=%d must be a multiple of %d", K, BK);
    KEDS_REQUIRE(lda >= K && ldc >= N && lda % 8 == 0 && ldc % 8 == 0, "keds_gemm_bt: bad row strides");
    KEDS_REQUIRE(epilogue != KEDS_EPI_PATCH_F32 || ldc == N, "keds_gemm_bt: EPI_PATCH needs a dense output");
    hipStream_t st = (hipStream_t)stream;
#define KEDS_GEMM_CASE(E) case E: return launch_gemm<E>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
    switch (epilogue) {
        KEDS_GEMM_CASE(KEDS_EPI_BIAS_BF16)
        KEDS_GEMM_CASE(KEDS_EPI_BIAS_QGELU_BF16)
        KEDS_GEMM_CASE(KEDS_EPI_BIAS_RELU_BF16)
        KEDS_GEMM_CASE(KEDS_EPI_BIAS_RESID_F32)
        KEDS_GEMM_CASE(KEDS_EPI_BIAS_F32)
        case KEDS_EPI_PATCH_F32:
            KEDS_REQUIRE(aux && aux_i > 0, "keds_gemm_bt: EPI_PATCH needs the positional embedding and G");
            return launch_gemm<KEDS_EPI_PATCH_F32>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
        case KEDS_EPI_LN_BIAS_BF16:
        case KEDS_EPI_LN_QGELU_BF16:
        case KEDS_EPI_LN_BIAS_BF16_H:
        case KEDS_EPI_LN_QGELU_BF16_H:
            KEDS_REQUIRE(bias && aux, "keds_gemm_bt: EPI_LN_* needs bias = [bias' | colsum] and aux = row statistics");
            if (epilogue == KEDS_EPI_LN_BIAS_BF16)
                return launch_gemm<KEDS_EPI_LN_BIAS_BF16>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
            if (epilogue == KEDS_EPI_LN_BIAS_BF16_H)
                return launch_gemm<KEDS_EPI_LN_BIAS_BF16_H>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
            if (epilogue == KEDS_EPI_LN_QGELU_BF16_H)
                return launch_gemm<KEDS_EPI_LN_QGELU_BF16_H>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
            return launch_gemm<KEDS_EPI_LN_QGELU_BF16>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
        case KEDS_EPI_RESID_STATS_F32:
            KEDS_REQUIRE(aux && aux2, "keds_gemm_bt: EPI_RESID_STATS needs aux = statistics and aux2 = bf16 copy");
            return launch_gemm<KEDS_EPI_RESID_STATS_F32>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
        KEDS_GEMM_CASE(KEDS_EPI_RESID_STATS_F16)
        case KEDS_EPI_BIAS_BF16_HEADF32:
            KEDS_REQUIRE(aux && aux_i >= 0, "keds_gemm_bt: EPI_BIAS_BF16_HEADF32 needs the fp32 head buffer and its row count");
            return launch_gemm<KEDS_EPI_BIAS_BF16_HEADF32>(A, W, bias, out, M, N, K, aux, aux_i, aux2, lda, ldc, st);
        case KEDS_EPI_X3_BIAS_F32:
        case KEDS_EPI_X3_RESID_F32:
        case KEDS_EPI_X3_QGELU_PAIR:
            keds_set_error("keds_gemm_bt: split-operand epilogues go through keds_gemm_x3");
            return KEDS_E_ARG;
        default: keds_set_error("keds_gemm_bt: unknown epilogue %d", epilogue); return KEDS_E_ARG;
    }
#undef KEDS_GEMM_CASE
}

extern "C" int keds_gemm_x3(const void* a, int64_t a_plane, int64_t lda, const void* w, int64_t w_plane, const float* bias, void* out,
                            int64_t ldc, int M, int N, int K, int epilogue, int aux_i, int w_exp, void* stream) {
    KEDS_REQUIRE(a && w && out && M > 0 && N > 0 && K > 0, "keds_gemm_x3: bad argument");
    KEDS_REQUIRE(N % BN == 0 && K % BK == 0 && lda >= K && ldc >= N && lda % 8 == 0 && ldc % 8 == 0, "keds_gemm_x3: bad shape / strides");
    // buffer offsets are 32-bit: a plane must be reachable from a tile's first row
    KEDS_REQUIRE(a_plane > 0 && w_plane > 0 && a_plane * 2 + 256LL * lda * 2 < (1LL << 31) && w_plane * 2 + 256LL * K * 2 < (1LL << 31),
                 "keds_gemm_x3: plane strides out of range");
    hipStream_t st = (hipStream_t)stream;
    g_x3_aplane = a_plane;
    KEDS_REQUIRE(w_exp >= -100 && w_exp <= 100, "keds_gemm_x3: w_exp %d out of range", w_exp);
    g_x3_wplane = x3_pack_wplane(w_plane, w_exp);
    switch (epilogue) {
        case KEDS_EPI_X3_BIAS_F32:
            return launch_gemm<KEDS_EPI_X3_BIAS_F32>(a, w, bias, out, M, N, K, nullptr, 0, nullptr, lda, ldc, st);
        case KEDS_EPI_X3_RESID_F32:
            return launch_gemm<KEDS_EPI_X3_RESID_F32>(a, w, bias, out, M, N, K, nullptr, 0, nullptr, lda, ldc, st);
        case KEDS_EPI_X3_QGELU_PAIR:
            KEDS_REQUIRE(aux_i > 0, "keds_gemm_x3: KEDS_EPI_X3_QGELU_PAIR needs the output planes' stride");
            return launch_gemm<KEDS_EPI_X3_QGELU_PAIR>(a, w, bias, out, M, N, K, nullptr, aux_i, nullptr, lda, ldc, st);
        default: keds_set_error("keds_gemm_x3: epilogue %d is not a split-operand epilogue", epilogue); return KEDS_E_ARG;
    }
}

extern "C" int keds_gemm_bt_ex(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                               int M, int N, int K, int epilogue, const float* aux, int aux_i, void* stream) {
    return keds_gemm_bt_ex2(A, lda, W, bias, out, ldc, M, N, K, epilogue, aux, aux_i, nullptr, stream);
}

extern "C" int keds_gemm_bt(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                            int epilogue, const float* aux, int aux_i, void* stream) {
    return keds_gemm_bt_ex2(A, K, W, bias, out, N, M, N, K, epilogue, aux, aux_i, nullptr, stream);
}
