// MXFP8 GEMM for BASELINE config 5 ("fp8 (CDNA4 MFMA) ViT-L/14 encoders"): OCP e4m3 elements with one e8m0 scale per
// 32 consecutive K (OCP MX), multiplied by v_mfma_scale_f32_16x16x128_f8f6f4, which applies the block scales in
// hardware and runs at twice the bf16 MFMA rate.
//
// Operand / scale lane maps of the instruction, measured with exact data (tools/micro/mx_layout_probe.hip,
// mx_scale_probe.hip): lane l = (c = l & 15, g = l >> 4) holds, for row (A) / column (B) c, the 16 bytes
// k = 16g .. 16g+15 in operand bytes 0..15 and k = 64 + 16g .. 64 + 16g + 15 in bytes 16..31, and ITS scale (byte 0 of
// the scale VGPR) applies to the 32-element block k = 32g .. 32g+31 of that row.  A literal scale operand is mis-read;
// scales must come from a VGPR.  So a 128-byte LDS row (one K-tile of 128 fp8) is read exactly like the bf16 kernel
// reads its two K-steps (16-byte chunk g and chunk 4+g), and the whole 256 x 256 tile machinery of gemm.hip carries over
// with one MFMA step per K-tile: same XOR swizzle, DMA pieces, W-row permutation, supertile mapping.
//
// Scales live in HBM as [K/128][rows] dwords (byte b of the dword of (k-tile, row) = block 4*ktile + b), so the 256 rows
// of a tile are 1 KiB contiguous per K-tile and ride along as one extra LDS-DMA piece per operand.
#include "keds_common.h"
#include "gemm_quad_gen.h"       // accumulator read-back / drain of the 4-wave kernels
#include "gemm_fp8_quad_gen.h"
#include <cstdlib>

namespace {

constexpr int TM = 256, TN = 256, TKB = 128;         // K-tile = 128 fp8 = 128 bytes per LDS row
constexpr int OP_BYTES = 256 * 128;                  // 32 KiB per operand per K-tile
constexpr int SC_BYTES = 256 * 4;                    // scale dwords of one operand per K-tile
constexpr int PBUF_BYTES = 2 * OP_BYTES + 2 * SC_BYTES;   // X | W | sX | sW
constexpr int SIDE_OFF = 2 * PBUF_BYTES;             // side area of the LN epilogues: {rstd, -mean rstd}[256] | bias'[256] | colsum[256]
constexpr int LDS_BYTES = SIDE_OFF + 4096;           // 136 KiB

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ int swz_f8(int row) { return (row >> 1) & 7; }
// same W-row permutation as gemm.hip: a lane's accumulators of n-tiles (2p, 2p+1) are 8 consecutive output columns
__device__ __forceinline__ int perm_w8(int R) {
    const int t = R >> 4, i = R & 15;
    return 64 * (t >> 2) + 32 * ((t >> 1) & 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
}

// ---- quantisation: one wave per row, lane handles 8 consecutive elements (4 lanes per 32-block) ---------------------
template <bool IN_BF16>
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const void* __restrict__ x, int rows, int K, int rows_pad,
                                                             unsigned char* __restrict__ q, unsigned char* __restrict__ scales) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        float v[8];
        if constexpr (IN_BF16) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(x) + (size_t)r * K + k0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
        } else {
            const float* p = reinterpret_cast<const float*>(x) + (size_t)r * K + k0;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = a[j];
                v[4 + j] = b[j];
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));       // the 4 lanes of one 32-element block
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        const int e = mx_block_exp(amax);
        *reinterpret_cast<uint2*>(q + (size_t)r * K + k0) = mx_pack8(v, e);
        if ((lane & 3) == 0) {
            scales[mx_scale_index(k0 >> 5, r, rows_pad)] = (unsigned char)(e + 127);
        }
    }
}

// One wave per output row n of an nn.Linear weight W [N,K]: (optionally) fold the LayerNorm that feeds it, quantise to
// MXFP8 and emit what the LN epilogue needs: bias' = bias + W.beta (fp32) and csum = row sum of the DEQUANTISED W.diag(gamma)
// (what the MFMA multiplies the row mean with).  gamma == nullptr: plain quantisation, bias copied, csum still written.
__global__ __launch_bounds__(256) void fold_quantize_mxfp8_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int N, int K, int n_pad, unsigned char* __restrict__ wq,
                                                                  unsigned char* __restrict__ wscale, float* __restrict__ bias_csum) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float cs = 0.f, bb = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        float v[8];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = W[(size_t)n * K + k0 + j];
            v[j] = gamma ? w * gamma[k0 + j] : w;
            if (beta) bb += w * beta[k0 + j];
            amax = fmaxf(amax, fabsf(v[j]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        const int e = mx_block_exp(amax);
        const uint2 q = mx_pack8(v, e);
        *reinterpret_cast<uint2*>(wq + (size_t)n * K + k0) = q;
        if ((lane & 3) == 0) wscale[mx_scale_index(k0 >> 5, n, n_pad)] = (unsigned char)(e + 127);
        const float sc = e == -127 ? 0.f : __uint_as_float((unsigned)(e + 127) << 23);
        const float d = ((__builtin_amdgcn_cvt_f32_fp8(q.x, 0) + __builtin_amdgcn_cvt_f32_fp8(q.x, 1)) +
                         (__builtin_amdgcn_cvt_f32_fp8(q.x, 2) + __builtin_amdgcn_cvt_f32_fp8(q.x, 3))) +
                        ((__builtin_amdgcn_cvt_f32_fp8(q.y, 0) + __builtin_amdgcn_cvt_f32_fp8(q.y, 1)) +
                         (__builtin_amdgcn_cvt_f32_fp8(q.y, 2) + __builtin_amdgcn_cvt_f32_fp8(q.y, 3)));
        cs += d * sc;
    }
    cs = wave_sum(cs);
    bb = wave_sum(bb);
    if (lane == 0) {
        bias_csum[n] = (bias ? bias[n] : 0.f) + bb;
        bias_csum[N + n] = cs;
    }
}

// ---- the epilogues of one wave's 128 x 64 block of accumulators (both tile kernels): wave row wm (0..1), wave column wn (0..3)
// timing-only ablations of the 4-wave kernel (tools/fp8_ablate.sh rebuilds with -DKEDS_FQ_ABL=n; results are wrong), bits:
// 1: no epilogue (read-back only)   2: the K-loop runs its first two and last two K-tiles only
// residual epilogues:  4: no statistics atomics   8: no MXFP8 copy / scale stores   16: no residual store   32: no residual load
// K-loop (tools/fp8_kloop_ab.sh, with -DKEDS_FQ_STAMP):  64: no DMA pieces   128: no fragment / scale reads   256: no wait +
//          barrier per K-tile   512: the wait without the barrier   1024: the barrier without the vmcnt wait   2048: the barrier alone
#ifndef KEDS_FQ_ABL
#define KEDS_FQ_ABL 0
#endif
// ---- the epilogues, one ROW GROUP at a time: the 16 rows 16 mi + c of one wave's 128 x 64 block, lane (g, c) owning row c and
// the columns 64 wn + 32 pp + 8 g + 0..7 (a[2 pp], a[2 pp + 1]) of both 32-column MX blocks pp.  The four lanes g = 0..3 of a row
// hold exactly one MX block per pp, so block amax / row sums are two xor-shuffles.  (Row groups outermost since round 4: the
// 4-wave kernel reads its accumulators back 16 registers at a time -- with a 64-register quarter, the residual chunks and the
// state carried between the two blocks of all eight row groups the allocator ran out and parked values in accumulator AGPRs.)
struct MxSide {                   // per MX block pp: bias' (or bias) and the LayerNorm column sums of the lane's 8 columns
    f32x4 b0[2], b1[2], c0[2], c1[2];
};
template <int EPI>
__device__ __forceinline__ void mx_side(MxSide& sd, const char* side, const float* __restrict__ bias, int n0, int wn, int g) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        sd.b0[pp] = sd.b1[pp] = sd.c0[pp] = sd.c1[pp] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == 1 || EPI == 2) {
            const char* sb = side + 2048 + (64 * wn + 32 * pp + 8 * g) * 4;
            sd.b0[pp] = *reinterpret_cast<const f32x4*>(sb);
            sd.b1[pp] = *reinterpret_cast<const f32x4*>(sb + 16);
            sd.c0[pp] = *reinterpret_cast<const f32x4*>(sb + 1024);
            sd.c1[pp] = *reinterpret_cast<const f32x4*>(sb + 1024 + 16);
        } else if (bias) {
            const int n = n0 + 64 * wn + 32 * pp + 8 * g;
            sd.b0[pp] = *reinterpret_cast<const f32x4*>(bias + n);
            sd.b1[pp] = *reinterpret_cast<const f32x4*>(bias + n + 4);
        }
    }
}
// PRE: `pre[pp]` holds the row group's fp16 residual chunks (EPI 4), loaded by the caller ahead of time; stores go through
//      32-bit offsets from the tile's base (no pointer pair per row)
// STAT (residual epilogues): 2 = one atomic pair per row and 64-column block; 4 = the partials go to LDS instead
//      (`red`: [wave column 0..3][row] {sum, sum sq}; the caller adds the four and issues ONE atomic pair per row and tile:
//      2,048 -> 512 atomics per tile, 7 us of out-proj's 69)
template <int EPI, bool PRE = false, int STAT = 2>
__device__ __forceinline__ void mx_epilogue_rows(const f32x4 (&a)[4], int mi, const MxSide& sd, const char* side, void* __restrict__ out,
                                                 int m0, int n0, int N, int wm, int wn, int g, int c, float* __restrict__ aux,
                                                 float* __restrict__ aux2, unsigned char* __restrict__ qout,
                                                 unsigned char* __restrict__ qscale, int q_pad, const f16x8 (&pre)[2],
                                                 float* red = nullptr) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int m = m0 + 128 * wm + 16 * mi + c;                   // M is a multiple of 256: every row is valid
    [[maybe_unused]] float rstd = 1.f, nmr = 0.f;
    if constexpr (EPI == 1 || EPI == 2) {
        const f32x2 cf = *reinterpret_cast<const f32x2*>(side + (128 * wm + 16 * mi + c) * 8);
        rstd = cf[0];
        nmr = cf[1];
        if (aux2 && n0 == 0 && wn == 0 && g == 0) {               // (a zero made HERE: as a hoisted constant it is four registers held -- or spilled -- across the K-loop)
            unsigned z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            *reinterpret_cast<u32x4*>(reinterpret_cast<keds_stat_t*>(aux2) + 2 * (size_t)m) = u32x4{z, z, z, z};
        }
    }
    [[maybe_unused]] float rs = 0.f, rss = 0.f;
    [[maybe_unused]] uint2 mxk = uint2{0u, 0u};
    [[maybe_unused]] int mxe = 0;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int n = n0 + 64 * wn + 32 * pp + 8 * g;
        f32x4 v0, v1;
        if constexpr (EPI == 1 || EPI == 2) {
            v0 = a[2 * pp] * rstd + (sd.c0[pp] * nmr + sd.b0[pp]);
            v1 = a[2 * pp + 1] * rstd + (sd.c1[pp] * nmr + sd.b1[pp]);
        } else {
            v0 = a[2 * pp] + sd.b0[pp];
            v1 = a[2 * pp + 1] + sd.b1[pp];
        }
        if constexpr (EPI == 2) {
            // x * sigmoid(1.702 x)  (src/model/model.py:300-302) on whole vectors: the scale, the + 1 and the product are packed
            // operations (gemm.hip, pair_ln_epilogue: the scalar form compiled to twice the issue slots)
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
        if constexpr (EPI == 3 || EPI == 4) {
            if constexpr (EPI == 3) {
                float* o = reinterpret_cast<float*>(out) + (size_t)m * N + n;
                v0 += *reinterpret_cast<const f32x4*>(o);
                v1 += *reinterpret_cast<const f32x4*>(o + 4);
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            } else {                                          // fp16 residual stream (cf. KEDS_EPI_RESID_STATS_F16)
                f16x8* o = reinterpret_cast<f16x8*>(reinterpret_cast<f16_t*>(out) + (size_t)m * N + n);
                f16x8 r;
                if constexpr (PRE && !(KEDS_FQ_ABL & 32)) r = pre[pp];
                else if constexpr ((KEDS_FQ_ABL & 32) != 0) r = f16x8{(f16_t)1.f, (f16_t)2.f, (f16_t)-1.f, (f16_t)0.5f, (f16_t)1.f, (f16_t)2.f, (f16_t)-1.f, (f16_t)0.5f};
                else r = *o;
                v0 += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                v1 += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
                const f16x8 nr = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3],
                                       (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
                if constexpr ((KEDS_FQ_ABL & 16) != 0) { if (nr[0] == (f16_t)123.25f) *o = nr; }
                else if constexpr (PRE)
                    keds_store16<8>(nr, reinterpret_cast<f16_t*>(out) + (size_t)m0 * N + n0, (unsigned)(((size_t)(m - m0) * N + (n - n0)) * 2));
                else
                    *o = nr;
            }
            rs += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
            rss += ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) +
                   ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
        }
        if constexpr (EPI == 0 || EPI == 1) {
            keds_store16<KEDS_ST_FP8_BF16>(bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                                                   (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]},
                                           reinterpret_cast<bf16_t*>(out) + (size_t)m0 * N + n0,
                                           (unsigned)(((size_t)(m - m0) * N + (n - n0)) * 2));
        } else {                                              // MXFP8 copy: one 32-column block per (row, pp)
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
            amax = rows_max(amax);
            const int e = mx_block_exp(amax);
            const uint2 pk = mx_pack8(v, e);
            // 16-byte stores: the lane's 8 bytes of block pp = 0 are kept until pp = 1, then the four lanes of the row
            // swap (v_permlane16_swap: the odd 16-lane rows of the first operand against the even rows of the second)
            // so that g = 0 / 2 own columns 0-15 / 16-31 of block 0 and g = 1 / 3 those of block 1; the two scale
            // bytes of the row (blocks 2 wn and 2 wn + 1 of its dword) go out as one 16-bit store
            if (pp == 0) {      // (8-byte + 1-byte stores per block: +0.35 ms of the 16.5 ms fp8 step, same-box A/B)
                mxk = pk;
                mxe = e;
            } else {
                const auto s0 = __builtin_amdgcn_permlane16_swap(mxk.x, pk.x, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(mxk.y, pk.y, false, false);
                const int blk = g & 1, half = g >> 1;          // after the swap: this lane's block and 16-column half
                const unsigned qoff = (unsigned)((size_t)(m - m0) * N + (64 * wn + 32 * blk + 16 * half));
                if constexpr ((KEDS_FQ_ABL & 8) != 0) {
                    if (s0[0] == 0x12345678u) qout[0] = 1;
                } else if constexpr (EPI == 2)       // MLP hidden (MXFP8): read once by c_proj
                    keds_store16<KEDS_ST_FP8_MX>(u32x4{s0[0], s1[0], s0[1], s1[1]}, qout + (size_t)m0 * N + n0, qoff);
                else                           // MXFP8 copy of the residual stream: the next GEMM's A operand
                    keds_store16<(PRE && KEDS_ST_FP8_MXR == 0) ? 8 : KEDS_ST_FP8_MXR>(u32x4{s0[0], s1[0], s0[1], s1[1]}, qout + (size_t)m0 * N + n0, qoff);
                if (g == 0 && !(KEDS_FQ_ABL & 8))
                    *reinterpret_cast<unsigned short*>(qscale + mx_scale_index((n0 + 64 * wn) >> 5, m, q_pad)) =
                        (unsigned short)((mxe + 127) | ((e + 127) << 8));
            }
        }
    }
    if constexpr (EPI == 3 || EPI == 4) {
        rs = rows_sum(rs);
        rss = rows_sum(rss);
        if constexpr (STAT == 4) {
            if (g == 0) *reinterpret_cast<f32x2*>(red + (wn * 256 + 128 * wm + 16 * mi + c) * 2) = f32x2{rs, rss};
        } else {
            if (g == 0 && !(KEDS_FQ_ABL & 4)) keds_stat_add(reinterpret_cast<keds_stat_t*>(aux) + 2 * (size_t)m, rs, rss);
        }
    }
}
// a wave's whole 128 x 64 block from acc[ni][mi] (8-wave kernel)
template <int EPI>
__device__ __forceinline__ void mx_epilogue(f32x4 (&acc)[4][8], const char* side, const float* __restrict__ bias, void* __restrict__ out,
                                            int m0, int n0, int N, int wm, int wn, int g, int c, float* __restrict__ aux,
                                            float* __restrict__ aux2, unsigned char* __restrict__ qout,
                                            unsigned char* __restrict__ qscale, int q_pad) {
    MxSide sd;
    mx_side<EPI>(sd, side, bias, n0, wn, g);
    const f16x8 none[2] = {};
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const f32x4 a4[4] = {acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi]};
        mx_epilogue_rows<EPI>(a4, mi, sd, side, out, m0, n0, N, wm, wn, g, c, aux, aux2, qout, qscale, q_pad, none);
    }
}

// ---- 256 x 256 x 128 MXFP8 tile kernel: 8 waves (2 along m x 4 along n, 128 x 64 outputs each) -----------------------
// out bf16 [M,N] = A[M,K] . W[N,K]^T + bias
// EPI: 0 = out bf16 = acc + bias
//      1 = LayerNorm folded in (gemm.hip, KEDS_EPI_LN_BIAS_BF16): bias = [bias' | csum], aux = row statistics, aux2 = statistics
//          buffer to clear; out bf16
//      2 = the same + QuickGELU, emitted as MXFP8: qout / qscale (rows padded to q_pad) instead of `out`
//      3 = out fp32 += acc + bias (residual stream), its MXFP8 copy to qout / qscale and row {sum, sum sq} atomically to aux
template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void gemm_mxfp8_kernel(const unsigned char* __restrict__ X, const unsigned char* __restrict__ sX,
                                                            const unsigned char* __restrict__ W, const unsigned char* __restrict__ sW,
                                                            const float* __restrict__ bias, void* __restrict__ out, int M, int N,
                                                            int K, int n_tiles, int m_pad, int n_pad, float* __restrict__ aux,
                                                            float* __restrict__ aux2, unsigned char* __restrict__ qout,
                                                            unsigned char* __restrict__ qscale, int q_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    const int m_tiles = gridDim.x / n_tiles;
    if ((m_tiles & 7) == 0 && (n_tiles & 3) == 0) {                // 8 x 4 supertile per XCD (gemm.hip)
        const int grp = bid >> 5, within = bid & 31;
        const int gcols = n_tiles >> 2;
        const int gm = grp / gcols, gn = grp - gm * gcols;
        tm = gm * 8 + (within & 7);
        tn = gn * 4 + (within >> 3);
    } else {
        tm = bid / n_tiles;
        tn = bid - tm * n_tiles;
    }
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging: data pieces as in the bf16 kernel (piece = 8 LDS rows; wave owns pieces wave + 8*i); waves 0 / 1
    // also bring the 1 KiB of X / W scale dwords of the K-tile
    const int R0 = 8 * wave + (lane >> 3);
    const int sch = (lane & 7) ^ swz_f8(R0);
    // uniform tile bases (SGPRs) + 32-bit lane offsets: 64-bit per-lane pointers cost 6 VGPRs this kernel does not have
    const unsigned char* xt = X + (size_t)m0 * K;
    const unsigned char* wt = W + (size_t)n0 * K;
    const unsigned xoff = (unsigned)R0 * (unsigned)K + sch * 16;
    const unsigned woff = (unsigned)perm_w8(R0) * (unsigned)K + sch * 16;
    const unsigned rstride = 64u * (unsigned)K;
    const unsigned char* st_base = wave == 0 ? sX + (size_t)m0 * 4 : sW + (size_t)n0 * 4;
    const size_t sstride = (size_t)(wave == 0 ? m_pad : n_pad) * 4;               // next K-tile's dwords
    auto issue = [&](int p, int q) {
        const int i = q & 3;
        const unsigned char* src = (q < 4 ? xt + (xoff + i * rstride + (unsigned)p * TKB) : wt + (woff + i * rstride + (unsigned)p * TKB));
        char* dst = smem + (p & 1) * PBUF_BYTES + (q < 4 ? 0 : OP_BYTES) + (wave + 8 * i) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_scales = [&](int p) {
        if (wave < 2) {
            char* dst = smem + (p & 1) * PBUF_BYTES + 2 * OP_BYTES + wave * SC_BYTES;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st_base + (size_t)p * sstride + lane * 16),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xrow = (128 * wm + c) * 128;                         // + mi * 2048
    const int wrow = OP_BYTES + (64 * wn + c) * 128;               // + ni * 2048
    // scale dwords: X row 128*wm + 16*mi + c; W LDS row R = 64*wn + 16*ni + c holds W row perm_w8(R & 63) + 64*(R >> 6)
    const int sx_off = 2 * OP_BYTES + (128 * wm + c) * 4;          // + mi * 64
    // perm_w8(16*ni + c) = [8*(c>>2) + (c&3)] + [32*((ni>>1)&1) + 4*(ni&1)]: one lane-dependent base + a constant per ni
    const int sw_base = 2 * OP_BYTES + SC_BYTES + (64 * wn + 8 * (c >> 2) + (c & 3)) * 4;

    f32x4 acc[4][8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int np = K / TKB;                                        // >= 2
    // LN epilogues (cf. gemm_bt_pair_kernel): thread t < 256 fetches row t's statistics, thread 256 + j column j's bias' and
    // column sum, BEFORE the first DMA piece; they become the side-area image while K-tile 0 is in flight
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    [[maybe_unused]] u32x4 st_raw = u32x4{0, 0, 0, 0};
    [[maybe_unused]] float pb = 0.f, pc = 0.f;
    if constexpr (EPI == 1 || EPI == 2) {
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
#pragma unroll
    for (int q = 0; q < 8; ++q) issue(0, q);
    issue_scales(0);
    if constexpr (EPI == 1 || EPI == 2) {
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(st_raw), "+v"(pb), "+v"(pc)::"memory");   // older than the (8 or 9) DMA pieces
        if (tid < 256) {
            const float invk_ = 1.0f / (float)K;
            const float mean = keds_stat_value((keds_stat_t)(((unsigned long long)st_raw[1] << 32) | st_raw[0])) * invk_;
            const float ss = keds_stat_value((keds_stat_t)(((unsigned long long)st_raw[3] << 32) | st_raw[2]));
            const float rsd = rsqrtf(fmaxf(ss * invk_ - mean * mean, 0.f) + 1e-5f);
            *reinterpret_cast<f32x2*>(smem + SIDE_OFF + tid * 8) = f32x2{rsd, -mean * rsd};
        } else {
            *reinterpret_cast<float*>(smem + SIDE_OFF + 2048 + (tid - 256) * 4) = pb;
            *reinterpret_cast<float*>(smem + SIDE_OFF + 3072 + (tid - 256) * 4) = pc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    auto load_frag = [&](const char* buf, int row_off) {
        i32x8 v;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(buf + row_off + slot0);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(buf + row_off + slot1);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };

    // Ping-pong wave groups: the waves of one SIMD (w and w+4) run the two phases of a K-tile in opposite order inside
    // each barrier interval,
    //     group 0:  read fragments of tile p      -> multiply tile p
    //     group 1:  multiply tile p-1 (registers) -> read fragments of tile p
    // with a single barrier per K-tile; all fragments of a tile live in registers (96 VGPRs) between the phases and
    // tile p+1 streams into the other buffer meanwhile.
    // MEASURED (tools/bench_fp8.py, proj shape 32768 x 1024 x 4096): MFMA only 83 us, DMA + LDS reads only 84 us,
    // everything 174 us -- the phases ADD even though they now overlap in time on every SIMD, and they added in the
    // plain "read, then multiply" loop as well (82 + 98 = 178 us).  A register-only MFMA loop already holds only
    // ~1.6 GHz: the chip is power-limited here, the clock falls when the LDS / DMA traffic runs under the MFMAs, and
    // wall time follows the ENERGY per K-tile rather than the critical path.  What helps is fewer bytes moved per flop
    // (fp8 itself: 1.3x over bf16 at K = 4096), not a tighter schedule.
    const int grp = wave >> 2;
    i32x8 wf[4], xf[8];
    int swp = 0, sxp[2] = {0, 0};          // e8m0 scales packed four to a register, picked by the MFMA's op_sel byte
    auto multiply = [&]() {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#define KEDS_FP8_MFMA(NI, OB)                                                                                         \
    acc[NI][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[NI], xf[mi], acc[NI][mi], 0, 0, NI, swp, OB, sxp[mi >> 2]);
#define KEDS_FP8_ROW(OB) KEDS_FP8_MFMA(0, OB) KEDS_FP8_MFMA(1, OB) KEDS_FP8_MFMA(2, OB) KEDS_FP8_MFMA(3, OB)
            if constexpr (DBG == 2) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni][mi][0] += __int_as_float(wf[ni][0] ^ xf[mi][3] ^ swp ^ sxp[mi >> 2]);
            } else {
                switch (mi & 3) {                      // op_sel is an immediate: the unrolled mi makes this a constant
                    case 0: KEDS_FP8_ROW(0) break;
                    case 1: KEDS_FP8_ROW(1) break;
                    case 2: KEDS_FP8_ROW(2) break;
                    default: KEDS_FP8_ROW(3) break;
                }
            }
#undef KEDS_FP8_ROW
#undef KEDS_FP8_MFMA
        }
    };
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (DBG == 3) {                                      // diagnostic build only
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int p = 0; p < np; ++p) {
        const char* cb = smem + (p & 1) * PBUF_BYTES;
        // tile p has landed (own pieces) and, past the barrier, everybody is done with the other buffer
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (p + 1 < np && DBG != 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) issue(p + 1, q);
            issue_scales(p + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1 && p > 0 && DBG != 1) multiply();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 1 && DBG != 4) {
            swp = 0;
            sxp[0] = sxp[1] = 0;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                wf[ni] = load_frag(cb, wrow + ni * 2048);
                swp |= ((*reinterpret_cast<const unsigned*>(cb + sw_base + (32 * ((ni >> 1) & 1) + 4 * (ni & 1)) * 4) >> (8 * g)) & 0xFFu) << (8 * ni);
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                xf[mi] = load_frag(cb, xrow + mi * 2048);
                sxp[mi >> 2] |= ((*reinterpret_cast<const unsigned*>(cb + sx_off + mi * 64) >> (8 * g)) & 0xFFu) << (8 * (mi & 3));
            }
        } else if (p == 0) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[mi][j] = lane * 0x01010101 + j + mi;
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni] = xf[ni] + ni;
            swp = 0x7F7F7F7F;
            sxp[0] = sxp[1] = 0x7F7F7F7F + (lane & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0 && DBG != 1) multiply();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 1 && DBG != 1) multiply();

    if constexpr (DBG == 3) {   // (core-clock ticks, 100 MHz ticks) of the K-loop into the first words of this tile's output
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<bf16_t*>(out) + (size_t)m0 * N + n0);
            dbg[0] = t1 - t0;
            dbg[1] = r1 - r0;
        }
        return;
    }
    mx_epilogue<EPI>(acc, smem + SIDE_OFF, bias, out, m0, n0, N, wm, wn, g, c, aux, aux2, qout, qscale, q_pad);
}


// ---- the 4-wave form: 2 x 2 waves, 128 x 128 outputs each, 256 accumulators in fixed AGPRs, persistent -------------------
// The recipe of gemm.hip's gemm_bt_quad_kernel carried over to the block-scaled instruction (round 4).  Why it pays MORE here
// than for bf16: the 8-wave kernel's 128 x 64 wave tiles read 192 KiB of fragments per K-tile, 1,536 LDS cycles, and the DMA
// writes another 512 -- as long as the K-tile's 2,048 matrix cycles: LDS and matrix pipe are co-critical and their phases add.
// 128 x 128 wave tiles read 128 KiB.  One K-tile (128 fp8 per row) is ONE step of 64 MFMAs (32 cycles each); a lane's
// fragment is the row's 16-byte chunks g and 4 + g (two ds_read_b128).
//   registers: X fragments double buffered (xa / xb, 64 + 64), W fragments refilled IN PLACE: group j (W fragment j against the
//     eight X fragments) is the only user of w[j], so the NEXT K-tile's w[j] is requested right behind the group's last MFMA
//     (an MFMA reads its A / B operands when it issues; the LDS answer comes a hundred cycles later) -- 64 more; 192 + scales;
//   LDS: two K-tile buffers; every read of step p targets the buffer of K-tile p + 1, so K-tile p + 2 streams into the buffer
//     of K-tile p during step p; one wait + barrier per K-tile;
//   gaps: one behind every MFMA, at most one memory instruction each: DMA piece n / 4 in gaps n % 4 == 0, fragment reads in the
//     odd gaps (W fragment j at gaps 8j + 7 / 8j + 9), scale dwords in gaps n % 4 == 2 (read in one gap, packed into the byte
//     the instruction's op_sel picks in the next: four e8m0 scales per register);
//   tiles: persistent walk, the next tile's first two K-tiles and side data requested before this tile's epilogue.
// the lane id from the execution mask, made where it is used: thread-id arithmetic kept in registers across the K-loop is what the
// 4-wave kernel has no room for (round 6: the allocator spilled the thread id and reloaded it behind `s_waitcnt vmcnt(0)` -- a wait
// for every DMA piece in flight -- in front of the epilogue)
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
namespace fq {
// LDS: X0 | X1 | W0 | W1 | sX0 | sX1 | sW0 | sW1 | side areas.  The two K-tile buffers of an operand are 32 KiB (scales: 1 KiB)
// apart, so the buffer is part of a read's 16-bit immediate offset and ONE address register per (operand, chunk) serves both.
constexpr int XB = 0, WB = 2 * OP_BYTES, SXB = 4 * OP_BYTES, SWB = SXB + 2 * SC_BYTES;
constexpr int SIDE0 = SWB + 2 * SC_BYTES;        // two side areas (tile i uses i & 1)
constexpr int RAW = SIDE0 + 2 * 4096;            // the next tile's row statistics (4 KiB) | bias' (1 KiB) | column sums (1 KiB), by LDS-DMA
constexpr int LDS_BYTES = RAW + 6144;            // 146 KiB
}  // namespace fq

#define KEDS_FQ_M(FIRST, j, mi, xc, swc, sxc)                                                                    \
    if constexpr (FIRST) { KEDS_FQ_MFMAZ_##j##_##mi(w[j], xc[mi], swc[(j) >> 2], sxc[(mi) >> 2]) }               \
    else { KEDS_FQ_MFMA_##j##_##mi(w[j], xc[mi], swc[(j) >> 2], sxc[(mi) >> 2]) }
// half hf (0: chunk g, 1: chunk 4 + g) of the next K-tile's X fragment fr / W fragment fr
#define KEDS_FQ_RD_X(fr, hf, xn, nb)                                                                             \
    {                                                                                                            \
        const i32x4 v_ = *reinterpret_cast<const i32x4*>(smem + ((hf) ? xrd1 : xrd0) + (nb) * OP_BYTES + (fr) * 2048); \
        xn[fr][4 * (hf)] = v_[0]; xn[fr][4 * (hf) + 1] = v_[1]; xn[fr][4 * (hf) + 2] = v_[2]; xn[fr][4 * (hf) + 3] = v_[3]; \
    }
#define KEDS_FQ_RD_W(fr, hf, nb)                                                                                 \
    {                                                                                                            \
        const i32x4 v_ = *reinterpret_cast<const i32x4*>(smem + ((hf) ? wrd1 : wrd0) + (nb) * OP_BYTES + (fr) * 2048); \
        w[fr][4 * (hf)] = v_[0]; w[fr][4 * (hf) + 1] = v_[1]; w[fr][4 * (hf) + 2] = v_[2]; w[fr][4 * (hf) + 3] = v_[3]; \
    }
// scale dword k (0..7: X fragment k, 8..15: W fragment k - 8) of the next K-tile: read / pack its byte g into the register
#define KEDS_FQ_SC_READ(k, nb)                                                                                   \
    sct[(k) == 15 ? 2 : ((k) & 1)] = *reinterpret_cast<const unsigned*>(                                         \
        smem + (nb) * SC_BYTES + ((k) < 8 ? sx_off + (k) * 64 : sw_base + (64 * (((k) - 8) >> 2) + 32 * ((((k) - 8) >> 1) & 1) + 4 * (((k) - 8) & 1)) * 4));
// (one v_perm_b32 per scale: byte g of the dword just read into byte k & 3 of the packed register, the other bytes kept; as asm
// volatile, because the compiler SINKS a plain shift / or chain to the register's first use -- the next step -- and keeps all 16
// dwords alive until then: with them the K-loop spilled into the accumulators' AGPRs)
#define KEDS_FQ_SC_PACK(k, sxn, swn)                                                                             \
    {                                                                                                            \
        if constexpr (((k) & 3) == 0)                                                                            \
            asm volatile("v_perm_b32 %0, %1, %1, %2" : "=v"(*((k) < 8 ? &sxn[(k) >> 2] : &swn[((k) - 8) >> 2]))   \
                         : "v"(sct[(k) == 15 ? 2 : ((k) & 1)]), "v"(psel[0]));                                   \
        else                                                                                                     \
            asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(*((k) < 8 ? &sxn[(k) >> 2] : &swn[((k) - 8) >> 2]))   \
                         : "v"(sct[(k) == 15 ? 2 : ((k) & 1)]), "v"(psel[(k) & 3]));                             \
    }
// gap n (0..63) behind MFMA n of a step
// (Round 4, measured and not kept.  Stamped ablations, cycles of one tile's K-loop of 8 K-tiles (profiles/r04_fp8_kloop_ablation.txt):
// MFMAs alone 16.65 k; + fragment reads + DMA pieces 16.95 k; + the barrier 17.3 k; + `s_waitcnt vmcnt(0)` 20.7 k -- 400 cycles per
// K-tile waiting for pieces requested less than a K-step earlier.  (a) All 17 pieces in the step's first half: 20.6 k, and 3 %
// fewer img/s (r04_fp8_sched_ab.txt): a piece takes longer than a K-step to land.  (b) The A operand one K-tile further ahead in
// the same two LDS buffers -- X-fragment reads in the first half of the step, a mid-step barrier, the X pieces of K-tile p + 3 in
// the second half, a counted vmcnt(8) in front of step p + 1; bit-identical: 20.1 k cycles, the tile 37.1 k instead of 38.8 k -- and
// on the same box qkv +2 %, c_fc +7 %, the bench -1.6 % in TIME (r04_fp8_deep_prefetch_ab.txt).  The cycles it saves are stall
// cycles; at the power cap they are not what the time is made of.)
#define KEDS_FQ_GAP(n_, xn, sxn, swn, nb, ISSUE, ip, PF)                                                         \
    {                                                                                                            \
        constexpr int gn_ = (n_);                                                                                \
        if constexpr ((gn_ & 3) == 0) {                                                                          \
            if constexpr (ISSUE) issue((ip), gn_ >> 2);                                                          \
        } else if constexpr (gn_ & 1) {                                                                          \
            constexpr int s_ = gn_ >> 1;                                                                         \
            if constexpr (s_ == 30) {                                                                            \
                if constexpr (ISSUE) issue_scales(ip);                                                           \
            } else if constexpr (PF) {                                                                           \
                if constexpr ((s_ & 3) == 3) {                                                                   \
                    KEDS_FQ_RD_W(s_ >> 2, 0, nb)                                                                 \
                    if constexpr (s_ == 31) KEDS_FQ_RD_W(7, 1, nb)                                               \
                } else if constexpr ((s_ & 3) == 0 && s_ > 0) {                                                  \
                    KEDS_FQ_RD_W((s_ >> 2) - 1, 1, nb)                                                           \
                } else {                                                                                         \
                    constexpr int t_ = s_ == 0 ? 0 : 2 * (s_ >> 2) + (s_ & 3);                                   \
                    KEDS_FQ_RD_X(t_ >> 1, t_ & 1, xn, nb)                                                        \
                }                                                                                                \
            }                                                                                                    \
        } else if constexpr (PF) {                                                                               \
            constexpr int k_ = gn_ >> 2;                                                                         \
            if constexpr (k_ > 0 && k_ < 15) KEDS_FQ_SC_PACK(k_ - 1, sxn, swn)                                   \
            if constexpr (k_ < 15) { KEDS_FQ_SC_READ(k_, nb) }                                                   \
            if constexpr (k_ == 14) { KEDS_FQ_SC_READ(15, nb) }                                                  \
            if constexpr (k_ == 15) { KEDS_FQ_SC_PACK(14, sxn, swn) KEDS_FQ_SC_PACK(15, sxn, swn) }              \
        }                                                                                                        \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#define KEDS_FQ_ONE(FIRST, j, mi, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                 \
    KEDS_FQ_M(FIRST, j, mi, xc, swc, sxc) KEDS_FQ_GAP(8 * (j) + (mi), xn, sxn, swn, nb, ISSUE, ip, PF)
#define KEDS_FQ_ROW(FIRST, j, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
    KEDS_FQ_ONE(FIRST, j, 0, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 1, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 2, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 3, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 4, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 5, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 6, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                      \
    KEDS_FQ_ONE(FIRST, j, 7, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)
// One K-tile: 64 MFMAs from (w, xc) with the scales (swc, sxc); the next K-tile's fragments / scales go to (w in place, xn,
// swn, sxn) from buffer `nb`; ISSUE: the DMA pieces of K-tile `ip`; SYNC_ 1: the next K-tile has landed and its predecessor's
// buffer is free (wait + barrier), 2: the buffer is free (every wave's fragment reads of it are done; no wait for pieces)
#define KEDS_FQ_STEP(FIRST, xc, sxc, swc, xn, sxn, swn, nb, SYNC_, ISSUE_, ip, PF_)                              \
    {                                                                                                            \
        constexpr int SYNCM = (KEDS_FQ_ABL & 256) ? 0 : (int)(SYNC_);                                            \
        constexpr bool SYNC = SYNCM == 1, ISSUE = (ISSUE_) && !(KEDS_FQ_ABL & 64),                               \
                       PF = (PF_) && !(KEDS_FQ_ABL & 128);                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if constexpr (SYNC && (KEDS_FQ_ABL & 512)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        else if constexpr ((SYNC && (KEDS_FQ_ABL & 1024)) || SYNCM == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else if constexpr (SYNC && (KEDS_FQ_ABL & 2048)) asm volatile("s_barrier" ::: "memory");                  \
        else if constexpr (SYNC) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        KEDS_FQ_ROW(FIRST, 0, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 1, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 2, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 3, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 4, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 5, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 6, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
        KEDS_FQ_ROW(FIRST, 7, xc, sxc, swc, xn, sxn, swn, nb, ISSUE, ip, PF)                                     \
    }

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_mxfp8_quad_kernel(const unsigned char* __restrict__ X, const unsigned char* __restrict__ sX,
                                                                 const unsigned char* __restrict__ W, const unsigned char* __restrict__ sW,
                                                                 const float* __restrict__ bias, void* __restrict__ out, int M, int N,
                                                                 int K, int n_tiles, int m_pad, int n_pad, float* __restrict__ aux,
                                                                 float* __restrict__ aux2, unsigned char* __restrict__ qout,
                                                                 unsigned char* __restrict__ qscale, int q_pad, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr bool LN = EPI == 1 || EPI == 2;
    const int m_tiles = ntiles / n_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn2 = wave & 1, wm = wave >> 1;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging: piece = 8 LDS rows (1 KiB); this wave owns pieces wave + 4 i (rows R0 + 32 i), i < 8, of either operand
    const int R0 = 8 * wave + (lane >> 3);                         // 0..31
    const int sch = (lane & 7) ^ swz_f8(R0);                       // swz_f8(R0 + 32 i) == swz_f8(R0)
    const unsigned xoff = (unsigned)R0 * (unsigned)K + sch * 16;
    const unsigned woff = (unsigned)perm_w8(R0) * (unsigned)K + sch * 16;   // perm_w8(R0 + 32 i) == perm_w8(R0) + 32 i
    const unsigned rstride = 32u * (unsigned)K;
    const unsigned soff = lane * 16;
    const unsigned sstride = (unsigned)(wave == 0 ? m_pad : n_pad) * 4u;    // the scale dwords of the next K-tile
#define make_rs(base) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(base)), 0, 0x7FFFFFFF, 0x00020000)
    auto xrs = make_rs(X), wrs = make_rs(W), srs = make_rs(sX);
    auto issue = [&](int p, int q) {                               // DMA piece q (0..15: X pieces 0..7, W pieces 0..7) of K-tile p
        const int i = q & 7;
        char* dst = smem + (q < 8 ? fq::XB : fq::WB) + (p & 1) * OP_BYTES + (wave + 4 * i) * 1024;
        const unsigned so = i * rstride + (unsigned)p * TKB;
        if (q < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, xoff, so, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, woff, so, 0, 0);
    };
    auto issue_scales = [&](int p) {                               // waves 0 / 1: the 1 KiB of X / W scale dwords of K-tile p
        if (wave < 2) {
            char* dst = smem + (wave == 0 ? fq::SXB : fq::SWB) + (p & 1) * SC_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void*)dst, 16, soff, (unsigned)p * sstride, 0, 0);
        }
    };
    auto point_at = [&](int m0_, int n0_) {
        xrs = make_rs(X + (size_t)m0_ * K);
        wrs = make_rs(W + (size_t)n0_ * K);
        srs = make_rs(wave == 0 ? sX + (size_t)m0_ * 4 : sW + (size_t)n0_ * 4);
    };
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    // read addresses (buffer 0; + OP_BYTES / SC_BYTES for buffer 1, + 2048 per fragment / 64 per X scale row group)
    const int xrd0 = fq::XB + (128 * wm + c) * 128 + slot0, xrd1 = fq::XB + (128 * wm + c) * 128 + slot1;
    const int wrd0 = fq::WB + (128 * wn2 + c) * 128 + slot0, wrd1 = fq::WB + (128 * wn2 + c) * 128 + slot1;
    const int sx_off = fq::SXB + (128 * wm + c) * 4;
    // W LDS row R = 128 wn2 + 16 j + c holds W row perm_w8(R & 63) + 64 (R >> 6): lane part + a constant per j
    const int sw_base = fq::SWB + (128 * wn2 + 8 * (c >> 2) + (c & 3)) * 4;
    // v_perm_b32 selectors: result byte q = byte g of the first source (selector value 4 + g), the other bytes from the second
    unsigned psel[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) psel[q] = (0x03020100u & ~(0xFFu << (8 * q))) | ((4u + (unsigned)g) << (8 * q));
    const int np = K / TKB;                                        // even, >= 4 (launcher)
    const int step = (int)gridDim.x;

    // side data of a tile (LN epilogues): row t's statistics, column t's bias' / column sum, one element per thread.  They travel by
    // LDS-DMA into a raw area (no registers live across the previous tile's epilogue) and become the side area behind the wait +
    // barrier that opens the tile
    auto side_load = [&](int m0_, int n0_) {
        if constexpr (LN) {
            // (buffer form: a FLAT-encoded global_load_lds makes the compiler's wait-count pass treat every later LDS wait as
            // out of order -- lgkmcnt(0) in front of each scale pack, a stall per gap)
            // (lane offsets made here: as loop invariants they would be two more registers held -- or spilled -- across the K-loop)
            const int te = wave * 64 + lane_now();
            const auto strs = make_rs(reinterpret_cast<const keds_stat_t*>(aux) + 2 * (size_t)m0_);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(strs, (__attribute__((address_space(3))) void*)(smem + fq::RAW + wave * 1024), 16,
                                                     te * 16, 0, 0, 0);
            if (wave < 2) {
                const auto brs = make_rs(bias + (wave ? N : 0) + n0_);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)(smem + fq::RAW + 4096 + wave * 1024), 16,
                                                         (te & 63) * 16, 0, 0, 0);
            }
        }
    };

    int id = blockIdx.x;
    int tm, tn;
    quad_tile_coords(xcd_remap(id, ntiles), m_tiles, n_tiles, tm, tn);
    int m0 = tm * TM, n0 = tn * TN;
    point_at(m0, n0);
    side_load(m0, n0);
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(0, q);
    issue_scales(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(1, q);
    issue_scales(1);

#ifdef KEDS_FQ_STAMP
    // diagnostic build (tools/fp8_stamp.py; EPI 1 only: qout carries a buffer of 8 x 64-bit words per (workgroup, wave)): core-clock
    // ticks of the phases of the workgroup's SECOND tile -- wait for its K-tiles | fragment reads of K-tile 0 | K-loop | requests of the
    // next tile | epilogue
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#define KEDS_FQ_TS(i) if (it == 1) ts[i] = __builtin_amdgcn_s_memtime();
#else
#define KEDS_FQ_TS(i)
#endif
    for (int it = 0;; ++it) {
        KEDS_FQ_TS(0)
        // this tile's K-tiles 0 and 1 and its raw side data have landed.  They were requested in the last two K-steps of the previous
        // tile, AHEAD of its epilogue's stores; vmcnt retires in order (loads and stores alike on this family), so the wait leaves
        // the epilogue's last NST stores in flight instead of sitting out their write latency (~1 k cycles per tile, round 4 stamps)
        // (12: every epilogue issues at least 32 stores behind the last DMA piece; a store takes ~250 cycles to issue and ~1-2 k to
        // retire, so the youngest dozen are the ones still in flight here)
        if (it == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        KEDS_FQ_TS(1)
        char* side = smem + fq::SIDE0 + (it & 1) * 4096;
        const int nid = id + step;
        const bool more = nid < ntiles;
        int nm0 = 0, nn0 = 0;
        if (more) {
            int ntm, ntn;
            quad_tile_coords(xcd_remap(nid, ntiles), m_tiles, n_tiles, ntm, ntn);
            nm0 = ntm * TM;
            nn0 = ntn * TN;
        }
        // K-tile 0: all fragments and scales from buffer 0.  EVERY LDS read of the tile start is issued back to back -- raw side
        // data, the 32 fragment chunks, the 16 scale dwords -- and only then does arithmetic begin: the side data (a dependent chain
        // of fp64 conversions) under the fragment reads, the scale bytes packed last.  (Round 4's order -- side data read, converted
        // and written first, the scale dwords read two at a time between their shifts -- cost four LDS round trips behind the other
        // waves' 32 KiB of fragment reads: 4.2 k cycles per tile where the reads themselves need ~1.2 k, profiles/r04_fp8_tile_phases.txt.)
        i32x8 xa[8], xb[8], w[8];
        unsigned sxa[2], sxb[2], swa[2], swb[2], sct[3];
        {
            [[maybe_unused]] u32x4 st_raw = u32x4{0, 0, 0, 0};
            [[maybe_unused]] float pb = 0.f, pc = 0.f;
            const int te = wave * 64 + lane_now();                 // (see side_load)
            if constexpr (LN) {
                st_raw = *reinterpret_cast<const u32x4*>(smem + fq::RAW + te * 16);
                pb = *reinterpret_cast<const float*>(smem + fq::RAW + 4096 + te * 4);
                pc = *reinterpret_cast<const float*>(smem + fq::RAW + 5120 + te * 4);
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(smem + xrd0 + mi * 2048);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(smem + xrd1 + mi * 2048);
                xa[mi] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(smem + wrd0 + j * 2048);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(smem + wrd1 + j * 2048);
                w[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            unsigned sraw[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sraw[k] = *reinterpret_cast<const unsigned*>(smem + sx_off + k * 64);
                sraw[8 + k] = *reinterpret_cast<const unsigned*>(smem + sw_base + (64 * (k >> 2) + 32 * ((k >> 1) & 1) + 4 * (k & 1)) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LN) {
                const float invk_ = 1.0f / (float)K;
                const float mean = keds_stat_value((keds_stat_t)(((unsigned long long)st_raw[1] << 32) | st_raw[0])) * invk_;
                const float ss = keds_stat_value((keds_stat_t)(((unsigned long long)st_raw[3] << 32) | st_raw[2]));
                const float rsd = rsqrtf(fmaxf(ss * invk_ - mean * mean, 0.f) + 1e-5f);
                *reinterpret_cast<f32x2*>(side + te * 8) = f32x2{rsd, -mean * rsd};      // (read in the epilogue, a dozen barriers from here)
                *reinterpret_cast<float*>(side + 2048 + te * 4) = pb;
                *reinterpret_cast<float*>(side + 3072 + te * 4) = pc;
            }
            const unsigned sh8 = 8 * ((unsigned)(te & 63) >> 4);   // 8 g
            sxa[0] = sxa[1] = swa[0] = swa[1] = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sxa[k >> 2] |= ((sraw[k] >> sh8) & 0xFFu) << (8 * (k & 3));
                swa[k >> 2] |= ((sraw[8 + k] >> sh8) & 0xFFu) << (8 * (k & 3));
            }
        }
        // K-tiles 0 | 1 | [2j, 2j + 1] | np - 2 | np - 1
#ifdef KEDS_FQ_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        KEDS_FQ_TS(2)
        // (step 0 syncs in mode 2: K-tile 1 landed behind the wait that opened the tile; what the step needs is buffer 0 free of
        // every wave's K-tile-0 reads -- and a vmcnt(0) here would sit out the previous epilogue's stores after all)
        KEDS_FQ_STEP(true, xa, sxa, swa, xb, sxb, swb, 1, 2, true, 2, true)
        KEDS_FQ_STEP(false, xb, sxb, swb, xa, sxa, swa, 0, 1, true, 3, true)
        for (int p = 2; p + 2 < np && !(KEDS_FQ_ABL & 2); p += 2) {
            KEDS_FQ_STEP(false, xa, sxa, swa, xb, sxb, swb, 1, 1, true, p + 2, true)
            KEDS_FQ_STEP(false, xb, sxb, swb, xa, sxa, swa, 0, 1, true, p + 3, true)
        }
        // the last two K-steps.  FOLD: their DMA slots carry the NEXT tile's K-tiles 0 and 1 (and its raw side data goes out in
        // front of them): K-tile np - 2's fragments are in registers when step np - 2 starts, so buffer 0 is free behind that step's
        // barrier, and buffer 1 behind the last step's (mode 2: every wave's reads of K-tile np - 1 are done; no wait for pieces).
        // Round 4 issued these 34 requests between the K-loop and the epilogue -- 3.6 k cycles per tile with the matrix pipe idle
        // (profiles/r04_fp8_tile_phases.txt).  The fp16-residual epilogue keeps that order: its residual chunks must leave ahead of the
        // pieces (below); folded, out-proj gains 3 % alone and the step nothing (profiles/r06_fp8_tile_switch_ab.txt).  A workgroup on its last tile re-requests its own tile's K-tiles (same code path, 132 KB
        // once per launch) and waits for them before it ends.
        constexpr bool FOLD = EPI != 4;
        if constexpr (FOLD) {
            if (more) {
                side_load(nm0, nn0);
                point_at(nm0, nn0);
            }
            KEDS_FQ_STEP(false, xa, sxa, swa, xb, sxb, swb, 1, 1, true, 0, true)
            KEDS_FQ_STEP(false, xb, sxb, swb, xa, sxa, swa, 0, 2, true, 1, false)
        } else {
            KEDS_FQ_STEP(false, xa, sxa, swa, xb, sxb, swb, 1, 1, false, 0, true)
            KEDS_FQ_STEP(false, xb, sxb, swb, xa, sxa, swa, 0, 0, false, 0, false)
        }

        KEDS_FQ_TS(3)
        // (the lane coordinates through an opaque move: everything the epilogue derives from them is otherwise loop invariant, and
        // hoisted out of the tile loop it is ~60 registers the K-loop does not have)
        const int le = lane_now();
        const int ge = le >> 4, ce = le & 15;
        // fp16 residual epilogue: the tile's residual chunks are requested HERE, in front of the next tile's 34 DMA pieces (vmcnt
        // retires in order: behind them a chunk waits for all of them, and loaded one by one between the stores of the epilogue --
        // the compiler cannot move a load above a store to the same array -- each chunk pays its own round trip: a cold
        // residual tile cost this kernel 15-20 us per launch against 5-7 for the 8-wave kernel, tools/fp8_cold_matrix.py).
        // The fragment registers are dead by now, but all 32 chunks (128 registers) beside a read-back quarter and the carry do
        // not fit (the allocator loaded chunks straight into accumulator AGPRs).
        // One 64-column half (16 chunks, 64 registers) goes out here, the other right behind the next tile's requests.
        [[maybe_unused]] f16x8 rpre[2][8][2];
        auto resid_request = [&](int h) {               // buffer loads: tile base in SGPRs, one lane offset, a scalar per chunk
            const auto ors = make_rs(reinterpret_cast<const f16_t*>(out) + (size_t)m0 * N + n0);
            const int voff = ((128 * wm + ce) * N + 128 * wn2 + 8 * ge) * 2;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    rpre[h][mi][pp] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ors, voff, (16 * mi * N + 64 * h + 32 * pp) * 2, 0));
        };
        if constexpr (EPI == 4) resid_request(0);
        if constexpr (!FOLD) {
            if (more) {
                // every wave has issued its last fragment reads (they completed before its last step's MFMAs could start): both
                // buffers are free for the next tile's K-tiles 0, 1
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                side_load(nm0, nn0);
                point_at(nm0, nn0);
#pragma unroll
                for (int q = 0; q < 16; ++q) issue(0, q);
                issue_scales(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) issue(1, q);
                issue_scales(1);
            }
        }
        if constexpr (EPI == 4) resid_request(1);
        KEDS_FQ_TS(4)
        // ---- epilogue: the 8-wave kernel's (its wave column 2 wn2 + h), one row group of 16 accumulator registers at a time
        KEDS_QUAD_DRAIN
        constexpr bool RES = EPI == 3 || EPI == 4;      // (their row partials meet the other three wave columns' in LDS)
        float* red = reinterpret_cast<float*>(smem + fq::SIDE0);
        MxSide sd;
#define KEDS_FQ_EPG(h, mi)                                                                                               \
    {                                                                                                                    \
        f32x4 av[4];                                                                                                     \
        KEDS_QUAD_READ_G##h##mi(av)                                                                                      \
        if constexpr (!(KEDS_FQ_ABL & 1))                                                                                \
            mx_epilogue_rows<EPI, EPI == 4, RES ? 4 : 2>(av, mi, sd, side, out, m0, n0, N, wm, 2 * wn2 + h, ge, ce, aux, aux2, qout,   \
                                                         qscale, q_pad, rpre[h][mi], red);                               \
    }
#define KEDS_FQ_EPH(h)                                                                                                   \
    mx_side<EPI>(sd, side, bias, n0, 2 * wn2 + h, ge);                                                                   \
    KEDS_FQ_EPG(h, 0) KEDS_FQ_EPG(h, 1) KEDS_FQ_EPG(h, 2) KEDS_FQ_EPG(h, 3)                                              \
    KEDS_FQ_EPG(h, 4) KEDS_FQ_EPG(h, 5) KEDS_FQ_EPG(h, 6) KEDS_FQ_EPG(h, 7)
        KEDS_FQ_EPH(0)
        KEDS_FQ_EPH(1)
#undef KEDS_FQ_EPH
#undef KEDS_FQ_EPG
        if constexpr (RES && !(KEDS_FQ_ABL & 1)) {      // row t: the four wave columns' partials (the side areas: unused by these epilogues)
            __syncthreads();
            const f32x2* rr = reinterpret_cast<const f32x2*>(red) + tid;
            const f32x2 p0 = rr[0], p1 = rr[256], p2 = rr[512], p3 = rr[768];
            if (!(KEDS_FQ_ABL & 4))
                keds_stat_add(reinterpret_cast<keds_stat_t*>(aux) + 2 * (size_t)(m0 + tid), (p0[0] + p1[0]) + (p2[0] + p3[0]),
                              (p0[1] + p1[1]) + (p2[1] + p3[1]));
        }
#ifdef KEDS_FQ_STAMP
        if constexpr (EPI == 1) {
            if (it == 1) {
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                if (lane == 0 && qout) {
                    unsigned long long* o = reinterpret_cast<unsigned long long*>(qout) + ((size_t)blockIdx.x * 4 + wave) * 8;
                    o[0] = ts[1] - ts[0]; o[1] = ts[2] - ts[1]; o[2] = ts[3] - ts[2]; o[3] = ts[4] - ts[3]; o[4] = t5 - ts[4]; o[5] = ts[0]; o[6] = t5;
                }
            }
        }
#endif
        if (!more) break;
        id = nid;
        m0 = nm0;
        n0 = nn0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the last tile's re-requested K-tiles: no LDS-DMA write may outlive the workgroup)
#undef KEDS_FQ_TS
#undef make_rs
}

}  // namespace

int g_fp8_debug = 0;   // timing-only ablations of gemm_mxfp8_kernel
extern "C" int keds_mxfp8_debug(int variant) {
    g_fp8_debug = variant;
    return KEDS_OK;
}

extern "C" size_t keds_mxfp8_scale_bytes(int rows_pad, int K) {
    if (rows_pad <= 0 || K <= 0 || K % 128) return 0;
    return (size_t)(K / 128) * rows_pad * 4;
}

extern "C" int keds_quantize_mxfp8(const void* x, int x_is_bf16, int rows, int K, int rows_pad, void* q, void* scales,
                                   void* stream) {
    KEDS_REQUIRE(x && q && scales && rows > 0, "keds_quantize_mxfp8: bad argument");
    KEDS_REQUIRE(K % 128 == 0 && K >= 128 && rows_pad >= rows, "keds_quantize_mxfp8: K %% 128 == 0 and rows_pad >= rows");
    hipStream_t st = (hipStream_t)stream;
    if (x_is_bf16)
        quantize_mxfp8_kernel<true><<<(rows + 3) / 4, 256, 0, st>>>(x, rows, K, rows_pad, (unsigned char*)q, (unsigned char*)scales);
    else
        quantize_mxfp8_kernel<false><<<(rows + 3) / 4, 256, 0, st>>>(x, rows, K, rows_pad, (unsigned char*)q, (unsigned char*)scales);
    return keds_check_launch("quantize_mxfp8_kernel");
}

namespace {
template <int EPI, int DBG>
int launch_mxfp8(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad, const float* bias,
                 void* out, int M, int N, int K, float* aux, float* aux2, void* qout, void* qscale, int q_pad, hipStream_t st) {
    if (int rc = keds_func_lds_once((const void*)gemm_mxfp8_kernel<EPI, DBG>, LDS_BYTES, "gemm_mxfp8_kernel")) return rc;
    const int m_tiles = M / TM, n_tiles = N / TN;
    KEDS_LAUNCH((gemm_mxfp8_kernel<EPI, DBG>), m_tiles * n_tiles, 512, LDS_BYTES, st,
                (const unsigned char*)Aq, (const unsigned char*)As, (const unsigned char*)Wq, (const unsigned char*)Ws, bias, out, M, N, K,
                n_tiles, m_pad, n_pad, aux, aux2, (unsigned char*)qout, (unsigned char*)qscale, q_pad);
    return keds_check_launch("gemm_mxfp8_kernel");
}

// the 4-wave persistent kernel: K-tiles in pairs (K % 256 == 0), at least four
bool fp8_quad_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = keds_exp_env("KEDS_FP8_QUAD");
        on = !(e && e[0] == '0');
    }
    return on != 0;
}
template <int EPI>
int launch_mxfp8_quad(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad, const float* bias,
                      void* out, int M, int N, int K, float* aux, float* aux2, void* qout, void* qscale, int q_pad, hipStream_t st) {
    if (int rc = keds_func_lds_once((const void*)gemm_mxfp8_quad_kernel<EPI>, fq::LDS_BYTES, "gemm_mxfp8_quad_kernel")) return rc;
    const int m_tiles = M / TM, n_tiles = N / TN, ntiles = m_tiles * n_tiles;
    int cus = keds_device_cus();
    if (cus > 256) cus = 256;
    cus &= ~7;                                    // whole XCD groups: workgroup b and its tile ids b, b + grid, ... share an XCD label
    // (A/B, round 4: the residual epilogues with one tile per workgroup, so that the hardware deals the tiles and the side lane's
    // small launches slot in between them: the GEMM class takes 0.5 ms more per step, the gaps in front of the attention launches
    // shrink by as much -- 14.87-15.06 against 14.92-15.00 ms per step, four alternating runs on one box)
    const int grid = (cus >= 8 && ntiles > cus) ? cus : ntiles;
    KEDS_LAUNCH((gemm_mxfp8_quad_kernel<EPI>), grid, 256, fq::LDS_BYTES, st,
                (const unsigned char*)Aq, (const unsigned char*)As, (const unsigned char*)Wq, (const unsigned char*)Ws, bias, out, M, N, K,
                n_tiles, m_pad, n_pad, aux, aux2, (unsigned char*)qout, (unsigned char*)qscale, q_pad, ntiles);
    return keds_check_launch("gemm_mxfp8_quad_kernel");
}
}  // namespace

extern "C" int keds_gemm_mxfp8_ex(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad,
                                  const float* bias, void* out, int M, int N, int K, int epilogue, float* aux, float* aux2,
                                  void* qout, void* qscale, int q_pad, void* stream) {
    KEDS_REQUIRE(Aq && As && Wq && Ws, "keds_gemm_mxfp8: null pointer");
    KEDS_REQUIRE(M > 0 && M % TM == 0 && N > 0 && N % TN == 0, "keds_gemm_mxfp8: M and N must be multiples of 256 (M=%d N=%d)", M, N);
    KEDS_REQUIRE(K % TKB == 0 && K >= 2 * TKB, "keds_gemm_mxfp8: K=%d must be a multiple of 128, >= 256", K);
    KEDS_REQUIRE(m_pad >= M && n_pad >= N && m_pad % 4 == 0 && n_pad % 4 == 0, "keds_gemm_mxfp8: bad scale row padding");
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_GEMM, st, /*lazy: KEDS_LAUNCH binds the pair*/ true);
    prof.work(2.0 * M * N * K);
    // keds_mxfp8_debug(16): the 8-wave kernel whatever the shape (the bit-identity test's reference).  The fp32-residual epilogue
    // (3: API only, the towers keep their stream in fp16) stays on the 8-wave kernel: beside its 256-bit residual chunks the
    // 4-wave K-loop has no registers left
    const bool quad = g_fp8_debug == 0 && fp8_quad_enabled() && K % (2 * TKB) == 0 && K >= 4 * TKB;
#define KEDS_FP8_GO(E, D)                                                                                                         \
    {                                                                                                                             \
        if constexpr ((D) == 0 && (E) != 3)                                                                                        \
            if (quad) return launch_mxfp8_quad<E>(Aq, As, m_pad, Wq, Ws, n_pad, bias, out, M, N, K, aux, aux2, qout, qscale, q_pad, st); \
        return launch_mxfp8<E, D>(Aq, As, m_pad, Wq, Ws, n_pad, bias, out, M, N, K, aux, aux2, qout, qscale, q_pad, st);              \
    }
    switch (epilogue) {
        case KEDS_FP8_EPI_BIAS_BF16:
            KEDS_REQUIRE(out != nullptr, "keds_gemm_mxfp8: null output");
            if (g_fp8_debug == 1) KEDS_FP8_GO(0, 1)
            if (g_fp8_debug == 2) KEDS_FP8_GO(0, 2)
            if (g_fp8_debug == 3) KEDS_FP8_GO(0, 3)
            if (g_fp8_debug == 4) KEDS_FP8_GO(0, 4)
            KEDS_FP8_GO(0, 0)
        case KEDS_FP8_EPI_LN_BIAS_BF16:
            KEDS_REQUIRE(out && bias && aux, "keds_gemm_mxfp8: LN epilogue needs out, bias = [bias' | csum] and row statistics");
            KEDS_FP8_GO(1, 0)
        case KEDS_FP8_EPI_LN_QGELU_MX:
            KEDS_REQUIRE(bias && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: LN+QuickGELU MX epilogue arguments");
            KEDS_FP8_GO(2, 0)
        case KEDS_FP8_EPI_RESID_STATS_MX:
            KEDS_REQUIRE(out && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: residual MX epilogue arguments");
            KEDS_FP8_GO(3, 0)
        case KEDS_FP8_EPI_RESID_STATS_MX_H:
            KEDS_REQUIRE(out && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: residual MX epilogue arguments");
            KEDS_FP8_GO(4, 0)
        default: keds_set_error("keds_gemm_mxfp8: unknown epilogue %d", epilogue); return KEDS_E_ARG;
    }
#undef KEDS_FP8_GO
}

extern "C" int keds_gemm_mxfp8(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad,
                               const float* bias, void* out, int M, int N, int K, void* stream) {
    return keds_gemm_mxfp8_ex(Aq, As, m_pad, Wq, Ws, n_pad, bias, out, M, N, K, KEDS_FP8_EPI_BIAS_BF16, nullptr, nullptr, nullptr,
                              nullptr, 0, stream);
}

extern "C" int keds_fold_layernorm_mxfp8(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                         int n_pad, void* wq, void* wscale, float* bias_csum, void* stream) {
    KEDS_REQUIRE(W && wq && wscale && bias_csum && N > 0, "keds_fold_layernorm_mxfp8: bad argument");
    KEDS_REQUIRE((gamma == nullptr) == (beta == nullptr), "keds_fold_layernorm_mxfp8: gamma and beta come together");
    KEDS_REQUIRE(K % 128 == 0 && K >= 128 && n_pad >= N, "keds_fold_layernorm_mxfp8: K %% 128 == 0 and n_pad >= N");
    fold_quantize_mxfp8_kernel<<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(W, bias, gamma, beta, N, K, n_pad, (unsigned char*)wq,
                                                                            (unsigned char*)wscale, bias_csum);
    return keds_check_launch("fold_quantize_mxfp8_kernel");
}
