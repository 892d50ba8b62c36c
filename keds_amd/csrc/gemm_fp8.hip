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
    // ---- epilogue: lane (g,c) owns rows m0 + 128*wm + 16*mi + c, columns n0 + 64*wn + 32*pp + 8*g + 0..7.  The four lanes
    // g = 0..3 of a row hold exactly one 32-column MX block per pp, so block amax / row sums are two xor-shuffles.
    float rstd[8], nmr[8];
    if constexpr (EPI == 1 || EPI == 2) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + 128 * wm + 16 * mi + c;
            const f32x2 cf = *reinterpret_cast<const f32x2*>(smem + SIDE_OFF + (128 * wm + 16 * mi + c) * 8);
            rstd[mi] = cf[0];
            nmr[mi] = cf[1];
            if (aux2 && n0 == 0 && wn == 0 && g == 0) keds_stat_zero(reinterpret_cast<keds_stat_t*>(aux2) + 2 * (size_t)m);
        }
    }
    float rs[8], rss[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) rs[mi] = rss[mi] = 0.f;
    [[maybe_unused]] uint2 mxk[8];
    [[maybe_unused]] int mxe[8];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int n = n0 + 64 * wn + 32 * pp + 8 * g;
        f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0, c0 = b0, c1 = b0;
        if constexpr (EPI == 1 || EPI == 2) {
            const char* sb = smem + SIDE_OFF + 2048 + (64 * wn + 32 * pp + 8 * g) * 4;
            b0 = *reinterpret_cast<const f32x4*>(sb);
            b1 = *reinterpret_cast<const f32x4*>(sb + 16);
            c0 = *reinterpret_cast<const f32x4*>(sb + 1024);
            c1 = *reinterpret_cast<const f32x4*>(sb + 1024 + 16);
        } else if (bias) {
            b0 = *reinterpret_cast<const f32x4*>(bias + n);
            b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + 128 * wm + 16 * mi + c;           // M is a multiple of 256: every row is valid
            f32x4 v0, v1;
            if constexpr (EPI == 1 || EPI == 2) {
                v0 = acc[2 * pp][mi] * rstd[mi] + (c0 * nmr[mi] + b0);
                v1 = acc[2 * pp + 1][mi] * rstd[mi] + (c1 * nmr[mi] + b1);
            } else {
                v0 = acc[2 * pp][mi] + b0;
                v1 = acc[2 * pp + 1][mi] + b1;
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
                    const f16x8 r = *o;
                    v0 += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                    v1 += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
                    *o = f16x8{(f16_t)v0[0], (f16_t)v0[1], (f16_t)v0[2], (f16_t)v0[3],
                               (f16_t)v1[0], (f16_t)v1[1], (f16_t)v1[2], (f16_t)v1[3]};
                }
                rs[mi] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                rss[mi] += ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) +
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
                    mxk[mi] = pk;
                    mxe[mi] = e;
                } else {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(mxk[mi].x, pk.x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(mxk[mi].y, pk.y, false, false);
                    const int blk = g & 1, half = g >> 1;          // after the swap: this lane's block and 16-column half
                    const unsigned qoff = (unsigned)((size_t)(m - m0) * N + (64 * wn + 32 * blk + 16 * half));
                    if constexpr (EPI == 2)       // MLP hidden (MXFP8): read once by c_proj
                        keds_store16<KEDS_ST_FP8_MX>(u32x4{s0[0], s1[0], s0[1], s1[1]}, qout + (size_t)m0 * N + n0, qoff);
                    else                           // MXFP8 copy of the residual stream: the next GEMM's A operand
                        keds_store16<KEDS_ST_FP8_MXR>(u32x4{s0[0], s1[0], s0[1], s1[1]}, qout + (size_t)m0 * N + n0, qoff);
                    if (g == 0)
                        *reinterpret_cast<unsigned short*>(qscale + mx_scale_index((n0 + 64 * wn) >> 5, m, q_pad)) =
                            (unsigned short)((mxe[mi] + 127) | ((e + 127) << 8));
                }

            }
        }
    }
    if constexpr (EPI == 3 || EPI == 4) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + 128 * wm + 16 * mi + c;
            float a = rs[mi], b2 = rss[mi];
            a = rows_sum(a);
            b2 = rows_sum(b2);
            if (g == 0) keds_stat_add(reinterpret_cast<keds_stat_t*>(aux) + 2 * (size_t)m, a, b2);
        }
    }
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
    gemm_mxfp8_kernel<EPI, DBG><<<m_tiles * n_tiles, 512, LDS_BYTES, st>>>(
        (const unsigned char*)Aq, (const unsigned char*)As, (const unsigned char*)Wq, (const unsigned char*)Ws, bias, out, M, N, K,
        n_tiles, m_pad, n_pad, aux, aux2, (unsigned char*)qout, (unsigned char*)qscale, q_pad);
    return keds_check_launch("gemm_mxfp8_kernel");
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
    KedsProfScope prof(KEDS_PROF_GEMM, st);
    prof.work(2.0 * M * N * K);
#define KEDS_FP8_GO(E, D) return launch_mxfp8<E, D>(Aq, As, m_pad, Wq, Ws, n_pad, bias, out, M, N, K, aux, aux2, qout, qscale, q_pad, st)
    switch (epilogue) {
        case KEDS_FP8_EPI_BIAS_BF16:
            KEDS_REQUIRE(out != nullptr, "keds_gemm_mxfp8: null output");
            if (g_fp8_debug == 1) KEDS_FP8_GO(0, 1);
            if (g_fp8_debug == 2) KEDS_FP8_GO(0, 2);
            if (g_fp8_debug == 3) KEDS_FP8_GO(0, 3);
            if (g_fp8_debug == 4) KEDS_FP8_GO(0, 4);
            KEDS_FP8_GO(0, 0);
        case KEDS_FP8_EPI_LN_BIAS_BF16:
            KEDS_REQUIRE(out && bias && aux, "keds_gemm_mxfp8: LN epilogue needs out, bias = [bias' | csum] and row statistics");
            KEDS_FP8_GO(1, 0);
        case KEDS_FP8_EPI_LN_QGELU_MX:
            KEDS_REQUIRE(bias && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: LN+QuickGELU MX epilogue arguments");
            KEDS_FP8_GO(2, 0);
        case KEDS_FP8_EPI_RESID_STATS_MX:
            KEDS_REQUIRE(out && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: residual MX epilogue arguments");
            KEDS_FP8_GO(3, 0);
        case KEDS_FP8_EPI_RESID_STATS_MX_H:
            KEDS_REQUIRE(out && aux && qout && qscale && q_pad >= M, "keds_gemm_mxfp8: residual MX epilogue arguments");
            KEDS_FP8_GO(4, 0);
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
