// Multi-head self-attention core: out = softmax(q k^T / sqrt(64) [+ causal mask]) v on a packed
// qkv buffer.  Reference: nn.MultiheadAttention inside ResidualAttentionBlock
// (src/model/model.py:309,319-321), causal mask src/model/model.py:543-549.
//
// gfx950 design.  Sequences are short (257 / 77 tokens), so one workgroup owns one (batch, head):
// all K rows and V^T stay resident in LDS (74 KB at S=257 -> 2 workgroups per CU) and no online
// softmax is needed.  Each wave takes 32 queries at a time (two 16-query MFMA tiles that share every K / V^T
// fragment read: the loop is LDS-read bound otherwise).  A start-time phase shift between the two workgroups of a
// CU was tried and measured neutral: staging and compute already overlap across workgroups.  Scores are computed "swapped",
// S^T = K . Q^T, so a lane holds ONE query column and 4 keys per 16-key tile: the row max / sum
// are in-lane reductions plus two xor-shuffles, and the probabilities, packed to bf16 in
// registers, are already the B operand of O^T = V^T . P^T (the MFMA k-slot order is permuted to
// match: k-slot (g, j) of step u is key 32u + 16*(j>>2) + 4g + (j&3); V^T is staged in that order).
#include "keds_common.h"
#include <math.h>
#include <hip/hip_ext.h>

namespace {

constexpr int DH = 64;

template <int NKT>
struct AttnCfg {
    static constexpr int KEYS = NKT * 16;
    static constexpr int K_BYTES = KEYS * DH * 2;          // [key][64] bf16, 128-byte rows, swizzled
    static constexpr int VT_ROW = KEYS * 2 + 16;           // bytes per dh row of V^T (+16: odd chunk stride)
    static constexpr int VT_BYTES = DH * VT_ROW;
    static constexpr int LDS = K_BYTES + VT_BYTES;
    static_assert(NKT % 2 == 0, "PV consumes key tiles in pairs");
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

struct AttnCtx {
    const bf16_t* base;   // qkv of (b, h): row stride ld
    bf16_t* out;          // out of (b, h): row stride d
    const char* k_lds;
    const char* vt_lds;
    int S, q_limit, ld, d, g, c;
    // fp8 towers: rows < q8_rows of the [B*S, d] output go out as MXFP8 (e4m3 + e8m0 per 32 columns) INSTEAD of bf16
    unsigned char* q8;    // element (row, col) at q8[row*d + col]; nullptr = bf16 everywhere
    unsigned char* s8;    // scale dwords [d/128][q8_rows]
    int q8_rows, row0, col0;   // first global row of this sample, first column of this head
    const char* tail;          // TAIL kernels: [K row of the last key: 64 bf16, unswizzled][its V row: 64 bf16]
};

// NQ consecutive 16-query tiles starting at tile qt0, for one wave.
// TAIL: the sequence is 16*NKT + 1 keys; the MFMA tiles cover the first 16*NKT and the last key is a rank-1 VALU update
// (its score from the lane's 16 query dims + a reduction over the four lanes of the query, its P.V term 16 FMAs per tile)
// instead of two more key tiles of which 31 of 32 columns would be padding.
template <int NKT, bool CAUSAL, int NFULL, int DBG, int NQ, bool TAIL = false>
__device__ __forceinline__ void attn_tiles(const AttnCtx& cx, int qt0) {
    using C = AttnCfg<NKT>;
    const int g = cx.g, c = cx.c, S = cx.S;
    const float sl2 = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    int qidx[NQ];
    bf16x8 qf[NQ][2];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        qidx[t] = (qt0 + t) * 16 + c;
        const int qrow = qidx[t] < S ? qidx[t] : S - 1;
        const bf16_t* qp = cx.base + (size_t)qrow * cx.ld + 8 * g;
        qf[t][0] = *reinterpret_cast<const bf16x8*>(qp);
        qf[t][1] = *reinterpret_cast<const bf16x8*>(qp + 32);
    }
    // ---- S^T tiles
    f32x4 sc[NQ][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int key = kt * 16 + c;
        const char* kr = cx.k_lds + key * 128;
        const int f = (key >> 1) & 7;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(kr + ((g ^ f) << 4));
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(kr + (((4 + g) ^ f) << 4));
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (DBG != 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[t][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[t][1], acc, 0, 0, 0);
            } else {
                acc[0] = (float)qf[t][0][kt & 7];
            }
            sc[t][kt] = acc;
        }
        if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // bound how far LDS reads are hoisted (VGPR pressure)
    }
    // ---- TAIL: score of the last key for the lane's query (replicated over the four g lanes after the reduction)
    [[maybe_unused]] float tsc[NQ];
    if constexpr (TAIL) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(cx.tail + g * 16);
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(cx.tail + (4 + g) * 16);
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (float)qf[t][0][j] * (float)k0[j] + (float)qf[t][1][j] * (float)k1[j];
            tsc[t] = rows_sum(a);
        }
    }
    // ---- mask, row max, exp, row sum (lane holds query qidx[t], keys kt*16 + 4g + r)
    float inv[NQ];
    [[maybe_unused]] float tp[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sc[t][kt][r];
                if (CAUSAL || kt >= NFULL) {      // resolved at compile time once the kt loop is unrolled
                    const int kidx = kt * 16 + 4 * g + r;
                    const bool ok = kidx < S && (!CAUSAL || kidx <= qidx[t]);
                    v = ok ? v : -INFINITY;
                    sc[t][kt][r] = v;
                }
                mx = fmaxf(mx, v);
            }
        mx = rows_max(mx);
        if constexpr (TAIL) mx = fmaxf(mx, tsc[t]);
        const float nmx = -mx * sl2;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32: two elements per VALU issue); v_exp_f32 stays per element
        const f32x2 vs = f32x2{sl2, sl2}, vn = f32x2{nmx, nmx};
        f32x2 vsum = f32x2{0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x2 e = f32x2{sc[t][kt][2 * hh], sc[t][kt][2 * hh + 1]};
                if constexpr (DBG != 3) {
                    e = pk_fma(e, vs, vn);
                    e[0] = __builtin_amdgcn_exp2f(e[0]);   // exp2(-inf) = 0
                    e[1] = __builtin_amdgcn_exp2f(e[1]);
                }
                sc[t][kt][2 * hh] = e[0];
                sc[t][kt][2 * hh + 1] = e[1];
                vsum += e;
            }
        }
        float sum = vsum[0] + vsum[1];
        sum = rows_sum(sum);
        if constexpr (TAIL) {
            const float e = __builtin_amdgcn_exp2f(tsc[t] * sl2 + nmx);
            sum += e;
            tp[t] = (float)(bf16_t)e;             // rounded like the probabilities the MFMA path multiplies
        }
        inv[t] = 1.0f / sum;
    }
    // ---- O^T = V^T . P^T
    f32x4 o[NQ][4];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
        bf16x8 pf[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const f32x4 p0 = sc[t][2 * u], p1 = sc[t][2 * u + 1];
            pf[t] = bf16x8{(bf16_t)p0[0], (bf16_t)p0[1], (bf16_t)p0[2], (bf16_t)p0[3],
                           (bf16_t)p1[0], (bf16_t)p1[1], (bf16_t)p1[2], (bf16_t)p1[3]};
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if constexpr (DBG != 4) {
                const bf16x8 a =
                    *reinterpret_cast<const bf16x8*>(cx.vt_lds + (dt * 16 + c) * C::VT_ROW + (4 * u + g) * 16);
#pragma unroll
                for (int t = 0; t < NQ; ++t) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[t], o[t][dt], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < NQ; ++t) o[t][dt][0] += (float)pf[t][dt];
            }
        }
        if (u % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TAIL) {      // + p_last * V[last key][16 dt + 4 g + r]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(cx.tail + 128 + (16 * dt + 4 * g) * 2);
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][dt][r] += tp[t] * (float)v[r];
        }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        if (qidx[t] < cx.q_limit) {
            const int row = cx.row0 + qidx[t];
            if (cx.q8 && row < cx.q8_rows) {
                // lane (g, c): head columns 16*dt + 4g + r; a 32-column MX block = dt in {2b, 2b+1} over the four g lanes
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const f32x4 v0 = o[t][2 * b2] * inv[t], v1 = o[t][2 * b2 + 1] * inv[t];
                    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    float amax = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
                    amax = rows_max(amax);
                    const int e = mx_block_exp(amax);
                    const uint2 pk = mx_pack8(v, e);
                    unsigned char* qp = cx.q8 + (size_t)row * cx.d + cx.col0 + 32 * b2 + 4 * g;
                    *reinterpret_cast<unsigned*>(qp) = pk.x;
                    *reinterpret_cast<unsigned*>(qp + 16) = pk.y;
                    if (g == 0) cx.s8[mx_scale_index((cx.col0 >> 5) + b2, row, cx.q8_rows)] = (unsigned char)(e + 127);
                }
            } else {
                bf16_t* op = cx.out + (size_t)qidx[t] * cx.d + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f32x4 v = o[t][dt] * inv[t];
                    *reinterpret_cast<bf16x4*>(op + dt * 16) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                }
            }
        }
    }
}

// DBG (timing-only ablations): 1 = no K/V staging, 2 = no QK^T MFMA/reads, 3 = no softmax math, 4 = no PV, 5 = no q loop
// CAUSAL: text tower mask.  NFULL: key tiles [0, NFULL) are known at compile time to lie entirely below S and
// need no mask (non-causal only) -- evaluating the mask for all 72 score registers cost half the loop's instructions.
template <int NKT, bool CAUSAL, int NFULL, int DBG = 0, bool NQ2 = (NKT == 18 && !CAUSAL)>
__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S,
                                                        int heads, int q_limit, unsigned char* __restrict__ q8,
                                                        unsigned char* __restrict__ s8, int q8_rows,
                                                        const int* __restrict__ seq_off) {
    using C = AttnCfg<NKT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* vt_lds = smem + C::K_BYTES;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int d = heads * DH;
    const int ld = 3 * d;  // qkv row stride (elements)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    // PACKED rows (keds_attention_packed: the text tower's captions end at different columns): sample b owns rows
    // [seq_off[b], seq_off[b + 1]) of qkv / out; S was the launch's upper bound and becomes the sample's own length
    int row0 = b * S;
    if (seq_off) {
        row0 = seq_off[b];
        S = seq_off[b + 1] - row0;
        q_limit = q_limit < S ? q_limit : S;
    }
    const bf16_t* base = qkv + (size_t)row0 * ld + h * DH;

    // ---- stage K (swizzled rows) and V^T (permuted key order); keys >= S are zero.
    // Work item = (4 consecutive keys, one 8-wide dh chunk): all global loads of a thread are issued before the
    // first LDS write (one HBM round trip instead of one per item), K goes in with ds_write_b128, and the four
    // keys of an item are adjacent in the permuted V^T row, so V^T goes in with 8-byte stores.
    if constexpr (DBG != 1) {
        constexpr int ITEMS = (C::KEYS / 4) * 8;
        constexpr int PER = (ITEMS + 255) / 256;
        bf16x8 kreg[PER][4], vreg[PER][4];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int id = tid + it * 256;
            const int quad = id >> 3, ch = id & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = quad * 4 + e;
                kreg[it][e] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                vreg[it][e] = kreg[it][e];
                if (id < ITEMS && key < S) {
                    const bf16_t* row = base + (size_t)key * ld + ch * 8;
                    kreg[it][e] = *reinterpret_cast<const bf16x8*>(row + d);
                    vreg[it][e] = *reinterpret_cast<const bf16x8*>(row + 2 * d);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int id = tid + it * 256;
            if (id >= ITEMS) continue;
            const int quad = id >> 3, ch = id & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = quad * 4 + e;
                *reinterpret_cast<bf16x8*>(k_lds + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[it][e];
            }
            // keys 4*quad .. 4*quad+3: u = key>>5, w = key&31, chunk (4u + ((w&15)>>2)), element 4*(w>>4) + (w&3)
            const int k0 = quad * 4;
            const int u = k0 >> 5, w = k0 & 31;
            const int pos = (4 * u + ((w & 15) >> 2)) * 16 + ((w >> 4) << 2) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<bf16x4*>(vt_lds + (ch * 8 + j) * C::VT_ROW + pos) =
                    bf16x4{vreg[it][0][j], vreg[it][1][j], vreg[it][2][j], vreg[it][3][j]};
        }
    }
    __syncthreads();

    // ---- queries: NQ = 2 tiles (32 queries) per step share every K / V^T fragment read from LDS (the loop is
    // LDS-bandwidth bound: 72 ds_read_b128 per 16-query tile); an odd last tile runs alone on a rotating wave.
    const int nqt = DBG == 5 ? 0 : (q_limit + 15) >> 4;      // only the first q_limit query rows are computed and stored
    AttnCtx cx{base, out + (size_t)row0 * d + h * DH, k_lds, vt_lds, S, q_limit, ld, d, g, c, q8, s8, q8_rows, row0, h * DH};
    if constexpr (NQ2) {
        const int npair = nqt >> 1;
        for (int qp = wave; qp < npair; qp += 4) attn_tiles<NKT, CAUSAL, NFULL, DBG, 2>(cx, 2 * qp);
        if ((nqt & 1) && wave == ((blockIdx.x + npair) & 3)) attn_tiles<NKT, CAUSAL, NFULL, DBG, 1>(cx, nqt - 1);
    } else {
        for (int qt = wave; qt < nqt; qt += 4) attn_tiles<NKT, CAUSAL, NFULL, DBG, 1>(cx, qt);
    }
}

// ---- S = 16 * NKT + 1, non-causal (ViT-L/14: 257 = 16 * 16 + 1 tokens) ------------------------------------------------
// The generic kernel pads 257 keys to 288 and 257 queries to 272: a fifth of its MFMA, exp and LDS-read work is padding, and
// the 17th query tile lands on one wave as a third step where the others run two.  Here the 256 leading keys / queries are
// 16 key tiles x 8 query-tile pairs (two per wave: balanced), the last KEY is a rank-1 VALU update inside attn_tiles<TAIL>,
// and the last QUERY is one row of plain VALU work split over the four waves after their tile loops: wave w scores keys
// [64 w, 64 w + 64) (one per lane), the scores and then the probabilities cross waves through 2 KB of LDS, and wave w
// accumulates output dims [16 w, 16 w + 16) (lane = dim x key quarter, reduced over the quarters with permlane swaps).
constexpr int TAIL_LDS = 256 + 2 * 1056;      // last key's K and V rows | scores [257+] | probabilities [257+]

template <int NKT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void attention_tail1_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                 int heads, int q_limit) {
    using C = AttnCfg<NKT>;
    constexpr int S = 16 * NKT + 1, LAST = 16 * NKT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* vt_lds = smem + C::K_BYTES;
    char* tail = smem + C::LDS;
    float* sc_lds = reinterpret_cast<float*>(tail + 256);
    float* p_lds = sc_lds + 264;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int d = heads * DH;
    const int ld = 3 * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * DH;

    // ---- stage K (swizzled rows), V^T (permuted key order) of the 16 * NKT leading keys, and the last key's two rows
    {
        constexpr int ITEMS = (C::KEYS / 4) * 8;
        constexpr int PER = ITEMS / 256;
        static_assert(ITEMS % 256 == 0, "staging items must divide over the workgroup");
        bf16x8 kreg[PER][4], vreg[PER][4];
        bf16x8 treg = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int id = tid + it * 256;
            const int quad = id >> 3, ch = id & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16_t* row = base + (size_t)(quad * 4 + e) * ld + ch * 8;
                kreg[it][e] = *reinterpret_cast<const bf16x8*>(row + d);
                vreg[it][e] = *reinterpret_cast<const bf16x8*>(row + 2 * d);
            }
        }
        if (tid < 16) treg = *reinterpret_cast<const bf16x8*>(base + (size_t)LAST * ld + (tid < 8 ? d : 2 * d) + (tid & 7) * 8);
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int id = tid + it * 256;
            const int quad = id >> 3, ch = id & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = quad * 4 + e;
                *reinterpret_cast<bf16x8*>(k_lds + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[it][e];
            }
            const int k0 = quad * 4;
            const int u = k0 >> 5, w = k0 & 31;
            const int pos = (4 * u + ((w & 15) >> 2)) * 16 + ((w >> 4) << 2) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<bf16x4*>(vt_lds + (ch * 8 + j) * C::VT_ROW + pos) =
                    bf16x4{vreg[it][0][j], vreg[it][1][j], vreg[it][2][j], vreg[it][3][j]};
        }
        if (tid < 16) *reinterpret_cast<bf16x8*>(tail + tid * 16) = treg;
    }
    __syncthreads();

    const int ql = q_limit < LAST ? q_limit : LAST;               // query rows covered by the MFMA tiles
    const int nqt = DBG == 5 ? 0 : (ql + 15) >> 4;
    AttnCtx cx{base, out + (size_t)b * S * d + h * DH, k_lds, vt_lds, S, q_limit, ld, d, g, c, nullptr, nullptr, 0, b * S, h * DH, tail};
    const int npair = nqt >> 1;
    for (int qp = wave; qp < npair; qp += 4) attn_tiles<NKT, false, NKT, DBG, 2, true>(cx, 2 * qp);
    if ((nqt & 1) && wave == ((blockIdx.x + npair) & 3)) attn_tiles<NKT, false, NKT, DBG, 1, true>(cx, nqt - 1);
    if (q_limit <= LAST) return;                                   // kernel-uniform: nobody waits at the barriers below

    // ---- the last query row: scores of this wave's 64 keys (lane = key)
    const float sl2 = 0.125f * 1.4426950408889634f;
    const bf16_t* qrow = base + (size_t)LAST * ld;
    float s = 0.f;
    {
        const int key = 64 * wave + lane;
        const char* kr = k_lds + key * 128;
        const int f = (key >> 1) & 7;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qrow + 8 * ch);       // same address in every lane: one fetch
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + ((ch ^ f) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)qv[j] * (float)kv[j];
        }
        sc_lds[key] = s;
    }
    const float q_own = (float)qrow[lane];                                            // lane = head dim
    const float s_last = wave_sum(q_own * (float)reinterpret_cast<const bf16_t*>(tail)[lane]);
    __syncthreads();
    // every wave: softmax statistics over all 16 * NKT + 1 scores (lane reads keys lane, lane + 64, ...)
    float mx = s_last;
#pragma unroll
    for (int i = 0; i < LAST / 64; ++i) mx = fmaxf(mx, sc_lds[lane + 64 * i]);
    mx = wave_max(mx);
    const float nmx = -mx * sl2;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LAST / 64; ++i) sum += __builtin_amdgcn_exp2f(sc_lds[lane + 64 * i] * sl2 + nmx);
    sum = wave_sum(sum);
    const float e_last = __builtin_amdgcn_exp2f(s_last * sl2 + nmx);
    sum += e_last;
    p_lds[64 * wave + lane] = (float)(bf16_t)__builtin_amdgcn_exp2f(s * sl2 + nmx);   // bf16-rounded like the MFMA path's P
    __syncthreads();
    // P.V: lane = (dim 16 wave + c, key quarter g); chunk 4u + gg of a V^T row holds keys 32u + 4gg + {0..3}, 32u + 16 + 4gg + {0..3}
    {
        const int dim = 16 * wave + c;
        const char* vrow = vt_lds + dim * C::VT_ROW;
        float acc = 0.f;
#pragma unroll
        for (int uu = 0; uu < NKT / 8; ++uu) {
            const int u = g * (NKT / 8) + uu;                       // this quarter's 32-key steps
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(vrow + (4 * u + gg) * 16);
                const f32x4 pa = *reinterpret_cast<const f32x4*>(p_lds + 32 * u + 4 * gg);
                const f32x4 pb = *reinterpret_cast<const f32x4*>(p_lds + 32 * u + 16 + 4 * gg);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += pa[j] * (float)v[j] + pb[j] * (float)v[4 + j];
            }
        }
        acc = rows_sum(acc);
        acc += (float)(bf16_t)e_last * (float)reinterpret_cast<const bf16_t*>(tail + 128)[dim];
        if (g == 0) cx.out[(size_t)LAST * d + dim] = (bf16_t)(acc / sum);
    }
}

// ---- S = 257, non-causal, EIGHT waves per (batch, head) (round 2; the product kernel of the ViT-L/14 tower) ------------
// The 4-wave kernels above run three serial phases per wave (all S^T tiles -> softmax over 128 score registers -> P.V) at two
// waves per SIMD, and their ablations are additive: no staging -19 us, no QK^T -15, no exp -10, no PV -10 of 81 us -- nothing
// overlaps.  The softmax is the largest term: on a 16-lane SIMD a wave64 VALU instruction holds the issue port 4 cycles
// (v_exp_f32: 8), i.e. ~20 cycles per score register against 32 MFMA-pipe cycles per 16 of them.  This kernel trades the
// register-resident score matrix for occupancy and lets the hardware interleave the phases of different waves:
//   * 512 threads: wave w owns queries [32 w, 32 w + 32) and walks the 256 leading keys in eight 32-key tiles with
//     v_mfma_f32_32x32x16_bf16 (half the issue slots per flop of the 16x16x32 form): S^T tile = K tile . Q^T (4 MFMAs, the query
//     on the lane, 16 keys in registers), exp2 in place, P packed to bf16 IS the B operand of O^T += V^T tile . P^T (4 MFMAs).
//     ~110 VGPRs -> four waves per SIMD (two workgroups per CU), so one wave's exp / pack / LDS reads run under another's MFMAs.
//   * No running-max rescale.  The probabilities are taken relative to the row maximum of the FIRST key tile (m0): bf16 and
//     fp32 keep their relative precision at any exponent, so exp2(s - m0) is as good as exp2(s - max) as long as it cannot
//     overflow.  The lane tracks the true maximum on the side (v_max3, 1 cycle per score); if it exceeds m0 by more than 64
//     (log2 units) for any query of the block -- it never does on real activations -- the block is recomputed once with the now
//     known maximum.  Exact, branch-free in the common case.
//   * K and V go to LDS by LDS-DMA (global_load_lds_dwordx4, 8 per wave, no staging registers, no LDS-write issue): both
//     row-major [key][64] with a chunk swizzle applied on the SOURCE side; V^T fragments are read with ds_read_b64_tr_b16
//     (hardware transpose), so no V^T image is built (the 8-byte V^T writes of the 4-wave kernels were 8-way bank conflicted).
//   * key 256 is a rank-1 VALU update per block, query 256 one VALU row split over the eight waves after their blocks.
namespace s257 {
constexpr int K_OFF = 0, V_OFF = 32768, TAIL_OFF = 65536;    // tail: row 256's q | k | v (128 B each)
constexpr int SC_OFF = TAIL_OFF + 384;                        // last query: f32 scores [264]
constexpr int PART_OFF = SC_OFF + 264 * 4;                    // last query: per wave {max, sum, -, ..., partial output [64]}: 72 floats
constexpr int CNT_OFF = PART_OFF + 8 * 72 * 4;                // last query: arrival counter
constexpr int DUMP_OFF = CNT_OFF + 16;                        // 256 B nobody reads: target of the L2 prefetch (KEDS_ATTN_PREFETCH)
constexpr int LDS = DUMP_OFF + 256;                           // 69,552 B -> two workgroups per CU
}  // namespace s257

__device__ __forceinline__ float halves_max(float x) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// c + sum_j a[j] b[j] with v_dot2c_f32_bf16 (two products per instruction, fp32 accumulate)
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float c) {
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_pair(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// all-VALU wave reductions (DPP within the 16-lane rows, permlane swaps across them); __shfl_xor is six LDS round trips
#define KEDS_DPP_F(x, CTRL) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), (CTRL), 0xF, 0xF, false))
__device__ __forceinline__ float wave_sum_v(float x) {
    x += KEDS_DPP_F(x, 0xB1);     // lane ^ 1
    x += KEDS_DPP_F(x, 0x4E);     // lane ^ 2
    x += KEDS_DPP_F(x, 0x141);    // row_half_mirror: the other group of four
    x += KEDS_DPP_F(x, 0x140);    // row_mirror: the other half of the row
    return rows_sum(x);
}
__device__ __forceinline__ float wave_max_v(float x) {
    x = fmaxf(x, KEDS_DPP_F(x, 0xB1));
    x = fmaxf(x, KEDS_DPP_F(x, 0x4E));
    x = fmaxf(x, KEDS_DPP_F(x, 0x141));
    x = fmaxf(x, KEDS_DPP_F(x, 0x140));
    return rows_max(x);
}

// Q8 (fp8 towers): rows < q8_rows of the [B*S, d] output go out as MXFP8 (e4m3 + one e8m0 scale per 32 columns: q8 / s8
// as in attention_kernel) INSTEAD of bf16; a 32-dim output tile of a query is exactly one MX block, held by two lanes.
template <int DBG = 0, bool Q8 = false>
__global__ __launch_bounds__(512, 4) void attention_s257_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                int heads, int q_limit,
                                                                unsigned long long* __restrict__ stamp,
                                                                unsigned char* __restrict__ q8 = nullptr,
                                                                unsigned char* __restrict__ s8 = nullptr, int q8_rows = 0) {
    using namespace s257;
    constexpr int S = 257, LAST = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    [[maybe_unused]] unsigned long long t_in = 0, t_staged = 0, t_lastq = 0, t_loop = 0;
    if constexpr (DBG == 8) t_in = __builtin_amdgcn_s_memtime();
    if constexpr (DBG == 6 || DBG == 7) {     // phase-shift experiment: half of the first-round workgroups start ~half a lifetime late
        const bool late = DBG == 6 ? (blockIdx.x & 1) : ((blockIdx.x >> 8) & 1);
        if (blockIdx.x < 512 && late) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < 12000ull) __builtin_amdgcn_s_sleep(16);
        }
    }
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int d = heads * DH;
    const int ld = 3 * d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * DH;
    bf16_t* obase = out + (size_t)b * S * d + h * DH;
    const float sl2 = 0.125f * 1.4426950408889634f;

    // ---- row 256 (q, k, v: 3 x 128 B) through registers, this wave's 32 queries as B operands (k = head dims), then the
    // 256 leading K and V rows by LDS-DMA: piece = 8 rows = 1 KiB = one wave-instruction; LDS slot (row, c) holds source
    // chunk c ^ f(row).  Keys 0-127 are issued first and waited for alone: the first four key tiles run while 128-255 land.
    bf16x8 trow = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < 24) trow = *reinterpret_cast<const bf16x8*>(base + (size_t)LAST * ld + (tid >> 3) * d + (tid & 7) * 8);
    bf16x8 qf[4];
    {
        const bf16_t* qp = base + (size_t)(32 * wave + r) * ld + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
    if constexpr (DBG != 1) {
        const int prow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 8 * i;
            const int row = 8 * piece + prow;
            const int fk = (row >> 1) & 7, fv = ((row >> 1) & 1) << 2;
            const bf16_t* src = base + (size_t)row * ld;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + d + ((slot ^ fk) << 3)),
                                             (__attribute__((address_space(3))) void*)(smem + K_OFF + piece * 1024), 16, 0, KEDS_LD_ATTN_AUX);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * d + ((slot ^ fv) << 3)),
                                             (__attribute__((address_space(3))) void*)(smem + V_OFF + piece * 1024), 16, 0, KEDS_LD_ATTN_AUX);
        }
    }
    // tail image: [q row 256 | k row 256 | v row 256], 128 B each, unswizzled; the last-query arrival counter
    if (tid < 24) *reinterpret_cast<bf16x8*>(smem + TAIL_OFF + tid * 16) = trow;
    if (tid == 24) *reinterpret_cast<int*>(smem + CNT_OFF) = 0;
    if constexpr (DBG != 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything older than the last four DMA pieces
    __syncthreads();
    if constexpr (DBG == 8) t_staged = __builtin_amdgcn_s_memtime();
    const char* tq = smem + TAIL_OFF;
    const char* tk = smem + TAIL_OFF + 128;
    const char* tv = smem + TAIL_OFF + 256;

    // ---- the last query row, without a barrier: wave w takes keys [32 w, 32 w + 32) (wave 0 also key 256) and leaves
    // {its maximum, its sum, its P.V partial relative to that maximum} in LDS; the wave that arrives last (LDS counter)
    // combines the eight partials.  (As a phase behind three barriers this row cost 5-6 k of a wave's 25-30 k cycles.)
    auto last_query_partial = [&]() {
        float* rec = reinterpret_cast<float*>(smem + PART_OFF) + wave * 72;      // [m, l, -, -, -, -, -, -, acc[64]]
        const int key = 32 * wave + r;
        const int fk = (key >> 1) & 7;
        float sq = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int ch = 4 * hh + c4;
            sq = dot8(*reinterpret_cast<const bf16x8*>(tq + ch * 16),
                      *reinterpret_cast<const bf16x8*>(smem + K_OFF + key * 128 + ((ch ^ fk) << 4)), sq);
        }
        sq = halves_sum(sq);                                        // lanes r and r + 32: score of key 32 w + r
        float s256 = -INFINITY;
        if (wave == 0) {                                            // key 256: lanes 0-7 take one chunk each
            const float t = lane < 8 ? dot8(*reinterpret_cast<const bf16x8*>(tq + (lane & 7) * 16),
                                            *reinterpret_cast<const bf16x8*>(tk + (lane & 7) * 16), 0.f) : 0.f;
            s256 = wave_sum_v(t);
        }
        const float mw = fmaxf(wave_max_v(sq), s256);
        const float nmw = -mw * sl2;
        const float pk = (float)(bf16_t)__builtin_amdgcn_exp2f(__builtin_fmaf(sq, sl2, nmw));   // bf16-rounded like the MFMA path's P
        const float e256 = wave == 0 ? __builtin_amdgcn_exp2f(__builtin_fmaf(s256, sl2, nmw)) : 0.f;
        const float lw = wave_sum_v(hh == 0 ? __builtin_amdgcn_exp2f(__builtin_fmaf(sq, sl2, nmw)) : 0.f) + e256;
        // P.V over the wave's 32 keys: lane = (8-dim chunk ch, key subset sub): keys 32 w + sub + 8 i
        const int ch = lane & 7, sub = lane >> 3;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = sub + 8 * i;                              // key within the wave's slice = the lane that holds its p
            const float p = __uint_as_float(__builtin_amdgcn_ds_bpermute(kk << 2, __float_as_uint(pk)));
            const int key2 = 32 * wave + kk;
            const int fv = ((key2 >> 1) & 1) << 2;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + V_OFF + key2 * 128 + ((ch ^ fv) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += p * (float)v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = acc[j];
            x += KEDS_DPP_F(x, 0x128);                                     // row_ror:8 = lane ^ 8
            acc[j] = rows_sum(x);                                          // lanes ^ 16, ^ 32
        }
        if (lane < 8) {
            if (wave == 0) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(tv + ch * 16);
                const float p = (float)(bf16_t)e256;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += p * (float)v[j];
            }
            *reinterpret_cast<f32x4*>(rec + 8 + 8 * ch) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(rec + 8 + 8 * ch + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            if (lane == 0) *reinterpret_cast<f32x2*>(rec) = f32x2{mw, lw};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(reinterpret_cast<int*>(smem + CNT_OFF), 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == 7) {                                          // every partial is in LDS: lane = head dim
            const float* all = reinterpret_cast<const float*>(smem + PART_OFF);
            float M = all[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) M = fmaxf(M, all[w * 72]);
            float L = 0.f, o = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float f = __builtin_amdgcn_exp2f((all[w * 72] - M) * sl2);
                L += all[w * 72 + 1] * f;
                o += all[w * 72 + 8 + lane] * f;
            }
            const float val = o / L;
            const int row = b * S + LAST;
            if (Q8 && row < q8_rows) {                               // lane = head dim: one MX block per 32-lane half
                float amax = fabsf(val);
                amax = fmaxf(amax, KEDS_DPP_F(amax, 0xB1));
                amax = fmaxf(amax, KEDS_DPP_F(amax, 0x4E));
                amax = fmaxf(amax, KEDS_DPP_F(amax, 0x141));
                amax = fmaxf(amax, KEDS_DPP_F(amax, 0x140));
                {
                    auto a16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
                    amax = fmaxf(__uint_as_float(a16[0]), __uint_as_float(a16[1]));
                }
                const int e = mx_block_exp(amax);
                const float inv2 = e == -127 ? 0.f : __uint_as_float((unsigned)(127 - e) << 23);
                const float sv = fminf(fmaxf(val * inv2, -448.f), 448.f);
                const unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(sv, 0.f, 0u, false);
                q8[(size_t)row * d + h * DH + lane] = (unsigned char)(pk & 0xFFu);
                if ((lane & 31) == 0) s8[mx_scale_index((h * DH + lane) >> 5, row, q8_rows)] = (unsigned char)(e + 127);
            } else {
                obase[(size_t)LAST * d + lane] = (bf16_t)val;
            }
        }
    };

    const int ql = q_limit < LAST ? q_limit : LAST;
    const int nblk = DBG == 5 ? 0 : (ql + 31) >> 5;
    const bool do_last = q_limit > LAST && DBG != 9;
    if (wave < nblk) {
        // lane-constant LDS offsets: K fragment of k-step ks (row r of the tile), V^T blocks of d-tile 0 / 1
        const int fk = (r >> 1) & 7;
        int ka[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ka[ks] = K_OFF + r * 128 + (((2 * ks + hh) ^ fk) << 4);
        const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gi = (lane >> 4) & 1;
        const int fq = (q4 >> 1) & 1;                                   // V swizzle: chunk bit 2 flips on rows 2, 3 (mod 4)
        const int vrow = V_OFF + (4 * hh + q4) * 128 + (p4 & 1) * 8 + ((2 * gi + (p4 >> 1)) << 4);
        const int vb0 = vrow + ((fq ? 4 : 0) << 4), vb1 = vrow + ((fq ? 0 : 4) << 4);

        float mgiven = 0.f;
        bool have = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            f32x16 o0, o1;
#pragma unroll
            for (int i = 0; i < 16; ++i) o0[i] = 0.f, o1[i] = 0.f;
            float nm = 0.f, lsum = 0.f, mref = 0.f;
            if (have) {            // recompute pass (rare): the exact row maximum first, scores only
                float mrun = -INFINITY;
#pragma unroll 1
                for (int kt = 0; kt < 8; ++kt) {
                    f32x16 sc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(smem + ka[ks] + kt * 4096), qf[ks], sc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mrun = fmaxf(mrun, sc[i]);
                }
                mgiven = fmaxf(halves_max(mrun), mgiven);       // mgiven came in as the last key's score
            }
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && attempt == 0) {                           // keys 128-255 (kernel-uniform count of barriers:
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // a recomputing wave does not come here again)
                    __syncthreads();
                    if constexpr (DBG == 8) t_lastq = __builtin_amdgcn_s_memtime();
#if KEDS_ATTN_PREFETCH
                    {
                        // L2 prefetch for the workgroup that follows this one on the XCD (block id + KEDS_ATTN_PREFETCH: a multiple
                        // of 8 keeps the XCD): its 771 lines (257 rows x {q, k, v} x 128 B) as 4-byte LDS-DMA requests, one line per
                        // lane, into a dump area -- no register, nothing waits for them but this workgroup's last vmcnt(0).  A
                        // workgroup spends ~4.5 of its ~16 us waiting for the first byte of its rows.
                        const int nb = (int)blockIdx.x + KEDS_ATTN_PREFETCH;
                        if (nb < (int)gridDim.x) {
                            const int b2 = nb / heads, h2 = nb - b2 * heads;
                            const bf16_t* base2 = qkv + (size_t)b2 * S * ld + h2 * DH;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int line = 64 * (2 * wave + i) + lane;
                                if (line < 3 * S) {
                                    const int row = line / 3, part = line - 3 * row;
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base2 + (size_t)row * ld + part * d),
                                                                     (__attribute__((address_space(3))) void*)(smem + DUMP_OFF), 4, 0, 0);
                                }
                            }
                        }
                    }
#endif
                    if (do_last) last_query_partial();
                }
                const char* kb = smem + half * 16384;
                bf16x8 a[4];                                               // K fragments of the NEXT tile: read under the P.V MFMAs
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(kb + ka[ks]);
                // four tiles unrolled: every LDS address is a lane constant + an immediate
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    f32x16 sc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
                    if constexpr (DBG != 2) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], qf[ks], sc, 0, 0, 0);
                    } else {
                        sc[0] = (float)qf[k4][0] + (float)a[k4][0];
                    }
                    if (k4 == 0 && half == 0) {                            // the reference: first tile's row maximum
                        float t = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
                        for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, sc[i]), sc[i + 1]);
                        t = fmaxf(t, sc[15]);
                        mref = have ? mgiven : halves_max(t);
                        nm = -mref * sl2;
                    }
                    if constexpr (DBG != 3) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], sl2, nm));
                            lsum += e;
                            sc[i] = e;
                        }
                    }
                    const bf16x8 p0 = bf16x8{(bf16_t)sc[0], (bf16_t)sc[1], (bf16_t)sc[2], (bf16_t)sc[3],
                                             (bf16_t)sc[4], (bf16_t)sc[5], (bf16_t)sc[6], (bf16_t)sc[7]};
                    const bf16x8 p1 = bf16x8{(bf16_t)sc[8], (bf16_t)sc[9], (bf16_t)sc[10], (bf16_t)sc[11],
                                             (bf16_t)sc[12], (bf16_t)sc[13], (bf16_t)sc[14], (bf16_t)sc[15]};
                    if (k4 < 3) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(kb + ka[ks] + (k4 + 1) * 4096);
                    }
                    if constexpr (DBG != 4) {
                        const char* v0 = kb + vb0 + k4 * 4096;
                        const char* v1 = kb + vb1 + k4 * 4096;
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(v0, v0 + 1024), p0, o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(v1, v1 + 1024), p0, o1, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(v0 + 2048, v0 + 3072), p1, o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(v1 + 2048, v1 + 3072), p1, o1, 0, 0, 0);
                    } else {
                        o0[0] += (float)p0[0] + (float)p1[0];
                    }
                    __builtin_amdgcn_sched_barrier(0);                     // keep the tiles apart: hoisted reads spill at 128 VGPRs
                }
            }
            if constexpr (DBG == 8) t_loop = __builtin_amdgcn_s_memtime();
            // ---- the last key: score from this lane's 32 query dims, combined over the two halves
            float ts = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ts = dot8(qf[ks], *reinterpret_cast<const bf16x8*>(tk + (2 * ks + hh) * 16), ts);
            ts = halves_sum(ts);
            // Overflow check of the reference: every e <= the row sum, so a bounded sum bounds every probability and every
            // P.V term; (ts - mref) covers the rank-1 key.  A NaN row fails the test too (recomputed once, stays NaN).
            const float ltot = halves_sum(lsum);
            const bool bad = !(ltot <= 0x1p80f) || !((ts - mref) * sl2 <= 64.0f);
            if (!have && __builtin_amdgcn_ballot_w64(bad) != 0ull) {
                mgiven = ts;
                have = true;
                continue;
            }
            const float et = __builtin_amdgcn_exp2f(__builtin_fmaf(ts, sl2, nm));
            const float inv = 1.0f / (ltot + et);
            const float tp = (float)(bf16_t)et;                         // rounded like the probabilities the MFMA path multiplies
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const bf16x4 va = *reinterpret_cast<const bf16x4*>(tv + (8 * g4 + 4 * hh) * 2);
                const bf16x4 vb = *reinterpret_cast<const bf16x4*>(tv + (32 + 8 * g4 + 4 * hh) * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[4 * g4 + j] += tp * (float)va[j];
                    o1[4 * g4 + j] += tp * (float)vb[j];
                }
            }
            // 16-byte stores: lane (query r, half hh) holds dims 8 g + 4 hh + {0..3} of every 8-dim group g; the two halves swap
            // the odd / even groups (v_permlane32_swap: the upper lanes of the first operand against the lower lanes of the
            // second), after which hh = 0 owns the even groups whole and hh = 1 the odd ones
            const int query = 32 * wave + r;
            if (Q8 && b * S + query < q8_rows) {      // (every row but the remainder rows of the tower's last half tile)
                if (query < q_limit) {
                    const int row = b * S + query;
                    int ex[2];
#pragma unroll
                    for (int tile = 0; tile < 2; ++tile) {
                        const f32x16& o = tile ? o1 : o0;
                        float amax = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(o[i] * inv));
                        amax = halves_max(amax);                           // the block's other 16 dims sit in lane r + 32
                        const int e = mx_block_exp(amax);
                        ex[tile] = e;
                        const float inv2 = e == -127 ? 0.f : __uint_as_float((unsigned)(127 - e) << 23);
                        unsigned dw[4];
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            float sv[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) sv[j] = fminf(fmaxf(o[4 * g4 + j] * inv * inv2, -448.f), 448.f);
                            unsigned w2 = __builtin_amdgcn_cvt_pk_fp8_f32(sv[0], sv[1], 0u, false);
                            dw[g4] = __builtin_amdgcn_cvt_pk_fp8_f32(sv[2], sv[3], w2, true);
                        }
                        // lane (r, hh) holds dims 8 g4 + 4 hh + {0..3}; after the swaps hh = 0 owns dims 0-15, hh = 1 dims 16-31
                        const auto s02 = __builtin_amdgcn_permlane32_swap(dw[0], dw[2], false, false);
                        const auto s13 = __builtin_amdgcn_permlane32_swap(dw[1], dw[3], false, false);
                        *reinterpret_cast<u32x4*>(q8 + (size_t)row * d + h * DH + 32 * tile + 16 * hh) =
                            u32x4{s02[0], s02[1], s13[0], s13[1]};
                    }
                    if (hh == 0)
                        *reinterpret_cast<unsigned short*>(s8 + mx_scale_index((h * DH) >> 5, row, q8_rows)) =
                            (unsigned short)((ex[0] + 127) | ((ex[1] + 127) << 8));
                }
                break;
            }
            u32x4 st[4];
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
                for (int pr2 = 0; pr2 < 2; ++pr2) {
                    const int ge = 2 * pr2, go = 2 * pr2 + 1;
                    const f32x16& o = tile ? o1 : o0;
                    const bf16x4 pe = bf16x4{(bf16_t)(o[4 * ge] * inv), (bf16_t)(o[4 * ge + 1] * inv), (bf16_t)(o[4 * ge + 2] * inv),
                                             (bf16_t)(o[4 * ge + 3] * inv)};
                    const bf16x4 po = bf16x4{(bf16_t)(o[4 * go] * inv), (bf16_t)(o[4 * go + 1] * inv), (bf16_t)(o[4 * go + 2] * inv),
                                             (bf16_t)(o[4 * go + 3] * inv)};
                    const u32x2 ue = __builtin_bit_cast(u32x2, pe), uo = __builtin_bit_cast(u32x2, po);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
                    st[2 * tile + pr2] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
            if (query < q_limit) {
                // (row `query`, element 8 hh: hh = 0 holds groups 0, 2; hh = 1 groups 1, 3)
#pragma unroll
                for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                    for (int pr2 = 0; pr2 < 2; ++pr2)
                        keds_store16<KEDS_ST_ATTN>(st[2 * tile + pr2], obase, (unsigned)(((size_t)query * d + 8 * hh + 32 * tile + 16 * pr2) * 2));
            }
            break;
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // the barrier in front of keys 128-255
    }
    if constexpr (DBG == 8) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0 && stamp) {
            unsigned long long* o = stamp + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = t_staged - t_in;     // row 256 + Q loads issued, K / V DMA landed, barrier
            o[1] = t_lastq - t_staged;  // key tiles 0-3, wait for keys 128-255, barrier
            o[2] = t_loop - t_lastq;    // last query partial + key tiles 4-7
            o[3] = t_issued - t_loop;   // last key, normalisation, stores issued
            o[4] = t_in;
            o[5] = t_end;
            o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |                 // HW_ID: where the wave ran
                   ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);          // XCC_ID
            o[7] = t_end - t_issued;    // store drain
        }
    }
}

// NOTE (measured, round 2): a PERSISTENT form of the kernel above was built (two workgroups per CU walking the (batch, head)
// items; the two 128-key halves of the LDS image as a ring: keys 0-127 of item i+1 requested behind the mid-item barrier of
// item i, keys 128-255 behind the top barrier; row 256 by LDS-DMA; overflow handled per tile with a lazy online-softmax
// rescale so that no item is revisited).  It aimed at the 8-9 k of a wave's 27 k cycles spent waiting for its item's rows
// (PMC: waves parked 39 %, issue ports 60 % busy).  Correct, and 195 us against 70 us: at the 128-VGPR budget of four waves
// per SIMD the item loop spills 36-110 registers (hoisted lane offsets, the next item's query fragments beside the output
// tiles), and every spill reload is a vmcnt wait that, vmcnt being in-order, also waits for the LDS-DMA in flight -- the
// prefetch it was built for.  Not kept; an assembly item loop is the way to do this.
// NOTE (measured, round 3; profiles/r03_attention_experiments.txt): three more forms, all bit-identical or equally accurate,
// none faster than the kernel above.
//  (a) ONE 8-wave workgroup per CU at the 256-register budget with TWO full item images in LDS (2 x 69 KB), item i+1
//      requested behind the top barrier of item i, the per-item body unchanged: no spill (196 VGPRs), 84 us against 63 us
//      (B = 128; 44 vs 34 at B = 64).  Two co-resident one-item workgroups already hide each other's row waits AND give every
//      SIMD four waves to interleave exp / MFMA / LDS reads; the persistent form keeps the first and halves the second.
//      160 KB of LDS holds two images either way: "prefetch" and "four waves per SIMD" exclude each other at bf16 K / V.
//  (b) one vmcnt(0) wait at the top and no mid-item barrier; the same with the last-query share of waves 4-7 moved in front
//      of their tiles (a half-tile stagger of the SIMD partners); the same with s_setprio 1 for waves 4-7: 60.6-64.4,
//      62.3-68.1, 60.7-66.6 us against 59.8-64.6 on the same boxes -- inside the box-to-box spread.
//  (c) the last query row on the matrix pipe: each wave runs one more key-tile step on its OWN key tile with row 256's query
//      broadcast into all 32 B-operand columns (4 + 4 MFMAs, 16 exp2, no cross-lane reduction, lanes 0 / 32 store the
//      partial; key 256 joins in the combine).  118 VGPRs instead of 128, same error (last rows 1.7e-3 rel-L2), 59.8-64.7 us
//      against 60.3-64.0 for the VALU form: the row costs one key-tile step per wave (a ninth of the tile work, 4.5 us:
//      "partials only" ablation 62 us, no last row 57.5) whichever unit runs it.
//  (d) S^T of key tile t+1 issued in front of P.V of tile t (so that the exp block never waits for its own MFMA chain): the
//      second score tile is live across the P.V step, 16 registers more than the 128 of four waves per SIMD hold -- 58-82
//      spilled registers, 22 scratch accesses per four tiles in the loop, with the K fragments read early or late.  Not run.
//  (e) the round-2 ring again, on this kernel's body and with everything this round learned (keys 128-255 requested behind the
//      top barrier, keys 0-127 and row 256 of the NEXT item behind the mid-item barrier, its query fragments loaded into the
//      registers of this item's right behind their last use, a counted vmcnt(4) at the top so that the four output stores
//      stay in flight, the last-query shares of waves 0-3 before the first half is overwritten; no overflow recompute yet):
//      bit-identical, the tile loop itself free of scratch, but 48 spilled registers around it (item top, last-query share,
//      epilogue) -- 122 us against 62.  A scratch reload is a vmcnt wait behind the DMA in flight; the compiler cannot be
//      told.  The item loop needs hand-allocated registers.
// In-kernel stamps with XCC_ID: every CU runs exactly 8 workgroups, 1.84 of 2 resident on average, the next workgroup enters
// 700-900 cycles after an exit, per-CU span 114.6 k cycles mean / 127 k max: a tenth of the launch is the spread between CUs.
// NOTE (measured, round 2): the "keys 0-127 first" wait of the kernel above is not what the hardware executes: __syncthreads()
// is a workgroup-scope fence and drains vmcnt in front of the barrier, the compiler puts s_waitcnt vmcnt(0) in front of the
// first ds_write behind an LDS-DMA and in front of the first ds_read_tr builtin (it cannot tell them from the DMA in flight),
// and a plain query load is waited for with vmcnt(0) at its first use.  A build with raw s_barrier asm, row 256 by LDS-DMA,
// inline-asm query loads and inline-asm transposed reads (so that tiles 0-3 really run under the DMA of keys 128-255) is
// correct and SLOWER, 92 us against 70: the asm statements pin the schedule and spill 15 registers, and the wait for keys
// 0-127 alone is as long as the wait for all 256 (10 k cycles: first-byte latency, not volume).  Not kept.

int g_attn_debug = 0;   // timing-only ablations (ViT kernel)
int g_attn_tail = 1;    // A/B hook: 0 routes S = 257 through the generic (padded) kernel

int launch_attn_tail1(const void* qkv, void* out, int B, int heads, int q_limit, hipStream_t st) {
    using C = AttnCfg<16>;
    constexpr int LDS = C::LDS + TAIL_LDS;
    if (int rc = keds_func_lds_once((const void*)attention_tail1_kernel<16>, LDS, "attention_tail1_kernel")) return rc;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    attention_tail1_kernel<16><<<B * heads, 256, LDS, st>>>((const bf16_t*)qkv, (bf16_t*)out, heads, q_limit);
    return keds_check_launch("attention_tail1_kernel");
}

int g_attn_s257 = 1;    // A/B hook: 0 routes S = 257 through the 4-wave tail kernel

unsigned long long* g_attn_stamp = nullptr;   // stamped build (code 8): 8 counters per wave, keds_attention_stamp_buffer
int g_attn_s257_dbg = 0;   // timing-only ablations of the 8-wave kernel (keds_attention_debug bit 6 + code)

template <int V>
int launch_attn_s257_dbg(const void* qkv, void* out, int B, int heads, int q_limit, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)attention_s257_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, s257::LDS);
    attention_s257_kernel<V><<<B * heads, 512, s257::LDS, st>>>((const bf16_t*)qkv, (bf16_t*)out, heads, q_limit, g_attn_stamp);
    return keds_check_launch("attention_s257_kernel<dbg>");
}

// The towers fork their remainder-row chain behind this launch.  An event RECORDED on the launching stream is a marker packet of
// its own between two kernels (a kernel trace shows 7.9 us between the attention's end and the next GEMM's start with it, 1.4
// without: profiles/r05_trace_block.txt); as the launch's STOP event it is the kernel's completion signal and costs nothing.
thread_local hipEvent_t tl_attn_stop = nullptr;
thread_local bool tl_attn_stop_taken = false;
template <typename K, typename... A>
void launch_s257(K kernel, int grid, hipStream_t st, A... args) {
    hipEvent_t stop = tl_attn_stop;
    tl_attn_stop = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stop && (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) stop = nullptr;   // (a graph capture records it itself)
    if (stop) {
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(512), s257::LDS, st, nullptr, stop, 0, args...);
        tl_attn_stop_taken = true;
    } else {
        kernel<<<grid, 512, s257::LDS, st>>>(args...);
    }
}

int launch_attn_s257_q8(const void* qkv, void* out, int B, int heads, int q_limit, void* q8, void* s8, int q8_rows,
                        hipStream_t st) {
    if (int rc = keds_func_lds_once((const void*)attention_s257_kernel<0, true>, s257::LDS, "attention_s257_kernel<q8>")) return rc;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    launch_s257(attention_s257_kernel<0, true>, B * heads, st, (const bf16_t*)qkv, (bf16_t*)out, heads, q_limit,
                (unsigned long long*)nullptr, (unsigned char*)q8, (unsigned char*)s8, q8_rows);
    return keds_check_launch("attention_s257_kernel<q8>");
}

int launch_attn_s257(const void* qkv, void* out, int B, int heads, int q_limit, hipStream_t st) {
    switch (g_attn_s257_dbg) {
        case 1: return launch_attn_s257_dbg<1>(qkv, out, B, heads, q_limit, st);
        case 2: return launch_attn_s257_dbg<2>(qkv, out, B, heads, q_limit, st);
        case 3: return launch_attn_s257_dbg<3>(qkv, out, B, heads, q_limit, st);
        case 4: return launch_attn_s257_dbg<4>(qkv, out, B, heads, q_limit, st);
        case 5: return launch_attn_s257_dbg<5>(qkv, out, B, heads, q_limit, st);
        case 6: return launch_attn_s257_dbg<6>(qkv, out, B, heads, q_limit, st);
        case 7: return launch_attn_s257_dbg<7>(qkv, out, B, heads, q_limit, st);
        case 8: return launch_attn_s257_dbg<8>(qkv, out, B, heads, q_limit, st);
        case 9: return launch_attn_s257_dbg<9>(qkv, out, B, heads, q_limit, st);
        default: break;
    }
    if (int rc = keds_func_lds_once((const void*)attention_s257_kernel<0>, s257::LDS, "attention_s257_kernel")) return rc;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    launch_s257(attention_s257_kernel<0>, B * heads, st, (const bf16_t*)qkv, (bf16_t*)out, heads, q_limit, (unsigned long long*)nullptr,
                (unsigned char*)nullptr, (unsigned char*)nullptr, 0);
    return keds_check_launch("attention_s257_kernel");
}

template <int NKT, bool CAUSAL, int NFULL>
int launch_attn(const void* qkv, void* out, int B, int S, int heads, int q_limit, void* q8, void* s8, int q8_rows,
                hipStream_t st, const int* seq_off = nullptr) {
    using C = AttnCfg<NKT>;
    if (int rc = keds_func_lds_once((const void*)attention_kernel<NKT, CAUSAL, NFULL>, C::LDS, "attention_kernel")) return rc;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    if constexpr (NKT == 18 && !CAUSAL && NFULL == 16) {
        if (g_attn_debug) {
#define KEDS_ATTN_DBG(V)                                                                                          \
    {                                                                                                            \
        (void)hipFuncSetAttribute((const void*)attention_kernel<NKT, CAUSAL, NFULL, V>,                          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);                           \
        attention_kernel<NKT, CAUSAL, NFULL, V><<<B * heads, 256, C::LDS, st>>>((const bf16_t*)qkv, (bf16_t*)out, S, heads, q_limit, (unsigned char*)q8, (unsigned char*)s8, q8_rows, seq_off); \
    }
            switch (g_attn_debug) {
                case 1: KEDS_ATTN_DBG(1) break;
                case 2: KEDS_ATTN_DBG(2) break;
                case 3: KEDS_ATTN_DBG(3) break;
                case 4: KEDS_ATTN_DBG(4) break;
                default: KEDS_ATTN_DBG(5) break;
            }
#undef KEDS_ATTN_DBG
            return keds_check_launch("attention_kernel<dbg>");
        }
    }
    attention_kernel<NKT, CAUSAL, NFULL><<<B * heads, 256, C::LDS, st>>>((const bf16_t*)qkv, (bf16_t*)out, S, heads, q_limit,
                                                                         (unsigned char*)q8, (unsigned char*)s8, q8_rows, seq_off);
    return keds_check_launch("attention_kernel");
}

}  // namespace

void keds_attention_stop_event(hipEvent_t ev) {
    tl_attn_stop = ev;
    tl_attn_stop_taken = false;
}
bool keds_attention_stop_event_taken() {
    tl_attn_stop = nullptr;                 // (a kernel form that does not take it: the caller records the event itself)
    return tl_attn_stop_taken;
}

extern "C" int keds_attention_stamp_buffer(void* buf) {      // diagnostic: B * heads * 64 uint64 for keds_attention_debug(64 + 8)
    g_attn_stamp = (unsigned long long*)buf;
    return KEDS_OK;
}

extern "C" int keds_attention_debug(int variant) {
    g_attn_s257_dbg = (variant >> 6) & 1 ? (variant & 15) : 0;   // bit 6: the code applies to the 8-wave S = 257 kernel
    g_attn_debug = (variant >> 6) & 1 ? 0 : (variant & 15);
    g_attn_tail = (variant >> 4) & 1 ? 0 : 1;      // bit 4: S = 257 through the generic kernel (A/B)
    g_attn_s257 = (variant >> 5) & 1 ? 0 : 1;      // bit 5: S = 257 through the 4-wave tail kernel (A/B)
    return KEDS_OK;
}

extern "C" int keds_attention_ex(const void* qkv, void* out, int B, int S, int heads, int causal, int q_limit,
                                 void* stream) {
    return keds_attention_mx(qkv, out, B, S, heads, causal, q_limit, nullptr, nullptr, 0, stream);
}

extern "C" int keds_attention_mx(const void* qkv, void* out, int B, int S, int heads, int causal, int q_limit, void* q8,
                                 void* s8, int q8_rows, void* stream) {
    KEDS_REQUIRE(qkv && out && B > 0 && heads > 0, "keds_attention: bad argument");
    KEDS_REQUIRE((q8 == nullptr) == (s8 == nullptr) && (q8 == nullptr || (q8_rows > 0 && (heads * 64) % 128 == 0)),
                 "keds_attention_mx: q8, s8 and q8_rows come together; width must be a multiple of 128");
    KEDS_REQUIRE(S >= 1 && S <= 288, "keds_attention: S=%d unsupported (1..288)", S);
    if (q_limit <= 0 || q_limit > S) q_limit = S;
    hipStream_t st = (hipStream_t)stream;
    if (causal) {
        if (S <= 32) return launch_attn<2, true, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
        if (S <= 96) return launch_attn<6, true, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
        return launch_attn<18, true, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
    }
    if (S <= 32) return launch_attn<2, false, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
    if (S <= 96) return launch_attn<6, false, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
    if (S == 257 && !q8 && g_attn_tail && g_attn_s257 && !g_attn_debug) return launch_attn_s257(qkv, out, B, heads, q_limit, st);
    if (S == 257 && q8 && g_attn_tail && g_attn_s257 && !g_attn_debug && !g_attn_s257_dbg)
        return launch_attn_s257_q8(qkv, out, B, heads, q_limit, q8, s8, q8_rows, st);
    if (S == 257 && !q8 && g_attn_tail && !g_attn_debug) return launch_attn_tail1(qkv, out, B, heads, q_limit, st);
    if (S >= 256) return launch_attn<18, false, 16>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);   // ViT-L/14: 257 tokens
    return launch_attn<18, false, 0>(qkv, out, B, S, heads, q_limit, q8, s8, q8_rows, st);
}

// Packed rows: sample b is rows [seq_off[b], seq_off[b + 1]) of qkv / out (device int32 [B + 1], lengths 1 .. s_max).  The text
// tower's captions end at different columns and the mask is causal: rows behind a caption's read-out column reach nothing that is
// read (model.py:543-549), so the tower computes none of them (keds_text_run_packed).
extern "C" int keds_attention_packed(const void* qkv, void* out, int B, int s_max, const int32_t* seq_off, int heads, int causal,
                                     void* stream) {
    KEDS_REQUIRE(qkv && out && seq_off && B > 0 && heads > 0, "keds_attention_packed: bad argument");
    KEDS_REQUIRE(s_max >= 1 && s_max <= 288, "keds_attention_packed: s_max=%d unsupported (1..288)", s_max);
    hipStream_t st = (hipStream_t)stream;
    if (causal) {
        if (s_max <= 32) return launch_attn<2, true, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
        if (s_max <= 96) return launch_attn<6, true, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
        return launch_attn<18, true, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
    }
    if (s_max <= 32) return launch_attn<2, false, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
    if (s_max <= 96) return launch_attn<6, false, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
    return launch_attn<18, false, 0>(qkv, out, B, s_max, heads, s_max, nullptr, nullptr, 0, st, seq_off);
}

extern "C" int keds_attention(const void* qkv, void* out, int B, int S, int heads, int causal, void* stream) {
    return keds_attention_ex(qkv, out, B, S, heads, causal, S, stream);
}
