// Attention of the fp32x3 operating point (CLIP.set_precision("fp32x3"), keds_tower_params.f32 = 2): fp32-grade
// softmax(q k^T / 8 [+ causal mask]) v per (sample, head) -- model.py:319-321 (nn.MultiheadAttention inside
// ResidualAttentionBlock), the causal mask model.py:543-549 -- with BOTH products on the fp16 matrix instruction.
//
// The f32-input form (f32path.hip, attention_f32_kernel) spends 64 cycles per v_mfma_f32_32x32x2_f32: 680 us per ViT-L/14 layer at
// B = 128, a fifth of the fp32x3 step.  Here every fp32 operand is two fp16 planes, x = hi + lo (22 significant bits; the low
// plane of small values lives in fp16 subnormals, which the matrix unit does not flush), and a product is hi.hi + hi.lo + lo.hi
// accumulated in fp32 -- the arithmetic of csrc/gemm.hip's KEDS_EPI_X3_* kernels: 3 x 32 cycles per 32 x 32 x 16 block where the
// f32 form needs 8 x 64.  Softmax statistics, exponentials (full precision) and the running rescale stay fp32.
//
// One workgroup (9 waves) per (sample, head).  Staging splits K and V while they go to LDS:
//   K   [SP][72] halves per plane (144-byte rows: the 16 lanes of a ds_read_b128 group land on 16 distinct 16-byte slots)
//   V^T [64][SP + 8] halves per plane, the keys of a 32-key tile stored in the order the probabilities leave the score
//       accumulator (bits 2 and 3 of the key offset swapped): one ds_read_b128 is one MFMA operand
// S = 257: 82,944 + 75,776 = 158,720 bytes of the 160 KiB.  A wave owns 32-query tiles qt = wave, wave + 9, ... and walks the
// key tiles with an online softmax, as attention_f32_kernel does:
//   S^T tile  = K tile . Q^T    12 MFMAs 32x32x16 (4 K-steps x 3 products); the lane's q values (/ 8: exact) sit in registers
//               as planes; result: lane = query, register r = key (r & 3) + 8 (r >> 2) + 4 h
//   p = exp(s - m_new) in fp32 (sum l from the fp32 values), then split into planes in place
//   O^T tile += V^T tile . P^T  12 MFMAs: MFMA c of a tile sums over the keys of registers 8c .. 8c + 7
// The output goes out as fp32 and / or as the two fp16 planes the out-projection's split-operand GEMM reads (saves the
// separate split pass over the attention output).  S <= 288.
#include "keds_common.h"
#include <math.h>

#ifndef KEDS_AX_DBG
#define KEDS_AX_DBG 0      // timing only: 1 = staging alone, 2 = no staging, 4 = no exponentials, 8 = no MFMAs
#endif

namespace {

constexpr int AX_WAVES = 9;
constexpr int AX_KP = 72;                    // halves per K row in LDS

__device__ __forceinline__ int ax_vpos(int o) {          // key offset in its 32-key tile -> position in the V^T row
    return (o & ~12) | ((o & 4) << 1) | ((o & 8) >> 1);
}

// exp(x) for x <= 0 to fp32 rounding level: 2^(x log2 e) with the product's rounding error carried into the result
// (v_exp_f32 is accurate to 1 ulp; a bare exp2f(x * log2e) loses |x| * 2^-24 relative)
__device__ __forceinline__ float ax_exp(float x) {
    if (KEDS_AX_DBG & 4) return x * 0.5f;
    x = fmaxf(x, -120.0f);                               // masked scores (-inf) and anything below fp32's range: 2^-173 -> 0
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float t = x * L2E;
    const float e = fmaf(x, L2E, -t) + x * L2E_LO;       // x log2 e = t + e
    const float y = __builtin_amdgcn_exp2f(t);
    return fmaf(y, e * 0.693147182464599609375f, y);
}

struct AxSplit {
    f16x8 hi, lo;
};
__device__ __forceinline__ f32x16 ax_mfma(f16x8 a, f16x8 b, f32x16 c) {
    if (KEDS_AX_DBG & 8) {
        c[0] += (float)a[0] * (float)b[0];
        return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void ax_split(float v, f16_t& hi, f16_t& lo) {
    hi = (f16_t)v;
    lo = (f16_t)(v - (float)hi);
}

__global__ __launch_bounds__(64 * AX_WAVES) void attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                       f16_t* __restrict__ pair, long long plane, int S, int heads,
                                                                       int causal, int q_limit, int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) f16_t ax_lds[];
    const int nkt = (S + 31) >> 5, SP = nkt * 32, VP = SP + 8;
    f16_t* Kh = ax_lds;                                   // [SP][72]
    f16_t* Kl = Kh + (size_t)SP * AX_KP;
    f16_t* Vh = Kl + (size_t)SP * AX_KP;                  // [64][VP]
    f16_t* Vl = Vh + (size_t)64 * VP;
    const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
    const int d = heads * 64, ld = 3 * d;
    const float* base = qkv + (size_t)b * S * ld + hd * 64;
    bool bad = false;
    // ---- staging: one item = two adjacent keys x four dims (adjacent keys are adjacent in the V^T row)
    for (int idx = threadIdx.x; idx < ((KEDS_AX_DBG & 2) ? 0 : (SP >> 1) * 16); idx += 64 * AX_WAVES) {
        const int r2 = idx >> 4, c4 = idx & 15, row = 2 * r2;
        f32x4 k0 = f32x4{0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
        if (row < S) {
            k0 = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + d + 4 * c4);
            v0 = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 2 * d + 4 * c4);
        }
        if (row + 1 < S) {
            k1 = *reinterpret_cast<const f32x4*>(base + (size_t)(row + 1) * ld + d + 4 * c4);
            v1 = *reinterpret_cast<const f32x4*>(base + (size_t)(row + 1) * ld + 2 * d + 4 * c4);
        }
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        f16x4 h0, l0, h1, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bad |= !(fmaxf(fmaxf(fabsf(k0[e]), fabsf(k1[e])), fmaxf(fabsf(v0[e]), fabsf(v1[e]))) < 65504.0f);   // (NaN: fmaxf drops it; the guard's isfinite check of the output sees it)
            f16_t a, c;
            ax_split(k0[e], a, c);
            h0[e] = a, l0[e] = c;
            ax_split(k1[e], a, c);
            h1[e] = a, l1[e] = c;
        }
        *reinterpret_cast<f16x4*>(Kh + (size_t)row * AX_KP + 4 * c4) = h0;
        *reinterpret_cast<f16x4*>(Kl + (size_t)row * AX_KP + 4 * c4) = l0;
        *reinterpret_cast<f16x4*>(Kh + (size_t)(row + 1) * AX_KP + 4 * c4) = h1;
        *reinterpret_cast<f16x4*>(Kl + (size_t)(row + 1) * AX_KP + 4 * c4) = l1;
        const int vp = (row & ~31) + ax_vpos(row & 31);   // even: the pair (row, row + 1) is one dword
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16_t a0, c0, a1, c1;
            ax_split(v0[e], a0, c0);
            ax_split(v1[e], a1, c1);
            *reinterpret_cast<f16x2*>(Vh + (size_t)(4 * c4 + e) * VP + vp) = f16x2{a0, a1};
            *reinterpret_cast<f16x2*>(Vl + (size_t)(4 * c4 + e) * VP + vp) = f16x2{c0, c1};
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 31, hh = lane >> 5;
    const int nq = q_limit < S ? q_limit : S;
    const int nqt = (nq + 31) >> 5;
    // the first query tile's q rows travel while the staging writes land
    __syncthreads();
    if (KEDS_AX_DBG & 1) return;
    for (int qt = wave; qt < nqt; qt += AX_WAVES) {
        const int q = qt * 32 + j;
        const float* qrow = base + (size_t)(q < S ? q : S - 1) * ld + 8 * hh;
        f16x8 qh[4], ql[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                            // dims 16 t + 8 h + e
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * t), c = *reinterpret_cast<const f32x4*>(qrow + 16 * t + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e < 4 ? a[e] : c[e - 4]) * 0.125f;              // 1 / sqrt(64): exact scaling
                bad |= !(fabsf(v) < 65504.0f);
                f16_t x, y;
                ax_split(v, x, y);
                qh[t][e] = x, ql[t][e] = y;
            }
        }
        float m = -INFINITY, l = 0.f;
        f32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) o0[e] = 0.f, o1[e] = 0.f;
        int kt_end = nkt;
        if (causal) {                                                            // key tiles that hold a key <= the tile's last query
            const int lastq = qt * 32 + 31;
            kt_end = (lastq >> 5) + 1 < nkt ? (lastq >> 5) + 1 : nkt;
        }
        for (int kt = 0; kt < kt_end; ++kt) {
            f32x16 sc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[e] = 0.f;
            const f16_t* kr = Kh + (size_t)(kt * 32 + j) * AX_KP + 8 * hh;
            const size_t klo = (size_t)SP * AX_KP;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(kr + 16 * t), al = *reinterpret_cast<const f16x8*>(kr + klo + 16 * t);
                sc = ax_mfma(al, qh[t], sc);
                sc = ax_mfma(ah, ql[t], sc);
                sc = ax_mfma(ah, qh[t], sc);
            }
            float tmax = -INFINITY;
            const bool edge = (kt * 32 + 32 > S) || (causal && kt * 32 + 31 > qt * 32);   // (wave-uniform) a tile with masked keys
            if (edge) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const bool valid = key < S && (!causal || key <= q);
                    sc[r] = valid ? sc[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(m, tmax);
            const float resc = m == -INFINITY ? 0.f : ax_exp(m - mnew);          // (first tile, or nothing valid so far)
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = mnew == -INFINITY ? 0.f : ax_exp(sc[r] - mnew);          // masked keys: exp(-inf) = 0
                psum += sc[r];
            }
            l = l * resc + psum;
#pragma unroll
            for (int e = 0; e < 16; ++e) o0[e] *= resc, o1[e] *= resc;
            const f16_t* vr = Vh + (size_t)j * VP + kt * 32 + 8 * hh;
            const size_t vlo = (size_t)64 * VP, vblk = (size_t)32 * VP;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f16_t x, y;
                    ax_split(sc[8 * c + e], x, y);
                    ph[e] = x, pl[e] = y;
                }
                const f16x8 v0h = *reinterpret_cast<const f16x8*>(vr + 16 * c), v0l = *reinterpret_cast<const f16x8*>(vr + vlo + 16 * c);
                const f16x8 v1h = *reinterpret_cast<const f16x8*>(vr + vblk + 16 * c), v1l = *reinterpret_cast<const f16x8*>(vr + vblk + vlo + 16 * c);
                o0 = ax_mfma(v0l, ph, o0);
                o1 = ax_mfma(v1l, ph, o1);
                o0 = ax_mfma(v0h, pl, o0);
                o1 = ax_mfma(v1h, pl, o1);
                o0 = ax_mfma(v0h, ph, o0);
                o1 = ax_mfma(v1h, ph, o1);
            }
            m = mnew;
        }
        const float ltot = l + __shfl_xor(l, 32, 64);
        if (q < nq) {                                                            // lane = query q; register r = dim (r & 3) + 8 (r >> 2) + 4 h
            const size_t off = ((size_t)b * S + q) * d + hd * 64 + 4 * hh;
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& o = blk ? o1 : o0;
                    const f32x4 v = f32x4{o[4 * g] / ltot, o[4 * g + 1] / ltot, o[4 * g + 2] / ltot, o[4 * g + 3] / ltot};
                    if (out) *reinterpret_cast<f32x4*>(out + off + 32 * blk + 8 * g) = v;
                    if (pair) {
                        f16x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f16_t x, y;
                            ax_split(v[e], x, y);
                            vh[e] = x, vl[e] = y;
                        }
                        *reinterpret_cast<f16x4*>(pair + off + 32 * blk + 8 * g) = vh;
                        *reinterpret_cast<f16x4*>(pair + plane + off + 32 * blk + 8 * g) = vl;
                    }
                }
        }
    }
    if (bad && overflow) *overflow = 1;                   // |q / 8|, |k| or |v| >= 65504: no fp16 hi plane (the caller falls back)
}

}  // namespace

// qkv fp32 [B, S, 3 * heads * 64]; out fp32 [B, S, heads * 64] (nullable); pair: fp16 planes [2][plane] of the same rows
// (nullable; plane >= B * S * heads * 64 elements); at least one of them.  q_limit > 0: the first q_limit queries of every sample
// only (the CLS query of the last ViT block).  overflow (nullable): raised when an operand does not fit an fp16 hi plane.
extern "C" int keds_attention_x3(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal,
                                 int q_limit, int* overflow, void* stream) {
    KEDS_REQUIRE(qkv && (out || pair) && B > 0 && heads > 0, "keds_attention_x3: bad argument");
    KEDS_REQUIRE(S >= 1 && S <= 288, "keds_attention_x3: S must be in [1, 288] (got %d)", S);
    KEDS_REQUIRE(!pair || plane >= (int64_t)B * S * heads * 64, "keds_attention_x3: plane stride shorter than the output");
    const int SP = (S + 31) / 32 * 32;
    const int lds = 2 * (SP * AX_KP + 64 * (SP + 8)) * (int)sizeof(f16_t);
    int rc = keds_func_lds_once((const void*)attention_x3_kernel, lds, "attention_x3_kernel");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    attention_x3_kernel<<<B * heads, 64 * AX_WAVES, lds, st>>>(qkv, out, (f16_t*)pair, plane, S, heads, causal,
                                                                q_limit > 0 ? q_limit : S, overflow);
    return keds_check_launch("attention_x3_kernel");
}
