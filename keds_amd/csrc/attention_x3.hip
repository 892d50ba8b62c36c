// Attention of the fp32x3 operating point (CLIP.set_precision("fp32x3"), keds_tower_params.f32 = 2): fp32-grade
// softmax(q k^T / 8 [+ causal mask]) v per (sample, head) -- model.py:319-321 (nn.MultiheadAttention inside
// ResidualAttentionBlock), the causal mask model.py:543-549 -- with BOTH products on the fp16 matrix instruction.
//
// The f32-input form (f32path.hip, attention_f32_kernel) spends 64 cycles per v_mfma_f32_32x32x2_f32: 680 us per ViT-L/14 layer at
// B = 128, a fifth of the fp32x3 step.  Here every fp32 operand is two fp16 planes, x = hi + lo (22 significant bits; the low
// plane of small values lives in fp16 subnormals, which the matrix unit does not flush), and a product is hi.hi + hi.lo + lo.hi
// accumulated in fp32 -- the arithmetic of csrc/gemm.hip's KEDS_EPI_X3_* kernels: 3 x 32 cycles per 32 x 32 x 16 block where the
// f32 form needs 8 x 64.  Softmax statistics, exponentials and the running rescale stay fp32.
//
// One workgroup (8 waves: two per SIMD) per (sample, head).  Staging splits K and V while they go to LDS:
//   K   [SP][72] halves per plane (144-byte rows: the 16 lanes of a ds_read_b128 group land on 16 distinct 16-byte slots)
//   V^T [64][SP + 8] halves per plane, the keys of a 32-key tile stored in the order the probabilities leave the score
//       accumulator (bits 2 and 3 of the key offset swapped): one ds_read_b128 is one MFMA operand
// S = 257: 82,944 + 75,776 (+ 2,112) = 160,832 bytes of the 160 KiB.  A wave owns 32-query tiles qt = wave, wave + 8, ... and walks the
// key tiles with an online softmax, as attention_f32_kernel does:
//   S^T tile  = K tile . Q^T    12 MFMAs 32x32x16 (4 K-steps x 3 products); the lane's q values (/ 8: exact) sit in registers
//               as planes; result: lane = query, register r = key (r & 3) + 8 (r >> 2) + 4 h
//   p = 2^(s log2 e - m) in fp32 (one fma + v_exp_f32; sum l from the fp32 values), then split into planes in place
//   O^T tile += V^T tile . P^T  12 MFMAs: MFMA c of a tile sums over the keys of registers 8c .. 8c + 7
// The output goes out as fp32 and / or as the two fp16 planes the out-projection's split-operand GEMM reads (saves the
// separate split pass over the attention output).  S <= 288.
#include "keds_common.h"
#include <math.h>

#ifndef KEDS_AX_PHASE
#define KEDS_AX_PHASE 0
#endif
#ifndef KEDS_AX_DBG
#define KEDS_AX_DBG 0      // timing only: 1 = staging alone, 2 = no staging, 8 = no MFMAs, 16 = phase stamps (s_memtime) of wave 0 into the fp32 output of the first 256 workgroups
#endif

namespace {

constexpr int AX_WAVES = 8;
constexpr int AX_KP = 72;                    // halves per K row in LDS

__device__ __forceinline__ int ax_vpos(int o) {          // key offset in its 32-key tile -> position in the V^T row
    return (o & ~12) | ((o & 4) << 1) | ((o & 8) >> 1);
}

constexpr float AX_L2E = 1.44269502162933349609375f;     // log2 e
// the two 32-lane halves of a wave hold the two halves of a query's keys: all-VALU exchange (a __shfl_xor is an LDS round trip)
__device__ __forceinline__ float ax_halves_max(float x) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ax_halves_sum(float x) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
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
                                                                       int causal, int q_limit, int* __restrict__ overflow,
                                                                       const int* __restrict__ seq_off) {
    extern __shared__ __attribute__((aligned(16))) f16_t ax_lds[];
    // packed rows (keds_attention_x3_packed; towers.hip PackedRows): sample b owns rows [seq_off[b], seq_off[b + 1]); S was the
    // launch's upper bound (it sized the LDS) and becomes the sample's own length
    long long row_base = (long long)(blockIdx.x / heads) * S;
    if (seq_off) {
        row_base = seq_off[blockIdx.x / heads];
        S = seq_off[blockIdx.x / heads + 1] - (int)row_base;
    }
    const int nkt = (S + 31) >> 5, SP = nkt * 32, VP = SP + 8;
    f16_t* Kh = ax_lds;                                   // [SP][72]
    f16_t* Kl = Kh + (size_t)SP * AX_KP;
    f16_t* Vh = Kl + (size_t)SP * AX_KP;                  // [64][VP]
    f16_t* Vl = Vh + (size_t)64 * VP;
    const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
    const int d = heads * 64, ld = 3 * d;
    const float* base = qkv + (size_t)row_base * ld + hd * 64;
    bool bad = false;
    [[maybe_unused]] unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define AX_STAMP(i) do { if (KEDS_AX_DBG & 16) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#if KEDS_AX_PHASE
    // the first round's workgroups start staggered (quarters of ~KEDS_AX_PHASE us): every CU stages K / V from HBM and then computes
    // with the memory idle -- in lock step the chip alternates between the two; staggered, one CU's staging runs under another's MFMAs
    if (blockIdx.x < 256 && (blockIdx.x & 3)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)(blockIdx.x & 3) * (KEDS_AX_PHASE * 450ull);
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    AX_STAMP(0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 31, hh = lane >> 5;
    const int nq = q_limit < S ? q_limit : S;
    const int nqt = (nq + 31) >> 5;
    // A last query tile that holds ONE query (ViT: S = 257 = 8 x 32 + 1; the CLS-only call of the last block: 1) would cost its
    // wave a full walk over the keys for one useful lane.  All waves share it instead: wave w takes key tiles w, w + 8, ...,
    // the partial (reference point, sum, output row) meet in LDS.  (S <= 288: at most one own tile per wave then.)
    const bool lone = !causal && (nq & 31) == 1;
    const int nqt_own = lone ? nqt - 1 : nqt;
    float* part = reinterpret_cast<float*>(Vl + (size_t)64 * VP);                // [AX_WAVES][66]: reference point, sum, 64 output dims
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    struct QRaw {
        f32x4 v[8];                                                              // dims 16 t + 8 h + 0..7 of the lane's query: v[2 t], v[2 t + 1]
    };
    auto fetch_q = [&](int q) {
        QRaw r;
        const float* qrow = base + (size_t)(q < S ? q : S - 1) * ld + 8 * hh;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            r.v[2 * t] = *reinterpret_cast<const f32x4*>(qrow + 16 * t);
            r.v[2 * t + 1] = *reinterpret_cast<const f32x4*>(qrow + 16 * t + 4);
        }
        return r;
    };
    // the wave's first own query tile and the shared query travel while K and V are staged
#ifndef KEDS_AX_QPF
#define KEDS_AX_QPF 2      // (A/B) bit 0: the own tile's q requested before the staging, bit 1: the shared query's
#endif
    QRaw q_own, q_lone;
    if (KEDS_AX_QPF & 1) q_own = fetch_q((wave < nqt_own ? wave : 0) * 32 + j);
    if (KEDS_AX_QPF & 2) q_lone = fetch_q(nq - 1);
    // ---- staging: one item = two adjacent keys x four dims (adjacent keys are adjacent in the V^T row).  (All of a thread's
    // loads in flight before its first conversion -- 80 registers -- is SLOWER: 22 k instead of 15 k cycles, r05_x3_attention_stamps.)
    for (int idx = threadIdx.x; idx < ((KEDS_AX_DBG & 2) ? 0 : (SP >> 1) * 16); idx += 64 * AX_WAVES) {
        const int c4 = idx & 15, row = 2 * (idx >> 4);
        f32x4 k0 = f32x4{0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
        if (row < S) {
            k0 = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + d + 4 * c4);
            v0 = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 2 * d + 4 * c4);
        }
        if (row + 1 < S) {
            k1 = *reinterpret_cast<const f32x4*>(base + (size_t)(row + 1) * ld + d + 4 * c4);
            v1 = *reinterpret_cast<const f32x4*>(base + (size_t)(row + 1) * ld + 2 * d + 4 * c4);
        }
        f16x4 h0, l0, h1, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bad = bad || !(fabsf(k0[e]) < 65504.0f) || !(fabsf(k1[e]) < 65504.0f) || !(fabsf(v0[e]) < 65504.0f) || !(fabsf(v1[e]) < 65504.0f);
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
    AX_STAMP(1);
    __syncthreads();
    AX_STAMP(2);
    if (KEDS_AX_DBG & 1) return;
    // (half a key tile of start skew between the two waves of a SIMD, s_sleep 6 .. 40 for waves 4-7: no difference, 273-278 us)
    f16x8 qh[4], ql[4];
    float m, l;
    f32x16 o0, o1;
    auto split_q = [&](const QRaw& r) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e < 4 ? r.v[2 * t][e] : r.v[2 * t + 1][e - 4]) * 0.125f;   // 1 / sqrt(64): exact scaling
                bad = bad || !(fabsf(v) < 65504.0f);
                f16_t x, y;
                ax_split(v, x, y);
                qh[t][e] = x, ql[t][e] = y;
            }
        m = -1.0e30f, l = 0.f;                                                   // (finite: -inf - -inf would be NaN)
#pragma unroll
        for (int e = 0; e < 16; ++e) o0[e] = 0.f, o1[e] = 0.f;
    };
    // one key tile against the wave's query tile (q: the lane's query row, for the causal mask)
    auto key_tile = [&](int kt, int q) {
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
        const bool edge = (kt * 32 + 32 > S) || (causal && kt * 32 + 31 > (q & ~31));   // (wave-uniform) a tile with masked keys
        if (edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool valid = key < S && (!causal || key <= q);
                sc[r] = valid ? sc[r] : -INFINITY;
            }
        }
        float tmax = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
        const float tl = ax_halves_max(tmax) * AX_L2E;
        // The reference point m (log2 units) only moves when some query's maximum has grown by more than 8: until then the
        // probabilities are taken relative to the old one (<= 2^8: exact scaling, the planes keep their relative precision) and
        // the rescale of the 32 output registers is skipped -- the same sums, one wave-uniform branch.
        if (__builtin_amdgcn_ballot_w64(tl > m + 8.0f)) {
            const float mnew = fmaxf(m, tl);
            const float resc = __builtin_amdgcn_exp2f(m - mnew);                 // first tile: 2^(-1e30 - m) = 0
            l *= resc;
#pragma unroll
            for (int e = 0; e < 16; ++e) o0[e] *= resc, o1[e] *= resc;
            m = mnew;
        }
        // p = 2^(s log2 e - m): ONE fma + v_exp_f32 per score.  The fma rounds (s log2 e - m) once, so the error is relative to the
        // distance from the reference point, not to |s| -- and a probability with a large distance weighs nothing; the rounding of
        // m itself is a common factor of the row that the final 1 / l removes.  Masked keys: fma(-inf, ., .) = -inf -> 0.
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], AX_L2E, -m));
            psum += sc[r];
        }
        l += psum;
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
    };
    // lane = query q; register r of a0 / a1 = dim (r & 3) + 8 (r >> 2) + 4 h (+ 32)
    auto store_tile = [&](int q, const f32x16& a0, const f32x16& a1) {
        if (q >= nq) return;
        const size_t off = ((size_t)row_base + q) * d + hd * 64 + 4 * hh;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& o = blk ? a1 : a0;
                const f32x4 v = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                if (out && !(KEDS_AX_DBG & 16)) *reinterpret_cast<f32x4*>(out + off + 32 * blk + 8 * g) = v;
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
    };
    f32x16 r0, r1;                                                               // the own tile's result, stored behind the shared query's barrier
#pragma unroll
    for (int e = 0; e < 16; ++e) r0[e] = 0.f, r1[e] = 0.f;
    for (int qt = wave; qt < nqt_own; qt += AX_WAVES) {
        const int q = qt * 32 + j;
        split_q((qt == wave && (KEDS_AX_QPF & 1)) ? q_own : fetch_q(q));
        AX_STAMP(3);
        int kt_end = nkt;
        if (causal) {                                                            // key tiles that hold a key <= the tile's last query
            const int lastq = qt * 32 + 31;
            kt_end = (lastq >> 5) + 1 < nkt ? (lastq >> 5) + 1 : nkt;
        }
        for (int kt = 0; kt < kt_end; ++kt) key_tile(kt, q);
        AX_STAMP(4);
        const float inv = 1.0f / ax_halves_sum(l);
#pragma unroll
        for (int e = 0; e < 16; ++e) r0[e] = o0[e] * inv, r1[e] = o1[e] * inv;
        if (!lone) store_tile(q, r0, r1);          // (with a shared query: no store in flight at its barrier -- a barrier waits for them)
    }
    AX_STAMP(5);
    if (lone) {                                                                  // (uniform over the workgroup)
        const int q = nq - 1;
        split_q((KEDS_AX_QPF & 2) ? q_lone : fetch_q(q));                        // every lane of the tile holds the one query
        for (int kt = wave; kt < nkt; kt += AX_WAVES) key_tile(kt, q);
        const float lt = ax_halves_sum(l);
        if (j == 0) {                                                            // lanes 0 and 32: dims (r & 3) + 8 (r >> 2) + 4 h (+ 32)
            float* pw = part + wave * 66;
            if (hh == 0) pw[0] = m, pw[1] = lt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pw[2 + (r & 3) + 8 * (r >> 2) + 4 * hh] = o0[r];
                pw[2 + 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] = o1[r];
            }
        }
        AX_STAMP(6);
        __syncthreads();
        if (wave < nqt_own) {
            if (pair && !out) {
                // Planes only (every block but the last): a lane holds 4-dim pieces of one query -- 32 stores of 8 bytes, each
                // instruction a scatter of partial lines.  K / V are dead behind the barrier: the wave turns its 32 x 64 tile
                // through its own 9 KiB of their LDS (144-byte rows) and stores 16 bytes per lane, eight whole 128-byte rows per
                // instruction, 4 + 4 instead of 32 + 32.
                f16_t* tl = ax_lds + (size_t)wave * (2 * 32 * AX_KP);            // [2 planes][32 queries][72 halves]
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& o = blk ? r1 : r0;
                        f16x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f16_t x, y;
                            ax_split(o[4 * g + e], x, y);
                            vh[e] = x, vl[e] = y;
                        }
                        f16_t* w = tl + (size_t)j * AX_KP + 32 * blk + 8 * g + 4 * hh;
                        *reinterpret_cast<f16x4*>(w) = vh;
                        *reinterpret_cast<f16x4*>(w + 32 * AX_KP) = vl;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the wave's own writes; no other wave touches this region)
                const size_t row0 = ((size_t)row_base + wave * 32) * d + hd * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + (lane >> 3), ch = lane & 7;          // (lone mode: all 32 queries of an own tile exist)
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(tl + (size_t)row * AX_KP + 8 * ch);
                    const f16x8 vl = *reinterpret_cast<const f16x8*>(tl + (size_t)(32 + row) * AX_KP + 8 * ch);
                    *reinterpret_cast<f16x8*>(pair + row0 + (size_t)row * d + 8 * ch) = vh;
                    *reinterpret_cast<f16x8*>(pair + plane + row0 + (size_t)row * d + 8 * ch) = vl;
                }
            } else {
                store_tile(wave * 32 + j, r0, r1);
            }
        }
        if (wave == AX_WAVES - 1) {                                              // lane = output dim
            float mm = -1.0e30f;
#pragma unroll
            for (int w = 0; w < AX_WAVES; ++w) mm = fmaxf(mm, part[w * 66]);     // (a wave without a key tile left -1e30, 0, 0)
            float lsum = 0.f, acc = 0.f;
#pragma unroll
            for (int w = 0; w < AX_WAVES; ++w) {
                const float f = __builtin_amdgcn_exp2f(part[w * 66] - mm);       // reference points are in log2 units
                lsum = fmaf(part[w * 66 + 1], f, lsum);
                acc = fmaf(part[w * 66 + 2 + lane], f, acc);
            }
            const float v = acc / lsum;
            const size_t off = ((size_t)row_base + q) * d + hd * 64 + lane;
            if (out && !(KEDS_AX_DBG & 16)) out[off] = v;
            if (pair) {
                f16_t x, y;
                ax_split(v, x, y);
                pair[off] = x, pair[plane + off] = y;
            }
        }
    }
    AX_STAMP(7);
    if ((KEDS_AX_DBG & 16) && out && blockIdx.x < 256 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(out)[blockIdx.x * 8 + i] = ts[i];
    if (bad && overflow) *overflow = 1;                   // |q / 8|, |k| or |v| >= 65504: no fp16 hi plane (the caller falls back)
}

}  // namespace

// qkv fp32 [B, S, 3 * heads * 64]; out fp32 [B, S, heads * 64] (nullable); pair: fp16 planes [2][plane] of the same rows
// (nullable; plane >= B * S * heads * 64 elements); at least one of them.  q_limit > 0: the first q_limit queries of every sample
// only (the CLS query of the last ViT block).  overflow (nullable): raised when an operand does not fit an fp16 hi plane.
// seq_off (nullable; device int32 [B + 1]): packed rows -- sample b is rows [seq_off[b], seq_off[b + 1]), S the longest sample
int keds_attention_x3_impl(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal, int q_limit,
                           int* overflow, const int32_t* seq_off, void* stream) {
    KEDS_REQUIRE(qkv && (out || pair) && B > 0 && heads > 0, "keds_attention_x3: bad argument");
    KEDS_REQUIRE(S >= 1 && S <= 288, "keds_attention_x3: S must be in [1, 288] (got %d)", S);
    KEDS_REQUIRE(!pair || seq_off || plane >= (int64_t)B * S * heads * 64, "keds_attention_x3: plane stride shorter than the output");
    const int SP = (S + 31) / 32 * 32;
    const int lds = 2 * (SP * AX_KP + 64 * (SP + 8)) * (int)sizeof(f16_t) + AX_WAVES * 66 * (int)sizeof(float);
    int rc = keds_func_lds_once((const void*)attention_x3_kernel, lds, "attention_x3_kernel");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    attention_x3_kernel<<<B * heads, 64 * AX_WAVES, lds, st>>>(qkv, out, (f16_t*)pair, plane, S, heads, causal,
                                                                q_limit > 0 ? q_limit : S, overflow, seq_off);
    return keds_check_launch("attention_x3_kernel");
}

extern "C" int keds_attention_x3(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal,
                                 int q_limit, int* overflow, void* stream) {
    return keds_attention_x3_impl(qkv, out, pair, plane, B, S, heads, causal, q_limit, overflow, nullptr, stream);
}
