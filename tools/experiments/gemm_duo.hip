// out[M,N] = epilogue(X[M,K] . W[N,K]^T + bias) with the EPILOGUE UNDER THE MFMAs (round 5).
// (reference: src/model/model.py:305-326 -- in_proj and c_fc of every ResidualAttentionBlock, LayerNorm folded.)
//
// Why.  The persistent 4-wave kernel (gemm.hip, gemm_bt_quad_kernel) spends 30 % of every 256 x 256 tile with the matrix pipe
// idle: prologue 4.3 k + K-loop 36.0 k + epilogue 10.4 k + drain 1.1 k cycles on the qkv shape (profiles/r03_gemm_stamps.txt).
// With the epilogue REMOVED the same launches take 13-33 % less time (profiles/r05_gemm_noepi_bound.txt: qkv 173 -> 143 us, c_fc
// 230 -> 188, out-proj 69 -> 46, c_proj 204 -> 176), and the K-loop's gaps carry one vector instruction per MFMA pair for
// +1.4-4.4 %.  At this part's power cap throughput is bought with matrix-pipe utilisation (a lower clock at a lower voltage
// point for the same MFMAs), so the epilogue has to run BESIDE MFMAs, and one wave per SIMD can only do that if the
// accumulators it reads back are not the ones the MFMAs write:
//
//   * a UNIT is 128 rows x 256 columns (half a 256 x 256 tile: the two units of a tile share the W panel, the operand every
//     workgroup of the XCD re-reads from L2); a wave owns all 128 rows x 64 columns = 8 x 4 MFMA tiles = 128 accumulators;
//   * the 256 AGPRs hold TWO such sets.  Unit u accumulates into set u & 1 while the epilogue of unit u - 1 reads set
//     (u - 1) & 1 back, four registers per K-step ("sub-slice" (mi, j): one 16 x 16 tile = 4 consecutive columns of 16 rows),
//     computes LayerNorm affine / QuickGELU / bf16 and stores 16 bytes per lane every other K-step -- all of it single vector
//     instructions placed in the gaps behind MFMA pairs (an MFMA 16x16x32 holds the SIMD's issue port 8 of its 16 cycles);
//   * the K-tile stream never stops: LDS is a THREE-deep ring of 48 KiB K-tiles (X 128 x 64 + W 256 x 64); K-tile q + 2 is
//     requested during K-tile q, across unit and tile boundaries (the next unit's first two K-tiles go out during this unit's
//     last two), six LDS-DMA pieces per K-step, one counted wait + barrier per K-tile.  There is no prologue per tile, no
//     store burst, no drain -- the price is 96 KiB of fragment reads and 48 DMA pieces per 64 K of a 128 x 256 unit where the
//     256 x 256 tile needs 128 KiB and 64 pieces for twice the area (the 8-wave kernel's LDS traffic).
//
// The literal-register asm (gemm_duo_gen.h) is the only code that touches AGPRs; tools/check_quad_asm.py checks that.
#include "keds_common.h"
#include <cstdlib>
#include "gemm_shared.h"
#include "gemm_duo_gen.h"

// TIMING ONLY (make EXTRA="-DKEDS_DUO_DBG=n", tools/rounds/r05_duo_ablate.sh; results are wrong): bit 0 = no sub-slices (the K-loop alone),
// bit 1 = no DMA pieces, bit 2 = no fragment reads, bit 3 = sub-slices without their stores; what a store costs, and why:
// bit 4 = every store of a workgroup into the same 64 KiB (never leaves L2 with the plain policy), bit 5 = lane-linear
// addresses (one store instruction = 1 KiB contiguous = 8 whole lines instead of 16 half lines), bit 6 = every other store only,
// bit 7 = every other DMA piece only (is the stores' cost a queueing effect of a vector-memory path the pieces nearly fill?),
// bit 8 = the K-tile waits leave 16 more operations in flight (is it the in-order vmcnt waiting for a store's acknowledgement?)
#ifndef KEDS_DUO_DBG
#define KEDS_DUO_DBG 0
#endif
// cache policy of the in-loop output stores (keds_common.h, keds_store16: 0 plain, 1 nt, 3 sc1 nt, ...): A/B with EXTRA
#ifndef KEDS_ST_DUO
#define KEDS_ST_DUO KEDS_ST_LN
#endif

namespace {

namespace du {
constexpr int UM = 128, UN = 256, TK = 64;
constexpr int XB = UM * 128;                    // X K-tile: 128 rows x 128 bytes
constexpr int WB = UN * 128;                    // W K-tile: 256 rows x 128 bytes
constexpr int BUF = XB + WB;                    // 48 KiB
constexpr int SIDE0 = 3 * BUF;                  // two side areas: float2 {rstd, -mean rstd}[128 rows] | bias'[256] | colsum[256]
constexpr int SIDE_BYTES = 3072;
constexpr int LDS_BYTES = SIDE0 + 2 * SIDE_BYTES;   // 150 KiB
constexpr int PEEL = 16;                        // K-tiles of a unit that carry the previous unit's 32 sub-slices
}  // namespace du

// The previous unit's epilogue, carried through the K-loop of the current one (everything in registers / SGPRs).
struct DuoEpi {
    float a[4];                 // the sub-slice's accumulators (row 16 mi + c, 4 consecutive columns), then its values
    float t[4];
    float rs, nm;               // row coefficients of row 16 mi + c
    unsigned pk[4];             // packed bf16 of the (mi, pp) pair: [0..1] from j = 2 pp, [2..3] from j = 2 pp + 1
    f32x4 bc[4], cc[4];         // bias' / column sums of the lane's 16 columns: [j]
    const char* side;           // that unit's side area (LDS)
    char* out_tile;             // wave-uniform: first byte of that unit's output rows / columns
    char* dbg_base;             // (KEDS_DUO_DBG bit 4 only)
    unsigned lane_off;          // (c N + 64 w + 8 g) * 2
    unsigned nrec;              // (uniform) records of the store descriptor: 0 for the first unit of a workgroup -- the range check
                                // then DROPS the store, but it is still issued and counted (the K-tile waits count it)
};

// Single vector instructions (the compiler packs adjacent fp32 multiplies / adds / FMAs into v_pk_*_f32 under -O3, and a packed
// instruction beside MFMAs costs ~22 cycles more than the two plain ones it replaces: MI355X_MICROARCH.md, cycle constants).
// Same IEEE results as the packed forms of pair_ln_epilogue (gemm.hip).
__device__ __forceinline__ float fma1(float a, float b, float c) {
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float mul1(float a, float b) {
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float add1(float a, float b) {
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// gap g (0..15) of a K-step carries stage g of sub-slice (MI, J) of the previous unit.  LayerNorm affine:
//   v = acc * rstd + (colsum * (-mean rstd) + bias')          (the expression of pair_ln_epilogue, gemm.hip: same contraction)
// QuickGELU: v * rcp(1 + exp2(-1.702 log2(e) v)).  One kind of instruction per gap, at most four plain or two transcendental
// ones (8 issue cycles each) beside the gap's memory instruction.
// INLOOP = false (the last unit's epilogue, nothing beside it): plain C++ arithmetic -- there the stages of a sub-slice follow
// each other directly, and the hazard recogniser does not see an inline-asm consumer of a transcendental result (no wait state
// between v_exp_f32 / v_rcp_f32 and the asm instruction that reads it: measured as 16 wrong elements per 16 x 16 block of the
// first pair, profiles/r05_duo_debug_v1.txt); inside the K-loop two MFMAs sit between any two stages.
template <int EPI, int G, int J, bool INLOOP = true>
__device__ __forceinline__ void duo_epi_gap(DuoEpi& e, int mi, int N) {
    constexpr bool GELU = epi_base(EPI) == KEDS_EPI_BIAS_QGELU_BF16;
    auto fma_ = [](float a, float b, float c) { return INLOOP ? fma1(a, b, c) : __builtin_fmaf(a, b, c); };
    auto mul_ = [](float a, float b) { return INLOOP ? mul1(a, b) : a * b; };
    auto add_ = [](float a, float b) { return INLOOP ? add1(a, b) : a + b; };
    if constexpr (G == 0) {
        const int c = threadIdx.x & 15;
        const f32x2 cf = *reinterpret_cast<const f32x2*>(e.side + (16 * mi + c) * 8);
        e.rs = cf[0];
        e.nm = cf[1];
    }
    // (G == 1: the accumulator read-back, literal registers: KEDS_DUO_READ in the step macro)
    if constexpr (G == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e.t[r] = fma_(e.cc[J][r], e.nm, e.bc[J][r]);
    }
    if constexpr (G == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e.a[r] = fma_(e.a[r], e.rs, e.t[r]);
    }
    if constexpr (GELU) {
        if constexpr (G == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r) e.t[r] = mul_(e.a[r], -2.4554669595930157f);
        }
        if constexpr (G == 6 || G == 7) {
#pragma unroll
            for (int r = 2 * (G - 6); r < 2 * (G - 6) + 2; ++r) e.t[r] = __builtin_amdgcn_exp2f(e.t[r]);
        }
        if constexpr (G == 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) e.t[r] = add_(e.t[r], 1.0f);
        }
        if constexpr (G == 9 || G == 10) {
#pragma unroll
            for (int r = 2 * (G - 9); r < 2 * (G - 9) + 2; ++r) e.t[r] = __builtin_amdgcn_rcpf(e.t[r]);
        }
        if constexpr (G == 11) {
#pragma unroll
            for (int r = 0; r < 4; ++r) e.a[r] = mul_(e.a[r], e.t[r]);
        }
    }
    if constexpr (G == (GELU ? 12 : 5)) {
        const bf16x2 p0 = bf16x2{(bf16_t)e.a[0], (bf16_t)e.a[1]}, p1 = bf16x2{(bf16_t)e.a[2], (bf16_t)e.a[3]};
        e.pk[2 * (J & 1)] = __builtin_bit_cast(unsigned, p0);
        e.pk[2 * (J & 1) + 1] = __builtin_bit_cast(unsigned, p1);
    }
    if constexpr (G == 15 && (J & 1)) {
        // 8 consecutive columns of row 16 mi + c: the 16-byte store of pair_ln_epilogue.  The LAST memory instruction of its
        // K-step, behind the step's last DMA piece: vmcnt retires in order and counts stores, and a store's acknowledgement takes
        // longer than a K-tile -- issued in front of pieces the next K-tile wait needs, every such wait also waited for the
        // store (first build: 192 us on the qkv shape against 161 without the stores, profiles/r05_duo_ablate_v1.txt).  Here it
        // is YOUNGER than every piece that wait is for, and the wait leaves it in flight (vmcnt(7)).
        if constexpr (!(KEDS_DUO_DBG & 8)) {
            constexpr int aux = KEDS_ST_DUO == 1 ? 2 : KEDS_ST_DUO == 2 ? 17 : KEDS_ST_DUO == 3 ? 18 : KEDS_ST_DUO == 4 ? 16 : 0;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(e.out_tile, 0, e.nrec, 0x00020000);
            if constexpr (KEDS_DUO_DBG & 16) {
                const auto r2 = __builtin_amdgcn_make_buffer_rsrc(e.dbg_base, 0, e.nrec, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{e.pk[0], e.pk[1], e.pk[2], e.pk[3]}, r2, (int)((threadIdx.x & 255) * 16),
                                                       ((2 * mi + (J >> 1)) & 15) * 4096, aux);
            } else if constexpr (KEDS_DUO_DBG & 32) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{e.pk[0], e.pk[1], e.pk[2], e.pk[3]}, rs, (int)((threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024),
                                                       (16 * mi * N + 4096 * (J >> 1)) * 2, aux);
            } else if constexpr (KEDS_DUO_DBG & 64) {
                if (J == 1)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{e.pk[0], e.pk[1], e.pk[2], e.pk[3]}, rs, (int)e.lane_off,
                                                           (16 * mi * N + 32 * (J >> 1)) * 2, aux);
            } else
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{e.pk[0], e.pk[1], e.pk[2], e.pk[3]}, rs, (int)e.lane_off,
                                                   (16 * mi * N + 32 * (J >> 1)) * 2, aux);
        }
    }
}

#define make_rs(base) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(base)), 0, 0x7FFFFFFF, 0x00020000)

// memory instruction(s) of gap n: pieces in gaps 0, 3, .., 15 (piece n / 3 of the step's six), the twelve fragment reads of the
// NEXT K-step in the other ten (X fragments 0-7, then W fragments 0-3; gaps 1 and 4 take two)
#define KEDS_DRD(r_, xn, wn_)                                                                                          \
    if constexpr ((r_) < 8) xn[(r_)] = *reinterpret_cast<const bf16x8*>(smem + xa_ + (r_) * 2048);             \
    else wn_[(r_) - 8] = *reinterpret_cast<const bf16x8*>(smem + wa_ + ((r_) - 8) * 2048);
#define KEDS_DMEM(n_, xn, wn_, IH)                                                                             \
    if constexpr ((n_) % 3 == 0) { if constexpr (!(KEDS_DUO_DBG & 2) && !((KEDS_DUO_DBG & 128) && ((n_) / 3) % 2)) issue(6 * (IH) + (n_) / 3) } \
    else if constexpr (KEDS_DUO_DBG & 4) {}                                                                    \
    else if constexpr ((n_) == 1) { KEDS_DRD(0, xn, wn_) KEDS_DRD(1, xn, wn_) }                                                  \
    else if constexpr ((n_) == 2) { KEDS_DRD(2, xn, wn_) }                                                              \
    else if constexpr ((n_) == 4) { KEDS_DRD(3, xn, wn_) KEDS_DRD(4, xn, wn_) }                                                  \
    else if constexpr ((n_) == 5) { KEDS_DRD(5, xn, wn_) }                                                              \
    else if constexpr ((n_) == 7) { KEDS_DRD(6, xn, wn_) }                                                              \
    else if constexpr ((n_) == 8) { KEDS_DRD(7, xn, wn_) }                                                              \
    else if constexpr ((n_) == 10) { KEDS_DRD(8, xn, wn_) }                                                             \
    else if constexpr ((n_) == 11) { KEDS_DRD(9, xn, wn_) }                                                             \
    else if constexpr ((n_) == 13) { KEDS_DRD(10, xn, wn_) }                                                            \
    else if constexpr ((n_) == 14) { KEDS_DRD(11, xn, wn_) }

#define KEDS_DMFMA(S, FIRST, j, mi, wc, xc)                                                                    \
    if constexpr (FIRST) {                                                                                     \
        if constexpr (epi_f16(EPI)) { KEDS_DUO_MFMAZ_##S##_##j##_##mi("v_mfma_f32_16x16x32_f16", wc[j], xc[mi]) } \
        else { KEDS_DUO_MFMAZ_##S##_##j##_##mi("v_mfma_f32_16x16x32_bf16", wc[j], xc[mi]) }                    \
    } else {                                                                                                   \
        if constexpr (epi_f16(EPI)) { KEDS_DUO_MFMA_##S##_##j##_##mi("v_mfma_f32_16x16x32_f16", wc[j], xc[mi]) } \
        else { KEDS_DUO_MFMA_##S##_##j##_##mi("v_mfma_f32_16x16x32_bf16", wc[j], xc[mi]) }                     \
    }
// gap n behind the MFMA pair (j; a, b): memory instruction, the sub-slice's stage n, then the scheduling barrier that makes
// program order issue order
#define KEDS_DGAP(n_, xn, wn_, IH, SL, PS, MI, J)                                                              \
    KEDS_DMEM(n_, xn, wn_, IH)                                                                                 \
    if constexpr (SL && !(KEDS_DUO_DBG & 1)) {                                                                 \
        if constexpr ((n_) == 1) { KEDS_DUO_READ_##PS##_##J##_##MI(e.a[0], e.a[1], e.a[2], e.a[3]) }           \
        duo_epi_gap<EPI, n_, J>(e, MI, N);                                                                     \
    }                                                                                                          \
    if constexpr ((n_) == 2) { hook(); }                                                                       \
    __builtin_amdgcn_sched_barrier(0);
#define KEDS_DGROUP(S, FIRST, j, xc, wc, xn, wn_, IH, SL, PS, MI, J)                                           \
    KEDS_DMFMA(S, FIRST, j, 0, wc, xc) KEDS_DMFMA(S, FIRST, j, 1, wc, xc) KEDS_DGAP(4 * (j), xn, wn_, IH, SL, PS, MI, J)     \
    KEDS_DMFMA(S, FIRST, j, 2, wc, xc) KEDS_DMFMA(S, FIRST, j, 3, wc, xc) KEDS_DGAP(4 * (j) + 1, xn, wn_, IH, SL, PS, MI, J) \
    KEDS_DMFMA(S, FIRST, j, 4, wc, xc) KEDS_DMFMA(S, FIRST, j, 5, wc, xc) KEDS_DGAP(4 * (j) + 2, xn, wn_, IH, SL, PS, MI, J) \
    KEDS_DMFMA(S, FIRST, j, 6, wc, xc) KEDS_DMFMA(S, FIRST, j, 7, wc, xc) KEDS_DGAP(4 * (j) + 3, xn, wn_, IH, SL, PS, MI, J)
// One K-step: 32 MFMAs of set S from (xc, wc); the 12 fragment reads of the NEXT K-step go to (xn, wn_) from LDS byte offset
// `rb` at chunk offset `nslot`; six DMA pieces (half IH of the issue cursor's K-tile); SL: sub-slice (MI, J) of set PS.
// SYNC: the K-tile the next step reads has landed and every wave is done with the buffer the pieces go to -- the youngest
// memory operations stay in flight: this K-tile's first six pieces and, in front of them, the output store that closed the
// previous K-tile when that one carried sub-slices (`pend_store`, uniform).
#define KEDS_DUO_STEP(S, FIRST, xc, wc, xn, wn_, rb, nslot, SYNC, IH, SL, PS, MI, J)                           \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr (SYNC) {                                                                                  \
            if (pend_store) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(((KEDS_DUO_DBG & 128) ? 4 : 7) + ((KEDS_DUO_DBG & 256) ? 16 : 0)) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(((KEDS_DUO_DBG & 128) ? 3 : 6) + ((KEDS_DUO_DBG & 256) ? 16 : 0)) : "memory"); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const int xa_ = xlane + (rb) + (nslot), wa_ = wlane + (rb) + (nslot);                                  \
        const auto xrs_ = make_rs(xbase_i), wrs_ = make_rs(wbase_i);                                           \
        KEDS_DGROUP(S, FIRST, 0, xc, wc, xn, wn_, IH, SL, PS, MI, J)                                           \
        KEDS_DGROUP(S, FIRST, 1, xc, wc, xn, wn_, IH, SL, PS, MI, J)                                           \
        KEDS_DGROUP(S, FIRST, 2, xc, wc, xn, wn_, IH, SL, PS, MI, J)                                           \
        KEDS_DGROUP(S, FIRST, 3, xc, wc, xn, wn_, IH, SL, PS, MI, J)                                           \
    }
// One K-tile = steps (p, 0) and (p, 1); with SL the (MI, PP) pair of the previous unit: sub-slices J = 2 PP and 2 PP + 1.
// HOOK (run in gap 2 of the second step, i.e. in a SYNC step in front of its last piece -- it may issue memory operations):
// 1 = request this unit's side data, 2 = turn it into the side area.
#define KEDS_DUO_KTILE(S, FIRST, SL, PS, MI, JA, JB, HOOK)                                                     \
    {                                                                                                          \
        {                                                                                                      \
            auto hook = [&]() {};                                                                              \
            KEDS_DUO_STEP(S, FIRST, xa, wa, xb, wb, rb_cur, slot1, false, 0, SL, PS, MI, JA)                   \
        }                                                                                                      \
        {                                                                                                      \
            auto hook = [&]() {                                                                                \
                if constexpr ((HOOK) == 1) side_request(m0u, n0);                                              \
                if constexpr ((HOOK) == 2) side_write(smem + du::SIDE0 + (S) * du::SIDE_BYTES, m0u, n0);       \
            };                                                                                                 \
            KEDS_DUO_STEP(S, false, xb, wb, xa, wa, rb_nxt, slot0, true, 1, SL, PS, MI, JB)                    \
        }                                                                                                      \
        advance_issue()                                                                                        \
        pend_store = (SL) && !(KEDS_DUO_DBG & 9);                                                              \
        const int t_ = rb_cur;                                                                                 \
        rb_cur = rb_nxt;                                                                                       \
        rb_nxt = rb_nn;                                                                                        \
        rb_nn = t_;                                                                                            \
    }

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bt_duo_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, void* __restrict__ out,
                                                             int M, int N, int K, int n_tiles,
                                                             const float* __restrict__ aux, void* __restrict__ aux2,
                                                             int* __restrict__ guard, int ntiles) {
    static_assert(epi_is_ln(EPI), "two-accumulator-set kernel: LayerNorm-folded epilogues");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m_tiles = ntiles / n_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int grid = (int)gridDim.x;
    const int np = K / du::TK;                                    // >= du::PEEL (the launcher checks)

    // ---- staging: piece j (8 LDS rows = 1 KiB) of an operand; this wave owns pieces wave + 4 i: X i < 4, W i < 8
    const int R0 = 8 * wave + (lane >> 3);                        // 0..31
    const int sch = (lane & 7) ^ swz_f(R0);                       // swz_f(R0 + 32 i) == swz_f(R0)
    const unsigned xoff = (unsigned)R0 * (unsigned)K * 2u + sch * 16;
    const unsigned woff = (unsigned)perm_w(R0) * (unsigned)K * 2u + sch * 16;   // perm_w(R0 + 32 i) == perm_w(R0) + 32 i
    const unsigned rstride = 32u * (unsigned)K * 2u;

    // ---- the issue cursor: the K-tile whose pieces go out next (two K-tiles ahead of the MFMAs, across units and tiles)
    int i_id = blockIdx.x, i_half = 0, i_m0, i_n0;
    {
        int tm, tn;
        quad_tile_coords(xcd_remap(i_id, ntiles), m_tiles, n_tiles, tm, tn);
        i_m0 = tm * 256;
        i_n0 = tn * 256;
    }
    // (the cursor is plain scalars -- operand base pointers, K offset, LDS buffer -- and the descriptors are rebuilt from the
    // pointers once per K-step: descriptor VARIABLES that a lambda re-points end up in scratch, reloaded through a
    // v_readfirstlane waterfall loop per piece)
    const char* xbase_i = reinterpret_cast<const char*>(X + (size_t)i_m0 * K);
    const char* wbase_i = reinterpret_cast<const char*>(W + (size_t)i_n0 * K);
    unsigned koff_i = 0;
    int ib = 0;                                                   // LDS byte offset of the buffer the cursor's K-tile goes to
    // piece q of the cursor's K-tile: X pieces 0..3, W pieces 4..11 (xrs_ / wrs_: the enclosing step's descriptors)
#define issue(q)                                                                                               \
    {                                                                                                          \
        if constexpr ((q) < 4)                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs_, (__attribute__((address_space(3))) void*)(smem + ib + (wave + 4 * (q)) * 1024), 16, \
                                                     xoff, (q) * rstride + koff_i, 0, 0);                     \
        else                                                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs_, (__attribute__((address_space(3))) void*)(smem + ib + du::XB + (wave + 4 * ((q) - 4)) * 1024), 16, \
                                                     woff, ((q) - 4) * rstride + koff_i, 0, 0);               \
    }
    // behind a K-tile's twelve pieces: the next K-tile of the unit, or (uniform branch) the next unit -- the tile's lower
    // half, or the next tile; when nothing is left the last unit's K-tiles are requested again, unused
#define advance_issue()                                                                                        \
    {                                                                                                          \
        koff_i += du::TK * 2;                                                                                  \
        ib = ib == 2 * du::BUF ? 0 : ib + du::BUF;                                                             \
        if (koff_i == (unsigned)K * 2u) {                                                                      \
            koff_i = 0;                                                                                        \
            if (i_half == 0) {                                                                                 \
                i_half = 1;                                                                                    \
                xbase_i += (size_t)du::UM * K * 2;                                                             \
            } else if (i_id + grid < ntiles) {                                                                 \
                i_id += grid;                                                                                  \
                i_half = 0;                                                                                    \
                int tm_, tn_;                                                                                  \
                quad_tile_coords(xcd_remap(i_id, ntiles), m_tiles, n_tiles, tm_, tn_);                         \
                xbase_i = reinterpret_cast<const char*>(X + (size_t)tm_ * 256 * K);                            \
                wbase_i = reinterpret_cast<const char*>(W + (size_t)tn_ * 256 * K);                            \
            }                                                                                                  \
        }                                                                                                      \
    }

    // ---- fragment read offsets: X rows 16 mi + c, W rows 64 wave + 16 j + c of a K-tile buffer
    const int f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xlane = c * 128;                                     // + mi * 2048
    const int wlane = du::XB + (64 * wave + c) * 128;             // + j * 2048

    // ---- side data of a unit (rows m0u.., columns n0..): requested with inline-asm loads (a plain load would make the compiler
    // wait vmcnt(0) -- for every DMA piece in flight -- at its first use), turned into the LDS side area two K-tiles later
    u32x4 st_raw = u32x4{0, 0, 0, 0};
    float pb = 0.f, pc = 0.f;
    auto side_request = [&](int m0u_, int n0_) {
        const keds_stat_t* sp = reinterpret_cast<const keds_stat_t*>(aux) + 2 * (size_t)(m0u_ + (tid & 127));
        const float* bp = bias + n0_ + tid;
        const float* cp = bp + N;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st_raw) : "v"(sp) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pb) : "v"(bp) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pc) : "v"(cp) : "memory");
    };
    auto side_write = [&](char* side, int m0u_, int n0_) {        // (behind a SYNC wait that retired the three loads)
        asm volatile("" : "+v"(st_raw), "+v"(pb), "+v"(pc));
        if (tid < 128) {
            float rs, nm;
            ln_coeff_from((keds_stat_t)(((unsigned long long)st_raw[1] << 32) | st_raw[0]),
                          (keds_stat_t)(((unsigned long long)st_raw[3] << 32) | st_raw[2]), 1.0f / (float)K, rs, nm,
                          n0_ == 0 ? guard : nullptr);
            *reinterpret_cast<f32x2*>(side + tid * 8) = f32x2{rs, nm};
            // the launch's first column tile clears the OTHER statistics buffer (the next producer accumulates into it)
            if (aux2 && n0_ == 0) keds_stat_zero(reinterpret_cast<keds_stat_t*>(aux2) + 2 * (size_t)(m0u_ + tid));
        }
        *reinterpret_cast<float*>(side + 1024 + tid * 4) = pb;
        *reinterpret_cast<float*>(side + 2048 + tid * 4) = pc;
    };

    // ---- prologue: K-tiles 0 and 1 of the first unit
#define KEDS_DUO_ISSUE_ALL                                                                                     \
    {                                                                                                          \
        const auto xrs_ = make_rs(xbase_i), wrs_ = make_rs(wbase_i);                                           \
        issue(0) issue(1) issue(2) issue(3) issue(4) issue(5) issue(6) issue(7) issue(8) issue(9) issue(10) issue(11) \
    }
    KEDS_DUO_ISSUE_ALL
    advance_issue()
    KEDS_DUO_ISSUE_ALL
    advance_issue()
#undef KEDS_DUO_ISSUE_ALL
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    int rb_cur = 0, rb_nxt = du::BUF, rb_nn = 2 * du::BUF;
    bool pend_store = false;                                      // (uniform) the previous K-tile closed with an output store
    bf16x8 xa[8], wa[4], xb[8], wb[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) xa[mi] = *reinterpret_cast<const bf16x8*>(smem + xlane + slot0 + mi * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[j] = *reinterpret_cast<const bf16x8*>(smem + wlane + slot0 + j * 2048);

    DuoEpi e;
    e.nrec = 0u;
    e.side = smem + du::SIDE0;
    e.out_tile = reinterpret_cast<char*>(out);
    e.dbg_base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * 65536;
    e.lane_off = ((unsigned)c * (unsigned)N + (unsigned)(64 * wave + 8 * g)) * 2u;
    e.rs = e.nm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) e.bc[j] = e.cc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) e.a[r] = e.t[r] = 0.f, e.pk[r] = 0u;

    // the epilogue context of the unit (rows m0u_, columns n0_, side area `area`) that just finished its K-loop
    auto epi_begin = [&](int m0u_, int n0_, int area) {
        e.nrec = 0x7FFFFFFFu;
        e.side = smem + du::SIDE0 + area * du::SIDE_BYTES;
        e.out_tile = reinterpret_cast<char*>(out) + ((size_t)m0u_ * N + n0_) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = 64 * wave + 32 * (j >> 1) + 8 * g + 4 * (j & 1);
            e.bc[j] = *reinterpret_cast<const f32x4*>(e.side + 1024 + nl * 4);
            e.cc[j] = *reinterpret_cast<const f32x4*>(e.side + 2048 + nl * 4);
        }
    };

    // One unit: rows m0u, accumulator set S; the 32 sub-slices of the previous unit (set PS) ride on its first 16 K-tiles.
#define KEDS_DUO_UNIT(S, PS)                                                                                   \
    {                                                                                                          \
        KEDS_DUO_KTILE(S, true, true, PS, 0, 0, 1, 1)                                                          \
        KEDS_DUO_KTILE(S, false, true, PS, 0, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 1, 0, 1, 2)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 1, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 2, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 2, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 3, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 3, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 4, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 4, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 5, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 5, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 6, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 6, 2, 3, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 7, 0, 1, 0)                                                         \
        KEDS_DUO_KTILE(S, false, true, PS, 7, 2, 3, 0)                                                         \
        for (int p = du::PEEL; p < np; ++p) KEDS_DUO_KTILE(S, false, false, PS, 0, 0, 1, 0)                    \
    }

    for (int id = blockIdx.x; id < ntiles; id += grid) {
        int tm, tn;
        quad_tile_coords(xcd_remap(id, ntiles), m_tiles, n_tiles, tm, tn);
        const int n0 = tn * 256;
        {
            const int m0u = tm * 256;                              // unit A: accumulates into set 0; the previous tile's unit B rides along
            KEDS_DUO_UNIT(0, 1)
            epi_begin(m0u, n0, 0);
        }
        {
            const int m0u = tm * 256 + du::UM;                     // unit B: set 1; unit A's epilogue rides along
            KEDS_DUO_UNIT(1, 0)
            epi_begin(m0u, n0, 1);
        }
    }
    // ---- the last unit's epilogue, with nothing beside it
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define KEDS_DUO_TAIL(MI, J)                                                                                   \
    {                                                                                                          \
        duo_epi_gap<EPI, 0, J, false>(e, MI, N);                                                               \
        KEDS_DUO_READ_1_##J##_##MI(e.a[0], e.a[1], e.a[2], e.a[3])                                             \
        duo_epi_gap<EPI, 3, J, false>(e, MI, N); duo_epi_gap<EPI, 4, J, false>(e, MI, N); duo_epi_gap<EPI, 5, J, false>(e, MI, N); \
        duo_epi_gap<EPI, 6, J, false>(e, MI, N); duo_epi_gap<EPI, 7, J, false>(e, MI, N); duo_epi_gap<EPI, 8, J, false>(e, MI, N); \
        duo_epi_gap<EPI, 9, J, false>(e, MI, N); duo_epi_gap<EPI, 10, J, false>(e, MI, N); duo_epi_gap<EPI, 11, J, false>(e, MI, N); \
        duo_epi_gap<EPI, 12, J, false>(e, MI, N); duo_epi_gap<EPI, 15, J, false>(e, MI, N);                    \
    }
#define KEDS_DUO_TAIL_ROW(MI) KEDS_DUO_TAIL(MI, 0) KEDS_DUO_TAIL(MI, 1) KEDS_DUO_TAIL(MI, 2) KEDS_DUO_TAIL(MI, 3)
    KEDS_DUO_TAIL_ROW(0) KEDS_DUO_TAIL_ROW(1) KEDS_DUO_TAIL_ROW(2) KEDS_DUO_TAIL_ROW(3)
    KEDS_DUO_TAIL_ROW(4) KEDS_DUO_TAIL_ROW(5) KEDS_DUO_TAIL_ROW(6) KEDS_DUO_TAIL_ROW(7)
#undef KEDS_DUO_TAIL_ROW
#undef KEDS_DUO_TAIL
#undef KEDS_DUO_UNIT
}

#undef make_rs
#undef issue
#undef advance_issue

constexpr int DUO_DEFAULT = 0;   // (off until its parity and its timing are both green on the GPU: see DESIGN.md section 0)
int duo_env() {          // KEDS_GEMM_DUO=0 / 1 in the environment: the round-4 kernels / this kernel (A/B, bisecting)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("KEDS_GEMM_DUO");
        v = e && e[0] ? (e[0] != '0') : DUO_DEFAULT;
    }
    return v;
}
int g_duo = -1;          // run-time override (keds_gemm_duo_enable): -1 = the environment

template <int EPI>
int launch_duo(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux, void* aux2,
               hipStream_t st) {
    const int m_tiles = M / 256, n_tiles = N / 256, ntiles = m_tiles * n_tiles;
    int cus = keds_device_cus();
    if (cus > 256) cus = 256;
    cus &= ~7;                                   // whole XCD groups: workgroup b and tile ids b, b + grid, ... share an XCD label
    if (cus > ntiles) cus = ntiles;
    if (int rc = keds_func_lds_once((const void*)gemm_bt_duo_kernel<EPI>, du::LDS_BYTES, "gemm_bt_duo_kernel")) return rc;
    gemm_bt_duo_kernel<EPI><<<cus, 256, du::LDS_BYTES, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K, n_tiles, aux,
                                                             aux2, keds_numerics_guard(), ntiles);
    return keds_check_launch("gemm_bt_duo_kernel");
}

}  // namespace

// true when (EPI, M, N, K) runs on the two-accumulator-set kernel: LayerNorm-folded fp16-operand epilogues, whole 256 x 256
// tiles, at least 16 K-tiles (the previous unit's 32 sub-slices ride on a unit's first 32 K-steps)
bool keds_gemm_duo_ok(int epi, int M, int N, int K) {
    const bool on = g_duo < 0 ? duo_env() != 0 : g_duo != 0;
    return on && (epi == KEDS_EPI_LN_BIAS_BF16_H || epi == KEDS_EPI_LN_QGELU_BF16_H) && M > 0 && M % 256 == 0 && N % 256 == 0 &&
           K % 64 == 0 && K / 64 >= du::PEEL;
}

int keds_gemm_duo_launch(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                         void* aux2, hipStream_t st) {
    if (epi == KEDS_EPI_LN_BIAS_BF16_H) return launch_duo<KEDS_EPI_LN_BIAS_BF16_H>(A, W, bias, out, M, N, K, aux, aux2, st);
    return launch_duo<KEDS_EPI_LN_QGELU_BF16_H>(A, W, bias, out, M, N, K, aux, aux2, st);
}

extern "C" int keds_gemm_duo_enable(int on) {
    g_duo = on < 0 ? -1 : (on ? 1 : 0);
    return KEDS_OK;
}
