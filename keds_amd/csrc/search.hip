// Similarity + top-k over a resident database: the replacement for faiss.IndexFlatL2
// (reference: src/eval_retrieval.py:289-298, src/eval_utils.py:169-180, src/trainer.py:246-257).
//
// Pipeline (all on one stream, no host sync):
//   qprep   : fp32 queries -> (optionally L2-normalised) fp32 copy + bf16 block of 128 rows
//   scan    : HBM-bound.  One persistent 512-thread workgroup per CU streams its contiguous
//             range of 32-key stages HBM -> LDS with LDS-DMA (the HBM image IS the LDS image,
//             pre-swizzled at pack time), keeps the 128 queries as MFMA B-operands in
//             registers, and folds every 16x16 score tile into per-lane exact top-16 lists
//             (lane = one query x one quarter of the keys; the list threshold makes inserts rare).
//   merge   : per query, tournament over the sorted per-lane lists -> 64 candidates
//   rerank  : exact fp32 distance of each candidate from the fp32 rows
//   select  : order the 64 by (distance, id) -> top-k  (+ optional row gather)
//
// Score used by the scan: s = q.x - 0.5*||x||^2 (L2; monotone in -||q-x||^2) or q.x (IP);
// the per-key bias enters as the MFMA C-in operand, read from the stage's fp32 tail.
#include "keds_common.h"
#include <math.h>

namespace {

constexpr int STAGE_KEYS = KEDS_SCAN_STAGE_KEYS;  // 32
constexpr int LISTK = KEDS_SCAN_LIST;             // 16
constexpr int NCAND = KEDS_SCAN_CAND;             // 64 candidates re-ranked per query for k <= 16
constexpr int NCAND_WIDE = 256;                   // ... and for 16 < k <= KEDS_SCAN_MAX_K
constexpr int MAXK = KEDS_SCAN_MAX_K;             // 128
constexpr int TRAILER = 128;                      // bytes behind the last stage: DbBounds
// Bounds over the database rows that make the candidate selection CERTIFIABLE (see certify_and_select_kernel):
// xt = max ||bf16(x)||, r = max ||x - bf16(x)||, x = max ||x||, written by the pack kernel with integer atomicMax.
struct DbBounds {
    float xt, r, x, pad;
};
constexpr int QBLOCK = KEDS_SCAN_MAX_QUERIES;     // 128
constexpr int SCAN_THREADS = 512;

template <int D>
struct ScanCfg {
    static constexpr int ROWB = D * 2;                   // bytes per bf16 key row
    static constexpr int CHUNKS = D / 8;                 // 16-byte chunks per row
    static constexpr int KEYB = STAGE_KEYS * ROWB;       // key bytes per stage
    static constexpr int STAGEB = KEYB + 128;            // HBM blob: keys + 32 fp32 biases
    static constexpr int LDS_STAGE = KEYB + 8 * 128;     // LDS slot: keys + 8 wave-private bias copies
    static constexpr int NST_RAW = (160 * 1024) / LDS_STAGE;
    static constexpr int NST = NST_RAW > 8 ? 8 : NST_RAW;  // ring depth
    static constexpr int PIECES = KEYB / 1024;           // 1-KiB DMA pieces per stage
    static constexpr int PPW = PIECES / 8;               // per wave
    static constexpr int PW = PPW + 1;                   // + the bias piece: vm ops per wave per stage
    static constexpr int KSTEPS = D / 32;
    static_assert(D % 128 == 0, "row must be a whole number of 256-byte bank rows");
    static_assert(PIECES % 8 == 0, "pieces must split evenly over 8 waves");
    static_assert(NST >= 2, "need at least a double buffer");
    static_assert((NST - 2) * PW <= 63, "vmcnt immediate");
};

// 16-byte chunk `ch` of key row `row` lives at slot (ch ^ row) in its 16-chunk (256-byte) group:
// 16 lanes reading the same chunk of 16 different rows then hit 16 different 16-byte slots.
__host__ __device__ __forceinline__ int swz_chunk(int ch, int row) { return (ch & ~15) | ((ch ^ row) & 15); }

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int PW, int MAXAHEAD>
__device__ __forceinline__ void wait_stage_and_barrier(int ahead) {
    // `ahead` = stages issued after the one we are about to read (wave-uniform)
    if constexpr (MAXAHEAD == 0) {
        wait_vm_barrier<0>();
    } else {
        if (ahead >= MAXAHEAD) wait_vm_barrier<MAXAHEAD * PW>();
        else wait_stage_and_barrier<PW, MAXAHEAD - 1>(ahead);
    }
}

// ------------------------------------------------------------------------------------------
// DBG (timing-only ablations; results wrong unless 0): 1 = no list update, 2 = no MFMA (and no list update),
// 3 = no LDS fragment reads
// L = per-lane list depth: LISTK (16) for the candidate pass, 4 for the threshold pass
// order-preserving integer image of a score: a > b  <=>  ord_key(a) > ord_key(b); 0 is below every real score's key
__device__ __forceinline__ unsigned ord_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// NT: LDS-DMA cache policy of the key stream (aux = 2: non-temporal -- the image is read once per search and is far larger
// than the caches; measured 3-6 % faster on the whole 0.5 M-row search).
// Output: the VALID entries of the workgroup's four lists of a query (its 16-lane key quarters), packed to the front of the
// workgroup's segment out_pairs[(query * nwg + wg) * 4L ..] as (ord_key(score) << 32 | row id), and
// out_meta[query * nwg + wg] = {entries written, largest key sitting in the LAST slot of a full list (0: none)}.
// With insert thresholds almost every list slot stays empty: the merge then reads a handful of pairs per segment instead
// of all 4L slots of it.
template <int D, int L = LISTK, int DBG = 0, int NT = 0>
__global__ __launch_bounds__(SCAN_THREADS, 2) void scan_topk_kernel(
    const char* __restrict__ packed, int stage_begin, int total_stages, const bf16_t* __restrict__ qb,
    const float* __restrict__ thr, unsigned long long* __restrict__ out_pairs, uint2* __restrict__ out_meta) {
    using C = ScanCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int nwg = gridDim.x;
    const int s0 = stage_begin + (int)(((long long)blockIdx.x * total_stages) / nwg);
    const int s1 = stage_begin + (int)(((long long)(blockIdx.x + 1) * total_stages) / nwg);

    // blockIdx.y = query block of 128; the workgroups of one block split the stage range between them
    const int qglob = blockIdx.y * QBLOCK + wave * 16 + c;
    // queries of this wave as MFMA B operands: lane (g,c) holds Q[16*wave + c][32*s + 8*g .. +7]
    bf16x8 qf[C::KSTEPS];
    {
        const bf16_t* qrow = qb + (size_t)qglob * D + 8 * g;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 32 * s);
    }
    float lv[L];
    int li[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        lv[j] = -INFINITY;
        li[j] = -1;
    }
    const float thr0 = thr ? thr[qglob] : -INFINITY;
    float lmin = thr0;
    int nocc = 0;                                   // valid entries of this lane's list

    // make sure the query loads are consumed before any LDS-DMA is counted on vmcnt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // LDS-DMA in its BUFFER form (round 4).  As `__builtin_amdgcn_global_load_lds` (FLAT encoding) the requests inside the stage
    // loop made the compiler's wait-count pass treat every LDS wait of the loop as possibly out of order: all 31 waits between the
    // fragment reads and their MFMAs were `s_waitcnt lgkmcnt(0)` -- each MFMA waited for every read in flight, also those of the
    // MFMAs behind it.  With buffer instructions the waits are counted.  The descriptor's base is this workgroup's first stage
    // (uniform), offsets are SIGNED 32-bit and the descriptor covers 2 GiB: launch_scan refuses a launch whose per-workgroup stage
    // span reaches that (a 768-wide index above ~40 M rows scanned by ~32 workgroups per query block).
    // (KEYB_ / STAGEB_ as LOCAL constants: with the class template's static member `C::KEYB` written inside the builtin's argument
    // list hipcc 7.2 drops this kernel's host-side stub without a diagnostic -- every instantiation links as an undefined symbol.)
    constexpr int KEYB_ = C::KEYB, STAGEB_ = C::STAGEB, LDS_STAGE_ = C::LDS_STAGE;
    auto krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(packed + (size_t)s0 * STAGEB_)), 0, 0x7FFFFFFF,
                                                 0x00020000);
    const int voff = lane * 16;
    auto issue = [&](int stage, int slot) {
        const int so = (stage - s0) * STAGEB_;
        char* dst = smem + slot * LDS_STAGE_;
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
            const int piece = wave + 8 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, voff,
                                                     so + piece * 1024, 0, NT ? 2 : 0);
        }
        if (lane < 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(dst + KEYB_ + wave * 128), 16, voff,
                                                     so + KEYB_, 0, NT ? 2 : 0);
        }
    };

    // prologue: NST-1 stages in flight
#pragma unroll
    for (int i = 0; i < C::NST - 1; ++i)
        if (s0 + i < s1) issue(s0 + i, i);

    // per-lane LDS read offset of chunk (4*s + g) of row c (row 16+c has the same swizzle)
    const int row_off = c * C::ROWB;

    int slot = 0;
    for (int t = s0; t < s1; ++t) {
        int ahead = s1 - 1 - t;
        if (ahead > C::NST - 2) ahead = C::NST - 2;
        wait_stage_and_barrier<C::PW, C::NST - 2>(ahead);   // stage t landed for every wave
        {
            const int nt = t + C::NST - 1;                   // refill the slot read in iteration t-1
            int nslot = slot + C::NST - 1;
            if (nslot >= C::NST) nslot -= C::NST;
            if (nt < s1) issue(nt, nslot);
        }
        const char* sb = smem + slot * C::LDS_STAGE;
        f32x4 acc0 = *reinterpret_cast<const f32x4*>(sb + C::KEYB + wave * 128 + g * 16);
        f32x4 acc1 = *reinterpret_cast<const f32x4*>(sb + C::KEYB + wave * 128 + 64 + g * 16);
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int ch = swz_chunk(4 * s + g, c);
            bf16x8 a0 = qf[s], a1 = qf[s];
            if constexpr (DBG != 3) {
                a0 = *reinterpret_cast<const bf16x8*>(sb + row_off + ch * 16);
                a1 = *reinterpret_cast<const bf16x8*>(sb + row_off + 16 * C::ROWB + ch * 16);
            }
            if constexpr (DBG != 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[s], acc1, 0, 0, 0);
            } else {
                asm volatile("" ::"v"(a0), "v"(a1));
            }
        }
        // D layout: lane (g,c) holds query c, keys 4g..4g+3 of each 16-key tile
        const int kbase = t * STAGE_KEYS + g * 4;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = tt == 0 ? acc0[r] : acc1[r];
                if constexpr (DBG == 1 || DBG == 2) asm volatile("" ::"v"(sc));
                if ((DBG == 0 || DBG == 3) && sc > lmin) {
                    // branch-free sorted insert with no serial chain: every slot looks only at the OLD list.
                    // new v[j] = med3(v[j-1], v[j], x); id follows the same three cases (x > v[j] strict, so an
                    // equal score stays behind the earlier key).
                    // With insert thresholds a list holds 0-2 entries, but at a 0.2 % accept rate one compare in eight still
                    // brings its whole wave here: when every inserting lane holds fewer than SHORT entries, the slots
                    // from SHORT on are empty before and after, and only the first SHORT are touched (wave-uniform choice).
                    const int ci = kbase + tt * 16 + r;
                    float pv = INFINITY;
                    int pi = -1;
                    bool pgt = false;
                    constexpr int SHORT = L > 4 ? 4 : L;
                    if (L > 4 && __builtin_amdgcn_ballot_w64(nocc >= SHORT) == 0ull) {      // -10 us per 0.5 M-row search (same-box A/B)
#pragma unroll
                        for (int j = 0; j < SHORT; ++j) {
                            const float ov = lv[j];
                            const int oi = li[j];
                            const bool gt = sc > ov;
                            lv[j] = __builtin_amdgcn_fmed3f(pv, ov, sc);
                            li[j] = gt ? (pgt ? pi : ci) : oi;
                            pv = ov;
                            pi = oi;
                            pgt = gt;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < L; ++j) {
                            const float ov = lv[j];
                            const int oi = li[j];
                            const bool gt = sc > ov;
                            lv[j] = __builtin_amdgcn_fmed3f(pv, ov, sc);
                            li[j] = gt ? (pgt ? pi : ci) : oi;
                            pv = ov;
                            pi = oi;
                            pgt = gt;
                        }
                    }
                    nocc = nocc < L ? nocc + 1 : L;
                    lmin = fmaxf(thr0, lv[L - 1]);
                }
            }
        }
        slot = slot + 1 == C::NST ? 0 : slot + 1;
    }

    // valid entries out (a list is sorted, its valid entries come first), packed over the four lanes of the query
    int n = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) n += li[j] >= 0 ? 1 : 0;
    const int n0 = __shfl(n, c, 64), n1 = __shfl(n, c + 16, 64), n2 = __shfl(n, c + 32, 64), n3 = __shfl(n, c + 48, 64);
    const int base = g == 0 ? 0 : (g == 1 ? n0 : (g == 2 ? n0 + n1 : n0 + n1 + n2));
    unsigned lk = n == L ? ord_key(lv[L - 1]) : 0u;
    {
        auto a = __builtin_amdgcn_permlane16_swap(lk, lk, false, false);
        lk = a[0] > a[1] ? a[0] : a[1];
        auto b = __builtin_amdgcn_permlane32_swap(lk, lk, false, false);
        lk = b[0] > b[1] ? b[0] : b[1];
    }
    const size_t seg = (size_t)qglob * nwg + blockIdx.x;
    unsigned long long* op = out_pairs + seg * (4 * L) + base;
#pragma unroll
    for (int j = 0; j < L; ++j)
        if (j < n) op[j] = ((unsigned long long)ord_key(lv[j]) << 32) | (unsigned)li[j];
    if (g == 0) out_meta[seg] = uint2{(unsigned)(n0 + n1 + n2 + n3), lk};
}

// NOTE (measured, round 2): what the candidate pass is NOT bound by.  At D = 768 it runs 5.4 k cycles per 32-key stage per
// CU (4.6 TB/s) where the matrix pipe needs 1.5 k and HBM 3 k.  Timing-only ablations: without list updates 137 us, without
// the MFMAs as well 135 us, without the LDS fragment reads no change (168 us): the bare LDS-DMA ring reaches 5.7 TB/s and the
// compute adds ~30 us on top of it.  (a) Bytes in flight: the same image streamed in 16-key half stages through six 24.5 KB
// slots (five halves = 122 KB in flight instead of two stages = 96 KB, one score tile per wave per barrier) is SLOWER:
// whole search 247 vs 227 us, the 8-shard configuration 360 vs 330 us.  (b) Which stages a workgroup reads: dealing them
// round-robin (the CUs read neighbouring blobs at any one time, like a grid-stride copy) instead of one contiguous range per
// workgroup changes nothing (226.5 vs 228.7 us).  Both removed.

// ------------------------------------------------------------------------------------------
// pack: fp32 rows -> swizzled bf16 stage blobs + fp32 bias tail.  One 256-thread block per stage.
template <int D>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ db, long long n, int metric,
                                                   char* __restrict__ packed, DbBounds* __restrict__ bounds,
                                                   long long first_stage) {
    using C = ScanCfg<D>;
    const long long stage = first_stage + blockIdx.x;
    char* blob = packed + (size_t)stage * C::STAGEB;
    for (int id = threadIdx.x; id < STAGE_KEYS * C::CHUNKS; id += 256) {
        const int row = id / C::CHUNKS, ch = id % C::CHUNKS;
        const long long key = stage * STAGE_KEYS + row;
        bf16x8 v;
        if (key < n) {
            const float* p = db + (size_t)key * D + ch * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
            v = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                       (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
        } else {
            v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        *reinterpret_cast<bf16x8*>(blob + row * C::ROWB + swz_chunk(ch, row) * 16) = v;
    }
    // bias tail + the rounding bounds of the certificate: one wave per 8 rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m_xt = 0.f, m_r = 0.f, m_x = 0.f;
    for (int row = wave * 8; row < wave * 8 + 8; ++row) {
        const long long key = stage * STAGE_KEYS + row;
        float bias;
        if (key < n) {
            float s = 0.f, st = 0.f, sr = 0.f;
            const float* p = db + (size_t)key * D;
            for (int i = lane; i < D; i += 64) {
                const float v = p[i], vt = (float)(bf16_t)v;
                s += v * v;
                st += vt * vt;
                sr += (v - vt) * (v - vt);
            }
            s = wave_sum(s);
            st = wave_sum(st);
            sr = wave_sum(sr);
            bias = metric == KEDS_METRIC_L2 ? -0.5f * s : 0.f;
            m_xt = fmaxf(m_xt, st);
            m_r = fmaxf(m_r, sr);
            m_x = fmaxf(m_x, s);
        } else {
            bias = -INFINITY;
        }
        if (lane == 0) reinterpret_cast<float*>(blob + C::KEYB)[row] = bias;
    }
    if (lane == 0) {      // non-negative floats order like their bit patterns; sqrt rounded up by one ulp-ish factor
        int* b = reinterpret_cast<int*>(bounds);
        atomicMax(b + 0, __float_as_int(sqrtf(m_xt) * 1.000001f));
        atomicMax(b + 1, __float_as_int(sqrtf(m_r) * 1.000001f));
        atomicMax(b + 2, __float_as_int(sqrtf(m_x) * 1.000001f));
    }
}

// ------------------------------------------------------------------------------------------
// qprep: one wave per query row (rows >= nq of the bf16 block are zeroed)
// qstat[row] = {||q||^2, ||q - bf16(q)||, ||bf16(q)||, 0} of the query as searched (after the optional normalisation);
// thread 0 also clears the per-launch-set counters {failed certificates, next fallback slot}.
__global__ __launch_bounds__(256) void qprep_kernel(const float* __restrict__ q, int nq, int dim, int normalize,
                                                    float* __restrict__ qn, bf16_t* __restrict__ qb, int qb_rows,
                                                    f32x4* __restrict__ qstat, int* __restrict__ counters) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[0] = 0;
    if (row >= qb_rows) return;
    constexpr int MAXV = 4;                       // dim <= 1024: four float4 per lane, all loads in flight together
    const int nv = dim >> 8;                      // 256 floats per wave pass (dim % 128 == 0: a last half pass is predicated)
    const bool half = (dim & 255) != 0;
    bf16_t* qbr = qb + (size_t)row * dim;
    if (row >= nq) {
        for (int i = lane * 4; i < dim; i += 256) *reinterpret_cast<bf16x4*>(qbr + i) = bf16x4{0, 0, 0, 0};
        return;
    }
    const float* p = q + (size_t)row * dim;
    f32x4 v[MAXV + 1];
#pragma unroll
    for (int j = 0; j <= MAXV; ++j) {
        const int i = lane * 4 + 256 * j;
        const bool on = j < nv || (j == nv && half && i < dim);
        v[j] = on ? *reinterpret_cast<const f32x4*>(p + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j <= MAXV; ++j) s += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
    s = wave_sum(s);
    const float nrm = sqrtf(s);
    float s2 = 0.f, st = 0.f, sr = 0.f;
#pragma unroll
    for (int j = 0; j <= MAXV; ++j) {
        const int i = lane * 4 + 256 * j;
        const bool on = j < nv || (j == nv && half && i < dim);
        if (!on) continue;
        f32x4 x = v[j];
        if (normalize) x = f32x4{x[0] / nrm, x[1] / nrm, x[2] / nrm, x[3] / nrm};
        const bf16x4 xb = bf16x4{(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3]};
        *reinterpret_cast<f32x4*>(qn + (size_t)row * dim + i) = x;
        *reinterpret_cast<bf16x4*>(qbr + i) = xb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float vt = (float)xb[e];
            s2 += x[e] * x[e];
            st += vt * vt;
            sr += (x[e] - vt) * (x[e] - vt);
        }
    }
    s2 = wave_sum(s2);
    st = wave_sum(st);
    sr = wave_sum(sr);
    if (lane == 0) qstat[row] = f32x4{s2, sqrtf(sr) * 1.000001f, sqrtf(st) * 1.000001f, 0.f};
}

// ------------------------------------------------------------------------------------------
// merge: exact top-ncand of the union of a query's scan lists, by (score, smaller id at the cut), + the score bound of
// everything left out of it (certificate).  One 256-thread block per query; thread t owns the segment of scan workgroup t
// (meta = {valid entries, last-slot key}, pairs packed at the front).
//   V <= 4096 valid entries (the normal case: with insert thresholds ~1 entry per list): the pairs are packed into LDS,
//     ONE wave takes all keys into registers and finds the cut T = the ncand-th largest key by bisection on the bits that
//     differ, counting with wave-wide DPP reductions (no barrier per bit), and the block collects from LDS;
//   more (dense lists: single-phase scans of small databases): every thread keeps its segment's <= 64 pairs in registers
//     and the block bisects together (two barriers per bit).
// thr_out[q] = T (or -inf if fewer than ncand valid entries): the insert threshold of the candidate pass.
// sbound_out[q]: upper bound on the bf16 score of every row that is NOT among the candidates.  Such a row was (a) in the
//   union and cut: score <= T; (b) rejected by the insert threshold: score <= thr_in; (c) pushed out of a full per-lane
//   list: score <= that list's last entry (meta.y).  (a) only exists when V > ncand, and then T > thr_in.
template <class Op>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned x, Op op) {       // DPP (lane bits 0-3) + permlane swaps (4, 5)
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true));    // row_half_mirror
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true));    // row_mirror
    auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    x = op(a[0], a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return op(b[0], b[1]);
}

__device__ unsigned long long* g_merge_stamp = nullptr;     // diagnostic (keds_merge_stamp_buffer): 8 stamps per block
#define KEDS_MSTAMP(i)                                                                   \
    if (mst && threadIdx.x == 0) mst[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();

constexpr int MERGE_THREADS = 256;
constexpr int MERGE_WAVES = MERGE_THREADS / 64;
constexpr int MERGE_SLOTS = 4 * LISTK;    // pairs per segment at most (64)
constexpr int MERGE_LDS_PAIRS = 4096;
constexpr int MERGE_SPEC = 16;            // segment slots loaded before the segment's count is known

// FUSED (round 4: the final merge of a search): the block that has merged a query's lists also re-ranks its `ncand` candidates
// with the exact fp32 distance (its four waves take the candidates in turn, four rows in flight each) and runs the selection
// + certificate on them -- what rerank_kernel and certify_select_kernel did as two more launches (6.4 + 10 us, each a grid
// of short dependent round trips, + two kernel boundaries): one launch less to wait for, and the candidate ids / scores never
// leave the CU.  Same arithmetic in the same order as the two kernels.  Measured neutral (see search_tail_fused): an A/B form.
struct TailArgs {
    const float* db;
    const float* qn;
    const f32x4* qstat;
    const DbBounds* bounds;
    float* D;
    long long* I;
    int* counters;
    int* fail_ids;
    int* fslot;
    float* dk;
    int* status;
    long long id_base;
    int dim, metric, k, force_fail;
};

template <bool FUSED>
__global__ __launch_bounds__(MERGE_THREADS) void merge_pairs_kernel(const unsigned long long* __restrict__ pairs,
                                                                    const uint2* __restrict__ meta, int nwg, int slots,
                                                                    int* __restrict__ cand_idx, float* __restrict__ cand_val,
                                                                    float* __restrict__ thr_out, int ncand,
                                                                    const float* __restrict__ thr_in,
                                                                    float* __restrict__ sbound_out, TailArgs ta) {
    __shared__ unsigned long long lp[MERGE_LDS_PAIRS];
    [[maybe_unused]] __shared__ int s_cid[FUSED ? NCAND_WIDE : 1];
    [[maybe_unused]] __shared__ float s_cd[FUSED ? NCAND_WIDE : 1];
    [[maybe_unused]] __shared__ float s_sb;
    [[maybe_unused]] __shared__ int s_nvalid;
    [[maybe_unused]] __shared__ float s_dk;
    __shared__ unsigned s_red[4][MERGE_WAVES];
    __shared__ unsigned s_T, s_rem, s_out, s_eq;
    __shared__ __attribute__((aligned(16))) unsigned hist[256];
    __shared__ int eq_idx[256];
    __shared__ float eq_val[256];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long* mst = g_merge_stamp;
    KEDS_MSTAMP(0)
    auto umax = [](unsigned a, unsigned b) { return a > b ? a : b; };
    auto umin = [](unsigned a, unsigned b) { return a < b ? a : b; };
    auto uadd = [](unsigned a, unsigned b) { return a + b; };
    // A dependent global round trip costs ~5 us in these small kernels: the first MERGE_SPEC slots of the segment are
    // loaded SPECULATIVELY together with the segment's count (a segment rarely holds more with insert thresholds, and the
    // threshold pass has exactly 16), so the usual merge waits for memory once.
    const unsigned long long* seg = pairs + ((size_t)q * nwg + (tid < nwg ? tid : 0)) * slots;
    unsigned long long pr[MERGE_SLOTS];
#pragma unroll
    for (int j = 0; j < MERGE_SLOTS; ++j) pr[j] = 0ull;
#pragma unroll
    for (int j = 0; j < MERGE_SPEC; ++j)
        if (j < slots) pr[j] = seg[j];                                                  // slots is kernel-uniform
    const uint2 m = tid < nwg ? meta[(size_t)q * nwg + tid] : uint2{0u, 0u};
    const unsigned cnt = m.x;
    const unsigned wmax = wave_reduce_u32(cnt, umax);
#pragma unroll
    for (int j = MERGE_SPEC; j < MERGE_SLOTS; ++j)
        if ((unsigned)j < wmax) pr[j] = (unsigned)j < cnt ? seg[j] : 0ull;             // wave-uniform skip of the tail
#pragma unroll
    for (int j = 0; j < MERGE_SPEC; ++j)
        if ((unsigned)j >= cnt) pr[j] = 0ull;                                           // stale bytes of earlier searches
    unsigned kmax = 0, kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < MERGE_SLOTS; ++j)
        if ((unsigned)j < cnt) {
            const unsigned k = (unsigned)(pr[j] >> 32);
            kmax = k > kmax ? k : kmax;
            kmin = k < kmin ? k : kmin;
        }
    // block totals: V, key range, last-slot key; and this thread's offset into the packed LDS list
    unsigned incl = cnt;                                           // inclusive prefix over the lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const unsigned wsum = __shfl(incl, 63, 64);
    kmax = wave_reduce_u32(kmax, umax);
    kmin = wave_reduce_u32(kmin, umin);
    unsigned lastkey = wave_reduce_u32(m.y, umax);
    if (tid == 0) {
        s_out = 0;
        s_eq = 0;
    }
    if (lane == 0) {
        s_red[0][wv] = wsum;
        s_red[1][wv] = kmax;
        s_red[2][wv] = kmin;
        s_red[3][wv] = lastkey;
    }
    KEDS_MSTAMP(1)
    __syncthreads();
    KEDS_MSTAMP(2)
    unsigned V = 0, wbase = 0;
#pragma unroll
    for (int w = 0; w < MERGE_WAVES; ++w) {
        if (w < wv) wbase += s_red[0][w];
        V += s_red[0][w];
        kmax = umax(kmax, s_red[1][w]);
        kmin = umin(kmin, s_red[2][w]);
        lastkey = umax(lastkey, s_red[3][w]);
    }
    const unsigned want = V < (unsigned)ncand ? V : (unsigned)ncand;
    unsigned T = 0, rem = want;
    const unsigned kx = kmax ^ kmin;
    const int top = (V > 0 && kx) ? 31 - __builtin_clz(kx) : -1;                  // highest bit in which two valid keys differ
    unsigned prefix = top < 0 ? kmax : (top >= 31 ? 0u : (kmax & ~((2u << top) - 1u)));
    const bool small = V <= (unsigned)MERGE_LDS_PAIRS;                            // block-uniform
    // collect: key > T always; key == T into the tie buffer (only when a cut exists)
    auto take = [&](unsigned long long p2, unsigned Tc) {
        const unsigned k = (unsigned)(p2 >> 32);
        const int id = (int)(unsigned)(p2 & 0xFFFFFFFFu);
        if (V <= (unsigned)ncand || k > Tc) {
            const unsigned o = atomicAdd(&s_out, 1u);
            if constexpr (FUSED) s_cid[o] = id;
            else {
                cand_idx[q * ncand + o] = id;
                cand_val[q * ncand + o] = key_float(k);
            }
        } else if (k == Tc) {
            const unsigned o = atomicAdd(&s_eq, 1u);
            if (o < 256) {
                eq_idx[o] = id;
                eq_val[o] = key_float(k);
            }
        }
    };
    if (small) {
        const unsigned off = wbase + incl - cnt;
#pragma unroll
        for (int j = 0; j < MERGE_SLOTS; ++j)
            if ((unsigned)j < cnt) lp[off + j] = pr[j];
        __syncthreads();
        KEDS_MSTAMP(3)
        if (V > (unsigned)ncand) {
            // T = the want-th largest key by MSB-first radix selection, 8 bits per pass over the bits in which keys differ
            // (scores of one query share their high bits: usually three passes): LDS histogram of the digit among the keys
            // that match the prefix so far, then one wave walks the 256 bins from the top.  (A one-wave bisection over
            // registers, one bit per step, took 57 k of this kernel's 76 k cycles.)
            unsigned rem_l = want;                                     // how many keys are still to be taken at / below the prefix
            unsigned pfx = 0;                                          // digits decided so far (the bits above `shift + 8`)
            int shift = top >= 0 ? (top & ~7) : -8;                    // first digit that can differ (all keys equal: none)
            if (top >= 0) pfx = shift >= 24 ? 0u : (kmax >> (shift + 8));
            else pfx = kmax;
            for (; shift >= 0; shift -= 8) {
                hist[tid] = 0;
                __syncthreads();
                for (unsigned p2 = tid; p2 < V; p2 += MERGE_THREADS) {
                    const unsigned k = (unsigned)(lp[p2] >> 32);
                    const bool in = shift >= 24 ? true : (k >> (shift + 8)) == pfx;
                    if (in) atomicAdd(&hist[(k >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (wv == 0) {
                    const u32x4 h4 = *reinterpret_cast<const u32x4*>(&hist[4 * lane]);
                    const unsigned tl = h4[0] + h4[1] + h4[2] + h4[3];
                    unsigned suf = tl;                                 // inclusive suffix sum over the lanes (bins 4 lane ..)
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned dn = __shfl_down(suf, o, 64);
                        if (lane + o < 64) suf += dn;
                    }
                    unsigned above = suf - tl;                         // keys in bins above this lane's four
                    // the bin b with above(b) < rem <= above(b) + h[b]: exactly one (lane, bin)
#pragma unroll
                    for (int e = 3; e >= 0; --e) {
                        if (above < rem_l && rem_l <= above + h4[e]) {
                            s_T = (unsigned)(4 * lane + e);
                            s_rem = rem_l - above;
                        }
                        above += h4[e];
                    }
                }
                __syncthreads();
                pfx = (pfx << 8) | s_T;
                rem_l = s_rem;
            }
            T = pfx;                                                   // all 32 bits decided: the want-th largest key
            rem = rem_l;                                               // of the keys == T, this many are taken
        }
        KEDS_MSTAMP(4)
        for (unsigned p2 = tid; p2 < V; p2 += MERGE_THREADS) take(lp[p2], T);
        KEDS_MSTAMP(5)
    } else {
        // dense lists: block-wide bisection over the registers (two barriers per bit)
        auto block_sum = [&](unsigned x) -> unsigned {
            x = wave_reduce_u32(x, uadd);
            __syncthreads();
            if (lane == 0) s_red[0][wv] = x;
            __syncthreads();
            return s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        };
#pragma unroll 1
        for (int bit = top; bit >= 0; --bit) {
            const unsigned cand = prefix | (1u << bit);
            unsigned c2 = 0;
#pragma unroll
            for (int j = 0; j < MERGE_SLOTS; ++j) c2 += ((unsigned)j < cnt && (unsigned)(pr[j] >> 32) >= cand) ? 1u : 0u;
            if (block_sum(c2) >= want) prefix = cand;
        }
        T = prefix;
        unsigned gt = 0;
#pragma unroll
        for (int j = 0; j < MERGE_SLOTS; ++j) gt += ((unsigned)j < cnt && (unsigned)(pr[j] >> 32) > T) ? 1u : 0u;
        rem = want - block_sum(gt);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MERGE_SLOTS; ++j)
            if ((unsigned)j < cnt) take(pr[j], T);
    }
    __syncthreads();
    if (V > (unsigned)ncand) {
        // ties at the threshold: take the `rem` smallest ids (rank by counting; ties are rare, <= 256 handled)
        const unsigned neq = s_eq < 256 ? s_eq : 256;
        const unsigned base = s_out;
        if (tid < (int)neq) {
            const int my = eq_idx[tid];
            unsigned rank = 0;
            for (unsigned j = 0; j < neq; ++j) rank += eq_idx[j] < my ? 1u : 0u;
            if (rank < rem) {
                if constexpr (FUSED) s_cid[base + rank] = my;
                else {
                    cand_idx[q * ncand + base + rank] = my;
                    cand_val[q * ncand + base + rank] = eq_val[tid];
                }
            }
        }
    }
    for (int o = (int)want + tid; o < ncand; o += MERGE_THREADS) {       // fillers when fewer than ncand valid entries
        if constexpr (FUSED) s_cid[o] = -1;
        else {
            cand_idx[q * ncand + o] = -1;
            cand_val[q * ncand + o] = -INFINITY;
        }
    }
    KEDS_MSTAMP(6)
    if (mst && tid == 0) mst[(size_t)blockIdx.x * 8 + 7] = V;
    if (tid == 0) {
        if (thr_out) thr_out[q] = V > (unsigned)ncand ? key_float(T) : -INFINITY;
        if (sbound_out || FUSED) {
            float b2 = V > (unsigned)ncand ? key_float(T) : (thr_in ? thr_in[q] : -INFINITY);
            if (lastkey) b2 = fmaxf(b2, key_float(lastkey));
            if (sbound_out) sbound_out[q] = b2;
            if constexpr (FUSED) s_sb = b2;
        }
        if constexpr (FUSED) {
            s_nvalid = 0;
            s_dk = ta.metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
        }
    }
    if constexpr (FUSED) {
        __syncthreads();
        // ---- rerank_kernel's arithmetic: exact fp32 distance of every candidate, one wave each, four rows requested at a time
        const int dim = ta.dim, metric = ta.metric;
        const float* qq = ta.qn + (size_t)q * dim;
        for (int c0 = 4 * wv; c0 < ncand; c0 += 4 * MERGE_WAVES) {
            float acc[4];
            int ids[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ids[u] = c0 + u < ncand ? s_cid[c0 + u] : -1;
                acc[u] = 0.f;
            }
            for (int i = lane * 4; i < dim; i += 256) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(qq + i);
                f32x4 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    b[u] = ids[u] >= 0 ? *reinterpret_cast<const f32x4*>(ta.db + (size_t)ids[u] * dim + i) : a;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (metric == KEDS_METRIC_L2) {
                        const f32x4 d = a - b[u];
                        acc[u] += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
                    } else {
                        acc[u] += a[0] * b[u][0] + a[1] * b[u][1] + a[2] * b[u][2] + a[3] * b[u][3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sum = wave_sum(acc[u]);
                if (lane == 0 && c0 + u < ncand)
                    s_cd[c0 + u] = ids[u] >= 0 ? sum : (metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY);
            }
        }
        __syncthreads();
        // ---- certify_select_kernel's body on the block's own candidates (threads 0 .. ncand-1)
        const int k = ta.k;
        const int t = tid;
        const bool act = t < ncand;
        const float d = act ? s_cd[t] : 0.f;
        const int id = act ? s_cid[t] : -1;
        const f32x4 qs = ta.qstat[q];
        const float xt = ta.bounds->xt, rm = ta.bounds->r;
        const float key = metric == KEDS_METRIC_L2 ? d : -d;           // smaller is better
        int rank = 0;
        if (act) {
            for (int j = 0; j < ncand; ++j) {
                const float dj = s_cd[j];
                const float kj = metric == KEDS_METRIC_L2 ? dj : -dj;
                const int ij = s_cid[j];
                const bool before = ij >= 0 && (id < 0 || kj < key || (kj == key && ij < id));
                rank += (before && j != t) ? 1 : 0;
            }
            if (id >= 0) atomicAdd(&s_nvalid, 1);
            if (id < 0) rank = ncand + t;                              // never selected before a valid one
            if (rank < k) {
                ta.D[(size_t)q * k + rank] = d;
                ta.I[(size_t)q * k + rank] = id + ta.id_base;
            }
            if (id >= 0 && rank == k - 1) s_dk = d;
        }
        __syncthreads();
        const int nvalid = s_nvalid;
        if (t >= nvalid && t < k) {                                    // fewer than k valid candidates: fillers like faiss
            ta.D[(size_t)q * k + t] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
            ta.I[(size_t)q * k + t] = -1;
        }
        if (t == 0) {
            const float sb = s_sb;
            bool ok = sb == -INFINITY;                                 // nothing was ever left out of the candidate set
            if (ta.force_fail) ok = false;
            else if (!ok && nvalid >= k) {
                const float tk = metric == KEDS_METRIC_L2 ? 0.5f * (qs[0] - s_dk) : s_dk;
                const float eps = (qs[1] * xt + qs[2] * rm + qs[1] * rm + 2e-4f * qs[2] * xt) * 1.01f +
                                  1e-6f * (fabsf(tk) + qs[0] + 1.0f);
                ok = tk - eps > sb;
            }
            if (ok) {
                ta.fslot[q] = -1;
                if (ta.status) atomicAdd(ta.status + 0, 1);
            } else {
                const int f = atomicAdd(ta.counters, 1);
                ta.fail_ids[f] = q;
                ta.fslot[q] = f;
                ta.counters[8 + f] = 0;                                // chunks of this query the exact pass has finished
                ta.dk[q] = nvalid >= k ? s_dk : (metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY);
                if (ta.status) atomicAdd(ta.status + 1, 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// rerank: exact fp32 distance (L2: sum (q-x)^2; IP: q.x) of every candidate; one wave each.
__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ db, int dim, int metric,
                                                     const float* __restrict__ qn, const int* __restrict__ cand_idx,
                                                     float* __restrict__ cand_d, int total, int ncand) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= total) return;
    const int q = w / ncand;
    const int id = cand_idx[w];
    if (id < 0) {
        if (lane == 0) cand_d[w] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
        return;
    }
    const float* x = db + (size_t)id * dim;
    const float* qq = qn + (size_t)q * dim;
    float s = 0.f;
    if (metric == KEDS_METRIC_L2) {
        for (int i = lane * 4; i < dim; i += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qq + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x + i);
            const f32x4 d = a - b;
            s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    } else {
        for (int i = lane * 4; i < dim; i += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qq + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x + i);
            s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
    }
    s = wave_sum(s);
    if (lane == 0) cand_d[w] = s;
}

// select + certificate: one block of `ncand` threads (64 or 256) per query.  Ranks the candidates by (exact distance, id),
// writes the top-k, and then PROVES that no row outside the candidate set can belong to the exact top-k:
//   every such row has a bf16 scan score  s <= sbound[q]                         (merge_select_kernel)
//   its exact score t = q.x - 0.5||x||^2 (L2) or q.x (IP) obeys  |t - s| <= eps(q)
//       eps(q) = ||q - q~|| max||x~|| + ||q~|| max||x - x~|| + ||q - q~|| max||x - x~||  (Cauchy-Schwarz on
//       q.x - q~.x~ = (q - q~).x~ + q~.(x - x~) + (q - q~).(x - x~); q~, x~ = the bf16 roundings the scan multiplies)
//       + 2e-4 ||q~|| max||x~||  (fp32 accumulation of the MFMA chain: 768 adds x 2^-24 relative, 4x margin)
//   so if the k-th best candidate's exact score t_k satisfies  t_k - eps(q) > sbound[q],  every outside row is strictly
//   worse than k candidates: the result is the exact top-k, ties included.  Otherwise the query goes to the exact
//   fallback (exact_chunk_kernel, which also merges) with the pruning bound dk = the k-th candidate distance.
// counters[0] = failed certificates of this launch set; status[0] / [1] += certified / fallback queries (nullable).
__global__ __launch_bounds__(256) void certify_select_kernel(const int* __restrict__ cand_idx, const float* __restrict__ cand_d,
                                                             int ncand, int metric, int k, long long id_base,
                                                             float* __restrict__ D, long long* __restrict__ I,
                                                             const f32x4* __restrict__ qstat, const float* __restrict__ sbound,
                                                             const DbBounds* __restrict__ bounds, int* __restrict__ counters,
                                                             int* __restrict__ fail_ids, int* __restrict__ fslot,
                                                             float* __restrict__ dk, int* __restrict__ status,
                                                             int force_fail) {
    __shared__ float s_key[NCAND_WIDE];
    __shared__ int s_id[NCAND_WIDE];
    __shared__ int s_nvalid;
    __shared__ float s_dk;
    const int q = blockIdx.x, t = threadIdx.x;
    const float d = cand_d[q * ncand + t];
    const int id = cand_idx[q * ncand + t];
    // (loaded up front by every thread: the certificate at the end then waits for no further memory round trip)
    const f32x4 qs = qstat[q];
    const float sb = sbound[q];
    const float xt = bounds->xt, rm = bounds->r;
    const float key = metric == KEDS_METRIC_L2 ? d : -d;           // smaller is better
    s_key[t] = key;
    s_id[t] = id;
    if (t == 0) {
        s_nvalid = 0;
        s_dk = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
    }
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < ncand; ++j) {
        const float kj = s_key[j];
        const int ij = s_id[j];
        const bool before = ij >= 0 && (id < 0 || kj < key || (kj == key && ij < id));
        rank += (before && j != t) ? 1 : 0;
    }
    if (id >= 0) atomicAdd(&s_nvalid, 1);
    if (id < 0) rank = ncand + t;            // never selected before a valid one
    if (rank < k) {
        D[(size_t)q * k + rank] = d;
        I[(size_t)q * k + rank] = id + id_base;
    }
    if (id >= 0 && rank == k - 1) s_dk = d;
    __syncthreads();
    const int nvalid = s_nvalid;
    // fewer than k valid candidates: (inf | -inf, -1) fillers like faiss
    if (t >= nvalid && t < k) {
        D[(size_t)q * k + t] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
        I[(size_t)q * k + t] = -1;
    }
    if (t == 0) {
        bool ok = sb == -INFINITY;           // nothing was ever left out of the candidate set
        if (force_fail) ok = false;          // test hook (keds_scan_debug bit 5): every query takes the exact pass
        else if (!ok && nvalid >= k) {
            const float tk = metric == KEDS_METRIC_L2 ? 0.5f * (qs[0] - s_dk) : s_dk;
            const float eps = (qs[1] * xt + qs[2] * rm + qs[1] * rm + 2e-4f * qs[2] * xt) * 1.01f +
                              1e-6f * (fabsf(tk) + qs[0] + 1.0f);
            ok = tk - eps > sb;
        }
        if (ok) {
            fslot[q] = -1;
            if (status) atomicAdd(status + 0, 1);
        } else {
            const int f = atomicAdd(counters, 1);
            fail_ids[f] = q;
            fslot[q] = f;
            counters[8 + f] = 0;             // chunks of this query the exact pass has finished (its last one merges)
            dk[q] = nvalid >= k ? s_dk : (metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY);
            if (status) atomicAdd(status + 1, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Exact fallback, pass 1.  Work item (chunk c of `chunk_rows` rows, failed query f): exact fp32 distances of the chunk's
// rows from the fp32 matrix, keep the rows that are not worse than the pruning bound dk[q], rank them by (distance, id)
// and write the best min(count, k) in order to plist[f][c][0..k).  Persistent blocks walk the items with the failed
// queries fastest, so the blocks that run together stream the same rows (L2).  Exits at once when nothing failed.
__device__ __noinline__ void exact_merge_query(int q, int f, int nchunks, int k, int metric, long long id_base,
                                               const unsigned long long* __restrict__ plist, const int* __restrict__ pcnt,
                                               float* __restrict__ D, long long* __restrict__ I);
__device__ __forceinline__ unsigned long long exact_key(float d, int id, int metric) {
    const float kf = metric == KEDS_METRIC_L2 ? d : -d;            // smaller is better
    return ((unsigned long long)ord_key(kf) << 32) | (unsigned)id;
}

__global__ __launch_bounds__(256) void exact_chunk_kernel(const float* __restrict__ db, long long n, int dim, int metric,
                                                          const float* __restrict__ qn, const int* __restrict__ counters,
                                                          const int* __restrict__ fail_ids, const float* __restrict__ dk,
                                                          int chunk_rows, int nchunks, int k,
                                                          unsigned long long* __restrict__ plist, int* __restrict__ pcnt,
                                                          int* __restrict__ done, long long id_base, float* __restrict__ Dq,
                                                          long long* __restrict__ Iq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nfail = counters[0];
    if (nfail == 0) return;
    float* sq = reinterpret_cast<float*>(smem);                                              // [dim]
    unsigned long long* surv = reinterpret_cast<unsigned long long*>(smem + dim * 4);         // [chunk_rows]
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long items = (long long)nfail * nchunks;
    for (long long w = blockIdx.x; w < items; w += gridDim.x) {
        const int c = (int)(w / nfail), f = (int)(w - (long long)c * nfail);
        const int q = fail_ids[f];
        __syncthreads();                                   // previous item's LDS is no longer read
        for (int i = tid; i < dim; i += 256) sq[i] = qn[(size_t)q * dim + i];
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const float bound = dk[q];
        const long long r0 = (long long)c * chunk_rows;
        const int rows = (int)((n - r0 < chunk_rows) ? (n - r0) : chunk_rows);
        for (int r = wave; r < rows; r += 4) {
            const float* x = db + (size_t)(r0 + r) * dim;
            float s = 0.f;
            if (metric == KEDS_METRIC_L2) {
                for (int i = lane * 4; i < dim; i += 256) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sq + i);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(x + i);
                    const f32x4 dd = a - b;
                    s += dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] + dd[3] * dd[3];
                }
            } else {
                for (int i = lane * 4; i < dim; i += 256) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sq + i);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(x + i);
                    s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
                }
            }
            s = wave_sum(s);                               // same summation order as rerank_kernel: identical bits
            const bool keep = metric == KEDS_METRIC_L2 ? s <= bound : s >= bound;
            if (lane == 0 && keep) surv[atomicAdd(&s_cnt, 1)] = exact_key(s, (int)(r0 + r), metric);
        }
        __syncthreads();
        const int cnt = s_cnt;
        unsigned long long* out = plist + ((size_t)f * nchunks + c) * k;
        for (int i = tid; i < cnt; i += 256) {             // rank by counting: keys are unique (the id is part of them)
            const unsigned long long me = surv[i];
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += surv[j] < me ? 1 : 0;
            if (rank < k) out[rank] = me;
        }
        if (tid == 0) pcnt[(size_t)f * nchunks + c] = cnt < k ? cnt : k;
        // the block that finishes the query's last chunk merges its lists (release / acquire through the counter)
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            s_cnt = atomicAdd(&done[f], 1) == nchunks - 1 ? -1 : 0;
        }
        __syncthreads();
        if (s_cnt == -1) {
            __threadfence();
            exact_merge_query(q, f, nchunks, k, metric, id_base, plist, pcnt, Dq, Iq);
        }
    }
}

// Exact fallback, pass 2: one block per query whose certificate failed merges its chunk lists (each sorted) into the
// final top-k: thread t owns the lists t, t + 256, ... and the block pops the smallest head k times.
// Exact fallback, merge: the k best of a failed query's chunk lists (each sorted), by the block that finished the query's LAST
// chunk (no second launch: a search that certifies every query -- the normal case -- pays one empty launch, not two).
__device__ __noinline__ void exact_merge_query(int q, int f, int nchunks, int k, int metric, long long id_base,
                                               const unsigned long long* __restrict__ plist, const int* __restrict__ pcnt,
                                               float* __restrict__ D, long long* __restrict__ I) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int OWN = 8;                                 // lists per thread: nchunks <= 2048
    __shared__ unsigned long long s_best[4];
    __shared__ int s_who[4];
    int pos[OWN], cnt[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int c = tid + 256 * o;
        pos[o] = 0;
        cnt[o] = c < nchunks ? pcnt[(size_t)f * nchunks + c] : 0;
    }
    const unsigned long long NONE = ~0ull;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = NONE;
        int bo = -1;
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            if (pos[o] < cnt[o]) {
                const unsigned long long h = plist[((size_t)f * nchunks + tid + 256 * o) * k + pos[o]];
                if (h < best) {
                    best = h;
                    bo = o;
                }
            }
        }
        unsigned long long wbest = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(wbest, o, 64);
            wbest = other < wbest ? other : wbest;
        }
        __syncthreads();
        if (best == wbest && best != NONE) {               // keys are unique: at most one lane per wave matches
            s_best[wave] = wbest;
            s_who[wave] = tid;
        }
        if (lane == 0 && wbest == NONE) s_best[wave] = NONE;
        __syncthreads();
        unsigned long long bb = s_best[0];
        int who = s_who[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_best[w] < bb) {
                bb = s_best[w];
                who = s_who[w];
            }
        if (bb == NONE) {
            if (tid == 0) {
                D[(size_t)q * k + r] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
                I[(size_t)q * k + r] = -1;
            }
            continue;
        }
        if (tid == who) {
#pragma unroll
            for (int o = 0; o < OWN; ++o)
                if (o == bo) pos[o] += 1;
            const unsigned ku = (unsigned)(bb >> 32);
            const float kf = __uint_as_float((ku & 0x80000000u) ? (ku & 0x7FFFFFFFu) : ~ku);
            D[(size_t)q * k + r] = metric == KEDS_METRIC_L2 ? kf : -kf;
            I[(size_t)q * k + r] = (long long)(unsigned)(bb & 0xFFFFFFFFu) + id_base;
        }
    }
}

__global__ void fill_invalid_kernel(float* D, long long* I, long long count, int metric) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < count) {
        D[i] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
        I[i] = -1;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ db, int dim,
                                                          const long long* __restrict__ idx, long long id_base,
                                                          long long count, float* __restrict__ out) {
    const long long w = blockIdx.x * 4LL + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= count) return;
    const long long id = idx[w] - id_base;
    float* o = out + (size_t)w * dim;
    if (idx[w] < 0) {
        for (int i = lane * 4; i < dim; i += 256) *reinterpret_cast<f32x4*>(o + i) = f32x4{0, 0, 0, 0};
        return;
    }
    const float* x = db + (size_t)id * dim;
    for (int i = lane * 4; i < dim; i += 256) *reinterpret_cast<f32x4*>(o + i) = *reinterpret_cast<const f32x4*>(x + i);
}

// merge of per-shard partial results [parts, nq, k] -> [nq, k]; one thread per query (tiny).
__global__ void merge_parts_kernel(const float* __restrict__ Dp, const long long* __restrict__ Ip, int parts, int nq,
                                   int k, int metric, float* __restrict__ D, long long* __restrict__ I) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int pos[64];
    for (int p = 0; p < parts; ++p) pos[p] = 0;
    for (int r = 0; r < k; ++r) {
        int bp = -1;
        float bk = 0.f;
        long long bi = 0;
        for (int p = 0; p < parts; ++p) {
            if (pos[p] >= k) continue;
            const size_t o = ((size_t)p * nq + q) * k + pos[p];
            const long long id = Ip[o];
            if (id < 0) continue;
            const float key = metric == KEDS_METRIC_L2 ? Dp[o] : -Dp[o];
            if (bp < 0 || key < bk || (key == bk && id < bi)) {
                bp = p;
                bk = key;
                bi = id;
            }
        }
        if (bp < 0) {
            D[(size_t)q * k + r] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
            I[(size_t)q * k + r] = -1;
        } else {
            D[(size_t)q * k + r] = metric == KEDS_METRIC_L2 ? bk : -bk;
            I[(size_t)q * k + r] = bi;
            pos[bp] += 1;
        }
    }
}

int device_cus() { return keds_device_cus(); }

constexpr int MAXQB = 8;   // query blocks searched per launch set (1024 queries)
constexpr size_t EXACT_BUDGET = (size_t)256 << 20;   // bytes of chunk lists the exact fallback may use

// rows per chunk of the exact fallback: at most 2048 chunks (exact_merge_query: 8 lists per thread), chunk lists for
// every query of a launch set within EXACT_BUDGET, and a survivor list that fits LDS (16384 rows = 128 KiB); 0 = impossible
int exact_chunk_rows(int64_t n, int nq_set, int k) {
    for (int rows = 2048; rows <= 16384; rows *= 2) {
        const int64_t chunks = (n + rows - 1) / rows;
        if (chunks <= 2048 && (size_t)chunks * nq_set * k * 8 <= EXACT_BUDGET) return rows;
    }
    return 0;
}

struct SearchWs {
    float* qn;       // [nq_pad, dim]
    bf16_t* qb;      // [1024, dim]
    unsigned long long* lpairs;   // [1024 queries x (256 / nqb) workgroups][4 * LISTK] packed (key, id) pairs of the scan
    uint2* lmeta;                 // [1024 x (256 / nqb)] {valid pairs, last-slot key}
    int* cidx;       // [1024, ncand]
    float* cval;
    float* cdist;
    int* aidx;       // [1024, ncand] phase-A candidates
    float* aval;
    float* thr;      // [1024] phase-B insert thresholds
    float* sbound;   // [1024] bf16 score bound of everything outside the candidate set
    f32x4* qstat;    // [1024] query norms for the certificate
    float* dk;       // [1024] pruning bound of the exact fallback
    int* fail_ids;   // [1024]
    int* fslot;      // [1024]
    int* counters;   // [8 + 1024]
    unsigned long long* plist;   // [nq_set, nchunks, k] exact fallback chunk lists
    int* pcnt;       // [nq_set, nchunks]
    int chunk_rows, nchunks;
    size_t bytes;
};

SearchWs carve(void* ws, int nq, int dim, int64_t n, int k) {
    SearchWs w;
    char* p = (char*)ws;
    const int nwg = 256;  // upper bound used for sizing
    size_t off = 0;
    auto take = [&](size_t b) {
        char* r = p ? p + off : nullptr;
        off += keds_align_up(b, 256);
        return r;
    };
    const int ncand = k > LISTK ? NCAND_WIDE : NCAND;
    const size_t nq_pad = keds_align_up((size_t)nq, QBLOCK);
    const size_t set = MAXQB * QBLOCK;
    w.qn = (float*)take(nq_pad * dim * sizeof(float));
    w.qb = (bf16_t*)take(set * dim * 2);
    w.lpairs = (unsigned long long*)take((size_t)QBLOCK * nwg * 4 * LISTK * sizeof(unsigned long long));
    w.lmeta = (uint2*)take((size_t)QBLOCK * nwg * sizeof(uint2));
    w.cidx = (int*)take(set * ncand * sizeof(int));
    w.cval = (float*)take(set * ncand * sizeof(float));
    w.cdist = (float*)take(set * ncand * sizeof(float));
    w.aidx = (int*)take(set * ncand * sizeof(int));
    w.aval = (float*)take(set * ncand * sizeof(float));
    w.thr = (float*)take(set * sizeof(float));
    w.sbound = (float*)take(set * sizeof(float));
    w.qstat = (f32x4*)take(set * sizeof(f32x4));
    w.dk = (float*)take(set * sizeof(float));
    w.fail_ids = (int*)take(set * sizeof(int));
    w.fslot = (int*)take(set * sizeof(int));
    w.counters = (int*)take((8 + set) * sizeof(int));   // [0] failed certificates | [8 + f] finished chunks of fail slot f
    const int nq_set = nq < (int)set ? nq : (int)set;
    w.chunk_rows = exact_chunk_rows(n, nq_set, k);
    w.nchunks = w.chunk_rows ? (int)((n + w.chunk_rows - 1) / w.chunk_rows) : 0;
    w.plist = (unsigned long long*)take((size_t)nq_set * w.nchunks * k * 8);
    w.pcnt = (int*)take((size_t)nq_set * w.nchunks * sizeof(int));
    w.bytes = off;
    return w;
}

int g_scan_debug = 0;   // timing-only ablations of the D=768 scan kernel
int g_scan_phases = 0;  // test hook: 1 forces the single-phase scan (no thresholds)
int g_force_exact = 0;  // test hook: 1 sends every query through the exact fallback
int g_scan_nt = 1;      // non-temporal LDS-DMA for the D = 768 candidate pass (A/B hook: keds_scan_debug bit 6 turns it off)
int g_thr_depth = 0;    // threshold-pass list depth: 0 = by launch shape, 1 / 4 forced (A/B hook: keds_scan_debug bits 7-9)
int g_tail_fused = 0; // keds_scan_debug bit 10: merge + re-rank + certificate in ONE launch (A/B, tests; see search_tail_fused)
// (Measured, round 4, same box, tools/search_profile.py: fused 234-242 us per search, three launches 237 -- one block per
// query re-ranks its 64 candidates on four waves where rerank_kernel spreads 8,192 waves over the chip; the two kernel
// boundaries it saves cost less than that.  OFF by default; KEDS_SEARCH_FUSED=1 / keds_scan_debug bit 10 select it.)
#ifdef KEDS_EXPERIMENTS
bool search_tail_fused() {
    static int env = -1;
    if (env < 0) {
        const char* e = keds_exp_env("KEDS_SEARCH_FUSED");
        env = e && e[0] == '1';
    }
    return env || g_tail_fused;
}
#endif

template <int D, int L>
int launch_scan(const void* packed, int stage_begin, int total_stages, const bf16_t* qb, const float* thr,
                unsigned long long* lval, uint2* lidx, int nwg, int nqb, hipStream_t st) {
    using C = ScanCfg<D>;
    const size_t lds = (size_t)C::NST * C::LDS_STAGE;
    // the key stream's buffer descriptor starts at the workgroup's first stage and takes signed 32-bit offsets (2 GiB): an
    // out-of-range buffer load returns zeros SILENTLY -- wrong neighbours, no fault -- so the span is checked here
    KEDS_REQUIRE(nwg > 0 && ((long long)(total_stages + nwg - 1) / nwg + 1) * (long long)C::STAGEB < (1LL << 31),
                 "keds_index_search: %d stages over %d workgroups: a workgroup's key stream would span 2 GiB or more "
                 "(search the index in row ranges)", total_stages, nwg);
    if (int rc = keds_func_lds_once((const void*)scan_topk_kernel<D, L>, (int)lds, "scan_topk_kernel")) return rc;
    KedsProfScope prof(KEDS_PROF_SCAN, st, /*lazy*/ !g_scan_debug);
    if constexpr (D == 768 && L == LISTK) {
        if (g_scan_debug) {
#define KEDS_SCAN_DBG(V)                                                                                            \
    {                                                                                                              \
        (void)hipFuncSetAttribute((const void*)scan_topk_kernel<D, L, V>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                       \
        scan_topk_kernel<D, L, V><<<dim3(nwg, nqb), SCAN_THREADS, lds, st>>>((const char*)packed, stage_begin, total_stages, qb, thr, \
                                                               lval, lidx);                                        \
    }
            switch (g_scan_debug) {
                case 1: KEDS_SCAN_DBG(1) break;
                case 2: KEDS_SCAN_DBG(2) break;
                default: KEDS_SCAN_DBG(3) break;
            }
#undef KEDS_SCAN_DBG
            return keds_check_launch("scan_topk_kernel<dbg>");
        }
    }
    if constexpr (D == 768 && L == LISTK) {
        if (g_scan_nt) {
            if (int rc = keds_func_lds_once((const void*)scan_topk_kernel<D, L, 0, 1>, (int)lds, "scan_topk_kernel<nt>")) return rc;
            KEDS_LAUNCH((scan_topk_kernel<D, L, 0, 1>), dim3(nwg, nqb), SCAN_THREADS, lds, st, (const char*)packed, stage_begin, total_stages, qb,
                        thr, lval, lidx);
            return keds_check_launch("scan_topk_kernel<nt>");
        }
    }
    KEDS_LAUNCH((scan_topk_kernel<D, L>), dim3(nwg, nqb), SCAN_THREADS, lds, st, (const char*)packed, stage_begin, total_stages, qb, thr, lval,
                lidx);
    return keds_check_launch("scan_topk_kernel");
}

// pack the stages that hold rows >= old_n (a partly filled last stage is re-packed); the bounds of the rows before
// old_n are carried over from where the trailer sat when the image ended at old_n
template <int D>
int launch_pack(const float* db, int64_t old_n, int64_t n, int metric, void* packed, hipStream_t st) {
    using C = ScanCfg<D>;
    const int64_t stages = (n + STAGE_KEYS - 1) / STAGE_KEYS;
    const int64_t old_stages = (old_n + STAGE_KEYS - 1) / STAGE_KEYS, first = old_n / STAGE_KEYS;
    DbBounds* bounds = (DbBounds*)((char*)packed + (size_t)stages * C::STAGEB);
    hipError_t e = hipSuccess;
    if (old_n == 0)
        e = hipMemsetAsync(bounds, 0, TRAILER, st);
    else if (old_stages != stages)
        e = hipMemcpyAsync(bounds, (char*)packed + (size_t)old_stages * C::STAGEB, TRAILER, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        keds_set_error("keds_index_pack: trailer set-up failed");
        return KEDS_E_LAUNCH;
    }
    pack_kernel<D><<<(unsigned)(stages - first), 256, 0, st>>>(db, n, metric, (char*)packed, bounds, first);
    return keds_check_launch("pack_kernel");
}

bool dim_supported(int dim) { return dim == 128 || dim == 256 || dim == 512 || dim == 768 || dim == 1024; }

size_t stage_bytes(int dim) { return (size_t)STAGE_KEYS * dim * 2 + 128; }

}  // namespace

extern "C" int keds_merge_stamp_buffer(void* buf) {     // diagnostic: >= 8 uint64 per query; nullptr switches the stamps off
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_merge_stamp), &p, sizeof(p)) == hipSuccess ? KEDS_OK : KEDS_E_LAUNCH;
}

extern "C" int keds_scan_debug(int variant) {
    g_scan_debug = variant & 15;          // bits 0-3: timing-only ablation
    g_scan_phases = (variant >> 4) & 1;   // bit 4: force the single-phase scan (exact as well; for A/B tests)
    g_force_exact = (variant >> 5) & 1;   // bit 5: fail every certificate (tests of the exact fallback)
    g_scan_nt = (variant >> 6) & 1 ? 0 : 1;   // bit 6: default-policy key stream instead of non-temporal (A/B)
    g_thr_depth = (variant >> 7) & 7;         // bits 7-9: threshold-pass list depth 1 or 4 forced (0: by launch shape)
    if (g_thr_depth != 1 && g_thr_depth != 4) g_thr_depth = 0;
    g_tail_fused = (variant >> 10) & 1;     // bit 10: merge + re-rank + certificate in one launch (A/B, tests)
    return KEDS_OK;
}

extern "C" size_t keds_index_packed_bytes(int64_t n, int dim) {
    if (n < 0 || !dim_supported(dim)) return 0;
    const int64_t stages = (n + STAGE_KEYS - 1) / STAGE_KEYS;
    return (size_t)stages * stage_bytes(dim) + TRAILER;      // + DbBounds (the certificate's rounding bounds)
}

extern "C" int keds_index_pack_append(const float* db, int64_t old_n, int64_t n, int dim, int metric, void* packed,
                                      void* stream) {
    KEDS_REQUIRE(db && packed && n > 0, "keds_index_pack: null pointer or empty database");
    KEDS_REQUIRE(old_n >= 0 && old_n < n, "keds_index_pack_append: old_n must be in [0, n)");
    KEDS_REQUIRE(dim_supported(dim), "keds_index_pack: dim %d unsupported (128,256,512,768,1024)", dim);
    KEDS_REQUIRE(metric == KEDS_METRIC_L2 || metric == KEDS_METRIC_IP, "keds_index_pack: bad metric");
    // the exact fp32 pass that backs the certificate walks a shard in at most 2,048 chunks of at most 16,384 rows
    // (exact_chunk_rows): a larger shard could be packed but never searched -- refuse it here, where the caller builds it
    KEDS_REQUIRE(n <= 2048LL * 16384, "keds_index_pack: at most %lld rows per shard (shard the database: keds_index_set_base)",
                 2048LL * 16384);
    hipStream_t st = (hipStream_t)stream;
    switch (dim) {
        case 128: return launch_pack<128>(db, old_n, n, metric, packed, st);
        case 256: return launch_pack<256>(db, old_n, n, metric, packed, st);
        case 512: return launch_pack<512>(db, old_n, n, metric, packed, st);
        case 768: return launch_pack<768>(db, old_n, n, metric, packed, st);
        default: return launch_pack<1024>(db, old_n, n, metric, packed, st);
    }
}

extern "C" int keds_index_pack(const float* db, int64_t n, int dim, int metric, void* packed, void* stream) {
    return keds_index_pack_append(db, 0, n, dim, metric, packed, stream);
}

extern "C" size_t keds_index_search_workspace_bytes_ex(int nq, int dim, int64_t n, int k) {
    if (nq <= 0 || dim <= 0 || n <= 0 || k < 1 || k > MAXK) return 0;
    return carve(nullptr, nq, dim, n, k).bytes;
}

// (sized for k <= 16 over at most 4 M rows; use the _ex form for anything else)
extern "C" size_t keds_index_search_workspace_bytes(int nq, int dim) {
    return keds_index_search_workspace_bytes_ex(nq, dim, (int64_t)1 << 22, LISTK);
}

extern "C" int keds_index_search_packed_ex(const void* packed, const float* db, int64_t n, int dim, int metric,
                                           const float* queries, int nq, int normalize_q, int k, int64_t id_base, float* D,
                                           int64_t* I, float* rows_out, void* workspace, size_t workspace_bytes,
                                           int32_t* status, void* stream) {
    KEDS_REQUIRE(packed && db && queries && D && I && workspace, "keds_index_search_packed: null pointer");
    KEDS_REQUIRE(dim_supported(dim), "keds_index_search_packed: dim %d unsupported", dim);
    KEDS_REQUIRE(n > 0 && nq > 0, "keds_index_search_packed: empty database or query set");
    KEDS_REQUIRE(k >= 1 && k <= MAXK, "keds_index_search_packed: k must be in [1,%d] (got %d)", MAXK, k);
    KEDS_REQUIRE(metric == KEDS_METRIC_L2 || metric == KEDS_METRIC_IP, "keds_index_search_packed: bad metric");
    SearchWs w = carve(workspace, nq, dim, n, k);
    KEDS_REQUIRE(w.chunk_rows > 0, "keds_index_search_packed: %lld rows x %d queries x k=%d exceed the exact-fallback budget; "
                 "search fewer queries per call", (long long)n, nq, k);
    if (workspace_bytes < w.bytes) {
        keds_set_error("keds_index_search_packed: workspace %zu < %zu bytes (keds_index_search_workspace_bytes_ex)",
                       workspace_bytes, w.bytes);
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int total_stages = (int)((n + STAGE_KEYS - 1) / STAGE_KEYS);
    const DbBounds* bounds = (const DbBounds*)((const char*)packed + (size_t)total_stages * stage_bytes(dim));
    const int ncand = k > LISTK ? NCAND_WIDE : NCAND;
    int cus = device_cus();
    if (cus > 256) cus = 256;
    // Two phases so that list inserts are rare for ANY data: phase A scans the first 1/16 of the stages with empty
    // lists (every lane fills its list: insert-heavy, but on 6 % of the bytes), the merge of phase A yields the
    // ncand-th best score per query, and phase B scans everything accepting only scores above that threshold
    // (expected accept rate 64 / rows(A): ~0.2 % at 0.5 M rows).  Small databases use one phase.
    // (Measured, round 2: the sample size is flat around 1/16 -- 1/8 and 1/12 give the same whole-search time, 1/32 +8 us,
    // 1/64 +45 us, 1/128 +75 us at 0.5 M rows: at a 0.2 % accept rate one compare in eight still takes the 16-slot insert for
    // its whole wave, which is what a larger sample buys back.)
    const bool two_phase = g_scan_phases != 1 && total_stages >= 16 * 64;
    // The sample: 1/16 of the rows, but at least 16 k rows (512 stages) where a quarter of the database allows it.  Round 4: at the
    // 8-GPU shard shape (1,024 queries x 62.5 k rows) a 1/16 sample is 3.9 k rows, the accept rate 64 / 3.9 k = 1.6 % instead of
    // the 0.2 % of the 0.5 M-row search, and with eight query blocks per launch set nothing hides the inserts: the candidate
    // pass took 236 us of which 125 were list updates (keds_scan_debug ablations under rocprofv3).  Whole search at that shape with
    // 1/16, 1/8, 1/6, 1/4, 1/3 of the rows sampled: 342, 301, 294, 292, 299 us; the 0.5 M-row search is unchanged (its 1/16 is 31 k
    // rows).  KEDS_SCAN_SAMPLE_DIV=<d> forces 1/d (A/B).
    static int sample_div = -1;
    if (sample_div < 0) {
        const char* e = keds_exp_env("KEDS_SCAN_SAMPLE_DIV");
        sample_div = e && atoi(e) >= 2 ? atoi(e) : 0;
    }
    int stagesA = total_stages / 16;
    if (sample_div) stagesA = total_stages / sample_div;
    else if (stagesA < 512) stagesA = total_stages / 4 < 512 ? total_stages / 4 : 512;
    int rc;
    // Up to MAXQB query blocks (1024 queries) share one set of launches: grid.y = query block, and the `cus`
    // workgroups are divided between the blocks, each block's workgroups splitting the stage range.  A row-sharded
    // multi-GPU search (every rank scans its small shard for ALL ranks' queries) then costs one launch set, and the
    // shard is re-streamed per block from the Infinity Cache rather than HBM.
    for (int q0 = 0; q0 < nq; q0 += MAXQB * QBLOCK) {
        const int nb = nq - q0 < MAXQB * QBLOCK ? nq - q0 : MAXQB * QBLOCK;      // queries in this launch set
        const int nqb = (nb + QBLOCK - 1) / QBLOCK;
        int per = cus / nqb;
        if (per < 1) per = 1;
        const int nwgA = stagesA < per ? (stagesA < 1 ? 1 : stagesA) : per;
        const int nwgB = total_stages < per ? total_stages : per;
        float* qn = w.qn + (size_t)q0 * dim;
        qprep_kernel<<<nqb * QBLOCK / 4, 256, 0, st>>>(queries + (size_t)q0 * dim, nb, dim, normalize_q, qn, w.qb,
                                                      nqb * QBLOCK, w.qstat, w.counters);
        if ((rc = keds_check_launch("qprep_kernel"))) return rc;
        // Threshold pass list depth.  Round 3: ONE entry per lane (its running maximum) wherever that still leaves 4 x ncand
        // lane maxima per query: the ncand-th largest of nwgA * 4 group maxima is reached by ncand distinct rows (all a
        // threshold has to guarantee), and with ~30 rows per group it sits within a few places of the exact ncand-th best of
        // the sample -- the top ncand rows fall into distinct groups almost surely -- while the pass drops from an
        // insert-bound 2.2 TB/s (depth-4 lists filling from empty) to streaming speed and its merge reads a quarter of the
        // entries.  Few workgroups per query block (row-sharded search of many query blocks) keep depth 4.
        const int depthA = g_thr_depth ? g_thr_depth : ((long)nwgA * 4 >= 4L * ncand ? 1 : 4);
        auto scan_thr = [&]() -> int {                             // shallow lists, no threshold
            if (depthA == 1) {
                switch (dim) {
                    case 128: return launch_scan<128, 1>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                    case 256: return launch_scan<256, 1>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                    case 512: return launch_scan<512, 1>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                    case 768: return launch_scan<768, 1>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                    default: return launch_scan<1024, 1>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                }
            }
            switch (dim) {
                case 128: return launch_scan<128, 4>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                case 256: return launch_scan<256, 4>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                case 512: return launch_scan<512, 4>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                case 768: return launch_scan<768, 4>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
                default: return launch_scan<1024, 4>(packed, 0, stagesA, w.qb, nullptr, w.lpairs, w.lmeta, nwgA, nqb, st);
            }
        };
        auto scan_all = [&](const float* thr) -> int {            // depth-16 lists over every stage
            switch (dim) {
                case 128: return launch_scan<128, LISTK>(packed, 0, total_stages, w.qb, thr, w.lpairs, w.lmeta, nwgB, nqb, st);
                case 256: return launch_scan<256, LISTK>(packed, 0, total_stages, w.qb, thr, w.lpairs, w.lmeta, nwgB, nqb, st);
                case 512: return launch_scan<512, LISTK>(packed, 0, total_stages, w.qb, thr, w.lpairs, w.lmeta, nwgB, nqb, st);
                case 768: return launch_scan<768, LISTK>(packed, 0, total_stages, w.qb, thr, w.lpairs, w.lmeta, nwgB, nqb, st);
                default: return launch_scan<1024, LISTK>(packed, 0, total_stages, w.qb, thr, w.lpairs, w.lmeta, nwgB, nqb, st);
            }
        };
        if (two_phase) {
            // Threshold pass: every lane keeps the best 4 scores of its share of the first 1/16 of the rows; the
            // ncand-th best of their union is a score that ncand real rows reach, so the final candidates all beat or
            // equal it.  The candidate pass then rescans everything, inserting only above that threshold, so list
            // inserts are rare for ANY data.
            if ((rc = scan_thr())) return rc;
            KedsProfScope prof(KEDS_PROF_OTHER, st);
            merge_pairs_kernel<false><<<nb, MERGE_THREADS, 0, st>>>(w.lpairs, w.lmeta, nwgA, 4 * depthA, w.aidx, w.aval, w.thr, ncand,
                                                                    nullptr, nullptr, TailArgs{});
            if ((rc = keds_check_launch("merge_pairs_kernel(thr)"))) return rc;
        }
        if ((rc = scan_all(two_phase ? w.thr : nullptr))) return rc;
        {
            KedsProfScope prof(KEDS_PROF_OTHER, st);
            float* Dq = D + (size_t)q0 * k;
            long long* Iq = (long long*)I + (size_t)q0 * k;
#ifdef KEDS_EXPERIMENTS
            if (search_tail_fused()) {       // merge + exact re-rank + selection + certificate: one launch
                const TailArgs ta{db, qn, w.qstat, bounds, Dq, Iq, w.counters, w.fail_ids, w.fslot, w.dk, status, (long long)id_base,
                                  dim, metric, k, g_force_exact};
                merge_pairs_kernel<true><<<nb, MERGE_THREADS, 0, st>>>(w.lpairs, w.lmeta, nwgB, 4 * LISTK, w.cidx, w.cval, nullptr, ncand,
                                                                       two_phase ? w.thr : nullptr, w.sbound, ta);
                if ((rc = keds_check_launch("merge_pairs_kernel<fused>"))) return rc;
            } else
#endif
            {
                merge_pairs_kernel<false><<<nb, MERGE_THREADS, 0, st>>>(w.lpairs, w.lmeta, nwgB, 4 * LISTK, w.cidx, w.cval, nullptr, ncand,
                                                                        two_phase ? w.thr : nullptr, w.sbound, TailArgs{});
                if ((rc = keds_check_launch("merge_pairs_kernel"))) return rc;
                rerank_kernel<<<(nb * ncand + 3) / 4, 256, 0, st>>>(db, dim, metric, qn, w.cidx, w.cdist, nb * ncand, ncand);
                if ((rc = keds_check_launch("rerank_kernel"))) return rc;
                certify_select_kernel<<<nb, ncand, 0, st>>>(w.cidx, w.cdist, ncand, metric, k, (long long)id_base, Dq, Iq, w.qstat,
                                                            w.sbound, bounds, w.counters, w.fail_ids, w.fslot, w.dk, status,
                                                            g_force_exact);
                if ((rc = keds_check_launch("certify_select_kernel"))) return rc;
            }
            // exact fallback for the queries whose certificate failed (both kernels return at once when none did)
            const size_t lds = (size_t)dim * 4 + (size_t)w.chunk_rows * 8;
            if ((rc = keds_func_lds_once((const void*)exact_chunk_kernel, (int)lds, "exact_chunk_kernel"))) return rc;
            exact_chunk_kernel<<<4 * cus, 256, lds, st>>>(db, n, dim, metric, qn, w.counters, w.fail_ids, w.dk, w.chunk_rows,
                                                         w.nchunks, k, w.plist, w.pcnt, w.counters + 8, (long long)id_base, Dq, Iq);
            if ((rc = keds_check_launch("exact_chunk_kernel"))) return rc;
        }
    }
    if (rows_out) {
        const long long cnt = (long long)nq * k;
        gather_rows_kernel<<<(unsigned)((cnt + 3) / 4), 256, 0, st>>>(db, dim, (const long long*)I, (long long)id_base,
                                                                       cnt, rows_out);
        if ((rc = keds_check_launch("gather_rows_kernel"))) return rc;
    }
    return KEDS_OK;
}

extern "C" int keds_index_search_packed(const void* packed, const float* db, int64_t n, int dim, int metric,
                                 const float* queries, int nq, int normalize_q, int k, int64_t id_base, float* D,
                                 int64_t* I, float* rows_out, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    return keds_index_search_packed_ex(packed, db, n, dim, metric, queries, nq, normalize_q, k, id_base, D, I, rows_out,
                                       workspace, workspace_bytes, nullptr, stream);
}

extern "C" int keds_topk_merge_parts(const float* D_parts, const int64_t* I_parts, int parts, int nq, int k,
                                     int metric, float* D, int64_t* I, void* stream) {
    KEDS_REQUIRE(D_parts && I_parts && D && I, "keds_topk_merge_parts: null pointer");
    KEDS_REQUIRE(parts >= 1 && parts <= 64 && nq > 0 && k > 0, "keds_topk_merge_parts: bad sizes");
    merge_parts_kernel<<<(nq + 63) / 64, 64, 0, (hipStream_t)stream>>>(D_parts, (const long long*)I_parts, parts, nq,
                                                                       k, metric, D, (long long*)I);
    return keds_check_launch("merge_parts_kernel");
}

// ---- packed exchange of a row-sharded search: one message per peer ----------------------------------
// A partial list entry travels as E = 3 int32 words: distance bits | id low | id high -- or, with the winner's fp32 row,
// E = 4 + dim: the same three, one pad word (rows stay 16-byte aligned), the row.
// send[w][b][j][E] (part w starts at w * w_stride words) = this shard's list entry j for query b of rank w.
__global__ __launch_bounds__(256) void exchange_pack_kernel(const float* __restrict__ Dp, const long long* __restrict__ Ip,
                                                            const float* __restrict__ Rp, int B, int k, int dim, int E,
                                                            long long w_stride, int* __restrict__ send) {
    const int list = blockIdx.x;                                  // w * B + b
    const int w = list / B, b = list - w * B;
    int* dst = send + (size_t)w * w_stride + (size_t)b * k * E;
    const float* d = Dp + (size_t)list * k;
    const long long* ids = Ip + (size_t)list * k;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        const long long id = ids[j];
        dst[(size_t)j * E + 0] = __float_as_int(d[j]);
        dst[(size_t)j * E + 1] = (int)(unsigned)(id & 0xFFFFFFFFll);
        dst[(size_t)j * E + 2] = (int)(id >> 32);
    }
    if (Rp) {
        const f32x4* r = reinterpret_cast<const f32x4*>(Rp + (size_t)list * k * dim);
        const int per = dim >> 2;
        for (int i = threadIdx.x; i < k * per; i += blockDim.x) {
            const int j = i / per, c = i - j * per;
            *reinterpret_cast<f32x4*>(dst + (size_t)j * E + 4 + 4 * c) = r[i];
        }
        for (int j = threadIdx.x; j < k; j += blockDim.x) dst[(size_t)j * E + 3] = 0;
    }
}

constexpr int XCHG_MAX_ENTRIES = 4096;                            // world * k entries of one query in LDS

__device__ __forceinline__ unsigned ordered_key(float v) {      // unsigned order == float order (-0 folded into +0)
    const unsigned u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// recv[w][b][j][E] (part w = what shard w found for MY query b) -> the k best keyed on (distance, id) + their rows.
// Every list is sorted by (key, id) with its invalid entries (id < 0) at the tail, and ids are unique across shards, so an
// entry's place in the merged order is its own position + the number of smaller entries in every other list (binary search).
__global__ __launch_bounds__(256) void exchange_merge_kernel(const int* __restrict__ recv, int world, int B, int k, int E,
                                                             long long w_stride, int metric, float* __restrict__ D,
                                                             long long* __restrict__ I, float* __restrict__ rows, int dim) {
    __shared__ unsigned keys[XCHG_MAX_ENTRIES];
    __shared__ long long ids[XCHG_MAX_ENTRIES];
    __shared__ int nvalid[64];
    __shared__ int src[KEDS_SCAN_MAX_K];
    __shared__ int total;
    const int b = blockIdx.x, t = threadIdx.x, n = world * k;
    if (t < world) nvalid[t] = 0;
    if (t == 0) total = 0;
    for (int r = t; r < k; r += blockDim.x) src[r] = -1;
    __syncthreads();
    for (int e = t; e < n; e += blockDim.x) {
        const int w = e / k, j = e - w * k;
        const int* p = recv + (size_t)w * w_stride + ((size_t)b * k + j) * E;
        const float d = __int_as_float(p[0]);
        const long long id = ((long long)p[2] << 32) | (unsigned)p[1];
        keys[e] = ordered_key(metric == KEDS_METRIC_L2 ? d : -d);
        ids[e] = id;
        if (id >= 0) atomicAdd(&nvalid[w], 1);
    }
    __syncthreads();
    for (int e = t; e < n; e += blockDim.x) {
        const int w = e / k, j = e - w * k;
        if (j >= nvalid[w]) continue;
        const unsigned key = keys[e];
        const long long id = ids[e];
        int rank = j;
        for (int o = 0; o < world; ++o) {
            if (o == w) continue;
            int lo = 0, hi = nvalid[o];                           // first entry of list o that is NOT smaller than (key, id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const unsigned km = keys[o * k + mid];
                if (km < key || (km == key && ids[o * k + mid] < id)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            const int* p = recv + (size_t)w * w_stride + ((size_t)b * k + j) * E;
            D[(size_t)b * k + rank] = __int_as_float(p[0]);
            I[(size_t)b * k + rank] = id;
            src[rank] = e;
        }
        if (j == 0) atomicAdd(&total, nvalid[w]);
    }
    __syncthreads();
    for (int r = total + t; r < k; r += blockDim.x) {           // fewer than k valid entries in all shards together
        D[(size_t)b * k + r] = metric == KEDS_METRIC_L2 ? INFINITY : -INFINITY;
        I[(size_t)b * k + r] = -1;
    }
    if (rows) {
        const int per = dim >> 2;
        for (int i = t; i < k * per; i += blockDim.x) {
            const int r = i / per, c = i - r * per;
            const int e = src[r];
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e >= 0) {
                const int w = e / k, j = e - w * k;
                v = *reinterpret_cast<const f32x4*>(recv + (size_t)w * w_stride + ((size_t)b * k + j) * E + 4 + 4 * c);
            }
            *reinterpret_cast<f32x4*>(rows + ((size_t)b * k + r) * dim + 4 * c) = v;
        }
    }
}

extern "C" int keds_exchange_pack(const float* D_p, const int64_t* I_p, const float* rows_p, int world, int B, int k, int dim,
                                  int64_t w_stride, int32_t* send, void* stream) {
    KEDS_REQUIRE(D_p && I_p && send && world >= 1 && B > 0 && k > 0, "keds_exchange_pack: bad argument");
    KEDS_REQUIRE(!rows_p || (dim > 0 && dim % 4 == 0), "keds_exchange_pack: rows need dim %% 4 == 0");
    const int E = rows_p ? dim + 4 : 3;
    KEDS_REQUIRE(w_stride >= (int64_t)B * k * E && (!rows_p || w_stride % 4 == 0), "keds_exchange_pack: bad w_stride");
    exchange_pack_kernel<<<world * B, 256, 0, (hipStream_t)stream>>>(D_p, (const long long*)I_p, rows_p, B, k, dim, E,
                                                                    (long long)w_stride, send);
    return keds_check_launch("exchange_pack_kernel");
}

extern "C" int keds_exchange_merge(const int32_t* recv, int world, int B, int k, int dim, int64_t w_stride, int metric,
                                   float* D, int64_t* I, float* rows, void* stream) {
    KEDS_REQUIRE(recv && D && I && world >= 1 && world <= 64 && B > 0 && k > 0 && k <= KEDS_SCAN_MAX_K,
                 "keds_exchange_merge: bad argument");
    KEDS_REQUIRE((long)world * k <= XCHG_MAX_ENTRIES, "keds_exchange_merge: world * k = %ld exceeds %d", (long)world * k,
                 XCHG_MAX_ENTRIES);
    KEDS_REQUIRE(!rows || (dim > 0 && dim % 4 == 0), "keds_exchange_merge: rows need dim %% 4 == 0");
    const int E = rows ? dim + 4 : 3;
    KEDS_REQUIRE(w_stride >= (int64_t)B * k * E && (!rows || (w_stride % 4 == 0)), "keds_exchange_merge: bad w_stride");
    exchange_merge_kernel<<<B, 256, 0, (hipStream_t)stream>>>(recv, world, B, k, E, (long long)w_stride, metric, D,
                                                             (long long*)I, rows, dim);
    return keds_check_launch("exchange_merge_kernel");
}

extern "C" int keds_gather_rows(const float* db, int dim, const int64_t* idx, int64_t count, float* out,
                                void* stream) {
    KEDS_REQUIRE(db && idx && out && count > 0, "keds_gather_rows: bad argument");
    KEDS_REQUIRE(dim % 4 == 0, "keds_gather_rows: dim must be a multiple of 4");
    gather_rows_kernel<<<(unsigned)((count + 3) / 4), 256, 0, (hipStream_t)stream>>>(db, dim, (const long long*)idx, 0,
                                                                                      count, out);
    return keds_check_launch("gather_rows_kernel");
}
