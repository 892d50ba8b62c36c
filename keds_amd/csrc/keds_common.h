// Shared device/host helpers for libkeds_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/keds_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define KEDS_WAVE 64

// ---- switches of the EXPERIMENT build (host) -------------------------------------------------
// The product library reads four environment variables, all documented in INTEGRATION.md: KEDS_DETERMINISTIC, KEDS_SIDE_STREAM,
// KEDS_TEXT_TRIM (and, in Python, KEDS_PRECISION / the tokenizer's KEDS_BPE_VOCAB).  Every other KEDS_* switch selects a kernel form
// or a schedule that lost a measured A/B; they exist only in a library built with `make EXTRA=-DKEDS_EXPERIMENTS` (what the A/B
// tools under tools/ build), so that the product has one dispatch path per shape.
#include <cstdlib>
#ifdef KEDS_EXPERIMENTS
inline const char* keds_exp_env(const char* name) { return getenv(name); }
#else
inline const char* keds_exp_env(const char*) { return nullptr; }
#endif

// Packed rows of a causal tower (round 6, keds_text_run_packed): sample b owns rows [off[b], off[b + 1]) of every activation buffer
// instead of [b S, (b + 1) S).  The GEMMs and LayerNorm statistics are row-wise and do not care; the attention kernels and the
// read-out-row gather take the offsets (towers.hip, f32path.hip).
struct PackedRows {
    const int32_t* off;     // device int32 [B + 1]
    int rows;               // rows the tower runs (whole 256-row tiles where the workspace allows)
    int valid;              // off[B]: rows [valid, rows) belong to no sample -- zero rows, kept finite (the attention never writes them)
};

// ---- error plumbing (host) ---------------------------------------------------------
void keds_set_error(const char* fmt, ...);
int keds_check_launch(const char* what);

#define KEDS_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            keds_set_error(__VA_ARGS__);         \
            return KEDS_E_ARG;                   \
        }                                        \
    } while (0)

// ---- profiling (host): event pair around a launch when enabled -----------------------
struct KedsProfScope {
    int klass;
    hipStream_t stream;
    void* slot;
    bool lazy, taken;          // lazy: the launches under this scope bind the pair themselves (KEDS_LAUNCH below)
    double work_units;
    hipEvent_t ev_a, ev_b;
    KedsProfScope* outer;
    KedsProfScope(int klass, hipStream_t s, bool lazy = false);
    ~KedsProfScope();
    void work(double units);   // algorithmic flops (GEMM) / bytes (scan) of this launch; counted only if it carries events
};
// Round 5: an event RECORDED on a stream is a marker packet of its own between two kernels (~3.5 us per pair: 0.9 % of the bench
// step with the pairs of every fourth step's 100 GEMM launches); as the START / STOP events of the launch itself
// (hipExtLaunchKernelGGL) they are the kernel's own dispatch signals and cost nothing.  A `lazy` scope records nothing: its
// launches go through KEDS_LAUNCH, which hands the first launch the start event and every launch the stop event (stream order: the
// last one's completion stays).  A lazy scope under which no KEDS_LAUNCH ran drops its pair.
struct KedsLaunchEvents {
    hipEvent_t start, stop;
};
KedsLaunchEvents keds_prof_launch_events(hipStream_t st);
#define KEDS_LAUNCH(kernel, grid, block, lds, st, ...)                                                                    \
    do {                                                                                                                  \
        const KedsLaunchEvents le_ = keds_prof_launch_events(st);                                                         \
        if (le_.stop)                                                                                                     \
            hipExtLaunchKernelGGL((kernel), dim3(grid), dim3(block), (unsigned)(lds), (st), le_.start, le_.stop, 0, __VA_ARGS__); \
        else                                                                                                              \
            hipLaunchKernelGGL((kernel), dim3(grid), dim3(block), (unsigned)(lds), (st), __VA_ARGS__);                    \
    } while (0)

// ---- side lane (host): a second, high-priority stream per device for the remainder-row chain of the towers ----------
// Callers fork to it / join from it with their own events (keds_stream_order).  nullptr when disabled (KEDS_SIDE_STREAM=0)
// or when the stream cannot be created: callers then run everything on their own stream.
struct KedsSideLane {
    hipStream_t s;
    hipEvent_t fork, join;      // created once with the lane (timing disabled): a stream wait binds to the event's record at
                                // the time of the call, so re-recording them on the next tower pass is safe
};
KedsSideLane* keds_side_lane();
bool keds_side_lane_enabled();
// record on `from`, make `to` wait: everything enqueued on `to` afterwards runs after everything enqueued on `from` so far
int keds_stream_order(hipStream_t from, hipEvent_t ev, hipStream_t to);
// an event recorded BY A LAUNCH (its stop event): lock, launch, keds_stream_wait_locked(ev, to), unlock (api.hip)
void keds_order_lock();
void keds_order_unlock();
int keds_stream_wait_locked(hipEvent_t ev, hipStream_t to);
// (attention.hip) the next S = 257 attention launch of this thread records `ev` as its stop event; returns through
// keds_attention_stop_event_taken() whether a launch consumed it (other kernel forms do not: the caller records it itself then)
void keds_attention_stop_event(hipEvent_t ev);
bool keds_attention_stop_event_taken();

static inline size_t keds_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- per-device launch state (host) --------------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute and the session API creates contexts on any
// device from any thread: raise it once per (kernel, device), under a lock.  Returns KEDS_OK or KEDS_E_LAUNCH.
int keds_func_lds_once(const void* func, int bytes, const char* what);
// compute units of the CURRENT device (cached per device id)
int keds_device_cus();

// ---- split-K scratch of the small-M GEMMs (host) -------------------------------------------------------------------
// A GEMM launch that splits K writes fp32 partial tiles to scratch memory and reduces them in a second launch.  A tower
// pass (keds_tower_forward: the only composite call whose shapes split K -- remainder rows / CLS tail at K >= 2048) carves
// KEDS_SPLITK_BYTES out of ITS OWN caller-supplied workspace and makes it the scratch of the GEMMs it enqueues for the
// duration of the call (thread-local scope): two handles, threads or streams never share partial sums.  The knowledge path
// runs its two CrossFormer chains on two streams AT ONCE, so it opens a scope with a NULL buffer = "no GEMM of this call
// splits K" (its K is 512 / 768: nothing would split today; the scope makes that a guarantee instead of a coincidence).
// Direct keds_gemm_bt* calls outside any scope use the per-device buffer registered with keds_gemm_set_workspace (one
// stream at a time per device), or do not split.
#define KEDS_SPLITK_BYTES ((size_t)8 << 20)      /* splits * tiles <= 128 tiles of 128 x 128 fp32 */
struct KedsSplitKScope {
    float* prev_p;
    size_t prev_bytes;
    bool prev_off;
    KedsSplitKScope(void* p, size_t bytes);
    ~KedsSplitKScope();
};
void keds_splitk_scratch(float** p, size_t* bytes);   // innermost scope of this thread, else the device registration

// ---- numerics guard (host): device int32 flag of the calling thread's current composite call, or nullptr -----------
int* keds_numerics_guard();

// ---- output-tile stores (device) -----------------------------------------------------------------------------------
// 16-byte store of an output tile that ANOTHER kernel reads next.  Policy per site class, compile-time (same-box A/B:
// tools/ab_nt.sh rebuilds with -DKEDS_ST_<class>=<policy>):
//   0 plain (write-back: the line stays dirty in the XCD's L2 until it is evicted or the kernel ends)
//   1 nt    (streaming hint; still write-back)
//   2 sc0 sc1 (write-through at system scope: the bytes leave L2 with the store and the line is dropped -- nothing is left for
//     the end-of-kernel write-back, whose cost the NEXT kernel's start pays: MI355X_MICROARCH.md price list, row "boundary":
//     + B / 6 TB/s for B bytes left dirty)
//   3 sc1 nt     4 sc1     5 sc0 sc1 nt     6 sc0     7 sc0 nt      (the remaining combinations of the three cache-policy bits)
#ifndef KEDS_ST_LN
#define KEDS_ST_LN 3        /* LayerNorm-epilogue outputs of the GEMMs (qkv, MLP hidden): read once by the next kernel.  nt since
                               round 2 (+0.55 ms of 21.7 against plain stores); round 4: sc1 nt as a buffer-store builtin, +0.4 %
                               same-box (profiles/r04_store_policy_ab_3.txt).  WARNING kept for the next reader: as an inline-asm
                               global_store_dwordx4 WITHOUT the s_nop 1 hazard pad the same policy "gained" 10 % -- the compiler
                               overwrote the data registers before the store had read them, the garbage it wrote toggles fewer
                               bits, and at the power cap fewer toggles are clock (profiles/r04_store_policy_ab_1/2.txt) */
#endif
#ifndef KEDS_ST_RESID
#define KEDS_ST_RESID 0     /* the fp16 residual stream, read-modify-written in place */
#endif
#ifndef KEDS_ST_ATTN
#define KEDS_ST_ATTN 0      /* attention output */
#endif
#ifndef KEDS_ST_FP8_BF16
#define KEDS_ST_FP8_BF16 0  /* MXFP8 GEMM: bf16 output (qkv) */
#endif
#ifndef KEDS_ST_FP8_MX
#define KEDS_ST_FP8_MX 0    /* MXFP8 GEMM: MXFP8 output (MLP hidden) */
#endif
#ifndef KEDS_ST_FP8_MXR
#define KEDS_ST_FP8_MXR 0   /* MXFP8 GEMM: MXFP8 copy of the residual stream */
#endif
// cache-policy bits of the LDS-DMA / buffer-load builtins' `aux` operand on gfx950: 1 = sc0, 2 = nt, 16 = sc1
#ifndef KEDS_LD_ATTN_AUX
#define KEDS_LD_ATTN_AUX 0  /* attention: K / V rows of one (sample, head), read once by one workgroup */
#endif
#ifndef KEDS_LD_A3_AUX
#define KEDS_LD_A3_AUX 0    /* A operand of the 4-wave residual GEMM's three-deep ring (attention output / MLP hidden: read by the
                               four column tiles of one XCD group, then dead) */
#endif
#ifndef KEDS_LD_RESID_AUX
#define KEDS_LD_RESID_AUX 0 /* the residual tile the fp16-residual epilogue reads, modifies and writes back */
#endif
#ifndef KEDS_ATTN_PREFETCH
#define KEDS_ATTN_PREFETCH 0 /* S = 257 attention: block distance of an L2 prefetch of a later workgroup's rows (0 = off).  Measured,
                                round 4 (profiles/r04_attn_prefetch_ab.txt): 256 / 512 / 1024 all cost the kernel 27 % (1.74 -> 2.2 ms
                                per step): 771 four-byte requests per workgroup are dearer than the first-byte latency they hide */
#endif
// `base` is wave-uniform (the tile's first byte), `off` the lane's 32-bit byte offset inside it: policies other than 0 go out
// as a buffer store whose `aux` operand carries the cache-policy bits -- a builtin, so the compiler counts the store and pads
// its data-register hazard (an inline-asm global_store_dwordx4 gets neither: cdna_hip_programming.md section 5.7).
template <int POLICY, typename T>
__device__ __forceinline__ void keds_store16(T v, void* base, unsigned off) {
    static_assert(sizeof(T) == 16, "16-byte stores only");
    if constexpr (POLICY == 0) {
        *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off) = v;
    } else if constexpr (POLICY >= 10 && POLICY < 20) {
        // experiment: the store as an inline-asm statement the compiler cannot move (s_nop 1: the store-data hazard pad hipcc
        // does not add for asm, cdna_hip_programming.md section 5.7); 10 plain, 11 nt, 12 sc0 sc1, 13 sc1 nt, 14 sc1
        void* p = reinterpret_cast<char*>(base) + off;
        if constexpr (POLICY == 10) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
        else if constexpr (POLICY == 11) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
        else if constexpr (POLICY == 12) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
        else if constexpr (POLICY == 13) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
    } else if constexpr (POLICY >= 20) {
        // experiment: the builtin store of policy POLICY - 20, pinned in program order by a scheduling barrier
        keds_store16<POLICY - 20>(v, base, off);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        // (8: the plain policy as a buffer store -- uniform base in SGPRs + a 32-bit lane offset instead of a 64-bit pointer per store)
        constexpr int aux = POLICY == 1 ? 2 : POLICY == 2 ? 17 : POLICY == 3 ? 18 : POLICY == 4 ? 16 : POLICY == 5 ? 19 : POLICY == 6 ? 1 : POLICY == 8 ? 0 : 3;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)off, 0, aux);
    }
}

// ---- device helpers ----------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Reductions over the lane bits 4 and 5 (the four 16-lane rows of a wave: the lanes that share an MFMA output column /
// row) with gfx950's v_permlane16_swap / v_permlane32_swap: VALU instructions instead of the ds_bpermute round trips that
// __shfl_xor(x, 16 / 32) compiles to.  With both operands = x, swap16 gives {row0,row0,row2,row2} and {row1,row1,row3,row3}.
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// XCD-aware bijective remap of a 1-D block id: blocks that the dispatcher deals to one XCD
// (b, b+8, b+16, ...) get a contiguous run of logical ids, so neighbouring tiles share an L2.
// Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// logical tile id -> (tm, tn): the 8 x 4 supertiles per XCD of the 256 x 256 tile kernels (gemm.hip, gemm_fp8.hip), m fastest
// (m_tiles need not be a multiple of 8 -- a tower's ragged 129th row tile: the first m_tiles & ~7 row tiles form the
// supertiles, the rest follow in plain order)
__device__ __forceinline__ void quad_tile_coords(int bid, int m_tiles, int n_tiles, int& tm, int& tn) {
    const int m8 = m_tiles & ~7;
    if (m8 && (n_tiles & 3) == 0 && bid < m8 * n_tiles) {
        const int grp = bid >> 5, within = bid & 31;
        const int grows = m8 >> 3;
        const int gn = grp / grows, gm = grp - gn * grows;
        tm = gm * 8 + (within & 7);          // (round 4 re-measured 16 x 2 on the round-3 kernels: 1 % slower, as in round 2)
        tn = gn * 4 + (within >> 3);
    } else if (m8 && (n_tiles & 3) == 0) {
        const int r = bid - m8 * n_tiles;
        tm = m8 + r / n_tiles;
        tn = r - (tm - m8) * n_tiles;
    } else {
        tm = bid / n_tiles;
        tn = bid - tm * n_tiles;
    }
}

// ---- LayerNorm row statistics {sum, sum of squares} accumulated across workgroups ---------------------------------
// Stored as 64-bit FIXED POINT (value * 2^28) and added with integer atomics: integer addition is associative, so the
// result does not depend on the order in which tiles arrive and every run produces the same bits (fp32 atomics did
// not: last-bit differences, amplified by e4m3 rounding in the fp8 towers).  One row = two long long (16 bytes).
typedef long long keds_stat_t;
#define KEDS_STAT_SCALE 268435456.0f            /* 2^28: exact for |partial| >= 2^-5, 2^-28 absolute below; sums < 3.4e10 */
__device__ __forceinline__ keds_stat_t keds_stat_fixed(float v) { return (keds_stat_t)__float2ll_rn(v * KEDS_STAT_SCALE); }
__device__ __forceinline__ float keds_stat_value(keds_stat_t a) { return (float)((double)a * (1.0 / 268435456.0)); }
__device__ __forceinline__ void keds_stat_add(keds_stat_t* row, float s, float ss) {
    atomicAdd(reinterpret_cast<unsigned long long*>(row), (unsigned long long)keds_stat_fixed(s));
    atomicAdd(reinterpret_cast<unsigned long long*>(row + 1), (unsigned long long)keds_stat_fixed(ss));
}
__device__ __forceinline__ void keds_stat_zero(keds_stat_t* row) {
    row[0] = 0;
    row[1] = 0;
}

// ---- OCP MX (e4m3 elements, one e8m0 scale per 32) helpers shared by gemm_fp8.hip and attention.hip --------------
// OCP MX shared exponent of a 32-element block from its amax: floor(log2(amax)) - 8 (emax of e4m3), clamped to e8m0
__device__ __forceinline__ int mx_block_exp(float amax) {
    int e = amax > 0.f ? (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8 : -127;
    return e < -127 ? -127 : (e > 127 ? 127 : e);
}
// eight values scaled by 2^-e, saturated at +-448, rounded to nearest even e4m3: two dwords of four bytes
__device__ __forceinline__ uint2 mx_pack8(const float (&v)[8], int e) {
    const float inv = e == -127 ? 0.f : __uint_as_float((unsigned)(127 - e) << 23);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = fminf(fmaxf(v[j] * inv, -448.f), 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(s[0], s[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(s[2], s[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(s[4], s[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(s[6], s[7], hi, true);
    return uint2{lo, hi};
}
// scale byte of block `blk` (32 columns) of row r in the [K/128][rows_pad] dword layout
__device__ __forceinline__ size_t mx_scale_index(int blk, int r, int rows_pad) {
    return ((size_t)(blk >> 2) * rows_pad + r) * 4 + (blk & 3);
}
