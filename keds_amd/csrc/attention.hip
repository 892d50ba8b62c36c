// Multi-head self-attention core: out = softmax(q k^T / sqrt(64) [+ causal mask]) v on a packed
// qkv buffer.  Reference: nn.MultiheadAttention inside ResidualAttentionBlock
// (src/model/model.py:309,319-321), causal mask src/model/model.py:543-549.
//
// gfx950 design.  Sequences are short (257 / 77 tokens), so one workgroup owns one (batch, head):
// all K rows and V^T stay resident in LDS (74 KB at S=257 -> 2 workgroups per CU) and no online
// softmax is needed.  Each wave takes 16-query tiles.  Scores are computed "swapped",
// S^T = K . Q^T, so a lane holds ONE query column and 4 keys per 16-key tile: the row max / sum
// are in-lane reductions plus two xor-shuffles, and the probabilities, packed to bf16 in
// registers, are already the B operand of O^T = V^T . P^T (the MFMA k-slot order is permuted to
// match: k-slot (g, j) of step u is key 32u + 16*(j>>2) + 4g + (j&3); V^T is staged in that order).
#include "keds_common.h"
#include <math.h>

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

template <int NKT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S,
                                                        int heads, int causal) {
    using C = AttnCfg<NKT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* vt_lds = smem + C::K_BYTES;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int d = heads * DH;
    const int ld = 3 * d;  // qkv row stride (elements)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * DH;

    // ---- stage K (swizzled rows) and V^T (permuted key order); keys >= S are zero
    for (int id = tid; id < C::KEYS * 8; id += 256) {
        const int key = id >> 3, ch = id & 7;
        bf16x8 kv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (key < S) {
            const bf16_t* row = base + (size_t)key * ld + ch * 8;
            kv = *reinterpret_cast<const bf16x8*>(row + d);
            vv = *reinterpret_cast<const bf16x8*>(row + 2 * d);
        }
        *reinterpret_cast<bf16x8*>(k_lds + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kv;
        const int u = key >> 5, w = key & 31;
        const int pos = (4 * u + ((w & 15) >> 2)) * 16 + ((w & 3) + ((w >> 4) << 2)) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<bf16_t*>(vt_lds + (ch * 8 + j) * C::VT_ROW + pos) = vv[j];
    }
    __syncthreads();

    const float sl2 = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    const int nqt = (S + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int qidx = qt * 16 + c;
        const int qrow = qidx < S ? qidx : S - 1;
        bf16x8 qf[2];
        {
            const bf16_t* qp = base + (size_t)qrow * ld + 8 * g;
            qf[0] = *reinterpret_cast<const bf16x8*>(qp);
            qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
        }
        // ---- S^T tiles
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int key = kt * 16 + c;
            const char* kr = k_lds + key * 128;
            const int f = (key >> 1) & 7;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(kr + ((g ^ f) << 4));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(kr + (((4 + g) ^ f) << 4));
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[1], acc, 0, 0, 0);
            sc[kt] = acc;
            __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later tiles from being hoisted (VGPR pressure)
        }
        // ---- mask + row max (lane holds query qidx, keys kt*16 + 4g + r)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kt * 16 + 4 * g + r;
                const bool ok = kidx < S && (!causal || kidx <= qidx);
                const float v = ok ? sc[kt][r] : -INFINITY;
                sc[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f((sc[kt][r] - mx) * sl2);   // exp2(-inf) = 0 for masked keys
                sc[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        // ---- O^T = V^T . P^T
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            const f32x4 p0 = sc[2 * u], p1 = sc[2 * u + 1];
            const bf16x8 pf = bf16x8{(bf16_t)p0[0], (bf16_t)p0[1], (bf16_t)p0[2], (bf16_t)p0[3],
                                     (bf16_t)p1[0], (bf16_t)p1[1], (bf16_t)p1[2], (bf16_t)p1[3]};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 a =
                    *reinterpret_cast<const bf16x8*>(vt_lds + (dt * 16 + c) * C::VT_ROW + (4 * u + g) * 16);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qidx < S) {
            const float inv = 1.0f / sum;
            bf16_t* op = out + ((size_t)b * S + qidx) * d + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = o[dt] * inv;
                *reinterpret_cast<bf16x4*>(op + dt * 16) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
        }
    }
}

template <int NKT>
int launch_attn(const void* qkv, void* out, int B, int S, int heads, int causal, hipStream_t st) {
    using C = AttnCfg<NKT>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attention_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS) != hipSuccess) {
            keds_set_error("attention: cannot set dynamic LDS size %d", C::LDS);
            return KEDS_E_LAUNCH;
        }
        attr_set = true;
    }
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    attention_kernel<NKT><<<B * heads, 256, C::LDS, st>>>((const bf16_t*)qkv, (bf16_t*)out, S, heads, causal);
    return keds_check_launch("attention_kernel");
}

}  // namespace

extern "C" int keds_attention(const void* qkv, void* out, int B, int S, int heads, int causal, void* stream) {
    KEDS_REQUIRE(qkv && out && B > 0 && heads > 0, "keds_attention: bad argument");
    KEDS_REQUIRE(S >= 1 && S <= 288, "keds_attention: S=%d unsupported (1..288)", S);
    hipStream_t st = (hipStream_t)stream;
    if (S <= 32) return launch_attn<2>(qkv, out, B, S, heads, causal, st);
    if (S <= 96) return launch_attn<6>(qkv, out, B, S, heads, causal, st);
    return launch_attn<18>(qkv, out, B, S, heads, causal, st);
}
