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

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;       // X tile + W tile
constexpr int GEMM_LDS = 2 * BUF_BYTES;         // double buffer = 64 KiB

// LDS row r (128 bytes = 8 chunks of 16 B): chunk c is stored at slot c ^ f(r)
__device__ __forceinline__ int swz_f(int row) { return (row >> 1) & 7; }

// LDS row R (0..127) of the W tile holds W row n0 + perm_w(R): with i = R&15 (MFMA row), g = i>>2,
// r = i&3, tile t = R>>4: n = 64*(t>>2) + 32*((t>>1)&1) + 8*g + 4*(t&1) + r, so the accumulator
// registers of tiles (2p, 2p+1) of one lane are 8 consecutive output columns.
__device__ __forceinline__ int perm_w(int R) {
    const int t = R >> 4, i = R & 15;
    return 64 * (t >> 2) + 32 * ((t >> 1) & 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
}

__device__ __forceinline__ float qgelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bt_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ bias, void* __restrict__ out,
                                                         int M, int N, int K, int n_tiles,
                                                         const float* __restrict__ aux, int aux_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / n_tiles, tn = bid - tm * n_tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int g = lane >> 4, c = lane & 15;

    // ---- staging addresses: wave w stages LDS rows [32w, 32w+32) of both tiles, 8 rows per DMA
    const int srow = lane >> 3, sslot = lane & 7;
    const char* xsrc[4];
    const char* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = 32 * wave + 8 * i + srow;
        const int ch = sslot ^ swz_f(R);
        xsrc[i] = reinterpret_cast<const char*>(X + (size_t)(m0 + R) * K) + ch * 16;
        wsrc[i] = reinterpret_cast<const char*>(W + (size_t)(n0 + perm_w(R)) * K) + ch * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* xb = smem + buf * BUF_BYTES + (32 * wave) * 128;
        char* wb = xb + TILE_BYTES;
        const int koff = kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + koff),
                                             (__attribute__((address_space(3))) void*)(xb + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + koff),
                                             (__attribute__((address_space(3))) void*)(wb + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a tile), per kk = 0,1
    int xoff[2], woff[2];
    {
        const int xr = 64 * wm + c, wr = 64 * wn + c;   // + 16*mi / 16*ni: same swizzle (f depends on (row>>1)&7)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            xoff[kk] = xr * 128 + (((4 * kk + g) ^ swz_f(xr)) << 4);
            woff[kk] = wr * 128 + (((4 * kk + g) ^ swz_f(wr)) << 4);
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* xt = smem + cur * BUF_BYTES;
        const char* wt = xt + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // rows +16*i keep (row>>1)&7 only if 16*i>>1 = 8i is 0 mod 8: yes
                xf[i] = *reinterpret_cast<const bf16x8*>(xt + xoff[kk] + i * 16 * 128);
                wf[i] = *reinterpret_cast<const bf16x8*>(wt + woff[kk] + i * 16 * 128);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        // next tile landed (own pieces) + everyone finished reading `cur`
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue: lane (g,c) owns rows m = m0 + 64*wm + 16*mi + c, columns n0 + 64*wn + 32*p + 8*g + 0..7
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int n = n0 + 64 * wn + 32 * p + 8 * g;
        f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (bias) {
            b0 = *reinterpret_cast<const f32x4*>(bias + n);
            b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + 64 * wm + 16 * mi + c;
            if (m >= M) continue;
            f32x4 v0 = acc[2 * p][mi] + b0;
            f32x4 v1 = acc[2 * p + 1][mi] + b1;
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
            if constexpr (EPI == KEDS_EPI_BIAS_BF16 || EPI == KEDS_EPI_BIAS_QGELU_BF16 ||
                          EPI == KEDS_EPI_BIAS_RELU_BF16) {
                bf16x8 o = bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                                  (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + (size_t)m * N + n) = o;
            } else if constexpr (EPI == KEDS_EPI_BIAS_RESID_F32) {
                float* o = reinterpret_cast<float*>(out) + (size_t)m * N + n;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(o);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(o + 4);
                *reinterpret_cast<f32x4*>(o) = r0 + v0;
                *reinterpret_cast<f32x4*>(o + 4) = r1 + v1;
            } else if constexpr (EPI == KEDS_EPI_BIAS_F32) {
                float* o = reinterpret_cast<float*>(out) + (size_t)m * N + n;
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
    }
}

template <int EPI>
int launch_gemm(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, const float* aux,
                int aux_i, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_bt_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GEMM_LDS) != hipSuccess) {
            keds_set_error("gemm: cannot set dynamic LDS size");
            return KEDS_E_LAUNCH;
        }
        attr_set = true;
    }
    const int m_tiles = (M + BM - 1) / BM, n_tiles = N / BN;
    KedsProfScope prof(KEDS_PROF_GEMM, st);
    gemm_bt_kernel<EPI><<<m_tiles * n_tiles, 256, GEMM_LDS, st>>>((const bf16_t*)A, (const bf16_t*)W, bias, out, M, N, K,
                                                                   n_tiles, aux, aux_i);
    return keds_check_launch("gemm_bt_kernel");
}

}  // namespace

extern "C" int keds_gemm_bt(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                            int epilogue, const float* aux, int aux_i, void* stream) {
    KEDS_REQUIRE(A && W && out, "keds_gemm_bt: null pointer");
    KEDS_REQUIRE(M > 0 && N > 0 && K > 0, "keds_gemm_bt: empty problem");
    KEDS_REQUIRE(N % BN == 0, "keds_gemm_bt: N=%d must be a multiple of %d", N, BN);
    KEDS_REQUIRE(K % BK == 0, "keds_gemm_bt: K=%d must be a multiple of %d", K, BK);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case KEDS_EPI_BIAS_BF16: return launch_gemm<KEDS_EPI_BIAS_BF16>(A, W, bias, out, M, N, K, aux, aux_i, st);
        case KEDS_EPI_BIAS_QGELU_BF16:
            return launch_gemm<KEDS_EPI_BIAS_QGELU_BF16>(A, W, bias, out, M, N, K, aux, aux_i, st);
        case KEDS_EPI_BIAS_RELU_BF16:
            return launch_gemm<KEDS_EPI_BIAS_RELU_BF16>(A, W, bias, out, M, N, K, aux, aux_i, st);
        case KEDS_EPI_BIAS_RESID_F32:
            return launch_gemm<KEDS_EPI_BIAS_RESID_F32>(A, W, bias, out, M, N, K, aux, aux_i, st);
        case KEDS_EPI_BIAS_F32: return launch_gemm<KEDS_EPI_BIAS_F32>(A, W, bias, out, M, N, K, aux, aux_i, st);
        case KEDS_EPI_PATCH_F32:
            KEDS_REQUIRE(aux && aux_i > 0, "keds_gemm_bt: EPI_PATCH needs the positional embedding and G");
            return launch_gemm<KEDS_EPI_PATCH_F32>(A, W, bias, out, M, N, K, aux, aux_i, st);
        default: keds_set_error("keds_gemm_bt: unknown epilogue %d", epilogue); return KEDS_E_ARG;
    }
}
