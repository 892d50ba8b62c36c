// The fp32-ACCURATE operating point of the encoder path (CLIP.set_precision("fp32"), keds_tower_params.f32 = 1).
//
// The reference evaluates in fp32 (src/eval_retrieval.py:108-109, `--precision` src/params.py:227-232: the default `amp`
// converts the model to fp32 and no autocast is entered in eval), and north_star asks for Recall@k EQUAL to that path.  The
// default flow of this library rounds GEMM operands to bf16 / fp16 and differs from the reference by a
// few 1e-3 in the embeddings -- enough to flip a (query, k) outcome the reference itself decides by a margin below 5e-4.  This
// file is the flow without operand rounding: every product runs on the f32-input matrix instruction
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate = an fmaf chain, cdna_hip_programming.md section 3), the
// residual stream, the LayerNorm output, q / k / v, the attention probabilities and the MLP hidden layer all stay fp32.
// Throughput is secondary here (the f32 matrix rate is 1/16 of the bf16 one): simple tiles, no side lane, no fused LayerNorm.
//
//   gemm_f32_kernel       out[M,N] = epilogue(X[M,K] . W[N,K]^T + bias)      model.py:309-326 (in/out projections, MLP), :381 (conv1)
//   attention_f32_kernel  softmax(q k^T / 8 [+ causal mask]) v per (sample, head) model.py:319-321, nn.MultiheadAttention
//   im2col_f32            patch rows of the conv1-as-GEMM                     model.py:394-398
#include "keds_common.h"
#include <math.h>

int keds_layernorm_impl(const float* x, long long x_stride, const int* row_map, int row_mul, const float* gamma,
                        const float* beta, void* out, int out_f32, int rows, int dim, hipStream_t st);

int keds_attention_x3_impl(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal, int q_limit,
                           int* overflow, const int32_t* seq_off, void* stream);
extern "C" int keds_attention_x3(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal,
                                 int q_limit, int* overflow, void* stream);

namespace {

namespace g32 {
constexpr int TM = 128, TN = 128, TK = 16;
constexpr int LD = TM + 4;            // k-major LDS image [TK][LD]: fragment reads are consecutive dwords (conflict free)
}  // namespace g32

enum { F32_EPI_BIAS = 0, F32_EPI_QGELU = 1, F32_EPI_RESID = 2, F32_EPI_RELU = 3, F32_EPI_PATCH = 4 };

// 128 x 128 x 16 tile, 4 waves (2 x 2), each wave 64 (n) x 64 (m) = 2 x 2 tiles of v_mfma_f32_32x32x2_f32.  W is the MFMA A
// operand and X the B operand (as in gemm.hip): the accumulator then holds, per lane, ONE output row m and runs of four
// consecutive n -> 16-byte stores.  Operand tiles go HBM -> registers -> LDS transposed to k-major ([k][row]: lane l of an
// MFMA reads element (k = 2 kk + (l >> 5), row = l & 31), i.e. 32 consecutive dwords per half wave), register-prefetched
// one K-tile ahead, two LDS buffers, one barrier per K-tile.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, long long lda, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out, long long ldc,
                                                       int M, int N, int K, const float* __restrict__ aux, int aux_i) {
    using namespace g32;
    __shared__ float sX[2][TK * LD];
    __shared__ float sW[2][TK * LD];
    const int n_tiles = N / TN;
    const int tm = blockIdx.x / n_tiles, tn = blockIdx.x - tm * n_tiles;       // n fastest: neighbours share their X rows
    const int m0 = tm * TM, n0 = tn * TN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = t >> 2, kq = t & 3;                                           // staging: rows r, r + 64; k quarter kq
    const float* xp[2];
    const float* wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + r + 64 * i;
        m = m < M ? m : M - 1;                                                  // ragged last row tile: rows clamped, not stored
        xp[i] = X + (size_t)m * lda + 4 * kq;
        wp[i] = W + (size_t)(n0 + r + 64 * i) * K + 4 * kq;
    }
    f32x4 xr[2], wr[2];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xr[i] = *reinterpret_cast<const f32x4*>(xp[i] + kt * TK);
            wr[i] = *reinterpret_cast<const f32x4*>(wp[i] + kt * TK);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sX[buf][(4 * kq + j) * LD + r + 64 * i] = xr[i][j];
                sW[buf][(4 * kq + j) * LD + r + 64 * i] = wr[i][j];
            }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / TK;
    const int h = lane >> 5, c = lane & 31;
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load(kt + 1);
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            const float* ws = sW[buf] + (2 * kk + h) * LD + 64 * wn + c;
            const float* xs = sX[buf] + (2 * kk + h) * LD + 64 * wm + c;
            const float a0 = ws[0], a1 = ws[32], b0 = xs[0], b1 = xs[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store(buf ^ 1);
        __syncthreads();
    }
    // C/D map of the 32 x 32 forms: column (here: m) = lane & 31, row (here: n) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 64 * wm + 32 * j + c;
        if (m >= M) continue;
        size_t orow = (size_t)m;
        const float* pos = nullptr;
        if constexpr (EPI == F32_EPI_PATCH) {                                    // token row (m / G) (G + 1) + 1 + m % G, + positional embedding
            const int G = aux_i, b = m / G, pidx = m - b * G;
            orow = (size_t)b * (G + 1) + 1 + pidx;
            pos = aux + (size_t)(1 + pidx) * N;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 64 * wn + 32 * i + 8 * g + 4 * h;
                f32x4 v = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (bias) v = v + *reinterpret_cast<const f32x4*>(bias + n);
                float* o = out + orow * ldc + n;
                if constexpr (EPI == F32_EPI_QGELU) {                            // x * sigmoid(1.702 x), model.py:300-302
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-1.702f * v[e]));
                }
                if constexpr (EPI == F32_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (EPI == F32_EPI_RESID) v = v + *reinterpret_cast<const f32x4*>(o);
                if constexpr (EPI == F32_EPI_PATCH) v = v + *reinterpret_cast<const f32x4*>(pos + n);
                *reinterpret_cast<f32x4*>(o) = v;
            }
    }
}

// One workgroup (9 waves) per (sample, head), both products on the f32-input matrix instruction.  K (row stride 65 floats:
// lane i reads key row i conflict free) and V live in LDS as fp32, zero padded to whole 32-key tiles; a wave owns 32-query
// tiles qt = wave, wave + 9, ... (S = 257: nine tiles, one per wave) and walks the key tiles with an online softmax:
//   S^T tile  = K tile . Q^T   32 MFMAs 32x32x2: A[i = key][k] = K[key][2 kk + h], B[k][j = query] = q[query][2 kk + h] / 8
//               (the lane's 32 q values sit in registers); result: lane = query j, register r = key (r & 3) + 8 (r >> 2) + 4 h
//   p = exp(s - m_new), running maximum m (combined over the two lane halves), running sum l (per half, combined at the end),
//               O rescaled by exp(m - m_new)
//   O^T tile += V^T tile . P^T  16 + 16 MFMAs: step r takes the key PAIR {key_0(r), key_1(r)} the two lane halves hold in
//               register r -- B[k = h][j] is the lane's own p[r], A[i = dim][k = h] = V[key_h(r)][dim] -- so the probabilities
//               go from the score accumulator into the next product without leaving their registers
// expf / division in full precision.  (The first version kept the scores of one key per lane and broadcast with v_readlane:
// 4.4 ms per ViT-L/14 layer at B = 128, 37 % of the fp32 step; this one ~0.4 ms.)  S <= 288.
constexpr int A32_WAVES = 9;
__global__ __launch_bounds__(64 * A32_WAVES) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S,
                                                                        int heads, int causal, int q_limit,
                                                                        const int* __restrict__ seq_off) {
    extern __shared__ __attribute__((aligned(16))) float a32_lds[];
    // packed rows (towers.hip PackedRows): sample b owns rows [seq_off[b], seq_off[b + 1]); S (the longest sample: it sized the LDS)
    // becomes the sample's own length
    long long row_base = (long long)(blockIdx.x / heads) * S;
    if (seq_off) {
        row_base = seq_off[blockIdx.x / heads];
        S = seq_off[blockIdx.x / heads + 1] - (int)row_base;
        q_limit = q_limit < S ? q_limit : S;
    }
    const int nkt = (S + 31) >> 5, SP = nkt * 32;
    float* Ks = a32_lds;                           // [SP][65]
    float* Vs = a32_lds + (size_t)SP * 65;         // [SP][64]
    const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
    const int d = heads * 64, ld = 3 * d;
    const float* base = qkv + (size_t)row_base * ld + hd * 64;
    for (int idx = threadIdx.x; idx < SP * 16; idx += 64 * A32_WAVES) {
        const int row = idx >> 4, c4 = idx & 15;
        f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (row < S) {
            kv = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + d + 4 * c4);
            vv = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 2 * d + 4 * c4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Ks[row * 65 + 4 * c4 + e] = kv[e];
            Vs[row * 64 + 4 * c4 + e] = vv[e];
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 31, hh = lane >> 5;
    const int nq = q_limit < S ? q_limit : S;
    const int nqt = (nq + 31) >> 5;
    for (int qt = wave; qt < nqt; qt += A32_WAVES) {
        const int q = qt * 32 + j;
        const float* qrow = base + (size_t)(q < S ? q : S - 1) * ld + hh;
        float qreg[32];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) qreg[kk] = qrow[2 * kk] * 0.125f;          // 1 / sqrt(64): exact scaling
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
            const float* kr = Ks + (size_t)(kt * 32 + j) * 65 + hh;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * kk], qreg[kk], sc, 0, 0, 0);
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool valid = key < S && (!causal || key <= q);
                sc[r] = valid ? sc[r] : -INFINITY;
                tmax = fmaxf(tmax, sc[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(m, tmax);
            const float resc = m == -INFINITY ? 0.f : expf(m - mnew);           // (first tile, or nothing valid so far)
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = mnew == -INFINITY ? 0.f : expf(sc[r] - mnew);           // masked keys: exp(-inf) = 0
                psum += sc[r];
            }
            l = l * resc + psum;
#pragma unroll
            for (int e = 0; e < 16; ++e) o0[e] *= resc, o1[e] *= resc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vr = Vs + (size_t)(kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 64 + j;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], sc[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], sc[r], o1, 0, 0, 0);
            }
            m = mnew;
        }
        const float ltot = l + __shfl_xor(l, 32, 64);
        if (q < nq) {                                                            // lane = query q; register r = dim (r & 3) + 8 (r >> 2) + 4 h
            float* orow = out + ((size_t)row_base + q) * d + hd * 64 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(orow + 8 * g) = f32x4{o0[4 * g] / ltot, o0[4 * g + 1] / ltot, o0[4 * g + 2] / ltot, o0[4 * g + 3] / ltot};
                *reinterpret_cast<f32x4*>(orow + 32 + 8 * g) = f32x4{o1[4 * g] / ltot, o1[4 * g + 1] / ltot, o1[4 * g + 2] / ltot, o1[4 * g + 3] / ltot};
            }
        }
    }
}

// one block per output row (b, patch); columns c*P*P + ky*P + kx, zero padded to Kpad (elementwise.hip's im2col in fp32)
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int R, int P,
                                                         int Kpad) {
    const int g = R / P;
    const int row = blockIdx.x;
    const int b = row / (g * g), pi = row % (g * g);
    const int py = pi / g, px = pi % g;
    const int kreal = 3 * P * P;
    const float* base = img + (size_t)b * 3 * R * R;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        float v = 0.f;
        if (k < kreal) {
            const int ch = k / (P * P), rem = k % (P * P);
            const int ky = rem / P, kx = rem % P;
            v = base[(size_t)ch * R * R + (size_t)(py * P + ky) * R + px * P + kx];
        }
        out[(size_t)row * Kpad + k] = v;
    }
}

__global__ void l2norm_rows_f32_kernel(float* __restrict__ x, int rows, int dim) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* p = x + (size_t)r * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s = fmaf(p[i], p[i], s);
    s = wave_sum(s);
    const float nrm = sqrtf(s);                   // x / x.norm(dim=-1, keepdim=True), eval_utils.py:162,704-710
    for (int i = lane; i < dim; i += 64) p[i] = p[i] / nrm;
}

template <int EPI>
int launch_gemm_f32(const float* A, long long lda, const float* W, const float* bias, float* out, long long ldc, int M, int N,
                    int K, const float* aux, int aux_i, hipStream_t st) {
    const int tiles = ((M + g32::TM - 1) / g32::TM) * (N / g32::TN);
    KedsProfScope prof(KEDS_PROF_GEMM, st);
    prof.work(2.0 * M * N * (double)K);
    gemm_f32_kernel<EPI><<<tiles, 256, 0, st>>>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i);
    return keds_check_launch("gemm_f32_kernel");
}

}  // namespace

extern "C" int keds_gemm_f32(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M,
                             int N, int K, int epilogue, const float* aux, int aux_i, void* stream) {
    KEDS_REQUIRE(A && W && out && M > 0, "keds_gemm_f32: bad argument");
    KEDS_REQUIRE(N > 0 && N % 128 == 0 && K > 0 && K % 16 == 0, "keds_gemm_f32: N must be a multiple of 128 and K of 16 (got N=%d K=%d)", N, K);
    KEDS_REQUIRE(lda >= K && lda % 4 == 0 && ldc % 4 == 0, "keds_gemm_f32: row strides must be multiples of 4 floats");
    KEDS_REQUIRE(epilogue != F32_EPI_PATCH || (aux && aux_i > 0), "keds_gemm_f32: the patch epilogue needs the positional embedding and G");
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case F32_EPI_BIAS: return launch_gemm_f32<F32_EPI_BIAS>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i, st);
        case F32_EPI_QGELU: return launch_gemm_f32<F32_EPI_QGELU>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i, st);
        case F32_EPI_RESID: return launch_gemm_f32<F32_EPI_RESID>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i, st);
        case F32_EPI_RELU: return launch_gemm_f32<F32_EPI_RELU>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i, st);
        case F32_EPI_PATCH: return launch_gemm_f32<F32_EPI_PATCH>(A, lda, W, bias, out, ldc, M, N, K, aux, aux_i, st);
    }
    keds_set_error("keds_gemm_f32: unknown epilogue %d", epilogue);
    return KEDS_E_ARG;
}

static int attention_f32_impl(const float* qkv, float* out, int B, int S, int heads, int causal, int q_limit, const int32_t* seq_off,
                              void* stream);
extern "C" int keds_attention_f32(const float* qkv, float* out, int B, int S, int heads, int causal, int q_limit, void* stream) {
    return attention_f32_impl(qkv, out, B, S, heads, causal, q_limit, nullptr, stream);
}
static int attention_f32_impl(const float* qkv, float* out, int B, int S, int heads, int causal, int q_limit, const int32_t* seq_off,
                              void* stream) {
    KEDS_REQUIRE(qkv && out && B > 0 && heads > 0, "keds_attention_f32: bad argument");
    KEDS_REQUIRE(S >= 1 && S <= 288, "keds_attention_f32: S must be in [1, 288] (got %d)", S);
    const int lds = ((S + 31) / 32 * 32) * (65 + 64) * (int)sizeof(float);           // K and V padded to whole 32-key tiles
    int rc = keds_func_lds_once((const void*)attention_f32_kernel, lds, "attention_f32_kernel");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_ATTN, st);
    attention_f32_kernel<<<B * heads, 64 * A32_WAVES, lds, st>>>(qkv, out, S, heads, causal, q_limit > 0 ? q_limit : S, seq_off);
    return keds_check_launch("attention_f32_kernel");
}

extern "C" int keds_im2col_f32(const float* image, float* out, int B, int R, int P, int Kpad, void* stream) {
    KEDS_REQUIRE(image && out && B > 0 && P > 0 && R % P == 0, "keds_im2col_f32: bad argument");
    KEDS_REQUIRE(Kpad >= 3 * P * P && Kpad % 16 == 0, "keds_im2col_f32: Kpad must cover 3*P*P and be a multiple of 16");
    const int g = R / P;
    KedsProfScope prof(KEDS_PROF_OTHER, (hipStream_t)stream);
    im2col_f32_kernel<<<B * g * g, 256, 0, (hipStream_t)stream>>>(image, out, R, P, Kpad);
    return keds_check_launch("im2col_f32_kernel");
}

// ---- whole towers ----------------------------------------------------------------------------------------------------
// scratch of one fp32 tower: ln [M,w] | qkv [M,3w] | att [M,w] | hid [M,4w], all fp32 (1.2 GB at B = 128, ViT-L/14)
size_t keds_tower_f32_workspace_bytes(int width, int seq, int B) {
    const size_t M = keds_align_up((size_t)B * seq, 256);
    return keds_align_up(M * width * 4, 256) + keds_align_up(M * (size_t)width * 3 * 4, 256) + keds_align_up(M * (size_t)width * 4, 256) +
           keds_align_up(M * (size_t)width * 4 * 4, 256);
}

int keds_gather_rows_impl(const void* src, void* dst, const int32_t* row, int S, int B, int dim, int mode, hipStream_t st,
                          bool global_rows = false);

int keds_layernorm_pair_impl(const float* x, long long x_stride, const float* gamma, const float* beta, void* out, long long plane,
                             int rows, int dim, hipStream_t st);
int* keds_numerics_guard();
bool keds_gemm_splits_rows(int M, int N, int K);

// keds_tower_params.f32 == 2: the "fp32x3" operating point (round 5).  The same fp32 flow -- fp32 residual stream, fp32
// LayerNorm / attention / QuickGELU -- with the four block GEMMs on SPLIT fp16 operands (keds_gemm_x3: x = hi + lo to 22 bits,
// products hi.hi + hi.lo + lo.hi on the fp16 MFMA, fp32 accumulate): fp32-grade results at about a third of the bf16 GEMMs'
// rate instead of the f32-input MFMA's sixteenth.  The weights arrive as fp16 planes [2][N][K] (keds_split_f16_pair at packing
// time); LayerNorm writes its output as planes, the attention output is split by a pass of its own, c_fc's epilogue writes the
// MLP hidden layer as planes.  A value beyond the fp16 range raises the caller's numerics guard (the host falls back to f32 = 1).
static int tower_forward_x3(const keds_tower_params* p, float* x, int B, void* ws, hipStream_t st, const int32_t* last_rows,
                            const PackedRows* pk) {
    const int w = p->width, S = p->seq, M = pk ? pk->rows : B * S;
    const size_t Mp = keds_align_up((size_t)B * S, 256);          // (the buffers are carved for the rectangular layout either way)
    const int32_t* soff = pk ? pk->off : nullptr;
    const int gbound = pk ? pk->valid : S;
    char* base = (char*)ws;
    _Float16* ln2 = (_Float16*)base;                              // planes [2][Mp][w] (the fp32 flow's ln buffer: same bytes)
    float* qkv = (float*)(base + keds_align_up(Mp * w * 4, 256));
    float* att = (float*)((char*)qkv + keds_align_up(Mp * (size_t)w * 3 * 4, 256));
    _Float16* hid2 = (_Float16*)((char*)att + keds_align_up(Mp * (size_t)w * 4, 256));      // planes [2][Mp][4w]
    const long long pl = (long long)Mp * w, plh = (long long)Mp * 4 * w;
    int* guard = keds_numerics_guard();
    int rc;
    const long long wq = 3LL * w * w, wo = (long long)w * w, wf = 4LL * w * w;         // elements between a weight's planes
    // Two lanes, as in the default flow (towers.hip, RowLanes): the rows beyond the last full 256-row tile (ViT-L/14 at B = 128:
    // 128 of 32,896) are a handful of workgroups per GEMM whose time is pure latency -- 150 us per block when they run between
    // the full-tile launches.  Rows only meet in the attention: the remainder rows' chain (out-proj, ln_2, MLP, next ln_1, next
    // in_proj) runs on the side lane beside the same chain on the full tiles; fork behind the attention, join in front of the next.
    struct Span {
        size_t r0;
        int n;
        hipStream_t st;
    };
    const int m_main = M / 256 * 256;
    KedsSideLane* lane = (m_main > 0 && m_main < M && keds_gemm_splits_rows(M, 3 * w, w)) ? keds_side_lane() : nullptr;
    Span spans[2] = {{0, lane ? m_main : M, st}, {(size_t)m_main, M - m_main, lane ? lane->s : st}};
    const int nspan = lane ? 2 : 1;
    auto pre = [&](const keds_block_params& k, const Span& sp) -> int {               // ln_1 + in_proj
        int r = keds_layernorm_pair_impl(x + sp.r0 * w, w, k.ln1_g, k.ln1_b, ln2 + sp.r0 * w, pl, sp.n, w, sp.st);
        if (r) return r;
        return keds_gemm_x3(ln2 + sp.r0 * w, pl, w, k.qkv_w, wq, k.qkv_b, qkv + sp.r0 * 3 * w, 3 * w, sp.n, 3 * w, w, KEDS_EPI_X3_BIAS_F32, 0, k.x3_exp[0], sp.st);
    };
    auto post = [&](const keds_block_params& k, const Span& sp) -> int {              // out-proj + ln_2 + MLP
        int r;
        if ((r = keds_gemm_x3(ln2 + sp.r0 * w, pl, w, k.out_w, wo, k.out_b, x + sp.r0 * w, w, sp.n, w, w, KEDS_EPI_X3_RESID_F32, 0, k.x3_exp[1], sp.st))) return r;
        if ((r = keds_layernorm_pair_impl(x + sp.r0 * w, w, k.ln2_g, k.ln2_b, ln2 + sp.r0 * w, pl, sp.n, w, sp.st))) return r;
        if ((r = keds_gemm_x3(ln2 + sp.r0 * w, pl, w, k.fc_w, wf, k.fc_b, hid2 + sp.r0 * 4 * w, 4 * w, sp.n, 4 * w, w, KEDS_EPI_X3_QGELU_PAIR, (int)plh, k.x3_exp[2], sp.st))) return r;
        return keds_gemm_x3(hid2 + sp.r0 * 4 * w, plh, 4 * w, k.proj_w, wf, k.proj_b, x + sp.r0 * w, w, sp.n, w, 4 * w, KEDS_EPI_X3_RESID_F32, 0, k.x3_exp[3], sp.st);
    };
    if (lane && (rc = keds_stream_order(st, lane->fork, lane->s))) return rc;
    for (int i = 0; i < nspan; ++i)
        if ((rc = pre(p->blocks[0], spans[i]))) return rc;
    for (int l = 0; l < p->layers; ++l) {
        const keds_block_params& k = p->blocks[l];
        const bool last = l == p->layers - 1;
        if (lane && (rc = keds_stream_order(lane->s, lane->join, st))) return rc;      // every row's q, k, v before the attention
        if (last && (last_rows || p->last_cls_only)) {
            // after the last block one row per sample is read (model.py:412 the CLS row; :587-589, 847-849 the read-out row):
            // attention (all rows for the text tower, the CLS query for the ViT), then the B rows in compact buffers
            float* att_c = qkv;                                   // [B, w] fp32 each, in the qkv buffer (dead after the attention)
            float* x_c = qkv + (size_t)B * w;
            if (last_rows) {
                if ((rc = keds_attention_x3_impl(qkv, att, nullptr, 0, B, S, p->heads, p->causal, S, guard, soff, st))) return rc;
                if ((rc = keds_gather_rows_impl(att, att_c, last_rows, gbound, B, w, 2, st, pk != nullptr))) return rc;
                if ((rc = keds_gather_rows_impl(x, x_c, last_rows, gbound, B, w, 2, st, pk != nullptr))) return rc;
            } else {
                if ((rc = keds_attention_x3(qkv, att, nullptr, 0, B, S, p->heads, p->causal, 1, guard, st))) return rc;
                if (hipMemcpy2DAsync(att_c, (size_t)w * 4, att, (size_t)S * w * 4, (size_t)w * 4, B, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                    hipMemcpy2DAsync(x_c, (size_t)w * 4, x, (size_t)S * w * 4, (size_t)w * 4, B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                    keds_set_error("keds_tower_forward_f32: CLS rows: %s", hipGetErrorString(hipGetLastError()));
                    return KEDS_E_LAUNCH;
                }
            }
            const long long pc = (long long)keds_align_up((size_t)B, 256) * w;
            if ((rc = keds_split_f16_pair(att_c, w, B, w, ln2, pc, guard, st))) return rc;
            if ((rc = keds_gemm_x3(ln2, pc, w, k.out_w, wo, k.out_b, x_c, w, B, w, w, KEDS_EPI_X3_RESID_F32, 0, k.x3_exp[1], st))) return rc;
            if ((rc = keds_layernorm_pair_impl(x_c, w, k.ln2_g, k.ln2_b, ln2, pc, B, w, st))) return rc;
            if ((rc = keds_gemm_x3(ln2, pc, w, k.fc_w, wf, k.fc_b, hid2, 4 * w, B, 4 * w, w, KEDS_EPI_X3_QGELU_PAIR, (int)(4 * pc), k.x3_exp[2], st))) return rc;
            if ((rc = keds_gemm_x3(hid2, 4 * pc, 4 * w, k.proj_w, wf, k.proj_b, x_c, w, B, w, 4 * w, KEDS_EPI_X3_RESID_F32, 0, k.x3_exp[3], st))) return rc;
            // the caller reads row b of a compact x (text tower) or row b * S (ViT: CLS rows in place)
            if (hipMemcpy2DAsync(x, (size_t)(last_rows ? w : S * w) * 4, x_c, (size_t)w * 4, (size_t)w * 4, B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                keds_set_error("keds_tower_forward_f32: read-out rows: %s", hipGetErrorString(hipGetLastError()));
                return KEDS_E_LAUNCH;
            }
            return KEDS_OK;
        }
        // (the attention writes the out-projection's A operand planes itself: no fp32 copy of its output, no split pass)
        if ((rc = keds_attention_x3_impl(qkv, nullptr, ln2, pl, B, S, p->heads, p->causal, S, guard, soff, st))) return rc;
        if (lane && (rc = keds_stream_order(st, lane->fork, lane->s))) return rc;
        for (int i = 0; i < nspan; ++i) {
            if ((rc = post(k, spans[i]))) return rc;
            if (!last && (rc = pre(p->blocks[l + 1], spans[i]))) return rc;
        }
    }
    return lane ? keds_stream_order(lane->s, lane->join, st) : KEDS_OK;
}

// last_rows (device int32 [B], nullable): towers.hip, rows_tail -- the text tower's read-out rows; on return x[b] = that row
// pk (nullable): packed rows of a causal tower with a read-out row per sample (keds_common.h) -- M = pk->rows, last_rows global
int keds_tower_forward_f32(const keds_tower_params* p, float* x, int B, void* ws, hipStream_t st, const int32_t* last_rows,
                           const PackedRows* pk) {
    if (pk && (!p->causal || !last_rows)) {
        keds_set_error("keds_tower_forward_f32: packed rows need a causal tower with a read-out row per sample");
        return KEDS_E_ARG;
    }
    if (p->f32 == 2) return tower_forward_x3(p, x, B, ws, st, last_rows, pk);
    const int w = p->width, S = p->seq, M = pk ? pk->rows : B * S;
    const size_t Mp = keds_align_up((size_t)B * S, 256);          // (the buffers are carved for the rectangular layout either way)
    const int32_t* soff = pk ? pk->off : nullptr;
    const int gbound = pk ? pk->valid : S;
    char* base = (char*)ws;
    float* ln = (float*)base;
    float* qkv = (float*)(base + keds_align_up(Mp * w * 4, 256));
    float* att = (float*)((char*)qkv + keds_align_up(Mp * (size_t)w * 3 * 4, 256));
    float* hid = (float*)((char*)att + keds_align_up(Mp * (size_t)w * 4, 256));
    int rc;
    // packed rows: the zero rows behind the last sample are nobody's queries -- their attention output is cleared once so that they
    // stay finite through the blocks
    if (pk && pk->rows > pk->valid &&
        hipMemsetAsync(att + (size_t)pk->valid * w, 0, (size_t)(pk->rows - pk->valid) * w * sizeof(float), st) != hipSuccess) {
        keds_set_error("keds_tower_forward_f32: packed rows: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    for (int l = 0; l < p->layers; ++l) {
        const keds_block_params& k = p->blocks[l];
        const bool last = l == p->layers - 1;
        const float *qkv_w = (const float*)k.qkv_w, *out_w = (const float*)k.out_w, *fc_w = (const float*)k.fc_w,
                    *proj_w = (const float*)k.proj_w;
        if ((rc = keds_layernorm_impl(x, w, nullptr, 1, k.ln1_g, k.ln1_b, ln, 1, M, w, st))) return rc;
        if ((rc = keds_gemm_f32(ln, w, qkv_w, k.qkv_b, qkv, 3 * w, M, 3 * w, w, F32_EPI_BIAS, nullptr, 0, st))) return rc;
        if (last && last_rows) {
            // after the last block only the read-out row of every sample is read (model.py:587-589, 847-849): attention on all
            // rows, then out-proj, ln_2 and the MLP on the B gathered rows (compact [B, w] buffers in the qkv buffer)
            if ((rc = attention_f32_impl(qkv, att, B, S, p->heads, p->causal, S, soff, st))) return rc;
            float* att_c = qkv;
            float* x_c = qkv + (size_t)B * w;
            if ((rc = keds_gather_rows_impl(att, att_c, last_rows, gbound, B, w, 2, st, pk != nullptr))) return rc;
            if ((rc = keds_gather_rows_impl(x, x_c, last_rows, gbound, B, w, 2, st, pk != nullptr))) return rc;
            if ((rc = keds_gemm_f32(att_c, w, out_w, k.out_b, x_c, w, B, w, w, F32_EPI_RESID, nullptr, 0, st))) return rc;
            if ((rc = keds_layernorm_impl(x_c, w, nullptr, 1, k.ln2_g, k.ln2_b, ln, 1, B, w, st))) return rc;
            if ((rc = keds_gemm_f32(ln, w, fc_w, k.fc_b, hid, 4 * w, B, 4 * w, w, F32_EPI_QGELU, nullptr, 0, st))) return rc;
            if ((rc = keds_gemm_f32(hid, 4 * w, proj_w, k.proj_b, x_c, w, B, w, 4 * w, F32_EPI_RESID, nullptr, 0, st))) return rc;
            if (hipMemcpyAsync(x, x_c, (size_t)B * w * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
                keds_set_error("keds_tower_forward_f32: read-out rows: %s", hipGetErrorString(hipGetLastError()));
                return KEDS_E_LAUNCH;
            }
            return KEDS_OK;
        }
        if (last && p->last_cls_only) {
            // after the last block only token 0 of every sample is read (model.py:412): its attention query, out-proj, ln_2
            // and MLP run on those B rows (row stride S*w in x / att)
            const long long ldr = (long long)S * w;
            if ((rc = keds_attention_f32(qkv, att, B, S, p->heads, p->causal, 1, st))) return rc;
            if ((rc = keds_gemm_f32(att, ldr, out_w, k.out_b, x, ldr, B, w, w, F32_EPI_RESID, nullptr, 0, st))) return rc;
            if ((rc = keds_layernorm_impl(x, w, nullptr, S, k.ln2_g, k.ln2_b, ln, 1, B, w, st))) return rc;
            if ((rc = keds_gemm_f32(ln, w, fc_w, k.fc_b, hid, 4 * w, B, 4 * w, w, F32_EPI_QGELU, nullptr, 0, st))) return rc;
            return keds_gemm_f32(hid, 4 * w, proj_w, k.proj_b, x, ldr, B, w, 4 * w, F32_EPI_RESID, nullptr, 0, st);
        }
        if ((rc = attention_f32_impl(qkv, att, B, S, p->heads, p->causal, S, soff, st))) return rc;
        if ((rc = keds_gemm_f32(att, w, out_w, k.out_b, x, w, M, w, w, F32_EPI_RESID, nullptr, 0, st))) return rc;
        if ((rc = keds_layernorm_impl(x, w, nullptr, 1, k.ln2_g, k.ln2_b, ln, 1, M, w, st))) return rc;
        if ((rc = keds_gemm_f32(ln, w, fc_w, k.fc_b, hid, 4 * w, M, 4 * w, w, F32_EPI_QGELU, nullptr, 0, st))) return rc;
        if ((rc = keds_gemm_f32(hid, 4 * w, proj_w, k.proj_b, x, w, M, w, 4 * w, F32_EPI_RESID, nullptr, 0, st))) return rc;
    }
    return KEDS_OK;
}

// ln_post / ln_final on the read-out rows + projection (+ L2 normalisation), fp32 (model.py:412-414, 586-589, 841-849)
size_t keds_readout_f32_workspace_bytes(int B, int d) { return keds_align_up((size_t)B, 128) * d * 4; }

int keds_readout_f32(const float* x, int S, const int32_t* row, const float* gamma, const float* beta, const float* proj_t,
                     float* out, int B, int d, int E, int normalize, void* workspace, hipStream_t st) {
    int rc = keds_layernorm_impl(x, d, row, S, gamma, beta, workspace, 1, B, d, st);
    if (rc) return rc;
    if ((rc = keds_gemm_f32((const float*)workspace, d, proj_t, nullptr, out, E, B, E, d, F32_EPI_BIAS, nullptr, 0, st))) return rc;
    if (normalize) {
        l2norm_rows_f32_kernel<<<(B + 3) / 4, 256, 0, st>>>(out, B, E);
        rc = keds_check_launch("l2norm_rows_f32_kernel");
    }
    return rc;
}

// ---- knowledge injection in fp32 (IM2TEXT + 2 x CrossFormer, model.py:37-123, eval_utils.py:661-672) ------------------
namespace {
// single-query cross-attention core (model.py:56-79) on projected fp32 rows: one wave per (sample, head), lane = dim of the
// 64-wide head; Q [B, inner], Kp / Vp [B*K, inner] -> out [B, inner]; K <= 64
__global__ __launch_bounds__(256) void cross_core_f32_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                             const float* __restrict__ Vp, float* __restrict__ out, int B, int K,
                                                             int heads) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= B * heads) return;
    const int b = wid / heads, h = wid - b * heads, inner = heads * 64;
    const float q = Q[(size_t)b * inner + h * 64 + lane] * 0.125f;           // dim_head ** -0.5, exact
    // one pass with a running maximum (K <= 64 keys: the rescales are exact to rounding, results within 1e-7 of the two-pass form)
    float mx = -INFINITY, sum = 0.f, o = 0.f;
    for (int j = 0; j < K; ++j) {
        const float sj = wave_sum(q * Kp[((size_t)b * K + j) * inner + h * 64 + lane]);
        if (sj > mx) {                                                       // (wave-uniform)
            const float sc = expf(mx - sj);                                  // 0 on the first key
            sum *= sc;
            o *= sc;
            mx = sj;
        }
        const float e = expf(sj - mx);
        sum += e;
        o = fmaf(e, Vp[((size_t)b * K + j) * inner + h * 64 + lane], o);
    }
    out[(size_t)b * inner + h * 64 + lane] = o / sum;
}

struct KnowF32Ws {
    float *x, *h1, *h2, *y, *qp, *kp, *vp, *core, *qt;
    size_t bytes;
};
KnowF32Ws carve_know_f32(const keds_knowledge_params* p, int B, int K, void* ws) {
    const size_t R = (size_t)B + 2 * (size_t)B * K;
    const int inner = p->fuse.heads * 64;
    KnowF32Ws w;
    char* base = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t n) {
        float* r = base ? (float*)(base + off) : nullptr;
        off += keds_align_up(n * sizeof(float), 256);
        return r;
    };
    w.x = take(R * p->i2t.dim_in);
    w.h1 = take(R * p->i2t.middle);
    w.h2 = take(R * p->i2t.middle);
    w.y = take(R * p->i2t.dim_out);
    w.qp = take((size_t)B * inner);
    w.kp = take((size_t)B * K * inner);
    w.vp = take((size_t)B * K * inner);
    w.core = take((size_t)B * inner);
    w.qt = take((size_t)B * p->fuse.dim);
    w.bytes = off;
    return w;
}

// q chained through the layers, k and v fixed (model.py:98-101); the last layer's output lands in `out` (row stride ldo)
int crossformer_f32(const keds_crossformer_params* p, const float* q, long long ldq, const float* kv, int B, int K, float* out,
                    long long ldo, const KnowF32Ws& w, hipStream_t st) {
    const int dim = p->dim, inner = p->heads * 64;
    int rc;
    for (int l = 0; l < p->layers; ++l) {
        const keds_cross_layer_params& c = p->layer[l];
        const bool last = l == p->layers - 1;
        if ((rc = keds_gemm_f32(q, ldq, (const float*)c.wq, c.bq, w.qp, inner, B, inner, dim, F32_EPI_BIAS, nullptr, 0, st))) return rc;
        if ((rc = keds_gemm_f32(kv, dim, (const float*)c.wk, c.bk, w.kp, inner, B * K, inner, dim, F32_EPI_BIAS, nullptr, 0, st))) return rc;
        if ((rc = keds_gemm_f32(kv, dim, (const float*)c.wv, c.bv, w.vp, inner, B * K, inner, dim, F32_EPI_BIAS, nullptr, 0, st))) return rc;
        cross_core_f32_kernel<<<(B * p->heads + 3) / 4, 256, 0, st>>>(w.qp, w.kp, w.vp, w.core, B, K, p->heads);
        if ((rc = keds_check_launch("cross_core_f32_kernel"))) return rc;
        float* dst = last ? out : w.qt;
        const long long ldd = last ? ldo : dim;
        if ((rc = keds_gemm_f32(w.core, inner, (const float*)c.wo, c.bo, dst, ldd, B, dim, inner, F32_EPI_BIAS, nullptr, 0, st))) return rc;
        q = w.qt;
        ldq = dim;
    }
    return KEDS_OK;
}
}  // namespace

extern "C" size_t keds_knowledge_f32_workspace_bytes(const keds_knowledge_params* p, int B, int K) {
    if (!p || B <= 0 || K <= 0) return 0;
    return carve_know_f32(p, B, K, nullptr).bytes;
}

// One stream of the knowledge injection with EVERY weight pointer of the params an fp32 array (unfused per-layer weights):
// tokens_out [B,3,dim] = [fuse(m, I, I), cond(m, T, T), m] with m = img2text(q), I / T = img2text(neighbours)
extern "C" int keds_knowledge_run_f32(const keds_knowledge_params* p, const float* q, const float* nbr_img, const float* nbr_txt,
                                      int B, int K, float* tokens_out, void* workspace, size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(p && q && nbr_img && nbr_txt && tokens_out && workspace && B > 0, "keds_knowledge_run_f32: bad argument");
    KEDS_REQUIRE(K >= 1 && K <= 64, "keds_knowledge_run_f32: K must be in [1, 64]");
    const keds_im2text_params& m = p->i2t;
    const int dim = p->fuse.dim;
    KEDS_REQUIRE(m.dim_in == dim && m.dim_out == dim && p->cond.dim == dim && p->cond.heads == p->fuse.heads && m.n_layer >= 1 &&
                     m.n_layer <= 4 && p->fuse.layer && p->cond.layer,
                 "keds_knowledge_run_f32: inconsistent module shapes");
    KnowF32Ws w = carve_know_f32(p, B, K, workspace);
    if (workspace_bytes < w.bytes) {
        keds_set_error("keds_knowledge_run_f32: workspace %zu < %zu", workspace_bytes, w.bytes);
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t BK = (size_t)B * K, R = B + 2 * BK;
    if (hipMemcpyAsync(w.x, q, (size_t)B * dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(w.x + (size_t)B * dim, nbr_img, BK * dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(w.x + ((size_t)B + BK) * dim, nbr_txt, BK * dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        keds_set_error("keds_knowledge_run_f32: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    int rc;
    const float* cur = w.x;
    int kdim = m.dim_in;
    for (int i = 0; i < m.n_layer; ++i) {                                   // (Linear, Dropout = identity, ReLU) x n_layer
        float* dst = (i & 1) ? w.h2 : w.h1;
        if ((rc = keds_gemm_f32(cur, kdim, (const float*)m.w[i], m.b[i], dst, m.middle, (int)R, m.middle, kdim, F32_EPI_RELU, nullptr, 0, st)))
            return rc;
        cur = dst;
        kdim = m.middle;
    }
    // fc_out: the mapped query rows go straight to token slot 2 (row stride 3 dim), the neighbours to y
    float* mapped = tokens_out + 2 * (size_t)dim;
    if ((rc = keds_gemm_f32(cur, m.middle, (const float*)m.out_w, m.out_b, mapped, 3LL * dim, B, dim, m.middle, F32_EPI_BIAS, nullptr, 0, st)))
        return rc;
    if ((rc = keds_gemm_f32(cur + (size_t)B * m.middle, m.middle, (const float*)m.out_w, m.out_b, w.y, dim, (int)(2 * BK), dim, m.middle,
                            F32_EPI_BIAS, nullptr, 0, st)))
        return rc;
    if ((rc = crossformer_f32(&p->fuse, mapped, 3LL * dim, w.y, B, K, tokens_out, 3LL * dim, w, st))) return rc;
    return crossformer_f32(&p->cond, mapped, 3LL * dim, w.y + BK * dim, B, K, tokens_out + dim, 3LL * dim, w, st);
}
