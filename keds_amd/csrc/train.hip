// Training step of the knowledge-injection modules: forward in training mode and backward of
// IM2TEXT + 2 x CrossFormer through the FROZEN text tower, the symmetric contrastive loss and AdamW
// (reference: src/trainer.py:44-165 get_loss_img2text_image, src/main.py:215-237 optimizer groups,
//  src/model/model.py:37-123 the modules, :305-326 the blocks the gradient crosses, :808-851 the token splice).
//
// Every matrix product of the backward pass runs on the MFMA GEMM of gemm.hip (out = X . W^T): dX = dY . W needs W^T as
// the "W" operand, dW = dY^T . X needs dY^T and X^T as the operands -- keds_transpose_to_bf16 makes those copies (weights
// once per step, activations as they are produced).  What is left are the element / row / head kernels below.  They are
// stateless like the rest of keds_hip.h (device pointers in, nothing allocated); the step is sequenced by the host
// (keds_amd/train.py), as the reference's is.
#include "keds_common.h"
#include <math.h>

namespace {

// ---- transposes ---------------------------------------------------------------------------------------------------
// out[c][r] = bf16(src[r][c]) for r < rows, 0 for rows <= r < ld_out   (32 x 32 tiles through LDS)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, long long ld_src, int rows, int cols,
                                                        bf16_t* __restrict__ out, int ld_out) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? (float)src[(size_t)r * ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < ld_out) out[(size_t)c * ld_out + r] = (bf16_t)tile[tx][ty + 8 * i];
    }
}

// ---- column sums (bias gradients): out[c] (+)= sum_r x[r][c]; one block per 64 columns, fixed order -> reproducible
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long long ld, int rows, int cols,
                                                     float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = w; r < rows; r += 4) s += (float)x[(size_t)r * ld + c];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---- IM2TEXT hidden layer: y = relu(dropout(z))  (model.py:112-116: Linear, Dropout, ReLU) ---------------------------
__global__ void dropout_relu_fwd_kernel(const bf16_t* __restrict__ z, const unsigned char* __restrict__ mask, float scale,
                                        bf16_t* __restrict__ y, long long n) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    float v = (float)z[i];
    if (mask) v = mask[i] ? v * scale : 0.f;
    y[i] = (bf16_t)fmaxf(v, 0.f);
}
template <typename G>
__global__ void dropout_relu_bwd_kernel(const G* __restrict__ dy, const bf16_t* __restrict__ z,
                                        const unsigned char* __restrict__ mask, float scale, bf16_t* __restrict__ dz,
                                        long long n) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const bool on = (float)z[i] > 0.f && (!mask || mask[i]);
    dz[i] = (bf16_t)(on ? (float)dy[i] * (mask ? scale : 1.f) : 0.f);
}
// counter-based mask: keep iff hash(seed, i) >= p * 2^32 (the host can pass its own mask instead: parity tests do)
__global__ void dropout_mask_kernel(unsigned char* __restrict__ mask, long long n, unsigned long long seed, float p) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    mask[i] = (float)(unsigned)(x >> 32) * (1.0f / 4294967296.0f) >= p ? 1 : 0;
}

// ---- QuickGELU on its own (the training forward keeps the pre-activation; model.py:300-302) ------------------------------
__global__ void qgelu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long long n) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const float x = (float)u[i];
    y[i] = (bf16_t)(x / (1.0f + __expf(-1.702f * x)));
}
__global__ void qgelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ du,
                                 long long n) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const float x = (float)u[i];
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
    du[i] = (bf16_t)((float)dy[i] * (s + 1.702f * x * s * (1.0f - s)));
}

// ---- LayerNorm, training form: one wave per row; stats = {mean, rstd} ------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_stats_kernel(const float* __restrict__ x, long long ld, const int* __restrict__ rowmap,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ y, float* __restrict__ stats, int rows, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)(rowmap ? rowmap[row] : row) * ld;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += p[i];
    const float mean = wave_sum(s) / dim;
    float v = 0.f;
    for (int i = lane; i < dim; i += 64) v += (p[i] - mean) * (p[i] - mean);
    const float rstd = rsqrtf(wave_sum(v) / dim + 1e-5f);
    for (int i = lane; i < dim; i += 64) y[(size_t)row * dim + i] = (bf16_t)((p[i] - mean) * rstd * gamma[i] + beta[i]);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}
// dx[map(row)] += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma, xhat = (x - mean) * rstd;
// dx_bf (nullable): bf16 copy of the updated dx row (the operand of the next backward GEMM)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long long ld,
                                                     const int* __restrict__ rowmap, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                                     bf16_t* __restrict__ dx_bf, int rows, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t xr = (size_t)(rowmap ? rowmap[row] : row);
    const float* p = x + xr * ld;
    const float* g = dy + (size_t)row * dim;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float a = 0.f, b = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float gi = g[i] * gamma[i];
        a += gi;
        b += gi * (p[i] - mean) * rstd;
    }
    a = wave_sum(a) / dim;
    b = wave_sum(b) / dim;
    for (int i = lane; i < dim; i += 64) {
        const float gi = g[i] * gamma[i], xh = (p[i] - mean) * rstd;
        const float v = dx[xr * ld + i] + rstd * (gi - a - xh * b);
        dx[xr * ld + i] = v;
        if (dx_bf) dx_bf[xr * ld + i] = (bf16_t)v;
    }
}

// ---- self-attention backward, one workgroup per (batch, head), fp32 in LDS (S <= 80: the text tower's 77 tokens) -----
// qkv bf16 [B*S, 3d] (q | k | v), dout bf16 [B*S, d]  ->  dqkv bf16 [B*S, 3d]
constexpr int AB_MAXS = 80;            // 135 KiB of LDS
__global__ __launch_bounds__(256) void attention_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            bf16_t* __restrict__ dqkv, int S, int heads, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = reinterpret_cast<float*>(smem);             // [S][65]
    float* k = q + AB_MAXS * 65;
    float* v = k + AB_MAXS * 65;
    float* go = v + AB_MAXS * 65;
    float* P = go + AB_MAXS * 65;                          // [S][S+1]: probabilities, then dS
    float* dP = P + AB_MAXS * (AB_MAXS + 1);
    __shared__ float delta[AB_MAXS];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, d = heads * 64, ld = 3 * d;
    const int tid = threadIdx.x;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * 64;
    for (int i = tid; i < S * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        q[r * 65 + c] = (float)base[(size_t)r * ld + c];
        k[r * 65 + c] = (float)base[(size_t)r * ld + d + c];
        v[r * 65 + c] = (float)base[(size_t)r * ld + 2 * d + c];
        go[r * 65 + c] = (float)dout[((size_t)b * S + r) * d + h * 64 + c];
    }
    __syncthreads();
    const int SP = AB_MAXS + 1;
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        float s = 0.f, g = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            s += q[i * 65 + c] * k[j * 65 + c];
            g += go[i * 65 + c] * v[j * 65 + c];
        }
        P[i * SP + j] = (causal && j > i) ? -INFINITY : s * 0.125f;
        dP[i * SP + j] = g;
    }
    __syncthreads();
    {   // softmax rows + delta: one wave per row
        const int lane = tid & 63, w = tid >> 6;
        for (int i = w; i < S; i += 4) {
            float m = -INFINITY;
            for (int j = lane; j < S; j += 64) m = fmaxf(m, P[i * SP + j]);
            m = wave_max(m);
            float z = 0.f;
            for (int j = lane; j < S; j += 64) {
                const float p = __expf(P[i * SP + j] - m);
                P[i * SP + j] = p;
                z += p;
            }
            z = 1.0f / wave_sum(z);
            float dl = 0.f;
            for (int j = lane; j < S; j += 64) {
                const float p = P[i * SP + j] * z;
                P[i * SP + j] = p;
                dl += p * dP[i * SP + j];
            }
            dl = wave_sum(dl);
            if (lane == 0) delta[i] = dl;
        }
    }
    __syncthreads();
    bf16_t* ob = dqkv + (size_t)b * S * ld + h * 64;
    // dV = P^T dO (needs P), then dS overwrites dP, then dQ = dS K / 8, dK = dS^T Q / 8
    for (int e = tid; e < S * 64; e += 256) {
        const int j = e >> 6, c = e & 63;
        float a = 0.f;
        for (int i = 0; i < S; ++i) a += P[i * SP + j] * go[i * 65 + c];
        ob[(size_t)j * ld + 2 * d + c] = (bf16_t)a;
    }
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        dP[i * SP + j] = P[i * SP + j] * (dP[i * SP + j] - delta[i]) * 0.125f;
    }
    __syncthreads();
    for (int e = tid; e < S * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        float aq = 0.f, ak = 0.f;
        for (int j = 0; j < S; ++j) {
            aq += dP[r * SP + j] * k[j * 65 + c];
            ak += dP[j * SP + r] * q[j * 65 + c];
        }
        ob[(size_t)r * ld + c] = (bf16_t)aq;
        ob[(size_t)r * ld + d + c] = (bf16_t)ak;
    }
}

// ---- single-query cross-attention core (model.py:56-79) and its backward: one wave per (sample, head), lane = head dim ----
__global__ __launch_bounds__(256) void cross_core_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                             const bf16_t* __restrict__ Vp, bf16_t* __restrict__ out, int B,
                                                             int K, int heads) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= B * heads) return;
    const int b = w / heads, h = w % heads, inner = heads * 64;
    const float q = (float)Q[(size_t)b * inner + h * 64 + lane];
    float sc[32], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        sc[j] = -INFINITY;
        if (j < K) {
            sc[j] = wave_sum(q * (float)Kp[((size_t)b * K + j) * inner + h * 64 + lane]) * 0.125f;
            mx = fmaxf(mx, sc[j]);
        }
    }
    float sum = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < K) {
            const float p = __expf(sc[j] - mx);
            sum += p;
            o += p * (float)Vp[((size_t)b * K + j) * inner + h * 64 + lane];
        }
    out[(size_t)b * inner + h * 64 + lane] = (bf16_t)(o / sum);
}
__global__ __launch_bounds__(256) void cross_core_bwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                             const bf16_t* __restrict__ Vp, const bf16_t* __restrict__ dout,
                                                             bf16_t* __restrict__ dQ, bf16_t* __restrict__ dK,
                                                             bf16_t* __restrict__ dV, int B, int K, int heads) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= B * heads) return;
    const int b = w / heads, h = w % heads, inner = heads * 64;
    const size_t qo = (size_t)b * inner + h * 64 + lane;
    const float q = (float)Q[qo], go = (float)dout[qo];
    float p[32], dp[32], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        p[j] = -INFINITY;
        dp[j] = 0.f;
        if (j < K) {
            const size_t o = ((size_t)b * K + j) * inner + h * 64 + lane;
            p[j] = wave_sum(q * (float)Kp[o]) * 0.125f;
            dp[j] = wave_sum(go * (float)Vp[o]);
            mx = fmaxf(mx, p[j]);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < K) {
            p[j] = __expf(p[j] - mx);
            sum += p[j];
        }
    const float inv = 1.0f / sum;
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < K) {
            p[j] *= inv;
            delta += p[j] * dp[j];
        }
    float dq = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < K) {
            const size_t o = ((size_t)b * K + j) * inner + h * 64 + lane;
            const float ds = p[j] * (dp[j] - delta) * 0.125f;
            dq += ds * (float)Kp[o];
            dK[o] = (bf16_t)(ds * q);
            dV[o] = (bf16_t)(p[j] * go);
        }
    dQ[qo] = (bf16_t)dq;
}

// ---- loss (trainer.py:78-127): logits = scale * I . T^T over N rows (this rank's B first), both cross-entropies -----------
__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ img, const float* __restrict__ txt, int N, int dim,
                                                     float scale, float* __restrict__ logits) {
    // one wave per (i, 64 j's): lane = j
    const int i = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
    if (j >= N) return;
    const float* a = img + (size_t)i * dim;
    const float* b = txt + (size_t)j * dim;
    float s = 0.f;
    for (int c = 0; c < dim; c += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + c), y = *reinterpret_cast<const f32x4*>(b + c);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    logits[(size_t)i * N + j] = s * scale;
}
// lse[0][i] = logsumexp_j logits[i][j] (rows), lse[1][j] = logsumexp_i logits[i][j] (columns); one wave per row / column
__global__ __launch_bounds__(256) void lse_kernel(const float* __restrict__ logits, int N, float* __restrict__ lse) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= 2 * N) return;
    const int which = w / N, r = w % N;
    const long long sr = which == 0 ? N : 1, sc = which == 0 ? 1 : N;
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, logits[r * sr + j * sc]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < N; j += 64) z += __expf(logits[r * sr + j * sc] - m);
    z = wave_sum(z);
    if (lane == 0) lse[w] = m + __logf(z);
}
// loss = (mean_i (lse_row_i - l_ii) + mean_j (lse_col_j - l_jj)) / 2;  one block
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ logits, const float* __restrict__ lse, int N,
                                                          float* __restrict__ loss) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) s += (lse[i] - logits[(size_t)i * N + i]) + (lse[N + i] - logits[(size_t)i * N + i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + (part[2] + part[3])) / (2.0f * N);
}
// dT[j][:] = scale * sum_i G_ij I_i for j < B, G_ij = (softmax_row + softmax_col - 2 delta_ij) / (2N); one block per j
__global__ __launch_bounds__(256) void loss_grad_text_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                             const float* __restrict__ img, int N, int dim, float scale,
                                                             float* __restrict__ dtxt) {
    extern __shared__ float g[];                              // [N]
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float l = logits[(size_t)i * N + j];
        g[i] = (__expf(l - lse[i]) + __expf(l - lse[N + j]) - (i == j ? 2.f : 0.f)) * (0.5f / N) * scale;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += 256) {
        float a = 0.f;
        for (int i = 0; i < N; ++i) a += g[i] * img[(size_t)i * dim + c];
        dtxt[(size_t)j * dim + c] = a;
    }
}
// y = x / ||x||  backward: dx = (dy - y (y . dy)) / ||x||; one wave per row
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int rows, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * dim;
    const float* g = dy + (size_t)row * dim;
    float nn = 0.f, dot = 0.f;
    for (int i = lane; i < dim; i += 64) {
        nn += p[i] * p[i];
        dot += p[i] * g[i];
    }
    nn = wave_sum(nn);
    dot = wave_sum(dot);
    const float inv = rsqrtf(nn);
    for (int i = lane; i < dim; i += 64) dx[(size_t)row * dim + i] = (g[i] - p[i] * dot / nn) * inv;
}

// ---- AdamW (torch.optim.AdamW semantics: decoupled decay, bias-corrected moments; main.py:227-237) ----------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    float w = p[i];
    w -= lr * wd * w;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}

// ---- row scatter / gather helpers -----------------------------------------------------------------------------------
// dst[map[r]][:] (+)= src[r][:]   (read-out rows of the tower, spliced token rows)
__global__ void rows_scatter_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst,
                                    long long ld_dst, int rows, int dim, int accumulate) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)rows * dim) return;
    const int r = (int)(i / dim), c = (int)(i % dim);
    float* d = dst + (size_t)map[r] * ld_dst + c;
    *d = accumulate ? *d + src[i] : src[i];
}
__global__ void rows_gather_kernel(const float* __restrict__ src, long long ld_src, const int* __restrict__ map,
                                   float* __restrict__ dst, int rows, int dim) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)rows * dim) return;
    const int r = (int)(i / dim), c = (int)(i % dim);
    dst[i] = src[(size_t)map[r] * ld_src + c];
}

unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define KEDS_LAUNCHED(name) return keds_check_launch(name)

extern "C" int keds_transpose_to_bf16(const void* src, int src_is_f32, int64_t ld_src, int rows, int cols, void* out, int ld_out,
                                      void* stream) {
    KEDS_REQUIRE(src && out && rows > 0 && cols > 0 && ld_out >= rows && ld_src >= cols, "keds_transpose_to_bf16: bad argument");
    const dim3 grid((cols + 31) / 32, (ld_out + 31) / 32);
    if (src_is_f32)
        transpose_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)src, ld_src, rows, cols, (bf16_t*)out, ld_out);
    else
        transpose_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, ld_src, rows, cols, (bf16_t*)out, ld_out);
    KEDS_LAUNCHED("transpose_kernel");
}

extern "C" int keds_colsum(const void* x, int x_is_f32, int64_t ld, int rows, int cols, float* out, int accumulate, void* stream) {
    KEDS_REQUIRE(x && out && rows > 0 && cols > 0, "keds_colsum: bad argument");
    if (x_is_f32)
        colsum_kernel<float><<<(cols + 63) / 64, 256, 0, (hipStream_t)stream>>>((const float*)x, ld, rows, cols, out, accumulate);
    else
        colsum_kernel<bf16_t><<<(cols + 63) / 64, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ld, rows, cols, out, accumulate);
    KEDS_LAUNCHED("colsum_kernel");
}

extern "C" int keds_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, float p, void* stream) {
    KEDS_REQUIRE(mask && n > 0 && p >= 0.f && p < 1.f, "keds_dropout_mask: bad argument");
    dropout_mask_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(mask, n, seed, p);
    KEDS_LAUNCHED("dropout_mask_kernel");
}

extern "C" int keds_dropout_relu_fwd(const void* z, const uint8_t* mask, float scale, void* y, int64_t n, void* stream) {
    KEDS_REQUIRE(z && y && n > 0, "keds_dropout_relu_fwd: bad argument");
    dropout_relu_fwd_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>((const bf16_t*)z, mask, scale, (bf16_t*)y, n);
    KEDS_LAUNCHED("dropout_relu_fwd_kernel");
}

extern "C" int keds_dropout_relu_bwd(const void* dy, int dy_is_f32, const void* z, const uint8_t* mask, float scale, void* dz,
                                     int64_t n, void* stream) {
    KEDS_REQUIRE(dy && z && dz && n > 0, "keds_dropout_relu_bwd: bad argument");
    if (dy_is_f32)
        dropout_relu_bwd_kernel<float><<<blocks_for(n), 256, 0, (hipStream_t)stream>>>((const float*)dy, (const bf16_t*)z, mask, scale,
                                                                                       (bf16_t*)dz, n);
    else
        dropout_relu_bwd_kernel<bf16_t><<<blocks_for(n), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (const bf16_t*)z, mask,
                                                                                        scale, (bf16_t*)dz, n);
    KEDS_LAUNCHED("dropout_relu_bwd_kernel");
}

extern "C" int keds_qgelu_fwd(const void* u, void* y, int64_t n, void* stream) {
    KEDS_REQUIRE(u && y && n > 0, "keds_qgelu_fwd: bad argument");
    qgelu_fwd_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>((const bf16_t*)u, (bf16_t*)y, n);
    KEDS_LAUNCHED("qgelu_fwd_kernel");
}

extern "C" int keds_qgelu_bwd(const void* dy, const void* u, void* du, int64_t n, void* stream) {
    KEDS_REQUIRE(dy && u && du && n > 0, "keds_qgelu_bwd: bad argument");
    qgelu_bwd_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du, n);
    KEDS_LAUNCHED("qgelu_bwd_kernel");
}

extern "C" int keds_ln_fwd_stats(const float* x, int64_t ld, const int32_t* rowmap, const float* gamma, const float* beta, void* y,
                                 float* stats, int rows, int dim, void* stream) {
    KEDS_REQUIRE(x && gamma && beta && y && stats && rows > 0 && dim > 0, "keds_ln_fwd_stats: bad argument");
    ln_fwd_stats_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ld, rowmap, gamma, beta, (bf16_t*)y, stats, rows, dim);
    KEDS_LAUNCHED("ln_fwd_stats_kernel");
}

extern "C" int keds_ln_bwd(const float* dy, const float* x, int64_t ld, const int32_t* rowmap, const float* stats,
                           const float* gamma, float* dx, void* dx_bf16, int rows, int dim, void* stream) {
    KEDS_REQUIRE(dy && x && stats && gamma && dx && rows > 0 && dim > 0, "keds_ln_bwd: bad argument");
    ln_bwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(dy, x, ld, rowmap, stats, gamma, dx, (bf16_t*)dx_bf16, rows, dim);
    KEDS_LAUNCHED("ln_bwd_kernel");
}

extern "C" int keds_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int S, int heads, int causal, void* stream) {
    KEDS_REQUIRE(qkv && dout && dqkv && B > 0 && heads > 0, "keds_attention_bwd: bad argument");
    KEDS_REQUIRE(S >= 1 && S <= AB_MAXS, "keds_attention_bwd: S=%d unsupported (1..%d: the text tower)", S, AB_MAXS);
    const int lds = (4 * AB_MAXS * 65 + 2 * AB_MAXS * (AB_MAXS + 1)) * 4;
    if (int rc = keds_func_lds_once((const void*)attention_bwd_kernel, lds, "attention_bwd_kernel")) return rc;
    attention_bwd_kernel<<<B * heads, 256, lds, (hipStream_t)stream>>>((const bf16_t*)qkv, (const bf16_t*)dout, (bf16_t*)dqkv, S, heads,
                                                                       causal);
    KEDS_LAUNCHED("attention_bwd_kernel");
}

extern "C" int keds_cross_core_fwd(const void* Q, const void* Kp, const void* Vp, void* out, int B, int K, int heads, void* stream) {
    KEDS_REQUIRE(Q && Kp && Vp && out && B > 0 && K >= 1 && K <= 32 && heads > 0, "keds_cross_core_fwd: bad argument (K in [1,32])");
    cross_core_fwd_kernel<<<(B * heads + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)Q, (const bf16_t*)Kp, (const bf16_t*)Vp,
                                                                                (bf16_t*)out, B, K, heads);
    KEDS_LAUNCHED("cross_core_fwd_kernel");
}

extern "C" int keds_cross_core_bwd(const void* Q, const void* Kp, const void* Vp, const void* dout, void* dQ, void* dK, void* dV,
                                   int B, int K, int heads, void* stream) {
    KEDS_REQUIRE(Q && Kp && Vp && dout && dQ && dK && dV && B > 0 && K >= 1 && K <= 32 && heads > 0,
                 "keds_cross_core_bwd: bad argument (K in [1,32])");
    cross_core_bwd_kernel<<<(B * heads + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)Q, (const bf16_t*)Kp, (const bf16_t*)Vp,
                                                                                (const bf16_t*)dout, (bf16_t*)dQ, (bf16_t*)dK,
                                                                                (bf16_t*)dV, B, K, heads);
    KEDS_LAUNCHED("cross_core_bwd_kernel");
}

extern "C" size_t keds_clip_loss_workspace_bytes(int N) { return N > 0 ? ((size_t)N * N + 2 * (size_t)N) * sizeof(float) + 256 : 0; }

extern "C" int keds_clip_loss(const float* img_n, const float* txt_n, int N, int B_local, int dim, float scale, float* loss,
                              float* dtxt_n, void* workspace, size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(img_n && txt_n && loss && dtxt_n && workspace && N > 0 && B_local > 0 && B_local <= N && dim % 4 == 0,
                 "keds_clip_loss: bad argument");
    KEDS_REQUIRE(N <= 8192, "keds_clip_loss: at most 8192 gathered rows");
    KEDS_REQUIRE(workspace_bytes >= keds_clip_loss_workspace_bytes(N), "keds_clip_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* logits = (float*)workspace;
    float* lse = logits + (size_t)N * N;
    int rc;
    logits_kernel<<<dim3(N, (N + 255) / 256), 256, 0, st>>>(img_n, txt_n, N, dim, scale, logits);
    if ((rc = keds_check_launch("logits_kernel"))) return rc;
    lse_kernel<<<(2 * N + 3) / 4, 256, 0, st>>>(logits, N, lse);
    if ((rc = keds_check_launch("lse_kernel"))) return rc;
    loss_reduce_kernel<<<1, 256, 0, st>>>(logits, lse, N, loss);
    if ((rc = keds_check_launch("loss_reduce_kernel"))) return rc;
    loss_grad_text_kernel<<<B_local, 256, (size_t)N * sizeof(float), st>>>(logits, lse, img_n, N, dim, scale, dtxt_n);
    KEDS_LAUNCHED("loss_grad_text_kernel");
}

extern "C" int keds_l2norm_bwd(const float* x, const float* dy, float* dx, int rows, int dim, void* stream) {
    KEDS_REQUIRE(x && dy && dx && rows > 0 && dim > 0, "keds_l2norm_bwd: bad argument");
    l2norm_bwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, dy, dx, rows, dim);
    KEDS_LAUNCHED("l2norm_bwd_kernel");
}

extern "C" int keds_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale, void* stream) {
    KEDS_REQUIRE(p && g && m && v && n > 0 && step >= 1, "keds_adamw_step: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                                                                 grad_scale);
    KEDS_LAUNCHED("adamw_kernel");
}

extern "C" int keds_rows_scatter(const float* src, const int32_t* map, float* dst, int64_t ld_dst, int rows, int dim, int accumulate,
                                 void* stream) {
    KEDS_REQUIRE(src && map && dst && rows > 0 && dim > 0, "keds_rows_scatter: bad argument");
    rows_scatter_kernel<<<blocks_for((long long)rows * dim), 256, 0, (hipStream_t)stream>>>(src, map, dst, ld_dst, rows, dim, accumulate);
    KEDS_LAUNCHED("rows_scatter_kernel");
}

extern "C" int keds_rows_gather(const float* src, int64_t ld_src, const int32_t* map, float* dst, int rows, int dim, void* stream) {
    KEDS_REQUIRE(src && map && dst && rows > 0 && dim > 0, "keds_rows_gather: bad argument");
    rows_gather_kernel<<<blocks_for((long long)rows * dim), 256, 0, (hipStream_t)stream>>>(src, ld_src, map, dst, rows, dim);
    KEDS_LAUNCHED("rows_gather_kernel");
}
