// Memory-bound helpers of the encoder path: LayerNorm (fp32 statistics), patch im2col,
// token embedding + pseudo-token splice + positional embedding, read-out, L2 normalisation.
// References: src/model/model.py:291-297 (LayerNorm), :394-398 (patch tokens), :579-581 and
// :817-837 (text embedding and splice), :412-414/:586-589/:841-849 (read-out),
// src/eval_utils.py:162,704-710 (normalise / mixture).
#include "keds_common.h"
#include <math.h>

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int LN_MAXV = 8;  // f32x4 chunks per lane: dim <= 2048

// one wave per output row r; source row = r*row_mul + (row_map ? row_map[r] : 0).
// NV = dim/256 f32x4 chunks per lane, known at compile time so all row loads (and gamma/beta) are issued
// back to back before the first reduction (predicated loads were being serialised: 2.6 TB/s -> see profiles/);
// NV == 0 is the generic fallback (dim == 128 or any dim <= 2048 that is a multiple of 4).
// OUT: 0 = bf16, 1 = fp32, 2 = a PAIR of fp16 planes (hi = out, lo = out + pair_plane elements: y = hi + lo to 22 bits, the A
// operand of a split-operand GEMM, keds_gemm_x3)
template <int OUT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long long x_stride,
                                                        const int* __restrict__ row_map, int row_mul,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out,
                                                        int rows, int dim, long long pair_plane) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    // (a gathered row outside its sequence -- a read-out column >= the cut the caller declared, keds_text_run_ex's seq_used -- would
    // read another sample's token: such a row comes out as NaN instead, loudly)
    const int rm = row_map ? row_map[r] : 0;
    const bool bad_row = row_map && (unsigned)rm >= (unsigned)row_mul;
    const long long src = (long long)r * row_mul + (bad_row ? 0 : rm);
    const float* p = x + src * x_stride;
    constexpr int MAXV = NV > 0 ? NV : LN_MAXV;
    f32x4 v[MAXV], gg[MAXV], bb[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        if (NV > 0 || i < dim) v[j] = *reinterpret_cast<const f32x4*>(p + i);
        else v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        if (NV > 0 || i < dim) {
            gg[j] = *reinterpret_cast<const f32x4*>(gamma + i);
            bb[j] = *reinterpret_cast<const f32x4*>(beta + i);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        if (NV > 0 || i < dim) {
            const f32x4 d = v[j] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = bad_row ? __builtin_nanf("") : rsqrtf(wave_sum(q) / (float)dim + LN_EPS);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        if (NV > 0 || i < dim) {
            const f32x4 y = (v[j] - mean) * rstd * gg[j] + bb[j];
            if constexpr (OUT == 1) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)r * dim + i) = y;
            } else if constexpr (OUT == 2) {
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                const f16x4 hi = f16x4{(f16_t)y[0], (f16_t)y[1], (f16_t)y[2], (f16_t)y[3]};
                const f16x4 lo = f16x4{(f16_t)(y[0] - (float)hi[0]), (f16_t)(y[1] - (float)hi[1]), (f16_t)(y[2] - (float)hi[2]),
                                       (f16_t)(y[3] - (float)hi[3])};
                f16_t* o = reinterpret_cast<f16_t*>(out) + (size_t)r * dim + i;
                *reinterpret_cast<f16x4*>(o) = hi;
                *reinterpret_cast<f16x4*>(o + pair_plane) = lo;
            } else {
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + (size_t)r * dim + i) =
                    bf16x4{(bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
            }
        }
    }
}

// one wave per row: bf16 copy + {sum, sum of squares} of a residual-stream row (what the RESID_STATS GEMM epilogue
// emits for every later block; this kernel provides them for the first one)
template <typename T>   // bf16_t, or f16_t for the fp16 residual stream
__global__ __launch_bounds__(256) void rowstats_cast_kernel(const float* __restrict__ x, T* __restrict__ xb,
                                                            float* __restrict__ stats, int rows, int dim) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* p = x + (size_t)r * dim;
    f32x4 v[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        v[j] = i < dim ? *reinterpret_cast<const f32x4*>(p + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int i = j * 256 + lane * 4;
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
        if (i < dim) {
            typedef __attribute__((ext_vector_type(4))) T tx4;
            *reinterpret_cast<tx4*>(xb + (size_t)r * dim + i) = tx4{(T)v[j][0], (T)v[j][1], (T)v[j][2], (T)v[j][3]};
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
        keds_stat_t* row = reinterpret_cast<keds_stat_t*>(stats) + 2 * (size_t)r;
        row[0] = keds_stat_fixed(s);
        row[1] = keds_stat_fixed(q);
    }
}

// one wave per output row n of W [N,K]: w_folded = bf16(W * gamma), csum = sum of the ROUNDED products (what the MFMA
// will multiply the row mean with), bias' = bias + W . beta in fp32
template <typename T>
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int N, int K,
                                                             T* __restrict__ wf, float* __restrict__ bias_csum) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float cs = 0.f, bb = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const T f = (T)(w * gamma[k]);
        wf[(size_t)n * K + k] = f;
        cs += (float)f;
        bb += w * beta[k];
    }
    cs = wave_sum(cs);
    bb = wave_sum(bb);
    if (lane == 0) {
        bias_csum[n] = (bias ? bias[n] : 0.f) + bb;
        bias_csum[N + n] = cs;
    }
}

// rows of the fp16 residual stream back to fp32 (row r at r * stride elements in both): one thread per 8 elements
__global__ __launch_bounds__(256) void cast_rows_f16_f32_kernel(const f16_t* __restrict__ x16, float* __restrict__ x32,
                                                                int rows, int dim, long long stride) {
    const int per_row = dim >> 3;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)rows * per_row) return;
    const int r = (int)(id / per_row), i = (int)(id - (long long)r * per_row) << 3;
    const f16x8 v = *reinterpret_cast<const f16x8*>(x16 + (size_t)r * stride + i);
    float* o = x32 + (size_t)r * stride + i;
    *reinterpret_cast<f32x4*>(o) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}

// one row per sample out of a [B, S, dim] stream into a compact [B, dim] buffer: dst[b] = src[b * S + row[b]]
// (the read-out rows of the text tower, model.py:587-589, 847-849: the last block runs on them only).
// MODE 0: 16-bit elements copied; 1: fp16 -> fp32; 2: fp32 copied.  One thread per 8 elements.
template <int MODE>
// GLOBAL (packed rows): row[b] is the row's index in src itself and S the number of rows src holds
__global__ __launch_bounds__(256) void gather_rows_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                          const int* __restrict__ row, int S, int B, int dim, int global_rows) {
    const int per_row = dim >> 3;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= B * per_row) return;
    const int b = id / per_row, i = (id - b * per_row) << 3;
    const size_t d_o = (size_t)b * dim + i;
    if ((unsigned)row[b] >= (unsigned)S) {        // a row outside its sequence (see layernorm_kernel): NaN, not a neighbour's token
        if constexpr (MODE == 0) {
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(dst) + d_o) = u32x4{0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu};
        } else {
            const float qn = __builtin_nanf("");
            float* o = reinterpret_cast<float*>(dst) + d_o;
            *reinterpret_cast<f32x4*>(o) = f32x4{qn, qn, qn, qn};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{qn, qn, qn, qn};
        }
        return;
    }
    const size_t so = ((global_rows ? (size_t)0 : (size_t)b * S) + row[b]) * dim + i;
    if constexpr (MODE == 0) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(dst) + d_o) =
            *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(src) + so);
    } else if constexpr (MODE == 1) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16_t*>(src) + so);
        float* o = reinterpret_cast<float*>(dst) + d_o;
        *reinterpret_cast<f32x4*>(o) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    } else {
        const float* q = reinterpret_cast<const float*>(src) + so;
        float* o = reinterpret_cast<float*>(dst) + d_o;
        *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(q);
        *reinterpret_cast<f32x4*>(o + 4) = *reinterpret_cast<const f32x4*>(q + 4);
    }
}

// ---- image preprocessing (src/model/clip.py:107-123 `_transform`, eval branch) on raw uint8 images ------------------
// Resize(n_px, bicubic) on the shorter side + CenterCrop(n_px) + ToTensor + Normalize, as PIL / torchvision do them:
// PIL resamples uint8 images in two separable passes (horizontal, then vertical), each with antialiasing support
// 2 * max(scale, 1), cubic a = -0.5, weights normalised to 1, and rounds the intermediate image back to uint8.
// One thread per output pixel re-creates that: for each source row under the vertical window it computes the
// horizontally resampled 8-bit value, then the vertical pass.  (PIL quantises the weights to 22 fractional bits; here
// they are fp32, so a rounding can flip by one 8-bit step on rare pixels.)
__device__ __forceinline__ float pil_bicubic(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}
__device__ __forceinline__ void pil_window(int out_i, float scale, int in_size, int& lo, int& hi, float& center, float& ss) {
    const float fs = fmaxf(scale, 1.f);
    const float support = 2.f * fs;
    center = (out_i + 0.5f) * scale;
    ss = 1.f / fs;
    lo = (int)(center - support + 0.5f);
    hi = (int)(center + support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    hi = hi > in_size ? in_size : hi;
}
__device__ __forceinline__ float clip8(float v) {               // PIL clip8 after its rounding shift
    v = floorf(v + 0.5f);
    return fminf(fmaxf(v, 0.f), 255.f);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ img, int B, int H, int W,
                                                         int RW, int RH, int left, int top, int n_px,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * n_px * n_px) return;
    const int b = idx / (n_px * n_px), rem = idx - b * n_px * n_px;
    const int oy = rem / n_px, ox = rem - oy * n_px;
    const unsigned char* src = img + (size_t)b * H * W * 3;
    const float sx = (float)W / (float)RW, sy = (float)H / (float)RH;
    int xlo, xhi, ylo, yhi;
    float xc, xss, yc, yss;
    pil_window(ox + left, sx, W, xlo, xhi, xc, xss);
    pil_window(oy + top, sy, H, ylo, yhi, yc, yss);
    float wxs = 0.f, wys = 0.f;
    for (int x = xlo; x < xhi; ++x) wxs += pil_bicubic((x - xc + 0.5f) * xss);
    for (int y = ylo; y < yhi; ++y) wys += pil_bicubic((y - yc + 0.5f) * yss);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    for (int y = ylo; y < yhi; ++y) {
        const float wy = pil_bicubic((y - yc + 0.5f) * yss) / wys;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
        if (RW == W) {                                         // PIL skips a pass that does not change the size
            const unsigned char* p = src + ((size_t)y * W + (ox + left)) * 3;
            h0 = p[0]; h1 = p[1]; h2 = p[2];
        } else {
            const unsigned char* row = src + (size_t)y * W * 3;
            for (int x = xlo; x < xhi; ++x) {
                const float wx = pil_bicubic((x - xc + 0.5f) * xss) / wxs;
                h0 += wx * row[x * 3];
                h1 += wx * row[x * 3 + 1];
                h2 += wx * row[x * 3 + 2];
            }
            h0 = clip8(h0); h1 = clip8(h1); h2 = clip8(h2);     // the horizontal pass is stored as uint8
        }
        if (RH == H) {
            if (y == oy + top) { acc0 = h0; acc1 = h1; acc2 = h2; }
        } else {
            acc0 += wy * h0; acc1 += wy * h1; acc2 += wy * h2;
        }
    }
    if (RH != H) { acc0 = clip8(acc0); acc1 = clip8(acc1); acc2 = clip8(acc2); }
    const size_t plane = (size_t)n_px * n_px;
    float* o = out + (size_t)b * 3 * plane + (size_t)oy * n_px + ox;
    o[0] = (acc0 * (1.f / 255.f) - m0) / s0;
    o[plane] = (acc1 * (1.f / 255.f) - m1) / s1;
    o[2 * plane] = (acc2 * (1.f / 255.f) - m2) / s2;
}

// The same pipeline with PIL's own arithmetic, bit for bit (src/model/clip.py:107-123 runs PIL's ImagingResample): the
// filter weights arrive as PIL's 22-bit fixed-point integers (computed on the host in float64 exactly as
// Resample.c:precompute_coeffs / normalize_coeffs_8bpc do -- keds_amd.ops.pil_bicubic_coeffs), every pass accumulates in
// int32 from the rounding constant 1 << 21 and stores clip8(acc >> 22), the horizontal pass first.  xb / yb: {first source
// index, taps} per OUTPUT column / row of the crop; xk / yk: their `ks` integer weights.  The uint8 result (optional u8
// output) equals PIL's resize + crop exactly, and the float output equals ToTensor + Normalize on it in fp32.
__global__ __launch_bounds__(256) void preprocess_pil_kernel(const unsigned char* __restrict__ img, int B, int H, int W,
                                                             int n_px, int need_h, int need_v, int left, int top,
                                                             const int* __restrict__ xb, const int* __restrict__ xk, int ksx,
                                                             const int* __restrict__ yb, const int* __restrict__ yk, int ksy,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out, unsigned char* __restrict__ out_u8) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * n_px * n_px) return;
    const int b = idx / (n_px * n_px), rem = idx - b * n_px * n_px;
    const int oy = rem / n_px, ox = rem - oy * n_px;
    const unsigned char* src = img + (size_t)b * H * W * 3;
    const int x0 = need_h ? xb[2 * ox] : ox + left, xn = need_h ? xb[2 * ox + 1] : 1;
    const int y0 = need_v ? yb[2 * oy] : oy + top, yn = need_v ? yb[2 * oy + 1] : 1;
    const int* kx = xk + (size_t)ox * ksx;
    const int* ky = yk + (size_t)oy * ksy;
    auto clip8 = [](int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); };
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    int r0 = 0, r1 = 0, r2 = 0;
    for (int y = 0; y < yn; ++y) {
        const unsigned char* row = src + ((size_t)(y0 + y) * W + x0) * 3;
        int h0, h1, h2;
        if (need_h) {
            h0 = h1 = h2 = 1 << 21;
            for (int x = 0; x < xn; ++x) {
                const int k = kx[x];
                h0 += k * row[3 * x];
                h1 += k * row[3 * x + 1];
                h2 += k * row[3 * x + 2];
            }
            h0 = clip8(h0); h1 = clip8(h1); h2 = clip8(h2);     // the horizontal pass is stored as uint8
        } else {
            h0 = row[0]; h1 = row[1]; h2 = row[2];
        }
        if (need_v) {
            const int k = ky[y];
            a0 += k * h0; a1 += k * h1; a2 += k * h2;
        } else {
            r0 = h0; r1 = h1; r2 = h2;
        }
    }
    if (need_v) { r0 = clip8(a0); r1 = clip8(a1); r2 = clip8(a2); }
    const size_t plane = (size_t)n_px * n_px;
    float* o = out + (size_t)b * 3 * plane + (size_t)oy * n_px + ox;
    o[0] = ((float)r0 / 255.0f - m0) / s0;                      // ToTensor then Normalize, fp32, same operations
    o[plane] = ((float)r1 / 255.0f - m1) / s1;
    o[2 * plane] = ((float)r2 / 255.0f - m2) / s2;
    if (out_u8) {
        unsigned char* u = out_u8 + ((size_t)b * plane + (size_t)oy * n_px + ox) * 3;
        u[0] = (unsigned char)r0; u[1] = (unsigned char)r1; u[2] = (unsigned char)r2;
    }
}

// one block per output row (b, patch); columns c*P*P + ky*P + kx, zero padded to Kpad
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int R,
                                                     int P, int Kpad) {
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
        out[(size_t)row * Kpad + k] = (bf16_t)v;
    }
}

// CLS rows: x[b*(G+1), :] = class_emb + pos[0]
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                int B, int S, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, j = i % d;
    x[(size_t)b * S * d + j] = cls[j] + pos[j];
}

// one block per (b, t)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* __restrict__ tokens,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ img_tokens, int n_tok,
                                                           int insert_col, float* __restrict__ x, int L, int Lx, int d,
                                                           const int* __restrict__ seq_off) {
    // Lx <= L: only the first Lx columns of every sequence are written, x is [B, Lx, d] (keds_text_run_ex: columns that
    // cannot reach the read-out under the causal mask are never embedded).  seq_off (packed rows, keds_text_run_packed): sample b
    // owns rows [seq_off[b], seq_off[b + 1]) of x and only its own columns are written
    const int b = blockIdx.x / Lx, t = blockIdx.x % Lx;
    size_t orow = (size_t)b * Lx + t;
    if (seq_off) {
        const int r0 = seq_off[b];
        if (t >= seq_off[b + 1] - r0) return;
        orow = (size_t)r0 + t;
    }
    const float* src;
    if (img_tokens && t >= insert_col && t < insert_col + n_tok) {
        src = img_tokens + ((size_t)b * n_tok + (t - insert_col)) * d;
    } else {
        const int col = (img_tokens && t >= insert_col + n_tok) ? t - (n_tok - 1) : t;
        src = table + (size_t)tokens[(size_t)b * L + col] * d;
    }
    const float* pe = pos + (size_t)t * d;
    float* o = x + orow * d;
    for (int i = threadIdx.x * 4; i < d; i += 1024)
        *reinterpret_cast<f32x4*>(o + i) =
            *reinterpret_cast<const f32x4*>(src + i) + *reinterpret_cast<const f32x4*>(pe + i);
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                     int dim) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* p = x + (size_t)r * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += p[i] * p[i];
    const float nrm = sqrtf(wave_sum(s));
    for (int i = lane; i < dim; i += 64) out[(size_t)r * dim + i] = p[i] / nrm;
}

__global__ __launch_bounds__(256) void mix_kernel(const float* __restrict__ a, const float* __restrict__ b, float wa,
                                                  float wb, float* __restrict__ an, float* __restrict__ bn,
                                                  float* __restrict__ mix, int rows, int dim) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* pa = a + (size_t)r * dim;
    const float* pb = b + (size_t)r * dim;
    float sa = 0.f, sb = 0.f;
    for (int i = lane; i < dim; i += 64) {
        sa += pa[i] * pa[i];
        sb += pb[i] * pb[i];
    }
    const float na = sqrtf(wave_sum(sa)), nb = sqrtf(wave_sum(sb));
    float sm = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float m = wa * (pa[i] / na) + wb * (pb[i] / nb);
        sm += m * m;
    }
    const float nm = sqrtf(wave_sum(sm));
    for (int i = lane; i < dim; i += 64) {
        const float va = pa[i] / na, vb = pb[i] / nb;
        an[(size_t)r * dim + i] = va;
        bn[(size_t)r * dim + i] = vb;
        mix[(size_t)r * dim + i] = (wa * va + wb * vb) / nm;
    }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long long n) {
    long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += step) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        *reinterpret_cast<bf16x4*>(out + i) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
    if (i < n && i + 3 >= n)
        for (long long j = i; j < n; ++j) out[j] = (bf16_t)x[j];
}

bool ln_dim_ok(int dim) { return dim > 0 && dim <= 2048 && dim % 4 == 0 && (dim % 256 == 0 || dim == 128); }

}  // namespace

template <int OUT>
static void ln_launch(const float* x, long long x_stride, const int* row_map, int row_mul, const float* gamma,
                      const float* beta, void* out, int rows, int dim, hipStream_t st, long long pair_plane = 0) {
    const unsigned grid = (rows + 3) / 4;
#define KEDS_LN(NV) layernorm_kernel<OUT, NV><<<grid, 256, 0, st>>>(x, x_stride, row_map, row_mul, gamma, beta, out, rows, dim, pair_plane)
    switch (dim) {
        case 256: KEDS_LN(1); break;
        case 512: KEDS_LN(2); break;
        case 768: KEDS_LN(3); break;
        case 1024: KEDS_LN(4); break;
        default: KEDS_LN(0); break;
    }
#undef KEDS_LN
}

int keds_layernorm_impl(const float* x, long long x_stride, const int* row_map, int row_mul, const float* gamma,
                        const float* beta, void* out, int out_f32, int rows, int dim, hipStream_t st) {
    KedsProfScope prof(KEDS_PROF_LN, st);
    if (out_f32) ln_launch<1>(x, x_stride, row_map, row_mul, gamma, beta, out, rows, dim, st);
    else ln_launch<0>(x, x_stride, row_map, row_mul, gamma, beta, out, rows, dim, st);
    return keds_check_launch("layernorm_kernel");
}

// The fp32x3 flow's LayerNorm -> planes pass as a STREAM (round 5): 2 x 24 launches per step at 84 us each (3.2 TB/s of read +
// write) in the one-wave-one-row form above -- every wave's life is one load latency, one reduction chain and one store burst,
// and gamma / beta come in again for every row.  Here a wave walks rows r, r + W, r + 2 W, ... with the NEXT row's loads in flight
// while it reduces and splits the current one, gamma / beta in registers for its whole life, a lane owning runs of 8 consecutive
// columns (16-byte plane stores, whole 128-byte lines per eight lanes), reductions on the VALU (DPP + permlane swaps, no LDS
// round trips).  dim = NJ * 512.  The arithmetic is the kernel's above (two-pass mean / variance in fp32) with another summation
// order inside the wave.
#define KEDS_LNS_DPP(x, CTRL) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), (CTRL), 0xF, 0xF, false))
__device__ __forceinline__ float lns_wave_sum(float x) {
    x += KEDS_LNS_DPP(x, 0xB1);     // lane ^ 1
    x += KEDS_LNS_DPP(x, 0x4E);     // lane ^ 2
    x += KEDS_LNS_DPP(x, 0x141);    // row_half_mirror: the other group of four
    x += KEDS_LNS_DPP(x, 0x140);    // row_mirror: the other half of the 16-lane row
    return rows_sum(x);             // the four rows (permlane16 / permlane32 swaps)
}
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_pair_stream_kernel(const float* __restrict__ x, long long x_stride,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    f16_t* __restrict__ out, long long plane, int rows) {
    constexpr int dim = NJ * 512;
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    f32x4 gg[NJ][2], bb[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gg[j][h] = *reinterpret_cast<const f32x4*>(gamma + j * 512 + lane * 8 + 4 * h);
            bb[j][h] = *reinterpret_cast<const f32x4*>(beta + j * 512 + lane * 8 + 4 * h);
        }
    f32x4 cur[NJ][2], nxt[NJ][2];
    auto fetch = [&](int row, f32x4 (&v)[NJ][2]) {
        const float* p = x + (long long)row * x_stride + lane * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v[j][0] = *reinterpret_cast<const f32x4*>(p + j * 512);
            v[j][1] = *reinterpret_cast<const f32x4*>(p + j * 512 + 4);
        }
    };
    fetch(r, cur);
    while (true) {
        const int rn = r + nw;
        if (rn < rows) fetch(rn, nxt);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) s += (cur[j][h][0] + cur[j][h][1]) + (cur[j][h][2] + cur[j][h][3]);
        const float mean = lns_wave_sum(s) * (1.0f / (float)dim);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 dv = cur[j][h] - mean;
                q += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
            }
        const float rstd = rsqrtf(lns_wave_sum(q) * (1.0f / (float)dim) + LN_EPS);
        f16_t* o = out + (size_t)r * dim + lane * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f16x8 hi, lo;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 y = (cur[j][h] - mean) * rstd * gg[j][h] + bb[j][h];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[4 * h + e] = (f16_t)y[e];
                    lo[4 * h + e] = (f16_t)(y[e] - (float)hi[4 * h + e]);
                }
            }
            *reinterpret_cast<f16x8*>(o + j * 512) = hi;
            *reinterpret_cast<f16x8*>(o + j * 512 + plane) = lo;
        }
        if (rn >= rows) break;
        r = rn;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cur[j][0] = nxt[j][0], cur[j][1] = nxt[j][1];
    }
}
static bool ln_stream_env() {          // KEDS_LN_STREAM=0 in the environment: the one-row-per-wave form (A/B)
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_LN_STREAM");
        v = !(e && e[0] == '0');
    }
    return v != 0;
}

// LayerNorm whose output is a pair of fp16 planes (hi = out, lo = out + plane elements; dense rows of `dim`)
int keds_layernorm_pair_impl(const float* x, long long x_stride, const float* gamma, const float* beta, void* out, long long plane,
                             int rows, int dim, hipStream_t st) {
    KedsProfScope prof(KEDS_PROF_LN, st);
    if ((dim == 1024 || dim == 512) && rows >= 2048 && x_stride % 4 == 0 && plane % 8 == 0 && ln_stream_env()) {
        // every wave resident at once (82 VGPRs: five per SIMD, five blocks per CU); at 32,768 rows a wave walks six or seven
        const int cus = keds_device_cus();
        int grid = cus * 5;
        if (grid * 4 > rows) grid = (rows + 3) / 4;
        if (dim == 1024) layernorm_pair_stream_kernel<2><<<grid, 256, 0, st>>>(x, x_stride, gamma, beta, (f16_t*)out, plane, rows);
        else layernorm_pair_stream_kernel<1><<<grid, 256, 0, st>>>(x, x_stride, gamma, beta, (f16_t*)out, plane, rows);
        return keds_check_launch("layernorm_pair_stream_kernel");
    }
    ln_launch<2>(x, x_stride, nullptr, 1, gamma, beta, out, rows, dim, st, plane);
    return keds_check_launch("layernorm_kernel<pair>");
}

// x fp32 [rows, cols] -> fp16 planes hi / lo (x = hi + lo: 22 significant bits for |x| >= 2^-3, an absolute 2^-25 below -- the low
// plane of a small value is an fp16 subnormal); one thread per 8 elements.  |x| >= 65504 does
// not fit an fp16 hi: the flag (nullable) is raised and the caller falls back to the f32-input MFMA flow.
// `scale` (an exact power of two; 1 for activations): the planes hold x * scale -- weights are split at a scale that puts their
// largest element in [2^13, 2^14), see keds_split_f16_weight.
__global__ __launch_bounds__(256) void split_f16_pair_kernel(const float* __restrict__ x, long long ld, long long rows, int cols,
                                                             f16_t* __restrict__ out, long long plane, int* __restrict__ overflow,
                                                             float scale) {
    const int per_row = cols >> 3;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * per_row) return;
    const long long r = id / per_row;
    const int i = (int)(id - r * per_row) << 3;
    const float* p = x + r * ld + i;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f16x8 hi, lo;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (j < 4 ? a[j] : b[j - 4]) * scale;
        bad |= !(fabsf(v) < 65504.0f);
        hi[j] = (f16_t)v;
        lo[j] = (f16_t)(v - (float)hi[j]);
    }
    f16_t* o = out + r * cols + i;
    *reinterpret_cast<f16x8*>(o) = hi;
    *reinterpret_cast<f16x8*>(o + plane) = lo;
    if (bad && overflow) *overflow = 1;
}

// max |x| over a dense array as the bit pattern of a non-negative float (atomicMax on unsigned keeps the float order); NaN / inf
// come out as a pattern >= 0x7F800000
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

extern "C" int keds_split_f16_pair(const float* x, int64_t ld, int64_t rows, int cols, void* out, int64_t plane, int* overflow,
                                   void* stream) {
    KEDS_REQUIRE(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 4 == 0 && plane >= rows * cols,
                 "keds_split_f16_pair: bad argument");
    KedsProfScope prof(KEDS_PROF_OTHER, (hipStream_t)stream);
    const long long threads = (long long)rows * (cols / 8);
    split_f16_pair_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, ld, rows, cols, (f16_t*)out, plane, overflow, 1.0f);
    return keds_check_launch("split_f16_pair_kernel");
}

// A weight matrix W [n, k] (dense fp32) as the planes of W * 2^e, e chosen so that max |W| * 2^e lies in [2^13, 2^14): the low
// plane lo = fp16(w 2^e - hi) of a typical element is then a NORMAL fp16 number and hi + lo carries w to 2^-22 relative down to
// |w| = 2^-16 max|W|, absolutely to 2^-39 max|W| below that.  (Round 5 split the weights as stored: with |w| ~ 1e-2 .. 1e-3, as in
// real CLIP checkpoints, lo fell into the fp16 subnormals -- spacing 2^-24 -- and hi + lo carried only 17 .. 14 bits: the advisor's
// finding on round 5.)  The scale is exact (a power of two) and comes back out in the GEMM's epilogue (keds_gemm_x3, w_exp).
// Packing-time call: it waits for the stream once (the exponent is needed on the host).
extern "C" int keds_split_f16_weight(const float* w, int64_t n, int k, void* out, int64_t plane, int* w_exp, void* stream) {
    KEDS_REQUIRE(w && out && w_exp && n > 0 && k > 0 && k % 8 == 0 && plane >= n * k, "keds_split_f16_weight: bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned* dmax = nullptr;
    if (hipMalloc(&dmax, sizeof(unsigned)) != hipSuccess) {
        keds_set_error("keds_split_f16_weight: out of device memory");
        return KEDS_E_LAUNCH;
    }
    unsigned hmax = 0;
    int rc = KEDS_OK;
    if (hipMemsetAsync(dmax, 0, sizeof(unsigned), st) != hipSuccess) rc = KEDS_E_LAUNCH;
    if (!rc) {
        absmax_bits_kernel<<<1024, 256, 0, st>>>(w, (long long)n * k, dmax);
        rc = keds_check_launch("absmax_bits_kernel");
    }
    if (!rc && (hipMemcpyAsync(&hmax, dmax, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess))
        rc = KEDS_E_LAUNCH;
    (void)hipFree(dmax);
    if (rc) {
        if (rc == KEDS_E_LAUNCH) keds_set_error("keds_split_f16_weight: device error while reading max |W|");
        return rc;
    }
    KEDS_REQUIRE(hmax < 0x7F800000u, "keds_split_f16_weight: the weight matrix holds a non-finite value");
    int e = 0;
    if (hmax) {                                       // floor(log2(max)) from the exponent field (subnormal max: scale by 2^40 at most)
        const int lg = (int)(hmax >> 23) - 127;
        e = 13 - lg;
        if (e > 40) e = 40;
        if (e < -100) e = -100;
    }
    *w_exp = e;
    const long long threads = (long long)n * (k / 8);
    split_f16_pair_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(w, k, n, k, (f16_t*)out, plane, nullptr, ldexpf(1.0f, e));
    return keds_check_launch("split_f16_pair_kernel");
}

extern "C" int keds_layernorm(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* out,
                              int out_f32, int rows, int dim, void* stream) {
    KEDS_REQUIRE(x && gamma && beta && out && rows > 0, "keds_layernorm: bad argument");
    KEDS_REQUIRE(ln_dim_ok(dim), "keds_layernorm: dim %d unsupported", dim);
    KEDS_REQUIRE(x_stride % 4 == 0, "keds_layernorm: row stride must be a multiple of 4");
    return keds_layernorm_impl(x, x_stride, nullptr, 1, gamma, beta, out, out_f32, rows, dim, (hipStream_t)stream);
}

extern "C" int keds_im2col(const float* image, void* out, int B, int R, int P, int Kpad, void* stream) {
    KEDS_REQUIRE(image && out && B > 0 && P > 0 && R % P == 0, "keds_im2col: bad argument");
    KEDS_REQUIRE(Kpad >= 3 * P * P && Kpad % 64 == 0, "keds_im2col: Kpad must cover 3*P*P and be a multiple of 64");
    const int g = R / P;
    KedsProfScope prof(KEDS_PROF_OTHER, (hipStream_t)stream);
    im2col_kernel<<<B * g * g, 256, 0, (hipStream_t)stream>>>(image, (bf16_t*)out, R, P, Kpad);
    return keds_check_launch("im2col_kernel");
}

int keds_cls_rows_impl(float* x, const float* cls, const float* pos, int B, int S, int d, hipStream_t st) {
    cls_rows_kernel<<<(B * d + 255) / 256, 256, 0, st>>>(x, cls, pos, B, S, d);
    return keds_check_launch("cls_rows_kernel");
}

int keds_embed_tokens_impl(const int32_t* tokens, const float* table, const float* pos, const float* img_tokens,
                           int n_tok, int insert_col, float* x, int B, int L, int Lx, int d, void* stream,
                           const int32_t* seq_off = nullptr);

extern "C" int keds_embed_tokens(const int32_t* tokens, const float* table, const float* pos, const float* img_tokens,
                                 int n_tok, int insert_col, float* x, int B, int L, int d, void* stream) {
    return keds_embed_tokens_impl(tokens, table, pos, img_tokens, n_tok, insert_col, x, B, L, L, d, stream);
}

// the first Lx columns only (x: [B, Lx, d]); a splice that reaches beyond Lx is cut there, like the context cut at L
int keds_embed_tokens_impl(const int32_t* tokens, const float* table, const float* pos, const float* img_tokens,
                           int n_tok, int insert_col, float* x, int B, int L, int Lx, int d, void* stream, const int32_t* seq_off) {
    KEDS_REQUIRE(tokens && table && pos && x && B > 0 && L > 0 && Lx > 0 && Lx <= L, "keds_embed_tokens: bad argument");
    KEDS_REQUIRE(d % 4 == 0, "keds_embed_tokens: d must be a multiple of 4");
    if (img_tokens) {
        KEDS_REQUIRE(n_tok >= 1 && insert_col >= 0 && insert_col + n_tok <= L,
                     "keds_embed_tokens: splice [%d,%d) outside the context of %d", insert_col, insert_col + n_tok, L);
    }
    KedsProfScope prof(KEDS_PROF_OTHER, (hipStream_t)stream);
    embed_tokens_kernel<<<B * Lx, 256, 0, (hipStream_t)stream>>>(tokens, table, pos, img_tokens, n_tok, insert_col, x, L,
                                                                  Lx, d, seq_off);
    return keds_check_launch("embed_tokens_kernel");
}

extern "C" size_t keds_readout_workspace_bytes(int B, int d) {
    return keds_align_up((size_t)B, 128) * d * 2;
}

extern "C" int keds_readout(const float* x, int S, const int32_t* row, const float* gamma, const float* beta,
                            const void* proj_t, float* out, int B, int d, int E, int normalize, void* workspace,
                            size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(x && gamma && beta && proj_t && out && workspace && B > 0, "keds_readout: bad argument");
    KEDS_REQUIRE(ln_dim_ok(d), "keds_readout: d %d unsupported", d);
    if (workspace_bytes < keds_readout_workspace_bytes(B, d)) {
        keds_set_error("keds_readout: workspace too small");
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = keds_layernorm_impl(x, d, row, S, gamma, beta, workspace, 0, B, d, st);
    if (rc) return rc;
    rc = keds_gemm_bt(workspace, proj_t, nullptr, out, B, E, d, KEDS_EPI_BIAS_F32, nullptr, 0, stream);
    if (rc) return rc;
    if (normalize) {
        l2norm_kernel<<<(B + 3) / 4, 256, 0, st>>>(out, out, B, E);
        rc = keds_check_launch("l2norm_kernel");
    }
    return rc;
}

extern "C" int keds_l2_normalize(const float* x, float* out, int rows, int dim, void* stream) {
    KEDS_REQUIRE(x && out && rows > 0 && dim > 0, "keds_l2_normalize: bad argument");
    l2norm_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, out, rows, dim);
    return keds_check_launch("l2norm_kernel");
}

extern "C" int keds_mix_normalize(const float* a, const float* b, float wa, float wb, float* a_n, float* b_n,
                                  float* mix, int rows, int dim, void* stream) {
    KEDS_REQUIRE(a && b && a_n && b_n && mix && rows > 0 && dim > 0, "keds_mix_normalize: bad argument");
    mix_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(a, b, wa, wb, a_n, b_n, mix, rows, dim);
    return keds_check_launch("mix_kernel");
}

extern "C" int keds_cast_bf16(const float* x, void* out, int64_t count, void* stream) {
    KEDS_REQUIRE(x && out && count > 0, "keds_cast_bf16: bad argument");
    long long blocks = (count / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    cast_bf16_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)out, count);
    return keds_check_launch("cast_bf16_kernel");
}

extern "C" int keds_rowstats_cast_ex(const float* x, void* xb, int out_f16, float* stats, int rows, int dim, void* stream) {
    KEDS_REQUIRE(x && xb && stats && rows > 0, "keds_rowstats_cast: bad argument");
    KEDS_REQUIRE(dim % 4 == 0 && dim >= 4 && dim <= 256 * LN_MAXV, "keds_rowstats_cast: dim %d unsupported", dim);
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_LN, st);
    if (out_f16)
        rowstats_cast_kernel<f16_t><<<(rows + 3) / 4, 256, 0, st>>>(x, (f16_t*)xb, stats, rows, dim);
    else
        rowstats_cast_kernel<bf16_t><<<(rows + 3) / 4, 256, 0, st>>>(x, (bf16_t*)xb, stats, rows, dim);
    return keds_check_launch("rowstats_cast_kernel");
}

extern "C" int keds_rowstats_cast(const float* x, void* xb, float* stats, int rows, int dim, void* stream) {
    return keds_rowstats_cast_ex(x, xb, 0, stats, rows, dim, stream);
}

int keds_cast_rows_f16_f32_impl(const void* x16, float* x32, int rows, int dim, long long stride, hipStream_t st) {
    KEDS_REQUIRE(x16 && x32 && rows > 0 && dim % 8 == 0 && stride >= dim, "cast_rows_f16_f32: bad argument");
    KedsProfScope prof(KEDS_PROF_OTHER, st);
    const long long threads = (long long)rows * (dim / 8);
    cast_rows_f16_f32_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>((const f16_t*)x16, x32, rows, dim, stride);
    return keds_check_launch("cast_rows_f16_f32_kernel");
}

int keds_gather_rows_impl(const void* src, void* dst, const int32_t* row, int S, int B, int dim, int mode, hipStream_t st,
                          bool global_rows) {
    KEDS_REQUIRE(src && dst && row && S > 0 && B > 0 && dim % 8 == 0 && mode >= 0 && mode <= 2, "gather_rows: bad argument");
    KedsProfScope prof(KEDS_PROF_OTHER, st);
    const unsigned grid = (unsigned)((B * (dim / 8) + 255) / 256);
    const int gr = global_rows ? 1 : 0;
    if (mode == 0) gather_rows_kernel<0><<<grid, 256, 0, st>>>(src, dst, row, S, B, dim, gr);
    else if (mode == 1) gather_rows_kernel<1><<<grid, 256, 0, st>>>(src, dst, row, S, B, dim, gr);
    else gather_rows_kernel<2><<<grid, 256, 0, st>>>(src, dst, row, S, B, dim, gr);
    return keds_check_launch("gather_rows_kernel");
}

extern "C" int keds_fold_layernorm_ex(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                      void* w_folded, int out_f16, float* bias_csum, void* stream) {
    KEDS_REQUIRE(W && gamma && beta && w_folded && bias_csum && N > 0 && K > 0, "keds_fold_layernorm: bad argument");
    if (out_f16)
        fold_layernorm_kernel<f16_t><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(W, bias, gamma, beta, N, K, (f16_t*)w_folded,
                                                                                   bias_csum);
    else
        fold_layernorm_kernel<bf16_t><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(W, bias, gamma, beta, N, K,
                                                                                    (bf16_t*)w_folded, bias_csum);
    return keds_check_launch("fold_layernorm_kernel");
}

extern "C" int keds_fold_layernorm(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                   void* w_folded, float* bias_csum, void* stream) {
    return keds_fold_layernorm_ex(W, bias, gamma, beta, N, K, w_folded, 0, bias_csum, stream);
}

extern "C" int keds_preprocess_pil(const unsigned char* images, int B, int H, int W, int n_px, int need_h, int need_v,
                                   int left, int top, const int32_t* xb, const int32_t* xk, int ksx, const int32_t* yb,
                                   const int32_t* yk, int ksy, const float* mean3, const float* std3, float* out,
                                   unsigned char* out_u8, void* stream) {
    KEDS_REQUIRE(images && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && n_px > 0, "keds_preprocess_pil: bad argument");
    KEDS_REQUIRE((!need_h || (xb && xk && ksx > 0)) && (!need_v || (yb && yk && ksy > 0)),
                 "keds_preprocess_pil: a resampled axis needs its bounds and weights");
    KEDS_REQUIRE(left >= 0 && top >= 0, "keds_preprocess_pil: negative crop offset");
    const long long total = (long long)B * n_px * n_px;
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_OTHER, st);
    preprocess_pil_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(images, B, H, W, n_px, need_h, need_v, left, top, xb, xk,
                                                                           ksx, yb, yk, ksy, mean3[0], mean3[1], mean3[2],
                                                                           std3[0], std3[1], std3[2], out, out_u8);
    return keds_check_launch("preprocess_pil_kernel");
}

extern "C" int keds_preprocess(const unsigned char* images, int B, int H, int W, int n_px, const float* mean3,
                               const float* std3, float* out, void* stream) {
    KEDS_REQUIRE(images && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && n_px > 0, "keds_preprocess: bad argument");
    // torchvision Resize(int): shorter side -> n_px, the other int(n_px * long / short); CenterCrop offsets round half up
    int RW, RH;
    if (W <= H) {
        RW = n_px;
        RH = (int)((long long)n_px * H / W);
    } else {
        RH = n_px;
        RW = (int)((long long)n_px * W / H);
    }
    KEDS_REQUIRE(RW >= n_px && RH >= n_px, "keds_preprocess: resized image smaller than the crop");
    const int left = (int)lrintf((RW - n_px) / 2.0f), top = (int)lrintf((RH - n_px) / 2.0f);
    const long long total = (long long)B * n_px * n_px;
    hipStream_t st = (hipStream_t)stream;
    KedsProfScope prof(KEDS_PROF_OTHER, st);
    preprocess_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(images, B, H, W, RW, RH, left, top, n_px, mean3[0],
                                                                       mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    return keds_check_launch("preprocess_kernel");
}
