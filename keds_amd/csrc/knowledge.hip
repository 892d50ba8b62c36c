// Dual-stream knowledge injection (one stream per keds_knowledge_run call):
//   m      = IM2TEXT(q)                                   (src/model/model.py:105-123)
//   I', T' = IM2TEXT(neighbour image rows), IM2TEXT(neighbour text rows)
//   fused  = CrossFormer_fuse(m, I', I');  cond = CrossFormer_cond(m, T', T')   (model.py:37-101)
//   tokens = [fused, cond, m]                              (src/eval_utils.py:661-672)
// IM2TEXT runs ONCE over the stacked rows [q; I; T] (B*(1+2K) rows) through the MFMA GEMM with
// fused bias+ReLU; the cross-attention core (1 query x K keys x heads) is one wave per
// (sample, head) with the head dimension (64) on the lanes.  keds_im2text_forward and
// keds_crossformer_forward expose the two modules on their own (IM2TEXT.forward /
// CrossFormer.forward of the reference API).
#include "keds_common.h"
#include <math.h>
#include <cstring>

namespace {

// one wave per (b, head): lane = head-dim element (dim_head = 64)
// (Kp / Vp rows have stride `kv_ld` elements: the fused form keeps every layer's projections side by side in one buffer)
__global__ __launch_bounds__(256) void cross_attn_core_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                              const bf16_t* __restrict__ Vp, bf16_t* __restrict__ out,
                                                              int B, int K, int heads, int kv_ld) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= B * heads) return;
    const int b = w / heads, h = w % heads;
    const int inner = heads * 64;
    const float q = (float)Q[(size_t)b * inner + h * 64 + lane];
    float sc[32];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < K) {
            const float kv = (float)Kp[((size_t)b * K + j) * kv_ld + h * 64 + lane];
            sc[j] = wave_sum(q * kv) * 0.125f;
            mx = fmaxf(mx, sc[j]);
        } else {
            sc[j] = -INFINITY;
        }
    }
    float sum = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < K) {
            const float p = __expf(sc[j] - mx);
            sum += p;
            o += p * (float)Vp[((size_t)b * K + j) * kv_ld + h * 64 + lane];
        }
    }
    out[(size_t)b * inner + h * 64 + lane] = (bf16_t)(o / sum);
}

// tokens[b, slot, :] = src[b, :]
__global__ void place_token_kernel(const float* __restrict__ src, float* __restrict__ tokens, int B, int dim, int slot,
                                   int nslots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i % dim;
    tokens[((size_t)b * nslots + slot) * dim + j] = src[i];
}

// rows [q; nbr_img; nbr_txt] (fp32) -> one bf16 matrix [B(1+2K), dim]: one launch instead of three casts
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ q, const float* __restrict__ ni,
                                                        const float* __restrict__ nt, bf16_t* __restrict__ rows, long long nq,
                                                        long long nn) {
    const long long i = (blockIdx.x * 256LL + threadIdx.x) * 8;
    if (i >= nq + 2 * nn) return;
    const float* src = i < nq ? q + i : (i < nq + nn ? ni + (i - nq) : nt + (i - nq - nn));
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    *reinterpret_cast<bf16x8*>(rows + i) = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                                                  (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}

// ---- fused CrossFormer weights (keds_hip.h, keds_crossformer_fused) -------------------------------------------------
__global__ void concat_rows_kernel(const bf16_t* __restrict__ src, const float* __restrict__ bsrc, bf16_t* __restrict__ dst,
                                   float* __restrict__ bdst, int rows, int cols) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < (long long)rows * cols) dst[i] = src[i];
    if (i < rows) bdst[i] = bsrc[i];
}
// W'[i][j] = sum_d Wq[i][d] Wo[d][j]  (fp32 sum of the bf16 weights, rounded once); b'[i] = sum_d Wq[i][d] bo[d] + bq[i]
__global__ __launch_bounds__(256) void fold_qo_kernel(const bf16_t* __restrict__ wq, const float* __restrict__ bq,
                                                      const bf16_t* __restrict__ wo, const float* __restrict__ bo, int inner,
                                                      int dim, bf16_t* __restrict__ wout, float* __restrict__ bout) {
    const int i = blockIdx.x, j0 = threadIdx.x;
    for (int j = j0; j < inner; j += 256) {
        float a = 0.f;
        for (int d = 0; d < dim; ++d) a += (float)wq[(size_t)i * dim + d] * (float)wo[(size_t)d * inner + j];
        wout[(size_t)i * inner + j] = (bf16_t)a;
    }
    if (j0 < 64) {
        float a = 0.f;
        for (int d = j0; d < dim; d += 64) a += (float)wq[(size_t)i * dim + d] * bo[d];
        a = wave_sum(a);
        if (j0 == 0) bout[i] = a + bq[i];
    }
}

size_t rpad(size_t r) { return keds_align_up(r, 128); }

int check_i2t(const keds_im2text_params* p, const char* who) {
    if (!p || p->n_layer < 1 || p->n_layer > 4 || !p->out_w) {
        keds_set_error("%s: bad IM2TEXT parameters", who);
        return KEDS_E_ARG;
    }
    if (p->dim_in % 128 || p->middle % 128 || p->dim_out % 128) {
        keds_set_error("%s: IM2TEXT dims (%d,%d,%d) must be multiples of 128", who, p->dim_in, p->middle, p->dim_out);
        return KEDS_E_ARG;
    }
    return KEDS_OK;
}

int check_xf(const keds_crossformer_params* p, const char* who) {
    if (!p || !p->layer || p->layers < 1 || p->heads < 1 || (p->heads * 64) % 128 || p->dim % 128) {
        keds_set_error("%s: bad CrossFormer parameters", who);
        return KEDS_E_ARG;
    }
    return KEDS_OK;
}

// ---- IM2TEXT on bf16 rows [R, dim_in] -> fp32 [R, dim_out]; scratch: 2 x [Rp, middle] bf16
size_t i2t_scratch(const keds_im2text_params* p, size_t R) { return 2 * keds_align_up(rpad(R) * p->middle * 2, 256); }

int i2t_run(const keds_im2text_params* p, const void* rows_bf, int R, float* out_f, char* scratch, void* stream) {
    char* hbuf[2] = {scratch, scratch + keds_align_up(rpad(R) * p->middle * 2, 256)};
    const void* cur = rows_bf;
    int cur_dim = p->dim_in, rc;
    for (int l = 0; l < p->n_layer; ++l) {
        if ((rc = keds_gemm_bt(cur, p->w[l], p->b[l], hbuf[l & 1], R, p->middle, cur_dim, KEDS_EPI_BIAS_RELU_BF16, nullptr,
                               0, stream)))
            return rc;
        cur = hbuf[l & 1];
        cur_dim = p->middle;
    }
    return keds_gemm_bt(cur, p->out_w, p->out_b, out_f, R, p->dim_out, p->middle, KEDS_EPI_BIAS_F32, nullptr, 0, stream);
}

// ---- CrossFormer on bf16 inputs: q [B,dim], k rows [B*K,dim], v rows [B*K,dim] -> fp32 [B,dim]
struct XfScratch {
    char *qcur, *Qp, *Kp, *Vp, *att;
    size_t bytes;
};
XfScratch carve_xf(const keds_crossformer_params* p, int B, int K, char* base) {
    XfScratch s;
    size_t off = 0;
    auto take = [&](size_t b) {
        char* r = base ? base + off : nullptr;
        off += keds_align_up(b, 256);
        return r;
    };
    const size_t Bp = rpad(B), BKp = rpad((size_t)B * K);
    const int inner = p->heads * 64;
    s.qcur = take(Bp * p->dim * 2);
    s.Qp = take(Bp * inner * 2);
    s.Kp = take(BKp * inner * 2);
    s.Vp = take(BKp * inner * 2);
    s.att = take(Bp * inner * 2);
    s.bytes = off;
    return s;
}

int xf_run(const keds_crossformer_params* p, const void* q_bf, const void* k_bf, const void* v_bf, int B, int K,
           float* out_f, char* scratch, void* stream) {
    XfScratch s = carve_xf(p, B, K, scratch);
    hipStream_t st = (hipStream_t)stream;
    const int dim = p->dim, inner = p->heads * 64, BK = B * K;
    const void* qin = q_bf;
    int rc;
    for (int l = 0; l < p->layers; ++l) {
        const keds_cross_layer_params& c = p->layer[l];
        if ((rc = keds_gemm_bt(qin, c.wq, c.bq, s.Qp, B, inner, dim, KEDS_EPI_BIAS_BF16, nullptr, 0, stream))) return rc;
        if ((rc = keds_gemm_bt(k_bf, c.wk, c.bk, s.Kp, BK, inner, dim, KEDS_EPI_BIAS_BF16, nullptr, 0, stream))) return rc;
        if ((rc = keds_gemm_bt(v_bf, c.wv, c.bv, s.Vp, BK, inner, dim, KEDS_EPI_BIAS_BF16, nullptr, 0, stream))) return rc;
        {
            KedsProfScope prof(KEDS_PROF_OTHER, st);
            cross_attn_core_kernel<<<(B * p->heads + 3) / 4, 256, 0, st>>>((const bf16_t*)s.Qp, (const bf16_t*)s.Kp,
                                                                           (const bf16_t*)s.Vp, (bf16_t*)s.att, B, K,
                                                                           p->heads, inner);
            if ((rc = keds_check_launch("cross_attn_core_kernel"))) return rc;
        }
        if (l == p->layers - 1) {
            if ((rc = keds_gemm_bt(s.att, c.wo, c.bo, out_f, B, dim, inner, KEDS_EPI_BIAS_F32, nullptr, 0, stream))) return rc;
        } else {
            if ((rc = keds_gemm_bt(s.att, c.wo, c.bo, s.qcur, B, dim, inner, KEDS_EPI_BIAS_BF16, nullptr, 0, stream)))
                return rc;
            qin = s.qcur;
        }
    }
    return KEDS_OK;
}

// Fused form (p->fused): ONE GEMM projects the neighbour rows for all layers' k and v, each later layer's query comes
// straight from the previous layer's attention output (folded Wq.Wo), and the last output projection writes fp32 rows with
// stride `ld_out` (e.g. a token slot of [B,3,dim]).  2 + 2*layers launches instead of 5*layers.
struct XfFusedScratch {
    char *KV, *Qp, *att;
    size_t bytes;
};
XfFusedScratch carve_xf_fused(const keds_crossformer_params* p, int B, int K, char* base) {
    XfFusedScratch s;
    size_t off = 0;
    auto take = [&](size_t b) {
        char* r = base ? base + off : nullptr;
        off += keds_align_up(b, 256);
        return r;
    };
    const int inner = p->heads * 64;
    s.KV = take(rpad((size_t)B * K) * (size_t)p->layers * 2 * inner * 2);
    s.Qp = take(rpad(B) * inner * 2);
    s.att = take(rpad(B) * inner * 2);
    s.bytes = off;
    return s;
}

int xf_run_fused(const keds_crossformer_params* p, const void* q_bf, const void* kv_bf, int B, int K, float* out_f,
                 long long ld_out, char* scratch, hipStream_t st) {
    const keds_crossformer_fused* f = p->fused;
    XfFusedScratch s = carve_xf_fused(p, B, K, scratch);
    const int dim = p->dim, inner = p->heads * 64, BK = B * K, kv_ld = p->layers * 2 * inner;
    int rc;
    if ((rc = keds_gemm_bt(kv_bf, f->wkv, f->bkv, s.KV, BK, kv_ld, dim, KEDS_EPI_BIAS_BF16, nullptr, 0, st))) return rc;
    for (int l = 0; l < p->layers; ++l) {
        if (l == 0) {
            if ((rc = keds_gemm_bt(q_bf, p->layer[0].wq, p->layer[0].bq, s.Qp, B, inner, dim, KEDS_EPI_BIAS_BF16, nullptr, 0, st)))
                return rc;
        } else if ((rc = keds_gemm_bt(s.att, f->wqn[l], f->bqn[l], s.Qp, B, inner, inner, KEDS_EPI_BIAS_BF16, nullptr, 0, st)))
            return rc;
        {
            KedsProfScope prof(KEDS_PROF_OTHER, st);
            const bf16_t* kp = (const bf16_t*)s.KV + (size_t)l * 2 * inner;
            cross_attn_core_kernel<<<(B * p->heads + 3) / 4, 256, 0, st>>>((const bf16_t*)s.Qp, kp, kp + inner, (bf16_t*)s.att, B, K,
                                                                           p->heads, kv_ld);
            if ((rc = keds_check_launch("cross_attn_core_kernel"))) return rc;
        }
    }
    const keds_cross_layer_params& last = p->layer[p->layers - 1];
    return keds_gemm_bt_ex(s.att, inner, last.wo, last.bo, out_f, ld_out, B, dim, inner, KEDS_EPI_BIAS_F32, nullptr, 0, st);
}

}  // namespace

extern "C" size_t keds_crossformer_fused_bytes(const keds_crossformer_params* p) {
    if (!p || p->layers < 1 || p->layers > 8) return 0;
    const size_t inner = (size_t)p->heads * 64;
    size_t b = keds_align_up((size_t)p->layers * 2 * inner * p->dim * 2, 256) + keds_align_up((size_t)p->layers * 2 * inner * 4, 256);
    b += (size_t)(p->layers - 1) * (keds_align_up(inner * inner * 2, 256) + keds_align_up(inner * 4, 256));
    return b;
}

extern "C" int keds_crossformer_fuse(const keds_crossformer_params* p, void* buffer, size_t buffer_bytes,
                                     keds_crossformer_fused* out, void* stream) {
    int rc = check_xf(p, "keds_crossformer_fuse");
    if (rc) return rc;
    KEDS_REQUIRE(buffer && out && p->layers <= 8, "keds_crossformer_fuse: bad argument (at most 8 layers)");
    KEDS_REQUIRE(buffer_bytes >= keds_crossformer_fused_bytes(p), "keds_crossformer_fuse: buffer too small");
    hipStream_t st = (hipStream_t)stream;
    const int inner = p->heads * 64, dim = p->dim;
    char* base = (char*)buffer;
    size_t off = 0;
    auto take = [&](size_t b) {
        char* r = base + off;
        off += keds_align_up(b, 256);
        return r;
    };
    bf16_t* wkv = (bf16_t*)take((size_t)p->layers * 2 * inner * dim * 2);
    float* bkv = (float*)take((size_t)p->layers * 2 * inner * 4);
    memset(out, 0, sizeof(*out));
    out->wkv = wkv;
    out->bkv = bkv;
    const unsigned blocks = (unsigned)(((size_t)inner * dim + 255) / 256);
    for (int l = 0; l < p->layers; ++l) {
        const keds_cross_layer_params& c = p->layer[l];
        concat_rows_kernel<<<blocks, 256, 0, st>>>((const bf16_t*)c.wk, c.bk, wkv + (size_t)(2 * l) * inner * dim, bkv + (2 * l) * inner,
                                                   inner, dim);
        concat_rows_kernel<<<blocks, 256, 0, st>>>((const bf16_t*)c.wv, c.bv, wkv + (size_t)(2 * l + 1) * inner * dim,
                                                   bkv + (2 * l + 1) * inner, inner, dim);
        if (l >= 1) {
            bf16_t* w = (bf16_t*)take((size_t)inner * inner * 2);
            float* b = (float*)take((size_t)inner * 4);
            const keds_cross_layer_params& prev = p->layer[l - 1];
            fold_qo_kernel<<<inner, 256, 0, st>>>((const bf16_t*)c.wq, c.bq, (const bf16_t*)prev.wo, prev.bo, inner, dim, w, b);
            out->wqn[l] = w;
            out->bqn[l] = b;
        }
    }
    return keds_check_launch("keds_crossformer_fuse");
}

// ---- standalone IM2TEXT ------------------------------------------------------------------------
extern "C" size_t keds_im2text_workspace_bytes(const keds_im2text_params* p, int rows) {
    if (!p || rows <= 0) return 0;
    return keds_align_up(rpad(rows) * p->dim_in * 2, 256) + i2t_scratch(p, rows);
}

extern "C" int keds_im2text_forward(const keds_im2text_params* p, const float* x, int rows, float* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    int rc = check_i2t(p, "keds_im2text_forward");
    if (rc) return rc;
    KEDS_REQUIRE(x && out && workspace && rows > 0, "keds_im2text_forward: bad argument");
    if (workspace_bytes < keds_im2text_workspace_bytes(p, rows)) {
        keds_set_error("keds_im2text_forward: workspace too small");
        return KEDS_E_WORKSPACE;
    }
    char* xb = (char*)workspace;
    char* scratch = xb + keds_align_up(rpad(rows) * p->dim_in * 2, 256);
    if ((rc = keds_cast_bf16(x, xb, (int64_t)rows * p->dim_in, stream))) return rc;
    return i2t_run(p, xb, rows, out, scratch, stream);
}

// ---- standalone CrossFormer --------------------------------------------------------------------
extern "C" size_t keds_crossformer_workspace_bytes(const keds_crossformer_params* p, int B, int K) {
    if (!p || B <= 0 || K <= 0) return 0;
    const size_t qb = keds_align_up(rpad(B) * p->dim * 2, 256);
    const size_t kb = keds_align_up(rpad((size_t)B * K) * p->dim * 2, 256);
    size_t sc = carve_xf(p, B, K, nullptr).bytes;
    if (p->fused && carve_xf_fused(p, B, K, nullptr).bytes > sc) sc = carve_xf_fused(p, B, K, nullptr).bytes;
    return qb + 2 * kb + sc;
}

extern "C" int keds_crossformer_forward(const keds_crossformer_params* p, const float* q, const float* k, const float* v,
                                        int B, int K, float* out, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    int rc = check_xf(p, "keds_crossformer_forward");
    if (rc) return rc;
    KEDS_REQUIRE(q && k && v && out && workspace && B > 0, "keds_crossformer_forward: bad argument");
    KEDS_REQUIRE(K >= 1 && K <= 32, "keds_crossformer_forward: K=%d must be in [1,32]", K);
    if (workspace_bytes < keds_crossformer_workspace_bytes(p, B, K)) {
        keds_set_error("keds_crossformer_forward: workspace too small");
        return KEDS_E_WORKSPACE;
    }
    const size_t qb = keds_align_up(rpad(B) * p->dim * 2, 256);
    const size_t kb = keds_align_up(rpad((size_t)B * K) * p->dim * 2, 256);
    char* base = (char*)workspace;
    char *q_bf = base, *k_bf = base + qb, *v_bf = base + qb + kb, *scratch = base + qb + 2 * kb;
    if ((rc = keds_cast_bf16(q, q_bf, (int64_t)B * p->dim, stream))) return rc;
    if ((rc = keds_cast_bf16(k, k_bf, (int64_t)B * K * p->dim, stream))) return rc;
    if (v != k) {
        if ((rc = keds_cast_bf16(v, v_bf, (int64_t)B * K * p->dim, stream))) return rc;
    } else {
        v_bf = k_bf;
        if (p->fused) return xf_run_fused(p, q_bf, k_bf, B, K, out, p->dim, scratch, (hipStream_t)stream);
    }
    return xf_run(p, q_bf, k_bf, v_bf, B, K, out, scratch, stream);
}

// ---- one full stream ---------------------------------------------------------------------------
namespace {
struct KnWs {
    char* rows_bf;   // [R, dim] bf16 stacked q | nbr_img | nbr_txt (R = B(1+2K))
    float* map_f;    // [R, dim] fp32 IM2TEXT output
    char* map_bf;    // [R + 128, dim] bf16
    float* outf;     // [Bp, dim] fp32 CrossFormer output
    char* i2t;       // IM2TEXT scratch
    char* xf;        // CrossFormer scratch (retrieval_fuse)
    char* xf2;       // ... of text_condition: the two chains run side by side on two lanes
    size_t bytes;
};
KnWs carve_kn(const keds_knowledge_params* p, int B, int K, void* ws) {
    KnWs w;
    char* base = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t b) {
        char* r = base ? base + off : nullptr;
        off += keds_align_up(b, 256);
        return r;
    };
    const size_t R = (size_t)B * (1 + 2 * K);
    const int dim = p->i2t.dim_out;
    w.rows_bf = take(rpad(R) * p->i2t.dim_in * 2);
    w.map_f = (float*)take(rpad(R) * dim * 4);
    w.map_bf = take((rpad(R) + 128) * dim * 2);   // GEMMs on sub-ranges may read up to 127 rows past R
    w.outf = (float*)take(rpad(B) * dim * 4);
    w.i2t = take(i2t_scratch(&p->i2t, R));
    size_t xb = carve_xf(&p->fuse, B, K, nullptr).bytes;
    if (p->fuse.fused && carve_xf_fused(&p->fuse, B, K, nullptr).bytes > xb) xb = carve_xf_fused(&p->fuse, B, K, nullptr).bytes;
    w.xf = take(xb);
    w.xf2 = take(xb);
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" size_t keds_knowledge_workspace_bytes(const keds_knowledge_params* p, int B, int K) {
    if (!p || B <= 0 || K <= 0) return 0;
    return carve_kn(p, B, K, nullptr).bytes;
}

extern "C" int keds_knowledge_run(const keds_knowledge_params* p, const float* q, const float* nbr_img,
                                      const float* nbr_txt, int B, int K, float* tokens_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(p && q && nbr_img && nbr_txt && tokens_out && workspace, "keds_knowledge_run: null pointer");
    KEDS_REQUIRE(B > 0 && K >= 1 && K <= 32, "keds_knowledge_run: K must be in [1,32]");
    int rc;
    if ((rc = check_i2t(&p->i2t, "keds_knowledge_run"))) return rc;
    if ((rc = check_xf(&p->fuse, "keds_knowledge_run"))) return rc;
    if ((rc = check_xf(&p->cond, "keds_knowledge_run"))) return rc;
    const int dim = p->i2t.dim_out;
    KEDS_REQUIRE(p->i2t.dim_in == dim && p->fuse.dim == dim && p->cond.dim == dim && p->fuse.heads == p->cond.heads,
                 "keds_knowledge_run: IM2TEXT and CrossFormer dims must agree");
    KnWs w = carve_kn(p, B, K, workspace);
    KedsSplitKScope no_split(nullptr, 0);     // the two CrossFormer chains run on two streams at once: no GEMM here may split K
    if (workspace_bytes < w.bytes) {
        keds_set_error("keds_knowledge_run: workspace %zu < %zu", workspace_bytes, w.bytes);
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int R = B * (1 + 2 * K), BK = B * K;
    bf16_t* rows = (bf16_t*)w.rows_bf;
    const bool fused = p->fuse.fused && p->cond.fused && p->fuse.layers == p->cond.layers;
    if (fused) {
        // 20 launches per stream instead of ~40 (x2 with the remainder-row splits of the GEMMs): one pack of the three
        // inputs, three IM2TEXT GEMMs (the last one writes the bf16 rows AND token slot 2), and per CrossFormer one k/v
        // GEMM for all layers + (query GEMM, attention core) per layer + the output GEMM straight into its token slot.
        // The two CrossFormer chains are independent: text_condition runs on the side lane beside retrieval_fuse.
        {
            KedsProfScope prof(KEDS_PROF_OTHER, st);
            const long long nq = (long long)B * dim, nn = (long long)BK * dim;
            pack_rows_kernel<<<(unsigned)(((nq + 2 * nn) / 8 + 255) / 256), 256, 0, st>>>(q, nbr_img, nbr_txt, rows, nq, nn);
            if ((rc = keds_check_launch("pack_rows_kernel"))) return rc;
        }
        char* hbuf[2] = {w.i2t, w.i2t + keds_align_up(rpad(R) * p->i2t.middle * 2, 256)};
        const void* cur = rows;
        int cur_dim = p->i2t.dim_in;
        for (int l = 0; l < p->i2t.n_layer; ++l) {
            if ((rc = keds_gemm_bt(cur, p->i2t.w[l], p->i2t.b[l], hbuf[l & 1], R, p->i2t.middle, cur_dim, KEDS_EPI_BIAS_RELU_BF16,
                                   nullptr, 0, stream)))
                return rc;
            cur = hbuf[l & 1];
            cur_dim = p->i2t.middle;
        }
        if ((rc = keds_gemm_bt(cur, p->i2t.out_w, p->i2t.out_b, w.map_bf, R, dim, p->i2t.middle, KEDS_EPI_BIAS_BF16_HEADF32,
                               tokens_out + 2 * dim, B, stream)))
            return rc;
        const bf16_t* map_bf = (const bf16_t*)w.map_bf;
        KedsSideLane* lane = keds_side_lane();
        hipStream_t s2 = st;
        if (lane) {
            s2 = lane->s;
            if ((rc = keds_stream_order(st, lane->fork, s2))) return rc;
        }
        rc = xf_run_fused(&p->cond, map_bf, map_bf + (size_t)(B + BK) * dim, B, K, tokens_out + dim, 3LL * dim, w.xf2, s2);
        if (!rc) rc = xf_run_fused(&p->fuse, map_bf, map_bf + (size_t)B * dim, B, K, tokens_out, 3LL * dim, w.xf, st);
        if (!rc && s2 != st) rc = keds_stream_order(s2, lane->join, st);
        return rc;
    }
    if ((rc = keds_cast_bf16(q, rows, (int64_t)B * dim, stream))) return rc;
    if ((rc = keds_cast_bf16(nbr_img, rows + (size_t)B * dim, (int64_t)BK * dim, stream))) return rc;
    if ((rc = keds_cast_bf16(nbr_txt, rows + (size_t)(B + BK) * dim, (int64_t)BK * dim, stream))) return rc;
    if ((rc = i2t_run(&p->i2t, rows, R, w.map_f, w.i2t, stream))) return rc;
    if ((rc = keds_cast_bf16(w.map_f, w.map_bf, (int64_t)R * dim, stream))) return rc;
    place_token_kernel<<<(B * dim + 255) / 256, 256, 0, st>>>(w.map_f, tokens_out, B, dim, 2, 3);
    if ((rc = keds_check_launch("place_token_kernel"))) return rc;
    const bf16_t* map_bf = (const bf16_t*)w.map_bf;
    for (int which = 0; which < 2; ++which) {
        const keds_crossformer_params* xf = which == 0 ? &p->fuse : &p->cond;
        const bf16_t* nb = map_bf + (size_t)(B + which * BK) * dim;
        if ((rc = xf_run(xf, map_bf, nb, nb, B, K, w.outf, w.xf, stream))) return rc;
        place_token_kernel<<<(B * dim + 255) / 256, 256, 0, st>>>(w.outf, tokens_out, B, dim, which, 3);
        if ((rc = keds_check_launch("place_token_kernel"))) return rc;
    }
    return KEDS_OK;
}
