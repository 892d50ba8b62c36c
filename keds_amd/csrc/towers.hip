// Whole-tower forward passes built from the kernels in gemm/attention/elementwise:
//   keds_tower_forward : N pre-LN residual blocks        (src/model/model.py:305-326,372-373)
//   keds_vit_run   : CLIP.encode_image, ViT branch   (src/model/model.py:569-575,393-415)
//   keds_text_run  : CLIP.encode_text / encode_text_img_retrieval (src/model/model.py:577-590,808-851)
// Host code only: every launch is asynchronous on the caller's stream, the caller owns the
// workspace, nothing is allocated or synchronised here.
#include "keds_common.h"
#include <cstdlib>

int keds_layernorm_impl(const float* x, long long x_stride, const int* row_map, int row_mul, const float* gamma,
                        const float* beta, void* out, int out_f32, int rows, int dim, hipStream_t st);
int keds_cls_rows_impl(float* x, const float* cls, const float* pos, int B, int S, int d, hipStream_t st);
int keds_cast_rows_f16_f32_impl(const void* x16, float* x32, int rows, int dim, long long stride, hipStream_t st);
int keds_gather_rows_impl(const void* src, void* dst, const int32_t* row, int S, int B, int dim, int mode, hipStream_t st,
                          bool global_rows = false);
int keds_embed_tokens_impl(const int32_t* tokens, const float* table, const float* pos, const float* img_tokens,
                           int n_tok, int insert_col, float* x, int B, int L, int Lx, int d, void* stream,
                           const int32_t* seq_off = nullptr);

bool keds_gemm_splits_rows(int M, int N, int K);   // gemm.hip
void keds_gemm_small_lds(int on);                  // gemm.hip: small GEMM launches of this thread take the 64 KiB-LDS kernel form

// f32path.hip: the fp32-accurate flow (keds_tower_params.f32)
size_t keds_tower_f32_workspace_bytes(int width, int seq, int B);
int keds_tower_forward_f32(const keds_tower_params* p, float* x, int B, void* ws, hipStream_t st, const int32_t* last_rows = nullptr,
                           const PackedRows* pk = nullptr);
size_t keds_readout_f32_workspace_bytes(int B, int d);
int keds_readout_f32(const float* x, int S, const int32_t* row, const float* gamma, const float* beta, const float* proj_t,
                     float* out, int B, int d, int E, int normalize, void* workspace, hipStream_t st);

namespace {

size_t pad_rows(size_t m) { return keds_align_up(m, 256); }     // (256: a tower may run its ragged last row tile as a full one)

// scratch of one tower: h [Mp,w] (the fp16 residual stream of the folded flow; the bf16 LayerNorm output of the unfolded
// one) | qkv [Mp,3w] | attn [Mp,w] | MLP hidden [Mp,4w] (bf16) |
// two row-statistics buffers [Mp,2] of 64-bit fixed point (LayerNorm folded into the GEMMs: ln_1 / ln_2 statistics).
// No buffer aliases another: the remainder-row chain runs beside the full-tile chain (see RowLanes) and the two touch
// disjoint ROWS of every buffer, which only keeps them apart if the buffers themselves are distinct.
struct TowerWs {
    bf16_t *h, *qkv, *att, *hid;
    keds_stat_t *st1, *st2;
    // MXFP8 operands of the full 256-row tiles (fp8 towers): residual copy, attention output, MLP hidden (+ scale dwords)
    unsigned char *xq, *xs, *aq, *as, *hq, *hs;
    char* splitk;          // fp32 partial tiles of the split-K launches of THIS call (remainder rows, CLS tail)
    size_t bytes;
};

TowerWs carve_tower(void* ws, int width, int seq, int B) {
    const size_t Mp = pad_rows((size_t)B * seq);
    TowerWs t;
    char* p = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t n) {
        char* r = p ? p + off : nullptr;
        off += keds_align_up(n, 256);
        return r;
    };
    t.h = (bf16_t*)take(Mp * width * 2);
    t.qkv = (bf16_t*)take(Mp * (size_t)width * 3 * 2);
    t.att = (bf16_t*)take(Mp * (size_t)width * 2);
    t.hid = (bf16_t*)take(Mp * (size_t)width * 4 * 2);
    t.st1 = (keds_stat_t*)take(Mp * 2 * sizeof(keds_stat_t));
    t.st2 = (keds_stat_t*)take(Mp * 2 * sizeof(keds_stat_t));
    const size_t Mm = (size_t)B * seq / 256 * 256;
    const size_t q1 = Mm * width, s1 = Mm * (size_t)(width / 32);   // 1 scale byte / 32
    t.xq = (unsigned char*)take(q1);
    t.xs = (unsigned char*)take(s1);
    t.aq = (unsigned char*)take(q1);
    t.as = (unsigned char*)take(s1);
    t.hq = (unsigned char*)take(4 * q1);
    t.hs = (unsigned char*)take(4 * s1);
    t.splitk = take(KEDS_SPLITK_BYTES);
    t.bytes = off;
    return t;
}

// Two lanes for one tower pass.  At B = 128 a ViT-L/14 tower has 32,896 rows = 128 full 256-row tiles + 128 remainder
// rows; every GEMM of the remainder is a handful of workgroups whose time is pure latency (13-14 us each, 61 us per block,
// 5.6 % of the step when it runs between the full-tile launches).  Rows only meet in the attention, so per block the
// remainder chain (out-proj, fc, proj, next qkv) runs on the side lane while the caller's stream runs the same four GEMMs
// on the full tiles: fork after the attention, join before the next one.
// Measured in round 2 (bench step, 128 images): 21.85 ms with the side lane, 22.85 ms with the remainder launches between
// the full-tile ones, 21.46 ms with the remainder rows not computed at all (timing only) -- the lane recovers 0.6 of the
// 1.0 ms these 0.39 % of the rows cost.  A side workgroup cannot share a CU with a 256 x 256 tile (registers), so each one
// displaces a tile of the kernel running beside it; forking the chain behind the out-proj launch instead of in front of it
// (so that it runs under c_fc's eight rounds of tiles rather than out-proj's two) changes nothing measurable.
struct RowLanes {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool split = false;
    int init(hipStream_t st, bool want) {
        main = side = st;
        KedsSideLane* lane = want ? keds_side_lane() : nullptr;
        if (!lane) return KEDS_OK;
        side = lane->s;
        fork = lane->fork;                                    // the lane's own two events, created once (round 1 made and
        join = lane->join;                                    // destroyed a pair on every tower call)
        split = true;
        return KEDS_OK;
    }
    int to_side() { return split ? keds_stream_order(main, fork, side) : KEDS_OK; }
    // the fork behind an attention launch: the launch records the event itself (its stop event: no marker packet between it and
    // the next GEMM on the main stream); kernel forms that do not take it get the recorded event.  KEDS_FORK_EXT=0: always recorded (A/B)
    template <typename F>
    int attention_then_side(F launch) {
        static const bool ext = [] {
            const char* e = keds_exp_env("KEDS_FORK_EXT");
            return !(e && e[0] == '0');
        }();
        if (!split || !ext) {
            const int rc = launch();
            return rc ? rc : to_side();
        }
        keds_order_lock();
        keds_attention_stop_event(fork);
        int rc = launch();
        const bool taken = keds_attention_stop_event_taken();
        if (!rc && taken) rc = keds_stream_wait_locked(fork, side);
        keds_order_unlock();
        if (!rc && !taken) rc = to_side();
        return rc;
    }
    int to_main() { return split ? keds_stream_order(side, join, main) : KEDS_OK; }
};

struct RowSpan {
    size_t r0;
    int n;
    hipStream_t st;
};

// The four GEMMs of a block on a span of rows, LayerNorm folded (keds_hip.h, KEDS_EPI_LN_*).  The residual stream lives
// in t.h as fp16 (the reference's own storage type after convert_weights, model.py:531-548): one copy that the residual
// GEMMs update in place (fp32 sum, rounded once) and the LN-folded GEMMs read as their fp16 operand.
int qkv_rows(const TowerWs& t, const keds_block_params& k, int w, RowSpan s) {
    return keds_gemm_bt_ex2(t.h + s.r0 * w, w, k.qkv_wf, k.qkv_bc, t.qkv + s.r0 * 3 * w, 3 * w, s.n, 3 * w, w,
                            KEDS_EPI_LN_BIAS_BF16_H, (const float*)(t.st1 + 2 * s.r0), 0, t.st2 + 2 * s.r0, s.st);
}
int out_rows(const TowerWs& t, const keds_block_params& k, int w, RowSpan s) {
    return keds_gemm_bt_ex2(t.att + s.r0 * w, w, k.out_w, k.out_b, t.h + s.r0 * w, w, s.n, w, w, KEDS_EPI_RESID_STATS_F16,
                            (const float*)(t.st2 + 2 * s.r0), 0, nullptr, s.st);
}
int fc_rows(const TowerWs& t, const keds_block_params& k, int w, RowSpan s) {
    return keds_gemm_bt_ex2(t.h + s.r0 * w, w, k.fc_wf, k.fc_bc, t.hid + s.r0 * 4 * w, 4 * w, s.n, 4 * w, w,
                            KEDS_EPI_LN_QGELU_BF16_H, (const float*)(t.st2 + 2 * s.r0), 0, t.st1 + 2 * s.r0, s.st);
}
// (the last block's output feeds no further ln_1: no statistics)
int proj_rows(const TowerWs& t, const keds_block_params& k, int w, bool last, RowSpan s) {
    return keds_gemm_bt_ex2(t.hid + s.r0 * 4 * w, 4 * w, k.proj_w, k.proj_b, t.h + s.r0 * w, w, s.n, w, 4 * w,
                            KEDS_EPI_RESID_STATS_F16, last ? nullptr : (const float*)(t.st1 + 2 * s.r0), 0, nullptr, s.st);
}

// After the last block only token 0 of every sample is read (ln_post(x[:,0,:]), model.py:412), so the attention
// queries, out-proj, ln_2 and the MLP run on those B rows only (row stride S*w in x / attn); fp32 rows, bf16 kernels.
// x16: the fp16 residual stream whose CLS rows are brought back to fp32 first (folded flow), or nullptr.
int cls_rows_tail(const keds_tower_params* p, const keds_block_params& k, const TowerWs& t, float* x, const void* x16, int B,
                  hipStream_t st) {
    const int w = p->width, S = p->seq;
    const long long ld = (long long)S * w;
    int rc;
    if (x16 && (rc = keds_cast_rows_f16_f32_impl(x16, x, B, w, ld, st))) return rc;
    if ((rc = keds_attention_ex(t.qkv, t.att, B, S, p->heads, p->causal, 1, st))) return rc;
    if ((rc = keds_gemm_bt_ex(t.att, ld, k.out_w, k.out_b, x, ld, B, w, w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st))) return rc;
    if ((rc = keds_layernorm_impl(x, w, nullptr, S, k.ln2_g, k.ln2_b, t.h, 0, B, w, st))) return rc;
    if ((rc = keds_gemm_bt(t.h, k.fc_w, k.fc_b, t.hid, B, 4 * w, w, KEDS_EPI_BIAS_QGELU_BF16, nullptr, 0, st))) return rc;
    return keds_gemm_bt_ex(t.hid, 4 * w, k.proj_w, k.proj_b, x, ld, B, w, 4 * w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st);
}

// The text tower's form of the same cut: after the last block only ONE row of every sample is read -- the EOT column
// (+ n_tok - 1 with spliced pseudo tokens; model.py:587-589, 847-849) -- a different row per sample, known on the device only
// (`rows`, int32 [B]).  The block's in_proj and attention run on all rows (every key is needed; the attention is 1.6 % of a
// block), then the B read-out rows of the attention output and of the residual stream are gathered into compact [B, w]
// buffers (in the qkv buffer, which nobody reads any more) and out-proj, ln_2 and the MLP run on those.  On return the first
// B rows of x hold the block's output for sample b in row b: the caller reads out with S = 1, row 0.
// pk (packed rows): `rows` are GLOBAL row indices (off[b] + the sample's read-out column).
int rows_tail(const keds_tower_params* p, const keds_block_params& k, const TowerWs& t, float* x, const void* x16, int B,
              const int32_t* rows, hipStream_t st, const PackedRows* pk = nullptr) {
    const int w = p->width, S = p->seq;
    int rc;
    if (pk) rc = keds_attention_packed(t.qkv, t.att, B, S, pk->off, p->heads, p->causal, st);
    else rc = keds_attention(t.qkv, t.att, B, S, p->heads, p->causal, st);
    if (rc) return rc;
    bf16_t* att_c = t.qkv;                                               // [B, w] bf16 (2 w bytes per row: a multiple of 256)
    float* x_c = (float*)((char*)t.qkv + (size_t)B * w * 2);             // [B, w] fp32; 6 B w <= the buffer's 6 w pad256(B S)
    const int bound = pk ? pk->valid : S;
    if ((rc = keds_gather_rows_impl(t.att, att_c, rows, bound, B, w, 0, st, pk != nullptr))) return rc;
    if ((rc = keds_gather_rows_impl(x16 ? x16 : (const void*)x, x_c, rows, bound, B, w, x16 ? 1 : 2, st, pk != nullptr))) return rc;
    if ((rc = keds_gemm_bt_ex(att_c, w, k.out_w, k.out_b, x_c, w, B, w, w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st))) return rc;
    if ((rc = keds_layernorm_impl(x_c, w, nullptr, 1, k.ln2_g, k.ln2_b, t.h, 0, B, w, st))) return rc;
    if ((rc = keds_gemm_bt(t.h, k.fc_w, k.fc_b, t.hid, B, 4 * w, w, KEDS_EPI_BIAS_QGELU_BF16, nullptr, 0, st))) return rc;
    if ((rc = keds_gemm_bt_ex(t.hid, 4 * w, k.proj_w, k.proj_b, x_c, w, B, w, 4 * w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st))) return rc;
    if (hipMemcpyAsync(x, x_c, (size_t)B * w * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        keds_set_error("keds_tower_forward: read-out rows: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    return KEDS_OK;
}

// KEDS_TAIL_ATTN=1 in the environment: the tail samples' attention on the side lane (round-4 experiment, OFF by default:
// bit-identical and neutral -- 6,759-6,770 vs 6,770-6,776 img/s in four same-box pairs, profiles/r04_tail_attention_ab.txt)
#ifdef KEDS_EXPERIMENTS
bool tail_attention_on_side() {
    static int v = -1;
    if (v < 0) {
        const char* e = keds_exp_env("KEDS_TAIL_ATTN");
        v = e && e[0] == '1';
    }
    return v != 0;
}
#else
constexpr bool tail_attention_on_side() { return false; }
#endif

bool bf16_rows_split(int M, int w) {
    return keds_gemm_splits_rows(M, 3 * w, w) && keds_gemm_splits_rows(M, w, w) && keds_gemm_splits_rows(M, 4 * w, w) &&
           keds_gemm_splits_rows(M, w, 4 * w);
}

// Round 3 experiment, OFF by default: the ragged last row tile as a FULL tile.  At B = 128 a ViT-L/14 tower has 32,896 rows =
// 128 x 256 + 128; the 128 remainder rows run as their own chain of four small GEMMs per block on the side lane (0.39 ms per
// step).  Rows only meet in the attention, so the GEMMs can just as well run on 129 full tiles: 128 filler rows behind the
// last token (a copy of the first rows; never read by the attention -- their attention output stays zero -- or by the
// read-out), 0.39 % more GEMM work, one lane.  Correct (test_ragged_row_tile_on_filler_rows_matches_the_two_lane_tower) and
// 4 ms per step SLOWER (23.9 vs 19.85 ms, same-box A/B): 129 row tiles x 4..16 column tiles is no longer a whole number of
// 256-workgroup rounds, and every one of the 94 GEMM launches of a step pays a nearly empty extra round (12..16 tiles on 256
// CUs) that the side lane's small launches used to fill.  KEDS_TOWER_FILL=1 / keds_tower_fill_enable(1) turn it on.
#ifdef KEDS_EXPERIMENTS
int g_tower_fill = -1;                          // -1: take KEDS_TOWER_FILL (default off)
int tower_fill_rows(int M, int w) {
    if (g_tower_fill < 0) {
        const char* e = keds_exp_env("KEDS_TOWER_FILL");
        g_tower_fill = e && e[0] == '1';
    }
    const int fill = (256 - M % 256) % 256;
    return g_tower_fill && fill && bf16_rows_split(M, w) && (long)fill * 64 <= M ? fill : 0;
}
#else
constexpr int tower_fill_rows(int, int) { return 0; }      // (the product library: remainder rows always take the side lane)
#endif

// BASELINE config 5: the four GEMMs of every block on MXFP8 operands (gemm_fp8.hip).  The residual stream stays fp32;
// its MXFP8 copy (xq, xs), the attention output (aq) and the MLP hidden (hq) are e4m3 + one e8m0 scale per 32 columns,
// produced by the GEMM epilogues themselves (the attention output by the attention kernel).  Rows beyond the last full
// 256-row tile (128 of 32,896 at B = 128) keep the bf16 path, on the side lane: every producer has a bf16 twin for them.
int tower_forward_fp8(const keds_tower_params* p, float* x, int B, const TowerWs& t, int Mm, hipStream_t st,
                      const int32_t* last_rows) {
    const int w = p->width, S = p->seq;
    const int M = B * S, Mt = M - Mm;
    RowLanes lanes;
    int rc;
    if ((rc = lanes.init(st, Mt > 0))) return rc;
    const RowSpan rem{(size_t)Mm, Mt, lanes.side};
    if ((rc = keds_rowstats_cast_ex(x, t.h, 1, (float*)t.st1, M, w, st))) return rc;
    if ((rc = keds_quantize_mxfp8(x, 0, Mm, w, Mm, t.xq, t.xs, st))) return rc;
    if ((rc = lanes.to_side())) return rc;
    for (int l = 0; l < p->layers; ++l) {
        const keds_block_params& k = p->blocks[l];
        const bool last = l == p->layers - 1;
        if ((rc = keds_gemm_mxfp8_ex(t.xq, t.xs, Mm, k.qkv_q8, k.qkv_s8, 3 * w, k.qkv_bc8, t.qkv, Mm, 3 * w, w,
                                     KEDS_FP8_EPI_LN_BIAS_BF16, (float*)t.st1, (float*)t.st2, nullptr, nullptr, 0, st)))
            return rc;
        if (Mt && (rc = qkv_rows(t, k, w, rem))) return rc;
        if ((rc = lanes.to_main())) return rc;
        if (last && last_rows) return rows_tail(p, k, t, x, t.h, B, last_rows, st);
        if (last && p->last_cls_only) return cls_rows_tail(p, k, t, x, t.h, B, st);
        // attention writes its output as MXFP8 for the full-tile rows and as bf16 for the remainder rows
        if ((rc = lanes.attention_then_side([&] { return keds_attention_mx(t.qkv, t.att, B, S, p->heads, p->causal, S, t.aq, t.as, Mm, st); })))
            return rc;
        if ((rc = keds_gemm_mxfp8_ex(t.aq, t.as, Mm, k.out_q8, k.out_s8, w, k.out_b, t.h, Mm, w, w,
                                     KEDS_FP8_EPI_RESID_STATS_MX_H, (float*)t.st2, nullptr, t.xq, t.xs, Mm, st)))
            return rc;
        if (Mt && (rc = out_rows(t, k, w, rem))) return rc;
        if ((rc = keds_gemm_mxfp8_ex(t.xq, t.xs, Mm, k.fc_q8, k.fc_s8, 4 * w, k.fc_bc8, nullptr, Mm, 4 * w, w,
                                     KEDS_FP8_EPI_LN_QGELU_MX, (float*)t.st2, (float*)t.st1, t.hq, t.hs, Mm, st)))
            return rc;
        if (Mt && (rc = fc_rows(t, k, w, rem))) return rc;
        if ((rc = keds_gemm_mxfp8_ex(t.hq, t.hs, Mm, k.proj_q8, k.proj_s8, w, k.proj_b, t.h, Mm, w, 4 * w,
                                     KEDS_FP8_EPI_RESID_STATS_MX_H, (float*)t.st1, nullptr, t.xq, t.xs, Mm, st)))
            return rc;
        if (Mt && (rc = proj_rows(t, k, w, last, rem))) return rc;
    }
    if ((rc = lanes.to_main())) return rc;
    return keds_cast_rows_f16_f32_impl(t.h, x, M, w, w, st);       // the caller reads x in fp32
}

// allow_fill: the caller's x holds pad_rows(B * seq) rows (keds_vit_run / keds_text_run carve it so); the public
// keds_tower_forward promises only a multiple of 128 and keeps the two-lane scheme
// last_rows (device int32 [B], nullable): the one row of every sample that is read after the last block (rows_tail): on
// return x[b] (row b of the first B rows) holds that row of sample b; without it x holds every row as before
// pk (nullable): packed rows -- M = pk->rows <= B * seq, the workspace is carved for B * seq
int tower_forward(const keds_tower_params* p, float* x, int B, void* ws, hipStream_t st, bool allow_fill,
                  const int32_t* last_rows = nullptr, const PackedRows* pk = nullptr) {
    const int w = p->width, S = p->seq;
    const int M = pk ? pk->rows : B * S;
    TowerWs t = carve_tower(ws, w, S, B);
    KedsSplitKScope splitk(t.splitk, KEDS_SPLITK_BYTES);   // this call's GEMMs split K into this call's scratch only
    int rc;
    // LayerNorm folded into the GEMMs (keds_hip.h, KEDS_EPI_LN_*): t.h holds the bf16 copy of the residual stream,
    // st1 / st2 the {sum, sum sq} of its rows as seen by ln_1 / ln_2.  Each LN-consuming GEMM also clears the
    // statistics buffer the next producer accumulates into, so no memset sits between the launches.
    bool folded = true, have8 = true;
    for (int l = 0; l < p->layers; ++l) {
        const keds_block_params& k = p->blocks[l];
        folded = folded && k.qkv_wf && k.fc_wf && k.qkv_bc && k.fc_bc;
        have8 = have8 && k.qkv_q8 && k.out_q8 && k.fc_q8 && k.proj_q8 && k.qkv_s8 && k.out_s8 && k.fc_s8 && k.proj_s8 &&
                k.qkv_bc8 && k.fc_bc8;
    }
    const int Mm = M / 256 * 256;                  // rows in full 256-row tiles: MXFP8 GEMMs; the rest stays on the bf16 kernels
    if (p->fp8 && !(folded && have8 && w % 256 == 0)) {
        keds_set_error("keds_tower_forward: fp8 needs the folded and MXFP8 weights and width %% 256 == 0");
        return KEDS_E_ARG;
    }
    const bool fp8 = p->fp8 && Mm > 0;             // fewer than 256 rows: everything is "remainder rows" (bf16 kernels)
    if (pk && (p->fp8 || !p->causal || !last_rows)) {
        keds_set_error("keds_tower_forward: packed rows need a causal bf16 tower with a read-out row per sample");
        return KEDS_E_ARG;
    }
    if (pk && pk->rows > pk->valid &&
        hipMemsetAsync(t.att + (size_t)pk->valid * w, 0, (size_t)(pk->rows - pk->valid) * w * sizeof(bf16_t), st) != hipSuccess) {
        keds_set_error("keds_tower_forward: packed rows: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    auto attention = [&](hipStream_t s_) {           // the block's attention on all samples
        return pk ? keds_attention_packed(t.qkv, t.att, B, S, pk->off, p->heads, p->causal, s_)
                  : keds_attention(t.qkv, t.att, B, S, p->heads, p->causal, s_);
    };
    if (fp8) return tower_forward_fp8(p, x, B, t, Mm, st, last_rows);
    if (folded) {
        // When every GEMM of the block would split into full 256-row tiles + a remainder launch anyway, the remainder
        // rows become their own chain on the side lane; otherwise one span covers all rows.
        const int fill = allow_fill ? tower_fill_rows(M, w) : 0;
        const bool two = !fill && bf16_rows_split(M, w);
        RowLanes lanes;
        if ((rc = lanes.init(st, two))) return rc;
        const RowSpan body{0, lanes.split ? Mm : M + fill, st};
        const RowSpan rem{(size_t)Mm, lanes.split ? M - Mm : 0, lanes.side};
        if (fill) {       // filler rows: a copy of the first rows of the stream; their attention output is zero in every block
            if (hipMemcpyAsync(x + (size_t)M * w, x, (size_t)fill * w * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemsetAsync(t.att + (size_t)M * w, 0, (size_t)fill * w * sizeof(bf16_t), st) != hipSuccess) {
                keds_set_error("keds_tower_forward: filler rows: %s", hipGetErrorString(hipGetLastError()));
                return KEDS_E_LAUNCH;
            }
        }
        if ((rc = keds_rowstats_cast_ex(x, t.h, 1, (float*)t.st1, M + fill, w, st))) return rc;
        if ((rc = lanes.to_side())) return rc;
        // Round-4 experiment (KEDS_TAIL_ATTN=1; off): the samples whose rows reach into the remainder ("tail" samples: b >= Mm / S;
        // one of 128 at ViT-L/14) get their attention ON THE SIDE LANE.  A kernel trace shows the main lane waiting 9 us per block
        // on average at its join in front of the attention launch (profiles/r04_trace_gaps.txt): the side chain's last launch
        // (the remainder rows' in_proj) cannot get a CU while the main lane's PERSISTENT in_proj holds all of them, so it runs
        // behind it.  Here the main lane's attention covers the samples that live in full tiles only and needs no join; the side
        // lane waits for the main in_proj, runs the tail samples' attention beside the main attention launch, and the main lane
        // waits for THAT in front of out-proj.  Bit-identical -- and neutral: the side launches (128 KiB of LDS each) do not fit
        // beside an attention workgroup either, so the wait only moves.
        const int b_tail = lanes.split && !pk && tail_attention_on_side() ? Mm / S : B;      // first tail sample (B: none)
        const bool tail_side = b_tail > 0 && b_tail < B;
        // Round 5: with the tail samples' attention on the side lane the WHOLE remainder chain of a block (out-proj, c_fc, c_proj,
        // the next in_proj of the remainder rows) starts beside the main ATTENTION launch instead of beside the main GEMMs, whose
        // one-wave-per-SIMD workgroups own every register of their CU -- provided its workgroups fit next to a resident attention
        // workgroup (74 KiB of LDS, half the CU's registers): the 64 KiB kernel form (keds_gemm_small_lds) instead of the 128 KiB one.
        struct SmallLds {
            bool on;
            explicit SmallLds(bool o) : on(o) { if (on) keds_gemm_small_lds(1); }
            ~SmallLds() { if (on) keds_gemm_small_lds(0); }
        } small_lds(tail_side);
        for (int l = 0; l < p->layers; ++l) {
            const keds_block_params& k = p->blocks[l];
            const bool last = l == p->layers - 1;
            if ((rc = qkv_rows(t, k, w, body))) return rc;
            if (rem.n && (rc = qkv_rows(t, k, w, rem))) return rc;
            if (last && (p->last_cls_only || last_rows)) {
                if ((rc = lanes.to_main())) return rc;
                return last_rows ? rows_tail(p, k, t, x, t.h, B, last_rows, st, pk) : cls_rows_tail(p, k, t, x, t.h, B, st);
            }
            if (tail_side) {
                if ((rc = lanes.to_side())) return rc;               // side: behind the main in_proj
                if ((rc = keds_attention(t.qkv, t.att, b_tail, S, p->heads, p->causal, st))) return rc;
                if ((rc = keds_attention(t.qkv + (size_t)b_tail * S * 3 * w, t.att + (size_t)b_tail * S * w, B - b_tail, S, p->heads,
                                         p->causal, lanes.side)))
                    return rc;
                if ((rc = lanes.to_main())) return rc;               // main: behind the tail samples' attention
            } else {
                if ((rc = lanes.to_main())) return rc;
                if ((rc = lanes.attention_then_side([&] { return attention(st); }))) return rc;
            }
            if ((rc = out_rows(t, k, w, body))) return rc;
            if (rem.n && (rc = out_rows(t, k, w, rem))) return rc;
            if ((rc = fc_rows(t, k, w, body))) return rc;
            if (rem.n && (rc = fc_rows(t, k, w, rem))) return rc;
            if ((rc = proj_rows(t, k, w, last, body))) return rc;
            if (rem.n && (rc = proj_rows(t, k, w, last, rem))) return rc;
        }
        if ((rc = lanes.to_main())) return rc;
        return keds_cast_rows_f16_f32_impl(t.h, x, M, w, w, st);   // the caller reads x in fp32
    }
    // KEDS_DETERMINISTIC / unfolded weights: stand-alone LayerNorm launches, plain bias epilogues
    for (int l = 0; l < p->layers; ++l) {
        const keds_block_params& k = p->blocks[l];
        const bool last = l == p->layers - 1;
        if ((rc = keds_layernorm_impl(x, w, nullptr, 1, k.ln1_g, k.ln1_b, t.h, 0, M, w, st))) return rc;
        if ((rc = keds_gemm_bt(t.h, k.qkv_w, k.qkv_b, t.qkv, M, 3 * w, w, KEDS_EPI_BIAS_BF16, nullptr, 0, st))) return rc;
        if (last && last_rows) return rows_tail(p, k, t, x, nullptr, B, last_rows, st, pk);
        if (last && p->last_cls_only) return cls_rows_tail(p, k, t, x, nullptr, B, st);
        if ((rc = attention(st))) return rc;
        if ((rc = keds_gemm_bt(t.att, k.out_w, k.out_b, x, M, w, w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st))) return rc;
        if ((rc = keds_layernorm_impl(x, w, nullptr, 1, k.ln2_g, k.ln2_b, t.h, 0, M, w, st))) return rc;
        if ((rc = keds_gemm_bt(t.h, k.fc_w, k.fc_b, t.hid, M, 4 * w, w, KEDS_EPI_BIAS_QGELU_BF16, nullptr, 0, st))) return rc;
        if ((rc = keds_gemm_bt(t.hid, k.proj_w, k.proj_b, x, M, w, 4 * w, KEDS_EPI_BIAS_RESID_F32, nullptr, 0, st))) return rc;
    }
    return KEDS_OK;
}

int check_tower(const keds_tower_params* p, const char* who) {
    if (!p || !p->blocks || p->layers <= 0) {
        keds_set_error("%s: missing tower parameters", who);
        return KEDS_E_ARG;
    }
    if (p->width % 128 != 0 || p->width != p->heads * 64) {
        keds_set_error("%s: width %d must be heads*64 and a multiple of 128", who, p->width);
        return KEDS_E_ARG;
    }
    if (p->seq < 1 || p->seq > 288) {
        keds_set_error("%s: sequence length %d unsupported", who, p->seq);
        return KEDS_E_ARG;
    }
    return KEDS_OK;
}

}  // namespace

extern "C" size_t keds_tower_workspace_bytes(int width, int seq, int B) {
    return carve_tower(nullptr, width, seq, B).bytes;
}

extern "C" int keds_tower_fill_enable(int on) {       // run-time override of KEDS_TOWER_FILL (A/B, tests): experiment build only
#ifdef KEDS_EXPERIMENTS
    g_tower_fill = on ? 1 : 0;
    return KEDS_OK;
#else
    if (!on) return KEDS_OK;
    keds_set_error("keds_tower_fill_enable: the filler-row tower is an experiment (round 3, slower): build with make EXTRA=-DKEDS_EXPERIMENTS");
    return KEDS_E_ARG;
#endif
}

extern "C" int keds_tower_side_rows(int width, int seq, int B, int fp8) {
    if (width <= 0 || seq <= 0 || B <= 0) return 0;
    const int M = B * seq, Mt = M % 256;
    if (!fp8 && tower_fill_rows(M, width)) return 0;       // the ragged tile runs as a full one on the caller's stream
    const bool split = fp8 ? (M >= 256 && Mt > 0) : bf16_rows_split(M, width);
    return split && keds_side_lane_enabled() ? Mt : 0;    // only a query: no stream is created here (no GPU needed)
}

extern "C" int keds_tower_forward(const keds_tower_params* p, float* x, int B, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    int rc = check_tower(p, "keds_tower_forward");
    if (rc) return rc;
    KEDS_REQUIRE(x && workspace && B > 0, "keds_tower_forward: bad argument");
    if (workspace_bytes < keds_tower_workspace_bytes_ex(p, B)) {
        keds_set_error("keds_tower_forward: workspace too small");
        return KEDS_E_WORKSPACE;
    }
    if (p->f32) return keds_tower_forward_f32(p, x, B, workspace, (hipStream_t)stream);
    return tower_forward(p, x, B, workspace, (hipStream_t)stream, false);
}

extern "C" size_t keds_tower_workspace_bytes_ex(const keds_tower_params* p, int B) {
    if (!p || B <= 0) return 0;
    return p->f32 ? keds_tower_f32_workspace_bytes(p->width, p->seq, B) : keds_tower_workspace_bytes(p->width, p->seq, B);
}

// ---- ViT -------------------------------------------------------------------------------------
namespace {
struct VitWs {
    float* x;      // [Mp, w] fp32 residual stream
    char* tower;   // tower scratch (the im2col matrix aliases its head)
    char* ro;      // read-out scratch
    size_t bytes;
};
VitWs carve_vit(const keds_vit_params* p, int B, void* ws) {
    VitWs v;
    char* base = (char*)ws;
    const int w = p->tower.width, S = p->tower.seq;
    const size_t Mp = pad_rows((size_t)B * S);
    const size_t xb = keds_align_up(Mp * w * sizeof(float), 256);
    size_t tb = keds_tower_workspace_bytes_ex(&p->tower, B);
    const size_t colb = keds_align_up(pad_rows((size_t)B * (S - 1)) * p->kpad * (p->tower.f32 ? 4 : 2), 256);
    if (colb > tb) tb = colb;
    const size_t rb = keds_align_up(p->tower.f32 ? keds_readout_f32_workspace_bytes(B, w) : keds_readout_workspace_bytes(B, w), 256);
    v.x = (float*)base;
    v.tower = base ? base + xb : nullptr;
    v.ro = base ? base + xb + tb : nullptr;
    v.bytes = xb + tb + rb;
    return v;
}
}  // namespace

extern "C" size_t keds_vit_workspace_bytes(const keds_vit_params* p, int B) {
    if (!p || B <= 0) return 0;
    return carve_vit(p, B, nullptr).bytes;
}

extern "C" int keds_vit_run(const keds_vit_params* p, const float* image, int B, float* out, int normalize,
                                void* workspace, size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(p && image && out && workspace && B > 0, "keds_vit_run: bad argument");
    int rc = check_tower(&p->tower, "keds_vit_run");
    if (rc) return rc;
    const int w = p->tower.width, S = p->tower.seq, G = S - 1;
    const int g = p->resolution / p->patch;
    KEDS_REQUIRE(p->resolution % p->patch == 0 && g * g == G, "keds_vit_run: seq must be (res/patch)^2 + 1");
    KEDS_REQUIRE(p->kpad % 64 == 0 && p->kpad >= 3 * p->patch * p->patch, "keds_vit_run: bad kpad");
    KEDS_REQUIRE(p->embed_dim % 128 == 0, "keds_vit_run: embed_dim must be a multiple of 128");
    VitWs v = carve_vit(p, B, workspace);
    if (workspace_bytes < v.bytes) {
        keds_set_error("keds_vit_run: workspace %zu < %zu", workspace_bytes, v.bytes);
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    void* col = v.tower;
    if (p->tower.f32) {
        // the fp32-accurate flow (f32path.hip): every weight pointer of the structs is an fp32 array
        if ((rc = keds_im2col_f32(image, (float*)col, B, p->resolution, p->patch, p->kpad, stream))) return rc;
        if ((rc = keds_gemm_f32((const float*)col, p->kpad, (const float*)p->conv_w, nullptr, v.x, w, B * G, w, p->kpad,
                                4 /* patch rows + positional embedding */, p->pos_emb, G, stream)))
            return rc;
        if ((rc = keds_cls_rows_impl(v.x, p->class_emb, p->pos_emb, B, S, w, st))) return rc;
        if ((rc = keds_layernorm_impl(v.x, w, nullptr, 1, p->ln_pre_g, p->ln_pre_b, v.x, 1, B * S, w, st))) return rc;
        if ((rc = keds_tower_forward_f32(&p->tower, v.x, B, v.tower, st))) return rc;
        return keds_readout_f32(v.x, S, nullptr, p->ln_post_g, p->ln_post_b, (const float*)p->proj_t, out, B, w, p->embed_dim,
                                normalize, v.ro, st);
    }
    if ((rc = keds_im2col(image, col, B, p->resolution, p->patch, p->kpad, stream))) return rc;
    if ((rc = keds_gemm_bt(col, p->conv_w, nullptr, v.x, B * G, w, p->kpad, KEDS_EPI_PATCH_F32, p->pos_emb, G, stream)))
        return rc;
    if ((rc = keds_cls_rows_impl(v.x, p->class_emb, p->pos_emb, B, S, w, st))) return rc;
    // ln_pre in place (each wave holds its whole row in registers before it stores)
    if ((rc = keds_layernorm_impl(v.x, w, nullptr, 1, p->ln_pre_g, p->ln_pre_b, v.x, 1, B * S, w, st))) return rc;
    if ((rc = tower_forward(&p->tower, v.x, B, v.tower, st, true))) return rc;
    return keds_readout(v.x, S, nullptr, p->ln_post_g, p->ln_post_b, p->proj_t, out, B, w, p->embed_dim, normalize, v.ro,
                        keds_readout_workspace_bytes(B, w), stream);
}

// ---- text ------------------------------------------------------------------------------------
namespace {
struct TextWs {
    float* x;
    char* tower;
    char* ro;
    size_t bytes;
};
TextWs carve_text_cols(const keds_text_params* p, int B, int S, void* ws) {        // the layout for S columns per sequence
    TextWs v;
    char* base = (char*)ws;
    const int w = p->tower.width;
    const size_t Mp = pad_rows((size_t)B * S);
    const size_t xb = keds_align_up(Mp * w * sizeof(float), 256);
    keds_tower_params tp = p->tower;
    tp.seq = S;
    const size_t tb = keds_tower_workspace_bytes_ex(&tp, B);
    const size_t rb = keds_align_up(p->tower.f32 ? keds_readout_f32_workspace_bytes(B, w) : keds_readout_workspace_bytes(B, w), 256);
    v.x = (float*)base;
    v.tower = base ? base + xb : nullptr;
    v.ro = base ? base + xb + tb : nullptr;
    v.bytes = xb + tb + rb;
    return v;
}
TextWs carve_text(const keds_text_params* p, int B, void* ws) { return carve_text_cols(p, B, p->tower.seq, ws); }
}  // namespace

extern "C" size_t keds_text_workspace_bytes(const keds_text_params* p, int B) {
    if (!p || B <= 0) return 0;
    return carve_text(p, B, nullptr).bytes;
}

// KEDS_TEXT_TRIM=0 in the environment: the round-4 flow (all L columns through all blocks) for an A/B or a bisect
static bool text_trim_on() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("KEDS_TEXT_TRIM");
        v = !(e && e[0] == '0');
    }
    return v != 0;
}
static int g_text_trim = -1;                     // run-time override (keds_text_trim_enable: tests, A/B); -1: the environment
extern "C" int keds_text_trim_enable(int on) {   // 0 off, 1 on, 2 the column cut only, 3 the read-out-row tail only
    g_text_trim = on < 0 || on > 3 ? -1 : on;
    return KEDS_OK;
}
extern "C" int keds_text_trim_mode(void) {       // the flow in force: 1 = cut + read-out-row tail (the default; what packed rows build on)
    return g_text_trim < 0 ? (text_trim_on() ? 1 : 0) : g_text_trim;
}

// Work that cannot reach the read-out is not done (round 5):
//  * the mask is causal (model.py:543-549), so columns to the right of the last read-out column change no row that is read:
//    `seq_used` (host-known: max(readout_row) + 1; 0 = unknown, all L columns) cuts the sequence there for the embedding, all
//    blocks and the workspace layout ([B, columns, w]);
//  * the last block's out-proj, ln_2 and MLP run on the B read-out rows only (rows_tail), as the ViT's do on the CLS rows.
extern "C" int keds_text_run_ex(const keds_text_params* p, const int32_t* tokens, const int32_t* readout_row,
                                const float* img_tokens, int n_tok, int insert_col, int B, int seq_used, float* out,
                                int normalize, void* workspace, size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(p && tokens && readout_row && out && workspace && B > 0, "keds_text_run: bad argument");
    int rc = check_tower(&p->tower, "keds_text_run");
    if (rc) return rc;
    KEDS_REQUIRE(p->embed_dim % 128 == 0, "keds_text_run: embed_dim must be a multiple of 128");
    KEDS_REQUIRE(seq_used >= 0 && seq_used <= p->tower.seq, "keds_text_run: seq_used %d outside [0, %d]", seq_used, p->tower.seq);
    TextWs v = carve_text(p, B, workspace);
    if (workspace_bytes < v.bytes) {
        keds_set_error("keds_text_run: workspace %zu < %zu", workspace_bytes, v.bytes);
        return KEDS_E_WORKSPACE;
    }
    const int w = p->tower.width, L = p->tower.seq;
    const int mode = g_text_trim < 0 ? (text_trim_on() ? 1 : 0) : g_text_trim;
    const bool cut = mode == 1 || mode == 2, trim = mode == 1 || mode == 3;
    int Lx = L;
    // (no rounding of the cut to whole row tiles: measured at B = 128 -- tools/text_shapes.py, profiles/r05_text_shapes.txt --
    // the four GEMMs of a block take 161 us at 43 columns, 170 at 44, 172-174 at 46-48: at these sizes the 128 x 128 kernel's
    // rounds of 512 workgroups decide, and fewer rows are never slower)
    if (cut && seq_used > 0 && seq_used < L && p->tower.causal) Lx = seq_used;
    keds_tower_params tp = p->tower;              // the tower on the columns that are computed
    tp.seq = Lx;
    v = carve_text_cols(p, B, Lx, workspace);     // (never larger than the full layout the size check above covers)
    if ((rc = keds_embed_tokens_impl(tokens, p->token_emb, p->pos_emb, img_tokens, n_tok, insert_col, v.x, B, L, Lx, w, stream)))
        return rc;
    const int32_t* last_rows = trim ? readout_row : nullptr;
    if (tp.f32) {
        if ((rc = keds_tower_forward_f32(&tp, v.x, B, v.tower, (hipStream_t)stream, last_rows))) return rc;
        return keds_readout_f32(v.x, last_rows ? 1 : Lx, last_rows ? nullptr : readout_row, p->ln_final_g, p->ln_final_b,
                                (const float*)p->proj_t, out, B, w, p->embed_dim, normalize, v.ro, (hipStream_t)stream);
    }
    if ((rc = tower_forward(&tp, v.x, B, v.tower, (hipStream_t)stream, true, last_rows))) return rc;
    return keds_readout(v.x, last_rows ? 1 : Lx, last_rows ? nullptr : readout_row, p->ln_final_g, p->ln_final_b, p->proj_t, out,
                        B, w, p->embed_dim, normalize, v.ro, keds_readout_workspace_bytes(B, w), stream);
}

// The same on PACKED rows (round 6).  Captions end at different columns and under the causal mask (model.py:543-549) a column to
// the right of a caption's own read-out column reaches nothing that is read of THAT caption: sample b needs columns
// [0, len_b) only, len_b = its read-out column + 1.  keds_text_run_ex cuts every sample at the batch's longest caption (B x max len
// rows); here sample b owns rows [seq_off[b], seq_off[b + 1]) and the tower runs sum(len_b) rows -- at bench.py's captions (read-out
// columns 10 .. 42) 6.9 k instead of 11 k rows for the dual workload's 2B-row pass, whole rounds of workgroups fewer in every GEMM.
//   seq_off         device int32 [B + 1]: 0 = seq_off[0] <= ... <= seq_off[B] = rows_total, 1 <= len_b <= seq_max
//   readout_global  device int32 [B]: seq_off[b] + the read-out column of sample b (a row outside [0, rows_total) comes out as NaN)
// Causal bf16 / fp32 / fp32x3 towers with the read-out-row tail (the default flow); anything else (the MXFP8 tower,
// KEDS_TEXT_TRIM=0) is the caller's to route through keds_text_run_ex.
extern "C" int keds_text_run_packed(const keds_text_params* p, const int32_t* tokens, const int32_t* seq_off,
                                    const int32_t* readout_global, int rows_total, int seq_max, const float* img_tokens, int n_tok,
                                    int insert_col, int B, float* out, int normalize, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    KEDS_REQUIRE(p && tokens && seq_off && readout_global && out && workspace && B > 0, "keds_text_run_packed: bad argument");
    int rc = check_tower(&p->tower, "keds_text_run_packed");
    if (rc) return rc;
    KEDS_REQUIRE(p->embed_dim % 128 == 0, "keds_text_run_packed: embed_dim must be a multiple of 128");
    const int w = p->tower.width, L = p->tower.seq;
    KEDS_REQUIRE(p->tower.causal && !p->tower.fp8, "keds_text_run_packed: a causal bf16 / fp32 / fp32x3 tower (the MXFP8 tower keeps the rectangular layout)");
    KEDS_REQUIRE(keds_text_trim_mode() == 1, "keds_text_run_packed: the A/B flows of keds_text_trim_enable / KEDS_TEXT_TRIM=0 go through keds_text_run_ex");
    KEDS_REQUIRE(seq_max >= 1 && seq_max <= L && rows_total >= B && (long long)rows_total <= (long long)B * seq_max,
                 "keds_text_run_packed: rows_total %d / seq_max %d do not fit B = %d sequences of <= %d columns", rows_total, seq_max, B, L);
    TextWs v = carve_text(p, B, workspace);
    if (workspace_bytes < v.bytes) {
        keds_set_error("keds_text_run_packed: workspace %zu < %zu", workspace_bytes, v.bytes);
        return KEDS_E_WORKSPACE;
    }
    keds_tower_params tp = p->tower;
    tp.seq = seq_max;
    v = carve_text_cols(p, B, seq_max, workspace);
    // the tower runs WHOLE 256-row tiles: the rows between rows_total and the next multiple of 256 are zero rows that belong to no
    // sample (no attention reads them, nothing gathers them).  A ragged last tile would send every GEMM of every block through the
    // remainder-row launches and the stricter fill rule of the 256 x 256 kernels (gemm.hip, big_tiles_ok).
    int rows_run = (rows_total + 255) / 256 * 256;
    if ((size_t)rows_run > pad_rows((size_t)B * seq_max)) rows_run = rows_total;
    if (rows_run > rows_total &&
        hipMemsetAsync(v.x + (size_t)rows_total * w, 0, (size_t)(rows_run - rows_total) * w * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        keds_set_error("keds_text_run_packed: %s", hipGetErrorString(hipGetLastError()));
        return KEDS_E_LAUNCH;
    }
    if ((rc = keds_embed_tokens_impl(tokens, p->token_emb, p->pos_emb, img_tokens, n_tok, insert_col, v.x, B, L, seq_max, w, stream, seq_off)))
        return rc;
    const PackedRows pk{seq_off, rows_run, rows_total};
    if (tp.f32) {
        if ((rc = keds_tower_forward_f32(&tp, v.x, B, v.tower, (hipStream_t)stream, readout_global, &pk))) return rc;
        return keds_readout_f32(v.x, 1, nullptr, p->ln_final_g, p->ln_final_b, (const float*)p->proj_t, out, B, w, p->embed_dim, normalize,
                                v.ro, (hipStream_t)stream);
    }
    if ((rc = tower_forward(&tp, v.x, B, v.tower, (hipStream_t)stream, false, readout_global, &pk))) return rc;
    return keds_readout(v.x, 1, nullptr, p->ln_final_g, p->ln_final_b, p->proj_t, out, B, w, p->embed_dim, normalize, v.ro,
                        keds_readout_workspace_bytes(B, w), stream);
}

extern "C" int keds_text_run(const keds_text_params* p, const int32_t* tokens, const int32_t* readout_row,
                                 const float* img_tokens, int n_tok, int insert_col, int B, float* out, int normalize,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return keds_text_run_ex(p, tokens, readout_row, img_tokens, n_tok, insert_col, B, 0, out, normalize, workspace,
                            workspace_bytes, stream);
}
