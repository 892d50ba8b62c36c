/*
 * keds_hip.h -- C ABI of libkeds_hip.so: the MI355X (gfx950) retrieval hot path of KEDs.
 *
 * The reference (suoych/KEDs) has no FFI seam: its seam is the Python object API
 * (SURVEY.md section 8b).  The Python facade in keds_amd/ keeps that API and binds the
 * functions below with ctypes; each entry point names the reference code it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never
 *     allocates, frees or synchronises, so every call is capturable in a hipGraph;
 *   - all launches go to the `stream` argument (a hipStream_t passed as void*);
 *   - return 0 on success, a negative KEDS_E_* code otherwise; keds_last_error()
 *     returns a thread-local message for the last failure;
 *   - bf16 tensors are raw 16-bit patterns (uint16_t), row-major, innermost dim contiguous.
 */
#ifndef KEDS_HIP_H
#define KEDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KEDS_OK 0
#define KEDS_E_ARG (-1)      /* bad argument / unsupported shape */
#define KEDS_E_LAUNCH (-2)   /* HIP launch or runtime error */
#define KEDS_E_WORKSPACE (-3)/* workspace too small */

#define KEDS_ABI_VERSION 8        /* 8 (round 6): keds_gemm_x3 takes w_exp, keds_split_f16_weight, keds_block_params.x3_exp, keds_text_run_packed / keds_attention_packed; keds_gemm_duo_enable left the product */

int keds_abi_version(void);
/* compiler flags of this build beyond the Makefile's defaults ("" for the product build; `make EXTRA="-D..."` variants of the
 * A/B and timing-only scripts record theirs here, and the evidence digest of keds_amd/_lib.py includes it) */
const char* keds_build_flags(void);
const char* keds_last_error(void);

/* ---- numerics guard of the fast tower flow ---------------------------------------------------------------------
 * The default tower flow keeps the residual stream in fp16 and evaluates LayerNorm inside the GEMMs on un-centred rows.
 * That is accurate while |row mean| / row std stays small (CLIP: a few) and the stream stays inside the fp16 range.
 * While a device flag is registered for the calling thread, every LayerNorm-consuming GEMM it enqueues sets *flag = 1 if
 * it meets a row with |mean| / std > 32 or non-finite statistics; the host reads the flag after the pass and re-runs it
 * on the fp32-stream flow (stand-alone LayerNorm: unfolded weights) -- keds_amd.CLIP does so automatically.
 * nullptr unregisters.  The flag is only ever set, never cleared, by the library. */
int keds_numerics_guard_set(int32_t* device_flag);

/* ---- profiling of kernel classes with hipEvents on the launch stream -------------
 * bench.py switches this on over its timed region to get the dominant kernel's average
 * launch duration (roofline.achieved).  Classes: see KEDS_PROF_*. */
#define KEDS_PROF_GEMM 0
#define KEDS_PROF_ATTN 1
#define KEDS_PROF_SCAN 2
#define KEDS_PROF_LN 3
#define KEDS_PROF_OTHER 4
#define KEDS_PROF_NCLASS 5
int keds_prof_enable(int on);                 /* 0 off; 1: event pair around every launch; else bit (k+1) = class k */
int keds_prof_reset(void);
int keds_prof_read(int klass, double* total_ms, int64_t* launches);  /* synchronises the events */
/* algorithmic work of the launches that carried event pairs since the last reset: 2*M*N*K flops per GEMM launch
 * (KEDS_PROF_GEMM; a patch-embedding GEMM counts its zero-padded K), scan-image bytes per scan launch (KEDS_PROF_SCAN) */
int keds_prof_read_work(int klass, double* units);

/* =====================================================================================
 * 1. Similarity + top-k  (replaces faiss.IndexFlatL2 + index_cpu_to_all_gpus + .add/.search,
 *    src/eval_retrieval.py:289-298, src/eval_utils.py:169,177; brute-force twin
 *    src/trainer.py:246-257)
 * ===================================================================================== */
#define KEDS_METRIC_L2 0   /* D = ||q - x||^2 ascending (IndexFlatL2) */
#define KEDS_METRIC_IP 1   /* D = q . x descending (trainer.py:246-249) */

#define KEDS_SCAN_STAGE_KEYS 32     /* keys per packed stage */
#define KEDS_SCAN_MAX_QUERIES 128   /* queries per scan launch (one query block) */
#define KEDS_SCAN_LIST 16           /* per-lane exact list depth */
#define KEDS_SCAN_CAND 64           /* candidates re-ranked in fp32 per query for k <= 16 (256 for larger k) */
#define KEDS_SCAN_MAX_K 128         /* largest k of keds_index_search_packed */

/* timing-only ablation hook of the D=768 scan kernel (0 = product path) */
int keds_scan_debug(int variant);
/* diagnostic: device buffer (>= 8 uint64 per query) that merge_pairs_kernel fills with s_memtime stamps of its phases
 * (tools/stamp_merge.py); nullptr switches them off */
int keds_merge_stamp_buffer(void* buf);

/* bytes of the packed bf16 scan image for n rows of dimension dim (dim % 128 == 0); the image ends with a 128-byte
 * trailer holding max ||bf16(x)||, max ||x - bf16(x)||, max ||x|| over the rows (the search certificate's bounds) */
size_t keds_index_packed_bytes(int64_t n, int dim);

/* `.add`: build the scan image from fp32 rows [n, dim] (device).  The image is the exact
 * LDS layout the scan kernel streams: per stage of 32 keys, XOR-swizzled bf16 rows followed
 * by 32 fp32 bias terms (-0.5*||x||^2 for L2, 0 for IP; -inf for rows >= n). */
int keds_index_pack(const float* db, int64_t n, int dim, int metric, void* packed, void* stream);   /* n <= 2,048 * 16,384 rows per shard */
/* chunked `.add`: `packed` already holds the image of rows [0, old_n) of `db` (same buffer, sized for n rows); pack rows
 * [old_n, n) only (the stage that old_n falls into is rewritten), keeping the bounds of the earlier rows. */
int keds_index_pack_append(const float* db, int64_t old_n, int64_t n, int dim, int metric, void* packed, void* stream);

/* workspace bytes for keds_index_search_packed(_ex): nq queries against n rows, top-k.  The two-argument form is sized
 * for k <= 16 over at most 4 M rows. */
size_t keds_index_search_workspace_bytes_ex(int nq, int dim, int64_t n, int k);
size_t keds_index_search_workspace_bytes(int nq, int dim);

/* `.search`: exact top-k (k <= KEDS_SCAN_MAX_K) of nq fp32 queries [nq, dim] against the index.
 *   normalize_q != 0 : L2-normalise the queries first (src/eval_utils.py:162)
 *   D [nq,k] fp32, I [nq,k] int64 (row ids offset by id_base, -1 if fewer than k rows)
 *   rows_out (nullable) [nq,k,dim] fp32 : gathered rows db[I]  (src/eval_utils.py:171-172,179-180)
 * Pipeline: bf16 MFMA scan with per-lane exact top-16 lists -> merge to 64 (k > 16: 256) candidates -> exact fp32
 * re-rank from `db` -> top-k -> CERTIFICATE per query: the k-th candidate's exact score minus a rigorous bound on the bf16
 * scan error must beat the best score any non-candidate row can have; queries that fail it (near-duplicate clusters
 * wider than the candidate list) are answered by an exact fp32 pass over all rows (IndexFlatL2 semantics in every case).
 * `db` is the fp32 matrix given to keds_index_pack.  status (nullable, device int32[2]): += {queries certified from the
 * candidates, queries answered by the exact pass}. */
int keds_index_search_packed_ex(const void* packed, const float* db, int64_t n, int dim, int metric,
                      const float* queries, int nq, int normalize_q, int k, int64_t id_base,
                      float* D, int64_t* I, float* rows_out,
                      void* workspace, size_t workspace_bytes, int32_t* status, void* stream);
int keds_index_search_packed(const void* packed, const float* db, int64_t n, int dim, int metric,
                      const float* queries, int nq, int normalize_q, int k, int64_t id_base,
                      float* D, int64_t* I, float* rows_out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* merge `parts` partial results [parts, nq, k] (each sorted) into the global top-k
 * (multi-GPU: after the all-gather of per-shard results).  Keyed on (D, I). */
int keds_topk_merge_parts(const float* D_parts, const int64_t* I_parts, int parts, int nq, int k,
                          int metric, float* D, int64_t* I, void* stream);

/* Packed exchange of a data-parallel, row-sharded search (replaces the replicated Faiss index of
 * src/eval_retrieval.py:289-298 and the two searches of src/eval_utils.py:169-183): every rank searched ITS shard for
 * the B queries of EVERY rank and holds D_p / I_p [world*B, k] (+ rows_p [world*B, k, dim], the rows of its partial
 * winners, when the caller needs the neighbours themselves).  keds_exchange_pack writes one message per peer,
 *   send[w][b][j][E] int32,  E = 3: distance bits | id low | id high;  with rows E = 4 + dim: those, a pad word, the row
 * (part w at send + w * w_stride words; w_stride >= B*k*E, a multiple of 4 with rows -- several databases share one
 * buffer by interleaving their parts).  After ONE all-to-all (block w to rank w) keds_exchange_merge turns the received
 * recv[w][b][j][E] (part w = what shard w found for MY query b) into this rank's D / I [B, k] keyed on (distance, id) --
 * the order of a single-GPU search of the whole database -- and, with rows != NULL, the winners' rows [B, k, dim].
 * world <= 64, world * k <= 4096, k <= KEDS_SCAN_MAX_K. */
int keds_exchange_pack(const float* D_p, const int64_t* I_p, const float* rows_p /* nullable */, int world, int B, int k,
                       int dim, int64_t w_stride, int32_t* send, void* stream);
int keds_exchange_merge(const int32_t* recv, int world, int B, int k, int dim, int64_t w_stride, int metric,
                        float* D, int64_t* I, float* rows /* nullable */, void* stream);

/* out[i,:] = db[idx[i],:]  (idx < 0 gives zeros) */
int keds_gather_rows(const float* db, int dim, const int64_t* idx, int64_t count, float* out, void* stream);

/* full gallery ranking for the recall metric (src/eval_utils.py:1040-1067):
 * order[q,:] = stable argsort of (1 - ref[q] . gallery[g]) ascending.  Up to 8192 gallery rows sort in LDS; larger
 * galleries (ImageNet domain-conversion targets) sort 8192-row chunks and merge the runs in global memory. */
size_t keds_rank_gallery_workspace_bytes(int nq, int ng);
int keds_rank_gallery(const float* ref, int nq, const float* gallery, int ng, int dim,
                      int32_t* order, void* workspace, size_t workspace_bytes, void* stream);

/* recall hits from a ranking with the reference image removed (eval_utils.py:1050-1065):
 * names are integer ids (basename-interned on the host).  rank_out[q] = position of target
 * id in order[q] after removing ref_id[q]; -1 if absent. */
int keds_cirr_target_rank(const int32_t* order, int nq, int ng, const int32_t* gallery_ids,
                          const int32_t* ref_ids, const int32_t* target_ids, int32_t* rank_out,
                          int32_t* counts_out /* nullable [nq,2]: reference matches, target matches */,
                          void* stream);

/* get_metrics_imgnet (src/eval_utils.py:1090-1134): hits_out[q,j] = number of gallery items among the first
 * ks[j] of order[q,:] whose label equals query_labels[q]; total_out[q] = that count over the whole gallery.
 * ks is a HOST array of nk <= 8 cut-offs. */
int keds_label_hits(const int32_t* order, int nq, int ng, const int32_t* gallery_labels,
                    const int32_t* query_labels, const int32_t* ks, int nk, int32_t* hits_out,
                    int32_t* total_out, void* stream);

/* =====================================================================================
 * 2. Encoder primitives (src/model/model.py:291-326).  bf16 GEMM inputs, fp32 accumulate,
 *    fp32 residual stream, fp32 LayerNorm / softmax statistics.
 * ===================================================================================== */
#define KEDS_EPI_BIAS_BF16 0        /* out bf16 = acc + bias */
#define KEDS_EPI_BIAS_QGELU_BF16 1  /* out bf16 = qgelu(acc + bias)          (model.py:300-302,311-315) */
#define KEDS_EPI_BIAS_RELU_BF16 2   /* out bf16 = relu(acc + bias)           (IM2TEXT, model.py:112-116) */
#define KEDS_EPI_BIAS_RESID_F32 3   /* out f32 += acc + bias (in place)      (model.py:324-325) */
#define KEDS_EPI_BIAS_F32 4         /* out f32 = acc + bias */
#define KEDS_EPI_PATCH_F32 5        /* out f32 row (m/G)*(G+1)+1+m%G = acc + aux[1+m%G]  (model.py:394-398) */
/* LayerNorm folded into the neighbouring GEMMs (ln_1 -> in_proj, ln_2 -> c_fc, model.py:305-326): the producer of the
 * residual stream also emits a bf16 copy of it and per-row {sum, sum of squares}; the consumer multiplies the UN-normalised
 * rows with W.diag(gamma) and finishes LayerNorm per output element:
 *     LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (b + W beta),   W' = bf16(W diag(gamma))
 * so the 2 x layers LayerNorm passes over the residual stream disappear (keds_fold_layernorm builds W', colsum, b'). */
#define KEDS_EPI_LN_BIAS_BF16 6     /* out bf16 = rstd[m] (acc - mean[m] csum[n]) + bias'[n];  bias = [bias'(N) | csum(N)],
                                       aux = row statistics [M,2] of A's rows (LayerNorm width = K): {sum, sum of squares} as
                                       64-bit fixed point (value * 2^28; integer atomics: order independent, deterministic),
                                       aux2 (nullable) = statistics buffer that is ZEROED for this launch's rows */
#define KEDS_EPI_LN_QGELU_BF16 7    /* same, then QuickGELU */
#define KEDS_EPI_RESID_STATS_F32 8  /* out f32 += acc + bias (in place); aux2 = bf16 copy [M,N] of the new rows;
                                       aux = statistics [M,2] (64-bit fixed point) += {sum, sum sq} of the new rows (zeroed before) */
#define KEDS_EPI_RESID_STATS_F16 9  /* out f16 += acc + bias (in place, fp32 sum rounded once): the residual stream kept in
                                       the reference's own storage type (convert_weights, model.py:531-548), one copy that is
                                       also the next GEMM's operand; aux = statistics as above (of the fp32 sums) or NULL */
#define KEDS_EPI_LN_BIAS_BF16_H 10  /* KEDS_EPI_LN_BIAS_BF16 with fp16 operands: A = the fp16 residual stream, W' fp16 */
#define KEDS_EPI_LN_QGELU_BF16_H 11 /* KEDS_EPI_LN_QGELU_BF16 with fp16 operands */
#define KEDS_EPI_BIAS_BF16_HEADF32 12 /* out bf16 = acc + bias, and rows m < aux_i also as fp32 to ((float*)aux)[m*3N + n]: IM2TEXT's
                                       * last layer writes the bf16 rows the CrossFormers read AND token slot 2 of [B,3,N] */
/* Split-operand GEMMs of the "fp32x3" operating point (round 5): BOTH operands are pairs of fp16 planes, x = hi + lo with
 * hi = fp16(x), lo = fp16(x - hi) (|x| < 65504; 22 significant bits while lo is a normal fp16 number, i.e. |x| >= 2^-3 -- below
 * that an absolute 2^-25), and the product is  hi.hi + hi.lo + lo.hi  on the fp16 MFMA with fp32 accumulation -- three times
 * the matrix work of a bf16 GEMM, ~1/5 of the time of the f32-input MFMA (157 TF), and fp32-grade results (the dropped lo.lo term
 * is 2^-22 relative).  A = planes [2][rows][K] (keds_split_f16_pair: activations are O(1)), W = planes [2][N][K] of W 2^e
 * (keds_split_f16_weight: an exact per-matrix power of two that lifts small weights out of the subnormal range; the epilogue
 * takes it out again); called through keds_gemm_x3. */
#define KEDS_EPI_X3_BIAS_F32 13     /* out f32 = acc + bias */
#define KEDS_EPI_X3_RESID_F32 14    /* out f32 += acc + bias (in place) */
#define KEDS_EPI_X3_QGELU_PAIR 15   /* out = fp16 planes [2][rows][N] of qgelu(acc + bias) (full-precision expf / division): the
                                       next split-operand GEMM's A operand; aux_i = elements between the two planes */

/* out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]).  A, W bf16 row-major (W is the nn.Linear
 * weight as stored).  N % 128 == 0, K % 64 == 0; rows of A / out up to the next multiple of
 * 128 above M must be addressable (workspaces are allocated padded).  `bias` may be NULL.
 * aux: EPI_PATCH: fp32 positional embedding [G+1, N], aux_i = G (patches per image). */
int keds_gemm_bt(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                 int epilogue, const float* aux, int aux_i, void* stream);
/* same with row strides (elements) of A and out: rows need not be contiguous (CLS-row-only last layer); A rows
 * past M are never read */
int keds_gemm_bt_ex(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                    int M, int N, int K, int epilogue, const float* aux, int aux_i, void* stream);

/* same with the second auxiliary pointer the KEDS_EPI_LN_* / KEDS_EPI_RESID_STATS_F32 epilogues use */
int keds_gemm_bt_ex2(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                     int M, int N, int K, int epilogue, const float* aux, int aux_i, void* aux2, void* stream);

/* Fold a LayerNorm (gamma, beta over K) into the nn.Linear that consumes it: W fp32 [N,K], bias fp32 [N] (nullable) ->
 * w_folded bf16 [N,K] = W diag(gamma), bias_csum fp32 [2N] = [bias + W beta | row sums of w_folded (as rounded)]. */
int keds_fold_layernorm(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                        void* w_folded, float* bias_csum, void* stream);
/* out_f16 != 0: w_folded in fp16 (operand of the KEDS_EPI_LN_*_H GEMMs, which read the fp16 residual stream) */
int keds_fold_layernorm_ex(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                           void* w_folded, int out_f16, float* bias_csum, void* stream);

/* Row statistics + bf16 copy of a residual stream (what KEDS_EPI_RESID_STATS_F32 emits, for the first block):
 * x fp32 [rows, dim] dense -> xb bf16 [rows, dim], stats [rows,2] = {sum, sum of squares} as 64-bit fixed point (* 2^28). */
int keds_rowstats_cast(const float* x, void* xb, float* stats, int rows, int dim, void* stream);
int keds_rowstats_cast_ex(const float* x, void* xb, int out_f16, float* stats, int rows, int dim, void* stream);  /* xb fp16 */

/* ---- MXFP8 (BASELINE config 5: fp8 encoders).  OCP e4m3 elements, one e8m0 scale per 32 consecutive K (OCP MX),
 * multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (block scales applied in hardware, 2x the bf16 MFMA rate).
 * Scales are stored as [K/128][rows_pad] dwords: byte b of dword (t, r) scales elements 128 t + 32 b .. + 31 of row r. */
size_t keds_mxfp8_scale_bytes(int rows_pad, int K);
int keds_mxfp8_debug(int variant);          /* timing-only ablation hook of the MXFP8 GEMM (0 = product path) */
/* x fp32 or bf16 [rows, K] dense -> q fp8 [rows, K], scales as above.  K % 128 == 0. */
int keds_quantize_mxfp8(const void* x, int x_is_bf16, int rows, int K, int rows_pad, void* q, void* scales, void* stream);
/* out bf16 [M,N] = A . W^T + bias with A [M,K], W [N,K] in MXFP8 (m_pad / n_pad: row counts of the scale arrays).
 * M % 256 == 0, N % 256 == 0, K % 128 == 0, K >= 256. */
int keds_gemm_mxfp8(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad,
                    const float* bias, void* out, int M, int N, int K, void* stream);
/* the same with the fused epilogues of the fp8 tower (cf. KEDS_EPI_LN_* / KEDS_EPI_RESID_STATS_F32 above):
 *   LN_BIAS_BF16   out bf16 = rstd (acc - mean csum) + bias';  bias = [bias' | csum], aux = row statistics [M,2],
 *                  aux2 (nullable) = statistics buffer cleared for these rows
 *   LN_QGELU_MX    the same, then QuickGELU, emitted as MXFP8 to qout [M,N] / qscale ([N/128][q_pad] dwords)
 *   RESID_STATS_MX out fp32 += acc + bias; MXFP8 copy of the new rows to qout / qscale; aux += {sum, sum sq} per row */
#define KEDS_FP8_EPI_BIAS_BF16 0
#define KEDS_FP8_EPI_LN_BIAS_BF16 1
#define KEDS_FP8_EPI_LN_QGELU_MX 2
#define KEDS_FP8_EPI_RESID_STATS_MX 3
#define KEDS_FP8_EPI_RESID_STATS_MX_H 4   /* the same on an fp16 residual stream: out fp16 += acc + bias */
int keds_gemm_mxfp8_ex(const void* Aq, const void* As, int m_pad, const void* Wq, const void* Ws, int n_pad,
                       const float* bias, void* out, int M, int N, int K, int epilogue, float* aux, float* aux2,
                       void* qout, void* qscale, int q_pad, void* stream);
/* Weight preparation for the fp8 tower: W fp32 [N,K] (times diag(gamma) when a LayerNorm is folded in) -> MXFP8 wq / wscale,
 * bias_csum fp32 [2N] = [bias + W beta | row sums of the dequantised weight].  gamma = beta = NULL: plain quantisation. */
int keds_fold_layernorm_mxfp8(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                              int n_pad, void* wq, void* wscale, float* bias_csum, void* stream);

/* Optional split-K scratch (fp32, 32 MiB is enough for every shape of the path).  Launches with fewer than ~64 output
 * tiles (remainder rows, M <= 256) then split K over up to 16 workgroups per tile and reduce in a second tiny kernel;
 * without it they run unsplit.  The buffer must stay valid until it is replaced; pass NULL to unregister. */
int keds_gemm_set_workspace(void* ptr, size_t bytes);

/* test/bench hook: bit 0 routes every GEMM through the 128x128 kernel, bit 8 skips the remainder-row launch (timing
 * only), bit 9 disables split-K; A/B switches of the 256x256 kernels: bit 10 residual tile as the accumulators' initial
 * value, bits 11-12 kernel form (1 = 4 waves, 2 = 4 waves persistent, 3 = 8 waves; 0 = by shape), bits 13-15 stamped
 * diagnostic build, bit 16 no three-deep A ring, bit 17 no deferred epilogue stores in the persistent kernel */
/* out = epilogue(A . W^T 2^-w_exp + bias) on split fp16 operands (KEDS_EPI_X3_*): A_hi = a, A_lo = a + a_plane elements (rows of
 * lda elements), W_hi = w, W_lo = w + w_plane (dense [N, K]); the W planes hold W 2^w_exp (keds_split_f16_weight; 0 for planes
 * of the matrix as stored).  M, N, K as keds_gemm_bt_ex2; ldc in elements of the output type. */
int keds_gemm_x3(const void* a, int64_t a_plane, int64_t lda, const void* w, int64_t w_plane, const float* bias, void* out,
                 int64_t ldc, int M, int N, int K, int epilogue, int aux_i, int w_exp, void* stream);
/* x fp32 [rows, cols] (row stride ld elements) -> fp16 planes hi = out, lo = out + plane (dense rows of `cols`): the A operand
 * of keds_gemm_x3.  x = hi + lo to 22 significant bits for |x| >= 2^-3; below that the low plane is an fp16 subnormal and the pair
 * carries x to an ABSOLUTE 2^-25 (activations are O(1): LayerNorm outputs, attention outputs, QuickGELU values).  Values beyond
 * the fp16 range raise *overflow (device int32, nullable) instead of going through as inf. */
int keds_split_f16_pair(const float* x, int64_t ld, int64_t rows, int cols, void* out, int64_t plane, int* overflow, void* stream);
/* A WEIGHT matrix w fp32 dense [n, k] -> the planes of w 2^e, with e = *w_exp (host int, written) chosen so that max |w| 2^e lies
 * in [2^13, 2^14): small weights (|w| ~ 1e-2 .. 1e-3 in real checkpoints) then keep 22 bits -- split as stored their low plane
 * falls into the fp16 subnormals and keeps 14-17.  Pass *w_exp to keds_gemm_x3.  Packing-time call: waits for the stream once. */
int keds_split_f16_weight(const float* w, int64_t n, int k, void* out, int64_t plane, int* w_exp, void* stream);
int keds_gemm_force_small(int on);

/* y = LayerNorm(x) * gamma + beta over the last dim (fp32 statistics, eps 1e-5).
 * x fp32 [rows, dim] with row stride x_stride (elements); out bf16 (out_f32 == 0) or fp32,
 * dense [rows, dim].  dim % 256 == 0 or dim == 128; dim <= 2048. */
int keds_layernorm(const float* x, int64_t x_stride, const float* gamma, const float* beta,
                   void* out, int out_f32, int rows, int dim, void* stream);

/* timing-only ablation hook of the S > 96 attention kernel (0 = product path) */
int keds_attention_debug(int variant);
/* diagnostic: device buffer of B * heads * 64 uint64 that the stamped build (keds_attention_debug(64 + 8)) of the S = 257
 * kernel fills with per-wave phase durations in shader cycles (tools/stamp_attn.py) */
int keds_attention_stamp_buffer(void* buf);

/* multi-head self attention core on a packed qkv buffer (nn.MultiheadAttention,
 * model.py:309,319-321): qkv bf16 [B*S, 3*d] (q | k | v, head h at columns h*64),
 * out bf16 [B*S, d] = softmax(q k^T / 8 [+ causal mask, model.py:543-549]) v.  dh = 64,
 * S <= 288. */
int keds_attention(const void* qkv, void* out, int B, int S, int heads, int causal, void* stream);
/* same, computing and storing only the first q_limit query rows of every sample (keys/values: all S rows) */
int keds_attention_ex(const void* qkv, void* out, int B, int S, int heads, int causal, int q_limit, void* stream);
/* fp8 towers: rows < q8_rows of the output are written as MXFP8 (q8 [q8_rows, d] e4m3, s8 scale dwords [d/128][q8_rows])
 * INSTEAD of bf16; the remaining rows go to `out` as usual */
int keds_attention_mx(const void* qkv, void* out, int B, int S, int heads, int causal, int q_limit, void* q8, void* s8,
                      int q8_rows, void* stream);
/* PACKED rows (round 6): sample b is rows [seq_off[b], seq_off[b + 1]) of qkv / out (device int32 [B + 1]; lengths 1 .. s_max
 * <= 288) instead of [b S, (b + 1) S): sequences of different lengths without padding rows (the text tower, keds_text_run_packed) */
int keds_attention_packed(const void* qkv, void* out, int B, int s_max, const int32_t* seq_off, int heads, int causal, void* stream);

/* patch im2col for conv1 (model.py:381,394-396): image fp32 [B,3,R,R] -> bf16 [B*G, Kpad],
 * column c*P*P + ky*P + kx, zero padded to Kpad (a multiple of 64). */
int keds_im2col(const float* image, void* out, int B, int R, int P, int Kpad, void* stream);

/* token embedding + optional pseudo-token splice + positional embedding
 * (model.py:579-581, 817-837): x fp32 [B, L, d].
 *   tokens int32 [B,L]; table bf16/fp32?  -> fp32 table [vocab, d]
 *   img_tokens (nullable) fp32 [B, n_tok, d] replaces column `insert_col`, the tail is shifted
 *   right by n_tok-1 and truncated to L. */
int keds_embed_tokens(const int32_t* tokens, const float* table, const float* pos,
                      const float* img_tokens, int n_tok, int insert_col,
                      float* x, int B, int L, int d, void* stream);

/* read-out: out[b,:] = LayerNorm(x[b*S + row[b], :]) . proj  (+ optional L2 normalisation)
 * (model.py:412-414, 586-589, 841-849).  proj_t bf16 [E, d] (the projection transposed),
 * row int32 [B] (NULL: row 0, the CLS token).  out fp32 [B, E]. */
int keds_readout(const float* x, int S, const int32_t* row, const float* gamma, const float* beta,
                 const void* proj_t, float* out, int B, int d, int E, int normalize,
                 void* workspace, size_t workspace_bytes, void* stream);
size_t keds_readout_workspace_bytes(int B, int d);

/* rows /= ||row||  (eval_utils.py:162,704-710);  out may alias x */
int keds_l2_normalize(const float* x, float* out, int rows, int dim, void* stream);
/* out = normalize(wa * normalize(a) + wb * normalize(b))  (eval_utils.py:704-710) */
int keds_mix_normalize(const float* a, const float* b, float wa, float wb,
                       float* a_n, float* b_n, float* mix, int rows, int dim, void* stream);
/* fp32 -> bf16 cast (weight packing, model.py:927-948 stand-in) */
int keds_cast_bf16(const float* x, void* out, int64_t count, void* stream);

/* =====================================================================================
 * 3. Whole towers
 * ===================================================================================== */
typedef struct {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;     /* fp32 [d] */
    const void *qkv_w, *out_w, *fc_w, *proj_w;      /* bf16 [3d,d] [d,d] [4d,d] [d,4d] */
    const float *qkv_b, *out_b, *fc_b, *proj_b;     /* fp32 */
    /* optional (all four or none): ln_1 folded into in_proj and ln_2 into c_fc by keds_fold_layernorm_ex(out_f16 = 1).
     * When present the tower runs without LayerNorm passes on an fp16 residual stream (KEDS_EPI_LN_*_H /
     * KEDS_EPI_RESID_STATS_F16); qkv_w / fc_w and the ln parameters are still needed (CLS-only last block). */
    const void *qkv_wf, *fc_wf;                     /* fp16 [3d,d], [4d,d] */
    const float *qkv_bc, *fc_bc;                    /* fp32 [2*3d], [2*4d]: folded bias | column sums */
    /* optional (all or none; needs the folded set above): MXFP8 copies for keds_tower_params.fp8
     * (keds_fold_layernorm_mxfp8: in_proj / c_fc with their LayerNorm folded in, out_proj / c_proj plain);
     * scales in the [K/128][N] dword layout */
    const void *qkv_q8, *out_q8, *fc_q8, *proj_q8;
    const void *qkv_s8, *out_s8, *fc_s8, *proj_s8;
    const float *qkv_bc8, *fc_bc8;                  /* fp32 [2*3d], [2*4d] for the MXFP8 weights */
    /* keds_tower_params.f32 == 2 only: the exponents keds_split_f16_weight returned for qkv_w, out_w, fc_w, proj_w */
    int x3_exp[4];
} keds_block_params;

typedef struct {
    int width, layers, heads, seq;                  /* seq = tokens per sample (257 / 77) */
    int causal;
    const keds_block_params* blocks;                /* HOST array [layers] of device pointers */
    int fp8;                                        /* 1: BASELINE config 5 -- the four GEMMs of every block run on MXFP8
                                                       operands (full 256-row tiles; remainder rows stay bf16); needs the
                                                       *_q8 block fields and width % 256 == 0 */
    int last_cls_only;                              /* 1: after the LAST block only token 0 of every sample is
                                                       defined (ViT read-out): its attention queries, out-proj, ln_2
                                                       and MLP run on B rows instead of B*seq */
    int f32;                                        /* 1: the fp32-ACCURATE flow (section 10; the reference's own eval
                                                       arithmetic, eval_retrieval.py:108-109): qkv_w / out_w / fc_w / proj_w of
                                                       every block -- and conv_w / proj_t of the enclosing vit / text struct --
                                                       point to FP32 arrays of the same shapes; the folded / MXFP8 fields are
                                                       unused; fp8 must be 0 */
} keds_tower_params;

typedef struct {
    keds_tower_params tower;
    int resolution, patch, kpad, embed_dim;
    const void* conv_w;                             /* bf16 [width, kpad] (im2col order, zero padded) */
    const float *class_emb, *pos_emb;               /* fp32 [width], [G+1, width] */
    const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    const void* proj_t;                             /* bf16 [embed_dim, width] */
} keds_vit_params;

typedef struct {
    keds_tower_params tower;
    int vocab, embed_dim;
    const float *token_emb, *pos_emb;               /* fp32 [vocab,d], [L,d] */
    const float *ln_final_g, *ln_final_b;
    const void* proj_t;                             /* bf16 [embed_dim, d] */
} keds_text_params;

size_t keds_tower_workspace_bytes(int width, int seq, int B);
size_t keds_tower_workspace_bytes_ex(const keds_tower_params* p, int B);   /* knows the fp32-accurate flow (p->f32) */
/* Rows (B*seq mod 256) that a tower pass of keds_vit_run / keds_text_run runs as their own chain on a second, high-priority
 * stream beside the full 256-row tiles (0: one stream; always 0 with KEDS_SIDE_STREAM=0, and 0 with KEDS_TOWER_FILL=1, an
 * experiment that runs the ragged last row tile as a full tile on filler rows -- measured slower, off by default).
 * Side-lane launches carry no profiling events: keds_prof_read(KEDS_PROF_GEMM) then covers the
 * full-tile launches only (bench.py scales the flops to match). */
int keds_tower_side_rows(int width, int seq, int B, int fp8);
int keds_side_lane_enable(int on);            /* run-time override of KEDS_SIDE_STREAM (default: on); results are identical */
int keds_tower_fill_enable(int on);           /* run-time override of KEDS_TOWER_FILL (default: off): ragged last row tile on filler rows */

/* x fp32 [B*seq (padded to 128), width] in place through all residual blocks (model.py:372-373); workspace: rows padded to 256 */
int keds_tower_forward(const keds_tower_params* p, float* x, int B,
                       void* workspace, size_t workspace_bytes, void* stream);

/* _transform of src/model/clip.py:107-123 (eval branch) on the device: raw uint8 images [B,H,W,3] of one size (device) ->
 * fp32 [B,3,n_px,n_px]: bicubic Resize of the shorter side to n_px (PIL semantics: two antialiased passes with a uint8
 * intermediate), CenterCrop, /255, (x - mean) / std.  mean3 / std3: HOST arrays of 3 floats. */
/* bit-exact form: PIL's integer resampling (22-bit fixed-point weights computed by the host exactly as PIL's Resample.c
 * does, horizontal pass first, uint8 intermediate).  need_h / need_v: the axis is resampled at all (PIL skips a pass that
 * keeps the size); xb/yb [n_px][2] = {first source index, taps} of every output column/row of the crop, xk/yk
 * [n_px][ksx|ksy] their int32 weights; left/top: crop offset when an axis is NOT resampled.  out_u8 (nullable)
 * [B,n_px,n_px,3]: the uint8 image PIL would hand to ToTensor. */
int keds_preprocess_pil(const unsigned char* images, int B, int H, int W, int n_px, int need_h, int need_v, int left, int top,
                        const int32_t* xb, const int32_t* xk, int ksx, const int32_t* yb, const int32_t* yk, int ksy,
                        const float* mean3, const float* std3, float* out, unsigned char* out_u8, void* stream);
int keds_preprocess(const unsigned char* images, int B, int H, int W, int n_px, const float* mean3, const float* std3,
                    float* out, void* stream);

/* CLIP.encode_image (model.py:569-575,393-415): image fp32 [B,3,R,R] -> out fp32 [B, embed_dim] */
size_t keds_vit_workspace_bytes(const keds_vit_params* p, int B);
int keds_vit_run(const keds_vit_params* p, const float* image, int B, float* out, int normalize,
                     void* workspace, size_t workspace_bytes, void* stream);

/* CLIP.encode_text / encode_text_img_retrieval (model.py:577-590, 808-851):
 * tokens int32 [B,L]; readout_row int32 [B]; img_tokens nullable fp32 [B,n_tok,d]. */
size_t keds_text_workspace_bytes(const keds_text_params* p, int B);
int keds_text_run(const keds_text_params* p, const int32_t* tokens, const int32_t* readout_row,
                      const float* img_tokens, int n_tok, int insert_col, int B, float* out, int normalize,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same with the host's knowledge of the read-out columns (ABI 7).  The mask is causal (model.py:543-549) and only the
 * read-out row of every sample is read (model.py:587-589, 847-849), so columns >= seq_used = max(readout_row) + 1 cannot
 * change the result: they are neither embedded nor run through the blocks (the sequence is cut there), and the last block's
 * out-proj / ln_2 / MLP run on the B read-out rows only.  seq_used = 0: unknown (all L columns; keds_text_run).  CONTRACT: every readout_row[b] < seq_used -- the rows live on
 * the device and are not checked.  keds_text_trim_enable(0) restores the all-columns / all-rows flow (A/B, bisecting), 2 =
 * the column cut only, 3 = the read-out-row tail only; -1 = the KEDS_TEXT_TRIM environment variable decides (default: on).
 * Numerics: the column cut changes no arithmetic (rows only land in other tiles); the tail runs the last block's out-proj /
 * ln_2 / MLP in the fp32-stream form (stand-alone LayerNorm, fp32 residual, as the ViT's CLS tail does) where the all-rows
 * flow uses the folded form on the fp16 stream: both are roundings of the same fp32 reference, 2-3e-3 apart. */
int keds_text_run_ex(const keds_text_params* p, const int32_t* tokens, const int32_t* readout_row,
                     const float* img_tokens, int n_tok, int insert_col, int B, int seq_used, float* out, int normalize,
                     void* workspace, size_t workspace_bytes, void* stream);
int keds_text_trim_enable(int on);
int keds_text_trim_mode(void);      /* the flow in force (0 .. 3 as above) */
/* The text tower on PACKED rows (round 6, ABI 8): captions end at different columns, and under the causal mask sample b needs its
 * columns [0, len_b) only (len_b = its read-out column + 1) -- keds_text_run_ex cuts every sample at the LONGEST caption of the
 * batch, here sample b owns rows [seq_off[b], seq_off[b + 1]) of every activation buffer and the tower runs sum(len_b) rows.
 *   seq_off         device int32 [B + 1], seq_off[0] = 0, seq_off[B] = rows_total, 1 <= len_b <= seq_max <= tower.seq
 *   readout_global  device int32 [B]: seq_off[b] + the read-out column of sample b (a row outside [0, rows_total) gives NaN)
 * Same arithmetic per row as keds_text_run_ex's default flow (rows only land in other tiles); causal bf16 towers only -- fp8 /
 * fp32 towers and the A/B flows of keds_text_trim_enable go through keds_text_run_ex.  Workspace: keds_text_workspace_bytes. */
int keds_text_run_packed(const keds_text_params* p, const int32_t* tokens, const int32_t* seq_off, const int32_t* readout_global,
                         int rows_total, int seq_max, const float* img_tokens, int n_tok, int insert_col, int B, float* out,
                         int normalize, void* workspace, size_t workspace_bytes, void* stream);

/* =====================================================================================
 * 4. Knowledge injection (IM2TEXT + 2 x CrossFormer, model.py:37-123, eval_utils.py:661-672)
 * ===================================================================================== */
typedef struct {
    const void *wq, *wk, *wv, *wo;                  /* bf16 [inner,dim] x3, [dim,inner] */
    const float *bq, *bk, *bv, *bo;
} keds_cross_layer_params;

typedef struct {                                    /* IM2TEXT: dim_in -> middle (x n_layer, ReLU) -> dim_out */
    int dim_in, middle, dim_out, n_layer;
    const void* w[4]; const float* b[4];            /* bf16 [middle, in] / fp32 [middle] per hidden layer */
    const void* out_w; const float* out_b;          /* bf16 [dim_out, middle] */
} keds_im2text_params;

/* Launch-saving re-arrangement of a CrossFormer's weights (built once by keds_crossformer_fuse from the per-layer ones):
 *  - k and v never change between the layers (model.py:98-101), so ALL layers' k / v projections are one GEMM:
 *    wkv [layers*2*inner, dim] = rows {Wk_0, Wv_0, Wk_1, Wv_1, ...}, bkv likewise;
 *  - layer l's output projection feeds only layer l+1's query projection (no residual, no norm, model.py:56-79), so the
 *    two are one matrix: wqn[l] = Wq_l . Wo_{l-1} [inner, inner], bqn[l] = Wq_l bo_{l-1} + bq_l   (l >= 1). */
typedef struct {
    const void* wkv; const float* bkv;
    const void* wqn[8]; const float* bqn[8];
} keds_crossformer_fused;

typedef struct {                                    /* CrossFormer: `layers` chained CrossAttention layers */
    int dim, heads, layers;                         /* dim_head = 64, inner = heads*64 */
    const keds_cross_layer_params* layer;           /* HOST array [layers] of device pointers */
    const keds_crossformer_fused* fused;            /* HOST struct of device pointers, or NULL (layer-by-layer launches) */
} keds_crossformer_params;

typedef struct {
    keds_im2text_params i2t;
    keds_crossformer_params fuse, cond;             /* retrieval_fuse, text_condition */
} keds_knowledge_params;

/* IM2TEXT.forward (model.py:120-123): x fp32 [rows, dim_in] -> out fp32 [rows, dim_out] */
size_t keds_im2text_workspace_bytes(const keds_im2text_params* p, int rows);
int keds_im2text_forward(const keds_im2text_params* p, const float* x, int rows, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* CrossFormer.forward (model.py:98-101) for single-query attention: q fp32 [B,dim],
 * k, v fp32 [B,K,dim] (K <= 32; v may alias k) -> out fp32 [B,dim] */
size_t keds_crossformer_workspace_bytes(const keds_crossformer_params* p, int B, int K);
int keds_crossformer_forward(const keds_crossformer_params* p, const float* q, const float* k, const float* v,
                             int B, int K, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* build `fused` (above) into a caller-owned device buffer of keds_crossformer_fused_bytes(p) bytes; layers <= 8 */
size_t keds_crossformer_fused_bytes(const keds_crossformer_params* p);
int keds_crossformer_fuse(const keds_crossformer_params* p, void* buffer, size_t buffer_bytes, keds_crossformer_fused* out,
                          void* stream);

size_t keds_knowledge_workspace_bytes(const keds_knowledge_params* p, int B, int K);
/* one stream of eval_utils.py:661-672: q [B,dim] fp32, nbr_img / nbr_txt [B,K,dim] fp32 ->
 * tokens_out [B,3,dim] fp32 = [retrieval_fuse(m, I, I), text_condition(m, T, T), m], m = img2text(q),
 * I = img2text(nbr_img), T = img2text(nbr_txt) */
int keds_knowledge_run(const keds_knowledge_params* p, const float* q, const float* nbr_img,
                           const float* nbr_txt, int B, int K, float* tokens_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* =====================================================================================
 * 9. Training step of the knowledge-injection modules: building blocks of the forward in training
 *    mode, of the backward through IM2TEXT / CrossFormer / the FROZEN text tower, of the symmetric contrastive loss and
 *    of AdamW (src/trainer.py:44-165, src/main.py:215-237).  The matrix products run on keds_gemm_bt* (dX = dY.W with
 *    W^T as the operand, dW = dY^T.X with both operands transposed: keds_transpose_to_bf16); the host (keds_amd/train.py)
 *    sequences the step, as the reference's Python does.  Gradients that feed a GEMM are bf16, the residual-stream
 *    gradient, LayerNorm, softmax, loss and optimizer arithmetic fp32.
 * ===================================================================================== */
/* out bf16 [cols, ld_out] = transpose of src [rows, cols] (fp32 or bf16, row stride ld_src), columns rows..ld_out-1 zero */
int keds_transpose_to_bf16(const void* src, int src_is_f32, int64_t ld_src, int rows, int cols, void* out, int ld_out, void* stream);
/* out[c] (+)= sum_r x[r][c]: bias gradients (fixed summation order: reproducible) */
int keds_colsum(const void* x, int x_is_f32, int64_t ld, int rows, int cols, float* out, int accumulate, void* stream);
/* IM2TEXT hidden layer y = relu(dropout(z)) (model.py:112-116); mask uint8 (1 = keep; NULL = no dropout), scale = 1/(1-p) */
int keds_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, float p, void* stream);
int keds_dropout_relu_fwd(const void* z_bf16, const uint8_t* mask, float scale, void* y_bf16, int64_t n, void* stream);
int keds_dropout_relu_bwd(const void* dy, int dy_is_f32, const void* z_bf16, const uint8_t* mask, float scale, void* dz_bf16,
                          int64_t n, void* stream);
/* QuickGELU with the pre-activation kept (model.py:300-302) */
int keds_qgelu_fwd(const void* u_bf16, void* y_bf16, int64_t n, void* stream);
int keds_qgelu_bwd(const void* dy_bf16, const void* u_bf16, void* du_bf16, int64_t n, void* stream);
/* LayerNorm with saved statistics {mean, rstd} per row; rowmap (nullable): row r reads x[rowmap[r]] */
int keds_ln_fwd_stats(const float* x, int64_t ld, const int32_t* rowmap, const float* gamma, const float* beta, void* y_bf16,
                      float* stats, int rows, int dim, void* stream);
/* dx[rowmap[r]] += dLN/dx (dy [rows, dim] compact fp32; gamma frozen: no parameter gradients); dx_bf16 (nullable): bf16 copy */
int keds_ln_bwd(const float* dy, const float* x, int64_t ld, const int32_t* rowmap, const float* stats, const float* gamma,
                float* dx, void* dx_bf16, int rows, int dim, void* stream);
/* self-attention backward on the packed qkv buffer of keds_attention (S <= 80): dout bf16 [B*S, d] -> dqkv bf16 [B*S, 3d] */
int keds_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int S, int heads, int causal, void* stream);
/* single-query cross-attention core (model.py:56-79) on projected bf16 rows Q [B, inner], K / V [B*K, inner], and backward */
int keds_cross_core_fwd(const void* Q, const void* Kp, const void* Vp, void* out, int B, int K, int heads, void* stream);
int keds_cross_core_bwd(const void* Q, const void* Kp, const void* Vp, const void* dout, void* dQ, void* dK, void* dV, int B,
                        int K, int heads, void* stream);
/* trainer.py:78-127: logits = scale * img_n . txt_n^T over N gathered rows (this rank's B_local first), loss = (CE rows + CE
 * columns) / 2 -> loss[0] (device), dtxt_n [B_local, dim] = dloss / d txt_n of this rank's rows (remote rows carry no grad) */
size_t keds_clip_loss_workspace_bytes(int N);
int keds_clip_loss(const float* img_n, const float* txt_n, int N, int B_local, int dim, float scale, float* loss, float* dtxt_n,
                   void* workspace, size_t workspace_bytes, void* stream);
/* y = x / ||x||:  dx = (dy - y (y . dy)) / ||x|| */
int keds_l2norm_bwd(const float* x, const float* dy, float* dx, int rows, int dim, void* stream);
/* torch.optim.AdamW update of one flat parameter buffer (step counts from 1); the gradient is read as g * grad_scale
 * (1 / world size after a SUM all-reduce) */
int keds_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* stream);
/* dst[map[r]] (+)= src[r]  /  dst[r] = src[map[r]]   (read-out rows, spliced token rows) */
int keds_rows_scatter(const float* src, const int32_t* map, float* dst, int64_t ld_dst, int rows, int dim, int accumulate,
                      void* stream);
int keds_rows_gather(const float* src, int64_t ld_src, const int32_t* map, float* dst, int rows, int dim, void* stream);

/* =====================================================================================
 * 10. The fp32-accurate operating point (csrc/f32path.hip)
 * =====================================================================================
 * The reference evaluates in fp32 (src/eval_retrieval.py:108-109, flag src/params.py:227-232).  These kernels run the same
 * path WITHOUT rounding any operand: products on the f32-input matrix instruction (exact fp32 products, fp32 accumulate),
 * fp32 residual stream / LayerNorm output / q, k, v / probabilities / MLP hidden layer.  keds_vit_run / keds_text_run /
 * keds_tower_forward take this flow when keds_tower_params.f32 = 1 (CLIP.set_precision("fp32"), KEDS_F32 compute of
 * keds_vit_create / keds_text_create).  About 1/10 of the default flow's throughput: an accuracy reference, not the headline. */
#define KEDS_F32_EPI_BIAS 0      /* out = acc + bias                                   */
#define KEDS_F32_EPI_QGELU 1     /* out = quick_gelu(acc + bias)   (model.py:300-302)  */
#define KEDS_F32_EPI_RESID 2     /* out += acc + bias                                  */
#define KEDS_F32_EPI_RELU 3      /* out = max(acc + bias, 0)                           */
#define KEDS_F32_EPI_PATCH 4     /* patch embedding: row remap + positional embedding (aux = pos_emb [G+1, N], aux_i = G) */
/* out[M,N] (row stride ldc) = epilogue(A[M,K] (row stride lda) . W[N,K]^T + bias); all fp32; N % 128 == 0, K % 16 == 0,
 * strides multiples of 4; bias nullable */
int keds_gemm_f32(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M, int N, int K,
                  int epilogue, const float* aux, int aux_i, void* stream);
/* keds_attention_ex on fp32 qkv [B*S, 3*d] -> out fp32 [B*S, d]; q_limit <= 0: all rows; S <= 288 */
int keds_attention_f32(const float* qkv, float* out, int B, int S, int heads, int causal, int q_limit, void* stream);
/* keds_attention_f32 at fp32 grade on the fp16 matrix instruction (the fp32x3 operating point, csrc/attention_x3.hip): q, k, v
 * and the probabilities as two fp16 planes each, three products per product, fp32 softmax.  out: fp32 [B*S, d] (nullable);
 * pair: fp16 planes [2][plane] of the same rows (nullable; what keds_gemm_x3 reads as its A operand); at least one of them.
 * overflow (nullable int on the device): raised when |q / 8|, |k| or |v| >= 65504.  Reference: src/model/model.py:319-321 */
int keds_attention_x3(const float* qkv, float* out, void* pair, int64_t plane, int B, int S, int heads, int causal, int q_limit,
                      int* overflow, void* stream);
/* keds_im2col with an fp32 patch matrix [B*G, Kpad] (Kpad % 16 == 0) */
int keds_im2col_f32(const float* image, float* out, int B, int R, int P, int Kpad, void* stream);
/* keds_knowledge_run with EVERY weight pointer of the params an FP32 array (per-layer weights; `fused` unused): one stream
 * of the knowledge injection, tokens_out fp32 [B,3,dim]; K <= 64 */
size_t keds_knowledge_f32_workspace_bytes(const keds_knowledge_params* p, int B, int K);
int keds_knowledge_run_f32(const keds_knowledge_params* p, const float* q, const float* nbr_img, const float* nbr_txt, int B,
                           int K, float* tokens_out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEDS_HIP_H */
