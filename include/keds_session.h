/* keds_session.h — handle-based C ABI of libkeds_hip.so (gfx950 / MI355X).
 *
 * The layer a non-torch host (C, C++, cgo, JNI, N-API, ctypes) binds: the library owns the packed
 * weights, the database, the workspaces and the RCCL communicator; the caller owns inputs and
 * outputs (plain device pointers) and the stream.  One call per starred method of the reference:
 *
 *   keds_vit_forward        <- CLIP.encode_image                 src/model/model.py:569-575, 393-415
 *   keds_text_forward       <- CLIP.encode_text / encode_text_img_retrieval
 *                                                                src/model/model.py:577-590, 808-851
 *   keds_knowledge_forward  <- img2text + retrieval_fuse + text_condition of one stream
 *                                                                src/eval_utils.py:661-672, model.py:37-123
 *   keds_index_create/add/search <- faiss.IndexFlatL2(768), index_cpu_to_all_gpus, .add, .search
 *                                                                src/eval_retrieval.py:289-298, eval_utils.py:169,177
 *   keds_comm_init + keds_index_search_sharded <- (no reference counterpart: Faiss replicates or
 *                                                  shards inside index_cpu_to_all_gpus) SURVEY.md 8e
 *
 * Every forward is built from the stateless entry points of keds_hip.h (same kernels, same
 * results bit for bit); nothing here computes on the CPU.  Weights are handed over as a list of
 * named tensors using the reference's own state_dict keys (model.py:381-391,483-507; build_model
 * model.py:951-991 infers the architecture from the shapes the same way), host or device memory,
 * fp32 / fp16 / bf16.  All functions return KEDS_OK (0) or a negative status; keds_last_error()
 * holds the message.  A handle is not thread-safe; different handles may be used from different
 * threads.  Launches are asynchronous on the caller's stream, except that the first call with a
 * larger batch than any before grows the handle's workspace (hipMalloc).
 */
#ifndef KEDS_SESSION_H
#define KEDS_SESSION_H

#include "keds_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { KEDS_F32 = 0, KEDS_BF16 = 1, KEDS_F16 = 2, KEDS_FP8 = 3 /* compute only: MXFP8 GEMM operands (config 5) */,
               KEDS_F32X3 = 4 /* compute only: the fp32 flow with the block GEMMs on split fp16 operands ("fp32x3", keds_hip.h:
                                 fp32-grade results, Recall@k equal, > 2x KEDS_F32's throughput; |activations| < 65504) */ } keds_dtype;

typedef struct {
    const char* name;        /* state_dict key, e.g. "visual.transformer.resblocks.0.attn.in_proj_weight" */
    const void* data;        /* host or device pointer, dense row-major */
    int dtype;               /* keds_dtype */
    int ndim;                /* 0..4 */
    int64_t shape[4];
} keds_tensor;

typedef struct keds_ctx keds_ctx;
typedef struct keds_vit keds_vit;
typedef struct keds_text keds_text;
typedef struct keds_knowledge keds_knowledge;
typedef struct keds_index keds_index;

/* one context per (process, device): selects the device, owns the split-K scratch and the communicator */
int keds_ctx_create(int device, keds_ctx** out);
int keds_ctx_destroy(keds_ctx* ctx);

/* ---- image tower: keys "visual.*" of a CLIP state_dict (other keys are ignored) ------------- */
int keds_vit_create(keds_ctx* ctx, const keds_tensor* weights, int n, int compute /* KEDS_BF16 | KEDS_FP8 | KEDS_F32 (the fp32-accurate flow, keds_hip.h section 10) | KEDS_F32X3 */, keds_vit** out);
int keds_vit_destroy(keds_vit* vit);
/* width, layers, heads, resolution, patch, embed_dim inferred from the shapes */
int keds_vit_info(const keds_vit* vit, int* width, int* layers, int* resolution, int* patch, int* embed_dim);
/* image [B,3,R,R] (dtype img_dtype, device) -> out fp32 [B, embed_dim] (not normalised, model.py:412-415) */
int keds_vit_forward(keds_vit* vit, const void* image, int img_dtype, int B, void* out, void* stream);

/* ---- text tower: keys token_embedding.weight, positional_embedding, transformer.*, ln_final.*,
 *      text_projection --------------------------------------------------------------------------- */
int keds_text_create(keds_ctx* ctx, const keds_tensor* weights, int n, int compute, keds_text** out);
int keds_text_destroy(keds_text* txt);
int keds_text_info(const keds_text* txt, int* width, int* layers, int* context, int* vocab, int* embed_dim);
/* tokens int32 [B, context]; img_tokens nullable fp32 [B, n_img_tok, width] spliced in at column
 * insert_idx (the position of the `*` token of row 0, model.py:820-834); readout_idx int32 [B] = the
 * row that is projected (EOT column, + n_img_tok - 1 after a splice, model.py:847-850). */
int keds_text_forward(keds_text* txt, const int32_t* tokens, const void* img_tokens, int n_img_tok,
                      int insert_idx, const int32_t* readout_idx, int B, void* out, void* stream);
/* the same when the HOST knows the read-out rows (it usually does: they come from the token ids): seq_used = max(readout_idx) + 1
 * lets the tower skip the columns behind it (causal mask: they cannot reach a read-out; keds_text_run_ex, keds_hip.h);
 * 0 = unknown.  Every readout_idx[b] must be < seq_used. */
int keds_text_forward_used(keds_text* txt, const int32_t* tokens, const void* img_tokens, int n_img_tok,
                           int insert_idx, const int32_t* readout_idx, int seq_used, int B, void* out, void* stream);
/* the same with the read-out columns themselves on the HOST (readout_host: host int32 [B]; round 6): a caption needs its own
 * columns [0, read-out column] only (causal mask), so the tower runs on PACKED rows -- the sum of the captions' lengths instead
 * of B x the longest -- where that pays (bf16 compute; keds_text_run_packed, keds_hip.h), the rectangular cut otherwise.  Same
 * result as keds_text_forward within the bf16 flow's own reordering of tiles. */
int keds_text_forward_packed(keds_text* txt, const int32_t* tokens, const void* img_tokens, int n_img_tok,
                             int insert_idx, const int32_t* readout_host, int B, void* out, void* stream);

/* ---- knowledge injection of ONE stream: IM2TEXT keys (layers.{i}.0.{weight,bias}, fc_out.*) and two
 *      CrossFormers (cross_layers.{i}.to_{q,k,v}.*, to_out.0.*): retrieval_fuse, text_condition ---- */
int keds_knowledge_create(keds_ctx* ctx, const keds_tensor* im2text, int n_im2text,
                          const keds_tensor* fuse, int n_fuse, const keds_tensor* cond, int n_cond,
                          keds_knowledge** out);
int keds_knowledge_destroy(keds_knowledge* kn);
/* q fp32 [B,dim], nbr_img / nbr_txt fp32 [B,K,dim] -> tokens_out fp32 [B,3,dim] (eval_utils.py:661-672) */
int keds_knowledge_forward(keds_knowledge* kn, const float* q, const float* nbr_img, const float* nbr_txt,
                           int B, int K, void* tokens_out, void* stream);

/* ---- flat exact index ------------------------------------------------------------------------ */
int keds_index_create(keds_ctx* ctx, int dim, int metric /* KEDS_METRIC_* */, int storage /* KEDS_BF16 scan image;
                      the fp32 rows are always kept for the exact re-rank */, keds_index** out);
int keds_index_destroy(keds_index* idx);
/* append n fp32 rows (host or device): the fp32 rows and the bf16 scan image live in buffers that grow geometrically,
 * only the new rows are copied and packed (amortised O(rows added)); returns once `rows` may be reused.  A chunked build
 * gives the same image, byte for byte, as one add of all rows.  Not to be called while a search of the SAME index is in
 * flight on another stream (Faiss's rule as well).  The call has no stream argument: a DEVICE source is waited for with one
 * device synchronisation before it is read (rows produced on any stream of the caller are complete by then); a host
 * source is read synchronously. */
int keds_index_add(keds_index* idx, const float* rows, int64_t n);
int64_t keds_index_ntotal(const keds_index* idx);
/* copy of the bf16 scan image (keds_hip.h: keds_index_packed_bytes(ntotal, dim) bytes, host or device destination): what a
 * host stores next to the fp32 rows so that a later run needs no re-pack (the Python facade's FlatIndex.save does) */
int keds_index_image(const keds_index* idx, void* packed_out, size_t bytes);
/* global id of local row 0 (row-sharded databases; default 0) */
int keds_index_set_base(keds_index* idx, int64_t row0);
/* q fp32 [B,dim] (device) -> D fp32 [B,k] ascending squared L2 (descending dot for IP), I int64 [B,k],
 * rows_out nullable fp32 [B,k,dim] = the winners' rows (eval_utils.py:171-172).  k <= 128 (KEDS_SCAN_MAX_K): exact, with
 * a per-query certificate and an exact fp32 pass for the queries that fail it (keds_hip.h). */
int keds_index_search(keds_index* idx, const void* q, int B, int k, float* D, int64_t* I, void* rows_out,
                      void* stream);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (librccl is dlopen'ed on first use) ------ */
#define KEDS_COMM_ID_BYTES 128
int keds_comm_unique_id(void* id_out /* KEDS_COMM_ID_BYTES, host */);          /* rank 0, then broadcast by the host */
int keds_comm_init(keds_ctx* ctx, int rank, int world, const void* unique_id);
/* Replaces the replicated multi-GPU Faiss index of src/eval_retrieval.py:289-298 and its two searches per query batch
 * (src/eval_utils.py:169-183).  Every rank passes its own B queries (same B everywhere) and holds rows
 * [row0, row0+ntotal) of the database: all-gather of the queries -> local exact search of all B*world queries -> ONE
 * all-to-all of the packed partial lists (distance | id | the winner's row when rows_out != NULL), block w to rank w ->
 * merge of this rank's B queries keyed on (distance, id).  k <= 128, world * k <= 4096.  D / I (/ rows_out [B,k,dim], the
 * winners' fp32 rows fetched from the shards that own them) are bit-identical to a single-GPU search of the whole
 * database. */
int keds_index_search_sharded(keds_index* idx, const void* q, int B, int k, float* D, int64_t* I, void* rows_out /* nullable */,
                              void* stream);

/* ---- host-side BPE tokenizer (no GPU) ------------------------------------------------------
 * Replaces `tokenize` of src/third_party/open_clip/clip.py:191-227 with SimpleTokenizer (simple_tokenizer.py:62-132):
 * html.unescape twice, strip, whitespace runs -> one space, lower (full Unicode), regex word pieces, byte symbols,
 * greedy lowest-rank merges; ids = position in [256 byte symbols | the same + "</w>" | merges | <|startoftext|> <|endoftext|>].
 * bpe_path: bpe_simple_vocab_16e6.txt(.gz) (first line is a header; not shipped with this library). */
typedef struct keds_tokenizer keds_tokenizer;
int keds_tokenizer_create(const char* bpe_path, keds_tokenizer** out);
void keds_tokenizer_destroy(keds_tokenizer* t);
int keds_tokenizer_special(const keds_tokenizer* t, int32_t* sot, int32_t* eot);
/* texts: n NUL-terminated UTF-8 strings -> out int32 [n, context_length] = <|startoftext|> ids <|endoftext|> 0 0 ...;
 * a row that does not fit is cut with its last id forced to <|endoftext|> (truncate != 0) or is an error (truncate == 0). */
int keds_tokenize(keds_tokenizer* t, const char* const* texts, int n, int context_length, int truncate, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* KEDS_SESSION_H */
