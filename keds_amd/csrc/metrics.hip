// Gallery ranking + CIRR recall bookkeeping on device.
// Reference: get_metrics_cirr, src/eval_utils.py:1040-1067 -- `1 - ref @ gallery.T`, a full
// ascending argsort per query, removal of the reference image from each ranking, one-hot against
// the target name.  The reference does the name work with a Python double loop over Q x G
// basenames; here names are interned to integer ids on the host once and compared on device.
#include "keds_common.h"
#include <math.h>

namespace {

// dist[q,g] = 1 - sum_k ref[q,k] * gal[g,k]  (fp32, 64x64 tile per 256-thread block, 4x4 per thread)
__global__ __launch_bounds__(256) void dist_kernel(const float* __restrict__ ref, int nq, const float* __restrict__ gal,
                                                   int ng, int dim, float* __restrict__ dist) {
    __shared__ float sa[16][64 + 1];
    __shared__ float sb[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int q0 = blockIdx.y * 64, g0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < dim; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, k = i & 15;
            sa[k][r] = (q0 + r < nq && k0 + k < dim) ? ref[(size_t)(q0 + r) * dim + k0 + k] : 0.f;
            sb[k][r] = (g0 + r < ng && k0 + k < dim) ? gal[(size_t)(g0 + r) * dim + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = sa[k][ty * 4 + i];
                b[i] = sb[k][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + ty * 4 + i, g = g0 + tx * 4 + j;
            if (q < nq && g < ng) dist[(size_t)q * ng + g] = 1.0f - acc[i][j];
        }
}

__device__ __forceinline__ unsigned int orderable(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one block per query: bitonic sort of (orderable(dist) << 32 | g) in LDS; P = padded power of two
__global__ __launch_bounds__(256) void sort_rows_kernel(const float* __restrict__ dist, int ng, int P,
                                                        int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < P; i += 256)
        keys[i] = i < ng ? (((unsigned long long)orderable(dist[(size_t)q * ng + i])) << 32) | (unsigned)i
                         : 0xFFFFFFFFFFFFFFFFull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = keys[i], b = keys[l];
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < ng; i += 256) order[(size_t)q * ng + i] = (int)(keys[i] & 0xFFFFFFFFu);
}

// ---- galleries beyond one LDS sort (> 8192 rows, e.g. the 17 k targets of the ImageNet domain-conversion eval):
// sort 8192-column chunks in LDS into 64-bit keys (orderable(dist) << 32 | global column), then merge the sorted runs
// pairwise in global memory; keys are unique, so an element's place in a merged pair is its offset in its own run plus
// the number of smaller keys in the other run (one binary search per element).
constexpr int SORT_CHUNK = 8192;

__global__ __launch_bounds__(256) void sort_chunk_keys_kernel(const float* __restrict__ dist, int ng,
                                                              unsigned long long* __restrict__ keys_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int q = blockIdx.x, c0 = blockIdx.y * SORT_CHUNK;
    const int len = min(SORT_CHUNK, ng - c0);
    for (int i = threadIdx.x; i < SORT_CHUNK; i += 256)
        keys[i] = i < len ? (((unsigned long long)orderable(dist[(size_t)q * ng + c0 + i])) << 32) | (unsigned)(c0 + i)
                          : 0xFFFFFFFFFFFFFFFFull;
    __syncthreads();
    for (int k = 2; k <= SORT_CHUNK; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < SORT_CHUNK; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = keys[i], b = keys[l];
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < len; i += 256) keys_out[(size_t)q * ng + c0 + i] = keys[i];
}

__device__ __forceinline__ int count_less(const unsigned long long* __restrict__ run, int n, unsigned long long key) {
    int lo = 0, hi = n;                                  // first position whose key is >= `key` == number of smaller keys
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (run[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// one thread per element: runs of length `run` (the last one shorter) are merged in pairs
__global__ __launch_bounds__(256) void merge_runs_kernel(const unsigned long long* __restrict__ in,
                                                         unsigned long long* __restrict__ out, int nq, int ng, int run) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)nq * ng) return;
    const int q = (int)(t / ng), i = (int)(t - (long long)q * ng);
    const unsigned long long* row = in + (size_t)q * ng;
    const int pair0 = (i / (2 * run)) * (2 * run);       // start of this pair of runs
    const int a_len = min(run, ng - pair0), b0 = pair0 + a_len, b_len = max(0, min(run, ng - b0));
    const unsigned long long key = row[i];
    int pos;
    if (i < b0) pos = (i - pair0) + count_less(row + b0, b_len, key);
    else pos = (i - b0) + count_less(row + pair0, a_len, key);
    out[(size_t)q * ng + pair0 + pos] = key;
}

__global__ __launch_bounds__(256) void keys_to_order_kernel(const unsigned long long* __restrict__ keys, long long total,
                                                            int* __restrict__ order) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) order[t] = (int)(keys[t] & 0xFFFFFFFFu);
}

// one wave per query
__global__ __launch_bounds__(256) void cirr_rank_kernel(const int* __restrict__ order, int nq, int ng,
                                                        const int* __restrict__ gallery_ids,
                                                        const int* __restrict__ ref_ids,
                                                        const int* __restrict__ target_ids, int* __restrict__ rank_out,
                                                        int* __restrict__ counts_out) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    const int rid = ref_ids[q], tid_ = target_ids[q];
    int removed = 0, nref = 0, ntgt = 0, rank = -1;
    for (int base = 0; base < ng; base += 64) {
        const int i = base + lane;
        const int id = i < ng ? gallery_ids[order[(size_t)q * ng + i]] : -2147483647;
        const bool is_ref = i < ng && id == rid;
        const bool is_tgt = i < ng && id == tid_ && !is_ref;
        const unsigned long long mref = __ballot(is_ref);
        const unsigned long long mtgt = __ballot(is_tgt);
        if (mtgt && rank < 0) {
            const int first = __ffsll((long long)mtgt) - 1;
            const int refs_before = __popcll(mref & ((1ull << first) - 1ull));
            rank = base + first - removed - refs_before;
        }
        removed += __popcll(mref);
        nref += __popcll(mref);
        ntgt += __popcll(mtgt);
    }
    if (lane == 0) {
        rank_out[q] = rank;
        if (counts_out) {
            counts_out[2 * q] = nref;
            counts_out[2 * q + 1] = ntgt;
        }
    }
}

struct KList {
    int k[8];
};

// One wave per query: hits[q, j] = #{ i < ks[j] : label[order[q,i]] == qlabel[q] },  total[q] = same over all i.
// (get_metrics_imgnet, src/eval_utils.py:1090-1134: the consistency / num_correct / num_total sums)
__global__ void label_hits_kernel(const int32_t* __restrict__ order, int nq, int ng,
                                  const int32_t* __restrict__ gallery_labels, const int32_t* __restrict__ query_labels,
                                  KList ks, int nk, int32_t* __restrict__ hits_out, int32_t* __restrict__ total_out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int ql = query_labels[q];
    int hits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int total = 0;
    for (int i = lane; i < ng; i += 64) {
        const int m = gallery_labels[order[(size_t)q * ng + i]] == ql ? 1 : 0;
        total += m;
#pragma unroll
        for (int j = 0; j < 8; ++j) hits[j] += (j < nk && i < ks.k[j]) ? m : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        total += __shfl_xor(total, o, 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) hits[j] += __shfl_xor(hits[j], o, 64);
    }
    if (lane == 0) {
        total_out[q] = total;
        for (int j = 0; j < nk; ++j) hits_out[(size_t)q * nk + j] = hits[j];
    }
}

int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace

extern "C" size_t keds_rank_gallery_workspace_bytes(int nq, int ng) {
    if (nq <= 0 || ng <= 0) return 0;
    const size_t d = keds_align_up((size_t)nq * ng * sizeof(float), 256);
    if (ng <= SORT_CHUNK) return d;
    return d + 2 * keds_align_up((size_t)nq * ng * sizeof(unsigned long long), 256);   // two key buffers for the run merges
}

extern "C" int keds_rank_gallery(const float* ref, int nq, const float* gallery, int ng, int dim, int32_t* order,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    KEDS_REQUIRE(ref && gallery && order && workspace && nq > 0 && ng > 0 && dim > 0, "keds_rank_gallery: bad argument");
    KEDS_REQUIRE((long long)nq * ng < (1LL << 40), "keds_rank_gallery: problem too large");
    if (workspace_bytes < keds_rank_gallery_workspace_bytes(nq, ng)) {
        keds_set_error("keds_rank_gallery: workspace too small");
        return KEDS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* dist = (float*)workspace;
    dim3 grid((ng + 63) / 64, (nq + 63) / 64);
    dist_kernel<<<grid, 256, 0, st>>>(ref, nq, gallery, ng, dim, dist);
    int rc = keds_check_launch("dist_kernel");
    if (rc) return rc;
    if (ng > SORT_CHUNK) {
        if ((rc = keds_func_lds_once((const void*)sort_chunk_keys_kernel, 65536, "sort_chunk_keys_kernel"))) return rc;
        const size_t db = keds_align_up((size_t)nq * ng * sizeof(float), 256);
        const size_t kb = keds_align_up((size_t)nq * ng * sizeof(unsigned long long), 256);
        unsigned long long* ka = (unsigned long long*)((char*)workspace + db);
        unsigned long long* kbuf = (unsigned long long*)((char*)workspace + db + kb);
        const int chunks = (ng + SORT_CHUNK - 1) / SORT_CHUNK;
        sort_chunk_keys_kernel<<<dim3(nq, chunks), 256, 65536, st>>>(dist, ng, ka);
        if ((rc = keds_check_launch("sort_chunk_keys_kernel"))) return rc;
        const long long total = (long long)nq * ng;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        for (int run = SORT_CHUNK; run < ng; run *= 2) {
            merge_runs_kernel<<<blocks, 256, 0, st>>>(ka, kbuf, nq, ng, run);
            if ((rc = keds_check_launch("merge_runs_kernel"))) return rc;
            unsigned long long* t = ka;
            ka = kbuf;
            kbuf = t;
        }
        keys_to_order_kernel<<<blocks, 256, 0, st>>>(ka, total, order);
        return keds_check_launch("keys_to_order_kernel");
    }
    const int P = next_pow2(ng);
    const size_t lds = (size_t)P * 8;
    if ((rc = keds_func_lds_once((const void*)sort_rows_kernel, 65536, "sort_rows_kernel"))) return rc;
    sort_rows_kernel<<<nq, 256, lds, st>>>(dist, ng, P, order);
    return keds_check_launch("sort_rows_kernel");
}

extern "C" int keds_cirr_target_rank(const int32_t* order, int nq, int ng, const int32_t* gallery_ids,
                                     const int32_t* ref_ids, const int32_t* target_ids, int32_t* rank_out,
                                     int32_t* counts_out, void* stream) {
    KEDS_REQUIRE(order && gallery_ids && ref_ids && target_ids && rank_out && nq > 0 && ng > 0,
                 "keds_cirr_target_rank: bad argument");
    cirr_rank_kernel<<<(nq + 3) / 4, 256, 0, (hipStream_t)stream>>>(order, nq, ng, gallery_ids, ref_ids, target_ids,
                                                                    rank_out, counts_out);
    return keds_check_launch("cirr_rank_kernel");
}

extern "C" int keds_label_hits(const int32_t* order, int nq, int ng, const int32_t* gallery_labels,
                               const int32_t* query_labels, const int32_t* ks, int nk, int32_t* hits_out,
                               int32_t* total_out, void* stream) {
    KEDS_REQUIRE(order && gallery_labels && query_labels && ks && hits_out && total_out && nq > 0 && ng > 0,
                 "keds_label_hits: bad argument");
    KEDS_REQUIRE(nk >= 1 && nk <= 8, "keds_label_hits: 1..8 cut-offs");
    KList kl;
    for (int j = 0; j < 8; ++j) kl.k[j] = j < nk ? ks[j] : 0;
    label_hits_kernel<<<(nq + 3) / 4, 256, 0, (hipStream_t)stream>>>(order, nq, ng, gallery_labels, query_labels, kl, nk,
                                                                     hits_out, total_out);
    return keds_check_launch("label_hits_kernel");
}
