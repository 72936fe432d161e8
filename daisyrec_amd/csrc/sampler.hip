// Uniform negative sampler + CSR builder + epoch permutation.
//
// Reference: daisy/utils/sampler.py:55-103 (uniform branch :82-89) draws, for
// EVERY user id, `num_ng` items uniformly WITH replacement from
// setdiff1d(arange(item_num), train_ur[u]) — an O(U*I) host loop.  Here each
// (user, k) is one thread: a counter-based Philox4x32-10 draw r in
// [0, I-deg(u)) is mapped to the r-th element of the complement by a binary
// search over the user's sorted CSR row (exactly uniform, no rejection loop,
// O(log deg)).  daisy/utils/utils.py:19-34 (get_ur) becomes a device CSR build.
#include "common.h"

namespace daisy {

__global__ void k_pack_pairs(const int32_t *__restrict__ users, const int32_t *__restrict__ items,
                             int64_t n, uint64_t *__restrict__ keys) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        keys[e] = ((uint64_t)(uint32_t)users[e] << 32) | (uint32_t)items[e];
}

__global__ void k_unpack_items(const uint64_t *__restrict__ keys, int64_t n,
                               int32_t *__restrict__ csr_items) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        csr_items[e] = (int32_t)(uint32_t)keys[e];
}

// indptr[u] = first position whose user >= u  (u = 0..U)
__global__ void k_indptr(const uint64_t *__restrict__ keys, int64_t n, int64_t U,
                         int64_t *__restrict__ indptr) {
    for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u <= U;
         u += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t target = (uint64_t)u << 32;
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        indptr[u] = lo;
    }
}

// r-th (0-based) element of {0,1,..} \ row  (row sorted, duplicate free):
// smallest t with row[t]-t > r, answer r+t
__device__ __forceinline__ int32_t kth_in_complement(const int32_t *__restrict__ row, int64_t deg,
                                                     int64_t r) {
    int64_t lo = 0, hi = deg;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row[mid] - mid > r) hi = mid;
        else lo = mid + 1;
    }
    return (int32_t)(r + lo);
}

__global__ void k_sample_per_user(const int64_t *__restrict__ indptr,
                                  const int32_t *__restrict__ csr_items, int64_t U, int64_t I,
                                  int num_ng, uint64_t seed, uint64_t epoch,
                                  int32_t *__restrict__ js) {
    const int64_t n = U * num_ng;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = e / num_ng;
        const int64_t lo = indptr[u], deg = indptr[u + 1] - lo;
        const int64_t free_ = I - deg;
        if (free_ <= 0) { js[e] = -1; continue; }
        const uint64_t x = philox_u64(seed, epoch, (uint64_t)e);
        const int64_t r = (int64_t)__umul64hi(x, (uint64_t)free_);
        js[e] = kth_in_complement(csr_items + lo, deg, r);
    }
}

// out[r][col0 + c] = smallest item i with cdf[i] > x * cdf[I-1], x uniform in [0,1) from Philox(seed, stream, r*k + c):
// np.random.choice(np.arange(I), size=k, p=prob) of sampler.py:76-80 (inverse CDF; the user's positives are NOT
// excluded there either)
__global__ void k_sample_categorical(const double *__restrict__ cdf, int64_t I, int64_t rows, int k, uint64_t seed,
                                     uint64_t stream, int32_t *__restrict__ out, int ld, int col0) {
    const int64_t n = rows * k;
    const double total = cdf[I - 1];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t x = philox_u64(seed, stream, (uint64_t)e);
        const double t = (double)(x >> 11) * (1.0 / 9007199254740992.0) * total;     // 53 random bits
        int64_t lo = 0, hi = I - 1;                      // the last item catches t == total after rounding
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] > t) hi = mid; else lo = mid + 1;
        }
        out[(e / k) * ld + col0 + (e % k)] = (int32_t)lo;
    }
}

// SkipGramNegativeSampler.sampling (sampler.py:133-155): for the element at position p of user u's sequence
// (the user's items in train-set order) the (target, context, 1) rows of its window [p-w, p+w] in window order,
// then as many (target, negative, 0) rows, negatives uniform from the complement of the user's row.  One thread per
// sequence element; rows [off[e], off[e+1]) of `out` are its own (off = exclusive scan of 2 * window size).
__global__ void k_skipgram_fill(const int32_t *__restrict__ seq_items, const int32_t *__restrict__ seq_user,
                                const int64_t *__restrict__ seq_ptr, const int64_t *__restrict__ off, int64_t n, int w,
                                const int64_t *__restrict__ ur_ptr, const int32_t *__restrict__ ur_items, int64_t I,
                                uint64_t seed, uint64_t stream, int32_t *__restrict__ out, int *__restrict__ bad) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t u = seq_user[e], target = seq_items[e];
        const int64_t a = seq_ptr[u], b = seq_ptr[u + 1], p = e - a;
        int64_t o = off[e];
        const int64_t first = o;
        for (int64_t j = (p - w > 0 ? p - w : 0); j <= p + w && j < b - a; ++j) {
            if (j == p) continue;
            out[3 * o] = target; out[3 * o + 1] = seq_items[a + j]; out[3 * o + 2] = 1;
            ++o;
        }
        const int64_t c = o - first;                                    // = (off[e+1] - off[e]) / 2
        const int64_t lo = ur_ptr[u], deg = ur_ptr[u + 1] - lo, free_ = I - deg;
        for (int64_t k = 0; k < c; ++k) {
            int32_t neg = -1;
            if (free_ > 0) {
                const uint64_t x = philox_u64(seed, stream, (uint64_t)(first / 2 + k));
                neg = kth_in_complement(ur_items + lo, deg, (int64_t)__umul64hi(x, (uint64_t)free_));
            } else {
                atomicOr(bad, 1);
            }
            out[3 * o] = target; out[3 * o + 1] = neg; out[3 * o + 2] = 0;
            ++o;
        }
    }
}

__global__ void k_expand_triples(const int32_t *__restrict__ users, const int32_t *__restrict__ items,
                                 int64_t n, const int32_t *__restrict__ js, int num_ng,
                                 int32_t *__restrict__ triples) {
    const int64_t m = n * num_ng;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < m;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / num_ng;
        const int k = (int)(e % num_ng);
        const int32_t u = users[row];
        triples[3 * e + 0] = u;
        triples[3 * e + 1] = items[row];
        triples[3 * e + 2] = js[(int64_t)u * num_ng + k];
    }
}

__global__ void k_resample_per_interaction(const int64_t *__restrict__ indptr,
                                           const int32_t *__restrict__ csr_items, int64_t I,
                                           int32_t *__restrict__ triples, int64_t n, uint64_t seed,
                                           uint64_t stream) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = triples[3 * e];
        const int64_t lo = indptr[u], deg = indptr[u + 1] - lo;
        const int64_t free_ = I - deg;
        if (free_ <= 0) { triples[3 * e + 2] = -1; continue; }
        const uint64_t x = philox_u64(seed, stream, (uint64_t)e);
        const int64_t r = (int64_t)__umul64hi(x, (uint64_t)free_);
        triples[3 * e + 2] = kth_in_complement(csr_items + lo, deg, r);
    }
}

__global__ void k_perm_keys(int64_t n, uint64_t seed, uint64_t stream, uint64_t *__restrict__ keys,
                            int64_t *__restrict__ vals) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        keys[e] = philox_u64(seed, stream, (uint64_t)e);
        vals[e] = e;
    }
}

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
constexpr uint64_t kStreamInteraction = 1ull << 63;
constexpr uint64_t kStreamPerm = 1ull << 62;

}  // namespace daisy

using namespace daisy;

extern "C" {

size_t daisy_csr_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return align_up((size_t)n * 8) * 2 + align_up(sort_keys_u64_temp_bytes(n));
}

int daisy_build_user_csr(const int32_t *users, const int32_t *items, int64_t n, int64_t user_num,
                         int64_t *indptr, int32_t *csr_items, void *workspace, size_t workspace_bytes,
                         daisy_stream_t stream) {
    DAISY_CHECK_ARG(users && items && indptr && csr_items && workspace && n > 0 && user_num > 0,
                    "build_user_csr: bad argument");
    DAISY_CHECK_ARG(workspace_bytes >= daisy_csr_workspace_bytes(n), "build_user_csr: workspace too small");
    hipStream_t s = S(stream);
    char *w = (char *)workspace;
    uint64_t *kin = (uint64_t *)w;   w += align_up((size_t)n * 8);
    uint64_t *kout = (uint64_t *)w;  w += align_up((size_t)n * 8);
    hipLaunchKernelGGL(k_pack_pairs, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, users, items, n, kin);
    DAISY_LAUNCH_CHECK();
    int rc = sort_keys_u64(w, sort_keys_u64_temp_bytes(n), kin, kout, n, 32 + bits_for(user_num), s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_items, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, kout, n, csr_items);
    DAISY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_indptr, dim3(grid_for(user_num + 1, kBlock)), dim3(kBlock), 0, s, kout, n,
                       user_num, indptr);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_sample_neg_per_user(const int64_t *indptr, const int32_t *csr_items, int64_t user_num,
                              int64_t item_num, int32_t num_ng, uint64_t seed, uint64_t epoch,
                              int32_t *js, daisy_stream_t stream) {
    DAISY_CHECK_ARG(indptr && csr_items && js && user_num > 0 && item_num > 0 && num_ng > 0,
                    "sample_neg_per_user: bad argument");
    DAISY_CHECK_ARG(epoch < kStreamPerm, "sample_neg_per_user: epoch out of range");
    hipLaunchKernelGGL(k_sample_per_user, dim3(grid_for(user_num * num_ng, kBlock)), dim3(kBlock), 0,
                       S(stream), indptr, csr_items, user_num, item_num, (int)num_ng, seed, epoch, js);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_skipgram_samples(const int32_t *seq_items, const int32_t *seq_user, const int64_t *seq_ptr,
                           const int64_t *row_offsets, int64_t n, int32_t context_window, const int64_t *ur_indptr,
                           const int32_t *ur_items, int64_t item_num, uint64_t seed, uint64_t stream_id, int32_t *out,
                           int32_t *bad_flag, daisy_stream_t stream) {
    DAISY_CHECK_ARG(seq_items && seq_user && seq_ptr && row_offsets && ur_indptr && ur_items && out && bad_flag &&
                        n > 0 && context_window > 0 && item_num > 0, "skipgram_samples: bad argument");
    hipLaunchKernelGGL(k_skipgram_fill, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, S(stream), seq_items, seq_user, seq_ptr,
                       row_offsets, n, (int)context_window, ur_indptr, ur_items, item_num, seed, stream_id, out, bad_flag);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_sample_categorical(const double *cdf, int64_t item_num, int64_t rows, int32_t k, uint64_t seed,
                             uint64_t stream_id, int32_t *out, int32_t ld, int32_t col0, daisy_stream_t stream) {
    DAISY_CHECK_ARG(cdf && out && item_num > 0 && rows > 0 && k > 0 && ld >= col0 + k && col0 >= 0,
                    "sample_categorical: bad argument");
    hipLaunchKernelGGL(k_sample_categorical, dim3(grid_for(rows * k, kBlock)), dim3(kBlock), 0, S(stream), cdf, item_num,
                       rows, (int)k, seed, stream_id, out, (int)ld, (int)col0);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_expand_triples(const int32_t *users, const int32_t *items, int64_t n, const int32_t *js,
                         int32_t num_ng, int32_t *triples, daisy_stream_t stream) {
    DAISY_CHECK_ARG(users && items && js && triples && n > 0 && num_ng > 0, "expand_triples: bad argument");
    hipLaunchKernelGGL(k_expand_triples, dim3(grid_for(n * num_ng, kBlock)), dim3(kBlock), 0, S(stream),
                       users, items, n, js, (int)num_ng, triples);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_resample_neg_per_interaction(const int64_t *indptr, const int32_t *csr_items,
                                       int64_t item_num, int32_t *triples, int64_t n, uint64_t seed,
                                       uint64_t epoch, daisy_stream_t stream) {
    DAISY_CHECK_ARG(indptr && csr_items && triples && n > 0 && item_num > 0,
                    "resample_neg_per_interaction: bad argument");
    DAISY_CHECK_ARG(epoch < kStreamPerm, "resample_neg_per_interaction: epoch out of range");
    hipLaunchKernelGGL(k_resample_per_interaction, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, S(stream),
                       indptr, csr_items, item_num, triples, n, seed, epoch | kStreamInteraction);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

size_t daisy_randperm_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return align_up((size_t)n * 8) * 3 + align_up(sort_pairs_u64_i64_temp_bytes(n));
}

int daisy_randperm(int64_t n, uint64_t seed, uint64_t epoch, int64_t *perm, void *workspace,
                   size_t workspace_bytes, daisy_stream_t stream) {
    DAISY_CHECK_ARG(perm && workspace && n > 0, "randperm: bad argument");
    DAISY_CHECK_ARG(epoch < kStreamPerm, "randperm: epoch out of range");
    DAISY_CHECK_ARG(workspace_bytes >= daisy_randperm_workspace_bytes(n), "randperm: workspace too small");
    hipStream_t s = S(stream);
    char *w = (char *)workspace;
    uint64_t *kin = (uint64_t *)w;   w += align_up((size_t)n * 8);
    uint64_t *kout = (uint64_t *)w;  w += align_up((size_t)n * 8);
    int64_t *vin = (int64_t *)w;     w += align_up((size_t)n * 8);
    hipLaunchKernelGGL(k_perm_keys, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, seed,
                       epoch | kStreamPerm, kin, vin);
    DAISY_LAUNCH_CHECK();
    return sort_pairs_u64_i64(w, sort_pairs_u64_i64_temp_bytes(n), kin, kout, vin, perm, n, 64, s);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Candidate sets for ranking (reference: daisy/utils/utils.py:53-85,
// build_candidates_set): per test user `cand_num` items = uniform negatives
// drawn WITH replacement from the items in neither the user's test row nor
// train row, followed by the test (ground-truth) items; a user with more than
// cand_num truths gets cand_num draws from the truths instead.
// The reference loops over users with np.setdiff1d (O(U*I)); here one thread per
// candidate: the r-th free item is found by a binary search on the item id with
// two rank queries (test row, train row; the rows are disjoint by construction
// of the split).  Truth items are appended in ascending id order.
// ---------------------------------------------------------------------------
namespace daisy {

__device__ __forceinline__ int64_t count_le(const int32_t *__restrict__ row, int64_t deg, int64_t x) {
    int64_t lo = 0, hi = deg;       // number of row elements <= x
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void k_build_candidates(const int64_t *__restrict__ ip_te, const int32_t *__restrict__ it_te,
                                   const int64_t *__restrict__ ip_tr, const int32_t *__restrict__ it_tr,
                                   const int64_t *__restrict__ users, int64_t n_users, int64_t I,
                                   int cand_num, uint64_t seed, int64_t *__restrict__ out) {
    const int64_t total = n_users * cand_num;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cand_num;
        const int k = (int)(e % cand_num);
        const int64_t u = users[row];
        const int32_t *te = it_te + ip_te[u];
        const int64_t dte = ip_te[u + 1] - ip_te[u];
        const uint64_t x = philox_u64(seed, 1ull << 60, (uint64_t)e);
        if (dte > cand_num) {                       // utils.py:72-73: draws from the truths
            out[e] = te[(int64_t)__umul64hi(x, (uint64_t)dte)];
            continue;
        }
        const int64_t n_neg = cand_num - dte;
        if (k >= n_neg) {                           // utils.py:79: ... followed by the truths
            out[e] = te[k - n_neg];
            continue;
        }
        const int32_t *tr = it_tr + ip_tr[u];
        const int64_t dtr = ip_tr[u + 1] - ip_tr[u];
        const int64_t free_ = I - dte - dtr;
        if (free_ <= 0) { out[e] = -1; continue; }
        const int64_t r = (int64_t)__umul64hi(x, (uint64_t)free_);   // r-th free item, 0-based
        int64_t lo = 0, hi = I - 1;                 // smallest id with (#free ids <= id) >= r+1
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const int64_t free_le = mid + 1 - count_le(te, dte, mid) - count_le(tr, dtr, mid);
            if (free_le >= r + 1) hi = mid;
            else lo = mid + 1;
        }
        out[e] = lo;
    }
}

}  // namespace daisy

extern "C" int daisy_build_candidates(const int64_t *indptr_test, const int32_t *items_test,
                                      const int64_t *indptr_train, const int32_t *items_train,
                                      const int64_t *users, int64_t n_users, int64_t item_num,
                                      int32_t cand_num, uint64_t seed, int64_t *out,
                                      daisy_stream_t stream) {
    DAISY_CHECK_ARG(indptr_test && items_test && indptr_train && items_train && users && out &&
                        n_users > 0 && item_num > 0 && cand_num > 0,
                    "build_candidates: bad argument");
    hipLaunchKernelGGL(daisy::k_build_candidates, dim3(daisy::grid_for(n_users * cand_num, daisy::kBlock)),
                       dim3(daisy::kBlock), 0, reinterpret_cast<hipStream_t>(stream), indptr_test, items_test,
                       indptr_train, items_train, users, n_users, item_num, (int)cand_num, seed, out);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}
