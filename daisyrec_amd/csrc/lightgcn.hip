// LightGCN (daisy/model/LightGCNRecommender.py) on gfx950: the normalised bipartite adjacency as a
// row-sorted entry list on the device, the propagation out = mean_k A^k E0 and its transpose.
// The sparse x dense products run on the item pass's segmented-reduction kernel (segsum_rows in
// bpr_train.hip): random 256-byte row gathers, 16 rows in flight per lane group, single-owner
// stores - an HBM-bound kernel, no MFMA.  Everything else of a LightGCN step is the MF path.
#include <stdlib.h>

#include "common.h"

namespace daisy {

__global__ void k_lg_pair_keys(const int32_t *__restrict__ users, const int32_t *__restrict__ items, int64_t n,
                               uint64_t *__restrict__ keys) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        keys[t] = ((uint64_t)(uint32_t)users[t] << 32) | (uint32_t)items[t];
}

// distinct (u, i) pairs sorted by (u, i): degree counts + the swapped keys (i, u) for the item rows
__global__ void k_lg_degrees(const uint64_t *__restrict__ pairs, int64_t m, int64_t U, int32_t *__restrict__ deg,
                             uint64_t *__restrict__ swapped) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < m; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = pairs[t];
        const uint32_t u = (uint32_t)(k >> 32), i = (uint32_t)k;
        atomicAdd(deg + u, 1);
        atomicAdd(deg + U + i, 1);
        swapped[t] = ((uint64_t)i << 32) | u;
    }
}

// entry e < m: user row (u -> U+i) from pairs[e]; entry e >= m: item row (U+i -> u) from swapped_sorted[e-m].
// value = D^-1/2 A D^-1/2 in float64, stored as float32 (LightGCNRecommender.py:93-105)
__global__ void k_lg_entries(const uint64_t *__restrict__ pairs, const uint64_t *__restrict__ swapped_sorted,
                             int64_t m, int64_t U, const int32_t *__restrict__ deg, uint32_t *__restrict__ ekey,
                             uint2 *__restrict__ esu, float2 *__restrict__ coef) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < 2 * m; e += (int64_t)gridDim.x * blockDim.x) {
        uint32_t row, col;
        if (e < m) {
            const uint64_t k = pairs[e];
            row = (uint32_t)(k >> 32);
            col = (uint32_t)U + (uint32_t)k;
        } else {
            const uint64_t k = swapped_sorted[e - m];
            row = (uint32_t)U + (uint32_t)(k >> 32);
            col = (uint32_t)k;
        }
        const double dr = pow((double)deg[row] + 1e-7, -0.5), dc = pow((double)deg[col] + 1e-7, -0.5);
        ekey[e] = row << 1;
        esu[e] = make_uint2((uint32_t)e, col);
        coef[e] = make_float2((float)((dr * 1.0) * dc), 0.f);
    }
}

__global__ void k_lg_read(const uint32_t *__restrict__ ekey, const uint2 *__restrict__ esu,
                          const float2 *__restrict__ coef, int64_t n, int32_t *__restrict__ row,
                          int32_t *__restrict__ col, float *__restrict__ val) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        row[e] = (int32_t)(ekey[e] >> 1);
        col[e] = (int32_t)esu[e].y;
        val[e] = coef[e].x;
    }
}

// reproducible product: the lane group that sees the first entry of a row owns it and sums the row's
// entries in stored order (fixed, so bitwise reproducible; serial over long rows - the throughput
// path is segsum_rows)
__global__ __launch_bounds__(kBlock) void k_lg_spmm_owner(const uint32_t *__restrict__ ekey,
                                                          const uint2 *__restrict__ esu,
                                                          const float2 *__restrict__ coef, int64_t n,
                                                          const float *__restrict__ X, int d,
                                                          float *__restrict__ Y) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int64_t pos = (int64_t)blockIdx.x * (kBlock / 16) + group; pos < n; pos += gstride) {
        const uint32_t key = ekey[pos];
        if (pos > 0 && ekey[pos - 1] == key) continue;          // not the head of its row
        const int64_t row = key >> 1;
        for (int c = lane; c < d; c += 16) {
            float acc = 0.f;
            for (int64_t q = pos; q < n && ekey[q] == key; ++q)
                acc = fmaf(coef[q].x, X[(int64_t)esu[q].y * d + c], acc);
            Y[row * d + c] = acc;
        }
    }
}

// y = a*x + b*y (elementwise; x may be null for y *= b)
__global__ void k_lg_axpby(const float *__restrict__ x, float a, float b, float *__restrict__ y, int64_t n) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        y[e] = (x ? a * x[e] : 0.f) + b * y[e];
}

// out[r] = sum of X[cols[e]] over e in [indptr[r], indptr[r+1]) - one lane group per row (rows here are a
// user's training items: tens to a few thousand entries)
__global__ __launch_bounds__(kBlock) void k_csr_row_sum(const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ cols,
                                                        const float *__restrict__ X, int64_t R, int d,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
        const int64_t e0 = indptr[r], e1 = indptr[r + 1];
        if (e0 == e1) continue;                          // rows without entries keep their contents
        for (int c = lane; c < d; c += 16) {
            float s = 0.f;
            for (int64_t e = e0; e < e1; ++e) s += X[(int64_t)cols[e] * d + c];
            out[r * d + c] = s;
        }
    }
}

__global__ void k_axpby_zero(float *__restrict__ x, float a, float b, float *__restrict__ y, int64_t n, int zero_x) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        y[e] = a * x[e] + b * y[e];
        if (zero_x) x[e] = 0.f;
    }
}

// regulariser gradient on the ego rows of a batch, in two deterministic passes: integer occurrence counts per
// node (as user | positive item, as negative item), then ONE update per touched row
//   dE0[row] += n_a * (reg_1*sign(e) + r_a*e) + n_b * (reg_1*sign(e) + r_b*e)
// (fp32 atomics would add the positive-slot and negative-slot terms of an item in arrival order: the replicas
// of a multi-GPU training drift apart by an ulp per step)
__global__ void k_lg_reg_count(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                               const int32_t *__restrict__ j, int64_t B, int64_t U, int pointwise,
                               int32_t *__restrict__ cnt) {
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        atomicAdd(cnt + 2 * (int64_t)u[b], 1);
        atomicAdd(cnt + 2 * (U + i[b]), 1);
        if (!pointwise) atomicAdd(cnt + 2 * (U + j[b]) + 1, 1);
    }
}

__global__ __launch_bounds__(kBlock) void k_lg_reg_apply(const float *__restrict__ E0, int64_t N, int64_t U, int d,
                                                         float reg_1, float reg_2, const double *__restrict__ stats,
                                                         int32_t *__restrict__ cnt, float *__restrict__ dE0) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    auto inv = [&](double x) { return x > 0.0 ? (float)((double)reg_2 / x) : 0.f; };   // d|X|_F/dX = 0 at X = 0
    const float r_u = inv(stats[DAISY_ST_NORM_U]), r_i = inv(stats[DAISY_ST_NORM_I]), r_j = inv(stats[DAISY_ST_NORM_J]);
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 16) + group; row < N; row += gstride) {
        const int na = cnt[2 * row], nb = cnt[2 * row + 1];
        if (na == 0 && nb == 0) continue;
        const float fa = (float)na, fb = (float)nb;
        const float ra = row < U ? r_u : r_i;
        for (int c = lane; c < d; c += 16) {
            const float e = E0[row * d + c];
            const float g = fa * fmaf(ra, e, reg_1 * sgn(e)) + fb * fmaf(r_j, e, reg_1 * sgn(e));
            dE0[row * d + c] += g;
        }
        if (lane == 0) { cnt[2 * row] = 0; cnt[2 * row + 1] = 0; }      // leave the workspace clean
    }
}

}  // namespace daisy

using namespace daisy;

struct daisy_lgcn_graph {
    int64_t U, I, nnz;
    void *arena;
    size_t arena_bytes;
    uint32_t *ekey;
    uint2 *esu;
    float2 *coef;
    int reproducible;     // daisy_lgcn_graph_set_reproducible
    void *edge_arena;     // scratch of the segmented reduction (edge records), sized for edge_d
    int edge_d;
    float *edge_vec, *edge_b;
    int32_t *edge_item, *edge_whole;
    int64_t *row_ptr_host;   // [N+1] first entry of every row (lazy: daisy_lgcn_spmm_rows)
};

// row_ptr[r] = first entry e with ekey[e] >= r << 1
__global__ void k_lg_row_ptr(const uint32_t *__restrict__ ekey, int64_t nnz, int64_t N, int64_t *__restrict__ row_ptr) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= N; r += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t target = (uint64_t)r << 1;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint64_t)ekey[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        row_ptr[r] = lo;
    }
}

// (re)allocate the edge-record scratch of the products for row width d
static int ensure_edges(daisy_lgcn_graph *g, int d) {
    if (g->edge_arena && g->edge_d == d) return DAISY_OK;
    if (g->edge_arena) { (void)hipFree(g->edge_arena); g->edge_arena = nullptr; }
    const size_t chunks = (size_t)segsum_chunks(g->nnz, d) + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_v = take(2 * chunks * (size_t)d * 4), o_i = take(2 * chunks * 4), o_b = take(2 * chunks * 4),
                 o_w = take(chunks * 4);
    DAISY_HIP(hipMalloc(&g->edge_arena, off));
    char *base = (char *)g->edge_arena;
    g->edge_vec = (float *)(base + o_v); g->edge_item = (int32_t *)(base + o_i);
    g->edge_b = (float *)(base + o_b); g->edge_whole = (int32_t *)(base + o_w);
    g->edge_d = d;
    return DAISY_OK;
}

static inline hipStream_t LS(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int daisy_lgcn_graph_create(daisy_lgcn_graph **out, const int32_t *users, const int32_t *items, int64_t n,
                            int64_t user_num, int64_t item_num, daisy_stream_t stream) {
    DAISY_CHECK_ARG(out != nullptr, "lgcn_graph_create: out is NULL");
    *out = nullptr;
    DAISY_CHECK_ARG(users && items && n > 0 && user_num > 0 && item_num > 0 &&
                        user_num + item_num < ((int64_t)1 << 31),
                    "lgcn_graph_create: bad argument");
    hipStream_t s = LS(stream);
    const int64_t N = user_num + item_num;
    // scratch: keys, sorted keys, unique pairs, swapped, swapped sorted, counts, runs, degrees, sort temp
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_k0 = take(n * 8), o_k1 = take(n * 8), o_un = take(n * 8), o_sw = take(n * 8), o_ss = take(n * 8);
    const size_t o_cnt = take(n * 4), o_runs = take(16), o_deg = take((size_t)N * 4);
    size_t tb = sort_keys_u64_temp_bytes(n);
    if (rle_u64_temp_bytes(n) > tb) tb = rle_u64_temp_bytes(n);
    const size_t o_tmp = take(tb);
    char *scratch = nullptr;
    DAISY_HIP(hipMalloc((void **)&scratch, off));
    auto fail = [&](int rc) { (void)hipFree(scratch); return rc; };
    uint64_t *k0 = (uint64_t *)(scratch + o_k0), *k1 = (uint64_t *)(scratch + o_k1), *un = (uint64_t *)(scratch + o_un);
    uint64_t *sw = (uint64_t *)(scratch + o_sw), *ss = (uint64_t *)(scratch + o_ss);
    uint32_t *cnt = (uint32_t *)(scratch + o_cnt), *runs = (uint32_t *)(scratch + o_runs);
    int32_t *deg = (int32_t *)(scratch + o_deg);
    void *tmp = scratch + o_tmp;
    hipLaunchKernelGGL(k_lg_pair_keys, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, s, users, items, n, k0);
    int rc = sort_keys_u64(tmp, tb, k0, k1, n, 64, s);
    if (rc) return fail(rc);
    rc = rle_u64(tmp, tb, k1, n, un, cnt, runs, s);
    if (rc) return fail(rc);
    uint32_t m32 = 0;
    if (hipMemcpyAsync(&m32, runs, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("lgcn_graph_create: reading the pair count failed");
        return fail(DAISY_ERR_HIP);
    }
    const int64_t m = m32;
    daisy_lgcn_graph *g = new daisy_lgcn_graph();
    g->U = user_num; g->I = item_num; g->nnz = 2 * m;
    g->reproducible = 0;
    g->edge_arena = nullptr; g->edge_d = 0;
    g->row_ptr_host = nullptr;
    size_t goff = 0;
    auto gtake = [&](size_t bytes) { size_t o = goff; goff += align_up(bytes); return o; };
    const size_t g_k = gtake((size_t)g->nnz * 4), g_s = gtake((size_t)g->nnz * 8), g_c = gtake((size_t)g->nnz * 8);
    g->arena_bytes = goff;
    if (hipMalloc(&g->arena, goff) != hipSuccess) {
        set_error("lgcn_graph_create: hipMalloc(%zu) failed", goff);
        delete g;
        return fail(DAISY_ERR_HIP);
    }
    g->ekey = (uint32_t *)((char *)g->arena + g_k);
    g->esu = (uint2 *)((char *)g->arena + g_s);
    g->coef = (float2 *)((char *)g->arena + g_c);
    (void)hipMemsetAsync(deg, 0, (size_t)N * 4, s);
    hipLaunchKernelGGL(k_lg_degrees, dim3(grid_for(m, kBlock * 4)), dim3(kBlock), 0, s, un, m, user_num, deg, sw);
    rc = sort_keys_u64(tmp, tb, sw, ss, m, 64, s);
    if (rc) { (void)hipFree(g->arena); delete g; return fail(rc); }
    hipLaunchKernelGGL(k_lg_entries, dim3(grid_for(2 * m, kBlock * 4)), dim3(kBlock), 0, s, un, ss, m, user_num, deg,
                       g->ekey, g->esu, g->coef);
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("lgcn_graph_create: kernel failure");
        (void)hipFree(g->arena); delete g;
        return fail(DAISY_ERR_HIP);
    }
    (void)hipFree(scratch);
    *out = g;
    return DAISY_OK;
}

int daisy_lgcn_graph_destroy(daisy_lgcn_graph *g) {
    if (!g) return DAISY_OK;
    if (g->arena) (void)hipFree(g->arena);
    if (g->edge_arena) (void)hipFree(g->edge_arena);
    if (g->row_ptr_host) free(g->row_ptr_host);
    delete g;
    return DAISY_OK;
}

int daisy_lgcn_graph_set_reproducible(daisy_lgcn_graph *g, int32_t flag) {
    DAISY_CHECK_ARG(g != nullptr, "lgcn_graph_set_reproducible: NULL graph");
    g->reproducible = flag ? 1 : 0;
    return DAISY_OK;
}

int64_t daisy_lgcn_graph_nnz(const daisy_lgcn_graph *g) { return g ? g->nnz : 0; }
size_t daisy_lgcn_graph_bytes(const daisy_lgcn_graph *g) { return g ? g->arena_bytes : 0; }

int daisy_lgcn_graph_read(const daisy_lgcn_graph *g, int32_t *row, int32_t *col, float *val,
                          daisy_stream_t stream) {
    DAISY_CHECK_ARG(g && row && col && val, "lgcn_graph_read: NULL argument");
    hipLaunchKernelGGL(k_lg_read, dim3(grid_for(g->nnz, kBlock * 4)), dim3(kBlock), 0, LS(stream), g->ekey, g->esu,
                       g->coef, g->nnz, row, col, val);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_lgcn_spmm(const daisy_lgcn_graph *g, const float *X, float *Y, int32_t d, daisy_stream_t stream) {
    DAISY_CHECK_ARG(g && X && Y && X != Y && d > 0, "lgcn_spmm: bad argument");
    hipStream_t s = LS(stream);
    DAISY_HIP(hipMemsetAsync(Y, 0, (size_t)(g->U + g->I) * d * 4, s));
    if (g->reproducible) {
        if (g->nnz > 0)
            hipLaunchKernelGGL(k_lg_spmm_owner, dim3(grid_for(g->nnz, kBlock / 16, kMaxGridSparse)), dim3(kBlock), 0, s,
                               g->ekey, g->esu, g->coef, g->nnz, X, (int)d, Y);
        DAISY_LAUNCH_CHECK();
        return DAISY_OK;
    }
    daisy_lgcn_graph *gm = const_cast<daisy_lgcn_graph *>(g);      // scratch only: the matrix itself is untouched
    int rc = ensure_edges(gm, d);
    if (rc) return rc;
    return segsum_rows(X, g->coef, g->ekey, g->esu, g->nnz, d, Y, gm->edge_vec, gm->edge_item, gm->edge_b,
                       gm->edge_whole, s);
}

int daisy_lgcn_spmm_rows(const daisy_lgcn_graph *g, const float *X, float *Yrows, int32_t d, int64_t row_lo,
                         int64_t row_hi, daisy_stream_t stream) {
    DAISY_CHECK_ARG(g && X && Yrows && d > 0, "lgcn_spmm_rows: bad argument");
    const int64_t N = g->U + g->I;
    DAISY_CHECK_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= N, "lgcn_spmm_rows: rows %lld..%lld outside 0..%lld",
                    (long long)row_lo, (long long)row_hi, (long long)N);
    hipStream_t s = LS(stream);
    daisy_lgcn_graph *gm = const_cast<daisy_lgcn_graph *>(g);      // caches and scratch only: the matrix is untouched
    if (!gm->row_ptr_host) {          // once: the first entry of every row (one host sync)
        int64_t *dev = nullptr;
        DAISY_HIP(hipMalloc((void **)&dev, (size_t)(N + 1) * 8));
        hipLaunchKernelGGL(k_lg_row_ptr, dim3(grid_for(N + 1, kBlock)), dim3(kBlock), 0, s, g->ekey, g->nnz, N, dev);
        gm->row_ptr_host = (int64_t *)malloc((size_t)(N + 1) * 8);
        hipError_t e = hipMemcpyAsync(gm->row_ptr_host, dev, (size_t)(N + 1) * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(dev);
        if (e != hipSuccess) {
            free(gm->row_ptr_host);
            gm->row_ptr_host = nullptr;
            set_error("lgcn_spmm_rows: reading the row offsets failed: %s", hipGetErrorString(e));
            return DAISY_ERR_HIP;
        }
    }
    if (row_hi == row_lo) return DAISY_OK;
    DAISY_HIP(hipMemsetAsync(Yrows, 0, (size_t)(row_hi - row_lo) * d * 4, s));
    int64_t e_lo = g->row_ptr_host[row_lo], e_hi = g->row_ptr_host[row_hi];
    if (e_hi == e_lo) return DAISY_OK;
    float *base = Yrows - row_lo * (int64_t)d;                     // row r of the product lands in Yrows[r - row_lo]
    // The segmented reduction takes an even number of entries: an odd range borrows one entry of the ADJACENT row, whose
    // partial sum lands in the spare row the caller provides before / after Yrows.  The entry next to the range belongs
    // to the next NON-EMPTY row, which is the adjacent one only if that row has entries (isolated nodes are common:
    // user_num / item_num come from the full dataset, the graph from the train split) - otherwise its partial sum would
    // land far outside the block, so such a block is summed by the row owners instead.
    bool owner = g->reproducible != 0;
    if (!owner && ((e_hi - e_lo) & 1)) {
        const int64_t *rp = g->row_ptr_host;
        if (row_hi < N && rp[row_hi + 1] > e_hi) ++e_hi;                  // first entry of row_hi
        else if (row_lo > 0 && rp[row_lo - 1] < e_lo) --e_lo;             // last entry of row_lo - 1
        else owner = true;
    }
    if (owner) {
        hipLaunchKernelGGL(k_lg_spmm_owner, dim3(grid_for(e_hi - e_lo, kBlock / 16, kMaxGridSparse)), dim3(kBlock), 0, s,
                           g->ekey + e_lo, g->esu + e_lo, g->coef + e_lo, e_hi - e_lo, X, (int)d, base);
        DAISY_LAUNCH_CHECK();
        return DAISY_OK;
    }
    int rc = ensure_edges(gm, d);
    if (rc) return rc;
    return segsum_rows(X, g->coef, g->ekey + e_lo, g->esu + e_lo, e_hi - e_lo, d, base, gm->edge_vec, gm->edge_item,
                       gm->edge_b, gm->edge_whole, s);
}

int daisy_lgcn_propagate(const daisy_lgcn_graph *g, const float *E0, int32_t d, int32_t num_layers,
                         float *work, float *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(g && E0 && work && out && d > 0 && num_layers >= 0, "lgcn_propagate: bad argument");
    hipStream_t s = LS(stream);
    const int64_t nel = (g->U + g->I) * (int64_t)d;
    const int grid = grid_for(nel, kBlock * 4);
    DAISY_HIP(hipMemcpyAsync(out, E0, (size_t)nel * 4, hipMemcpyDeviceToDevice, s));
    const float *x = E0;
    for (int k = 0; k < num_layers; ++k) {
        float *y = work + (int64_t)(k & 1) * nel;
        int rc = daisy_lgcn_spmm(g, x, y, d, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lg_axpby, dim3(grid), dim3(kBlock), 0, s, y, 1.f, 1.f, out, nel);   // out += E_{k+1}
        x = y;
    }
    hipLaunchKernelGGL(k_lg_axpby, dim3(grid), dim3(kBlock), 0, s, (const float *)nullptr, 0.f,
                       1.f / (float)(num_layers + 1), out, nel);                                  // mean (:126)
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_lgcn_backprop(const daisy_lgcn_graph *g, const float *G, int32_t d, int32_t num_layers, float *work,
                        float *dE0, daisy_stream_t stream) {
    DAISY_CHECK_ARG(g && G && work && dE0 && d > 0 && num_layers >= 0, "lgcn_backprop: bad argument");
    hipStream_t s = LS(stream);
    const int64_t nel = (g->U + g->I) * (int64_t)d;
    const int grid = grid_for(nel, kBlock * 4);
    const float *t = G;                               // Horner: T <- G + A T, L times
    for (int k = 0; k < num_layers; ++k) {
        float *y = work + (int64_t)(k & 1) * nel;
        int rc = daisy_lgcn_spmm(g, t, y, d, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lg_axpby, dim3(grid), dim3(kBlock), 0, s, G, 1.f, 1.f, y, nel);
        t = y;
    }
    hipLaunchKernelGGL(k_lg_axpby, dim3(grid), dim3(kBlock), 0, s, t, 1.f / (float)(num_layers + 1), 1.f, dE0, nel);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_axpby_f32(float *x, float a, float b, float *y, int64_t n, int32_t zero_x, daisy_stream_t stream) {
    DAISY_CHECK_ARG(x && y && n > 0, "axpby_f32: bad argument");
    hipLaunchKernelGGL(k_axpby_zero, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, LS(stream), x, a, b, y, n,
                       (int)zero_x);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_csr_row_sum(const int64_t *indptr, const int32_t *cols, const float *X, int64_t rows, int32_t d,
                      float *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(indptr && cols && X && out && rows > 0 && d > 0, "csr_row_sum: bad argument");
    hipLaunchKernelGGL(k_csr_row_sum, dim3(grid_for(rows, kBlock / 16)), dim3(kBlock), 0, LS(stream), indptr, cols, X,
                       rows, (int)d, out);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_lgcn_reg_grad(const float *E0, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B,
                        int64_t user_num, int64_t item_num, int32_t d, int32_t pointwise, float reg_1, float reg_2,
                        const double *stats, int32_t *count_ws, float *dE0, daisy_stream_t stream) {
    DAISY_CHECK_ARG(E0 && u && i && j && stats && dE0 && count_ws && B > 0 && d > 0 && user_num > 0 && item_num > 0,
                    "lgcn_reg_grad: bad argument");
    if (reg_1 == 0.f && reg_2 == 0.f) return DAISY_OK;
    hipStream_t s = LS(stream);
    const int64_t N = user_num + item_num;
    hipLaunchKernelGGL(k_lg_reg_count, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, u, i, j, B, user_num,
                       (int)pointwise, count_ws);
    hipLaunchKernelGGL(k_lg_reg_apply, dim3(grid_for(N, kBlock / 16 * 2)), dim3(kBlock), 0, s, E0, N, user_num, (int)d,
                       reg_1, reg_2, stats, count_ws, dE0);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // extern "C"
