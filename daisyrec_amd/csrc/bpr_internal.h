// Internal types shared by the training translation units (bpr_train.hip: the phase kernels
// over the sorted epoch plan; bpr_staged.hip: the partitioned epoch plan and the staged step).
#pragma once

#include "common.h"

namespace daisy {

constexpr uint32_t kNegBit = 0x80000000u;

// What the step kernels see of the current batch (pointers into an epoch plan).
//   sample s in [0,B):  user = ukey[s] & umask,  (pos item, neg item) = ij[s]
//                       samples of one user are contiguous (stable order)
//   entry  q in [0,2B): item = (ekey[q] & imask) >> 1, negative slot = ekey[q] & 1 (ascending, stable),
//                       esu[q] = (sample position s | kNegBit for the negative slot, user of s)
struct BatchView {
    const uint32_t *ukey;
    const int2 *ij;
    const uint32_t *ekey;
    const uint2 *esu;
    // run-length encoding of the batch's entry keys: run m in [run_off[0], run_off[1]) has
    // run_key[m] = item << 1 | neg and run_cnt[m] entries (an item owns 1 or 2 adjacent runs)
    const uint32_t *run_key, *run_cnt;
    const int32_t *run_off;
    uint32_t umask, imask;
    // point-wise losses (CL / SL, MFRecommender.py:75-81): ij[s] = (item, label); the "negative"
    // slot of a sample is an inert copy of its item (coefficient 0, not counted by the regulariser)
    int32_t pointwise;
    int64_t B;
    // FM (FMRecommender.py:61-68): score += u_bias[u] + i_bias[item] + bias_; bu == nullptr -> plain MF.
    // g_bi accumulates like gQ (zero between steps, consumed by k_item_apply); g_bu / g_b0 are only
    // written by the gradient-output user pass (Adam); the SGD user pass updates bu and b0 in place.
    float *bu, *bi, *b0;
    float *g_bu, *g_bi, *g_b0;
    // set for the duration of daisy_bpr_sgd_step / daisy_bpr_fit_epoch_sgd: the epoch's count of non-finite step
    // losses (epoch_acc[1]).  Once it is > 0 every kernel that changes a table returns at once: the epoch stops at its
    // first non-finite loss like the reference's loop (AbstractRecommender.py:122-123), with one host sync per epoch
    const double *halt;
};

__device__ __forceinline__ bool halted(const double *halt) { return halt != nullptr && *halt > 0.0; }

}  // namespace daisy

// What the STAGED step (bpr_staged.hip) sees of the current batch.  Either layout of the epoch plan
// can feed it:
//   sample s in [0,B):  partitioned layout: s_rec[s] = {user, pos item, neg item, epoch position}, stage slot of the sample
//                       = position - pos_base;   sorted layout (s_rec == NULL): user = s_user[s] & umask, (pos item, neg
//                       item) = s_ij[s], stage slot = s.  Samples of one user are contiguous; the slots are a bijection
//                       onto [0,B)
//   entry  q in [0,E):  item = (e_key[q*e_kstride] & imask) >> 1, negative slot = e_key[..] & 1 (ascending, stable);
//                       stage slot of the sample it belongs to = (e_pos[q*e_stride] & ~kNegBit) - pos_base
//                       (partitioned layout: 8-byte records {key, position}, both strides 2)
namespace daisy {
struct StreamView {
    const uint4 *s_rec;
    const uint32_t *s_user;
    const int2 *s_ij;
    const uint32_t *e_key;
    const uint32_t *e_pos;
    uint32_t umask, imask;
    int32_t e_kstride, e_stride;
    uint32_t pos_base;
    int64_t B, E;
    const double *halt;      // see BatchView::halt
    int32_t pointwise;       // rows are (user, item, label): E = B entries in the partitioned layout; the sorted
                             // layout keeps an inert negative slot per sample (E = 2B)
    int32_t p_stream;        // the user pass reads and writes P past the caches (tables far beyond them: set by the host)
};
// the sample as {user, pos item, neg item, stage slot} / the user alone / an entry's key  (any layout; the hot kernels
// read the layout they were compiled for directly)
__device__ __forceinline__ uint4 sv_sample(const StreamView &v, int64_t s) {
    if (v.s_rec) { uint4 r = v.s_rec[s]; r.w -= v.pos_base; return r; }
    const int2 ij = v.s_ij[s];
    return make_uint4(v.s_user[s] & v.umask, (uint32_t)ij.x, (uint32_t)ij.y, (uint32_t)s);
}
__device__ __forceinline__ uint32_t sv_user(const StreamView &v, int64_t s) {
    return v.s_rec ? v.s_rec[s].x : (v.s_user[s] & v.umask);
}
__device__ __forceinline__ uint32_t sv_key(const StreamView &v, int64_t e) { return v.e_key[e * v.e_kstride] & v.imask; }
}  // namespace daisy

// Static index of a training set (built once per fit): the triples in CSR (user-sorted) order and
// their item entries sorted by item.  The partitioned epoch plan is two stable one-digit partitions
// of these arrays by batch id.
struct daisy_train_index {
    int64_t n, U, I;
    int32_t user_base;
    const int32_t *triples;   // [n][3] CSR order: the caller's array, or `sorted_copy`
    int32_t *sorted_copy;     // owned copy when the caller's array was not user-sorted
    const uint32_t *orig;     // [n] row of the caller's array behind CSR row t (NULL: identity); epoch positions
                              //     (perm / Feistel / identity) always refer to the caller's rows
    uint32_t *ent_t;          // [2n] triple index | slot << 31, sorted by ent_key (stable: t ascending)
    uint32_t *ent_key;        // [2n] item << 1 | slot
    size_t bytes;
    int32_t pointwise;        // rows are (user, item, label): ONE entry per row (n_ent = n), else two (n_ent = 2n)
    int64_t n_ent;
    int64_t max_item_entries; // entries of the most frequent item (the longest segment an item pass can meet, scaled by
                              // the batch's share of the set: decides whether its edge chains are reduced in two levels)
};

// Epoch plan: the whole epoch laid out batch by batch (see the header comment of bpr_train.hip).
// Two layouts:
//   kind 0 (sorted, daisy_epoch_plan_build):          packed sort keys + run lists; every phase kernel reads it
//   kind 1 (partitioned, daisy_epoch_plan_build_indexed): plain SoA records, 32 B per interaction; staged step only
struct daisy_epoch_plan {
    int64_t max_triples, U, I;
    void *arena;          // kind 0 buffers (allocated by the first daisy_epoch_plan_build)
    size_t arena_bytes, temp_bytes;
    // double buffers of the two radix sorts
    uint32_t *k32[2];     // [2n] 32-bit keys
    uint64_t *k64[2];     // [2n] 64-bit keys (only when batch bits + id bits > 32)
    uint64_t *v64[2];     // [2n] payloads
    uint32_t *ukey;       // [n]  sorted sample keys (batch << ubits | user)
    uint64_t *uval;       // [n]  (i, j)
    uint32_t *ekey;       // [2n] sorted entry keys (batch << ibits | item)
    uint64_t *eval;       // [2n] (s | neg, u)
    uint32_t *run_key;    // [2n]  item << 1 | neg of every run of equal entry keys
    uint32_t *run_cnt;    // [2n]  its length
    int32_t *run_off;     // [max_triples+2] first run of every batch; [num_batches] = total
    uint32_t *run_total;  // [1]   number of runs (device)
    int *bad;             // [1]   bit 0: an id of the last build lay outside the tables, bit 1: a bad permutation entry
    uint32_t umask, imask;
    void *temp;
    int64_t n, batch_size, num_batches;
    int32_t pointwise;
    bool built;
    int32_t kind;
    // kind 1 buffers (allocated by the first daisy_epoch_plan_build_indexed): record set [x] of the LSD passes
    void *parena;
    size_t parena_bytes, ptemp_bytes;
    uint4 *p_srec[2];                   // [n]   sample records {user, pos item, neg item / label, epoch position}
    uint2 *p_erec[2];                   // [2n]  entry records {item << 1 | slot, epoch position of the sample}
    uint32_t *p_counts, *p_offsets;     // [ndig * ntiles] per-tile digit counts / their exclusive scan
    uint32_t *p_inv;                    // [n]   inverse of an explicit permutation (DAISY_ORDER_PERM)
    uint32_t *p_park;                   // [2n]  device shuffle: the epoch positions the entry records' counting kernel walked
                                        //       to, parked for their scatter kernel
    void *ptemp;                        // rocPRIM scan scratch
    void *parena2;                      // second record set (plans with more than 256 batches: LSD passes ping-pong)
    int32_t p_cur;                      // record set holding the finished plan
    double hot_item_share;              // max_item_entries / n_ent of the index the plan was built from
    uint64_t build_gen;                 // id of the build the plan currently holds, unique in the process (what a
                                        // batch index refers to: a context that computed something ahead for "batch k+1"
                                        // must not mistake a rebuilt - or another plan at the same address - for it)
    // daisy_epoch_plan_build_positions: this plan holds a SUBSET of the epoch's rows (one rank's share), batch k =
    // the held rows whose epoch position lies in [k*B, (k+1)*B): record ranges differ per batch
    int64_t *h_off;                     // host, [num_batches+1] first record of every batch (NULL: k*batch_size)
    int64_t *d_off;                     // device scratch of the same
    int64_t h_off_cap;
};

constexpr int kMaxItemSlices = 16;  // daisy_bpr_staged_item_slices
constexpr int kEdgeBlock = 32;      // chunks per block of the two-level edge chains
constexpr int kPreBlocks = 256;   // workgroups (= partial sums) of the staged step's pre-norm pass
constexpr int64_t kMergeMaxBatch = 131072;   // largest batch of the three-launch staged step (see staged_sgd_step)

struct daisy_bpr_ctx {
    int64_t max_batch, U, I;
    int d;
    void *arena;
    size_t arena_bytes;
    float2 *coef;        // (dL/dpos, dL/dneg) per sample   [max_batch]
    double *partials;    // per-workgroup sums              [kMaxGrid*8], then [kPreBlocks] of the staged step's
                         // pre-norm pass (k_unorm), which the user pass reads while it writes the first part
    int32_t *tmp_triples;  // [max_batch*3] staging for daisy_bpr_set_batch
    float *edge_vec;     // [2*nchunks][d] partial user gradients of runs that cross a chunk boundary
    int32_t *edge_user;  // [2*nchunks]    their user (-1: none); [2c] head edge, [2c+1] tail edge
    float *edge_n;       // [2*nchunks][2] their sample counts and (FM) coefficient sums
    int32_t *edge_whole; // [nchunks]      the head edge's run also fills the whole chunk
    float *p_stage;      // [max_batch][d] updated user rows of the fused step, committed after the item pass
    float *p_sqnorm;     // [U] cache of |P[u]|^2 (fused step: the user-side Frobenius norm before the pass)
    const float *p_sqnorm_of;   // table the cache describes (NULL = invalid)
    daisy_epoch_plan *own_plan;   // 1-batch plan used by set_batch / set_batch_from_triples
    daisy::BatchView v;
    daisy::StreamView sv;  // the same batch as the staged step sees it
    int32_t batch_kind;   // layout of the plan the current batch comes from (0: v and sv valid, 1: only sv)
    int64_t edge_chunks;  // chunks the edge-record arrays hold (two records per chunk)
    // a second, small set of edge records: the three-launch form of the staged step (batches up to kMergeMaxBatch
    // samples) runs the user pass's edge chains and the item pass in ONE launch, so the two cannot share theirs
    float *edge2_vec; int32_t *edge2_item; float *edge2_cnt; int32_t *edge2_whole;
    int64_t edge2_chunks;
    float *edge_cnt;      // staged step, edge records of the item pass: [2*nchunks][2] (n_pos, n_neg)
    // block sums of the item pass's edge chains (k_staged_item_edge_blocks): one record per kEdgeBlock chunks
    float *eb_vec; int32_t *eb_item; float *eb_cnt; int32_t *eb_through; int64_t eb_blocks;
    int64_t *slice_rng;   // staged step in item slices (multi-GPU pipelining): entry range of slice s = [slice_rng[s], slice_rng[s+1])
    int32_t n_slices;     //   of the current batch (0: not prepared)
    int32_t pointwise;   // batches set through set_batch* hold (user, item, label) rows
    int last_item_mode;  // mode of the last daisy_bpr_item_grad: the user pass of the same step follows it
    float *bu, *bi, *b0, *g_bu, *g_bi, *g_b0;   // FM bias parameters (daisy_bpr_ctx_set_bias); bu == nullptr: MF
    bool batch_set, fwd_done;
    // the plan batch the context is set to (partitioned layout), and - staged step, single GPU - the batch whose
    // pre-norm sum(|P[u]|^2) the previous step's item pass has already produced on the side (see staged_sgd_step)
    const daisy_epoch_plan *cur_plan; int64_t cur_k; uint64_t cur_gen;
    const daisy_epoch_plan *pre_plan; int64_t pre_k; uint64_t pre_gen;
    const float *pre_P; const double *pre_stats; int pre_n;     // pre_n > 0: partial sums in `partials`; 0: stats[SQ_U_PRE]
    bool pre_ready;
    int32_t p_stream_mode;      // daisy_bpr_ctx_set_p_stream: -1 automatic (by table size), 0 off, 1 on
};


namespace daisy {
uint64_t next_plan_build_id();          // capi.hip
// bpr_staged.hip
bool staged_supported(const daisy_bpr_ctx *ctx, int loss_type);
int staged_sgd_step(daisy_bpr_ctx *ctx, float *P, float *Q, int loss_type, float gamma, float lr, float reg_1,
                    float reg_2, double *stats, double *epoch_acc, double *step_loss, hipStream_t s);
StreamView plan_stream_view(const daisy_epoch_plan *plan, int64_t k);
int plan_read_batch_partitioned(const daisy_epoch_plan *plan, int64_t k, int32_t *u, int32_t *i, int32_t *j,
                                int32_t *ent_item, uint32_t *ent_s, int32_t *ent_u, int64_t *B_out_host,
                                hipStream_t s);
// bpr_small.hip: every step of an epoch inside one persistent workgroup (batches of a few hundred samples)
constexpr int kSmallBatchMax = 256;
bool small_epoch_supported(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int loss_type);
bool small_epoch_adam_pays(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan);
// adam != NULL: torch.optim.Adam in the exact lazy form (moments, last-step stamps, the table of per-step constants) from
// step `first_step` on; the caller flushes the rows no batch referenced (daisy_adam_lazy_flush)
struct SmallAdamArgs {
    float *mP, *vP; int32_t *lastP;
    float *mQ, *vQ; int32_t *lastQ;
    const float *table;
    float beta1, beta2, eps;
    int64_t first_step;
};
int small_fit_epoch(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q, int loss_type, float gamma,
                    float lr, float reg_1, float reg_2, double *stats, double *epoch_acc, double *step_losses,
                    hipStream_t s, const SmallAdamArgs *adam = nullptr);
// bpr_train.hip: fixed-order reduction of `nblocks` x 8 per-workgroup sums into stats[0..6,12]; finalize: also
// the norms and the loss (finalize_stats)
int launch_reduce_partials(const double *partials, int nblocks, double *stats, bool finalize, float reg_1,
                           float reg_2, double *epoch_acc, double *step_loss, hipStream_t s);
// shared device helpers
// MFRecommender.py:88-89,94-95: loss += reg_1*(L1 terms) + reg_2*(Frobenius terms)
// the step's loss and the three Frobenius norms from the seven batch sums (indexed like stats[0..6])
__device__ __forceinline__ double loss_from_sums(const double *__restrict__ s7, float reg_1, float reg_2, double &nU,
                                                 double &nI, double &nJ) {
    nU = sqrt(s7[DAISY_ST_SQ_U]);
    nI = sqrt(s7[DAISY_ST_SQ_I]);
    nJ = sqrt(s7[DAISY_ST_SQ_J]);
    return s7[DAISY_ST_LOSS_DATA] + (double)reg_1 * (s7[DAISY_ST_L1_I] + s7[DAISY_ST_L1_J]) + (double)reg_2 * (nI + nJ) +
           (double)reg_1 * s7[DAISY_ST_L1_U] + (double)reg_2 * nU;
}
__device__ __forceinline__ void finalize_stats(double *__restrict__ stats, float reg_1, float reg_2,
                                               double *__restrict__ epoch_acc,
                                               double *__restrict__ step_loss) {
    double nU, nI, nJ;
    const double loss = loss_from_sums(stats, reg_1, reg_2, nU, nI, nJ);
    stats[DAISY_ST_LOSS] = loss;
    stats[DAISY_ST_NORM_U] = nU;
    stats[DAISY_ST_NORM_I] = nI;
    stats[DAISY_ST_NORM_J] = nJ;
    if (epoch_acc) {
        epoch_acc[0] += loss;
        if (!(loss == loss) || isinf(loss)) epoch_acc[1] += 1.0;
    }
    if (step_loss) *step_loss = loss;
}


// one workgroup of kBlock threads: the column sums of partials[nblocks][8] in a fixed order - thread b adds rows b,
// b + kBlock, ..., every wave adds its 64 threads' sums (DPP steps in registers), the four wave sums are added in wave
// order - returned in LDS: (*sums)[k] for k < 8, readable by every thread after the call (two barriers: the wave sums,
// then their total; round 3's eight-level LDS tree cost eight, on the critical path of every step's reduction).
// Contract: called by ALL threads of a workgroup of exactly kBlock threads (the row loop strides by kBlock, sm holds
// kBlock / kWave wave sums); the returned pointer aims at a function-static LDS array, so it is valid only until the
// workgroup's next call (callers that need the sums longer copy them).
__device__ __forceinline__ const double (*partials_sums(const double *__restrict__ partials, int nblocks))[8] {
    __shared__ double sm[kBlock / kWave + 1][8];
    if (blockDim.x != kBlock) __builtin_trap();      // (uniform, one compare: a violated contract must not pass silently)
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = threadIdx.x;
    for (; b + kBlock < nblocks; b += 2 * kBlock) {          // two rows (16 loads) in flight per thread
        double a0[8], a1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0[k] = partials[(int64_t)b * 8 + k]; a1[k] = partials[(int64_t)(b + kBlock) * 8 + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (t[k] + a0[k]) + a1[k];
    }
    for (; b < nblocks; b += kBlock) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] += partials[(int64_t)b * 8 + k];
    }
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double w = wave_sum_f64_dpp(t[k]);
        if ((threadIdx.x % kWave) == 0) sm[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double tot = sm[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) tot += sm[w][threadIdx.x];
        sm[kBlock / kWave][threadIdx.x] = tot;
    }
    __syncthreads();
    return reinterpret_cast<const double (*)[8]>(&sm[kBlock / kWave]);
}

// stats[0..6, SUM_COEF] = those sums; finalize: also the norms and the loss (finalize_stats)
__device__ __forceinline__ void reduce_partials_block(const double *__restrict__ partials, int nblocks,
                                                      double *__restrict__ stats, bool finalize, float reg_1,
                                                      float reg_2, double *__restrict__ epoch_acc,
                                                      double *__restrict__ step_loss) {
    const double (*sums)[8] = partials_sums(partials, nblocks);
    if (threadIdx.x < 7) stats[threadIdx.x] = (*sums)[threadIdx.x];
    if (threadIdx.x == 7) stats[DAISY_ST_SUM_COEF] = (*sums)[7];
    if (finalize) {
        __syncthreads();
        if (threadIdx.x == 0) finalize_stats(stats, reg_1, reg_2, epoch_acc, step_loss);
    }
}


// torch.optim.Adam single-tensor math on one row fragment: the expressions of k_adam_dense (bpr_train.hip), shared by the
// lazy row updates there and by the row owners of the staged step
template <class C>
__device__ __forceinline__ void adam_row(Row<C> &w, Row<C> &m, Row<C> &v, const Row<C> &g, float step_size, float bc2_sqrt,
                                         float beta1, float beta2, float eps) {
    const float w1 = 1.f - beta1, w2 = 1.f - beta2;
#pragma unroll
    for (int k = 0; k < C::NE; ++k) {
        const float gg = g.v[k];
        const float mm = fmaf(w1, gg - m.v[k], m.v[k]);
        const float vv = fmaf(w2 * gg, gg, beta2 * v.v[k]);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        w.v[k] = w.v[k] - step_size * (mm / denom);
        m.v[k] = mm;
        v.v[k] = vv;
    }
}

// Optimiser of a table's row owners in the staged step (bpr_staged.hip).  SGD: w -= lr * g.  Adam (template flag): the
// lazy form of torch.optim.Adam - the rows a step references have been brought to step t-1 before the pass
// (daisy_bpr_staged_adam_catchup), the owner applies step t with the row's gradient and stamps last[row] = t.
struct RowOpt {
    float lr;
    float *m, *v;              // [rows][d] first / second moments
    int32_t *last;             // [rows] the step each row is current for
    float step_size, bc2_sqrt, beta1, beta2, eps;     // step t's lr / (1 - beta1^t) and sqrt(1 - beta2^t): the host's bits
    int32_t t;
};

// FM (FMRecommender.py:61-68): score += u_bias[u] + i_bias[item] + bias_.  bu == nullptr: plain MF.  grad_out: the
// bias gradients go to g_bu / g_bi (and stats[DAISY_ST_SUM_COEF] for bias_) for a dense optimiser; else SGD in place.
struct StagedBias { float *bu, *bi, *b0; float *g_bu, *g_bi; int32_t grad_out; };

// m_pre / v_pre (Adam): the row's moments, loaded by the caller together with the row itself - a commit then issues no
// load at all.  (Loaded here, they sit behind every store the wave has issued: the vector-memory counter counts stores,
// and s_waitcnt vmcnt(0) in front of the first use waits for all of them - profiles/r05_item_pass_counters.txt.)
template <class C, bool ADAM>
__device__ __forceinline__ void row_apply(Row<C> &w, const Row<C> &g, const RowOpt &o, int64_t row, int lane, int d,
                                          const Row<C> *m_pre = nullptr, const Row<C> *v_pre = nullptr) {
    if constexpr (ADAM) {
        Row<C> m, v;
        if (m_pre) { m = *m_pre; v = *v_pre; }             // (compile-time at every call site)
        else {
            m.load(o.m + row * d, lane, d);
            v.load(o.v + row * d, lane, d);
        }
        adam_row<C>(w, m, v, g, o.step_size, o.bc2_sqrt, o.beta1, o.beta2, o.eps);
        m.store(o.m + row * d, lane, d);
        v.store(o.v + row * d, lane, d);
        if (lane == 0) o.last[row] = o.t;
    } else {
#pragma unroll
        for (int k = 0; k < C::NE; ++k) w.v[k] = fmaf(-o.lr, g.v[k], w.v[k]);
    }
}

__device__ __forceinline__ float inv_or_zero(double n, float reg_2) {
    return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f;  // d|X|_F/dX = 0 at X = 0 (torch)
}
}  // namespace daisy
