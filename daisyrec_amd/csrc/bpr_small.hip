// Small-batch epochs (the reference's default batch_size = 256, basic.yaml:23; BASELINE configs[0]):
// one persistent workgroup runs EVERY step of an epoch.
//
// At B = 256 a step moves ~100 KB; the phase kernels of bpr_train.hip spend 23-29 us per step on it, all of
// it the dependency chain of five to seven kernel launches (measured: a captured hipGraph of the chain was
// not faster - the cost is kernel-boundary latency on the GPU, not host work).  Here the chain's phases are
// separated by workgroup barriers instead of kernel boundaries, inside ONE kernel per epoch, and the rows a
// step gathers are staged in LDS once (160 KB per CU) so that every later phase reads them from there:
//
//   A  forward: 3 row gathers per sample -> registers -> LDS; both scores, loss term, c = dL/dx -> LDS;
//      the seven batch sums reduced across the workgroup in fixed order              (MFRecommender.py:63-97)
//   B  item side: the lane group that sees the head of an item's run of entries owns the row, sums
//      c_e * p_u(e) over the run in plan order from the STAGED user rows, adds the regulariser (the
//      pre-step item row is the staged q of its own sample) and writes Q[i] in place;
//      user side, same phase: the group at the head of a user's run sums over the STAGED item rows and writes
//      P[u] in place - nobody reads the tables during this phase, only the pre-step copies in LDS.
//      (d too wide for three staged rows per sample: the user rows stay in global memory, and the user side
//      waits behind one more barrier for the item side to finish reading them.  Wider still - the reference's
//      default d = 100 at B = 256 - only the USER rows are staged: the user side runs first and reads its item
//      rows from the table, which nobody has written yet; after a barrier the item side reads the staged user
//      rows and the pre-step row of its own item from the table.)
//
// Global traffic per step = the 3B gathered rows + one write per distinct row; the index arrays of step k+1
// are fetched while step k computes.  Every sum runs in plan order on a single owner: bitwise reproducible.
// One workgroup = one CU: right for batches of a few hundred samples, hopeless for large ones (the dispatcher
// in bpr_train.hip only comes here for B <= kSmallBatchMax and rows that fit the LDS).
#include <stdlib.h>

#include "bpr_internal.h"

namespace daisy {

constexpr int kSmallThreads = 1024;
constexpr size_t kSmallLdsRows = 128 * 1024;      // dynamic LDS for the staged rows (static arrays take ~22 KB of the 160)

struct SmallPlan {           // the sorted layout of daisy_epoch_plan (kind 0), whole epoch
    const uint32_t *ukey;
    const int2 *ij;
    const uint32_t *ekey;
    const uint2 *esu;
    uint32_t umask, imask;
    int64_t n, B, nb;
};

// metadata of one step in LDS (filled one step ahead: no global index load sits on a step's critical path)
struct SmallMeta {
    uint32_t ukey[kSmallBatchMax];
    int2 ij[kSmallBatchMax];
    uint32_t item[2 * kSmallBatchMax];       // entry -> item
    uint2 su[2 * kSmallBatchMax];            // entry -> (sample | neg bit, user)
};

// row <-> LDS (the staged rows are laid out like table rows: d floats, chunk c of lane l at (c*LPR + l)*VEC)
template <class C>
__device__ __forceinline__ void lds_store_row(float *dst, const Row<C> &r, int lane, int d) {
#pragma unroll
    for (int c = 0; c < C::NV; ++c) {
        const int e = (c * C::LPR + lane) * C::VEC;
        if (C::EXACT || e < d) {
#pragma unroll
            for (int k = 0; k < C::VEC; ++k) dst[e + k] = r.v[c * C::VEC + k];
        }
    }
}
template <class C>
__device__ __forceinline__ void lds_load_row(Row<C> &r, const float *src, int lane, int d) {
#pragma unroll
    for (int c = 0; c < C::NV; ++c) {
        const int e = (c * C::LPR + lane) * C::VEC;
        const bool in = C::EXACT || e < d;                 // lanes past d read the row's start and drop the value:
        const int ec = in ? e : 0;                         // no LDS read behind a branch
#pragma unroll
        for (int k = 0; k < C::VEC; ++k) {
            const float t = src[ec + k];
            r.v[c * C::VEC + k] = in ? t : 0.f;
        }
    }
}

// torch.optim.Adam in the persistent epoch (round 6): the exact lazy form of bpr_train.hip (adam_claim_row) / the staged
// step - last[row] = the step a row is current for, table[s] = step s's (lr / (1 - beta1^s), sqrt(1 - beta2^s)) - with the
// phases of a step separated by workgroup barriers instead of six or seven kernel boundaries (48 us per step at B = 256):
//   0  every distinct row of the step (heads of the user runs / of the item runs) is brought to step t-1: the zero-gradient
//      replay of the steps it sat out, in registers, written back with last = t-1
//   A  as SGD: the caught-up rows are gathered and staged                           B  the owner of a row forms its gradient
//      as SGD does and applies step t with the row's moments (read and written once per distinct row), last = t
struct SmallAdam {
    float *mP, *vP; int32_t *lastP;
    float *mQ, *vQ; int32_t *lastQ;
    const float2 *table;
    float beta1, beta2, eps;
    int64_t first_step;
    // helpers > 0: workgroups 1 .. helpers run the catch-up of step j's rows AHEAD of the main workgroup (below);
    // sync[0] = steps the main workgroup has finished, sync[1 + j] = helpers that have finished step j's rows
    int helpers;
    int *sync;
};

// ---- the catch-up of the lazy Adam rows on helper workgroups (round 6) ----------------------------------------------
// Bringing a row from the step it was last touched to step t-1 replays every step in between (sqrt, two IEEE divides per
// element and step: ~40 VALU instructions); with the whole epoch inside ONE workgroup that replay alone kept the CU's four
// SIMDs busy for tens of microseconds per step (measured: 82 us per step at ml-100k shapes against 47 for the chain of
// launches it was meant to replace).  But which rows step j needs is known from the plan, and a row that step j-1 does
// NOT reference is not touched by the main workgroup while it runs step j-1: helper workgroups bring the rows of
// set(j) \ set(j-1) to step t_j - 1 during step j-1 (after the main workgroup has finished step j-2, whose owners may have
// written them); the rows of set(j) that step j-1 references leave that step current.  Hand-offs: one release / acquire
// pair per step and direction (MI355X_MICROARCH.md, "inter-workgroup visibility").
__device__ __forceinline__ void small_wait_ge(const int *flag, int want) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void small_publish_add(int *flag) {      // one lane, after the workgroup's barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// is `id` among the (ascending) masked keys a[0 .. n)?  shift: 0 for users, 1 for the entry keys (item << 1 | slot)
__device__ __forceinline__ bool small_has(const uint32_t *a, int n, uint32_t mask, int shift, uint32_t id) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (((a[mid] & mask) >> shift) < id) lo = mid + 1; else hi = mid;
    }
    return lo < n && ((a[lo] & mask) >> shift) == id;
}

template <class C>
__device__ void small_adam_helper(float *__restrict__ P, float *__restrict__ Q, const SmallPlan &pl, int d, const SmallAdam &ad,
                                  uint32_t *__restrict__ prev_u, uint32_t *__restrict__ prev_e) {
    constexpr int G = kSmallThreads / C::LPR;
    const int tid = threadIdx.x, lane = tid % C::LPR, group = tid / C::LPR;
    const int h = (int)blockIdx.x - 1, H = ad.helpers;
    for (int64_t j = 0; j < pl.nb; ++j) {
        if (j >= 2) {
            if (tid == 0) { small_wait_ge(ad.sync, (int)(j - 1)); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
            __syncthreads();
        }
        const int32_t t_step = (int32_t)(ad.first_step + j);
        const int64_t lo = j * pl.B;
        const int Bj = (int)((pl.n - lo < pl.B) ? (pl.n - lo) : pl.B);
        const int64_t lop = lo - pl.B;                                  // step j-1 (full batch)
        if (j >= 1) {
            // step j-1's (sorted) user and entry keys -> LDS: the membership tests below are eight-deep binary searches, and
            // as dependent GLOBAL loads they were most of a helper's time per step
            for (int x = tid; x < (int)pl.B; x += kSmallThreads) prev_u[x] = pl.ukey[lop + x] & pl.umask;
            for (int x = tid; x < (int)(2 * pl.B); x += kSmallThreads) prev_e[x] = (pl.ekey[2 * lop + x] & pl.imask) >> 1;
            __syncthreads();
        }
        for (int x = h * G + group; x < 3 * Bj; x += H * G) {
            const bool us = x < Bj;
            uint32_t row;
            if (us) {
                row = pl.ukey[lo + x] & pl.umask;
                if (x > 0 && (pl.ukey[lo + x - 1] & pl.umask) == row) continue;
                if (j >= 1 && small_has(prev_u, (int)pl.B, 0xFFFFFFFFu, 0, row)) continue;
            } else {
                const int e = x - Bj;
                row = (pl.ekey[2 * lo + e] & pl.imask) >> 1;
                if (e > 0 && ((pl.ekey[2 * lo + e - 1] & pl.imask) >> 1) == row) continue;
                if (j >= 1 && small_has(prev_e, (int)(2 * pl.B), 0xFFFFFFFFu, 0, row)) continue;
            }
            float *W = us ? P : Q, *M = us ? ad.mP : ad.mQ, *V = us ? ad.vP : ad.vQ;
            int32_t *last = us ? ad.lastP : ad.lastQ;
            const int32_t old = last[row];
            if (old >= t_step - 1) continue;
            Row<C> w, mm, vv, g0;
            w.load_clamped(W + (int64_t)row * d, lane, d); mm.load_clamped(M + (int64_t)row * d, lane, d);
            vv.load_clamped(V + (int64_t)row * d, lane, d);
            g0.zero();
            for (int32_t st = old + 1; st < t_step; ++st) {
                const float2 c = ad.table[st];
                adam_row<C>(w, mm, vv, g0, c.x, c.y, ad.beta1, ad.beta2, ad.eps);
            }
            w.store(W + (int64_t)row * d, lane, d); mm.store(M + (int64_t)row * d, lane, d); vv.store(V + (int64_t)row * d, lane, d);
            if (lane == 0) last[row] = t_step - 1;
        }
        __syncthreads();
        if (tid == 0) small_publish_add(ad.sync + 1 + j);
    }
}

template <class C, int MODE, bool ADAM = false>      // rows staged in LDS per sample: 3 = q_i, q_j, p_u; 2 = q_i, q_j; 1 = p_u only
__global__ __launch_bounds__(kSmallThreads) void k_small_epoch(
    float *__restrict__ P, float *__restrict__ Q, SmallPlan pl, int d, int dpad, int loss_type, float gamma, float lr,
    float reg_1, float reg_2, double *__restrict__ stats, double *__restrict__ epoch_acc,
    double *__restrict__ step_losses, SmallAdam ad = SmallAdam{}) {
    constexpr int G = kSmallThreads / C::LPR;
    constexpr int NW = kSmallThreads / kWave;
    constexpr int UN = (C::NE <= 4) ? 2 : 1;            // samples whose 3 row gathers are issued together per lane group
    constexpr int PT = (3 * kSmallBatchMax + kSmallThreads - 1) / kSmallThreads;   // metadata words per thread
    constexpr bool STAGE_P = MODE != 2, STAGE_Q = MODE != 1;
    extern __shared__ float s_rows[];                  // MODE 3/2: [B][dpad] q_i | [B][dpad] q_j | (3: [B][dpad] p_u);  MODE 1: [B][dpad] p_u
    __shared__ SmallMeta s_meta[2];
    __shared__ float2 s_coef[kSmallBatchMax];          // (dL/dpos, dL/dneg)
    __shared__ double s_part[NW][8];
    __shared__ double s_stats[DAISY_STATS_LEN];
    __shared__ float s_inv[3];                         // reg_2 / |P[u]|_F, / |Q[i]|_F, / |Q[j]|_F of the step

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const int wave = tid / kWave;
    float *const s_qi = s_rows;
    float *const s_qj = s_rows + (size_t)pl.B * dpad;
    float *const s_p = STAGE_Q ? s_rows + 2 * (size_t)pl.B * dpad : s_rows;
    double acc_epoch = 0.0, nan_epoch = 0.0;           // thread 0 only
    if constexpr (ADAM) {
        if (blockIdx.x > 0) {                          // helper workgroups: the catch-up of the steps ahead (see above)
            small_adam_helper<C>(P, Q, pl, d, ad, s_meta[1].ukey, s_meta[1].item);
            return;
        }
    }

    // element x of a step's metadata: x < B samples, then 2B entries
    uint32_t pre_a[PT];
    uint2 pre_b[PT];
    auto fetch = [&](int64_t k) {                      // global -> registers (issued a whole step ahead)
        const int64_t lo = k * pl.B;
        const int Bk = (int)((pl.n - lo < pl.B) ? (pl.n - lo) : pl.B);
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int x = t * kSmallThreads + tid;
            if (x < Bk) {
                pre_a[t] = pl.ukey[lo + x] & pl.umask;
                const int2 v = pl.ij[lo + x];
                pre_b[t] = make_uint2((uint32_t)v.x, (uint32_t)v.y);
            } else if (x < 3 * Bk) {
                pre_a[t] = (pl.ekey[2 * lo + (x - Bk)] & pl.imask) >> 1;
                pre_b[t] = pl.esu[2 * lo + (x - Bk)];
            }
        }
    };
    auto stash = [&](int64_t k) {                      // registers -> LDS buffer k & 1
        const int64_t lo = k * pl.B;
        const int Bk = (int)((pl.n - lo < pl.B) ? (pl.n - lo) : pl.B);
        SmallMeta &m = s_meta[k & 1];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int x = t * kSmallThreads + tid;
            if (x < Bk) { m.ukey[x] = pre_a[t]; m.ij[x] = make_int2((int)pre_b[t].x, (int)pre_b[t].y); }
            else if (x < 3 * Bk) { m.item[x - Bk] = pre_a[t]; m.su[x - Bk] = pre_b[t]; }
        }
    };
    if (pl.nb > 0) { fetch(0); stash(0); }
    __syncthreads();

    for (int64_t k = 0; k < pl.nb; ++k) {
        const int64_t lo = k * pl.B;
        const int Bk = (int)((pl.n - lo < pl.B) ? (pl.n - lo) : pl.B);
        const SmallMeta &m = s_meta[k & 1];
        if (k + 1 < pl.nb) fetch(k + 1);               // lands while this step computes

        [[maybe_unused]] const int32_t t_step = ADAM ? (int32_t)(ad.first_step + k) : 0;
        if constexpr (ADAM) {
            if (ad.helpers > 0) {
                // ---- 0: the helpers have brought the step's rows to step t-1 (or step k-1 left them there)
#ifndef DAISY_SMALL_NOWAIT          // (timing experiments only: the main workgroup's own step time, results meaningless)
                if (tid == 0) { small_wait_ge(ad.sync + 1 + k, ad.helpers); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#endif
                __syncthreads();
            } else
            // ---- 0: the step's distinct rows -> step t-1 (x < Bk: heads of the user runs; then heads of the item runs)
            for (int x = group; x < 3 * Bk; x += G) {
                const bool us = x < Bk;
                int64_t row;
                if (us) {
                    if (x > 0 && m.ukey[x - 1] == m.ukey[x]) continue;
                    row = m.ukey[x];
                } else {
                    const int e = x - Bk;
                    if (e > 0 && m.item[e - 1] == m.item[e]) continue;
                    row = m.item[e];
                }
                float *W = us ? P : Q, *M = us ? ad.mP : ad.mQ, *V = us ? ad.vP : ad.vQ;
                int32_t *last = us ? ad.lastP : ad.lastQ;
                const int32_t old = last[row];
                if (old >= t_step - 1) continue;
                Row<C> w, mm, vv, g0;
                w.load_clamped(W + row * d, lane, d); mm.load_clamped(M + row * d, lane, d); vv.load_clamped(V + row * d, lane, d);
                g0.zero();
                for (int32_t st = old + 1; st < t_step; ++st) {
                    const float2 c = ad.table[st];
                    adam_row<C>(w, mm, vv, g0, c.x, c.y, ad.beta1, ad.beta2, ad.eps);
                }
                w.store(W + row * d, lane, d); mm.store(M + row * d, lane, d); vv.store(V + row * d, lane, d);
                if (lane == 0) last[row] = t_step - 1;
            }
            __syncthreads();
        }

        // ---- A: forward; the gathered rows stay in LDS for phase B
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s0 = group; s0 < Bk; s0 += UN * G) {
            Row<C> p[UN], qi[UN], qj[UN];
#pragma unroll
            for (int y = 0; y < UN; ++y) {
                const int s = s0 + y * G;
                if (s < Bk) {
                    const int2 it = m.ij[s];
                    p[y].load_clamped(P + (int64_t)m.ukey[s] * d, lane, d);
                    qi[y].load_clamped(Q + (int64_t)it.x * d, lane, d);
                    qj[y].load_clamped(Q + (int64_t)it.y * d, lane, d);
                }
            }
#pragma unroll
            for (int y = 0; y < UN; ++y) {
                const int s = s0 + y * G;
                if (s < Bk) {
                    if constexpr (STAGE_Q) {
                        lds_store_row<C>(s_qi + (size_t)s * dpad, qi[y], lane, d);
                        lds_store_row<C>(s_qj + (size_t)s * dpad, qj[y], lane, d);
                    }
                    if constexpr (STAGE_P) lds_store_row<C>(s_p + (size_t)s * dpad, p[y], lane, d);
                    const float pos = row_dot<C>(p[y], qi[y]);
                    const float neg = row_dot<C>(p[y], qj[y]);
#pragma unroll
                    for (int e = 0; e < C::NE; ++e) {
                        acc[1] += fabsf(p[y].v[e]);
                        acc[2] += fabsf(qi[y].v[e]);
                        acc[3] += fabsf(qj[y].v[e]);
                        acc[4] = fmaf(p[y].v[e], p[y].v[e], acc[4]);
                        acc[5] = fmaf(qi[y].v[e], qi[y].v[e], acc[5]);
                        acc[6] = fmaf(qj[y].v[e], qj[y].v[e], acc[6]);
                    }
                    if (lane == 0) {
                        float term, cp, cn;
                        pair_coef(loss_type, pos, neg, gamma, term, cp, cn);
                        s_coef[s] = make_float2(cp, cn);
                        acc[0] += term;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double w = (double)wave_sum_f32_dpp(acc[q]);     // fp32 inside the wave (64 terms), fp64 across the waves
            if ((tid % kWave) == 0) s_part[wave][q] = w;
        }
        __syncthreads();
        if (tid < 7) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += s_part[w][tid];
            s_stats[tid] = t;
            // reg_2 / |X|_F once, by the thread that holds the sum (an fp64 sqrt and divide per THREAD was a
            // measurable part of the step)
            if (tid >= DAISY_ST_SQ_U) {
                const double nrm = sqrt(t);
                s_stats[DAISY_ST_NORM_U + (tid - DAISY_ST_SQ_U)] = nrm;
                s_inv[tid - DAISY_ST_SQ_U] = inv_or_zero(nrm, reg_2);
            }
        }
        __syncthreads();
        const float rU = s_inv[0], rI = s_inv[1], rJ = s_inv[2];
        // AbstractRecommender.py:122-123: a non-finite loss ends the training BEFORE that step's backward.  The loss is
        // finite iff its seven sums are; every thread reads the same seven words, so the exit is uniform.
        bool finite = true;
#pragma unroll
        for (int q = 0; q <= DAISY_ST_SQ_J; ++q) { const double x = s_stats[q]; finite = finite && (x == x) && !isinf(x); }
        if (tid == 0) {       // MFRecommender.py:88-89,94-95 (slots 7..10 are written here and read by nobody until the end)
            const double nU = s_stats[DAISY_ST_NORM_U], nI = s_stats[DAISY_ST_NORM_I], nJ = s_stats[DAISY_ST_NORM_J];
            const double loss = s_stats[DAISY_ST_LOSS_DATA] +
                                (double)reg_1 * (s_stats[DAISY_ST_L1_I] + s_stats[DAISY_ST_L1_J]) +
                                (double)reg_2 * (nI + nJ) + (double)reg_1 * s_stats[DAISY_ST_L1_U] + (double)reg_2 * nU;
            s_stats[DAISY_ST_LOSS] = loss;
            acc_epoch += loss;
            if (!(loss == loss) || isinf(loss)) nan_epoch += 1.0;
            if (step_losses) step_losses[k] = loss;
        }
        if (!finite) {         // the tables stay as step k-1 left them; the host raises when it reads epoch_acc[1]
            if constexpr (ADAM) {
                if (ad.helpers > 0 && tid == 0)      // (release the helpers: they run out their steps on rows nobody reads any more)
                    __hip_atomic_store(ad.sync, (int)pl.nb + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }

        // ---- B, item side: entries sorted by item; the head of a run owns Q[item]
        const int nE = 2 * Bk;
        auto item_side = [&]() {
            for (int e = group; e < nE; e += G) {
                const uint32_t r = m.item[e];
                if (e > 0 && m.item[e - 1] == r) continue;
                const uint32_t s_head = m.su[e].x;
                Row<C> a, qr;      // the pre-step Q[r]: the staged q of the head entry's own sample, or (MODE 1) the table row
                if constexpr (STAGE_Q)
                    lds_load_row<C>(qr, ((s_head & kNegBit) ? s_qj : s_qi) + (size_t)(s_head & ~kNegBit) * dpad, lane, d);
                else qr.load_clamped(Q + (int64_t)r * d, lane, d);
                a.zero();
                float np = 0.f, nn = 0.f;
                for (int q = e; q < nE && m.item[q] == r; ++q) {
                    const uint2 su = m.su[q];
                    const bool is_neg = (su.x & kNegBit) != 0;
                    const float2 c2 = s_coef[su.x & ~kNegBit];
                    const float c = is_neg ? c2.y : c2.x;
                    np += is_neg ? 0.f : 1.f;
                    nn += is_neg ? 1.f : 0.f;
                    Row<C> pr;
                    if constexpr (STAGE_P) lds_load_row<C>(pr, s_p + (size_t)(su.x & ~kNegBit) * dpad, lane, d);
                    else pr.load(P + (int64_t)su.y * d, lane, d);
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) a.v[x] = fmaf(c, pr.v[x], a.v[x]);
                }
                const float w1 = reg_1 * (np + nn), w2 = np * rI + nn * rJ;
                if constexpr (ADAM) {
                    Row<C> mm, vv;
                    mm.load_clamped(ad.mQ + (int64_t)r * d, lane, d); vv.load_clamped(ad.vQ + (int64_t)r * d, lane, d);
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) a.v[x] = a.v[x] + fmaf(w2, qr.v[x], w1 * sgn(qr.v[x]));
                    const float2 c = ad.table[t_step];
                    adam_row<C>(qr, mm, vv, a, c.x, c.y, ad.beta1, ad.beta2, ad.eps);
                    mm.store(ad.mQ + (int64_t)r * d, lane, d); vv.store(ad.vQ + (int64_t)r * d, lane, d);
                    if (lane == 0) ad.lastQ[r] = t_step;
                } else {
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) {
                        const float g = a.v[x] + fmaf(w2, qr.v[x], w1 * sgn(qr.v[x]));
                        qr.v[x] = fmaf(-lr, g, qr.v[x]);
                    }
                }
                qr.store(Q + (int64_t)r * d, lane, d);      // nobody reads Q before the next step's phase A
            }
        };
        // ---- B, user side: samples grouped by user; the head of a run owns P[u]
        auto user_side = [&]() {
            for (int s = group; s < Bk; s += G) {
                const uint32_t uu = m.ukey[s];
                if (s > 0 && m.ukey[s - 1] == uu) continue;
                Row<C> p, a;
                if constexpr (STAGE_P) lds_load_row<C>(p, s_p + (size_t)s * dpad, lane, d);
                else p.load(P + (int64_t)uu * d, lane, d);
                a.zero();
                float cnt = 0.f;
                for (int q = s; q < Bk && m.ukey[q] == uu; ++q) {
                    const float2 c = s_coef[q];
                    Row<C> qi, qj;
                    if constexpr (STAGE_Q) {
                        lds_load_row<C>(qi, s_qi + (size_t)q * dpad, lane, d);
                        lds_load_row<C>(qj, s_qj + (size_t)q * dpad, lane, d);
                    } else {                                // MODE 1: the item side has not run yet, Q is pre-step
                        const int2 it = m.ij[q];
                        qi.load_clamped(Q + (int64_t)it.x * d, lane, d);
                        qj.load_clamped(Q + (int64_t)it.y * d, lane, d);
                    }
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) a.v[x] = fmaf(c.x, qi.v[x], fmaf(c.y, qj.v[x], a.v[x]));
                    cnt += 1.f;
                }
                const float w1 = reg_1 * cnt, w2 = rU * cnt;
                if constexpr (ADAM) {
                    Row<C> mm, vv;
                    mm.load_clamped(ad.mP + (int64_t)uu * d, lane, d); vv.load_clamped(ad.vP + (int64_t)uu * d, lane, d);
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) a.v[x] = a.v[x] + fmaf(w2, p.v[x], w1 * sgn(p.v[x]));
                    const float2 c = ad.table[t_step];
                    adam_row<C>(p, mm, vv, a, c.x, c.y, ad.beta1, ad.beta2, ad.eps);
                    mm.store(ad.mP + (int64_t)uu * d, lane, d); vv.store(ad.vP + (int64_t)uu * d, lane, d);
                    if (lane == 0) ad.lastP[uu] = t_step;
                } else {
#pragma unroll
                    for (int x = 0; x < C::NE; ++x) {
                        const float g = a.v[x] + fmaf(w2, p.v[x], w1 * sgn(p.v[x]));
                        p.v[x] = fmaf(-lr, g, p.v[x]);
                    }
                }
                p.store(P + (int64_t)uu * d, lane, d);
            }
        };
        if constexpr (MODE == 3) {            // both sides read only the staged pre-step rows
            item_side();
            user_side();
        } else if constexpr (MODE == 2) {     // the item side reads P from the table: the user side writes it afterwards
            item_side();
            __syncthreads();
            user_side();
        } else {                              // the user side reads Q from the table: the item side writes it afterwards
            user_side();
            __syncthreads();
            item_side();
        }
        if (k + 1 < pl.nb) stash(k + 1);
        __syncthreads();
        if constexpr (ADAM) {
            if (ad.helpers > 0 && tid == 0) small_publish_add(ad.sync);        // step k's rows are written: sync[0] = k + 1
        }
    }
    if (tid < DAISY_STATS_LEN && pl.nb > 0) stats[tid] = (tid <= DAISY_ST_NORM_J) ? s_stats[tid] : 0.0;   // the last step's
    if (tid == 0 && epoch_acc) { epoch_acc[0] += acc_epoch; epoch_acc[1] += nan_epoch; }
}

// floats per staged row: float4-aligned; rows that are a multiple of 128 bytes get one float4 of padding when three
// row sets still fit (every row would start on the same banks otherwise: measured 11.3 vs 7.x us per step at d = 32
// with four-lane groups)
static inline int small_dpad(int d, int64_t B = kSmallBatchMax) {
    const int p = (d + 3) / 4 * 4;
    if (p % 32 == 0 && 3 * (size_t)B * (p + 4) * sizeof(float) <= kSmallLdsRows) return p + 4;
    return p;
}

// the Adam form pays where the rows a step references were touched a few steps ago (the zero-gradient replay is ~40 VALU
// instructions per element and skipped step): tables of at most 32 rows per sample of a batch - ml-100k at B = 256: 8;
// measured 35 us per step against 48 for the chain of launches there, but 750 against 250 at BASELINE configs[1] tables
// (4 300 rows per sample: a row sits out thousands of steps)
bool small_epoch_adam_pays(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan) {
    return ctx->U + ctx->I <= 32 * plan->batch_size;
}

bool small_epoch_supported(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int loss_type) {
    static const int enabled = getenv("DAISY_SMALL_EPOCH") ? atoi(getenv("DAISY_SMALL_EPOCH")) : 1;
    return enabled && plan->kind == 0 && !plan->pointwise && plan->batch_size <= kSmallBatchMax && !ctx->bu &&
           loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_TL && plan->batch_size <= ctx->max_batch &&
           (size_t)plan->batch_size * small_dpad(ctx->d, plan->batch_size) * sizeof(float) <= kSmallLdsRows;      // at least the user rows
}

int small_fit_epoch(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q, int loss_type, float gamma,
                    float lr, float reg_1, float reg_2, double *stats, double *epoch_acc, double *step_losses,
                    hipStream_t s, const SmallAdamArgs *adam) {
    SmallPlan pl;
    pl.ukey = plan->ukey;
    pl.ij = reinterpret_cast<const int2 *>(plan->uval);
    pl.ekey = plan->ekey;
    pl.esu = reinterpret_cast<const uint2 *>(plan->eval);
    pl.umask = plan->umask; pl.imask = plan->imask;
    pl.n = plan->n; pl.B = plan->batch_size; pl.nb = plan->num_batches;
    const int d = ctx->d, dpad = small_dpad(d, plan->batch_size);
    const size_t row_bytes = (size_t)plan->batch_size * dpad * sizeof(float);
    const int mode = (3 * row_bytes <= kSmallLdsRows) ? 3 : ((2 * row_bytes <= kSmallLdsRows) ? 2 : 1);
    const size_t shmem = (size_t)mode * row_bytes;
    // Lane groups half as wide as dispatch_d's (a lane then holds two float4 of a row instead of one): twice as
    // many groups share the serial part of phase B - the run heads each group works through one after the other,
    // a chain of dependent LDS reads - which is what bounds a step here, not bytes.  Measured at B=256: d=32 8.9 ->
    // see profiles/r02_small_epoch.txt.
    static const int tune_narrow = getenv("DAISY_SMALL_NARROW") ? atoi(getenv("DAISY_SMALL_NARROW")) : 1;
    auto dispatch_small = [&](auto &&f) -> int {
        if (tune_narrow && d % 4 == 0 && d <= 128) {
            if (d == 32) return f(RowCfg<4, 4, 2, true>{});
            if (d == 64) return f(RowCfg<8, 4, 2, true>{});
            if (d == 128) return f(RowCfg<8, 4, 4, true>{});
            if (d <= 32) return f(RowCfg<4, 4, 2>{});
            if (d <= 64) return f(RowCfg<8, 4, 2>{});
            return f(RowCfg<8, 4, 4>{});
        }
        return dispatch_d(d, f);
    };
    int rc = dispatch_small([&](auto cfg) -> int {
        using C = decltype(cfg);
        SmallAdam ad{};
        int grid = 1;
        if (adam) {
            ad.mP = adam->mP; ad.vP = adam->vP; ad.lastP = adam->lastP; ad.mQ = adam->mQ; ad.vQ = adam->vQ; ad.lastQ = adam->lastQ;
            ad.table = reinterpret_cast<const float2 *>(adam->table);
            ad.beta1 = adam->beta1; ad.beta2 = adam->beta2; ad.eps = adam->eps; ad.first_step = adam->first_step;
            // DAISY_SMALL_ADAM_HELPERS (read per call): workgroups that run the catch-up ahead of the main one; 0: inside it
            const char *env = getenv("DAISY_SMALL_ADAM_HELPERS");
            int helpers = env ? atoi(env) : 8;
            const size_t sync_ints = (size_t)plan->num_batches + 1;
            if (helpers < 0 || sync_ints * sizeof(int) > ((size_t)kMaxGrid * 8 + kPreBlocks) * 8) helpers = 0;
            if (helpers > 64) helpers = 64;
            if (helpers > 0) {
                ad.helpers = helpers;
                ad.sync = reinterpret_cast<int *>(ctx->partials);       // (scratch of the phase kernels: idle during the epoch call)
                DAISY_HIP(hipMemsetAsync(ad.sync, 0, sync_ints * sizeof(int), s));
                grid = 1 + helpers;
            }
        }
        auto launch = [&](auto kern) -> int {
            DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kSmallLdsRows));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kSmallThreads), shmem, s, P, Q, pl, d, dpad, loss_type, gamma, lr,
                               reg_1, reg_2, stats, epoch_acc, step_losses, ad);
            return DAISY_OK;
        };
        if (adam) {
            if (mode == 3) return launch(k_small_epoch<C, 3, true>);
            if (mode == 2) return launch(k_small_epoch<C, 2, true>);
            return launch(k_small_epoch<C, 1, true>);
        }
        if (mode == 3) return launch(k_small_epoch<C, 3>);
        if (mode == 2) return launch(k_small_epoch<C, 2>);
        return launch(k_small_epoch<C, 1>);
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    ctx->p_sqnorm_of = nullptr;          // P rows changed behind the staged step's row-norm cache
    ctx->pre_ready = false;              // (and behind a pre-norm computed ahead of its step)
    ctx->batch_set = false;
    ctx->fwd_done = false;
    return DAISY_OK;
}

}  // namespace daisy
