// MF + BPR training step for gfx950 (MI355X): hand-written gather / dot /
// loss-coefficient / scatter-update kernels.
//
// Reference semantics being reproduced (file:line in AmazingDD/daisyRec):
//   loader       daisy/utils/dataset.py:5-27 (DataLoader(shuffle=True) over BasicDataset)
//   forward      daisy/model/MFRecommender.py:63-68
//   loss         daisy/model/MFRecommender.py:70-97 + daisy/utils/loss.py:5-33
//   backward     autograd through 7 embedding lookups (AbstractRecommender.py:125)
//   SGD / Adam   daisy/model/AbstractRecommender.py:48-67,126
//
// The step is BATCH SYNCHRONOUS like autograd + optimizer.step: every gradient
// is formed from the tables as they were when the step began.  The kernel order
// guarantees it without a copy of the tables:
//   k_fwd          reads P,Q            writes coef, partial sums
//   k_item_grad_*  reads P,Q            writes gQ  (side buffer; also what the
//                                       multi-GPU path all-reduces)
//   k_user         reads P[u] (owner),Q writes P[u]   (one owner per user row:
//                                       every batch is grouped by user)
//   k_item_apply   reads gQ,Q           writes Q, zeroes gQ
//
// Measured on MI355X (profiles/r01_probe_*): random 256-B row gathers / plain
// stores run at 5-8 TB/s, fp32 global atomics at 0.3 TB/s.  So every scatter is
// organised around OWNERSHIP instead of atomics: an EPOCH PLAN (radix sorts, once
// per epoch) lays the epoch out batch by batch, each batch grouped by user, plus
// a per-batch list of item entries sorted by item; the item gradient is then a
// segmented reduction over that list.
//
// HBM/L2 view: a d=64 row is 256 B = 16 lanes x float4, one coalesced request
// per quarter wave; four rows are in flight per wave instruction.
#include <stdlib.h>
#include <string.h>

#include "bpr_internal.h"

namespace daisy {

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------
// epoch plan kernels
// ---------------------------------------------------------------------------
// order_mode: 0 identity, 1 explicit permutation (perm[p] = triple at position p), 2 Feistel
// ids outside [0,U) x [0,I) x [0,I) (point-wise rows: the third column is a label) raise *bad and are
// replaced by 0, so that no later kernel reads or writes outside the tables (daisy_epoch_plan_validate
// reports it; the reference raises IndexError in nn.Embedding, MFRecommender.py:64-65)
template <class KeyT>
__global__ void k_plan_keys(const int32_t *__restrict__ triples, const int64_t *__restrict__ perm,
                            int order_mode, FeistelKey fk, int64_t n, int64_t start, int64_t B,
                            int32_t user_base, int ubits, int64_t U, int64_t I, int pointwise,
                            int *__restrict__ bad, KeyT *__restrict__ key, uint64_t *__restrict__ val,
                            int64_t perm_limit) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t t, p;
        if (order_mode == 1) { p = e; t = perm[e]; }
        else if (order_mode == 2) { t = e; p = (int64_t)feistel_position((uint64_t)e, (uint64_t)n, fk); }
        else { t = e; p = e; }
        if (order_mode == 1 && (t < 0 || t >= perm_limit)) { atomicOr(bad, 2); t = 0; }     // (rows the entry may name)
        const int32_t *row = triples + 3 * (t + start);
        int64_t uu = (int64_t)row[0] - user_base;
        int32_t ri = row[1], rj = row[2];
        if (uu < 0 || uu >= U || ri < 0 || ri >= I || (!pointwise && (rj < 0 || rj >= I))) {
            atomicOr(bad, 1);
            uu = 0; ri = 0; rj = 0;
        }
        key[e] = (KeyT)(((uint64_t)(p / B) << ubits) | (uint64_t)uu);
        val[e] = ((uint64_t)(uint32_t)rj << 32) | (uint32_t)ri;      // (j, i)
    }
}

// from the user-grouped samples: the two item entries of every sample
template <class KeyT>
__global__ void k_plan_entries(const KeyT *__restrict__ skey, const uint64_t *__restrict__ sval,
                               int64_t n, int64_t B, int ibits, uint32_t umask, int pointwise,
                               KeyT *__restrict__ ekey, uint64_t *__restrict__ eval) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = (uint64_t)(p / B);
        const uint32_t s = (uint32_t)(p - (int64_t)k * B);
        const uint32_t uu = (uint32_t)skey[p] & umask;
        const uint64_t ij = sval[p];
        ekey[2 * p] = (KeyT)((k << (ibits + 1)) | ((uint64_t)(uint32_t)ij << 1));
        eval[2 * p] = ((uint64_t)uu << 32) | s;
        const uint32_t jn = pointwise ? (uint32_t)ij : (uint32_t)(ij >> 32);
        ekey[2 * p + 1] = (KeyT)((k << (ibits + 1)) | ((uint64_t)jn << 1) | 1u);
        eval[2 * p + 1] = ((uint64_t)uu << 32) | (s | kNegBit);
    }
}

// 64-bit sort keys -> the 32-bit id arrays the step kernels read
__global__ void k_narrow_keys(const uint64_t *__restrict__ in, int64_t n, uint64_t mask,
                              uint32_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        out[e] = (uint32_t)(in[e] & mask);
}

// runs of equal entry keys -> per-batch run offsets (lower bound of batch k's first key) and
// the narrowed run keys (item << 1 | neg)
template <class KeyT>
__global__ void k_run_finish(const KeyT *__restrict__ full_key, const uint32_t *__restrict__ run_total,
                             int64_t nb, int ibits1, uint32_t *__restrict__ run_key,
                             int32_t *__restrict__ run_off) {
    const int64_t R = *run_total;
    const uint64_t imask = ((uint64_t)1 << ibits1) - 1;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t m = tid; m < R; m += stride) run_key[m] = (uint32_t)((uint64_t)full_key[m] & imask);
    for (int64_t k = tid; k <= nb; k += stride) {
        const uint64_t target = (uint64_t)k << ibits1;
        int64_t lo = 0, hi = R;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint64_t)full_key[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        run_off[k] = (int32_t)lo;
    }
}

__global__ void k_pack_triples(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                               const int32_t *__restrict__ j, int64_t B, int32_t *__restrict__ out) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        out[3 * s] = u[s]; out[3 * s + 1] = i[s]; out[3 * s + 2] = j[s];
    }
}

__global__ void k_unpack_batch(BatchView v, int32_t *__restrict__ u, int32_t *__restrict__ i,
                               int32_t *__restrict__ j, int32_t *__restrict__ ent_item,
                               uint32_t *__restrict__ ent_s, int32_t *__restrict__ ent_u) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < 2 * v.B;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (e < v.B) {
            u[e] = (int32_t)(v.ukey[e] & v.umask);
            i[e] = v.ij[e].x;
            j[e] = v.ij[e].y;
        }
        if (ent_item) ent_item[e] = (int32_t)((v.ekey[e] & v.imask) >> 1);
        if (ent_s) ent_s[e] = v.esu[e].x;
        if (ent_u) ent_u[e] = (int32_t)v.esu[e].y;
    }
}

// out[k] = position of triple ids[k] (a rank's rows of a multi-GPU fit)
__global__ void k_feistel_at(const int64_t *__restrict__ ids, int64_t m, int64_t n, FeistelKey fk,
                             int64_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = ids[e];
        out[e] = (t >= 0 && t < n) ? (int64_t)feistel_position((uint64_t)t, (uint64_t)n, fk) : -1;
    }
}

__global__ void k_feistel_perm(int64_t n, FeistelKey fk, int64_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        out[e] = (int64_t)feistel_position((uint64_t)e, (uint64_t)n, fk);
}

// ---------------------------------------------------------------------------
// forward: scores, coefficients, the seven batch sums
// ---------------------------------------------------------------------------
// A lane group takes a run of LPR consecutive samples: lane x loads the ids of sample x (one
// coalesced read), the rows of FWD_SUB samples are in flight together, their two scores land in
// lane x, and the loss epilogue (exp/log) is evaluated once per run with one sample per lane
// instead of once per sample on a single lane.
template <class C>
struct FwdCfg {
    static constexpr int RUN = C::LPR;
#ifndef DAISY_FWD_SUB
#define DAISY_FWD_SUB 4
#endif
    static constexpr int SUB = (C::NE <= 4) ? DAISY_FWD_SUB : ((C::NE <= 8) ? 2 : 1);
};

template <class C, bool POINTWISE>
__global__ __launch_bounds__(kBlock) void k_fwd(const float *__restrict__ P,
                                                const float *__restrict__ Q, BatchView v, int d,
                                                int loss_type, float gamma,
                                                float2 *__restrict__ coef,
                                                double *__restrict__ partials) {
    constexpr int RUN = FwdCfg<C>::RUN, SUB = FwdCfg<C>::SUB;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t nruns = (v.B + RUN - 1) / RUN;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < nruns; r += gstride) {
        const int64_t t0 = r * RUN;
        const int cnt = (v.B - t0 < RUN) ? (int)(v.B - t0) : RUN;
        // the ids of a tail run are clamped to its last sample, so no load sits behind a branch
        // (a guarded load makes the compiler drain vmcnt first); the duplicates are masked below
        const int64_t il = t0 + ((lane < cnt) ? lane : cnt - 1);
        const int32_t my_u = (int32_t)(v.ukey[il] & v.umask);
        const int2 my_ij = v.ij[il];
        float my_pos = 0.f, my_neg = 0.f;
#pragma unroll
        for (int x0 = 0; x0 < RUN; x0 += SUB) {
            Row<C> p[SUB], qi[SUB], qj[SUB];
#pragma unroll
            for (int y = 0; y < SUB; ++y) {
                const int x = x0 + y;
                p[y].load_clamped(P + (int64_t)group_bcast<C>(my_u, x) * d, lane, d);
                qi[y].load_clamped(Q + (int64_t)group_bcast<C>(my_ij.x, x) * d, lane, d);
                if constexpr (!POINTWISE) qj[y].load_clamped(Q + (int64_t)group_bcast<C>(my_ij.y, x) * d, lane, d);
                else qj[y].zero();                       // point-wise: j is the label, no second row
            }
#pragma unroll
            for (int y = 0; y < SUB; ++y) {
                const float pos = row_dot<C>(p[y], qi[y]);
                const float neg = row_dot<C>(p[y], qj[y]);
                if (lane == x0 + y) { my_pos = pos; my_neg = neg; }
                const float w = (x0 + y < cnt) ? 1.f : 0.f;     // 0 for the clamped duplicates
#pragma unroll
                for (int k = 0; k < C::NE; ++k) {
                    acc[1] = fmaf(w, fabsf(p[y].v[k]), acc[1]);
                    acc[2] = fmaf(w, fabsf(qi[y].v[k]), acc[2]);
                    acc[4] = fmaf(w * p[y].v[k], p[y].v[k], acc[4]);
                    acc[5] = fmaf(w * qi[y].v[k], qi[y].v[k], acc[5]);
                    if constexpr (!POINTWISE) {
                        acc[3] = fmaf(w, fabsf(qj[y].v[k]), acc[3]);
                        acc[6] = fmaf(w * qj[y].v[k], qj[y].v[k], acc[6]);
                    }
                }
            }
        }
        if (lane < cnt) {
            if (v.bu) {                                   // FMRecommender.py:65-66
                const float b0 = v.b0[0], ub = v.bu[my_u];
                my_pos += (ub + v.bi[my_ij.x]) + b0;
                if constexpr (!POINTWISE) my_neg += (ub + v.bi[my_ij.y]) + b0;
            }
            float term, cp, cn;
            pair_coef(loss_type, my_pos, POINTWISE ? (float)my_ij.y : my_neg, gamma, term, cp, cn);
            coef[t0 + lane] = make_float2(cp, cn);
            acc[0] += term;
            acc[7] += cp + cn;                            // d loss / d bias_
        }
    }
    __shared__ double sm[kBlock / kWave][8];
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double w = wave_sum_f64((double)acc[k]);
        if ((threadIdx.x % kWave) == 0) sm[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += sm[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * 8 + threadIdx.x] = t;
    }
}

// fixed-order reduction of the per-workgroup sums -> stats[0..6]; FINALIZE: also the norms and
// the loss (single-GPU step: no all-reduce between the two); reduce_partials_block: bpr_internal.h
template <bool FINALIZE>
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const double *__restrict__ partials,
                                                            int nblocks, double *__restrict__ stats,
                                                            float reg_1, float reg_2,
                                                            double *__restrict__ epoch_acc,
                                                            double *__restrict__ step_loss) {
    reduce_partials_block(partials, nblocks, stats, FINALIZE, reg_1, reg_2, epoch_acc, step_loss);
}

__global__ void k_finalize(double *__restrict__ stats, float reg_1, float reg_2,
                           double *__restrict__ epoch_acc, double *__restrict__ step_loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) finalize_stats(stats, reg_1, reg_2, epoch_acc, step_loss);
}

// ---------------------------------------------------------------------------
// item gradient, legacy mode: one fp32 atomic row per entry (kept for A/B
// measurements; 0.3 TB/s on MI355X)
// ---------------------------------------------------------------------------
template <class C, bool REG>
__global__ __launch_bounds__(kBlock) void k_item_grad_atomic(
    const float *__restrict__ P, const float *__restrict__ Q, BatchView v,
    const float2 *__restrict__ coef, int d, const double *__restrict__ stats, float reg_1,
    float reg_2, float *__restrict__ gQ) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rI = REG ? inv_or_zero(stats[DAISY_ST_NORM_I], reg_2) : 0.f;
    const float rJ = REG ? inv_or_zero(stats[DAISY_ST_NORM_J], reg_2) : 0.f;
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < v.B; s += gstride) {
        const int64_t uu = v.ukey[s] & v.umask;
        const int2 ij = v.ij[s];
        const float2 c = coef[s];
        Row<C> p, gi, gj;
        p.load(P + uu * d, lane, d);
        if constexpr (REG) {
            Row<C> qi, qj;
            qi.load(Q + (int64_t)ij.x * d, lane, d);
            qj.load(Q + (int64_t)(v.pointwise ? ij.x : ij.y) * d, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k) {
                gi.v[k] = fmaf(c.x, p.v[k], fmaf(rI, qi.v[k], reg_1 * sgn(qi.v[k])));
                gj.v[k] = fmaf(c.y, p.v[k], fmaf(rJ, qj.v[k], reg_1 * sgn(qj.v[k])));
            }
        } else {
#pragma unroll
            for (int k = 0; k < C::NE; ++k) {
                gi.v[k] = c.x * p.v[k];
                gj.v[k] = c.y * p.v[k];
            }
        }
        gi.atomic_add_to(gQ + (int64_t)ij.x * d, lane, d);
        if (!v.pointwise) gj.atomic_add_to(gQ + (int64_t)ij.y * d, lane, d);
        if (v.g_bi && lane == 0) {
            unsafeAtomicAdd(v.g_bi + ij.x, c.x);
            if (!v.pointwise) unsafeAtomicAdd(v.g_bi + ij.y, c.y);
        }
    }
}

// ---------------------------------------------------------------------------
// item gradient, reproducible mode: the group that sees the head of an item's
// run owns the row and sums its entries in plan order (fixed, so bitwise
// reproducible).   gQ[r] = sum_e c_e*p_u(e) + reg terms
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_grad_sorted(
    const float *__restrict__ P, const float *__restrict__ Q, const float2 *__restrict__ coef,
    BatchView v, int d, const double *__restrict__ stats, float reg_1, float reg_2,
    float *__restrict__ gQ) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t n = 2 * v.B;
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t pos = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; pos < n; pos += gstride) {
        const uint32_t r = (v.ekey[pos] & v.imask) >> 1;
        if (pos > 0 && ((v.ekey[pos - 1] & v.imask) >> 1) == r) continue;  // not a segment head
        Row<C> acc;
        acc.zero();
        float n_pos = 0.f, n_neg = 0.f, csum = 0.f;
        for (int64_t q = pos; q < n && ((v.ekey[q] & v.imask) >> 1) == r; ++q) {
            const uint2 su = v.esu[q];
            const bool is_neg = (su.x & kNegBit) != 0;
            const float2 c2 = coef[su.x & ~kNegBit];
            const float c = is_neg ? c2.y : c2.x;
            csum += c;
            n_pos += is_neg ? 0.f : 1.f;
            n_neg += (is_neg && !v.pointwise) ? 1.f : 0.f;
            Row<C> p;
            p.load(P + (int64_t)su.y * d, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k) acc.v[k] = fmaf(c, p.v[k], acc.v[k]);
        }
        Row<C> qr;
        qr.load(Q + (int64_t)r * d, lane, d);
        const float w1 = reg_1 * (n_pos + n_neg);
        const float w2 = n_pos * rI + n_neg * rJ;
#pragma unroll
        for (int k = 0; k < C::NE; ++k) acc.v[k] += fmaf(w2, qr.v[k], w1 * sgn(qr.v[k]));
        acc.store(gQ + (int64_t)r * d, lane, d);
        if (v.g_bi && lane == 0) v.g_bi[r] = csum;
    }
}

// ---------------------------------------------------------------------------
// item gradient, throughput mode: segmented reduction over the item-sorted
// entries (data term  sum_e c_e * p_u(e)  only; the regulariser share is added
// by k_item_reg or folded into the commit kernel).
// A workgroup takes a chunk of G*RUN consecutive entries, every lane group a
// run of RUN of them: one load fetches the run's metadata (lane x <- entry x),
// then all RUN row gathers are in flight together.  A segment (= all entries of
// one item) that lies inside one run is summed in registers and stored straight
// to gQ by its group (single owner, no atomics, no LDS).  Only segments that
// cross a run boundary - at most one per boundary - go through an LDS
// accumulator slot (slot s>0: the segment that starts in run s-1; slot 0: the
// segment inherited from the previous chunk); after a barrier each used slot is
// written once: plain store if the segment lies inside the chunk, fp32 atomics
// only when it is shared with a neighbouring chunk (<= 2 rows per chunk).
// ---------------------------------------------------------------------------
template <class C, int RUN_OVERRIDE = 0>
struct RunCfg {
    // needs RUN <= LPR (lane x holds the metadata of entry x); RUN*NE row registers are live at once
    static constexpr int RUN_BY_REGS = (C::NE <= 4) ? 16 : ((C::NE <= 8) ? 4 : 2);
    static constexpr int RUN_AUTO = RUN_BY_REGS < C::LPR ? RUN_BY_REGS : C::LPR;
    static constexpr int RUN = (RUN_OVERRIDE > 0 && RUN_OVERRIDE <= C::LPR) ? RUN_OVERRIDE : RUN_AUTO;
    static constexpr int G = C::GROUPS_PER_BLOCK;
    static constexpr int E = G * RUN;
};

// DET (bitwise reproducible, no atomics): a group does not add its run-crossing partial sums into the
// slot but parks them (<= 2 per group: the segment it continues, the segment it hands on) and the
// slot's finisher adds them in group order; segments shared with a neighbouring chunk leave the chunk
// as edge records that k_item_edges chains in chunk order.  Same data movement, fixed summation order.
struct ItemEdges {
    float *vec;          // [2*nchunks][d]  partial gradient rows; [2c] head edge (inherited), [2c+1] tail edge
    int32_t *item;       // [2*nchunks]     their item (-1: none)
    float *b;            // [2*nchunks]     FM: partial coefficient sums
    int32_t *whole;      // [nchunks]       the head edge's segment also runs on into the next chunk
    const double *halt = nullptr;   // BatchView::halt of the step the records belong to
};

template <class C, int RUN_OVERRIDE = 0, bool DET = false, bool XH = false>      // XH: the gathered rows are bf16
__global__ __launch_bounds__(kBlock) void k_item_grad_chunked(const float *__restrict__ P,
                                                              const float2 *__restrict__ coef,
                                                              BatchView v, int d,
                                                              float *__restrict__ gQ, ItemEdges edges = ItemEdges{}) {
    if (halted(v.halt)) return;
    constexpr int G = RunCfg<C, RUN_OVERRIDE>::G, RUN = RunCfg<C, RUN_OVERRIDE>::RUN,
                  E = RunCfg<C, RUN_OVERRIDE>::E;
    constexpr int ROWF = C::NE * C::LPR;
    __shared__ float slot_acc[DET ? 1 : (G + 1) * ROWF];
    __shared__ float slot_b[G + 1];              // FM: sum of the coefficients (d loss / d i_bias)
    __shared__ int slot_item[G + 1], slot_shared[G + 1];
    __shared__ int run_first[G], run_last[G];
    __shared__ float part_acc[DET ? 2 * G * ROWF : 1];     // DET: [group][head|tail] parked partial sums
    __shared__ float part_b[DET ? 2 * G : 1];
    __shared__ int part_slot[DET ? 2 * G : 1];             //      the slot each belongs to (-1: unused)

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const int64_t n = 2 * v.B;
    const int64_t nchunks = (n + E - 1) / E;

    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * E;
        const int64_t t0 = c0 + (int64_t)group * RUN;
        const int64_t t1 = (t0 + RUN < n) ? (t0 + RUN) : n;
        const int cnt = (t0 < n) ? (int)(t1 - t0) : 0;       // entries of this run

        // ---- hop 1: the run's metadata, lane x <- entry x; plus the entries just before and
        // after the run (same address on every lane: one request)
        // Every load is unconditional on a clamped address (n >= 1 here): a branch around a
        // load makes the compiler drain vmcnt before it, which would serialise these requests
        // into one memory round trip each.
        const int64_t last = n - 1;
        const int64_t il = (t0 + lane < n) ? (t0 + lane) : last;
        const uint32_t k_me = v.ekey[il];
        const uint2 su_me = v.esu[il];
        const uint32_t k_prev = v.ekey[(t0 > 0) ? ((t0 - 1 < n) ? t0 - 1 : last) : 0];
        const uint32_t k_next = v.ekey[(t1 < n) ? t1 : last];
        const uint32_t k_cprev = v.ekey[(c0 > 0) ? c0 - 1 : 0];
        const int32_t my_item = (lane < cnt) ? (int32_t)((k_me & v.imask) >> 1) : -1;
        const uint2 my_su = (lane < cnt) ? su_me : make_uint2(0u, 0u);
        const int32_t item_prev = (cnt > 0 && t0 > 0) ? (int32_t)((k_prev & v.imask) >> 1) : -1;
        const int32_t item_next = (cnt > 0 && t1 < n) ? (int32_t)((k_next & v.imask) >> 1) : -1;
        const int32_t chunk_prev_item = (c0 > 0) ? (int32_t)((k_cprev & v.imask) >> 1) : -1;
        if constexpr (!DET) {
            for (int e = tid; e < (G + 1) * ROWF; e += kBlock) slot_acc[e] = 0.f;
        } else {
            if (tid < 2 * G) part_slot[tid] = -1;
        }
        if (tid <= G) { slot_item[tid] = -1; slot_shared[tid] = 0; slot_b[tid] = 0.f; }
        const int32_t item_first = group_bcast<C>(my_item, 0);
        const int32_t item_last = __shfl(my_item, cnt > 0 ? cnt - 1 : 0, C::LPR);
        if (lane == 0) {
            run_first[group] = cnt > 0 ? item_first : -2;
            run_last[group] = cnt > 0 ? item_last : -2;
        }

        // ---- hop 2: coefficient of entry `lane`, and all row gathers of the run
        const float2 c2 = coef[su_me.x & ~kNegBit];
        const float my_c = (lane < cnt) ? ((su_me.x & kNegBit) ? c2.y : c2.x) : 0.f;
        Row<C> p[RUN];
        if (__all(cnt == RUN)) {        // wave-uniform: every run of this wave is full
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const uint32_t ux = group_bcast<C>(my_su.y, x);
                if constexpr (XH) p[x].load_bf16(reinterpret_cast<const uint16_t *>(P) + (int64_t)ux * d, lane, d);
                else p[x].load(P + (int64_t)ux * d, lane, d);
            }
        } else {
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const uint32_t ux = group_bcast<C>(my_su.y, x);
                if (x >= cnt) p[x].zero();
                else if constexpr (XH) p[x].load_bf16(reinterpret_cast<const uint16_t *>(P) + (int64_t)ux * d, lane, d);
                else p[x].load(P + (int64_t)ux * d, lane, d);
            }
        }
        __syncthreads();

        if (cnt > 0) {
            const bool cont = (t0 > 0) && (item_prev == item_first);
            int cur_slot = -1;                      // >= 0: the current segment began before this run
            if (cont) {
                if (group == 0) cur_slot = 0;       // inherited from the previous chunk
                else {
                    int gs = group - 1;
                    while (gs > 0 && run_first[gs] == item_first && run_last[gs - 1] == item_first) --gs;
                    // the segment holds the last entry of run gs; it began there unless run 0 is
                    // all this item and the chunk itself continues the previous chunk
                    const bool inherited = (gs == 0) && (run_first[0] == item_first) && (c0 > 0) &&
                                           (chunk_prev_item == item_first);
                    cur_slot = inherited ? 0 : gs + 1;
                }
            }
            int32_t cur_item = item_first;
            Row<C> acc;
            acc.zero();
            float accb = 0.f;
            auto finish = [&](bool ends_here, bool to_next_chunk) {
                if (cur_slot < 0 && ends_here) {    // interior: this group owns gQ[cur_item]
                    acc.store(gQ + (int64_t)cur_item * d, lane, d);
                    if (v.g_bi && lane == 0) v.g_bi[cur_item] = accb;
                } else {                            // crosses a run boundary: LDS slot
                    const int s = (cur_slot >= 0) ? cur_slot : group + 1;
                    if constexpr (DET) {            // park it: head partial (continued segment) or tail partial
                        const int q = group * 2 + ((cur_slot >= 0) ? 0 : 1);
                        float *dst = part_acc + q * ROWF;
#pragma unroll
                        for (int k = 0; k < C::NE; ++k) dst[k * C::LPR + lane] = acc.v[k];
                        if (lane == 0) {
                            part_slot[q] = s;
                            part_b[q] = accb;
                            slot_item[s] = cur_item;
                            if (to_next_chunk) slot_shared[s] = 1;
                        }
                    } else {
                        float *dst = slot_acc + s * ROWF;
#pragma unroll
                        for (int k = 0; k < C::NE; ++k) atomicAdd(dst + k * C::LPR + lane, acc.v[k]);
                        if (lane == 0) {
                            slot_item[s] = cur_item;
                            atomicAdd(&slot_b[s], accb);
                            if (to_next_chunk) slot_shared[s] = 1;
                        }
                    }
                }
            };
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                if (x < cnt) {
                    const int32_t it = group_bcast<C>(my_item, x);
                    const float cx = group_bcast<C>(my_c, x);
                    if (it != cur_item) {           // previous segment ended inside this run
                        finish(true, false);
                        cur_item = it;
                        cur_slot = -1;
                        acc.zero();
                        accb = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) acc.v[k] = fmaf(cx, p[x].v[k], acc.v[k]);
                    accb += cx;
                }
            }
            const bool continues = (t1 < n) && (item_next == cur_item);
            finish(!continues, continues && (group == G - 1));
        }
        __syncthreads();

        if constexpr (DET) {
            if (tid == 0) { edges.item[2 * chunk] = -1; edges.item[2 * chunk + 1] = -1; edges.whole[chunk] = 0; }
            __syncthreads();
        }
        // one write per used slot (G+1 slots over G groups)
        for (int s = group; s <= G; s += G) {
            const int r = slot_item[s];
            if (r < 0) continue;
            Row<C> g;
            if constexpr (DET) {                    // the parked partials of this slot, in group order
                g.zero();
                float gb = 0.f;
                for (int q = 0; q < 2 * G; ++q) {
                    if (part_slot[q] != s) continue;
                    const float *src = part_acc + q * ROWF;
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) g.v[k] += src[k * C::LPR + lane];
                    gb += part_b[q];
                }
                const bool from_prev = (s == 0), to_next = slot_shared[s] != 0;
                if (from_prev || to_next) {
                    const int64_t e = 2 * chunk + (from_prev ? 0 : 1);
                    g.store(edges.vec + e * d, lane, d);
                    if (lane == 0) {
                        edges.item[e] = r;
                        edges.b[e] = gb;
                        if (from_prev && to_next) edges.whole[chunk] = 1;
                    }
                } else {
                    g.store(gQ + (int64_t)r * d, lane, d);
                    if (v.g_bi && lane == 0) v.g_bi[r] = gb;
                }
                continue;
            }
            const float *src = slot_acc + s * ROWF;
#pragma unroll
            for (int k = 0; k < C::NE; ++k) g.v[k] = src[k * C::LPR + lane];
            if (s == 0 || slot_shared[s]) {
                g.atomic_add_to(gQ + (int64_t)r * d, lane, d);
                if (v.g_bi && lane == 0) unsafeAtomicAdd(v.g_bi + r, slot_b[s]);
            } else {
                g.store(gQ + (int64_t)r * d, lane, d);
                if (v.g_bi && lane == 0) v.g_bi[r] = slot_b[s];
            }
        }
        __syncthreads();   // the slots are reused by the next chunk
    }
}

// DET: chains of edge records - the chunk whose TAIL edge starts a segment owns it and adds the head edges
// of the chunks it runs through, in chunk order (single writer per row, fixed order)
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_edges(ItemEdges edges, int64_t nchunks, int d,
                                                       float *__restrict__ gQ, float *__restrict__ g_bi) {
    if (halted(edges.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t c = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; c < nchunks; c += gstride) {
        const int it = edges.item[2 * c + 1];
        if (it < 0) continue;
        Row<C> acc, t;
        acc.load(edges.vec + (2 * c + 1) * d, lane, d);
        float sb = edges.b[2 * c + 1];
        for (int64_t k = c + 1; k < nchunks && edges.item[2 * k] == it; ++k) {
            t.load(edges.vec + (2 * k) * d, lane, d);
#pragma unroll
            for (int q = 0; q < C::NE; ++q) acc.v[q] += t.v[q];
            sb += edges.b[2 * k];
            if (!edges.whole[k]) break;
        }
        acc.store(gQ + (int64_t)it * d, lane, d);
        if (g_bi && lane == 0) g_bi[it] = sb;
    }
}

// regulariser share of the item gradient (MFRecommender.py:88-89), one lane group per
// distinct item of the batch (head run of the plan's run list):
//   gQ[item] += reg_1*(np+nn)*sign(q) + reg_2*(np/|Q[i]|_F + nn/|Q[j]|_F)*q
__device__ __forceinline__ bool run_head(const BatchView &v, int64_t m, int64_t m0, int64_t m1,
                                         int64_t &item, float &fp, float &fn) {
    const uint32_t key = v.run_key[m];
    if (m > m0 && (v.run_key[m - 1] >> 1) == (key >> 1)) return false;   // the item's second run
    item = key >> 1;
    const float c = (float)v.run_cnt[m];
    fp = (key & 1u) ? 0.f : c;
    fn = (key & 1u) ? c : 0.f;
    if (!(key & 1u) && m + 1 < m1 && (v.run_key[m + 1] >> 1) == (key >> 1)) fn = (float)v.run_cnt[m + 1];
    if (v.pointwise) fn = 0.f;      // the negative slots are inert copies
    return true;
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_item_reg(const float *__restrict__ Q, BatchView v, int d,
                                                     const double *__restrict__ stats, float reg_1,
                                                     float reg_2, float *__restrict__ gQ) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t m0 = v.run_off[0], m1 = v.run_off[1];
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t m = m0 + (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; m < m1; m += gstride) {
        int64_t r;
        float fp, fn;
        if (!run_head(v, m, m0, m1, r, fp, fn)) continue;
        Row<C> g, q;
        g.load(gQ + r * d, lane, d);
        q.load(Q + r * d, lane, d);
        const float w1 = reg_1 * (fp + fn), w2 = fp * rI + fn * rJ;
#pragma unroll
        for (int k = 0; k < C::NE; ++k) g.v[k] += fmaf(w2, q.v[k], w1 * sgn(q.v[k]));
        g.store(gQ + r * d, lane, d);
    }
}

// ---------------------------------------------------------------------------
// user rows: the batch is grouped by user, the group that sees the head of a
// user's run owns P[u]:  g = sum_b (cp*q_i + cn*q_j) + n*(reg_1*sign(p) + reg_2*p/|P[u]|_F)
//   SGD : P[u] -= lr*g  (in place, single writer)      GRAD: gP[u] = g
// ---------------------------------------------------------------------------
template <class C, bool SGD>
__global__ __launch_bounds__(kBlock) void k_user(float *__restrict__ P, const float *__restrict__ Q,
                                                 BatchView v, const float2 *__restrict__ coef, int d,
                                                 const double *__restrict__ stats, float lr,
                                                 float reg_1, float reg_2, float *__restrict__ gP) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rU = inv_or_zero(stats[DAISY_ST_NORM_U], reg_2);
    for (int64_t pos = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; pos < v.B; pos += gstride) {
        const uint32_t uu = v.ukey[pos] & v.umask;
        if (pos > 0 && (v.ukey[pos - 1] & v.umask) == uu) continue;  // not the head of this user's run
        Row<C> p, acc;
        p.load(P + (int64_t)uu * d, lane, d);
        acc.zero();
        float n = 0.f, csum = 0.f;
        for (int64_t q = pos; q < v.B && (v.ukey[q] & v.umask) == uu; ++q) {
            const float2 c = coef[q];
            csum += c.x + c.y;
            const int2 ij = v.ij[q];
            Row<C> qi, qj;
            qi.load(Q + (int64_t)ij.x * d, lane, d);
            qj.load(Q + (int64_t)(v.pointwise ? ij.x : ij.y) * d, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k)
                acc.v[k] = fmaf(c.x, qi.v[k], fmaf(c.y, qj.v[k], acc.v[k]));
            n += 1.f;
        }
        const float w1 = reg_1 * n, w2 = rU * n;
#pragma unroll
        for (int k = 0; k < C::NE; ++k) {
            const float g = acc.v[k] + fmaf(w2, p.v[k], w1 * sgn(p.v[k]));
            if constexpr (SGD) p.v[k] = fmaf(-lr, g, p.v[k]);
            else p.v[k] = g;
        }
        if constexpr (SGD) p.store(P + (int64_t)uu * d, lane, d);
        else p.store(gP + (int64_t)uu * d, lane, d);
        if (v.bu && lane == 0) {                           // d loss / d u_bias[u] = sum (cp + cn)
            if constexpr (SGD) v.bu[uu] = fmaf(-lr, csum, v.bu[uu]);
            else v.g_bu[uu] = csum;
        }
    }
    if (v.bu && blockIdx.x == 0 && threadIdx.x == 0) {     // the global bias (FMRecommender.py:59)
        const float g0 = (float)stats[DAISY_ST_SUM_COEF];
        if constexpr (SGD) v.b0[0] = fmaf(-lr, g0, v.b0[0]);
        else v.g_b0[0] = g0;
    }
}

// ---------------------------------------------------------------------------
// commit the item rows: Q[r] -= lr*gQ[r]; gQ[r] = 0, one lane group per distinct
// item of the batch (head run of the plan's run list; dense != 0: every row, after an
// all-reduce of gQ).  WITH_REG: gQ holds the data term only and the regulariser
// share is added here (single-GPU SGD step: saves one pass over the touched rows).
// ---------------------------------------------------------------------------
template <class C, bool WITH_REG>
__global__ __launch_bounds__(kBlock) void k_item_apply(float *__restrict__ Q, float *__restrict__ gQ,
                                                       BatchView v, int64_t n_dense, int d, float lr,
                                                       int dense, const double *__restrict__ stats,
                                                       float reg_1, float reg_2) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    float rI = 0.f, rJ = 0.f;
    if constexpr (WITH_REG) {
        rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
        rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    }
    const int64_t m0 = dense ? 0 : (int64_t)v.run_off[0];
    const int64_t m1 = dense ? n_dense : (int64_t)v.run_off[1];
    for (int64_t m = m0 + (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; m < m1; m += gstride) {
        int64_t r = m;
        float fp = 0.f, fn = 0.f;
        if (!dense && !run_head(v, m, m0, m1, r, fp, fn)) continue;
        Row<C> g, q, z;
        g.load(gQ + r * d, lane, d);
        q.load(Q + r * d, lane, d);
        z.zero();
        if constexpr (WITH_REG) {
            const float w1 = reg_1 * (fp + fn), w2 = fp * rI + fn * rJ;
#pragma unroll
            for (int k = 0; k < C::NE; ++k) g.v[k] += fmaf(w2, q.v[k], w1 * sgn(q.v[k]));
        }
#pragma unroll
        for (int k = 0; k < C::NE; ++k) q.v[k] = fmaf(-lr, g.v[k], q.v[k]);
        q.store(Q + r * d, lane, d);
        z.store(gQ + r * d, lane, d);
        if (v.bi && lane == 0) {
            v.bi[r] = fmaf(-lr, v.g_bi[r], v.bi[r]);
            v.g_bi[r] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------
// user rows, throughput mode: the same run/slot scheme as k_item_grad_chunked over the
// user-grouped samples, so every lane group streams RUN samples (2 item rows + the user
// row each, all gathers in flight together) whatever the run lengths are.
//   in-run user:            its group owns P[u]: P[u] -= lr*(sum + n*reg(p))     in place
//   run crossing groups:    LDS slot, finished by one group after the barrier
//   run crossing chunks:    partial (sum, n) to the chunk's head/tail EDGE record; k_user_edges
//                           walks each chain from the chunk where the run starts and updates P[u]
//                           (no global atomics, and P[u] is read before anybody writes it)
// ---------------------------------------------------------------------------
template <class C>
struct UserRunCfg {
#ifndef DAISY_USER_RUN
#define DAISY_USER_RUN 8      // samples per lane group per chunk (3 rows each in flight): 4 -> 8 was +6 % on the step
#endif
    static constexpr int RUN = (C::NE <= 4 && C::LPR >= DAISY_USER_RUN) ? DAISY_USER_RUN : ((C::LPR >= 4) ? 4 : C::LPR);
    static constexpr int G = C::GROUPS_PER_BLOCK;
    static constexpr int E = G * RUN;
};

template <class C>
__device__ __forceinline__ void user_finish_row(Row<C> &p, const Row<C> &acc, float n, float lr,
                                                float reg_1, float rU) {
    const float w1 = reg_1 * n, w2 = rU * n;
#pragma unroll
    for (int k = 0; k < C::NE; ++k) {
        const float g = acc.v[k] + fmaf(w2, p.v[k], w1 * sgn(p.v[k]));
        p.v[k] = fmaf(-lr, g, p.v[k]);
    }
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_user_chunked(
    float *__restrict__ P, const float *__restrict__ Q, BatchView v, const float2 *__restrict__ coef,
    int d, const double *__restrict__ stats, float lr, float reg_1, float reg_2,
    float *__restrict__ edge_vec, int32_t *__restrict__ edge_user, float *__restrict__ edge_n,
    int32_t *__restrict__ edge_whole) {
    if (halted(v.halt)) return;
    constexpr int G = UserRunCfg<C>::G, RUN = UserRunCfg<C>::RUN, E = UserRunCfg<C>::E;
    constexpr int ROWF = C::NE * C::LPR;
    // run-crossing partial sums are parked per group (<= 2: the run it continues, the run it hands on)
    // and added by the slot's finisher in group order: fixed summation order, no LDS atomics
    __shared__ float part_acc[2 * G * ROWF];
    __shared__ float part_n[2 * G], part_b[2 * G];   // part_b: FM, sum of (cp + cn) = d loss / d u_bias
    __shared__ int part_slot[2 * G];
    __shared__ int slot_user[G + 1], slot_next[G + 1];
    __shared__ int run_first[G], run_last[G];

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const int64_t n = v.B;
    const int64_t nchunks = (n + E - 1) / E;
    const float rU = inv_or_zero(stats[DAISY_ST_NORM_U], reg_2);

    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * E;
        const int64_t t0 = c0 + (int64_t)group * RUN;
        const int64_t t1 = (t0 + RUN < n) ? (t0 + RUN) : n;
        const int cnt = (t0 < n) ? (int)(t1 - t0) : 0;

        // ---- hop 1: metadata of the run (lane x <- sample x) and its two neighbours
        // (unconditional loads on clamped addresses: see k_item_grad_chunked)
        const int64_t last = n - 1;
        const int64_t il = (t0 + lane < n) ? (t0 + lane) : last;
        const uint32_t k_me = v.ukey[il];
        const int2 ij_me = v.ij[il];
        const float2 c_me = coef[il];
        const float2 my_c = (lane < cnt) ? c_me : make_float2(0.f, 0.f);
        const uint32_t k_prev = v.ukey[(t0 > 0) ? ((t0 - 1 < n) ? t0 - 1 : last) : 0];
        const uint32_t k_next = v.ukey[(t1 < n) ? t1 : last];
        const uint32_t k_cprev = v.ukey[(c0 > 0) ? c0 - 1 : 0];
        const int32_t my_user = (lane < cnt) ? (int32_t)(k_me & v.umask) : -1;
        const int2 my_ij = (lane < cnt) ? ij_me : make_int2(0, 0);
        const int32_t user_prev = (cnt > 0 && t0 > 0) ? (int32_t)(k_prev & v.umask) : -1;
        const int32_t user_next = (cnt > 0 && t1 < n) ? (int32_t)(k_next & v.umask) : -1;
        const int32_t chunk_prev_user = (c0 > 0) ? (int32_t)(k_cprev & v.umask) : -1;
        if (tid < 2 * G) part_slot[tid] = -1;
        if (tid <= G) { slot_user[tid] = -1; slot_next[tid] = 0; }
        const int32_t user_first = group_bcast<C>(my_user, 0);
        const int32_t user_last = __shfl(my_user, cnt > 0 ? cnt - 1 : 0, C::LPR);
        if (lane == 0) {
            run_first[group] = cnt > 0 ? user_first : -2;
            run_last[group] = cnt > 0 ? user_last : -2;
        }

        // ---- hop 2: the three rows of every sample of the run
        Row<C> qi[RUN], qj[RUN], pr[RUN];
        if (__all(cnt == RUN)) {        // wave-uniform: no branch between the 3*RUN gathers
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const int32_t ux = group_bcast<C>(my_user, x);
                const int ix = group_bcast<C>(my_ij.x, x);
                const int jx = v.pointwise ? ix : group_bcast<C>(my_ij.y, x);
                qi[x].load(Q + (int64_t)ix * d, lane, d);
                qj[x].load(Q + (int64_t)jx * d, lane, d);
                pr[x].load(P + (int64_t)ux * d, lane, d);
            }
        } else {
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const int32_t ux = group_bcast<C>(my_user, x);
                const int ix = group_bcast<C>(my_ij.x, x);
                const int jx = v.pointwise ? ix : group_bcast<C>(my_ij.y, x);
                if (x < cnt) {
                    qi[x].load(Q + (int64_t)ix * d, lane, d);
                    qj[x].load(Q + (int64_t)jx * d, lane, d);
                    pr[x].load(P + (int64_t)ux * d, lane, d);
                } else {
                    qi[x].zero(); qj[x].zero(); pr[x].zero();
                }
            }
        }
        __syncthreads();

        if (cnt > 0) {
            const bool cont = (t0 > 0) && (user_prev == user_first);
            int cur_slot = -1;
            if (cont) {
                if (group == 0) cur_slot = 0;
                else {
                    int gs = group - 1;
                    while (gs > 0 && run_first[gs] == user_first && run_last[gs - 1] == user_first) --gs;
                    const bool inherited = (gs == 0) && (run_first[0] == user_first) && (c0 > 0) &&
                                           (chunk_prev_user == user_first);
                    cur_slot = inherited ? 0 : gs + 1;
                }
            }
            int32_t cur_user = user_first;
            Row<C> pcur = pr[0];                     // P row of the current run's user
            Row<C> acc;
            acc.zero();
            float cn_ = 0.f, cb_ = 0.f;
            auto finish = [&](bool ends_here, bool to_next_chunk, const Row<C> &prow) {
                if (cur_slot < 0 && ends_here) {     // this group owns P[cur_user]
                    Row<C> pn = prow;
                    user_finish_row<C>(pn, acc, cn_, lr, reg_1, rU);
                    pn.store(P + (int64_t)cur_user * d, lane, d);
                    if (v.bu && lane == 0) v.bu[cur_user] = fmaf(-lr, cb_, v.bu[cur_user]);
                } else {
                    const int s = (cur_slot >= 0) ? cur_slot : group + 1;
                    const int q = group * 2 + ((cur_slot >= 0) ? 0 : 1);
                    float *dst = part_acc + q * ROWF;
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) dst[k * C::LPR + lane] = acc.v[k];
                    if (lane == 0) {
                        part_slot[q] = s;
                        part_n[q] = cn_;
                        part_b[q] = cb_;
                        slot_user[s] = cur_user;
                        if (to_next_chunk) slot_next[s] = 1;
                    }
                }
            };
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                if (x < cnt) {
                    const int32_t ux = group_bcast<C>(my_user, x);
                    const float cp = group_bcast<C>(my_c.x, x);
                    const float cn = group_bcast<C>(my_c.y, x);
                    if (ux != cur_user) {
                        finish(true, false, pcur);
                        cur_user = ux;
                        cur_slot = -1;
                        pcur = pr[x];
                        acc.zero();
                        cn_ = 0.f;
                        cb_ = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k)
                        acc.v[k] = fmaf(cp, qi[x].v[k], fmaf(cn, qj[x].v[k], acc.v[k]));
                    cn_ += 1.f;
                    cb_ += cp + cn;
                }
            }
            const bool continues = (t1 < n) && (user_next == cur_user);
            finish(!continues, continues && (group == G - 1), pcur);
        }
        __syncthreads();

        // ---- one finisher per used slot; runs shared with a neighbouring chunk go to the edges
        if (tid == 0) { edge_user[2 * chunk] = -1; edge_user[2 * chunk + 1] = -1; edge_whole[chunk] = 0; }
        __syncthreads();
        for (int s = group; s <= G; s += G) {
            const int uu = slot_user[s];
            if (uu < 0) continue;
            Row<C> g;
            g.zero();
            float ns = 0.f, sb = 0.f;
            for (int q = 0; q < 2 * G; ++q) {
                if (part_slot[q] != s) continue;
                const float *src = part_acc + q * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) g.v[k] += src[k * C::LPR + lane];
                ns += part_n[q];
                sb += part_b[q];
            }
            const bool from_prev = (s == 0), to_next = slot_next[s] != 0;
            if (!from_prev && !to_next) {
                Row<C> p;
                p.load(P + (int64_t)uu * d, lane, d);
                user_finish_row<C>(p, g, ns, lr, reg_1, rU);
                p.store(P + (int64_t)uu * d, lane, d);
                if (v.bu && lane == 0) v.bu[uu] = fmaf(-lr, sb, v.bu[uu]);
            } else {
                const int64_t e = 2 * chunk + (from_prev ? 0 : 1);
                g.store(edge_vec + e * d, lane, d);
                if (lane == 0) {
                    edge_user[e] = uu;
                    edge_n[2 * e] = ns;
                    edge_n[2 * e + 1] = sb;
                    if (from_prev && to_next) edge_whole[chunk] = 1;
                }
            }
        }
        __syncthreads();
    }
    if (v.bu && blockIdx.x == 0 && tid == 0)               // the global bias (FMRecommender.py:59)
        v.b0[0] = fmaf(-lr, (float)stats[DAISY_ST_SUM_COEF], v.b0[0]);
}

// chains of edge records: the chunk whose TAIL edge starts a run owns it
template <class C>
__global__ __launch_bounds__(kBlock) void k_user_edges(float *__restrict__ P, int64_t nchunks, int d,
                                                       const double *__restrict__ stats, float lr,
                                                       float reg_1, float reg_2,
                                                       const float *__restrict__ edge_vec,
                                                       const int32_t *__restrict__ edge_user,
                                                       const float *__restrict__ edge_n,
                                                       const int32_t *__restrict__ edge_whole,
                                                       float *__restrict__ u_bias, const double *__restrict__ halt) {
    if (halted(halt)) return;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rU = inv_or_zero(stats[DAISY_ST_NORM_U], reg_2);
    for (int64_t c = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; c < nchunks; c += gstride) {
        const int uu = edge_user[2 * c + 1];
        if (uu < 0) continue;
        Row<C> acc, t;
        acc.load(edge_vec + (2 * c + 1) * d, lane, d);
        float ns = edge_n[2 * (2 * c + 1)], sb = edge_n[2 * (2 * c + 1) + 1];
        for (int64_t k = c + 1; k < nchunks && edge_user[2 * k] == uu; ++k) {
            t.load(edge_vec + (2 * k) * d, lane, d);
#pragma unroll
            for (int q = 0; q < C::NE; ++q) acc.v[q] += t.v[q];
            ns += edge_n[2 * (2 * k)];
            sb += edge_n[2 * (2 * k) + 1];
            if (!edge_whole[k]) break;
        }
        Row<C> p;
        p.load(P + (int64_t)uu * d, lane, d);
        user_finish_row<C>(p, acc, ns, lr, reg_1, rU);
        p.store(P + (int64_t)uu * d, lane, d);
        if (u_bias && lane == 0) u_bias[uu] = fmaf(-lr, sb, u_bias[uu]);
    }
}

// torch.optim.Adam single-tensor math (exp_avg.lerp_, addcmul_, addcdiv_), dense
__global__ __launch_bounds__(kBlock) void k_adam_dense(float *__restrict__ W, float *__restrict__ g,
                                                       float *__restrict__ m, float *__restrict__ v,
                                                       int64_t n, float step_size, float beta1,
                                                       float beta2, float eps, float bc2_sqrt) {
    const float w1 = 1.f - beta1, w2 = 1.f - beta2;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[e];
        const float mm = fmaf(w1, gg - m[e], m[e]);          // lerp(m, g, 1-beta1)
        const float vv = fmaf(w2 * gg, gg, beta2 * v[e]);    // mul_(beta2).addcmul_(g,g,1-beta2)
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        W[e] = W[e] - step_size * (mm / denom);
        m[e] = mm;
        v[e] = vv;
        g[e] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Dense Adam without touching every row in every step.  torch's Adam moves every element in every step (the moments
// decay without a gradient), which at table sizes beyond the batch is most of the step's traffic.  But a row's
// update sequence depends on nothing but its own gradients: a row without gradient in steps a+1..b can be brought
// from its state after step a to its state after step b in registers (b - a zero-gradient updates, the very
// expressions of k_adam_dense with g = 0) whenever it is next needed - before the step that reads it, or at a flush.
// last[r] = the step row r has been updated to.  table[s] = (lr / (1 - beta1^s), sqrt(1 - beta2^s)) as the dense
// entry point computes them on the host, so the replay uses the same bits.  Result: identical to daisy_adam_dense
// in every step (tested bit for bit), HBM traffic proportional to the rows a step touches.
// ---------------------------------------------------------------------------------------------------------
struct AdamTable { float *W, *g, *m, *v; int32_t *last; };
struct AdamHyper { const float2 *table; float beta1, beta2, eps; int32_t t; };

// the lane group that raises last[row] from below `upto` wins the row: it replays the zero-gradient steps
// last+1 .. upto-1 (catch-up, WITH_G = false: upto = t, the row is then current for step t's forward) or applies step
// t itself with the row's gradient and clears it (WITH_G: upto = t + 1)
template <class C, bool WITH_G>
__device__ __forceinline__ void adam_claim_row(const AdamTable &T, int64_t row, int d, const AdamHyper &h, int lane) {
    const int32_t target = WITH_G ? h.t : h.t - 1;
    int32_t old = 0;
    if (lane == 0) old = atomicMax(T.last + row, target);
    old = group_bcast<C>(old, 0);
    if (old >= target) return;
    Row<C> w, m, v, g;
    w.load(T.W + row * d, lane, d); m.load(T.m + row * d, lane, d); v.load(T.v + row * d, lane, d);
    g.zero();
    for (int32_t s = old + 1; s < (WITH_G ? h.t : h.t); ++s) {           // zero-gradient steps old+1 .. t-1
        const float2 c = h.table[s];
        adam_row<C>(w, m, v, g, c.x, c.y, h.beta1, h.beta2, h.eps);
    }
    if constexpr (WITH_G) {
        g.load(T.g + row * d, lane, d);
        const float2 c = h.table[h.t];
        adam_row<C>(w, m, v, g, c.x, c.y, h.beta1, h.beta2, h.eps);
        g.zero();
        g.store(T.g + row * d, lane, d);
    }
    w.store(T.W + row * d, lane, d); m.store(T.m + row * d, lane, d); v.store(T.v + row * d, lane, d);
}

// every row reference of the current batch: user of sample s, its item(s)
template <class C, bool WITH_G>
__global__ __launch_bounds__(kBlock) void k_adam_batch_rows(BatchView v, int d, AdamTable TP, AdamTable TQ, AdamHyper h) {
    const int lane = threadIdx.x % C::LPR, group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < v.B; s += gstride) {
        const int2 ij = v.ij[s];
        adam_claim_row<C, WITH_G>(TP, (int64_t)(v.ukey[s] & v.umask), d, h, lane);
        adam_claim_row<C, WITH_G>(TQ, (int64_t)ij.x, d, h, lane);
        if (!v.pointwise) adam_claim_row<C, WITH_G>(TQ, (int64_t)ij.y, d, h, lane);
    }
}

// all rows up to step t (end of an epoch, before the tables are read by anything but a step)
template <class C>
__global__ __launch_bounds__(kBlock) void k_adam_flush(AdamTable T, int64_t rows, int d, AdamHyper h) {
    const int lane = threadIdx.x % C::LPR, group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < rows; r += gstride) {
        const int32_t old = T.last[r];
        if (old >= h.t) continue;
        Row<C> w, m, v, g;
        w.load(T.W + r * d, lane, d); m.load(T.m + r * d, lane, d); v.load(T.v + r * d, lane, d);
        g.zero();
        for (int32_t s = old + 1; s <= h.t; ++s) {
            const float2 c = h.table[s];
            adam_row<C>(w, m, v, g, c.x, c.y, h.beta1, h.beta2, h.eps);
        }
        w.store(T.W + r * d, lane, d); m.store(T.m + r * d, lane, d); v.store(T.v + r * d, lane, d);
        if (lane == 0) T.last[r] = h.t;
    }
}

// torch.optim.Adagrad single-tensor math (defaults): state_sum.addcmul_(g, g); w.addcdiv_(g, sqrt(state_sum) + eps, -lr)
__global__ __launch_bounds__(kBlock) void k_adagrad_dense(float *__restrict__ W, float *__restrict__ g,
                                                          float *__restrict__ ss, int64_t n, float lr, float eps) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[e];
        const float s2 = fmaf(gg, gg, ss[e]);
        W[e] = W[e] - lr * (gg / (sqrtf(s2) + eps));
        ss[e] = s2;
        g[e] = 0.f;
    }
}

// torch.optim.RMSprop single-tensor math (defaults): sq.mul_(alpha).addcmul_(g, g, 1-alpha); w.addcdiv_(g, sqrt(sq) + eps, -lr)
__global__ __launch_bounds__(kBlock) void k_rmsprop_dense(float *__restrict__ W, float *__restrict__ g,
                                                          float *__restrict__ sq, int64_t n, float lr, float alpha,
                                                          float eps) {
    const float w2 = 1.f - alpha;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[e];
        const float s2 = fmaf(w2 * gg, gg, alpha * sq[e]);
        W[e] = W[e] - lr * (gg / (sqrtf(s2) + eps));
        sq[e] = s2;
        g[e] = 0.f;
    }
}

// ---------------------------------------------------------------------------
// plan construction (host side)
// ---------------------------------------------------------------------------
static int plan_alloc(daisy_epoch_plan **out, int64_t max_triples, int64_t U, int64_t I) {
    daisy_epoch_plan *p = new daisy_epoch_plan();
    memset(p, 0, sizeof(*p));          // no device memory yet: each layout allocates at its first build
    p->max_triples = max_triples; p->U = U; p->I = I;
    *out = p;
    return DAISY_OK;
}

// buffers of the sorted layout (kind 0)
static int plan_need_sorted(daisy_epoch_plan *p) {
    if (p->arena) return DAISY_OK;
    const int64_t max_triples = p->max_triples;
    const size_t n2 = 2 * (size_t)max_triples;
    const size_t ta = sort_pairs_u32_u64_temp_bytes(n2), tb = sort_pairs_u64_u64_temp_bytes(n2);
    const size_t tc = rle_u32_temp_bytes(n2), td = rle_u64_temp_bytes(n2);
    p->temp_bytes = ta > tb ? ta : tb;
    if (tc > p->temp_bytes) p->temp_bytes = tc;
    if (td > p->temp_bytes) p->temp_bytes = td;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    size_t o_k32[2], o_v64[2];
    o_k32[0] = take(n2 * 4); o_k32[1] = take(n2 * 4);
    o_v64[0] = take(n2 * 8); o_v64[1] = take(n2 * 8);
    const size_t o_s32 = take((size_t)max_triples * 4);   // sorted sample keys survive the entry sort
    const size_t o_sv = take((size_t)max_triples * 8);
    const size_t o_ss = take(n2 * 4), o_sn = take(n2 * 4), o_so = take(((size_t)max_triples + 2) * 4);
    const size_t o_rt = take(256);
    const size_t o_tmp = take(p->temp_bytes);
    p->arena_bytes = off;
    hipError_t e = hipMalloc(&p->arena, p->arena_bytes);
    if (e != hipSuccess) {
        set_error("epoch_plan_build: hipMalloc(%zu) failed: %s", p->arena_bytes, hipGetErrorString(e));
        p->arena = nullptr;
        p->arena_bytes = 0;
        return DAISY_ERR_HIP;
    }
    char *b = (char *)p->arena;
    for (int k = 0; k < 2; ++k) {
        p->k32[k] = (uint32_t *)(b + o_k32[k]);
        p->v64[k] = (uint64_t *)(b + o_v64[k]);
        p->k64[k] = nullptr;                        // allocated on demand (rare: > 32 key bits)
    }
    p->ukey = (uint32_t *)(b + o_s32);
    p->uval = (uint64_t *)(b + o_sv);
    p->run_key = (uint32_t *)(b + o_ss);
    p->run_cnt = (uint32_t *)(b + o_sn);
    p->run_off = (int32_t *)(b + o_so);
    p->run_total = (uint32_t *)(b + o_rt);
    p->bad = (int *)(b + o_rt + 64);
    p->ekey = nullptr; p->eval = nullptr;
    p->umask = p->imask = 0;
    p->temp = b + o_tmp;
    return DAISY_OK;
}

static int plan_free(daisy_epoch_plan *p) {
    hipError_t e = p->arena ? hipFree(p->arena) : hipSuccess;
    for (int k = 0; k < 2; ++k)
        if (p->k64[k]) (void)hipFree(p->k64[k]);
    if (p->parena) (void)hipFree(p->parena);
    if (p->parena2) (void)hipFree(p->parena2);
    if (p->d_off) (void)hipFree(p->d_off);
    free(p->h_off);
    delete p;
    if (e != hipSuccess) {
        set_error("epoch_plan_destroy: hipFree failed: %s", hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    return DAISY_OK;
}

static int plan_need_k64(daisy_epoch_plan *p) {
    for (int k = 0; k < 2; ++k) {
        if (!p->k64[k]) {
            hipError_t e = hipMalloc((void **)&p->k64[k], 2 * (size_t)p->max_triples * 8);
            if (e != hipSuccess) {
                set_error("epoch_plan_build: hipMalloc of 64-bit key buffers failed: %s", hipGetErrorString(e));
                return DAISY_ERR_HIP;
            }
        }
    }
    return DAISY_OK;
}

// flags: DAISY_PLAN_TRIPLES_USER_SORTED -> the samples only need a stable partition by batch
// perm_limit: rows of `triples` a permutation entry may name (n for a permutation of the n rows; the whole array when the
// entries SELECT n of its rows: daisy_bpr_set_batch_from_triples - whose range check compared against n until round 4, so
// that a selection naming a row >= its own length was refused)
static int plan_build(daisy_epoch_plan *p, const int32_t *triples, int64_t n, int64_t start,
                      const int64_t *perm, int order_mode, uint64_t seed, uint64_t epoch,
                      int64_t batch_size, int32_t user_base, int32_t flags, hipStream_t s, int64_t perm_limit = -1) {
    if (perm_limit < 0) perm_limit = n;
    const int ubits = bits_for(p->U), ibits = bits_for(p->I);
    const int64_t nb = (n + batch_size - 1) / batch_size;
    const int bbits = (nb > 1) ? bits_for(nb) : 0;
    const uint32_t umask = (uint32_t)(((uint64_t)1 << ubits) - 1);
    const int ibits1 = ibits + 1;                    // item << 1 | negative-slot bit
    const uint32_t imask = (uint32_t)(((uint64_t)1 << ibits1) - 1);
    const bool wide = (ubits + bbits > 32) || (ibits1 + bbits > 32);
    const bool presorted = (flags & DAISY_PLAN_TRIPLES_USER_SORTED) && order_mode != DAISY_ORDER_PERM;
    const int pointwise = (flags & DAISY_PLAN_POINTWISE) ? 1 : 0;
    int rc = plan_need_sorted(p);
    if (rc) return rc;
    const int s_begin = presorted ? ubits : 0;      // user bits ride along unsorted
    FeistelKey fk = make_feistel_key((uint64_t)n, seed, epoch);
    const int g1 = grid_for(n, kBlock), g2 = grid_for(2 * n, kBlock);
    DAISY_HIP(hipMemsetAsync(p->bad, 0, sizeof(int), s));
    if (!wide) {
        hipLaunchKernelGGL((k_plan_keys<uint32_t>), dim3(g1), dim3(kBlock), 0, s, triples, perm,
                           order_mode, fk, n, start, batch_size, user_base, ubits, p->U, p->I, pointwise, p->bad,
                           p->k32[0], p->v64[0], perm_limit);
        DAISY_LAUNCH_CHECK();
        if (ubits + bbits > s_begin) {
            rc = sort_pairs_u32_u64(p->temp, p->temp_bytes, p->k32[0], p->ukey, p->v64[0], p->uval, n,
                                    s_begin, ubits + bbits, s);
            if (rc) return rc;
        } else {   // one batch of user-sorted triples: already in plan order
            DAISY_HIP(hipMemcpyAsync(p->ukey, p->k32[0], n * 4, hipMemcpyDeviceToDevice, s));
            DAISY_HIP(hipMemcpyAsync(p->uval, p->v64[0], n * 8, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL((k_plan_entries<uint32_t>), dim3(g1), dim3(kBlock), 0, s, p->ukey, p->uval,
                           n, batch_size, ibits, umask, pointwise, p->k32[0], p->v64[0]);
        DAISY_LAUNCH_CHECK();
        rc = sort_pairs_u32_u64(p->temp, p->temp_bytes, p->k32[0], p->k32[1], p->v64[0], p->v64[1], 2 * n,
                                0, ibits1 + bbits, s);
        if (rc) return rc;
        // runs of equal (batch, item, slot) keys: the distinct items of every batch + their counts
        rc = rle_u32(p->temp, p->temp_bytes, p->k32[1], 2 * n, p->k32[0], p->run_cnt, p->run_total, s);
        if (rc) return rc;
        hipLaunchKernelGGL((k_run_finish<uint32_t>), dim3(g2), dim3(kBlock), 0, s, p->k32[0], p->run_total,
                           nb, ibits1, p->run_key, p->run_off);
        DAISY_LAUNCH_CHECK();
        p->umask = umask;
        p->imask = imask;
    } else {
        if ((rc = plan_need_k64(p))) return rc;
        hipLaunchKernelGGL((k_plan_keys<uint64_t>), dim3(g1), dim3(kBlock), 0, s, triples, perm,
                           order_mode, fk, n, start, batch_size, user_base, ubits, p->U, p->I, pointwise, p->bad,
                           p->k64[0], p->v64[0], perm_limit);
        DAISY_LAUNCH_CHECK();
        rc = sort_pairs_u64_u64(p->temp, p->temp_bytes, p->k64[0], p->k64[1], p->v64[0], p->uval, n,
                                s_begin, ubits + bbits, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_narrow_keys, dim3(g1), dim3(kBlock), 0, s, p->k64[1], n, (uint64_t)umask,
                           p->ukey);
        DAISY_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_plan_entries<uint64_t>), dim3(g1), dim3(kBlock), 0, s, p->k64[1], p->uval,
                           n, batch_size, ibits, 0xFFFFFFFFu & umask, pointwise, p->k64[0], p->v64[0]);
        DAISY_LAUNCH_CHECK();
        rc = sort_pairs_u64_u64(p->temp, p->temp_bytes, p->k64[0], p->k64[1], p->v64[0], p->v64[1], 2 * n,
                                0, ibits1 + bbits, s);
        if (rc) return rc;
        rc = rle_u64(p->temp, p->temp_bytes, p->k64[1], 2 * n, p->k64[0], p->run_cnt, p->run_total, s);
        if (rc) return rc;
        hipLaunchKernelGGL((k_run_finish<uint64_t>), dim3(g2), dim3(kBlock), 0, s, p->k64[0], p->run_total,
                           nb, ibits1, p->run_key, p->run_off);
        DAISY_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_narrow_keys, dim3(g2), dim3(kBlock), 0, s, p->k64[1], 2 * n, (uint64_t)imask,
                           p->k32[1]);
        DAISY_LAUNCH_CHECK();
        p->umask = 0xFFFFFFFFu;
        p->imask = 0xFFFFFFFFu;
    }
    p->ekey = p->k32[1];
    p->eval = p->v64[1];
    free(p->h_off); p->h_off = nullptr; p->h_off_cap = 0;          // (a plan that held a rank's share before)
    if (p->d_off) { (void)hipFree(p->d_off); p->d_off = nullptr; }
    p->n = n; p->batch_size = batch_size; p->num_batches = nb; p->built = true;
    p->build_gen = next_plan_build_id();
    p->pointwise = pointwise;
    p->kind = 0;
    return DAISY_OK;
}

static BatchView plan_view(const daisy_epoch_plan *p, int64_t k) {
    const int64_t lo = k * p->batch_size;
    BatchView v;
    v.B = (p->n - lo < p->batch_size) ? (p->n - lo) : p->batch_size;
    v.ukey = p->ukey + lo;
    v.ij = reinterpret_cast<const int2 *>(p->uval + lo);
    v.ekey = p->ekey + 2 * lo;
    v.esu = reinterpret_cast<const uint2 *>(p->eval + 2 * lo);
    v.run_key = p->run_key;
    v.run_cnt = p->run_cnt;
    v.run_off = p->run_off + k;
    v.umask = p->umask;
    v.imask = p->imask;
    v.pointwise = p->pointwise;
    v.bu = v.bi = v.b0 = v.g_bu = v.g_bi = v.g_b0 = nullptr;
    v.halt = nullptr;
    return v;
}

int launch_reduce_partials(const double *partials, int nblocks, double *stats, bool finalize, float reg_1,
                           float reg_2, double *epoch_acc, double *step_loss, hipStream_t s) {
    if (finalize)
        hipLaunchKernelGGL((k_reduce_partials<true>), dim3(1), dim3(kBlock), 0, s, partials, nblocks, stats, reg_1,
                           reg_2, epoch_acc, step_loss);
    else
        hipLaunchKernelGGL((k_reduce_partials<false>), dim3(1), dim3(kBlock), 0, s, partials, nblocks, stats, 0.f,
                           0.f, (double *)nullptr, (double *)nullptr);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

// the same batch as the staged step reads it (stage slot = grouped sample position)
static StreamView stream_view_of(const BatchView &v) {
    StreamView sv;
    sv.s_rec = nullptr; sv.s_user = v.ukey; sv.s_ij = v.ij;
    sv.e_key = v.ekey; sv.e_kstride = 1; sv.e_pos = reinterpret_cast<const uint32_t *>(v.esu); sv.e_stride = 2;
    sv.umask = v.umask; sv.imask = v.imask; sv.pos_base = 0;
    sv.B = v.B; sv.E = 2 * v.B;
    sv.halt = nullptr;
    sv.pointwise = v.pointwise;
    sv.p_stream = 0;
    return sv;
}

}  // namespace daisy

using namespace daisy;

// out[row] = sum over the row's entries of coef[e].x * X[col_e]  (a sparse-matrix x dense-matrix product
// for entries sorted by row): the item pass's segmented reduction (bitwise reproducible variant) on a
// synthetic view - ekey[e] = row << 1, esu[e] = (e, col), coef[e] = (value, 0).  Rows without entries are
// not written; n_entries must be even.  `edges` is scratch for segsum_chunks(n_entries, d) chunks.
namespace daisy {
int64_t segsum_chunks(int64_t n_entries, int d) {
    int64_t nchunks = 0;
    (void)dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        nchunks = (n_entries + RunCfg<C>::E - 1) / RunCfg<C>::E;
        return DAISY_OK;
    });
    return nchunks;
}

int segsum_rows(const float *X, const float2 *coef, const uint32_t *ekey, const uint2 *esu, int64_t n_entries,
                int d, float *out, float *edge_vec, int32_t *edge_item, float *edge_b, int32_t *edge_whole,
                hipStream_t s, bool x_bf16) {
    if (n_entries <= 0) return DAISY_OK;
    if (n_entries & 1) { set_error("segsum_rows: odd entry count %lld", (long long)n_entries); return DAISY_ERR_ARG; }
    BatchView v{};
    v.ekey = ekey;
    v.esu = esu;
    v.imask = 0xFFFFFFFFu;
    v.B = n_entries / 2;
    ItemEdges ed{edge_vec, edge_item, edge_b, edge_whole};
    static const int tune_run = getenv("DAISY_SEGSUM_RUN") ? atoi(getenv("DAISY_SEGSUM_RUN")) : 4;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        auto go = [&](auto ro_tag) {
            // wide rows (d > 128: 16 floats per lane) default to 2 rows in flight per lane group; 4 measured faster
            // for the NeuMF tables (fewer, longer chunks: 3 barriers per chunk)
            constexpr int RO = decltype(ro_tag)::value;
            const int64_t nchunks = (n_entries + RunCfg<C, RO>::E - 1) / RunCfg<C, RO>::E;
            const dim3 g(grid_for(n_entries, RunCfg<C, RO>::E, 16384)), ge(grid_for(nchunks, C::GROUPS_PER_BLOCK));
            if (x_bf16) hipLaunchKernelGGL((k_item_grad_chunked<C, RO, true, true>), g, dim3(kBlock), 0, s, X, coef, v, d, out, ed);
            else hipLaunchKernelGGL((k_item_grad_chunked<C, RO, true>), g, dim3(kBlock), 0, s, X, coef, v, d, out, ed);
            hipLaunchKernelGGL((k_item_edges<C>), ge, dim3(kBlock), 0, s, ed, nchunks, d, out, (float *)nullptr);
        };
        if (C::NE == 16 && tune_run == 4) go(std::integral_constant<int, 4>{});
        else go(std::integral_constant<int, 0>{});
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}
}  // namespace daisy

static void view_bias(daisy_bpr_ctx *ctx) {
    BatchView &v = ctx->v;
    v.bu = ctx->bu; v.bi = ctx->bi; v.b0 = ctx->b0;
    v.g_bu = ctx->g_bu; v.g_bi = ctx->g_bi; v.g_b0 = ctx->g_b0;
}

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int daisy_epoch_plan_create(daisy_epoch_plan **out, int64_t max_triples, int64_t user_num,
                            int64_t item_num) {
    DAISY_CHECK_ARG(out != nullptr, "epoch_plan_create: out is NULL");
    DAISY_CHECK_ARG(max_triples > 0 && max_triples < ((int64_t)1 << 30),
                    "epoch_plan_create: max_triples=%lld out of range", (long long)max_triples);
    DAISY_CHECK_ARG(user_num > 0 && user_num <= INT32_MAX && item_num > 0 && item_num <= INT32_MAX,
                    "epoch_plan_create: user_num/item_num out of int32 range");
    return plan_alloc(out, max_triples, user_num, item_num);
}

int daisy_epoch_plan_destroy(daisy_epoch_plan *plan) {
    if (!plan) return DAISY_OK;
    return plan_free(plan);
}

size_t daisy_epoch_plan_bytes(const daisy_epoch_plan *plan) {
    return plan ? plan->arena_bytes + plan->parena_bytes : 0;
}

int64_t daisy_epoch_plan_num_batches(const daisy_epoch_plan *plan) {
    return (plan && plan->built) ? plan->num_batches : 0;
}

int daisy_epoch_plan_build(daisy_epoch_plan *plan, const int32_t *triples, int64_t n_triples,
                           const int64_t *perm, int32_t order_mode, uint64_t seed, uint64_t epoch,
                           int64_t batch_size, int32_t user_base, int32_t flags,
                           daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan && triples, "epoch_plan_build: NULL argument");
    DAISY_CHECK_ARG(n_triples > 0 && n_triples <= plan->max_triples,
                    "epoch_plan_build: n_triples=%lld not in 1..%lld", (long long)n_triples,
                    (long long)plan->max_triples);
    DAISY_CHECK_ARG(batch_size > 0, "epoch_plan_build: batch_size must be positive");
    DAISY_CHECK_ARG(order_mode >= DAISY_ORDER_IDENTITY && order_mode <= DAISY_ORDER_FEISTEL,
                    "epoch_plan_build: bad order_mode %d", order_mode);
    DAISY_CHECK_ARG(order_mode != DAISY_ORDER_PERM || perm != nullptr,
                    "epoch_plan_build: DAISY_ORDER_PERM needs perm");
    return plan_build(plan, triples, n_triples, 0, perm, order_mode, seed, epoch, batch_size, user_base,
                      flags, S(stream));
}

static int report_bad_ids(const int *bad_dev, const char *who, int64_t U, int64_t I, hipStream_t s) {
    int bad = 0;
    DAISY_HIP(hipMemcpyAsync(&bad, bad_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    DAISY_HIP(hipStreamSynchronize(s));
    if (bad & 2) { set_error("%s: index out of range in the epoch permutation", who); return DAISY_ERR_ARG; }
    if (bad & 1) {
        set_error("%s: index out of range in the batch: need 0 <= user - user_base < %lld and 0 <= item < %lld "
                  "(the reference raises IndexError in nn.Embedding, MFRecommender.py:64-65)", who, (long long)U,
                  (long long)I);
        return DAISY_ERR_ARG;
    }
    return DAISY_OK;
}

int daisy_epoch_plan_validate(const daisy_epoch_plan *plan, daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan != nullptr, "epoch_plan_validate: NULL plan");
    if (!plan->built) { set_error("epoch_plan_validate: plan has not been built"); return DAISY_ERR_STATE; }
    if (plan->kind == 1) return DAISY_OK;            // a train index is validated when it is created
    return report_bad_ids(plan->bad, "epoch_plan_build", plan->U, plan->I, S(stream));
}

int daisy_bpr_ctx_validate_batch(const daisy_bpr_ctx *ctx, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx != nullptr, "ctx_validate_batch: NULL context");
    if (!ctx->batch_set) { set_error("ctx_validate_batch: no batch set"); return DAISY_ERR_STATE; }
    if (!ctx->own_plan || !ctx->own_plan->built || ctx->v.ukey != ctx->own_plan->ukey) return DAISY_OK;   // from an epoch plan
    return report_bad_ids(ctx->own_plan->bad, "set_batch", ctx->U, ctx->I, S(stream));
}

int daisy_epoch_plan_read_batch(const daisy_epoch_plan *plan, int64_t k, int32_t *u, int32_t *i,
                                int32_t *j, int32_t *ent_item, uint32_t *ent_s, int32_t *ent_u,
                                int64_t *B_out_host, daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan && u && i && j, "epoch_plan_read_batch: NULL argument");
    if (!plan->built) { set_error("epoch_plan_read_batch: plan has not been built"); return DAISY_ERR_STATE; }
    DAISY_CHECK_ARG(k >= 0 && k < plan->num_batches, "epoch_plan_read_batch: batch %lld not in 0..%lld",
                    (long long)k, (long long)plan->num_batches);
    if (plan->kind == 1) return plan_read_batch_partitioned(plan, k, u, i, j, ent_item, ent_s, ent_u, B_out_host, S(stream));
    const BatchView v = plan_view(plan, k);
    hipLaunchKernelGGL(k_unpack_batch, dim3(grid_for(2 * v.B, kBlock)), dim3(kBlock), 0, S(stream), v, u, i,
                       j, ent_item, ent_s, ent_u);
    DAISY_LAUNCH_CHECK();
    if (B_out_host) *B_out_host = v.B;
    return DAISY_OK;
}

int daisy_feistel_positions(int64_t n, uint64_t seed, uint64_t epoch, int64_t *out,
                            daisy_stream_t stream) {
    DAISY_CHECK_ARG(out && n > 0 && n <= ((int64_t)1 << 30), "feistel_positions: n must be in 1..2^30");
    FeistelKey fk = make_feistel_key((uint64_t)n, seed, epoch);
    hipLaunchKernelGGL(k_feistel_perm, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, S(stream), n, fk, out);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_feistel_positions_at(const int64_t *ids, int64_t n_ids, int64_t n, uint64_t seed, uint64_t epoch,
                               int64_t *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ids && out && n_ids > 0 && n > 0 && n <= ((int64_t)1 << 30), "feistel_positions_at: bad argument");
    FeistelKey fk = make_feistel_key((uint64_t)n, seed, epoch);
    hipLaunchKernelGGL(k_feistel_at, dim3(grid_for(n_ids, kBlock)), dim3(kBlock), 0, S(stream), ids, n_ids, n, fk, out);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_ctx_create(daisy_bpr_ctx **out, int64_t max_batch, int32_t d, int64_t user_num,
                         int64_t item_num) {
    DAISY_CHECK_ARG(out != nullptr, "ctx_create: out is NULL");
    DAISY_CHECK_ARG(max_batch > 0 && max_batch < ((int64_t)1 << 30), "ctx_create: max_batch=%lld out of range",
                    (long long)max_batch);
    DAISY_CHECK_ARG(d > 0 && d <= kMaxD, "ctx_create: unsupported d=%d", d);
    DAISY_CHECK_ARG(user_num > 0 && user_num <= INT32_MAX && item_num > 0 && item_num <= INT32_MAX,
                    "ctx_create: user_num/item_num out of int32 range");
    daisy_bpr_ctx *c = new daisy_bpr_ctx();
    c->max_batch = max_batch; c->d = d; c->U = user_num; c->I = item_num;
    c->batch_set = false; c->fwd_done = false; c->own_plan = nullptr;
    c->last_item_mode = DAISY_ITEM_SORTED;
    c->pointwise = 0;
    c->bu = c->bi = c->b0 = c->g_bu = c->g_bi = c->g_b0 = nullptr;
    c->cur_plan = c->pre_plan = nullptr; c->cur_k = c->pre_k = -1; c->cur_gen = c->pre_gen = 0;
    c->pre_P = nullptr; c->pre_stats = nullptr; c->pre_n = 0; c->pre_ready = false;
    c->p_stream_mode = -1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_coef = take((size_t)max_batch * 8);
    const size_t o_part = take(((size_t)kMaxGrid * 8 + kPreBlocks) * 8);
    const size_t o_tt = take((size_t)max_batch * 12);
    // edge records: two per chunk of the user pass (B samples) or of the item pass (2B entries)
    size_t max_chunks = 0;
    (void)dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        const size_t cu = ((size_t)max_batch + UserRunCfg<C>::E - 1) / UserRunCfg<C>::E;
        const size_t ci = (2 * (size_t)max_batch + RunCfg<C>::E - 1) / RunCfg<C>::E;
        // staged passes: the item pass runs 128-thread workgroups (kStagedItemBlock) over 2 B entries, and its sparse
        // flavour keeps 4 (rows of <= 8 floats per lane) or 2 (wider rows) entries per lane group in flight: that is
        // its smallest chunk (StagedItemCfg<C, 128, true>::E)
        const size_t e_min = (size_t)(128 / C::LPR) * (C::NE <= 8 ? 4 : 2);
        const size_t cs = (2 * (size_t)max_batch + e_min - 1) / e_min;
        max_chunks = (cu > ci ? cu : ci);
        if (cs > max_chunks) max_chunks = cs;
        max_chunks += 2;
        return DAISY_OK;
    });
    const size_t n_edge = 2 * max_chunks;
    c->edge_chunks = (int64_t)max_chunks;
    const size_t o_ev = take(n_edge * (size_t)d * 4), o_eu = take(n_edge * 4), o_en = take(n_edge * 8);
    const size_t o_ew = take(n_edge * 4);
    const size_t o_ec = take(n_edge * 16);        // staged item pass: (n_pos, n_neg, coefficient sum, -) per edge record
    const size_t n_eb = max_chunks / kEdgeBlock + 2;      // block sums of long edge chains (k_staged_item_edge_blocks)
    const size_t o_ebv = take(n_eb * (size_t)d * 4), o_ebi = take(n_eb * 4), o_ebc = take(n_eb * 16), o_ebt = take(n_eb * 4);
    // the item pass's own edge records in the three-launch form (its chunks are at least 2 entries x 16 lane groups)
    const size_t merge_b = (size_t)(max_batch < kMergeMaxBatch ? max_batch : kMergeMaxBatch);
    const size_t chunks2 = 2 * merge_b / 32 + 2, n_edge2 = 2 * chunks2;
    const size_t o_e2v = take(n_edge2 * (size_t)d * 4), o_e2i = take(n_edge2 * 4), o_e2c = take(n_edge2 * 16);
    const size_t o_e2w = take(n_edge2 * 4);
    const size_t o_sr = take((kMaxItemSlices + 1) * 8);
    const size_t o_ps = take((size_t)max_batch * (size_t)d * 4);
    const size_t o_pn = take((size_t)user_num * 4);
    c->arena_bytes = off;
    hipError_t e = hipMalloc(&c->arena, c->arena_bytes);
    if (e != hipSuccess) {
        set_error("ctx_create: hipMalloc(%zu) failed: %s", c->arena_bytes, hipGetErrorString(e));
        delete c;
        return DAISY_ERR_HIP;
    }
    char *base = (char *)c->arena;
    c->coef = (float2 *)(base + o_coef);
    c->partials = (double *)(base + o_part);
    c->tmp_triples = (int32_t *)(base + o_tt);
    c->edge_vec = (float *)(base + o_ev);
    c->edge_user = (int32_t *)(base + o_eu);
    c->edge_n = (float *)(base + o_en);
    c->edge_whole = (int32_t *)(base + o_ew);
    c->edge_cnt = (float *)(base + o_ec);
    c->eb_vec = (float *)(base + o_ebv); c->eb_item = (int32_t *)(base + o_ebi);
    c->eb_cnt = (float *)(base + o_ebc); c->eb_through = (int32_t *)(base + o_ebt);
    c->eb_blocks = (int64_t)n_eb;
    c->edge2_vec = (float *)(base + o_e2v); c->edge2_item = (int32_t *)(base + o_e2i);
    c->edge2_cnt = (float *)(base + o_e2c); c->edge2_whole = (int32_t *)(base + o_e2w);
    c->edge2_chunks = (int64_t)chunks2;
    c->slice_rng = (int64_t *)(base + o_sr);
    c->n_slices = 0;
    c->batch_kind = 0;
    c->p_stage = (float *)(base + o_ps);
    c->p_sqnorm = (float *)(base + o_pn);
    c->p_sqnorm_of = nullptr;
    *out = c;
    return DAISY_OK;
}

int daisy_bpr_ctx_destroy(daisy_bpr_ctx *ctx) {
    if (!ctx) return DAISY_OK;
    int rc = DAISY_OK;
    if (ctx->own_plan) rc = plan_free(ctx->own_plan);
    hipError_t e = hipFree(ctx->arena);
    delete ctx;
    if (e != hipSuccess) {
        set_error("ctx_destroy: hipFree failed: %s", hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    return rc;
}

size_t daisy_bpr_ctx_scratch_bytes(const daisy_bpr_ctx *ctx) {
    if (!ctx) return 0;
    return ctx->arena_bytes + (ctx->own_plan ? ctx->own_plan->arena_bytes : 0);
}

static int ensure_own_plan(daisy_bpr_ctx *ctx) {
    if (ctx->own_plan) return DAISY_OK;
    return plan_alloc(&ctx->own_plan, ctx->max_batch, ctx->U, ctx->I);
}

int daisy_bpr_ctx_set_pointwise(daisy_bpr_ctx *ctx, int32_t pointwise) {
    DAISY_CHECK_ARG(ctx != nullptr, "ctx_set_pointwise: NULL argument");
    ctx->pointwise = pointwise ? 1 : 0;
    return DAISY_OK;
}

int daisy_bpr_ctx_set_bias(daisy_bpr_ctx *ctx, float *u_bias, float *i_bias, float *bias,
                           float *g_u_bias, float *g_i_bias, float *g_bias) {
    DAISY_CHECK_ARG(ctx != nullptr, "ctx_set_bias: NULL context");
    if (u_bias == nullptr) {                  // detach: plain MF again
        ctx->bu = ctx->bi = ctx->b0 = ctx->g_bu = ctx->g_bi = ctx->g_b0 = nullptr;
    } else {
        DAISY_CHECK_ARG(i_bias && bias && g_i_bias,
                        "ctx_set_bias: u_bias, i_bias, bias and g_i_bias must all be given");
        ctx->bu = u_bias; ctx->bi = i_bias; ctx->b0 = bias;
        ctx->g_bu = g_u_bias; ctx->g_bi = g_i_bias; ctx->g_b0 = g_bias;
    }
    if (ctx->batch_set) view_bias(ctx);
    return DAISY_OK;
}

int daisy_bpr_set_batch_from_plan(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int64_t k,
                                  daisy_stream_t stream) {
    (void)stream;
    DAISY_CHECK_ARG(ctx && plan, "set_batch_from_plan: NULL argument");
    if (!plan->built) { set_error("set_batch_from_plan: plan has not been built"); return DAISY_ERR_STATE; }
    DAISY_CHECK_ARG(k >= 0 && k < plan->num_batches, "set_batch_from_plan: batch %lld not in 0..%lld",
                    (long long)k, (long long)plan->num_batches);
    DAISY_CHECK_ARG(plan->batch_size <= ctx->max_batch && plan->U == ctx->U && plan->I == ctx->I,
                    "set_batch_from_plan: plan (batch %lld, U %lld, I %lld) does not fit the context",
                    (long long)plan->batch_size, (long long)plan->U, (long long)plan->I);
    if (plan->kind == 1) {            // partitioned layout: only the staged step can read it
        if (daisy_epoch_plan_batch_rows(plan, k) == 0) {
            set_error("set_batch_from_plan: batch %lld holds no rows of this plan (a rank's share of the epoch: "
                      "check daisy_epoch_plan_batch_rows first)", (long long)k);
            return DAISY_ERR_STATE;
        }
        ctx->sv = plan_stream_view(plan, k);
        memset(&ctx->v, 0, sizeof(ctx->v));
        ctx->v.B = ctx->sv.B;
        ctx->batch_kind = 1;
        ctx->cur_plan = plan; ctx->cur_k = k; ctx->cur_gen = plan->build_gen;
    } else {
        ctx->v = plan_view(plan, k);
        ctx->sv = stream_view_of(ctx->v);
        ctx->batch_kind = 0;
        view_bias(ctx);
        ctx->cur_plan = nullptr;
    }
    ctx->batch_set = true; ctx->fwd_done = false; ctx->n_slices = 0;
    return DAISY_OK;
}

int daisy_bpr_set_batch_from_triples(daisy_bpr_ctx *ctx, const int32_t *triples, int64_t n_triples,
                                     const int64_t *idx, int64_t start, int64_t B, int32_t user_base,
                                     daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && triples, "set_batch_from_triples: NULL argument");
    DAISY_CHECK_ARG(B > 0 && B <= ctx->max_batch, "set_batch_from_triples: B=%lld not in 1..%lld",
                    (long long)B, (long long)ctx->max_batch);
    DAISY_CHECK_ARG(idx || (start >= 0 && start + B <= n_triples),
                    "set_batch_from_triples: rows %lld..%lld outside 0..%lld", (long long)start,
                    (long long)(start + B), (long long)n_triples);
    int rc = ensure_own_plan(ctx);
    if (rc) return rc;
    // a one-batch plan over the selected rows
    rc = plan_build(ctx->own_plan, triples, B, idx ? 0 : start, idx, idx ? DAISY_ORDER_PERM : DAISY_ORDER_IDENTITY,
                    0, 0, B, user_base, ctx->pointwise ? DAISY_PLAN_POINTWISE : 0, S(stream), idx ? n_triples : -1);
    if (rc) return rc;
    ctx->v = plan_view(ctx->own_plan, 0);
    ctx->sv = stream_view_of(ctx->v);
    ctx->batch_kind = 0;
    view_bias(ctx);
    ctx->cur_plan = nullptr;
    ctx->batch_set = true; ctx->fwd_done = false; ctx->n_slices = 0;
    return DAISY_OK;
}

int daisy_bpr_set_batch(daisy_bpr_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j,
                        int64_t B, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && u && i && j, "set_batch: NULL argument");
    DAISY_CHECK_ARG(B > 0 && B <= ctx->max_batch, "set_batch: B=%lld not in 1..%lld", (long long)B,
                    (long long)ctx->max_batch);
    hipStream_t s = S(stream);
    hipLaunchKernelGGL(k_pack_triples, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, u, i, j, B,
                       ctx->tmp_triples);
    DAISY_LAUNCH_CHECK();
    return daisy_bpr_set_batch_from_triples(ctx, ctx->tmp_triples, B, nullptr, 0, B, 0, stream);
}

static int forward_impl(daisy_bpr_ctx *ctx, const float *P, const float *Q, int32_t loss_type,
                        float gamma, double *stats, bool finalize, float reg_1, float reg_2,
                        double *epoch_acc, double *step_loss, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "forward: NULL argument");
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_SL,
                    "Invalid loss type: %d", loss_type);
    if (!ctx->batch_set) { set_error("forward: no batch set"); return DAISY_ERR_STATE; }
    if (ctx->batch_kind != 0) {
        set_error("%s: the current batch comes from a partitioned plan (daisy_epoch_plan_build_indexed), which only "
                  "daisy_bpr_sgd_step / daisy_bpr_fit_epoch_sgd with DAISY_ITEM_FUSED and the daisy_bpr_staged_* phases read", "forward");
        return DAISY_ERR_STATE;
    }
    DAISY_CHECK_ARG((loss_type >= DAISY_LOSS_CL) == (ctx->v.pointwise != 0),
                    "forward: loss type %d does not match the batch layout (point-wise=%d)", loss_type,
                    ctx->v.pointwise);
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    int grid = 0;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        grid = grid_for(v.B, C::GROUPS_PER_BLOCK * FwdCfg<C>::RUN);
        if (v.pointwise)
            hipLaunchKernelGGL((k_fwd<C, true>), dim3(grid), dim3(kBlock), 0, s, P, Q, v, d, (int)loss_type,
                               gamma, ctx->coef, ctx->partials);
        else
            hipLaunchKernelGGL((k_fwd<C, false>), dim3(grid), dim3(kBlock), 0, s, P, Q, v, d, (int)loss_type,
                               gamma, ctx->coef, ctx->partials);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    if (finalize)
        hipLaunchKernelGGL((k_reduce_partials<true>), dim3(1), dim3(kBlock), 0, s, ctx->partials, grid,
                           stats, reg_1, reg_2, epoch_acc, step_loss);
    else
        hipLaunchKernelGGL((k_reduce_partials<false>), dim3(1), dim3(kBlock), 0, s, ctx->partials, grid,
                           stats, 0.f, 0.f, nullptr, nullptr);
    DAISY_LAUNCH_CHECK();
    ctx->fwd_done = true;
    return DAISY_OK;
}

int daisy_bpr_forward(daisy_bpr_ctx *ctx, const float *P, const float *Q, int32_t loss_type,
                      float gamma, double *stats, daisy_stream_t stream) {
    return forward_impl(ctx, P, Q, loss_type, gamma, stats, false, 0.f, 0.f, nullptr, nullptr, stream);
}

int daisy_bpr_finalize(daisy_bpr_ctx *ctx, double *stats, float reg_1, float reg_2,
                       double *epoch_acc, double *step_loss, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && stats, "finalize: NULL argument");
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1), 0, S(stream), stats, reg_1, reg_2, epoch_acc,
                       step_loss);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

static int item_grad_impl(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                          float reg_1, float reg_2, float *gQ, int32_t item_mode, bool data_only,
                          daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats && gQ, "item_grad: NULL argument");
    if (item_mode == DAISY_ITEM_FUSED) item_mode = DAISY_ITEM_CHUNKED;   // phase API: same item kernel
    DAISY_CHECK_ARG(item_mode >= DAISY_ITEM_ATOMIC && item_mode <= DAISY_ITEM_CHUNKED,
                    "item_grad: bad item_mode %d", item_mode);
    ctx->last_item_mode = item_mode;
    if (!ctx->fwd_done) { set_error("item_grad: forward has not run for this batch"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    const bool reg = (reg_1 != 0.f) || (reg_2 != 0.f);
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (item_mode == DAISY_ITEM_SORTED) {
            hipLaunchKernelGGL((k_item_grad_sorted<C>),
                               dim3(grid_for(2 * v.B, C::GROUPS_PER_BLOCK, kMaxGridSparse)), dim3(kBlock), 0,
                               s, P, Q, ctx->coef, v, d, stats, reg_1, reg_2, gQ);
        } else if (item_mode == DAISY_ITEM_CHUNKED) {
            static const int tune_run = getenv("DAISY_CHUNK_RUN") ? atoi(getenv("DAISY_CHUNK_RUN")) : 0;
            static const int tune_cap = getenv("DAISY_CHUNK_GRID") ? atoi(getenv("DAISY_CHUNK_GRID")) : 16384;
            // 1 (default): bitwise reproducible variant (parked partials + edge records); 0: the variant that
            // combines shared segments with fp32 atomics (kept for A/B measurements)
            static const int tune_det = getenv("DAISY_ITEM_DET") ? atoi(getenv("DAISY_ITEM_DET")) : 1;
            if (tune_run == 4)
                hipLaunchKernelGGL((k_item_grad_chunked<C, 4>), dim3(grid_for(2 * v.B, RunCfg<C, 4>::E, tune_cap)),
                                   dim3(kBlock), 0, s, P, ctx->coef, v, d, gQ);
            else if (tune_run == 16)
                hipLaunchKernelGGL((k_item_grad_chunked<C, 16>), dim3(grid_for(2 * v.B, RunCfg<C, 16>::E, tune_cap)),
                                   dim3(kBlock), 0, s, P, ctx->coef, v, d, gQ);
            else if (tune_run == 12)
                hipLaunchKernelGGL((k_item_grad_chunked<C, 12>), dim3(grid_for(2 * v.B, RunCfg<C, 12>::E, tune_cap)),
                                   dim3(kBlock), 0, s, P, ctx->coef, v, d, gQ);
            else if (tune_det) {
                const int64_t nchunks = (2 * v.B + RunCfg<C>::E - 1) / RunCfg<C>::E;
                ItemEdges ed{ctx->edge_vec, ctx->edge_user, ctx->edge_n, ctx->edge_whole, v.halt};
                hipLaunchKernelGGL((k_item_grad_chunked<C, 0, true>), dim3(grid_for(2 * v.B, RunCfg<C>::E, tune_cap)),
                                   dim3(kBlock), 0, s, P, ctx->coef, v, d, gQ, ed);
                hipLaunchKernelGGL((k_item_edges<C>), dim3(grid_for(nchunks, C::GROUPS_PER_BLOCK)), dim3(kBlock), 0, s,
                                   ed, nchunks, d, gQ, v.g_bi);
            } else
                hipLaunchKernelGGL((k_item_grad_chunked<C>), dim3(grid_for(2 * v.B, RunCfg<C>::E, tune_cap)),
                                   dim3(kBlock), 0, s, P, ctx->coef, v, d, gQ);
            if (reg && !data_only)
                hipLaunchKernelGGL((k_item_reg<C>),
                                   dim3(grid_for(2 * v.B < ctx->I ? 2 * v.B : ctx->I, C::GROUPS_PER_BLOCK * 2)),
                                   dim3(kBlock), 0, s, Q, v, d, stats, reg_1, reg_2, gQ);
        } else if (reg) {
            hipLaunchKernelGGL((k_item_grad_atomic<C, true>), dim3(grid_for(v.B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, v, ctx->coef, d, stats, reg_1, reg_2, gQ);
        } else {
            hipLaunchKernelGGL((k_item_grad_atomic<C, false>), dim3(grid_for(v.B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, v, ctx->coef, d, stats, reg_1, reg_2, gQ);
        }
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_item_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gQ, int32_t item_mode,
                        daisy_stream_t stream) {
    return item_grad_impl(ctx, P, Q, stats, reg_1, reg_2, gQ, item_mode, false, stream);
}

int daisy_bpr_item_grad_data(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                             float *gQ, int32_t item_mode, daisy_stream_t stream) {
    if (item_mode == DAISY_ITEM_FUSED) item_mode = DAISY_ITEM_CHUNKED;
    DAISY_CHECK_ARG(item_mode == DAISY_ITEM_CHUNKED,
                    "item_grad_data: only the chunked mode splits data term and regulariser (mode %d)", item_mode);
    return item_grad_impl(ctx, P, Q, stats, 0.f, 0.f, gQ, item_mode, true, stream);
}

int daisy_bpr_item_grad_reg(daisy_bpr_ctx *ctx, const float *Q, const double *stats, float reg_1,
                            float reg_2, float *gQ, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && Q && stats && gQ, "item_grad_reg: NULL argument");
    if (!ctx->batch_set) { set_error("item_grad_reg: no batch set"); return DAISY_ERR_STATE; }
    if (ctx->batch_kind != 0) {
        set_error("%s: the current batch comes from a partitioned plan (daisy_epoch_plan_build_indexed), which only "
                  "daisy_bpr_sgd_step / daisy_bpr_fit_epoch_sgd with DAISY_ITEM_FUSED and the daisy_bpr_staged_* phases read", "item_grad_reg");
        return DAISY_ERR_STATE;
    }
    if (reg_1 == 0.f && reg_2 == 0.f) return DAISY_OK;
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_item_reg<C>),
                           dim3(grid_for(2 * v.B < ctx->I ? 2 * v.B : ctx->I, C::GROUPS_PER_BLOCK * 2)),
                           dim3(kBlock), 0, S(stream), Q, v, d, stats, reg_1, reg_2, gQ);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

static int user_pass(daisy_bpr_ctx *ctx, float *P, const float *Q, const double *stats, float lr,
                     float reg_1, float reg_2, float *gP, bool sgd, daisy_stream_t stream,
                     bool chunked = false) {
    ctx->p_sqnorm_of = nullptr;   // P rows change behind the row-norm cache of the staged step
    ctx->pre_ready = false;       // (and behind a pre-norm the staged step computed ahead)
    if (!ctx->fwd_done) { set_error("user update: forward has not run for this batch"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    static const int user_kernel = getenv("DAISY_USER_KERNEL") ? atoi(getenv("DAISY_USER_KERNEL")) : 1;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        const int grid = grid_for(v.B, C::GROUPS_PER_BLOCK * 2);
        if (sgd && chunked && user_kernel == 1 && C::NE <= 4) {
            const int64_t nchunks = (v.B + UserRunCfg<C>::E - 1) / UserRunCfg<C>::E;
            hipLaunchKernelGGL((k_user_chunked<C>), dim3(grid_for(nchunks, 1, 16384)), dim3(kBlock), 0,
                               s, P, Q, v, ctx->coef, d, stats, lr, reg_1, reg_2, ctx->edge_vec,
                               ctx->edge_user, ctx->edge_n, ctx->edge_whole);
            hipLaunchKernelGGL((k_user_edges<C>), dim3(grid_for(nchunks, C::GROUPS_PER_BLOCK)),
                               dim3(kBlock), 0, s, P, nchunks, d, stats, lr, reg_1, reg_2, ctx->edge_vec,
                               ctx->edge_user, ctx->edge_n, ctx->edge_whole, v.bu, v.halt);
        } else if (sgd)
            hipLaunchKernelGGL((k_user<C, true>), dim3(grid), dim3(kBlock), 0, s, P, Q, v, ctx->coef, d,
                               stats, lr, reg_1, reg_2, gP);
        else {
            if (v.bu && !(v.g_bu && v.g_b0)) {
                set_error("user_grad: the context has biases but no g_u_bias / g_bias outputs");
                return DAISY_ERR_ARG;
            }
            hipLaunchKernelGGL((k_user<C, false>), dim3(grid), dim3(kBlock), 0, s, P, Q, v, ctx->coef, d,
                               stats, lr, reg_1, reg_2, gP);
        }
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_user_sgd(daisy_bpr_ctx *ctx, float *P, const float *Q, const double *stats, float lr,
                       float reg_1, float reg_2, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "user_sgd: NULL argument");
    // throughput kernel when the step's item gradient was formed in throughput mode,
    // the reproducible run-owner kernel otherwise
    return user_pass(ctx, P, Q, stats, lr, reg_1, reg_2, nullptr, true, stream,
                     ctx->last_item_mode == DAISY_ITEM_CHUNKED);
}

int daisy_bpr_user_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gP, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats && gP, "user_grad: NULL argument");
    return user_pass(ctx, const_cast<float *>(P), Q, stats, 0.f, reg_1, reg_2, gP, false, stream);
}

static int item_apply_impl(daisy_bpr_ctx *ctx, float *Q, float *gQ, float lr, int32_t dense,
                           const double *stats, float reg_1, float reg_2, bool with_reg,
                           daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && Q && gQ, "item_sgd_apply: NULL argument");
    if (!dense && !ctx->batch_set) { set_error("item_sgd_apply: no batch set"); return DAISY_ERR_STATE; }
    if (!dense && ctx->batch_kind != 0) {
        set_error("item_sgd_apply: the current batch comes from a partitioned plan; only dense != 0 applies there");
        return DAISY_ERR_STATE;
    }
    hipStream_t s = S(stream);
    const int d = ctx->d;
    const BatchView &v = ctx->v;
    const int64_t n = dense ? ctx->I : (2 * v.B < ctx->I ? 2 * v.B : ctx->I);   // upper bound of rows
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        const int grid = grid_for(n, C::GROUPS_PER_BLOCK * 2);
        if (with_reg)
            hipLaunchKernelGGL((k_item_apply<C, true>), dim3(grid), dim3(kBlock), 0, s, Q, gQ, v, ctx->I, d,
                               lr, (int)dense, stats, reg_1, reg_2);
        else
            hipLaunchKernelGGL((k_item_apply<C, false>), dim3(grid), dim3(kBlock), 0, s, Q, gQ, v, ctx->I, d,
                               lr, (int)dense, stats, reg_1, reg_2);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_item_sgd_apply(daisy_bpr_ctx *ctx, float *Q, float *gQ, float lr, int32_t dense,
                             daisy_stream_t stream) {
    return item_apply_impl(ctx, Q, gQ, lr, dense, nullptr, 0.f, 0.f, false, stream);
}

int daisy_adam_dense(float *W, float *g, float *m, float *v, int64_t n, float lr, float beta1,
                     float beta2, float eps, int64_t step, daisy_stream_t stream) {
    DAISY_CHECK_ARG(W && g && m && v && n > 0 && step >= 1, "adam_dense: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam_dense, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, S(stream), W, g, m,
                       v, n, (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2));
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_adam_lazy_table(float lr, float beta1, float beta2, int64_t n_steps, float *table_host) {
    DAISY_CHECK_ARG(table_host && n_steps >= 1, "adam_lazy_table: bad argument");
    table_host[0] = table_host[1] = 0.f;                              // step 0 does not exist
    for (int64_t s = 1; s <= n_steps; ++s) {                          // the host arithmetic of daisy_adam_dense
        const double bc1 = 1.0 - pow((double)beta1, (double)s), bc2 = 1.0 - pow((double)beta2, (double)s);
        table_host[2 * s] = (float)((double)lr / bc1);
        table_host[2 * s + 1] = (float)sqrt(bc2);
    }
    return DAISY_OK;
}

static int adam_lazy_batch(daisy_bpr_ctx *ctx, bool with_g, float *P, float *gP, float *mP, float *vP, int32_t *lastP,
                           float *Q, float *gQ, float *mQ, float *vQ, int32_t *lastQ, const float *table, float beta1,
                           float beta2, float eps, int64_t step, hipStream_t s) {
    if (!ctx->batch_set || ctx->batch_kind != 0) {
        set_error("adam_lazy: needs a batch of the sorted plan layout / daisy_bpr_set_batch*");
        return DAISY_ERR_STATE;
    }
    // rows of P change behind the staged step's row-norm cache and behind a pre-norm that rode on its last item pass
    // (a caller may hand the same LazyAdam state to this path and to daisy_bpr_staged_adam_step on one context)
    ctx->p_sqnorm_of = nullptr;
    ctx->pre_ready = false;
    const BatchView &v = ctx->v;
    const AdamTable TP{P, gP, mP, vP, lastP}, TQ{Q, gQ, mQ, vQ, lastQ};
    const AdamHyper h{reinterpret_cast<const float2 *>(table), beta1, beta2, eps, (int32_t)step};
    int rc = dispatch_d(ctx->d, [&](auto cfg) {
        using C = decltype(cfg);
        const dim3 g(grid_for(v.B, C::GROUPS_PER_BLOCK, 8192));
        if (with_g) hipLaunchKernelGGL((k_adam_batch_rows<C, true>), g, dim3(kBlock), 0, s, v, ctx->d, TP, TQ, h);
        else hipLaunchKernelGGL((k_adam_batch_rows<C, false>), g, dim3(kBlock), 0, s, v, ctx->d, TP, TQ, h);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_adam_lazy_catchup(daisy_bpr_ctx *ctx, float *P, float *mP, float *vP, int32_t *lastP, float *Q, float *mQ,
                            float *vQ, int32_t *lastQ, const float *table, float beta1, float beta2, float eps,
                            int64_t step, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && mP && vP && lastP && Q && mQ && vQ && lastQ && table && step >= 1, "adam_lazy_catchup: bad argument");
    return adam_lazy_batch(ctx, false, P, nullptr, mP, vP, lastP, Q, nullptr, mQ, vQ, lastQ, table, beta1, beta2, eps, step,
                           S(stream));
}

int daisy_adam_lazy_step(daisy_bpr_ctx *ctx, float *P, float *gP, float *mP, float *vP, int32_t *lastP, float *Q,
                         float *gQ, float *mQ, float *vQ, int32_t *lastQ, const float *table, float beta1, float beta2,
                         float eps, int64_t step, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && gP && mP && vP && lastP && Q && gQ && mQ && vQ && lastQ && table && step >= 1,
                    "adam_lazy_step: bad argument");
    return adam_lazy_batch(ctx, true, P, gP, mP, vP, lastP, Q, gQ, mQ, vQ, lastQ, table, beta1, beta2, eps, step, S(stream));
}

int daisy_adam_lazy_flush(float *W, float *m, float *v, int32_t *last, int64_t rows, int32_t d, const float *table,
                          float beta1, float beta2, float eps, int64_t step, daisy_stream_t stream) {
    DAISY_CHECK_ARG(W && m && v && last && table && rows > 0 && step >= 0, "adam_lazy_flush: bad argument");
    const AdamTable T{W, nullptr, m, v, last};
    const AdamHyper h{reinterpret_cast<const float2 *>(table), beta1, beta2, eps, (int32_t)step};
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_adam_flush<C>), dim3(grid_for(rows, C::GROUPS_PER_BLOCK * 2)), dim3(kBlock), 0, S(stream), T,
                           rows, (int)d, h);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_adagrad_dense(float *W, float *g, float *state_sum, int64_t n, float lr, float eps, daisy_stream_t stream) {
    DAISY_CHECK_ARG(W && g && state_sum && n > 0, "adagrad_dense: bad argument");
    hipLaunchKernelGGL(k_adagrad_dense, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, S(stream), W, g, state_sum, n, lr,
                       eps);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_rmsprop_dense(float *W, float *g, float *square_avg, int64_t n, float lr, float alpha, float eps,
                        daisy_stream_t stream) {
    DAISY_CHECK_ARG(W && g && square_avg && n > 0, "rmsprop_dense: bad argument");
    hipLaunchKernelGGL(k_rmsprop_dense, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, S(stream), W, g, square_avg, n,
                       lr, alpha, eps);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_sgd_step(daisy_bpr_ctx *ctx, float *P, float *Q, int32_t loss_type, float gamma,
                       float lr, float reg_1, float reg_2, float *gQ, double *stats,
                       double *epoch_acc, double *step_loss, int32_t item_mode,
                       daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "sgd_step: NULL argument");
    if (!ctx->batch_set) { set_error("sgd_step: no batch set"); return DAISY_ERR_STATE; }
    // with an epoch accumulator the step belongs to an epoch loop: it does nothing once a step of the epoch has had a
    // non-finite loss (AbstractRecommender.py:122-123 raises before that step's backward; here the phase kernels stop
    // exactly there, the staged step has already moved the user rows of the offending step when its loss is known)
    struct HaltScope {
        daisy_bpr_ctx *c;
        HaltScope(daisy_bpr_ctx *c_, const double *h) : c(c_) { c->v.halt = h; c->sv.halt = h; }
        ~HaltScope() { c->v.halt = nullptr; c->sv.halt = nullptr; }
    } halt_scope(ctx, epoch_acc ? epoch_acc + 1 : nullptr);
    if (item_mode == DAISY_ITEM_FUSED) {
        // the staged step (bpr_staged.hip): pairwise losses without FM biases; anything else runs the phase kernels
        if (staged_supported(ctx, loss_type))
            return staged_sgd_step(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, stats, epoch_acc, step_loss,
                                   S(stream));
        item_mode = DAISY_ITEM_CHUNKED;
    }
    DAISY_CHECK_ARG(gQ != nullptr, "sgd_step: gQ is NULL");
    int rc;
    if ((rc = forward_impl(ctx, P, Q, loss_type, gamma, stats, true, reg_1, reg_2, epoch_acc, step_loss,
                           stream))) return rc;
    // throughput mode: gQ carries the data term only, the commit kernel adds the regulariser
    const bool fold_reg = (item_mode == DAISY_ITEM_CHUNKED) && (reg_1 != 0.f || reg_2 != 0.f);
    if ((rc = item_grad_impl(ctx, P, Q, stats, reg_1, reg_2, gQ, item_mode, fold_reg, stream))) return rc;
    if ((rc = user_pass(ctx, P, Q, stats, lr, reg_1, reg_2, nullptr, true, stream,
                        item_mode == DAISY_ITEM_CHUNKED))) return rc;
    if ((rc = item_apply_impl(ctx, Q, gQ, lr, 0, stats, reg_1, reg_2, fold_reg, stream))) return rc;
    return DAISY_OK;
}

int daisy_bpr_fit_epoch_sgd(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q,
                            int32_t loss_type, float gamma, float lr, float reg_1, float reg_2,
                            float *gQ, double *stats, double *epoch_acc, double *step_losses,
                            int32_t item_mode, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && plan && P && Q && stats, "fit_epoch: NULL argument");
    if (!plan->built) { set_error("fit_epoch: plan has not been built"); return DAISY_ERR_STATE; }
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_SL, "Invalid loss type: %d", loss_type);
    // batches of a few hundred samples (the reference's default 256): all steps of the epoch inside one
    // persistent workgroup - the step is bound by kernel-boundary latency there, not by bandwidth
    if ((item_mode == DAISY_ITEM_CHUNKED || item_mode == DAISY_ITEM_FUSED) && small_epoch_supported(ctx, plan, loss_type)) {
        DAISY_CHECK_ARG(plan->U == ctx->U && plan->I == ctx->I, "fit_epoch: plan does not fit the context");
        return small_fit_epoch(ctx, plan, P, Q, loss_type, gamma, lr, reg_1, reg_2, stats, epoch_acc, step_losses,
                               S(stream));
    }
    for (int64_t k = 0; k < plan->num_batches; ++k) {
        int rc = daisy_bpr_set_batch_from_plan(ctx, plan, k, stream);
        if (rc) return rc;
        rc = daisy_bpr_sgd_step(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, gQ, stats, epoch_acc,
                                step_losses ? step_losses + k : nullptr, item_mode, stream);
        if (rc) return rc;
    }
    return DAISY_OK;
}

}  // extern "C"
