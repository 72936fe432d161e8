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
// organised around OWNERSHIP instead of atomics: an EPOCH PLAN (one radix sort
// per epoch) lays the epoch out batch by batch, each batch grouped by user, plus
// a per-batch list of item entries sorted by item; the item gradient is then a
// segmented reduction (LDS accumulators per 128-entry chunk, plain stores for
// rows a chunk owns, atomics only for the <=2 rows that straddle a chunk edge).
//
// HBM/L2 view: a d=64 row is 256 B = 16 lanes x float4, one coalesced request
// per quarter wave; four rows are in flight per wave instruction.
#include "common.h"

namespace daisy {

constexpr uint32_t kNegBit = 0x80000000u;

// what the step kernels see of the current batch
struct BatchView {
    const int32_t *u, *i, *j;    // [B] grouped by user (stable)
    const int32_t *ent_item;     // [2B] item of entry q, sorted ascending (stable)
    const uint32_t *ent_s;       // [2B] sample position s | kNegBit for the negative slot
    const int32_t *ent_u;        // [2B] u[s]
    int64_t B;
};

}  // namespace daisy

// Epoch plan: the whole epoch laid out batch by batch (see header comment).
struct daisy_epoch_plan {
    int64_t max_triples, U, I;
    void *arena;
    size_t arena_bytes, temp_bytes;
    int32_t *gu, *gi, *gj;              // [n]   plan order
    int32_t *ent_item, *ent_u;          // [2n]
    uint32_t *ent_s;                    // [2n]
    uint64_t *k64a, *k64b;              // [2n]  sort keys
    int32_t *v32a, *v32b;               // [2n]  sort payloads
    void *temp;
    int64_t n, batch_size, num_batches; // current build
    bool built;
};

struct daisy_bpr_ctx {
    int64_t max_batch, U, I;
    int d;
    void *arena;
    size_t arena_bytes, bitmap_bytes;
    float2 *coef;        // (dL/dpos, dL/dneg) per sample   [max_batch]
    double *partials;    // per-workgroup sums              [kMaxGrid*8]
    uint32_t *bitmap;    // touched item rows               [ceil(I/32)]
    int32_t *tmp_triples;  // [max_batch*3] staging for daisy_bpr_set_batch
    daisy_epoch_plan *own_plan;   // 1-batch plan used by set_batch / set_batch_from_triples
    daisy::BatchView v;
    bool batch_set, fwd_done;
};

namespace daisy {

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------
// epoch plan kernels
// ---------------------------------------------------------------------------
// order_mode: 0 identity, 1 explicit permutation (perm[p] = triple at position p), 2 Feistel
__global__ void k_plan_keys(const int32_t *__restrict__ triples, const int64_t *__restrict__ perm,
                            int order_mode, FeistelKey fk, int64_t n, int64_t start, int64_t B,
                            int32_t user_base, int ubits, uint64_t *__restrict__ key,
                            int32_t *__restrict__ val) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t t, p;
        if (order_mode == 1) { p = e; t = perm[e]; }
        else if (order_mode == 2) { t = e; p = (int64_t)feistel_position((uint64_t)e, (uint64_t)n, fk); }
        else { t = e; p = e; }
        t += start;
        const uint32_t uu = (uint32_t)(triples[3 * t] - user_base);
        key[e] = ((uint64_t)(p / B) << ubits) | uu;
        val[e] = (int32_t)(t - start);
    }
}

__global__ void k_plan_gather(const int32_t *__restrict__ triples, const int32_t *__restrict__ val,
                              int64_t n, int64_t start, int64_t B, int32_t user_base, int ibits,
                              int32_t *__restrict__ gu, int32_t *__restrict__ gi,
                              int32_t *__restrict__ gj, uint64_t *__restrict__ ekey,
                              int32_t *__restrict__ eval) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t *t = triples + 3 * ((int64_t)val[p] + start);
        const int32_t uu = t[0] - user_base, ii = t[1], jj = t[2];
        gu[p] = uu; gi[p] = ii; gj[p] = jj;
        const uint64_t k = (uint64_t)(p / B);
        const uint32_t s = (uint32_t)(p - (int64_t)k * B);
        ekey[2 * p] = (k << ibits) | (uint32_t)ii;
        eval[2 * p] = (int32_t)s;
        ekey[2 * p + 1] = (k << ibits) | (uint32_t)jj;
        eval[2 * p + 1] = (int32_t)(s | kNegBit);
    }
}

__global__ void k_plan_entries(const uint64_t *__restrict__ ekey, const int32_t *__restrict__ eval,
                               const int32_t *__restrict__ gu, int64_t n2, int64_t B, int ibits,
                               int32_t *__restrict__ ent_item, uint32_t *__restrict__ ent_s,
                               int32_t *__restrict__ ent_u) {
    const uint64_t imask = ((uint64_t)1 << ibits) - 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n2;
         q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ekey[q];
        const uint32_t sv = (uint32_t)eval[q];
        const int64_t k = (int64_t)(key >> ibits);
        ent_item[q] = (int32_t)(key & imask);
        ent_s[q] = sv;
        ent_u[q] = gu[k * B + (int64_t)(sv & ~kNegBit)];
    }
}

__global__ void k_pack_triples(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                               const int32_t *__restrict__ j, int64_t B, int32_t *__restrict__ out) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        out[3 * s] = u[s]; out[3 * s + 1] = i[s]; out[3 * s + 2] = j[s];
    }
}

__global__ void k_feistel_perm(int64_t n, FeistelKey fk, int64_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        out[e] = (int64_t)feistel_position((uint64_t)e, (uint64_t)n, fk);
}

// ---------------------------------------------------------------------------
// loss coefficient (daisy/utils/loss.py)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void pair_coef(int loss_type, float pos, float neg, float gamma,
                                          float &term, float &cp, float &cn) {
    if (loss_type == DAISY_LOSS_BPR) {  // loss.py:10-13
        const float s = sigmoidf_(pos - neg);
        const float t = gamma + s;
        term = -logf(t);
        cp = -(s * (1.f - s)) / t;
        cn = -cp;
    } else if (loss_type == DAISY_LOSS_HL) {  // loss.py:20-23 (clamp passes grad at equality)
        const float m = 1.f - (pos - neg);
        term = fmaxf(m, 0.f);
        cp = (m >= 0.f) ? -1.f : 0.f;
        cn = -cp;
    } else {  // TOP1, loss.py:30-33
        const float s1 = sigmoidf_(neg - pos);
        const float s2 = sigmoidf_(neg * neg);
        term = s1 + s2;
        const float d1 = s1 * (1.f - s1);
        cp = -d1;
        cn = d1 + 2.f * neg * s2 * (1.f - s2);
    }
}

// ---------------------------------------------------------------------------
// forward: scores, coefficients, the seven batch sums
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kBlock) void k_fwd(const float *__restrict__ P,
                                                const float *__restrict__ Q,
                                                const int32_t *__restrict__ u,
                                                const int32_t *__restrict__ i,
                                                const int32_t *__restrict__ j, int64_t B, int d,
                                                int loss_type, float gamma,
                                                float2 *__restrict__ coef,
                                                double *__restrict__ partials) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < B; s += gstride) {
        const int32_t uu = u[s], ii = i[s], jj = j[s];
        Row<C> p, qi, qj;
        p.load(P + (int64_t)uu * d, lane, d);
        qi.load(Q + (int64_t)ii * d, lane, d);
        qj.load(Q + (int64_t)jj * d, lane, d);
        const float pos = row_dot<C>(p, qi);
        const float neg = row_dot<C>(p, qj);
#pragma unroll
        for (int k = 0; k < C::NE; ++k) {
            acc[1] += fabsf(p.v[k]);
            acc[2] += fabsf(qi.v[k]);
            acc[3] += fabsf(qj.v[k]);
            acc[4] = fmaf(p.v[k], p.v[k], acc[4]);
            acc[5] = fmaf(qi.v[k], qi.v[k], acc[5]);
            acc[6] = fmaf(qj.v[k], qj.v[k], acc[6]);
        }
        if (lane == 0) {
            float term, cp, cn;
            pair_coef(loss_type, pos, neg, gamma, term, cp, cn);
            coef[s] = make_float2(cp, cn);
            acc[0] += term;
        }
    }
    __shared__ double sm[kBlock / kWave][8];
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double w = wave_sum_f64((double)acc[k]);
        if ((threadIdx.x % kWave) == 0) sm[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += sm[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * 8 + threadIdx.x] = t;
    }
}

// fixed-order reduction of the per-workgroup sums -> stats[0..6]
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const double *__restrict__ partials,
                                                            int nblocks, double *__restrict__ stats) {
    __shared__ double sm[kBlock][7];
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
#pragma unroll
        for (int k = 0; k < 7; ++k) t[k] += partials[(int64_t)b * 8 + k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) sm[threadIdx.x][k] = t[k];
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 7; ++k) sm[threadIdx.x][k] += sm[threadIdx.x + off][k];
        }
        __syncthreads();
    }
    if (threadIdx.x < 7) stats[threadIdx.x] = sm[0][threadIdx.x];
}

// MFRecommender.py:88-89,94-95: loss += reg_1*(L1 terms) + reg_2*(Frobenius terms)
__global__ void k_finalize(double *__restrict__ stats, float reg_1, float reg_2,
                           double *__restrict__ epoch_acc, double *__restrict__ step_loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double nU = sqrt(stats[DAISY_ST_SQ_U]);
    const double nI = sqrt(stats[DAISY_ST_SQ_I]);
    const double nJ = sqrt(stats[DAISY_ST_SQ_J]);
    const double loss = stats[DAISY_ST_LOSS_DATA] +
                        (double)reg_1 * (stats[DAISY_ST_L1_I] + stats[DAISY_ST_L1_J]) +
                        (double)reg_2 * (nI + nJ) + (double)reg_1 * stats[DAISY_ST_L1_U] +
                        (double)reg_2 * nU;
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

__device__ __forceinline__ float inv_or_zero(double n, float reg_2) {
    return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f;  // d|X|_F/dX = 0 at X = 0 (torch)
}

// ---------------------------------------------------------------------------
// item gradient, legacy mode: one fp32 atomic row per entry (kept for A/B
// measurements; 0.3 TB/s on MI355X)
// ---------------------------------------------------------------------------
template <class C, bool REG>
__global__ __launch_bounds__(kBlock) void k_item_grad_atomic(
    const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ u,
    const int32_t *__restrict__ i, const int32_t *__restrict__ j, const float2 *__restrict__ coef,
    int64_t B, int d, const double *__restrict__ stats, float reg_1, float reg_2,
    float *__restrict__ gQ, uint32_t *__restrict__ bitmap) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rI = REG ? inv_or_zero(stats[DAISY_ST_NORM_I], reg_2) : 0.f;
    const float rJ = REG ? inv_or_zero(stats[DAISY_ST_NORM_J], reg_2) : 0.f;
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < B; s += gstride) {
        const int32_t uu = u[s], ii = i[s], jj = j[s];
        const float2 c = coef[s];
        Row<C> p, gi, gj;
        p.load(P + (int64_t)uu * d, lane, d);
        if constexpr (REG) {
            Row<C> qi, qj;
            qi.load(Q + (int64_t)ii * d, lane, d);
            qj.load(Q + (int64_t)jj * d, lane, d);
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
        gi.atomic_add_to(gQ + (int64_t)ii * d, lane, d);
        gj.atomic_add_to(gQ + (int64_t)jj * d, lane, d);
        if (lane == 0) {
            atomicOr(bitmap + (ii >> 5), 1u << (ii & 31));
            atomicOr(bitmap + (jj >> 5), 1u << (jj & 31));
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
    const int32_t *__restrict__ ent_item, const uint32_t *__restrict__ ent_s,
    const int32_t *__restrict__ ent_u, int64_t n, int d, const double *__restrict__ stats,
    float reg_1, float reg_2, float *__restrict__ gQ, uint32_t *__restrict__ bitmap) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t pos = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; pos < n; pos += gstride) {
        const int32_t r = ent_item[pos];
        if (pos > 0 && ent_item[pos - 1] == r) continue;  // not a segment head
        Row<C> acc;
        acc.zero();
        float n_pos = 0.f, n_neg = 0.f;
        for (int64_t q = pos; q < n && ent_item[q] == r; ++q) {
            const uint32_t sv = ent_s[q];
            const bool is_neg = (sv & kNegBit) != 0;
            const float2 c2 = coef[sv & ~kNegBit];
            const float c = is_neg ? c2.y : c2.x;
            n_pos += is_neg ? 0.f : 1.f;
            n_neg += is_neg ? 1.f : 0.f;
            Row<C> p;
            p.load(P + (int64_t)ent_u[q] * d, lane, d);
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
        if (lane == 0) atomicOr(bitmap + (r >> 5), 1u << (r & 31));
    }
}

// ---------------------------------------------------------------------------
// item gradient, throughput mode: segmented reduction over the item-sorted
// entries.  A workgroup takes a chunk of E consecutive entries; every lane
// group streams RUN of them (row gathers 4 deep), keeps the running sum of the
// current item in registers and adds it to that item's LDS accumulator when the
// item changes; after a barrier each accumulator is written once: plain store
// when the chunk holds the whole run of that item, fp32 atomics only for the
// first/last item when its run continues in the neighbouring chunk.
// ---------------------------------------------------------------------------
template <class C>
struct ChunkCfg {
    static constexpr int ROWF = C::NE * C::LPR;                    // padded floats per row
    static constexpr int E_RAW = 8192 / ROWF;                      // 32 KB of accumulators
    static constexpr int E = E_RAW > kBlock ? kBlock : E_RAW;      // entries per chunk
    static constexpr int RUN = E / C::GROUPS_PER_BLOCK;            // entries per lane group
    static_assert(RUN >= 1 && E == RUN * C::GROUPS_PER_BLOCK, "chunk geometry");
};

template <class C, bool REG>
__global__ __launch_bounds__(kBlock) void k_item_grad_chunked(
    const float *__restrict__ P, const float *__restrict__ Q, const float2 *__restrict__ coef,
    const int32_t *__restrict__ ent_item, const uint32_t *__restrict__ ent_s,
    const int32_t *__restrict__ ent_u, int64_t n, int d, const double *__restrict__ stats,
    float reg_1, float reg_2, float *__restrict__ gQ, uint32_t *__restrict__ bitmap) {
    using K = ChunkCfg<C>;
    constexpr int E = K::E, RUN = K::RUN, ROWF = K::ROWF;
    __shared__ float acc_lds[E * ROWF];
    __shared__ int seg_item[E];
    __shared__ int cnt_pos[E], cnt_neg[E];
    __shared__ int lidx[E];
    __shared__ int wave_tot[kBlock / kWave];

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const float rI = REG ? inv_or_zero(stats[DAISY_ST_NORM_I], reg_2) : 0.f;
    const float rJ = REG ? inv_or_zero(stats[DAISY_ST_NORM_J], reg_2) : 0.f;
    const int64_t nchunks = (n + E - 1) / E;

    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * E;
        // ---- a. local segment index of every entry (block-wide scan of head flags) -------
        int my_item = -1;
        bool head = false;
        if (tid < E && c0 + tid < n) {
            my_item = ent_item[c0 + tid];
            head = (tid == 0) || (ent_item[c0 + tid - 1] != my_item);
        }
        const unsigned long long m = __ballot(head);
        const int wl = tid % kWave, wv = tid / kWave;
        const int incl = __popcll(m & ((wl == 63) ? ~0ull : ((1ull << (wl + 1)) - 1)));
        if (wl == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wv; ++w) off += wave_tot[w];
        int nseg = 0;
        for (int w = 0; w < kBlock / kWave; ++w) nseg += wave_tot[w];
        if (tid < E && c0 + tid < n) {
            const int l = off + incl - 1;
            lidx[tid] = l;
            if (head) { seg_item[l] = my_item; cnt_pos[l] = 0; cnt_neg[l] = 0; }
        }
        for (int e = tid; e < nseg * ROWF; e += kBlock) acc_lds[e] = 0.f;
        __syncthreads();

        // ---- b. stream the entries: RUN per lane group, 4 row gathers in flight ----------
        {
            const int t0 = group * RUN;
            int cur = -1;
            Row<C> acc;
            acc.zero();
            int np = 0, nn = 0;
            auto flush = [&](int l) {
                if (l < 0) return;
                float *dst = acc_lds + l * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) atomicAdd(dst + k * C::LPR + lane, acc.v[k]);
                if (lane == 0) {
                    if (np) atomicAdd(&cnt_pos[l], np);
                    if (nn) atomicAdd(&cnt_neg[l], nn);
                }
            };
            constexpr int UNR = (RUN % 4 == 0) ? 4 : ((RUN % 2 == 0) ? 2 : 1);
            for (int r0 = 0; r0 < RUN; r0 += UNR) {
                Row<C> p[UNR];
                float c[UNR];
                bool neg[UNR], ok[UNR];
#pragma unroll
                for (int x = 0; x < UNR; ++x) {
                    const int64_t q = c0 + t0 + r0 + x;
                    ok[x] = q < n;
                    const uint32_t sv = ok[x] ? ent_s[q] : 0u;
                    neg[x] = (sv & kNegBit) != 0;
                    const float2 c2 = ok[x] ? coef[sv & ~kNegBit] : make_float2(0.f, 0.f);
                    c[x] = neg[x] ? c2.y : c2.x;
                    if (ok[x]) p[x].load(P + (int64_t)ent_u[q] * d, lane, d);
                    else p[x].zero();
                }
#pragma unroll
                for (int x = 0; x < UNR; ++x) {
                    if (!ok[x]) continue;
                    const int l = lidx[t0 + r0 + x];
                    if (l != cur) {
                        flush(cur);
                        cur = l;
                        acc.zero();
                        np = nn = 0;
                    }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) acc.v[k] = fmaf(c[x], p[x].v[k], acc.v[k]);
                    if (neg[x]) ++nn; else ++np;
                }
            }
            flush(cur);
        }
        __syncthreads();

        // ---- c. one write per item of the chunk -------------------------------------------
        const bool first_shared = (c0 > 0) && (ent_item[c0 - 1] == seg_item[0]);
        const bool last_shared = (c0 + E < n) && (ent_item[c0 + E] == seg_item[nseg - 1]);
        for (int l = group; l < nseg; l += C::GROUPS_PER_BLOCK) {
            const int r = seg_item[l];
            Row<C> g;
            const float *src = acc_lds + l * ROWF;
#pragma unroll
            for (int k = 0; k < C::NE; ++k) g.v[k] = src[k * C::LPR + lane];
            if constexpr (REG) {
                Row<C> qr;
                qr.load(Q + (int64_t)r * d, lane, d);
                const float fp = (float)cnt_pos[l], fn = (float)cnt_neg[l];
                const float w1 = reg_1 * (fp + fn), w2 = fp * rI + fn * rJ;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) g.v[k] += fmaf(w2, qr.v[k], w1 * sgn(qr.v[k]));
            }
            const bool shared = (l == 0 && first_shared) || (l == nseg - 1 && last_shared);
            if (shared) g.atomic_add_to(gQ + (int64_t)r * d, lane, d);
            else g.store(gQ + (int64_t)r * d, lane, d);
            if (lane == 0) atomicOr(bitmap + (r >> 5), 1u << (r & 31));
        }
        __syncthreads();   // LDS is reused by the next chunk
    }
}

// ---------------------------------------------------------------------------
// user rows: the batch is grouped by user, the group that sees the head of a
// user's run owns P[u]:  g = sum_b (cp*q_i + cn*q_j) + n*(reg_1*sign(p) + reg_2*p/|P[u]|_F)
//   SGD : P[u] -= lr*g  (in place, single writer)      GRAD: gP[u] = g
// ---------------------------------------------------------------------------
template <class C, bool SGD>
__global__ __launch_bounds__(kBlock) void k_user(float *__restrict__ P, const float *__restrict__ Q,
                                                 const int32_t *__restrict__ u,
                                                 const int32_t *__restrict__ i,
                                                 const int32_t *__restrict__ j,
                                                 const float2 *__restrict__ coef, int64_t B, int d,
                                                 const double *__restrict__ stats, float lr,
                                                 float reg_1, float reg_2, float *__restrict__ gP) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rU = inv_or_zero(stats[DAISY_ST_NORM_U], reg_2);
    for (int64_t pos = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; pos < B; pos += gstride) {
        const int32_t uu = u[pos];
        if (pos > 0 && u[pos - 1] == uu) continue;  // not the head of this user's run
        Row<C> p, acc;
        p.load(P + (int64_t)uu * d, lane, d);
        acc.zero();
        float n = 0.f;
        for (int64_t q = pos; q < B && u[q] == uu; ++q) {
            const float2 c = coef[q];
            Row<C> qi, qj;
            qi.load(Q + (int64_t)i[q] * d, lane, d);
            qj.load(Q + (int64_t)j[q] * d, lane, d);
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
    }
}

// ---------------------------------------------------------------------------
// commit the item rows: Q[r] -= lr*gQ[r]; gQ[r] = 0
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_apply(float *__restrict__ Q, float *__restrict__ gQ,
                                                       const uint32_t *__restrict__ bitmap, int64_t I,
                                                       int d, float lr, int dense) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < I; r += gstride) {
        if (!dense && !((bitmap[r >> 5] >> (r & 31)) & 1u)) continue;
        Row<C> g, q, z;
        g.load(gQ + r * d, lane, d);
        q.load(Q + r * d, lane, d);
        z.zero();
#pragma unroll
        for (int k = 0; k < C::NE; ++k) q.v[k] = fmaf(-lr, g.v[k], q.v[k]);
        q.store(Q + r * d, lane, d);
        z.store(gQ + r * d, lane, d);
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

// ---------------------------------------------------------------------------
// plan construction (host side)
// ---------------------------------------------------------------------------
static int plan_alloc(daisy_epoch_plan **out, int64_t max_triples, int64_t U, int64_t I) {
    daisy_epoch_plan *p = new daisy_epoch_plan();
    p->max_triples = max_triples; p->U = U; p->I = I;
    p->n = 0; p->batch_size = 0; p->num_batches = 0; p->built = false;
    const size_t n = (size_t)max_triples;
    size_t t1 = sort_pairs_u64_i32_temp_bytes(2 * max_triples);
    p->temp_bytes = t1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_gu = take(n * 4), o_gi = take(n * 4), o_gj = take(n * 4);
    const size_t o_ei = take(2 * n * 4), o_eu = take(2 * n * 4), o_es = take(2 * n * 4);
    const size_t o_ka = take(2 * n * 8), o_kb = take(2 * n * 8);
    const size_t o_va = take(2 * n * 4), o_vb = take(2 * n * 4);
    const size_t o_tmp = take(p->temp_bytes);
    p->arena_bytes = off;
    hipError_t e = hipMalloc(&p->arena, p->arena_bytes);
    if (e != hipSuccess) {
        set_error("epoch_plan_create: hipMalloc(%zu) failed: %s", p->arena_bytes, hipGetErrorString(e));
        delete p;
        return DAISY_ERR_HIP;
    }
    char *b = (char *)p->arena;
    p->gu = (int32_t *)(b + o_gu); p->gi = (int32_t *)(b + o_gi); p->gj = (int32_t *)(b + o_gj);
    p->ent_item = (int32_t *)(b + o_ei); p->ent_u = (int32_t *)(b + o_eu); p->ent_s = (uint32_t *)(b + o_es);
    p->k64a = (uint64_t *)(b + o_ka); p->k64b = (uint64_t *)(b + o_kb);
    p->v32a = (int32_t *)(b + o_va); p->v32b = (int32_t *)(b + o_vb);
    p->temp = b + o_tmp;
    *out = p;
    return DAISY_OK;
}

static int plan_build(daisy_epoch_plan *p, const int32_t *triples, int64_t n, int64_t start,
                      const int64_t *perm, int order_mode, uint64_t seed, uint64_t epoch,
                      int64_t batch_size, int32_t user_base, hipStream_t s) {
    const int ubits = bits_for(p->U), ibits = bits_for(p->I);
    const int64_t nb = (n + batch_size - 1) / batch_size;
    const int bbits = bits_for(nb);
    FeistelKey fk = make_feistel_key((uint64_t)n, seed, epoch);
    const int g1 = grid_for(n, kBlock);
    hipLaunchKernelGGL(k_plan_keys, dim3(g1), dim3(kBlock), 0, s, triples, perm, order_mode, fk, n,
                       start, batch_size, user_base, ubits, p->k64a, p->v32a);
    DAISY_LAUNCH_CHECK();
    int rc = sort_pairs_u64_i32(p->temp, p->temp_bytes, p->k64a, p->k64b, p->v32a, p->v32b, n,
                                ubits + bbits, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_plan_gather, dim3(g1), dim3(kBlock), 0, s, triples, p->v32b, n, start,
                       batch_size, user_base, ibits, p->gu, p->gi, p->gj, p->k64a, p->v32a);
    DAISY_LAUNCH_CHECK();
    rc = sort_pairs_u64_i32(p->temp, p->temp_bytes, p->k64a, p->k64b, p->v32a, p->v32b, 2 * n,
                            ibits + bbits, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_plan_entries, dim3(grid_for(2 * n, kBlock)), dim3(kBlock), 0, s, p->k64b,
                       p->v32b, p->gu, 2 * n, batch_size, ibits, p->ent_item, p->ent_s, p->ent_u);
    DAISY_LAUNCH_CHECK();
    p->n = n; p->batch_size = batch_size; p->num_batches = nb; p->built = true;
    return DAISY_OK;
}

static void view_from_plan(daisy_bpr_ctx *ctx, const daisy_epoch_plan *p, int64_t k) {
    const int64_t lo = k * p->batch_size;
    const int64_t B = (p->n - lo < p->batch_size) ? (p->n - lo) : p->batch_size;
    ctx->v.u = p->gu + lo; ctx->v.i = p->gi + lo; ctx->v.j = p->gj + lo;
    ctx->v.ent_item = p->ent_item + 2 * lo;
    ctx->v.ent_s = p->ent_s + 2 * lo;
    ctx->v.ent_u = p->ent_u + 2 * lo;
    ctx->v.B = B;
    ctx->batch_set = true; ctx->fwd_done = false;
}

}  // namespace daisy

using namespace daisy;

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
    hipError_t e = hipFree(plan->arena);
    delete plan;
    if (e != hipSuccess) {
        set_error("epoch_plan_destroy: hipFree failed: %s", hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    return DAISY_OK;
}

size_t daisy_epoch_plan_bytes(const daisy_epoch_plan *plan) { return plan ? plan->arena_bytes : 0; }

int64_t daisy_epoch_plan_num_batches(const daisy_epoch_plan *plan) {
    return (plan && plan->built) ? plan->num_batches : 0;
}

int daisy_epoch_plan_build(daisy_epoch_plan *plan, const int32_t *triples, int64_t n_triples,
                           const int64_t *perm, int32_t order_mode, uint64_t seed, uint64_t epoch,
                           int64_t batch_size, int32_t user_base, daisy_stream_t stream) {
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
                      S(stream));
}

int daisy_epoch_plan_read_batch(const daisy_epoch_plan *plan, int64_t k, int32_t *u, int32_t *i,
                                int32_t *j, int32_t *ent_item, uint32_t *ent_s, int32_t *ent_u,
                                int64_t *B_out_host, daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan && u && i && j, "epoch_plan_read_batch: NULL argument");
    if (!plan->built) { set_error("epoch_plan_read_batch: plan has not been built"); return DAISY_ERR_STATE; }
    DAISY_CHECK_ARG(k >= 0 && k < plan->num_batches, "epoch_plan_read_batch: batch %lld not in 0..%lld",
                    (long long)k, (long long)plan->num_batches);
    const int64_t lo = k * plan->batch_size;
    const int64_t B = (plan->n - lo < plan->batch_size) ? (plan->n - lo) : plan->batch_size;
    hipStream_t s = S(stream);
    DAISY_HIP(hipMemcpyAsync(u, plan->gu + lo, B * 4, hipMemcpyDeviceToDevice, s));
    DAISY_HIP(hipMemcpyAsync(i, plan->gi + lo, B * 4, hipMemcpyDeviceToDevice, s));
    DAISY_HIP(hipMemcpyAsync(j, plan->gj + lo, B * 4, hipMemcpyDeviceToDevice, s));
    if (ent_item) DAISY_HIP(hipMemcpyAsync(ent_item, plan->ent_item + 2 * lo, 2 * B * 4, hipMemcpyDeviceToDevice, s));
    if (ent_s) DAISY_HIP(hipMemcpyAsync(ent_s, plan->ent_s + 2 * lo, 2 * B * 4, hipMemcpyDeviceToDevice, s));
    if (ent_u) DAISY_HIP(hipMemcpyAsync(ent_u, plan->ent_u + 2 * lo, 2 * B * 4, hipMemcpyDeviceToDevice, s));
    if (B_out_host) *B_out_host = B;
    return DAISY_OK;
}

int daisy_feistel_positions(int64_t n, uint64_t seed, uint64_t epoch, int64_t *out,
                            daisy_stream_t stream) {
    DAISY_CHECK_ARG(out && n > 0, "feistel_positions: bad argument");
    FeistelKey fk = make_feistel_key((uint64_t)n, seed, epoch);
    hipLaunchKernelGGL(k_feistel_perm, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, S(stream), n, fk, out);
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
    c->bitmap_bytes = align_up((size_t)((item_num + 31) / 32) * 4);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_coef = take((size_t)max_batch * 8);
    const size_t o_part = take((size_t)kMaxGrid * 8 * 8);
    const size_t o_bm = take(c->bitmap_bytes);
    const size_t o_tt = take((size_t)max_batch * 12);
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
    c->bitmap = (uint32_t *)(base + o_bm);
    c->tmp_triples = (int32_t *)(base + o_tt);
    e = hipMemset(c->bitmap, 0, c->bitmap_bytes);
    if (e != hipSuccess) {
        set_error("ctx_create: hipMemset failed: %s", hipGetErrorString(e));
        (void)hipFree(c->arena);
        delete c;
        return DAISY_ERR_HIP;
    }
    *out = c;
    return DAISY_OK;
}

int daisy_bpr_ctx_destroy(daisy_bpr_ctx *ctx) {
    if (!ctx) return DAISY_OK;
    int rc = DAISY_OK;
    if (ctx->own_plan) rc = daisy_epoch_plan_destroy(ctx->own_plan);
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
    view_from_plan(ctx, plan, k);
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
                    0, 0, B, user_base, S(stream));
    if (rc) return rc;
    view_from_plan(ctx, ctx->own_plan, 0);
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

int daisy_bpr_forward(daisy_bpr_ctx *ctx, const float *P, const float *Q, int32_t loss_type,
                      float gamma, double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "forward: NULL argument");
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_TL,
                    "Invalid loss type: %d", loss_type);
    if (!ctx->batch_set) { set_error("forward: no batch set"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    int grid = 0;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        grid = grid_for(v.B, C::GROUPS_PER_BLOCK * 4);
        hipLaunchKernelGGL((k_fwd<C>), dim3(grid), dim3(kBlock), 0, s, P, Q, v.u, v.i, v.j, v.B, d,
                           (int)loss_type, gamma, ctx->coef, ctx->partials);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, s, ctx->partials, grid, stats);
    DAISY_LAUNCH_CHECK();
    ctx->fwd_done = true;
    return DAISY_OK;
}

int daisy_bpr_finalize(daisy_bpr_ctx *ctx, double *stats, float reg_1, float reg_2,
                       double *epoch_acc, double *step_loss, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && stats, "finalize: NULL argument");
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1), 0, S(stream), stats, reg_1, reg_2, epoch_acc,
                       step_loss);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_item_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gQ, int32_t item_mode,
                        daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats && gQ, "item_grad: NULL argument");
    DAISY_CHECK_ARG(item_mode >= DAISY_ITEM_ATOMIC && item_mode <= DAISY_ITEM_CHUNKED,
                    "item_grad: bad item_mode %d", item_mode);
    if (!ctx->fwd_done) { set_error("item_grad: forward has not run for this batch"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    const bool reg = (reg_1 != 0.f) || (reg_2 != 0.f);
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (item_mode == DAISY_ITEM_SORTED) {
            hipLaunchKernelGGL((k_item_grad_sorted<C>), dim3(grid_for(2 * v.B, C::GROUPS_PER_BLOCK)),
                               dim3(kBlock), 0, s, P, Q, ctx->coef, v.ent_item, v.ent_s, v.ent_u,
                               2 * v.B, d, stats, reg_1, reg_2, gQ, ctx->bitmap);
        } else if (item_mode == DAISY_ITEM_CHUNKED) {
            const int grid = grid_for(2 * v.B, ChunkCfg<C>::E);
            if (reg)
                hipLaunchKernelGGL((k_item_grad_chunked<C, true>), dim3(grid), dim3(kBlock), 0, s, P, Q,
                                   ctx->coef, v.ent_item, v.ent_s, v.ent_u, 2 * v.B, d, stats, reg_1,
                                   reg_2, gQ, ctx->bitmap);
            else
                hipLaunchKernelGGL((k_item_grad_chunked<C, false>), dim3(grid), dim3(kBlock), 0, s, P, Q,
                                   ctx->coef, v.ent_item, v.ent_s, v.ent_u, 2 * v.B, d, stats, reg_1,
                                   reg_2, gQ, ctx->bitmap);
        } else if (reg) {
            hipLaunchKernelGGL((k_item_grad_atomic<C, true>), dim3(grid_for(v.B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, v.u, v.i, v.j, ctx->coef, v.B, d, stats, reg_1,
                               reg_2, gQ, ctx->bitmap);
        } else {
            hipLaunchKernelGGL((k_item_grad_atomic<C, false>), dim3(grid_for(v.B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, v.u, v.i, v.j, ctx->coef, v.B, d, stats, reg_1,
                               reg_2, gQ, ctx->bitmap);
        }
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

static int user_pass(daisy_bpr_ctx *ctx, float *P, const float *Q, const double *stats, float lr,
                     float reg_1, float reg_2, float *gP, bool sgd, daisy_stream_t stream) {
    if (!ctx->fwd_done) { set_error("user update: forward has not run for this batch"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const BatchView &v = ctx->v;
    const int d = ctx->d;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        const int grid = grid_for(v.B, C::GROUPS_PER_BLOCK * 2);
        if (sgd)
            hipLaunchKernelGGL((k_user<C, true>), dim3(grid), dim3(kBlock), 0, s, P, Q, v.u, v.i, v.j,
                               ctx->coef, v.B, d, stats, lr, reg_1, reg_2, gP);
        else
            hipLaunchKernelGGL((k_user<C, false>), dim3(grid), dim3(kBlock), 0, s, P, Q, v.u, v.i, v.j,
                               ctx->coef, v.B, d, stats, lr, reg_1, reg_2, gP);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_user_sgd(daisy_bpr_ctx *ctx, float *P, const float *Q, const double *stats, float lr,
                       float reg_1, float reg_2, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "user_sgd: NULL argument");
    return user_pass(ctx, P, Q, stats, lr, reg_1, reg_2, nullptr, true, stream);
}

int daisy_bpr_user_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gP, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats && gP, "user_grad: NULL argument");
    return user_pass(ctx, const_cast<float *>(P), Q, stats, 0.f, reg_1, reg_2, gP, false, stream);
}

int daisy_bpr_item_sgd_apply(daisy_bpr_ctx *ctx, float *Q, float *gQ, float lr, int32_t dense,
                             daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && Q && gQ, "item_sgd_apply: NULL argument");
    hipStream_t s = S(stream);
    const int d = ctx->d;
    const int64_t I = ctx->I;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_item_apply<C>), dim3(grid_for(I, C::GROUPS_PER_BLOCK * 4)), dim3(kBlock),
                           0, s, Q, gQ, ctx->bitmap, I, d, lr, (int)dense);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    DAISY_HIP(hipMemsetAsync(ctx->bitmap, 0, ctx->bitmap_bytes, s));
    return DAISY_OK;
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

int daisy_bpr_sgd_step(daisy_bpr_ctx *ctx, float *P, float *Q, int32_t loss_type, float gamma,
                       float lr, float reg_1, float reg_2, float *gQ, double *stats,
                       double *epoch_acc, double *step_loss, int32_t item_mode,
                       daisy_stream_t stream) {
    int rc;
    if ((rc = daisy_bpr_forward(ctx, P, Q, loss_type, gamma, stats, stream))) return rc;
    if ((rc = daisy_bpr_finalize(ctx, stats, reg_1, reg_2, epoch_acc, step_loss, stream))) return rc;
    if ((rc = daisy_bpr_item_grad(ctx, P, Q, stats, reg_1, reg_2, gQ, item_mode, stream))) return rc;
    if ((rc = daisy_bpr_user_sgd(ctx, P, Q, stats, lr, reg_1, reg_2, stream))) return rc;
    if ((rc = daisy_bpr_item_sgd_apply(ctx, Q, gQ, lr, 0, stream))) return rc;
    return DAISY_OK;
}

int daisy_bpr_fit_epoch_sgd(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q,
                            int32_t loss_type, float gamma, float lr, float reg_1, float reg_2,
                            float *gQ, double *stats, double *epoch_acc, double *step_losses,
                            int32_t item_mode, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && plan, "fit_epoch: NULL argument");
    if (!plan->built) { set_error("fit_epoch: plan has not been built"); return DAISY_ERR_STATE; }
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
