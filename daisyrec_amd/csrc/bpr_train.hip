// MF + BPR training step for gfx950 (MI355X): hand-written gather / dot /
// loss-coefficient / scatter-update kernels.
//
// Reference semantics being reproduced (file:line in AmazingDD/daisyRec):
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
//                                       the batch is grouped by user)
//   k_item_apply   reads gQ,Q           writes Q, zeroes gQ
// HBM/L2 view: a d=64 row is 256 B = 16 lanes x float4, one coalesced request
// per quarter wave; four samples are in flight per wave instruction.
#include "common.h"

namespace daisy {

struct CtxBuffers {
    int32_t *u, *i, *j;      // batch grouped by user           [max_batch]
    int32_t *tu, *ti, *tj;   // batch as collated               [max_batch]
    int32_t *kin, *kout;     // radix sort keys                 [2*max_batch]
    int32_t *vin, *vout;     // radix sort payload              [2*max_batch]
    float2 *coef;            // (dL/dpos, dL/dneg) per sample   [max_batch]
    double *partials;        // per-workgroup sums              [kMaxGrid*8]
    uint32_t *bitmap;        // touched item rows               [ceil(I/32)]
    void *sort_temp;
};

}  // namespace daisy

struct daisy_bpr_ctx {
    int64_t max_batch, U, I;
    int d;
    void *arena;
    size_t arena_bytes, sort_temp_bytes, bitmap_bytes;
    daisy::CtxBuffers b;
    int64_t B;          // current batch size
    int fwd_grid;       // workgroups used by the last forward
    bool batch_set, fwd_done;
};

namespace daisy {

// ---------------------------------------------------------------------------
// batch preparation
// ---------------------------------------------------------------------------
__global__ void k_gather_triples(const int32_t *__restrict__ triples, const int64_t *__restrict__ idx,
                                 int64_t start, int64_t B, int32_t user_base,
                                 int32_t *__restrict__ tu, int32_t *__restrict__ ti,
                                 int32_t *__restrict__ tj, int32_t *__restrict__ iota) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx ? idx[s] : (start + s);
        const int32_t *t = triples + 3 * r;
        tu[s] = t[0] - user_base;
        ti[s] = t[1];
        tj[s] = t[2];
        iota[s] = (int32_t)s;
    }
}

__global__ void k_copy_batch(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                             const int32_t *__restrict__ j, int64_t B, int32_t *__restrict__ tu,
                             int32_t *__restrict__ ti, int32_t *__restrict__ tj,
                             int32_t *__restrict__ iota) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        tu[s] = u[s];
        ti[s] = i[s];
        tj[s] = j[s];
        iota[s] = (int32_t)s;
    }
}

__global__ void k_permute_ij(const int32_t *__restrict__ perm, const int32_t *__restrict__ ti,
                             const int32_t *__restrict__ tj, int64_t B, int32_t *__restrict__ i,
                             int32_t *__restrict__ j) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = perm[s];
        i[s] = ti[p];
        j[s] = tj[p];
    }
}

__global__ void k_item_entries(const int32_t *__restrict__ i, const int32_t *__restrict__ j, int64_t B,
                               int32_t *__restrict__ key, int32_t *__restrict__ val) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < 2 * B;
         e += (int64_t)gridDim.x * blockDim.x) {
        key[e] = (e < B) ? i[e] : j[e - B];
        val[e] = (int32_t)e;
    }
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
// item gradient, throughput mode: fp32 atomics into the side buffer gQ
//   gQ[i] += cp*p_u + reg_1*sign(q_i) + reg_2*q_i/|Q[i]|_F      (same for j with cn)
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
// item gradient, reproducible mode: entries (item, e) sorted by item (stable);
// the group that sees a segment head owns the row and sums in batch order.
// e < B: positive of sample e, e >= B: negative of sample e-B.
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_grad_sorted(
    const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ u,
    const float2 *__restrict__ coef, const int32_t *__restrict__ key, const int32_t *__restrict__ val,
    int64_t B, int d, const double *__restrict__ stats, float reg_1, float reg_2,
    float *__restrict__ gQ, uint32_t *__restrict__ bitmap) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t n = 2 * B;
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t pos = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; pos < n; pos += gstride) {
        const int32_t r = key[pos];
        if (pos > 0 && key[pos - 1] == r) continue;  // not a segment head
        Row<C> acc;
        acc.zero();
        float n_pos = 0.f, n_neg = 0.f;
        for (int64_t q = pos; q < n && key[q] == r; ++q) {
            const int32_t e = val[q];
            const bool is_pos = e < B;
            const int64_t s = is_pos ? e : (e - B);
            const float2 c2 = coef[s];
            const float c = is_pos ? c2.x : c2.y;
            n_pos += is_pos ? 1.f : 0.f;
            n_neg += is_pos ? 0.f : 1.f;
            Row<C> p;
            p.load(P + (int64_t)u[s] * d, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k) acc.v[k] = fmaf(c, p.v[k], acc.v[k]);
        }
        Row<C> qr;
        qr.load(Q + (int64_t)r * d, lane, d);
        const float w1 = reg_1 * (n_pos + n_neg);
        const float w2 = n_pos * rI + n_neg * rJ;
#pragma unroll
        for (int k = 0; k < C::NE; ++k)
            acc.v[k] += fmaf(w2, qr.v[k], w1 * sgn(qr.v[k]));
        acc.store(gQ + (int64_t)r * d, lane, d);
        if (lane == 0) atomicOr(bitmap + (r >> 5), 1u << (r & 31));
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

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int group_by_user(daisy_bpr_ctx *ctx, int64_t B, hipStream_t s) {
    CtxBuffers &b = ctx->b;
    // stable sort of (user, position): equal users keep batch order
    int rc = sort_pairs_i32(b.sort_temp, ctx->sort_temp_bytes, b.tu, b.u, b.vin, b.vout, B,
                            bits_for(ctx->U), s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_permute_ij, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, b.vout, b.ti,
                       b.tj, B, b.i, b.j);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // namespace daisy

using namespace daisy;

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

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
    c->B = 0; c->batch_set = false; c->fwd_done = false; c->fwd_grid = 0;
    c->sort_temp_bytes = sort_pairs_i32_temp_bytes(2 * max_batch);
    c->bitmap_bytes = align_up((size_t)((item_num + 31) / 32) * 4);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_u = take(max_batch * 4), o_i = take(max_batch * 4), o_j = take(max_batch * 4);
    const size_t o_tu = take(max_batch * 4), o_ti = take(max_batch * 4), o_tj = take(max_batch * 4);
    const size_t o_kin = take(2 * max_batch * 4), o_kout = take(2 * max_batch * 4);
    const size_t o_vin = take(2 * max_batch * 4), o_vout = take(2 * max_batch * 4);
    const size_t o_coef = take(max_batch * 8);
    const size_t o_part = take((size_t)kMaxGrid * 8 * 8);
    const size_t o_bm = take(c->bitmap_bytes);
    const size_t o_tmp = take(c->sort_temp_bytes);
    c->arena_bytes = off;
    hipError_t e = hipMalloc(&c->arena, c->arena_bytes);
    if (e != hipSuccess) {
        set_error("ctx_create: hipMalloc(%zu) failed: %s", c->arena_bytes, hipGetErrorString(e));
        delete c;
        return DAISY_ERR_HIP;
    }
    char *base = (char *)c->arena;
    c->b.u = (int32_t *)(base + o_u); c->b.i = (int32_t *)(base + o_i); c->b.j = (int32_t *)(base + o_j);
    c->b.tu = (int32_t *)(base + o_tu); c->b.ti = (int32_t *)(base + o_ti); c->b.tj = (int32_t *)(base + o_tj);
    c->b.kin = (int32_t *)(base + o_kin); c->b.kout = (int32_t *)(base + o_kout);
    c->b.vin = (int32_t *)(base + o_vin); c->b.vout = (int32_t *)(base + o_vout);
    c->b.coef = (float2 *)(base + o_coef);
    c->b.partials = (double *)(base + o_part);
    c->b.bitmap = (uint32_t *)(base + o_bm);
    c->b.sort_temp = base + o_tmp;
    e = hipMemset(c->b.bitmap, 0, c->bitmap_bytes);
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
    hipError_t e = hipFree(ctx->arena);
    delete ctx;
    if (e != hipSuccess) {
        set_error("ctx_destroy: hipFree failed: %s", hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    return DAISY_OK;
}

size_t daisy_bpr_ctx_scratch_bytes(const daisy_bpr_ctx *ctx) { return ctx ? ctx->arena_bytes : 0; }

int daisy_bpr_set_batch_from_triples(daisy_bpr_ctx *ctx, const int32_t *triples, int64_t n_triples,
                                     const int64_t *idx, int64_t start, int64_t B, int32_t user_base,
                                     daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && triples, "set_batch_from_triples: NULL argument");
    DAISY_CHECK_ARG(B > 0 && B <= ctx->max_batch, "set_batch_from_triples: B=%lld not in 1..%lld",
                    (long long)B, (long long)ctx->max_batch);
    DAISY_CHECK_ARG(idx || (start >= 0 && start + B <= n_triples),
                    "set_batch_from_triples: rows %lld..%lld outside 0..%lld", (long long)start,
                    (long long)(start + B), (long long)n_triples);
    hipStream_t s = S(stream);
    hipLaunchKernelGGL(k_gather_triples, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, triples, idx,
                       start, B, user_base, ctx->b.tu, ctx->b.ti, ctx->b.tj, ctx->b.vin);
    DAISY_LAUNCH_CHECK();
    int rc = group_by_user(ctx, B, s);
    if (rc) return rc;
    ctx->B = B; ctx->batch_set = true; ctx->fwd_done = false;
    return DAISY_OK;
}

int daisy_bpr_set_batch(daisy_bpr_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j,
                        int64_t B, int32_t pre_grouped, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && u && i && j, "set_batch: NULL argument");
    DAISY_CHECK_ARG(B > 0 && B <= ctx->max_batch, "set_batch: B=%lld not in 1..%lld", (long long)B,
                    (long long)ctx->max_batch);
    hipStream_t s = S(stream);
    if (pre_grouped) {
        DAISY_HIP(hipMemcpyAsync(ctx->b.u, u, B * 4, hipMemcpyDeviceToDevice, s));
        DAISY_HIP(hipMemcpyAsync(ctx->b.i, i, B * 4, hipMemcpyDeviceToDevice, s));
        DAISY_HIP(hipMemcpyAsync(ctx->b.j, j, B * 4, hipMemcpyDeviceToDevice, s));
    } else {
        hipLaunchKernelGGL(k_copy_batch, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, u, i, j, B,
                           ctx->b.tu, ctx->b.ti, ctx->b.tj, ctx->b.vin);
        DAISY_LAUNCH_CHECK();
        int rc = group_by_user(ctx, B, s);
        if (rc) return rc;
    }
    ctx->B = B; ctx->batch_set = true; ctx->fwd_done = false;
    return DAISY_OK;
}

int daisy_bpr_forward(daisy_bpr_ctx *ctx, const float *P, const float *Q, int32_t loss_type,
                      float gamma, double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "forward: NULL argument");
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_TL,
                    "Invalid loss type: %d", loss_type);
    if (!ctx->batch_set) { set_error("forward: no batch set"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const int64_t B = ctx->B;
    const int d = ctx->d;
    CtxBuffers &b = ctx->b;
    int grid = 0;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        grid = grid_for(B, C::GROUPS_PER_BLOCK * 4);
        hipLaunchKernelGGL((k_fwd<C>), dim3(grid), dim3(kBlock), 0, s, P, Q, b.u, b.i, b.j, B, d,
                           (int)loss_type, gamma, b.coef, b.partials);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, s, b.partials, grid, stats);
    DAISY_LAUNCH_CHECK();
    ctx->fwd_grid = grid; ctx->fwd_done = true;
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
    DAISY_CHECK_ARG(item_mode == DAISY_ITEM_ATOMIC || item_mode == DAISY_ITEM_SORTED,
                    "item_grad: bad item_mode %d", item_mode);
    if (!ctx->fwd_done) { set_error("item_grad: forward has not run for this batch"); return DAISY_ERR_STATE; }
    hipStream_t s = S(stream);
    const int64_t B = ctx->B;
    const int d = ctx->d;
    CtxBuffers &b = ctx->b;
    if (item_mode == DAISY_ITEM_SORTED) {
        hipLaunchKernelGGL(k_item_entries, dim3(grid_for(2 * B, kBlock)), dim3(kBlock), 0, s, b.i, b.j,
                           B, b.kin, b.vin);
        DAISY_LAUNCH_CHECK();
        int rc = sort_pairs_i32(b.sort_temp, ctx->sort_temp_bytes, b.kin, b.kout, b.vin, b.vout, 2 * B,
                                bits_for(ctx->I), s);
        if (rc) return rc;
    }
    const bool reg = (reg_1 != 0.f) || (reg_2 != 0.f);
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (item_mode == DAISY_ITEM_SORTED) {
            hipLaunchKernelGGL((k_item_grad_sorted<C>), dim3(grid_for(2 * B, C::GROUPS_PER_BLOCK)),
                               dim3(kBlock), 0, s, P, Q, b.u, b.coef, b.kout, b.vout, B, d, stats,
                               reg_1, reg_2, gQ, b.bitmap);
        } else if (reg) {
            hipLaunchKernelGGL((k_item_grad_atomic<C, true>), dim3(grid_for(B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, b.u, b.i, b.j, b.coef, B, d, stats, reg_1,
                               reg_2, gQ, b.bitmap);
        } else {
            hipLaunchKernelGGL((k_item_grad_atomic<C, false>), dim3(grid_for(B, C::GROUPS_PER_BLOCK * 4)),
                               dim3(kBlock), 0, s, P, Q, b.u, b.i, b.j, b.coef, B, d, stats, reg_1,
                               reg_2, gQ, b.bitmap);
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
    const int64_t B = ctx->B;
    const int d = ctx->d;
    CtxBuffers &b = ctx->b;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        const int grid = grid_for(B, C::GROUPS_PER_BLOCK * 2);
        if (sgd)
            hipLaunchKernelGGL((k_user<C, true>), dim3(grid), dim3(kBlock), 0, s, P, Q, b.u, b.i, b.j,
                               b.coef, B, d, stats, lr, reg_1, reg_2, gP);
        else
            hipLaunchKernelGGL((k_user<C, false>), dim3(grid), dim3(kBlock), 0, s, P, Q, b.u, b.i, b.j,
                               b.coef, B, d, stats, lr, reg_1, reg_2, gP);
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
                           0, s, Q, gQ, ctx->b.bitmap, I, d, lr, (int)dense);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    DAISY_HIP(hipMemsetAsync(ctx->b.bitmap, 0, ctx->bitmap_bytes, s));
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

int daisy_bpr_fit_epoch_sgd(daisy_bpr_ctx *ctx, float *P, float *Q, const int32_t *triples,
                            int64_t n_triples, const int64_t *perm, int64_t batch_size,
                            int32_t user_base, int32_t loss_type, float gamma, float lr, float reg_1,
                            float reg_2, float *gQ, double *stats, double *epoch_acc,
                            double *step_losses, int32_t item_mode, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && triples && n_triples > 0 && batch_size > 0, "fit_epoch: bad argument");
    int64_t k = 0;
    for (int64_t start = 0; start < n_triples; start += batch_size, ++k) {
        const int64_t B = (n_triples - start < batch_size) ? (n_triples - start) : batch_size;
        int rc = daisy_bpr_set_batch_from_triples(ctx, triples, n_triples, perm ? perm + start : nullptr,
                                                  start, B, user_base, stream);
        if (rc) return rc;
        rc = daisy_bpr_sgd_step(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, gQ, stats, epoch_acc,
                                step_losses ? step_losses + k : nullptr, item_mode, stream);
        if (rc) return rc;
    }
    return DAISY_OK;
}

}  // extern "C"
