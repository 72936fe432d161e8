// NeuMF at the reference's own operating point (neumf.yaml: factors 24, 2 layers; basic.yaml:23: batch 256): a step of a few
// hundred rows through layers of < 100 columns.  The layer-by-layer path (csrc/neumf.hip) spends such a step in launches - five
// GEMM kernels of 9-17 us each on tiles that are 95 % padding, six reductions of their slices, predict, criterion, finalize: 17 of
// the step's 23 dispatches and ~100 of its 146 us lie between the gather and the scatter.  Here that stretch is ONE launch:
//
//   k_nmf_mid      a workgroup takes 32 rows of the step (16 samples: both rows of a pair meet in one workgroup) with every
//                  layer's weights in LDS: x_l = ReLU(x_{l-1} W_l^T + b_l) (* dropout), pred, the criterion (pair_coef),
//                  dZ_L .. dZ_1, dX0 (-> global, for the scatter), and the workgroup's share of gW_l, gb_l, gWp, gbp and the
//                  loss as one slab of partial sums.  fp32 FMAs on the vector ALU: 18 MFLOP per step over 16 workgroups, no
//                  tile of it is large enough for an MFMA pipeline to matter.
//   k_nmf_mid_reduce   the slabs added in workgroup order into the gradient tensors (+=), the loss and the norms of
//                  NeuMF.calc_loss (k_nmf_finalize's work): bitwise reproducible, like every other reduction of this library.
//
// Same arithmetic as the layered path up to the order of the fp32 sums (k ascending here, the MFMA's k grouping there); the
// dropout masks are the same function of (seed, layer, row, column).  Reference: daisy/model/NeuMFRecommender.py:104-137 (forward),
// :139-169 (calc_loss), AbstractRecommender.py:79-93 (criterion).
#include "neumf_internal.h"

namespace daisy {

constexpr int kMidBlock = 256;
constexpr int kMidRows = 32;         // rows of a workgroup's tile (pairwise: 16 positives + their 16 negatives)
constexpr int kMidRB = 8;            // rows per thread in the forward / input-gradient products
constexpr int kMidLdsBytes = 150 * 1024;

struct MidL16 { static constexpr int LPR = 16; };

struct MidLayout {
    int offW[DAISY_NEUMF_MAX_LAYERS], offB[DAISY_NEUMF_MAX_LAYERS];      // LDS, in floats: W_l rows of width[l-1] + 4 floats
    int offX[DAISY_NEUMF_MAX_LAYERS + 1];                               // x_0 .. x_L tiles [32][width[l]]
    int offDZ[2], offWp, offG;                                          // offG: the GMF products' tile [32][d]
    int lds_floats;
    int slabW[DAISY_NEUMF_MAX_LAYERS], slabB[DAISY_NEUMF_MAX_LAYERS], slabWp, slab;   // a workgroup's slab of partial sums
};

static MidLayout mid_layout(int L, const int *w, int d) {
    MidLayout y{};
    int o = 0;
    for (int l = 1; l <= L; ++l) { y.offW[l - 1] = o; o += w[l] * (w[l - 1] + 4); }
    for (int l = 1; l <= L; ++l) { y.offB[l - 1] = o; o += (w[l] + 3) / 4 * 4; }
    y.offWp = o; o += (d + w[L] + 3) / 4 * 4;
    for (int l = 0; l <= L; ++l) { y.offX[l] = o; o += kMidRows * w[l]; }
    for (int k = 0; k < 2; ++k) { y.offDZ[k] = o; o += kMidRows * w[1]; }
    y.offG = o; o += kMidRows * d;
    y.lds_floats = o;
    int sl = 0;
    for (int l = 1; l <= L; ++l) { y.slabW[l - 1] = sl; sl += w[l] * w[l - 1]; }
    for (int l = 1; l <= L; ++l) { y.slabB[l - 1] = sl; sl += w[l]; }
    y.slabWp = sl; sl += d + w[L];
    y.slab = (sl + 3) / 4 * 4;
    return y;
}

bool neumf_mid_fits(int L, const int *w, int d) {
    if (L < 1 || L > DAISY_NEUMF_MAX_LAYERS || d % 4 || w[L] % 4) return false;
    for (int l = 1; l <= L; ++l)
        if (w[l - 1] % 8 || w[l] % 4) return false;
    return (size_t)mid_layout(L, w, d).lds_floats * sizeof(float) <= (size_t)kMidLdsBytes;
}

size_t neumf_mid_ws_bytes(int L, const int *w, int d, int max_rows) {
    if (!neumf_mid_fits(L, w, d)) return 0;
    const size_t nb = ((size_t)max_rows + 15) / 16 + 1;       // (pointwise: 32 samples per workgroup - fewer)
    return nb * (size_t)mid_layout(L, w, d).slab * sizeof(float) + nb * 2 * sizeof(double) + 16;
}

#ifdef DAISY_MID_PROF
__device__ int mid_prof_calls = 0;
#define MID_MARK(k) { const long long now_ = wall_clock64(); prof[k] += now_ - pt0; pt0 = now_; }
#else
#define MID_MARK(k)
#endif

__global__ __launch_bounds__(kMidBlock) void k_nmf_mid(MidArgs a, MidLayout y, double *__restrict__ wsd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
#ifdef DAISY_MID_PROF
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = wall_clock64();
#endif
    __shared__ float preds[kMidRows], dps[kMidRows], lgb[kMidRows];
    __shared__ double lterm[kMidRows];
    __shared__ int grow_s[kMidRows];
    const int tid = threadIdx.x, L = a.L, d = a.d;
    const int TP = a.pointwise ? kMidRows : kMidRows / 2;
    const int b0 = (int)blockIdx.x * TP;
    float *__restrict__ slab = a.ws + (size_t)blockIdx.x * y.slab;
    if (tid < kMidRows) {             // local row -> row of the step (-1: past the batch)
        const int smp = a.pointwise ? b0 + tid : b0 + (tid & 15);
        const int gr = (a.pointwise || tid < 16) ? smp : a.B + smp;
        grow_s[tid] = (smp < a.B) ? gr : -1;
    }
    for (int l = 1; l <= L; ++l) {    // weights -> LDS, rows padded by 4 floats (16 lanes reading 16 rows' float4 hit 64 distinct banks)
        const int n_in = a.width[l - 1], n_out = a.width[l], q4 = n_in / 4, sw = n_in + 4;
        const float *__restrict__ W = a.W[l - 1];
        float *dst = sm + y.offW[l - 1];
        for (int e = tid; e < n_out * q4; e += kMidBlock) {
            const int n = e / q4, c = e % q4;
            *reinterpret_cast<float4 *>(dst + n * sw + 4 * c) = *reinterpret_cast<const float4 *>(W + (size_t)n * n_in + 4 * c);
        }
        for (int e = tid; e < n_out; e += kMidBlock) sm[y.offB[l - 1] + e] = a.b[l - 1][e];
    }
    const int wL = a.width[L];
    for (int e = tid; e < d + wL; e += kMidBlock) sm[y.offWp + e] = a.Wp[e];
    __syncthreads();
    MID_MARK(0)
    {
        const int w0 = a.width[0], q4 = w0 / 4;
        float *x0 = sm + y.offX[0];
        for (int e = tid; e < kMidRows * q4; e += kMidBlock) {
            const int lr = e / q4, c = e % q4, gr = grow_s[lr];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0) v = *reinterpret_cast<const float4 *>(a.X0 + (size_t)gr * w0 + 4 * c);
            *reinterpret_cast<float4 *>(x0 + lr * w0 + 4 * c) = v;
        }
        // (the GMF products too: read from global inside the predict layer's loops they were 32 dependent L2 round trips per thread -
        // 25 of the kernel's first 34 us)
        const int g4 = d / 4;
        float *gs = sm + y.offG;
        for (int e = tid; e < kMidRows * g4; e += kMidBlock) {
            const int lr = e / g4, c = e % g4, gr = grow_s[lr];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0) v = *reinterpret_cast<const float4 *>(a.G + (size_t)gr * d + 4 * c);
            *reinterpret_cast<float4 *>(gs + lr * d + 4 * c) = v;
        }
    }
    __syncthreads();
    MID_MARK(1)
    // ---- forward: a thread holds one output column of 8 rows
    for (int l = 1; l <= L; ++l) {
        const int n_in = a.width[l - 1], n_out = a.width[l], sw = n_in + 4;
        const float *Wl = sm + y.offW[l - 1], *xin = sm + y.offX[l - 1];
        float *xout = sm + y.offX[l];
        const bool drop = l < L && a.thresh != 0;
        for (int it = tid; it < n_out * (kMidRows / kMidRB); it += kMidBlock) {
            const int n = it % n_out, rg = it / n_out;
            const float bias = sm[y.offB[l - 1] + n];
            float acc[kMidRB];
#pragma unroll
            for (int q = 0; q < kMidRB; ++q) acc[q] = bias;
            const float *wrow = Wl + n * sw, *xr = xin + rg * kMidRB * n_in;
            for (int k = 0; k < n_in; k += 4) {
                const float4 w = *reinterpret_cast<const float4 *>(wrow + k);
#pragma unroll
                for (int q = 0; q < kMidRB; ++q) {
                    const float4 x = *reinterpret_cast<const float4 *>(xr + q * n_in + k);
                    acc[q] = fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, fmaf(x.x, w.x, acc[q]))));
                }
            }
#pragma unroll
            for (int q = 0; q < kMidRB; ++q) {
                const int lr = rg * kMidRB + q, gr = grow_s[lr];
                float v = fmaxf(acc[q], 0.f);
                if (gr < 0) v = 0.f;
                else if (drop) v = drop_keep(a.seed, (uint32_t)(l + 1), (uint64_t)gr * (uint64_t)n_out + (uint64_t)n, a.thresh) ? v * a.scale : 0.f;
                xout[lr * n_out + n] = v;
            }
        }
        __syncthreads();
        MID_MARK(1 + l)
    }
    // ---- predict layer (16 lanes per row, the summation order of k_nmf_predict)
    {
        const float *xL = sm + y.offX[L], *wp = sm + y.offWp, *gs = sm + y.offG;
        const int lane = tid % 16, group = tid / 16;
        for (int lr = group; lr < kMidRows; lr += kMidBlock / 16) {
            const int gr = grow_s[lr];
            float s = 0.f;
            for (int c = lane; c < d; c += 16) s = fmaf(wp[c], gs[lr * d + c], s);
            for (int c = lane; c < wL; c += 16) s = fmaf(wp[d + c], xL[lr * wL + c], s);
            s = group_sum<MidL16>(s);
            if (lane == 0) {
                const float pr = s + a.bp[0];
                preds[lr] = pr;
                if (gr >= 0) a.pred[gr] = pr;
            }
        }
    }
    __syncthreads();
    MID_MARK(5)
    // ---- criterion
    if (tid < TP) {
        const int smp = b0 + tid;
        float term = 0.f, cp = 0.f, cn = 0.f;
        if (smp < a.B) {
            pair_coef(a.loss_type, preds[tid], a.pointwise ? (float)a.j[smp] : preds[16 + tid], a.gamma, term, cp, cn);
            a.dpred[smp] = cp;
            if (!a.pointwise) a.dpred[a.B + smp] = cn;
        }
        dps[tid] = cp;
        if (!a.pointwise) dps[16 + tid] = cn;
        lterm[tid] = (double)term;
        lgb[tid] = cp + cn;           // paired per sample: exactly 0 under BPR / HL, as in the reference's autograd
    }
    __syncthreads();
    MID_MARK(6)
    if (tid == 0) {
        double t = 0.0, gbp = 0.0;
        for (int k = 0; k < TP; ++k) { t += lterm[k]; gbp += (double)lgb[k]; }
        wsd[2 * (size_t)blockIdx.x] = t;
        wsd[2 * (size_t)blockIdx.x + 1] = gbp;
    }
    // ---- predict layer backward: dZ_L = dpred * Wp[d:] gated by x_L > 0; gWp
    int cur = 0;
    {
        const float *xL = sm + y.offX[L], *wp = sm + y.offWp, *gs = sm + y.offG;
        float *dz = sm + y.offDZ[0];
        for (int e = tid; e < kMidRows * wL; e += kMidBlock) {
            const int lr = e / wL, c = e % wL;
            dz[e] = (xL[e] > 0.f) ? dps[lr] * wp[d + c] : 0.f;
        }
        for (int c = tid; c < d + wL; c += kMidBlock) {
            float t = 0.f;
            const float *col = (c < d) ? gs + c : xL + (c - d);       // (rows past the batch: dps = 0)
            const int pitch = (c < d) ? d : wL;
            for (int lr = 0; lr < kMidRows; ++lr) t = fmaf(dps[lr], col[lr * pitch], t);
            slab[y.slabWp + c] = t;
        }
    }
    __syncthreads();
    MID_MARK(7)
    // ---- the layers backward
    for (int l = L; l >= 1; --l) {
        const int n_in = a.width[l - 1], n_out = a.width[l], sw = n_in + 4;
        const float *Wl = sm + y.offW[l - 1], *xin = sm + y.offX[l - 1], *dz = sm + y.offDZ[cur];
        float *dzn = sm + y.offDZ[cur ^ 1];
        for (int n = tid; n < n_out; n += kMidBlock) {        // gb_l
            float t = 0.f;
            for (int lr = 0; lr < kMidRows; ++lr) t += dz[lr * n_out + n];
            slab[y.slabB[l - 1] + n] = t;
        }
        for (int it = tid; it < n_in * (n_out / 4); it += kMidBlock) {       // gW_l: a thread holds column k of 4 rows n
            const int k = it % n_in, nb = it / n_in;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int lr = 0; lr < kMidRows; ++lr) {
                const float x = xin[lr * n_in + k];
                const float4 z = *reinterpret_cast<const float4 *>(dz + lr * n_out + 4 * nb);
                acc[0] = fmaf(z.x, x, acc[0]); acc[1] = fmaf(z.y, x, acc[1]);
                acc[2] = fmaf(z.z, x, acc[2]); acc[3] = fmaf(z.w, x, acc[3]);
            }
            float *o = slab + y.slabW[l - 1] + (size_t)(4 * nb) * n_in + k;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[(size_t)q * n_in] = acc[q];
        }
        for (int it = tid; it < n_in * (kMidRows / kMidRB); it += kMidBlock) {      // dZ_{l-1}: column k of 8 rows
            const int k = it % n_in, rg = it / n_in;
            float acc[kMidRB];
#pragma unroll
            for (int q = 0; q < kMidRB; ++q) acc[q] = 0.f;
            const float *zr = dz + rg * kMidRB * n_out;
            for (int n = 0; n < n_out; n += 4) {
                const float w0 = Wl[n * sw + k], w1 = Wl[(n + 1) * sw + k], w2 = Wl[(n + 2) * sw + k], w3 = Wl[(n + 3) * sw + k];
#pragma unroll
                for (int q = 0; q < kMidRB; ++q) {
                    const float4 z = *reinterpret_cast<const float4 *>(zr + q * n_out + n);
                    acc[q] = fmaf(z.w, w3, fmaf(z.z, w2, fmaf(z.y, w1, fmaf(z.x, w0, acc[q]))));
                }
            }
#pragma unroll
            for (int q = 0; q < kMidRB; ++q) {
                const int lr = rg * kMidRB + q, gr = grow_s[lr];
                if (l > 1) {          // ReLU (and dropout) gate of x_{l-1}
                    dzn[lr * n_in + k] = (xin[lr * n_in + k] > 0.f) ? acc[q] * a.scale : 0.f;
                } else if (gr >= 0) {      // the concat input: its dropout mask
                    float v = acc[q];
                    if (a.thresh) v = drop_keep(a.seed, 1u, (uint64_t)gr * (uint64_t)n_in + (uint64_t)k, a.thresh) ? v * a.scale : 0.f;
                    a.DX0[(size_t)gr * n_in + k] = v;
                }
            }
        }
        __syncthreads();
        MID_MARK(8 + (l > 1 ? 1 : 0))
        cur ^= 1;
    }
#ifdef DAISY_MID_PROF
    if (tid == 0 && blockIdx.x == 0 && atomicAdd(&mid_prof_calls, 1) % 200 == 150)
        printf("k_nmf_mid wg 0, x10 ns: weights->LDS %lld  x0,g->LDS %lld  F1 %lld F2 %lld F3 %lld  predict %lld  criterion %lld  pred bwd %lld  "
               "B1 %lld  B2.. %lld\n", prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7], prof[8], prof[9]);
#endif
}

struct MidSegs {
    float *dst[2 * DAISY_NEUMF_MAX_LAYERS + 1];
    int off[2 * DAISY_NEUMF_MAX_LAYERS + 2];
    int n;
};

__global__ __launch_bounds__(kMidBlock) void k_nmf_mid_reduce(const float *__restrict__ ws, const double *__restrict__ wsd, int nb,
                                                             int slab, MidSegs segs, float *gbp, double *__restrict__ stats,
                                                             float reg_1, float reg_2, int pointwise) {
    if (blockIdx.x + 1 < gridDim.x) {
        const int c = (int)blockIdx.x * kMidBlock + (int)threadIdx.x;
        if (c >= segs.off[segs.n]) return;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int sl = 0;
        for (; sl + 3 < nb; sl += 4) {                       // four loads in flight, one fixed association
            t0 += ws[(size_t)sl * slab + c];
            t1 += ws[(size_t)(sl + 1) * slab + c];
            t2 += ws[(size_t)(sl + 2) * slab + c];
            t3 += ws[(size_t)(sl + 3) * slab + c];
        }
        for (; sl < nb; ++sl) t0 += ws[(size_t)sl * slab + c];
        const float t = (t0 + t1) + (t2 + t3);
        float *o = nullptr;
        for (int k = 0; k < segs.n; ++k)
            if (c >= segs.off[k] && c < segs.off[k + 1]) o = segs.dst[k] + (c - segs.off[k]);
        if (o) *o += t;
        return;
    }
    if (threadIdx.x) return;
    double loss = 0.0, gb = 0.0;
    for (int b = 0; b < nb; ++b) { loss += wsd[2 * (size_t)b]; gb += wsd[2 * (size_t)b + 1]; }
    stats[DAISY_NST_LOSS_DATA] += loss;
    gbp[0] += (float)gb;
    double l1 = 0.0, fro = 0.0;       // NeuMFRecommender.py:149-167: the negative item's GMF rows enter twice (k_nmf_finalize)
    for (int k = 0; k < 5; ++k) {
        const double n = sqrt(stats[DAISY_NST_SQ + k]);
        stats[DAISY_NST_NORM + k] = n;
        const double w = (k == 4) ? (pointwise ? 0.0 : 2.0) : 1.0;
        l1 += w * stats[DAISY_NST_L1 + k];
        fro += w * n;
    }
    stats[DAISY_NST_LOSS] = stats[DAISY_NST_LOSS_DATA] + (double)reg_1 * l1 + (double)reg_2 * fro;
}

int neumf_mid_step(const MidArgs &args, float *const *gW, float *const *gb, float *gWp, float *gbp, double *stats, float reg_1,
                   float reg_2, hipStream_t s) {
    const MidLayout y = mid_layout(args.L, args.width, args.d);
    const int TP = args.pointwise ? kMidRows : kMidRows / 2;
    const int nb = (args.B + TP - 1) / TP;
    static bool attr_set = false;
    if (!attr_set) {
        DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_nmf_mid), hipFuncAttributeMaxDynamicSharedMemorySize, kMidLdsBytes));
        attr_set = true;
    }
    double *wsd = reinterpret_cast<double *>(args.ws + (((size_t)nb * y.slab + 3) / 4) * 4);
    hipLaunchKernelGGL(k_nmf_mid, dim3((unsigned)nb), dim3(kMidBlock), (size_t)y.lds_floats * sizeof(float), s, args, y, wsd);
    MidSegs segs{};
    int n = 0;
    for (int l = 1; l <= args.L; ++l) { segs.dst[n] = gW[l - 1]; segs.off[n] = y.slabW[l - 1]; ++n; }
    for (int l = 1; l <= args.L; ++l) { segs.dst[n] = gb[l - 1]; segs.off[n] = y.slabB[l - 1]; ++n; }
    segs.dst[n] = gWp; segs.off[n] = y.slabWp; ++n;
    segs.off[n] = y.slabWp + args.d + args.width[args.L];
    segs.n = n;
    hipLaunchKernelGGL(k_nmf_mid_reduce, dim3((unsigned)((segs.off[n] + kMidBlock - 1) / kMidBlock + 1)), dim3(kMidBlock), 0, s,
                       args.ws, wsd, nb, y.slab, segs, gbp, stats, reg_1, reg_2, args.pointwise);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // namespace daisy
