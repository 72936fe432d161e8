// NeuMF at the reference's own operating point (neumf.yaml: factors 24, 2 layers; basic.yaml:23: batch 256): a step of a few
// hundred rows through layers of < 100 columns.  The layer-by-layer path (csrc/neumf.hip) spends such a step in launches - five
// GEMM kernels of 9-17 us each on tiles that are 95 % padding, six reductions of their slices, predict, criterion, finalize: 17 of
// the step's 23 dispatches and ~100 of its 146 us lie between the gather and the scatter.  Here that stretch is ONE launch:
//
//   k_nmf_mid      a workgroup takes 8 rows of the step (4 samples: both rows of a pair meet in one workgroup) with every
//                  layer's weights in LDS: x_l = ReLU(x_{l-1} W_l^T + b_l) (* dropout), pred, the criterion (pair_coef),
//                  dZ_L .. dZ_1, dX0 (-> global, for the scatter), and the workgroup's share of gW_l, gb_l, gWp, gbp and the
//                  loss as one slab of partial sums.  fp32 FMAs on the vector ALU: 18 MFLOP per step, every phase a chain of
//                  a few dozen dependent LDS reads and a few hundred dependent instructions on one wave per SIMD - what counts
//                  is how many workgroups share the rows (64 at 256 samples) and how many instructions the longest thread
//                  runs (first version: 32 rows per workgroup, 30 us).
//   k_nmf_mid_reduce   the slabs added in workgroup order into the gradient tensors (+=), the loss and the norms of
//                  NeuMF.calc_loss (k_nmf_finalize's work): bitwise reproducible, like every other reduction of this library.
//
// Same arithmetic as the layered path up to the order of the fp32 sums (k ascending here, the MFMA's k grouping there); the
// dropout masks are the same function of (seed, layer, row, column).  Reference: daisy/model/NeuMFRecommender.py:104-137 (forward),
// :139-169 (calc_loss), AbstractRecommender.py:79-93 (criterion).
#include "neumf_internal.h"

namespace daisy {

constexpr int kMidBlock = 256;
constexpr int kMidRows = 8;          // rows of a workgroup's tile (pairwise: 4 positives + their 4 negatives)
constexpr int kMidLdsBytes = 150 * 1024;

struct MidL16 { static constexpr int LPR = 16; };

struct MidLayout {
    int offW[DAISY_NEUMF_MAX_LAYERS], offB[DAISY_NEUMF_MAX_LAYERS];      // LDS, in floats: W_l rows of width[l-1] + 4 floats
    int offX[DAISY_NEUMF_MAX_LAYERS + 1];                               // x_0 .. x_L tiles [rows][width[l]]
    int offDZ[2], offWp, offG;                                          // offG: the GMF products' tile [rows][d]
    int lds_floats;
    int slabW[DAISY_NEUMF_MAX_LAYERS], slabB[DAISY_NEUMF_MAX_LAYERS], slabWp, slab;   // a workgroup's slab of partial sums
};

// the parameters' way into LDS: W_1 .. W_L (rows of q4 float4 -> rows of `pitch` floats), b_1 .. b_L, Wp (one row each) as one
// list of float4s - segment s holds elements start[s] .. start[s+1]
constexpr int kMidMaxLayers = 3;     // (what fits the LDS has at most three layers in practice; the segment list stays in SGPRs)
constexpr int kMidSegs = 2 * kMidMaxLayers + 1;
constexpr int kMidDoubles = 12;      // a workgroup's double sums: loss, gbp, L1[5], SQ[5]
struct MidCopy {
    const float *src[kMidSegs];
    int start[kMidSegs + 1], q4[kMidSegs], dst[kMidSegs], pitch[kMidSegs];
    uint32_t magic[kMidSegs];        // ceil(2^32 / q4): row of element e = mulhi(e, magic)
    int n;
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

static int mid_param_float4(int L, const int *w, int d) {
    int n = (d + w[L]) / 4;
    for (int l = 1; l <= L; ++l) n += w[l] * w[l - 1] / 4 + w[l] / 4;
    return n;
}

bool neumf_mid_fits(int L, const int *w, int d) {
    if (L < 1 || L > kMidMaxLayers || d % 4 || w[L] % 4) return false;
    for (int l = 1; l <= L; ++l)
        if (w[l - 1] % 8 || w[l] % 4) return false;
    // (the tile of x0 and of the GMF products is fetched into registers in one go: 4 and 1 float4 per thread; the parameters
    // in at most 16 float4 per thread)
    if (kMidRows * w[0] / 4 > 4 * kMidBlock || kMidRows * d / 4 > kMidBlock) return false;
    if (mid_param_float4(L, w, d) > 16 * kMidBlock) return false;
    return (size_t)mid_layout(L, w, d).lds_floats * sizeof(float) <= (size_t)kMidLdsBytes;
}

size_t neumf_mid_ws_bytes(int L, const int *w, int d, int max_rows) {
    if (!neumf_mid_fits(L, w, d)) return 0;
    const size_t nb = ((size_t)max_rows + kMidRows / 2 - 1) / (kMidRows / 2) + 1;       // (pointwise: twice the samples per workgroup)
    return nb * (size_t)mid_layout(L, w, d).slab * sizeof(float) + nb * kMidDoubles * sizeof(double) + 16;
}

#ifdef DAISY_MID_PROF
__device__ int mid_prof_calls = 0;
#define MID_MARK(k) { const long long now_ = wall_clock64(); prof[k] += now_ - pt0; pt0 = now_; }
#else
#define MID_MARK(k)
#endif

template <int NQ>
__global__ __launch_bounds__(kMidBlock) void k_nmf_mid(MidArgs a, MidLayout y, MidCopy cp, double *__restrict__ wsd) {
    constexpr int TR = kMidRows, RB = TR / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
#ifdef DAISY_MID_PROF
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = wall_clock64();
#endif
    __shared__ float preds[TR], dps[TR], lgb[TR];
    __shared__ double lterm[TR];
    __shared__ int grow_s[TR], uid_s[TR], iid_s[TR];
    __shared__ double regs_s[kMidBlock / kWave][10];
    const int tid = threadIdx.x, L = a.L, d = a.d, dm = a.dm;
    const int TP = a.pointwise ? TR : TR / 2;
    const int b0 = (int)blockIdx.x * TP;
    float *__restrict__ slab = a.ws + (size_t)blockIdx.x * y.slab;
    // ---- global -> LDS in two dependent steps: (the batch's ids + every parameter), then the tables' rows.  Nothing later in
    // the kernel reads global memory.  A dependent read of what the previous kernel wrote costs 0.2 - 0.4 us
    // (profiles/r06_latency_probe.txt) - little next to a launch, a lot inside a loop: the first version read the GMF products
    // inside the predict layer's backward loop (32 reads in a row per thread, 3.7 of its 33.6 us).
    int my_u = 0, my_i = 0, my_gr = -1;
    if (tid < TR) {                   // local row -> row of the step (-1: past the batch), its user and item
        const int smp = a.pointwise ? b0 + tid : b0 + (tid % (TR / 2));
        const bool pos = a.pointwise || tid < TR / 2;
        if (smp < a.B) {
            my_gr = pos ? smp : a.B + smp;
            my_u = a.u[smp];
            my_i = pos ? a.i[smp] : a.j[smp];
        }
    }
    // (every thread issues NQ loads - a count the compiler can wait on selectively.  Named registers, not an array: the loop
    // over an array of 16 float4 is unrolled after the pass that would have promoted the array, and it lands in scratch)
    auto locate = [&](int e, int &off) -> const float4 * {
        const float *sp = cp.src[0];
        int rem = 0, q4 = 1, pitch = 0, dsto = 0;
        uint32_t magic = 0;
        bool in = false;
        // (fully unrolled over the at most 7 segments: their descriptors are scalar loads issued once, up front - a loop over
        // cp.n re-read them from the kernel arguments per element and segment, 40 dependent scalar loads: 2.4 of this stage's 5.1 us)
#pragma unroll
        for (int sg = 0; sg < kMidSegs; ++sg)
            if (sg < cp.n && e >= cp.start[sg] && e < cp.start[sg + 1]) {
                in = true; sp = cp.src[sg]; rem = e - cp.start[sg]; q4 = cp.q4[sg]; pitch = cp.pitch[sg]; dsto = cp.dst[sg]; magic = cp.magic[sg];
            }
        const int n = (int)__umulhi((uint32_t)rem, magic);
        off = in ? dsto + n * pitch + 4 * (rem - n * q4) : -1;
        return reinterpret_cast<const float4 *>(sp) + rem;
    };
#define MID_ISSUE(q) float4 pv##q = make_float4(0.f, 0.f, 0.f, 0.f); int po##q = -1; \
    if constexpr (NQ > q) pv##q = *locate(q * kMidBlock + tid, po##q);
    MID_ISSUE(0) MID_ISSUE(1) MID_ISSUE(2) MID_ISSUE(3) MID_ISSUE(4) MID_ISSUE(5) MID_ISSUE(6) MID_ISSUE(7)
    MID_ISSUE(8) MID_ISSUE(9) MID_ISSUE(10) MID_ISSUE(11) MID_ISSUE(12) MID_ISSUE(13) MID_ISSUE(14) MID_ISSUE(15)
#undef MID_ISSUE
    if (tid < TR) { grow_s[tid] = my_gr; uid_s[tid] = my_u; iid_s[tid] = my_i; }
    __syncthreads();
    MID_MARK(10)
    const int w0 = a.width[0], wL = a.width[L];
    float4 rx[4], rgu, rgi;
    {
        const int q4 = w0 / 4, h4 = dm / 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int e = q * kMidBlock + tid;
            if (e >= TR * q4) e = 0;
            const int lr = e / q4, c4 = e % q4;
            const float *row = (c4 < h4) ? a.uM + (size_t)uid_s[lr] * dm + 4 * c4 : a.iM + (size_t)iid_s[lr] * dm + 4 * (c4 - h4);
            rx[q] = *reinterpret_cast<const float4 *>(row);
        }
        const int g4 = d / 4;
        const int e = (tid < TR * g4) ? tid : 0;
        rgu = *reinterpret_cast<const float4 *>(a.uG + (size_t)uid_s[e / g4] * d + 4 * (e % g4));
        rgi = *reinterpret_cast<const float4 *>(a.iG + (size_t)iid_s[e / g4] * d + 4 * (e % g4));
    }
    // the parameters: weight rows padded by 4 floats (16 lanes reading 16 rows' float4 hit 64 distinct banks)
#define MID_COMMIT(q) if (NQ > q && po##q >= 0) *reinterpret_cast<float4 *>(sm + po##q) = pv##q;
    MID_COMMIT(0) MID_COMMIT(1) MID_COMMIT(2) MID_COMMIT(3) MID_COMMIT(4) MID_COMMIT(5) MID_COMMIT(6) MID_COMMIT(7)
    MID_COMMIT(8) MID_COMMIT(9) MID_COMMIT(10) MID_COMMIT(11) MID_COMMIT(12) MID_COMMIT(13) MID_COMMIT(14) MID_COMMIT(15)
#undef MID_COMMIT
    MID_MARK(11)
    // x0 = [uM[u] | iM[item]] (* the dropout mask of layer 1), g = uG[u] * iG[item], and the rows' share of the regulariser sums
    // (NeuMFRecommender.py:149-167: the positive rows' four embeddings, the negative rows' GMF item embedding)
    float s1[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, s2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    {
        const int q4 = w0 / 4, h4 = dm / 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * kMidBlock + tid;
            if (e < TR * q4) {
                const int lr = e / q4, c4 = e % q4, gr = grow_s[lr];
                const bool first = a.pointwise || lr < TR / 2;
                float v[4] = {rx[q].x, rx[q].y, rx[q].z, rx[q].w};
                if (gr < 0) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                else {
                    const bool left = c4 < h4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (first && left) { s1[1] += fabsf(v[k]); s2[1] = fmaf(v[k], v[k], s2[1]); }
                        if (first && !left) { s1[3] += fabsf(v[k]); s2[3] = fmaf(v[k], v[k], s2[3]); }
                        if (a.thresh) v[k] = drop_keep(a.seed, 1u, (uint64_t)gr * (uint64_t)w0 + (uint64_t)(4 * c4 + k), a.thresh) ? v[k] * a.scale : 0.f;
                    }
                }
                *reinterpret_cast<float4 *>(sm + y.offX[0] + 4 * e) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (tid < TR * (d / 4)) {
            const int lr = tid / (d / 4), gr = grow_s[lr];
            const bool first = a.pointwise || lr < TR / 2;
            const float ua[4] = {rgu.x, rgu.y, rgu.z, rgu.w}, ia[4] = {rgi.x, rgi.y, rgi.z, rgi.w};
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0) {
                gv = make_float4(ua[0] * ia[0], ua[1] * ia[1], ua[2] * ia[2], ua[3] * ia[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (first) { s1[0] += fabsf(ua[k]); s2[0] = fmaf(ua[k], ua[k], s2[0]); s1[2] += fabsf(ia[k]); s2[2] = fmaf(ia[k], ia[k], s2[2]); }
                    else if (!a.pointwise) { s1[4] += fabsf(ia[k]); s2[4] = fmaf(ia[k], ia[k], s2[4]); }
                }
            }
            *reinterpret_cast<float4 *>(sm + y.offG + 4 * tid) = gv;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {      // waves in order, workgroups in order (k_nmf_mid_reduce): reproducible
            const double sa = wave_sum_f64_dpp((double)s1[k]), sb = wave_sum_f64_dpp((double)s2[k]);
            if (tid % kWave == 0) { regs_s[tid / kWave][k] = sa; regs_s[tid / kWave][5 + k] = sb; }
        }
    }
    __syncthreads();
    if (tid < 10) {
        double t = 0.0;
        for (int w = 0; w < kMidBlock / kWave; ++w) t += regs_s[w][tid];
        wsd[(size_t)blockIdx.x * kMidDoubles + 2 + tid] = t;
    }
    MID_MARK(0)
    // ---- forward: a thread holds one output column of RB rows
    for (int l = 1; l <= L; ++l) {
        const int n_in = a.width[l - 1], n_out = a.width[l], sw = n_in + 4;
        const float *Wl = sm + y.offW[l - 1], *xin = sm + y.offX[l - 1];
        float *xout = sm + y.offX[l];
        const bool drop = l < L && a.thresh != 0;
        for (int it = tid; it < n_out * (TR / RB); it += kMidBlock) {
            const int n = it % n_out, rg_ = it / n_out;
            const float bias = sm[y.offB[l - 1] + n];
            float acc[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) acc[q] = bias;
            const float *wrow = Wl + n * sw, *xr = xin + rg_ * RB * n_in;
#pragma unroll 2
            for (int k = 0; k < n_in; k += 4) {
                const float4 w = *reinterpret_cast<const float4 *>(wrow + k);
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const float4 x = *reinterpret_cast<const float4 *>(xr + q * n_in + k);
                    acc[q] = fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, fmaf(x.x, w.x, acc[q]))));
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int lr = rg_ * RB + q, gr = grow_s[lr];
                float v = fmaxf(acc[q], 0.f);
                if (gr < 0) v = 0.f;
                else if (drop) v = drop_keep(a.seed, (uint32_t)(l + 1), (uint64_t)gr * (uint64_t)n_out + (uint64_t)n, a.thresh) ? v * a.scale : 0.f;
                xout[lr * n_out + n] = v;
            }
        }
        __syncthreads();
        MID_MARK(l)
    }
    // ---- predict layer (16 lanes per row, the summation order of k_nmf_predict)
    {
        const float *xL = sm + y.offX[L], *wp = sm + y.offWp, *gs = sm + y.offG;
        const int lane = tid % 16, group = tid / 16;
        for (int lr = group; lr < TR; lr += kMidBlock / 16) {
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
            pair_coef(a.loss_type, preds[tid], a.pointwise ? (float)a.j[smp] : preds[TR / 2 + tid], a.gamma, term, cp, cn);
            a.dpred[smp] = cp;
            if (!a.pointwise) a.dpred[a.B + smp] = cn;
        }
        dps[tid] = cp;
        if (!a.pointwise) dps[TR / 2 + tid] = cn;
        lterm[tid] = (double)term;
        lgb[tid] = cp + cn;           // paired per sample: exactly 0 under BPR / HL, as in the reference's autograd
    }
    __syncthreads();
    MID_MARK(6)
    if (tid == 0) {
        double t = 0.0, gbp = 0.0;
        for (int k = 0; k < TP; ++k) { t += lterm[k]; gbp += (double)lgb[k]; }
        wsd[(size_t)blockIdx.x * kMidDoubles] = t;
        wsd[(size_t)blockIdx.x * kMidDoubles + 1] = gbp;
    }
    // ---- predict layer backward: dZ_L = dpred * Wp[d:] gated by x_L > 0; gWp
    int cur = 0;
    {
        const float *xL = sm + y.offX[L], *wp = sm + y.offWp, *gs = sm + y.offG;
        float *dz = sm + y.offDZ[0];
        for (int e = tid; e < TR * wL; e += kMidBlock) {
            const int lr = e / wL, c = e % wL;
            dz[e] = (xL[e] > 0.f) ? dps[lr] * wp[d + c] : 0.f;
        }
        for (int c = tid; c < d + wL; c += kMidBlock) {
            float t = 0.f;
            const float *col = (c < d) ? gs + c : xL + (c - d);       // (rows past the batch: dps = 0)
            const int pitch = (c < d) ? d : wL;
#pragma unroll
            for (int lr = 0; lr < TR; ++lr) t = fmaf(dps[lr], col[lr * pitch], t);
            slab[y.slabWp + c] = t;
        }
    }
    __syncthreads();
    MID_MARK(7)
    // ---- the layers backward
    for (int l = L; l >= 1; --l) {
        const int n_in = a.width[l - 1], n_out = a.width[l], sw = n_in + 4, k4n = n_in / 4;
        const float *Wl = sm + y.offW[l - 1], *xin = sm + y.offX[l - 1], *dz = sm + y.offDZ[cur];
        float *dzn = sm + y.offDZ[cur ^ 1];
        for (int n = tid; n < n_out; n += kMidBlock) {        // gb_l
            float t = 0.f;
#pragma unroll
            for (int lr = 0; lr < TR; ++lr) t += dz[lr * n_out + n];
            slab[y.slabB[l - 1] + n] = t;
        }
        for (int it = tid; it < k4n * (n_out / 4); it += kMidBlock) {       // gW_l: a thread holds 4 rows n x 4 columns k
            const int kb = it % k4n, nb = it / k4n;
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) acc[i][jx] = 0.f;
#pragma unroll
            for (int lr = 0; lr < TR; ++lr) {
                const float4 x = *reinterpret_cast<const float4 *>(xin + lr * n_in + 4 * kb);
                const float4 z = *reinterpret_cast<const float4 *>(dz + lr * n_out + 4 * nb);
                const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][0] = fmaf(zz[i], x.x, acc[i][0]); acc[i][1] = fmaf(zz[i], x.y, acc[i][1]);
                    acc[i][2] = fmaf(zz[i], x.z, acc[i][2]); acc[i][3] = fmaf(zz[i], x.w, acc[i][3]);
                }
            }
            float *o = slab + y.slabW[l - 1] + (size_t)(4 * nb) * n_in + 4 * kb;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4 *>(o + (size_t)i * n_in) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        }
        for (int it = tid; it < k4n * TR; it += kMidBlock) {      // dZ_{l-1}: 4 columns k of one row
            const int kb = it % k4n, lr = it / k4n;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *zr = dz + lr * n_out, *wc = Wl + 4 * kb;
#pragma unroll 2
            for (int n = 0; n < n_out; n += 4) {
                const float4 z = *reinterpret_cast<const float4 *>(zr + n);
                const float4 w0 = *reinterpret_cast<const float4 *>(wc + n * sw), w1 = *reinterpret_cast<const float4 *>(wc + (n + 1) * sw),
                             w2 = *reinterpret_cast<const float4 *>(wc + (n + 2) * sw), w3 = *reinterpret_cast<const float4 *>(wc + (n + 3) * sw);
                acc.x = fmaf(z.w, w3.x, fmaf(z.z, w2.x, fmaf(z.y, w1.x, fmaf(z.x, w0.x, acc.x))));
                acc.y = fmaf(z.w, w3.y, fmaf(z.z, w2.y, fmaf(z.y, w1.y, fmaf(z.x, w0.y, acc.y))));
                acc.z = fmaf(z.w, w3.z, fmaf(z.z, w2.z, fmaf(z.y, w1.z, fmaf(z.x, w0.z, acc.z))));
                acc.w = fmaf(z.w, w3.w, fmaf(z.z, w2.w, fmaf(z.y, w1.w, fmaf(z.x, w0.w, acc.w))));
            }
            float v[4] = {acc.x, acc.y, acc.z, acc.w};
            if (l > 1) {              // ReLU (and dropout) gate of x_{l-1}
                const float4 x = *reinterpret_cast<const float4 *>(xin + lr * n_in + 4 * kb);
                v[0] = (x.x > 0.f) ? v[0] * a.scale : 0.f; v[1] = (x.y > 0.f) ? v[1] * a.scale : 0.f;
                v[2] = (x.z > 0.f) ? v[2] * a.scale : 0.f; v[3] = (x.w > 0.f) ? v[3] * a.scale : 0.f;
                *reinterpret_cast<float4 *>(dzn + lr * n_in + 4 * kb) = make_float4(v[0], v[1], v[2], v[3]);
            } else {                  // the concat input: its dropout mask
                const int gr = grow_s[lr];
                if (gr >= 0) {
                    if (a.thresh) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = drop_keep(a.seed, 1u, (uint64_t)gr * (uint64_t)n_in + (uint64_t)(4 * kb + q), a.thresh) ? v[q] * a.scale : 0.f;
                    }
                    *reinterpret_cast<float4 *>(a.DX0 + (size_t)gr * n_in + 4 * kb) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();
        MID_MARK(8 + (l > 1 ? 1 : 0))
        cur ^= 1;
    }
#ifdef DAISY_MID_PROF
    if (tid == 0 && blockIdx.x == 0 && atomicAdd(&mid_prof_calls, 1) % 200 == 150)
        printf("k_nmf_mid wg 0, x10 ns: ids %lld  params->LDS %lld  rows,sums %lld  F1 %lld F2 %lld F3 %lld  predict %lld  criterion %lld  pred bwd %lld  "
               "B1 %lld  B2.. %lld\n", prof[10], prof[11], prof[0], prof[1], prof[2], prof[3], prof[5], prof[6], prof[7], prof[8], prof[9]);
#endif
}

struct MidSegs {
    float *dst[2 * DAISY_NEUMF_MAX_LAYERS + 1];
    int off[2 * DAISY_NEUMF_MAX_LAYERS + 2];
    int n;
};

// a workgroup takes 16 consecutive floats of the slab: 16 thread groups each adding every 16th workgroup's slab (four loads in
// flight), the groups' sums meet in LDS in group order - one fixed association per element
__global__ __launch_bounds__(kMidBlock) void k_nmf_mid_reduce(const float *__restrict__ ws, const double *__restrict__ wsd, int nb,
                                                             int slab, MidSegs segs, float *gbp, double *__restrict__ stats,
                                                             float reg_1, float reg_2, int pointwise) {
    constexpr int C = 16, G = kMidBlock / C;
    if (blockIdx.x + 1 < gridDim.x) {
        __shared__ float smr[G][C];
        const int cc = threadIdx.x % C, gg = threadIdx.x / C;
        const int c = (int)blockIdx.x * C + cc;
        const bool on = c < segs.off[segs.n];
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        if (on) {
            int sl = gg;
            for (; sl + 3 * G < nb; sl += 4 * G) {
                t0 += ws[(size_t)sl * slab + c];
                t1 += ws[(size_t)(sl + G) * slab + c];
                t2 += ws[(size_t)(sl + 2 * G) * slab + c];
                t3 += ws[(size_t)(sl + 3 * G) * slab + c];
            }
            for (; sl < nb; sl += G) t0 += ws[(size_t)sl * slab + c];
        }
        smr[gg][cc] = (t0 + t1) + (t2 + t3);
        __syncthreads();
        if (gg == 0 && on) {
            float t = smr[0][cc];
#pragma unroll
            for (int k = 1; k < G; ++k) t += smr[k][cc];
            float *o = nullptr;
            for (int k = 0; k < segs.n; ++k)
                if (c >= segs.off[k] && c < segs.off[k + 1]) o = segs.dst[k] + (c - segs.off[k]);
            if (o) *o += t;
        }
        return;
    }
    // the workgroups' twelve doubles each, 256 slabs at a time through LDS, added in workgroup order
    __shared__ double sd[kMidDoubles][kMidBlock];
    __shared__ double tot[kMidDoubles];
    double t = 0.0;
    for (int b0 = 0; b0 < nb; b0 += kMidBlock) {
        const int b = b0 + (int)threadIdx.x;
        for (int k = 0; k < kMidDoubles; ++k) sd[k][threadIdx.x] = (b < nb) ? wsd[(size_t)b * kMidDoubles + k] : 0.0;
        __syncthreads();
        if (threadIdx.x < kMidDoubles) {
            const int n = (nb - b0 < kMidBlock) ? nb - b0 : kMidBlock;
            for (int q = 0; q < n; ++q) t += sd[threadIdx.x][q];
        }
        __syncthreads();
    }
    if (threadIdx.x < kMidDoubles) tot[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x) return;
    // (this path's step never zeroes `stats`: every slot is written here)
    stats[DAISY_NST_LOSS_DATA] = tot[0];
    gbp[0] += (float)tot[1];
    double l1 = 0.0, fro = 0.0;       // NeuMFRecommender.py:149-167: the negative item's GMF rows enter twice (k_nmf_finalize)
    for (int k = 0; k < 5; ++k) {
        stats[DAISY_NST_L1 + k] = tot[2 + k];
        stats[DAISY_NST_SQ + k] = tot[7 + k];
        const double n = sqrt(tot[7 + k]);
        stats[DAISY_NST_NORM + k] = n;
        const double w = (k == 4) ? (pointwise ? 0.0 : 2.0) : 1.0;
        l1 += w * tot[2 + k];
        fro += w * n;
    }
    const double loss = tot[0] + (double)reg_1 * l1 + (double)reg_2 * fro;
    stats[DAISY_NST_LOSS] = loss;
    stats[DAISY_NST_LOSS_SUM] += loss;
}

int neumf_mid_step(const MidArgs &args, float *const *gW, float *const *gb, float *gWp, float *gbp, double *stats, float reg_1,
                   float reg_2, hipStream_t s) {
    const MidLayout y = mid_layout(args.L, args.width, args.d);
    const int TP = args.pointwise ? kMidRows : kMidRows / 2;
    const int nb = (args.B + TP - 1) / TP;
    static bool attr_set = false;
    if (!attr_set) {
        DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_nmf_mid<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMidLdsBytes));
        DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_nmf_mid<16>), hipFuncAttributeMaxDynamicSharedMemorySize, kMidLdsBytes));
        attr_set = true;
    }
    double *wsd = reinterpret_cast<double *>(args.ws + (((size_t)nb * y.slab + 3) / 4) * 4);
    MidCopy cp{};
    auto seg = [&](const float *src, int n4, int q4, int dst, int pitch) {
        cp.src[cp.n] = src; cp.start[cp.n + 1] = cp.start[cp.n] + n4; cp.q4[cp.n] = q4; cp.dst[cp.n] = dst; cp.pitch[cp.n] = pitch;
        cp.magic[cp.n] = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)q4 - 1) / (uint64_t)q4);
        ++cp.n;
    };
    for (int l = 1; l <= args.L; ++l) seg(args.W[l - 1], args.width[l] * args.width[l - 1] / 4, args.width[l - 1] / 4, y.offW[l - 1], args.width[l - 1] + 4);
    for (int l = 1; l <= args.L; ++l) seg(args.b[l - 1], args.width[l] / 4, args.width[l] / 4, y.offB[l - 1], 0);
    seg(args.Wp, (args.d + args.width[args.L]) / 4, (args.d + args.width[args.L]) / 4, y.offWp, 0);
    const size_t lds = (size_t)y.lds_floats * sizeof(float);
    if (cp.start[cp.n] <= 8 * kMidBlock) hipLaunchKernelGGL((k_nmf_mid<8>), dim3((unsigned)nb), dim3(kMidBlock), lds, s, args, y, cp, wsd);
    else hipLaunchKernelGGL((k_nmf_mid<16>), dim3((unsigned)nb), dim3(kMidBlock), lds, s, args, y, cp, wsd);
    MidSegs segs{};
    int n = 0;
    for (int l = 1; l <= args.L; ++l) { segs.dst[n] = gW[l - 1]; segs.off[n] = y.slabW[l - 1]; ++n; }
    for (int l = 1; l <= args.L; ++l) { segs.dst[n] = gb[l - 1]; segs.off[n] = y.slabB[l - 1]; ++n; }
    segs.dst[n] = gWp; segs.off[n] = y.slabWp; ++n;
    segs.off[n] = y.slabWp + args.d + args.width[args.L];
    segs.n = n;
    hipLaunchKernelGGL(k_nmf_mid_reduce, dim3((unsigned)((segs.off[n] + 15) / 16 + 1)), dim3(kMidBlock), 0, s,
                       args.ws, wsd, nb, y.slab, segs, gbp, stats, reg_1, reg_2, args.pointwise);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // namespace daisy
