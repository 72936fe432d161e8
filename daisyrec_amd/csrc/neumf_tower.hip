// NeuMF's MLP tower above the first layer as ONE persistent kernel (round 6; NeuMFRecommender.py:58-71,118-137,139-169).
//
// With the first layer run through the embedding tables (csrc/neumf.hip "FACT": x1[r] = relu(T_u[user] + T_i[item] + b1)),
// what is left of a training step's tower is, per row r of the step,
//     x2 = relu(W2 x1 + b2),  x3 = relu(W3 x2 + b3),  pred = <Wp, [uG[user]*iG[item] | x3]> + bp,
//     the criterion on (pos, neg) row pairs, and back:  dZ3 = dpred Wp[d:] . [x3 > 0],  gW3 += dZ3^T x2,
//     dZ2 = (dZ3 W3) . [x2 > 0],  gW2 += dZ2^T x1,  dZ1 = (dZ2 W2) . [x1 > 0].
// The layer-by-layer form moved every one of those activations through HBM (2.6 GB per step at 524 288 rows, 40 launches);
// MFMA time was a twentieth of the step.  Here a workgroup (4 waves, one per SIMD, the whole register file) keeps W2 and W3
// (bf16, 80 KB) in LDS for the life of the kernel, takes tiles of 64 rows = 32 (pos, neg) pairs, gathers x1 into LDS, chains
// the layers with v_mfma_f32_32x32x16_bf16 from LDS to LDS, and accumulates gW2 / gW3 in its registers ACROSS tiles (one slab
// per workgroup at the end, summed over the workgroups in a fixed order by k_tower_reduce: two runs of a step give the same
// bits).  Per row only dZ1 (bf16, 2 N1 bytes) and dpred leave the kernel.
//
// LDS layouts.  Every activation / weight buffer is read BOTH as a k-contiguous MFMA operand (one ds_read_b128 per fragment:
// lane -> row lane % 32, 8 consecutive k) and, in the backward pass, as the transposed operand through gfx950's
// ds_read_b64_tr_b16 (lds_frag_tr2).  One copy serves both: rows are unpadded (pitch = the feature count, a multiple of 64
// dwords) and the 16-byte chunk c of row r sits at chunk c ^ swz(r), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3):
//   b128 reads  - the 16 lanes the LDS services per cycle hold 16 rows with 16 distinct swz values -> 16 distinct bank groups;
//   tr reads    - a 32-lane group addresses 4 consecutive k rows (r & 3 = 0..3 -> XOR on bits 2..3 of the chunk: four
//                 different 16-bank groups) x 4 chunks x 2 halves -> all 64 banks once.
// The 64-wide buffer (x3 / dZ3) has rows of 8 chunks: pitch 96 halfwords (48 dwords: four consecutive rows start 16 banks
// apart) and chunk ^ ((r >> 2) & 3).
// Orientation of the products: D[m][n] leaves the MFMA with lane <-> n and four consecutive m per register quad, so the
// feature index is put on m and the row index on n: a lane then holds 4 consecutive features of ONE row - 8 bytes of bf16,
// one ds_write_b64 - instead of 16 rows of one feature.
#include <stdlib.h>

#include "neumf_internal.h"

namespace daisy {

struct TwL16 { static constexpr int LPR = 16; };
constexpr int kTwRows = 64;          // rows per tile (32 sample pairs, or 64 point-wise rows)
constexpr int kTwBlock = 256;

__device__ __forceinline__ int tw_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// halfword offset of 16-byte chunk `chunk` of row `row` in a wide (>= 128 features) buffer of pitch P
template <int P>
__device__ __forceinline__ int tw_off(int row, int chunk) { return row * P + ((chunk ^ tw_swz(row)) << 3); }
// the 64-feature buffer: pitch 96
constexpr int kTwP3 = 96;
__device__ __forceinline__ int tw_off3(int row, int chunk) { return row * kTwP3 + ((chunk ^ ((row >> 2) & 3)) << 3); }

// Transposed fragments (ds_read_b64_tr_b16).  A lane's two pieces of the fragment of feature block `fb` (a multiple of 32),
// k rows 16 ks .. 16 ks + 15, sit at k rows kr = 16 ks + 8 (lane / 32) + (lane % 16) / 4 and kr + 4, features fb + 16 ((lane % 32)
// / 16) + 4 (lane % 4) ...  With swz as above, swz(kr) does not depend on ks ((kr >> 2) & 3 = 2 (lane / 32), + 1 for the second
// piece) and fb only flips bits 5.. of the offset: piece address = (lane base ^ fb) + 16 ks * pitch, so the k steps of a
// stream are immediate offsets of ONE address register pair (the first version recomputed - or, hoisted out of the tile loop
// by the compiler, spilled - some 150 addresses per tile).
struct TrBase { int lo, hi; };                 // halfword offsets of the two pieces at fb = 0, ks = 0
template <int P>
__device__ __forceinline__ TrBase tw_tr_base(int lane) {
    const int h = lane >> 5, t = (lane & 15) >> 2, k0 = 8 * h + t;
    const int cc = 2 * ((lane & 31) >> 4) + ((lane & 3) >> 1), within = 4 * (lane & 1);
    return TrBase{k0 * P + ((cc ^ ((t << 2) | (2 * h))) << 3) + within, (k0 + 4) * P + ((cc ^ ((t << 2) | (2 * h + 1))) << 3) + within};
}
__device__ __forceinline__ TrBase tw_tr_base3(int lane) {      // the 64-feature buffer (pitch 96, chunk ^ ((row >> 2) & 3))
    const int h = lane >> 5, t = (lane & 15) >> 2, k0 = 8 * h + t;
    const int cc = 2 * ((lane & 31) >> 4) + ((lane & 3) >> 1), within = 4 * (lane & 1);
    return TrBase{k0 * kTwP3 + ((cc ^ (2 * h)) << 3) + within, (k0 + 4) * kTwP3 + ((cc ^ (2 * h + 1)) << 3) + within};
}
template <int P>
__device__ __forceinline__ bf16x8 tw_frag_tr(const uint16_t *buf, TrBase b, int fb, int ks) {
    return lds_frag_tr2(buf + (b.lo ^ fb) + 16 * ks * P, buf + (b.hi ^ fb) + 16 * ks * P);
}
__device__ __forceinline__ bf16x8 tw_frag_tr3(const uint16_t *buf, TrBase b, int fb, int ks) {
    return lds_frag_tr2(buf + b.lo + fb + 16 * ks * kTwP3, buf + b.hi + fb + 16 * ks * kTwP3);
}

__device__ __forceinline__ floatx16 tw_mfma(bf16x8 a, bf16x8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float tw_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float tw_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

// per-workgroup slabs of the workspace, segment by segment ([nb][len] each), in floats
template <int D>
struct TowerWs {
    static constexpr int N1 = 4 * D, N2 = 2 * D, N3 = D;
    static constexpr int64_t kW2 = (int64_t)N2 * N1, kW3 = (int64_t)N3 * N2, kB2 = N2, kB3 = N3, kWp = 2 * D;
    static constexpr int64_t kPerBlock = kW2 + kW3 + kB2 + kB3 + kWp;
    static constexpr int kDoubles = 12;       // loss terms, gbp, L1[5], SQ[5]
};

template <int D>
__global__ __launch_bounds__(kTwBlock, 1) void k_nmf_tower(TowerArgs a, int64_t ntiles) {
    constexpr int N1 = 4 * D, N2 = 2 * D, N3 = D;
    static_assert(D == 64, "tile shapes are laid out for factors = 64 (256 -> 128 -> 64)");
    __shared__ __attribute__((aligned(16))) uint16_t W2s[N2 * N1];
    __shared__ __attribute__((aligned(16))) uint16_t W3s[N3 * N2];
    __shared__ __attribute__((aligned(16))) uint16_t X1s[kTwRows * N1];      // x1, then dZ1 in place; float scratch at the end
    __shared__ __attribute__((aligned(16))) uint16_t X2s[kTwRows * N2];      // x2, then dZ2 in place
    __shared__ __attribute__((aligned(16))) uint16_t X3s[kTwRows * kTwP3];   // x3, then dZ3 in place
    __shared__ __attribute__((aligned(16))) float b1s[N1];
    __shared__ __attribute__((aligned(16))) float b2s[N2];
    __shared__ __attribute__((aligned(16))) float b3s[N3];
    __shared__ __attribute__((aligned(16))) float wps[2 * D];
    __shared__ float pred_s[kTwRows], dpred_s[kTwRows];
    __shared__ int32_t ids_s[2][kTwRows];      // users / items of the tile's rows
    __shared__ double red_d[kTwBlock / kWave][TowerWs<D>::kDoubles];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c = lane & 31;
    const int sc = tw_swz(c);                 // (rows c, 32 + c, 64 + c ... share it: it reads bits 0..3 of the row)

    // ---- the tower's weights: HBM -> LDS once per workgroup
    // (the fp32 master weights are rounded to bf16 - nearest even, as k_to_bf16 did in launches of its own - on their way in)
    auto w_chunk = [](const float *src) {
        const float4 lo = *reinterpret_cast<const float4 *>(src), hi = *reinterpret_cast<const float4 *>(src + 4);
        return u32x4{bf16_pack2(lo.x, lo.y), bf16_pack2(lo.z, lo.w), bf16_pack2(hi.x, hi.y), bf16_pack2(hi.z, hi.w)};
    };
    for (int e = tid; e < N2 * N1 / 8; e += kTwBlock) {
        const int row = e / (N1 / 8), ch = e % (N1 / 8);
        *reinterpret_cast<u32x4 *>(W2s + tw_off<N1>(row, ch)) = w_chunk(a.W2 + (int64_t)e * 8);
    }
    for (int e = tid; e < N3 * N2 / 8; e += kTwBlock) {
        const int row = e / (N2 / 8), ch = e % (N2 / 8);
        *reinterpret_cast<u32x4 *>(W3s + tw_off<N2>(row, ch)) = w_chunk(a.W3 + (int64_t)e * 8);
    }
    for (int e = tid; e < N1; e += kTwBlock) b1s[e] = a.b1[e];
    for (int e = tid; e < N2; e += kTwBlock) b2s[e] = a.b2[e];
    for (int e = tid; e < N3; e += kTwBlock) b3s[e] = a.b3[e];
    for (int e = tid; e < 2 * D; e += kTwBlock) wps[e] = a.Wp[e];
    const float bp = a.bp[0];
    __syncthreads();

    // ---- accumulators that live across tiles
    floatx16 gW2[2][4], gW3[2];               // gW2: N1 blocks 2w, 2w+1 x N2 blocks 0..3;  gW3: N2 block w x N3 blocks 0, 1
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int i = 0; i < 16; ++i) gW3[x][i] = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) gW2[x][o][i] = 0.f;
    }
    // bias / predict-layer gradients, this thread's share: gb2 in the MFMA result layout (16 features of the lane's rows), the
    // others for the 4 columns of the predict-phase mapping (below) - 16 columns per thread cost 48 registers for the life of
    // the kernel and, with the 160 of gW2 / gW3, pushed the register allocation into scratch
    float gb2p[16], gb3p[4], gwx[4], gwg[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) gb2p[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) gb3p[k] = gwx[k] = gwg[k] = 0.f;
    float s1[5] = {0, 0, 0, 0, 0}, s2[5] = {0, 0, 0, 0, 0};
    double loss_acc = 0.0;
    float gbp_acc = 0.f;

    const int rg = tid >> 4, cq = tid & 15;   // GMF product / predict layer: rows 4 rg .. 4 rg + 3, columns 4 cq .. 4 cq + 3
    const int pairs = a.pointwise ? kTwRows : kTwRows / 2;

    // The gather.  A tile's ids are read one tile ahead (64 threads, one row each) and handed to the workgroup through LDS.
    // The table rows are then read so that every wave instruction covers whole cache lines: the wave's k-th load takes rows
    // 16 w + 2 k and 16 w + 2 k + 1, eight lanes x 16 bytes per 128-byte line - a thread reads the SAME 16-byte chunk
    // (lane & 31) of eight different rows.  (The first version gave a thread 128 contiguous bytes of one row: each of its
    // eight loads touched 64 different lines for 16 bytes apiece, 4 096 line accesses per tile and table pair - the gather ran
    // at 9 bytes per clock and CU, a third of the tile's time, whatever was done about its latency: reading the ids ahead,
    // touching the rows ahead, starting the workgroups out of step - all measured, none moved it.)
    int32_t n_user = 0, n_item = 0;           // tid < 64: ids of row tid of the NEXT tile
    auto load_ids = [&](int64_t t) {
        if (tid < kTwRows) {
            const int64_t s0_ = t * pairs;
            const int64_t bidx = a.pointwise ? s0_ + tid : s0_ + (tid & 31);
            n_user = a.u[bidx];
            n_item = (a.pointwise || tid < 32) ? a.i[bidx] : a.j[bidx];
        }
    };
    struct Rows { u32x4 va[8], vb[8]; float4 ga[4], gb[4]; };
    const int gchunk = lane & 31;             // the 16-byte chunk of a table row this thread gathers (of N1 / 8 = 32)
#ifdef DAISY_TOWER_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = clock64();
#define TW_MARK(k) { const long long now_ = clock64(); prof[k] += now_ - pt0; pt0 = now_; }
#else
#define TW_MARK(k)
#endif
    Rows rows;
    if ((int64_t)blockIdx.x < ntiles) load_ids(blockIdx.x);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // an opaque zero per tile: the address arithmetic below depends on it, so it is redone per tile (a few dozen VALU
        // instructions) instead of being hoisted out of the loop as ~100 loop-invariant registers - which spilled, and a
        // spilled address is a scratch load (hundreds of cycles, one wave per SIMD: nothing to hide it behind) per use
        int opq;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
        const int hsc = ((h ^ sc) << 3) + opq;             // k-contiguous fragments: chunk 2 ks + h of row c -> (ks << 4) ^ hsc
        const int hs3 = ((h ^ ((c >> 2) & 3)) << 3) + opq; //   ... in the 64-feature buffer
        const TrBase tb1 = {tw_tr_base<N1>(lane).lo + opq, tw_tr_base<N1>(lane).hi + opq};
        const TrBase tb2 = {tw_tr_base<N2>(lane).lo + opq, tw_tr_base<N2>(lane).hi + opq};
        const TrBase tb3 = {tw_tr_base3(lane).lo + opq, tw_tr_base3(lane).hi + opq};
        const int64_t s0 = t * pairs;
        const int64_t tn = t + gridDim.x;
        const bool more = tn < ntiles;
        const int32_t my_user = n_user, my_item = n_item;    // (tid < 64: this tile's row tid)
        if (tid < kTwRows) { ids_s[0][tid] = my_user; ids_s[1][tid] = my_item; }
        __syncthreads();
        if (more) load_ids(tn);                              // (in flight for the whole tile)

        // ================= gather -> x1 = relu(T_u[user] + T_i[item] + b1) in LDS; GMF product
        float g[4][4];                        // uG[user] * iG[item] of the predict-phase mapping
        {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 16 * w + 2 * k + h;
                rows.va[k] = *reinterpret_cast<const u32x4 *>(a.tu + (int64_t)ids_s[0][row] * N1 + gchunk * 8);
                rows.vb[k] = *reinterpret_cast<const u32x4 *>(a.ti + (int64_t)ids_s[1][row] * N1 + gchunk * 8);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                rows.ga[x] = *reinterpret_cast<const float4 *>(a.uG + (int64_t)ids_s[0][4 * rg + x] * D + 4 * cq);
                rows.gb[x] = *reinterpret_cast<const float4 *>(a.iG + (int64_t)ids_s[1][4 * rg + x] * D + 4 * cq);
            }
            if (tid < kTwRows && (a.pointwise || tid < 32)) {       // the MLP rows' share of the regulariser sums, from the per-row table
                const float2 nu = a.nu[my_user], ni = a.ni[my_item];
                s1[1] += nu.x; s2[1] += nu.y; s1[3] += ni.x; s2[3] += ni.y;
            }
            const float4 c0 = *reinterpret_cast<const float4 *>(b1s + gchunk * 8), c1 = *reinterpret_cast<const float4 *>(b1s + gchunk * 8 + 4);
            const float cb[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int row = 16 * w + 2 * x + h;
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z0 = fmaxf((tw_lo(rows.va[x][k]) + tw_lo(rows.vb[x][k])) + cb[2 * k], 0.f);
                    const float z1 = fmaxf((tw_hi(rows.va[x][k]) + tw_hi(rows.vb[x][k])) + cb[2 * k + 1], 0.f);
                    o[k] = bf16_pack2(z0, z1);
                }
                *reinterpret_cast<u32x4 *>(X1s + tw_off<N1>(row, gchunk) + opq) = u32x4{o[0], o[1], o[2], o[3]};
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float av[4] = {rows.ga[x].x, rows.ga[x].y, rows.ga[x].z, rows.ga[x].w};
                const float bv[4] = {rows.gb[x].x, rows.gb[x].y, rows.gb[x].z, rows.gb[x].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g[x][k] = av[k] * bv[k];
                    if (a.pointwise || 4 * rg + x < 32) {
                        s1[0] += fabsf(av[k]); s2[0] = fmaf(av[k], av[k], s2[0]);
                        s1[2] += fabsf(bv[k]); s2[2] = fmaf(bv[k], bv[k], s2[2]);
                    } else {
                        s1[4] += fabsf(bv[k]); s2[4] = fmaf(bv[k], bv[k], s2[4]);
                    }
                }
            }
        }
        __syncthreads();
        TW_MARK(0)

        // ================= x2 = relu(W2 x1 + b2): wave w <- features 32w .. 32w+31 of both row blocks
        {
            floatx16 acc[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
            const uint16_t *wr = W2s + (32 * w + c) * N1, *x0 = X1s + c * N1, *x1 = X1s + (32 + c) * N1;
            // fragments of k step ks + 2 are requested before the MFMAs of step ks: with one wave per SIMD the LDS latency
            // is covered by this wave's own MFMAs or not at all
            bf16x8 fa[3], f0[3], f1[3];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int off = (p << 4) ^ hsc;
                fa[p] = *reinterpret_cast<const bf16x8 *>(wr + off);
                f0[p] = *reinterpret_cast<const bf16x8 *>(x0 + off);
                f1[p] = *reinterpret_cast<const bf16x8 *>(x1 + off);
            }
#pragma unroll
            for (int ks = 0; ks < N1 / 16; ++ks) {
                if (ks + 2 < N1 / 16) {
                    const int off = ((ks + 2) << 4) ^ hsc;
                    fa[(ks + 2) % 3] = *reinterpret_cast<const bf16x8 *>(wr + off);
                    f0[(ks + 2) % 3] = *reinterpret_cast<const bf16x8 *>(x0 + off);
                    f1[(ks + 2) % 3] = *reinterpret_cast<const bf16x8 *>(x1 + off);
                }
                acc[0] = tw_mfma(fa[ks % 3], f0[ks % 3], acc[0]);
                acc[1] = tw_mfma(fa[ks % 3], f1[ks % 3], acc[1]);
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int f = 32 * w + 8 * qd + 4 * h, row = 32 * rb + c;
                    const float4 bb = *reinterpret_cast<const float4 *>(b2s + f);
                    const uint2 o = make_uint2(bf16_pack2(fmaxf(acc[rb][4 * qd] + bb.x, 0.f), fmaxf(acc[rb][4 * qd + 1] + bb.y, 0.f)),
                                               bf16_pack2(fmaxf(acc[rb][4 * qd + 2] + bb.z, 0.f), fmaxf(acc[rb][4 * qd + 3] + bb.w, 0.f)));
                    *reinterpret_cast<uint2 *>(X2s + tw_off<N2>(row, f >> 3) + 4 * h + opq) = o;
                }
        }
        __syncthreads();
        TW_MARK(1)

        // ================= x3 = relu(W3 x2 + b3): wave w <- feature block w / 2, row block w % 2
        {
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int fb = w >> 1, rb = w & 1;
            const uint16_t *wr = W3s + (32 * fb + c) * N2, *xr = X2s + (32 * rb + c) * N2;
            bf16x8 fa[N2 / 16], fx[N2 / 16];                 // 8 k steps: all fragments up front (64 registers)
#pragma unroll
            for (int ks = 0; ks < N2 / 16; ++ks) {
                const int off = (ks << 4) ^ hsc;
                fa[ks] = *reinterpret_cast<const bf16x8 *>(wr + off);
                fx[ks] = *reinterpret_cast<const bf16x8 *>(xr + off);
            }
#pragma unroll
            for (int ks = 0; ks < N2 / 16; ++ks) acc = tw_mfma(fa[ks], fx[ks], acc);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int f = 32 * fb + 8 * qd + 4 * h, row = 32 * rb + c;
                const float4 bb = *reinterpret_cast<const float4 *>(b3s + f);
                const uint2 o = make_uint2(bf16_pack2(fmaxf(acc[4 * qd] + bb.x, 0.f), fmaxf(acc[4 * qd + 1] + bb.y, 0.f)),
                                           bf16_pack2(fmaxf(acc[4 * qd + 2] + bb.z, 0.f), fmaxf(acc[4 * qd + 3] + bb.w, 0.f)));
                *reinterpret_cast<uint2 *>(X3s + tw_off3(row, f >> 3) + 4 * h + opq) = o;
            }
        }
        __syncthreads();
        TW_MARK(2)

        // ================= predict layer, criterion, dZ3 (thread (rg, cq): rows 4 rg .., columns 4 cq .. of x3 and of the GMF product)
        float x3v[4][4];
        {
            const float4 wg = *reinterpret_cast<const float4 *>(wps + 4 * cq), wx = *reinterpret_cast<const float4 *>(wps + D + 4 * cq);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int rr = 4 * rg + x;
                const uint2 v = *reinterpret_cast<const uint2 *>(X3s + tw_off3(rr, cq >> 1) + 4 * (cq & 1) + opq);
                x3v[x][0] = tw_lo(v.x); x3v[x][1] = tw_hi(v.x); x3v[x][2] = tw_lo(v.y); x3v[x][3] = tw_hi(v.y);
                float sx = fmaf(wg.x, g[x][0], fmaf(wg.y, g[x][1], fmaf(wg.z, g[x][2], wg.w * g[x][3])));
                sx += fmaf(wx.x, x3v[x][0], fmaf(wx.y, x3v[x][1], fmaf(wx.z, x3v[x][2], wx.w * x3v[x][3])));
                sx = group_sum<TwL16>(sx);                      // the 16 threads of a row group (DPP: no LDS crossbar)
                if (cq == 0) pred_s[rr] = sx + bp;
            }
        }
        __syncthreads();
        if (tid < pairs) {
            const int64_t b = s0 + tid;
            float term, cp, cn;
            if (a.pointwise) {
                pair_coef(a.loss_type, pred_s[tid], (float)a.j[b], a.gamma, term, cp, cn);
                dpred_s[tid] = cp;
                a.dpred[b] = cp;
            } else {
                pair_coef(a.loss_type, pred_s[tid], pred_s[32 + tid], a.gamma, term, cp, cn);
                dpred_s[tid] = cp; dpred_s[32 + tid] = cn;
                a.dpred[b] = cp; a.dpred[a.B + b] = cn;
            }
            loss_acc += (double)term;
            gbp_acc += cp + cn;               // paired per sample: exactly 0 under BPR / HL, as in the reference's autograd
        }
        __syncthreads();
        {
            const float4 wx = *reinterpret_cast<const float4 *>(wps + D + 4 * cq);
            const float wv[4] = {wx.x, wx.y, wx.z, wx.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int rr = 4 * rg + x;
                const float dp = dpred_s[rr];
                float z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gwx[k] = fmaf(dp, x3v[x][k], gwx[k]);
                    gwg[k] = fmaf(dp, g[x][k], gwg[k]);
                    z[k] = (x3v[x][k] > 0.f) ? dp * wv[k] : 0.f;
                }
                const uint2 o = make_uint2(bf16_pack2(z[0], z[1]), bf16_pack2(z[2], z[3]));
                gb3p[0] += tw_lo(o.x); gb3p[1] += tw_hi(o.x); gb3p[2] += tw_lo(o.y); gb3p[3] += tw_hi(o.y);   // (the stored, rounded gradient)
                *reinterpret_cast<uint2 *>(X3s + tw_off3(rr, cq >> 1) + 4 * (cq & 1) + opq) = o;
            }
        }
        __syncthreads();
        TW_MARK(3)

        // ================= layer 3 backward: gW3 += dZ3^T x2 (k = the tile's rows);  dZ2 = (dZ3 W3) . [x2 > 0]
        {
            floatx16 acc[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
            {                                                 // D[m = x2 feature of block w][n = dZ3 feature]
                bf16x8 fa[kTwRows / 16], fb0[kTwRows / 16], fb1[kTwRows / 16];
#pragma unroll
                for (int ks = 0; ks < kTwRows / 16; ++ks) {
                    fa[ks] = tw_frag_tr<N2>(X2s, tb2, 32 * w, ks);
                    fb0[ks] = tw_frag_tr3(X3s, tb3, 0, ks);
                    fb1[ks] = tw_frag_tr3(X3s, tb3, 32, ks);
                }
#pragma unroll
                for (int ks = 0; ks < kTwRows / 16; ++ks) {
                    gW3[0] = tw_mfma(fa[ks], fb0[ks], gW3[0]);
                    gW3[1] = tw_mfma(fa[ks], fb1[ks], gW3[1]);
                }
            }
            {                                                 // D[m = z2 feature of block w][n = row]: k = z3 feature
                bf16x8 fa[N3 / 16], f0[N3 / 16], f1[N3 / 16];
#pragma unroll
                for (int ks = 0; ks < N3 / 16; ++ks) {
                    fa[ks] = tw_frag_tr<N2>(W3s, tb2, 32 * w, ks);
                    const int off = (ks << 4) ^ hs3;
                    f0[ks] = *reinterpret_cast<const bf16x8 *>(X3s + c * kTwP3 + off);
                    f1[ks] = *reinterpret_cast<const bf16x8 *>(X3s + (32 + c) * kTwP3 + off);
                }
#pragma unroll
                for (int ks = 0; ks < N3 / 16; ++ks) {
                    acc[0] = tw_mfma(fa[ks], f0[ks], acc[0]);
                    acc[1] = tw_mfma(fa[ks], f1[ks], acc[1]);
                }
            }
            __syncthreads();                                  // every wave is done with x2 as an operand: gate in place
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int f = 32 * w + 8 * qd + 4 * h, row = 32 * rb + c;
                    uint2 *px = reinterpret_cast<uint2 *>(X2s + tw_off<N2>(row, f >> 3) + 4 * h + opq);
                    const uint2 xv = *px;
                    const float z0 = bf16_positive((uint16_t)xv.x) ? acc[rb][4 * qd] : 0.f;
                    const float z1 = bf16_positive((uint16_t)(xv.x >> 16)) ? acc[rb][4 * qd + 1] : 0.f;
                    const float z2 = bf16_positive((uint16_t)xv.y) ? acc[rb][4 * qd + 2] : 0.f;
                    const float z3 = bf16_positive((uint16_t)(xv.y >> 16)) ? acc[rb][4 * qd + 3] : 0.f;
                    const uint2 o = make_uint2(bf16_pack2(z0, z1), bf16_pack2(z2, z3));
                    gb2p[4 * qd] += tw_lo(o.x); gb2p[4 * qd + 1] += tw_hi(o.x);
                    gb2p[4 * qd + 2] += tw_lo(o.y); gb2p[4 * qd + 3] += tw_hi(o.y);
                    *px = o;
                }
        }
        __syncthreads();
        TW_MARK(4)

        // ================= layer 2 backward: gW2 += dZ2^T x1;  dZ1 = (dZ2 W2) . [x1 > 0]
        {
            {                                                 // D[m = x1 feature of blocks 2w, 2w+1][n = z2 feature]
#pragma unroll
                for (int ks = 0; ks < kTwRows / 16; ++ks) {
                    const bf16x8 fa0 = tw_frag_tr<N1>(X1s, tb1, 32 * (2 * w), ks), fa1 = tw_frag_tr<N1>(X1s, tb1, 32 * (2 * w + 1), ks);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const bf16x8 fbb = tw_frag_tr<N2>(X2s, tb2, 32 * o, ks);
                        gW2[0][o] = tw_mfma(fa0, fbb, gW2[0][o]);
                        gW2[1][o] = tw_mfma(fa1, fbb, gW2[1][o]);
                    }
                }
            }
            // D[m = z1 feature][n = row], k = z2 feature: the wave's two feature blocks one after the other (32 accumulator
            // registers at a time; the dZ2 fragments are read twice - LDS bandwidth is not what this kernel lacks, registers are)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                floatx16 acc[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
                const int fbx = 32 * (2 * w + x);
#pragma unroll
                for (int ks = 0; ks < N2 / 16; ++ks) {
                    const bf16x8 fa = tw_frag_tr<N1>(W2s, tb1, fbx, ks);
                    const int off = (ks << 4) ^ hsc;
                    const bf16x8 f0 = *reinterpret_cast<const bf16x8 *>(X2s + c * N2 + off);
                    const bf16x8 f1 = *reinterpret_cast<const bf16x8 *>(X2s + (32 + c) * N2 + off);
                    acc[0] = tw_mfma(fa, f0, acc[0]);
                    acc[1] = tw_mfma(fa, f1, acc[1]);
                }
                // dZ1 leaves from the registers: a lane holds 4 consecutive features (8 bytes) of its row; the gate x1 > 0 is
                // read from LDS, which nobody writes before the barrier below
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int row = 32 * rb + c;
                    // (rows of the tile -> rows of the step: the lane's own row, not the gather thread's)
                    const int64_t gr = a.pointwise ? s0 + row : ((row < 32) ? s0 + row : a.B + s0 + (row - 32));
                    uint16_t *drow = a.dZ1 + gr * N1;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int f = fbx + 8 * qd + 4 * h;
                        const uint2 xv = *reinterpret_cast<const uint2 *>(X1s + tw_off<N1>(row, f >> 3) + 4 * h + opq);
                        const float z0 = bf16_positive((uint16_t)xv.x) ? acc[rb][4 * qd] : 0.f;
                        const float z1 = bf16_positive((uint16_t)(xv.x >> 16)) ? acc[rb][4 * qd + 1] : 0.f;
                        const float z2 = bf16_positive((uint16_t)xv.y) ? acc[rb][4 * qd + 2] : 0.f;
                        const float z3 = bf16_positive((uint16_t)(xv.y >> 16)) ? acc[rb][4 * qd + 3] : 0.f;
                        *reinterpret_cast<uint2 *>(drow + f) = make_uint2(bf16_pack2(z0, z1), bf16_pack2(z2, z3));
                    }
                }
            }
        }
        __syncthreads();                                      // x1 / x2 / x3 are free for the next tile
        TW_MARK(5)
    }
#ifdef DAISY_TOWER_PROF
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
        printf("tower wg %d cycles: x1->LDS %lld  F2 %lld  F3 %lld  predict %lld  B3 %lld  B2+out %lld\n", (int)blockIdx.x,
               prof[0], prof[1], prof[2], prof[3], prof[4], prof[5]);
#endif

    // ================= the workgroup's sums -> its slabs of the workspace
    using WS = TowerWs<D>;
    const int64_t nb = gridDim.x, blk = blockIdx.x;
    float *wsW2 = a.ws + blk * WS::kW2;
    float *wsW3 = a.ws + nb * WS::kW2 + blk * WS::kW3;
    float *wsB2 = a.ws + nb * (WS::kW2 + WS::kW3) + blk * WS::kB2;
    float *wsB3 = a.ws + nb * (WS::kW2 + WS::kW3 + WS::kB2) + blk * WS::kB3;
    float *wsWp = a.ws + nb * (WS::kW2 + WS::kW3 + WS::kB2 + WS::kB3) + blk * WS::kWp;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)                    // gW2[z2 feature 32o + c][x1 feature ...]
                *reinterpret_cast<float4 *>(wsW2 + (int64_t)(32 * o + c) * N1 + 32 * (2 * w + x) + 8 * qd + 4 * h) =
                    make_float4(gW2[x][o][4 * qd], gW2[x][o][4 * qd + 1], gW2[x][o][4 * qd + 2], gW2[x][o][4 * qd + 3]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)                        // gW3[z3 feature 32o + c][x2 feature 32w + ...]
            *reinterpret_cast<float4 *>(wsW3 + (int64_t)(32 * o + c) * N2 + 32 * w + 8 * qd + 4 * h) =
                make_float4(gW3[o][4 * qd], gW3[o][4 * qd + 1], gW3[o][4 * qd + 2], gW3[o][4 * qd + 3]);
    // gb2: a lane holds 16 features' sums over its own rows; the 32 lanes of a half-wave hold the other rows
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v = gb2p[k];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
        gb2p[k] = v;
    }
    if (c == 0)
#pragma unroll
        for (int k = 0; k < 16; ++k) wsB2[32 * w + 8 * (k >> 2) + 4 * h + (k & 3)] = gb2p[k];
    // gb3 / gWp: thread (rg, cq) holds the sums of columns 4 cq .. 4 cq + 3 over its row groups' rows: the 16 row groups meet in
    // LDS, added in row-group order
    float *scr = reinterpret_cast<float *>(X1s);              // 256 x 4 floats (the tile loop is over)
    auto colsum = [&](const float (&v)[4], float *dst) {
        __syncthreads();
        *reinterpret_cast<float4 *>(scr + tid * 4) = make_float4(v[0], v[1], v[2], v[3]);
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int gq = 0; gq < 16; ++gq) t += scr[(gq * 16 + (tid >> 2)) * 4 + (tid & 3)];
            dst[tid] = t;
        }
    };
    colsum(gb3p, wsB3);
    colsum(gwg, wsWp);
    colsum(gwx, wsWp + D);
    {
        double v[WS::kDoubles];
        v[0] = loss_acc; v[1] = (double)gbp_acc;
#pragma unroll
        for (int k = 0; k < 5; ++k) { v[2 + k] = (double)s1[k]; v[7 + k] = (double)s2[k]; }
#pragma unroll
        for (int k = 0; k < WS::kDoubles; ++k) {
            const double t = wave_sum_f64(v[k]);
            if (lane == 0) red_d[w][k] = t;
        }
        __syncthreads();
        if (tid < WS::kDoubles) {
            double t = 0.0;
            for (int ww = 0; ww < kTwBlock / kWave; ++ww) t += red_d[ww][tid];
            a.wsd[blk * WS::kDoubles + tid] = t;
        }
    }
}

// the workgroups' slabs, added in workgroup order into the gradients (+=), and the step's statistics: the criterion sum,
// the regulariser sums, their norms and NeuMF.calc_loss's value (what k_nmf_loss's atomics and k_nmf_finalize did)
template <int D>
__global__ __launch_bounds__(kTwBlock) void k_nmf_tower_reduce(const float *__restrict__ ws, const double *__restrict__ wsd,
                                                               int nb, float *gW2, float *gW3, float *gb2, float *gb3,
                                                               float *gWp, float *gbp, double *__restrict__ stats,
                                                               float reg_1, float reg_2, int pointwise) {
    using WS = TowerWs<D>;
    // a workgroup takes 64 consecutive floats of the concatenated segments: 16 lanes x float4, 16 thread groups each adding
    // every 16th workgroup's slab (8 loads in flight per thread; the first version kept 2 in flight: 54 us for 42 MB); the
    // groups' sums meet in LDS in group order - one fixed association per element
    constexpr int C = 64, G = kTwBlock / (C / 4);
    __shared__ float4 sm[G][C / 4];
    const int64_t nblk_f = WS::kPerBlock / C;                 // (every segment is a multiple of 64 floats)
    if ((int64_t)blockIdx.x < nblk_f) {
        int64_t e0 = (int64_t)blockIdx.x * C;
        const float *src;
        float *dst;
        int64_t len;
        if (e0 < WS::kW2) { src = ws; dst = gW2; len = WS::kW2; }
        else if ((e0 -= WS::kW2) < WS::kW3) { src = ws + (int64_t)nb * WS::kW2; dst = gW3; len = WS::kW3; }
        else if ((e0 -= WS::kW3) < WS::kB2) { src = ws + (int64_t)nb * (WS::kW2 + WS::kW3); dst = gb2; len = WS::kB2; }
        else if ((e0 -= WS::kB2) < WS::kB3) { src = ws + (int64_t)nb * (WS::kW2 + WS::kW3 + WS::kB2); dst = gb3; len = WS::kB3; }
        else { e0 -= WS::kB3; src = ws + (int64_t)nb * (WS::kW2 + WS::kW3 + WS::kB2 + WS::kB3); dst = gWp; len = WS::kWp; }
        const int cc = threadIdx.x % (C / 4), gg = threadIdx.x / (C / 4);
        const float *col = src + e0 + 4 * cc;
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        int sl = gg;
        for (; sl + 7 * G < nb; sl += 8 * G) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(col + (int64_t)(sl + k * G) * len);
                t[k].x += v.x; t[k].y += v.y; t[k].z += v.z; t[k].w += v.w;
            }
        }
        for (int k = 0; sl < nb; sl += G, ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(col + (int64_t)sl * len);
            t[k & 7].x += v.x; t[k & 7].y += v.y; t[k & 7].z += v.z; t[k & 7].w += v.w;
        }
        float4 acc = t[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { acc.x += t[k].x; acc.y += t[k].y; acc.z += t[k].z; acc.w += t[k].w; }
        sm[gg][cc] = acc;
        __syncthreads();
        if (gg == 0) {
            float4 r4 = sm[0][cc];
#pragma unroll
            for (int k = 1; k < G; ++k) { const float4 v = sm[k][cc]; r4.x += v.x; r4.y += v.y; r4.z += v.z; r4.w += v.w; }
            float *o = dst + e0 + 4 * cc;             // (the caller's gradient tensors: no alignment beyond a float's is assumed)
            o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
        }
        return;
    }
    // the last workgroup: the doubles (all slabs loaded at once, then added in workgroup order)
    __shared__ double sd[WS::kDoubles];
    __shared__ double sall[kTwBlock][WS::kDoubles];
    double tot = 0.0;
    for (int b0 = 0; b0 < nb; b0 += kTwBlock) {
        const int b = b0 + (int)threadIdx.x;
        if (b < nb)
#pragma unroll
            for (int k = 0; k < WS::kDoubles; ++k) sall[threadIdx.x][k] = wsd[(int64_t)b * WS::kDoubles + k];
        __syncthreads();
        if (threadIdx.x < WS::kDoubles) {
            const int n = (nb - b0 < kTwBlock) ? nb - b0 : kTwBlock;
            for (int b2 = 0; b2 < n; ++b2) tot += sall[b2][threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x < WS::kDoubles) sd[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every slot of the step's statistics is written here: this path's step does not zero them first)
        stats[DAISY_NST_LOSS_DATA] = sd[0];
        gbp[0] += (float)sd[1];
        double l1 = 0.0, fro = 0.0;
        for (int k = 0; k < 5; ++k) {
            const double L1 = sd[2 + k], SQ = sd[7 + k];
            stats[DAISY_NST_L1 + k] = L1;
            stats[DAISY_NST_SQ + k] = SQ;
            const double n = sqrt(SQ);
            stats[DAISY_NST_NORM + k] = n;
            const double wgt = (k == 4) ? (pointwise ? 0.0 : 2.0) : 1.0;     // NeuMFRecommender.py:158-161: iG[j] twice
            l1 += wgt * L1;
            fro += wgt * n;
        }
        const double loss = stats[DAISY_NST_LOSS_DATA] + (double)reg_1 * l1 + (double)reg_2 * fro;
        stats[DAISY_NST_LOSS] = loss;
        stats[DAISY_NST_LOSS_SUM] += loss;
    }
}

size_t neumf_tower_ws_bytes(int d, int nblocks) {
    if (d != 64) return 0;
    using WS = TowerWs<64>;
    return (size_t)nblocks * ((size_t)WS::kPerBlock * sizeof(float) + WS::kDoubles * sizeof(double));
}

int neumf_tower_blocks(int64_t rows_or_pairs_tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) cus = 256;
        else cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return (int)(rows_or_pairs_tiles < cus ? rows_or_pairs_tiles : cus);
}

int neumf_tower_step(const TowerArgs &args, int d, int64_t R, float *gW2, float *gW3, float *gb2, float *gb3, float *gWp,
                     float *gbp, double *stats, float reg_1, float reg_2, hipStream_t s) {
    if (d != 64 || R % kTwRows != 0 || R <= 0) {
        set_error("neumf tower: factors=%d rows=%lld do not tile (factors 64, rows %% 64 == 0)", d, (long long)R);
        return DAISY_ERR_STATE;
    }
    using WS = TowerWs<64>;
    const int64_t ntiles = R / kTwRows;
    const int nb = neumf_tower_blocks(ntiles);
    TowerArgs a = args;
    a.wsd = reinterpret_cast<double *>(a.ws + (size_t)nb * WS::kPerBlock);
    hipLaunchKernelGGL((k_nmf_tower<64>), dim3(nb), dim3(kTwBlock), 0, s, a, ntiles);
    DAISY_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_nmf_tower_reduce<64>), dim3((unsigned)(WS::kPerBlock / 64 + 1)), dim3(kTwBlock), 0, s, a.ws, a.wsd, nb,
                       gW2, gW3, gb2, gb3, gWp, gbp, stats, reg_1, reg_2, args.pointwise);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // namespace daisy
