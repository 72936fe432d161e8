// NeuMF (daisy/model/NeuMFRecommender.py) on gfx950.
//
//   forward   k_nmf_gather      x0 = [uM[u] | iM[item]] (dropout of layer 1 applied), g = uG[u]*iG[item],
//                               regulariser sums of the gathered rows (training only)
//             k_gemm<EPI_BIAS_RELU>   x_l = ReLU(x_{l-1} W_l^T + b_l) (* dropout mask of layer l+1):
//                               fp32 MFMA 32x32x2 tiles, LDS-staged, register-prefetched
//             k_nmf_predict     pred = <Wp, [g | x_L]> + bp
//   loss      k_nmf_loss        criterion epilogue shared with MF (pair_coef) -> d loss / d pred
//             k_nmf_finalize    norms + NeuMF.calc_loss value
//   backward  k_nmf_pred_bwd    dZ_L = dpred * Wp[mlp part] gated by x_L > 0;  gWp, gbp
//             k_gemm<EPI_ATOMIC>      gW_l += dZ_l^T x_{l-1}   (reduction over the batch rows, split over blocks)
//             k_colsum          gb_l += column sums of dZ_l
//             k_gemm<EPI_GATE>  dZ_{l-1} = (dZ_l W_l) gated by x_{l-1} > 0 (layer 1: dropout mask of x0)
//             k_nmf_scatter     embedding gradients (fp32 atomics into the dense gradient tables) +
//                               the regulariser gradients exactly as NeuMFRecommender.py:149-167 lists them
// Dropout: x_{l-1} is stored already masked and scaled, so "x > 0" carries mask and ReLU gate at once.
#include <stdlib.h>

#include "common.h"
#include "neumf_internal.h"

#include <type_traits>

#ifndef DAISY_BKH
#define DAISY_BKH 32        // k depth of the bf16-storage GEMM's tiles (32 or 64; -DDAISY_BKH=64 to try the other)
#endif

namespace daisy {


constexpr int kBK = 16;        // k depth of an LDS tile
constexpr int kGemmBM = 128;   // block tile rows (2 x 2 waves, each 64 rows)
constexpr int kLdsPad = 4;

enum { EPI_STORE = 0, EPI_BIAS_RELU = 1, EPI_GATE = 2, EPI_ATOMIC = 3 };

struct GemmOp {
    const float *A; int64_t sam, sak;     // A(m,k) = A[m*sam + k*sak]
    const float *B; int64_t sbn, sbk;     // B(n,k) = B[n*sbn + k*sbk]
    float *C; int64_t ldc, scn;           // C(m,n) = C[m*ldc + n*scn]   (scn = 0 means 1)
    int64_t M; int N; int64_t K;
    const float *bias;                    // EPI_BIAS_RELU
    const float *gate; int64_t ldg;       // EPI_GATE: out = acc * (gate(m,n) > 0 ? gate_scale : 0)
    float gate_scale;
    uint32_t drop_thresh, drop_stream;    // dropout on output element (m,n), idx = m*N + n; thresh 0: off
    float drop_scale;
    uint64_t drop_seed;
    int64_t k_chunk;                      // reduction range per blockIdx.z
    int64_t slice_stride;                 // EPI_ATOMIC: != 0 - slice z STORES its partial product at C + z * slice_stride
                                          // (summed in slice order by k_reduce_slices: reproducible); 0 - fp32 atomics into C
    int vec_a, vec_b;                     // set by launch_gemm: operand qualifies for the float4 path
    int bf16;                             // throughput mode: bf16-input MFMA where the tile shape allows it
    // bf16 STORAGE (precision level 2: activations and a copy of the weights live as bf16 in HBM): when A16 is set the
    // operands are read through A16 / B16 (same strides, in elements), the gate through G16, and the result goes to
    // C16 (EPI_BIAS_RELU / EPI_GATE) or, in fp32, to C (EPI_ATOMIC)
    const uint16_t *A16, *B16, *G16;
    uint16_t *C16;
};

template <int WN, int EPI, bool FAST, bool DROP>
__device__ __forceinline__ void gemm_epilogue(const GemmOp &op, floatx16 (&acc)[2][WN], int64_t m0, int n0, int wm,
                                              int wn, int lane, unsigned zslice) {
    // lane holds column (lane % 32), rows (i/4)*8 + (lane/32)*4 + i%4 of each 32x32 block (all MFMA
    // 32x32 shapes share this C/D map on gfx950).  epilogue: lane holds column (lane % 32), rows (i/4)*8 + (lane/32)*4 + i%4 of each 32x32 block.
    // 32-bit offsets from the tile origin (a tile spans < 2^31 elements of C: 128 rows x ldc)
    const int scn = op.scn ? (int)op.scn : 1;
    float *__restrict__ Ct = op.C + m0 * op.ldc + (int64_t)n0 * scn;
    // (the k slice this workgroup computed - k_gemm_h remaps workgroups to tiles, so that is NOT blockIdx.z there: until round 6
    // the slices of the bf16-storage weight gradients landed in the slot of blockIdx.z, two workgroups per slot whenever the
    // slice count was a multiple of 8 and an output had fewer than 8 tiles - partial products lost, unseen by tests that
    // allowed 25 %; found by the bf16 oracle at 2 %)
    if constexpr (EPI == EPI_ATOMIC) Ct += (int64_t)zslice * op.slice_stride;
    const float *__restrict__ Gt = (EPI == EPI_GATE && op.gate) ? op.gate + m0 * op.ldg + n0 : nullptr;
    const int ldc = (int)op.ldc, ldg = (int)op.ldg;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int nl = wn * 32 * WN + ni * 32 + lane % 32;
            float bias = 0.f;
            if constexpr (EPI == EPI_BIAS_RELU) bias = (FAST || n0 + nl < op.N) ? op.bias[n0 + nl] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ml = wm * 64 + mi * 32 + (i / 4) * 8 + (lane / 32) * 4 + (i % 4);
                if (!FAST && (m0 + ml >= op.M || n0 + nl >= op.N)) continue;
                float v = acc[mi][ni][i];
                if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
                if constexpr (EPI == EPI_GATE)
                    if (Gt) v = (Gt[ml * ldg + nl] > 0.f) ? v * op.gate_scale : 0.f;
                if constexpr (DROP)
                    if (op.drop_thresh)
                        v = drop_keep(op.drop_seed, op.drop_stream,
                                      (uint64_t)(m0 + ml) * (uint64_t)op.N + (uint64_t)(n0 + nl), op.drop_thresh)
                                ? v * op.drop_scale : 0.f;
                if constexpr (EPI == EPI_ATOMIC) {
                    if (op.slice_stride) Ct[ml * ldc + nl * scn] = v;
                    else unsafeAtomicAdd(Ct + ml * ldc + nl * scn, v);
                } else Ct[ml * ldc + nl * scn] = v;
            }
        }
}

// C = A * B^T-style contraction over k with arbitrary strides.  WN: 32-column MFMA blocks per wave
// (block tile = 128 x 64*WN).  Operand tiles go global -> registers -> LDS (k-major, so the MFMA
// fragment reads are conflict free; two LDS stages, one barrier per k tile) with the next tile's
// loads in flight during the MFMAs.  Interior tiles of 16-byte aligned operands take a branch-free
// float4 path (a guarded load costs a branch and a vmcnt drain each); edge tiles, k tails and
// unaligned operands take the guarded scalar path.
// FAST: every tile is interior, both operands qualify for the float4 path and the k range is a multiple
// of kBK (checked by launch_gemm) - the guarded loader and its address registers are compiled out, which
// is what lets four waves per SIMD share the MFMA pipe.
template <int WN, int EPI, bool FAST, bool DROP>
__device__ __forceinline__ void gemm_f32_tile(const GemmOp &op, unsigned bx, unsigned by, unsigned bz) {
    constexpr int BM = kGemmBM, BN = 64 * WN;
    constexpr int EA = BM * kBK / kBlock, EB = BN * kBK / kBlock;     // elements per thread per tile
    constexpr int LA = BM + kLdsPad, LB = BN + kLdsPad;
    __shared__ __attribute__((aligned(16))) float As[2][kBK * LA];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK * LB];
    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    const int wm = wave / 2, wn = wave % 2;
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;
    const int64_t k_lo = (int64_t)bz * op.k_chunk;
    const int64_t k_hi = (k_lo + op.k_chunk < op.K) ? (k_lo + op.k_chunk) : op.K;
    const bool a_kfast = (op.sak == 1), b_kfast = (op.sbk == 1);
    const bool a_vec = FAST || (op.vec_a && (m0 + BM <= op.M)), b_vec = FAST || (op.vec_b && (n0 + BN <= op.N));

    floatx16 acc[2][WN];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    float ra[EA], rb[EB];
    // one operand tile [rows x kBK] -> registers.  vec: float4 along the contiguous dimension
    auto load_op = [&](const float *__restrict__ P, int64_t srow, int64_t sk, bool kfast, bool vec, int64_t row0,
                       int64_t nrows_total, int64_t kt, auto &r, auto rows_c, auto elems_c) {
        constexpr int ROWS = decltype(rows_c)::value, E = decltype(elems_c)::value;
        if (FAST || (vec && kt + kBK <= k_hi)) {
            if (kfast) {                                   // 4 lanes cover the 16 k of one row
                const float *src = P + (row0 + tid / 4) * srow + kt + (tid % 4) * 4;
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)q * (kBlock / 4) * srow);
                    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
                }
            } else {                                       // ROWS/4 lanes cover one k
                const float *src = P + row0 + (tid % (ROWS / 4)) * 4 + (kt + tid / (ROWS / 4)) * sk;
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)q * (kBlock / (ROWS / 4)) * sk);
                    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
                }
            }
        } else if constexpr (!FAST) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int e = tid + q * kBlock;
                const int kk = kfast ? (e % kBK) : (e / ROWS);
                const int rr = kfast ? (e / kBK) : (e % ROWS);
                const int64_t row = row0 + rr, k = kt + kk;
                r[q] = (row < nrows_total && k < k_hi) ? P[row * srow + k * sk] : 0.f;
            }
        }
    };
    auto store_op = [&](float *__restrict__ S, int ld, bool kfast, bool vec, bool full, auto &r, auto rows_c,
                        auto elems_c) {
        constexpr int ROWS = decltype(rows_c)::value, E = decltype(elems_c)::value;
        if (FAST || (vec && full)) {
            if (kfast) {
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const int row = tid / 4 + q * (kBlock / 4), k = (tid % 4) * 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) S[(k + t) * ld + row] = r[4 * q + t];
                }
            } else {
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const int row = (tid % (ROWS / 4)) * 4, k = tid / (ROWS / 4) + q * (kBlock / (ROWS / 4));
                    *reinterpret_cast<float4 *>(S + k * ld + row) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
                }
            }
        } else if constexpr (!FAST) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int e = tid + q * kBlock;
                const int kk = kfast ? (e % kBK) : (e / ROWS);
                const int rr = kfast ? (e / kBK) : (e % ROWS);
                S[kk * ld + rr] = r[q];
            }
        }
    };
    using RA = std::integral_constant<int, BM>; using RB = std::integral_constant<int, BN>;
    using NA = std::integral_constant<int, EA>; using NB = std::integral_constant<int, EB>;

    if (k_lo < k_hi) {
        load_op(op.A, op.sam, op.sak, a_kfast, a_vec, m0, op.M, k_lo, ra, RA{}, NA{});
        load_op(op.B, op.sbn, op.sbk, b_kfast, b_vec, (int64_t)n0, (int64_t)op.N, k_lo, rb, RB{}, NB{});
        store_op(As[0], LA, a_kfast, a_vec, k_lo + kBK <= k_hi, ra, RA{}, NA{});
        store_op(Bs[0], LB, b_kfast, b_vec, k_lo + kBK <= k_hi, rb, RB{}, NB{});
        __syncthreads();
        int cur = 0;
        for (int64_t kt = k_lo; kt < k_hi; kt += kBK) {
            const bool more = kt + kBK < k_hi;
            if (more) {
                load_op(op.A, op.sam, op.sak, a_kfast, a_vec, m0, op.M, kt + kBK, ra, RA{}, NA{});
                load_op(op.B, op.sbn, op.sbk, b_kfast, b_vec, (int64_t)n0, (int64_t)op.N, kt + kBK, rb, RB{}, NB{});
            }
            const float *as = As[cur] + (lane / 32) * LA + wm * 64 + lane % 32;
            const float *bs = Bs[cur] + (lane / 32) * LB + wn * 32 * WN + lane % 32;
            float a[2][2], b[2][WN];          // fragments of k-step s live in slot s&1: the next step's LDS
#pragma unroll                                // reads are issued before this step's MFMAs
            for (int mi = 0; mi < 2; ++mi) a[0][mi] = as[mi * 32];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) b[0][ni] = bs[ni * 32];
#pragma unroll
            for (int ks = 0; ks < kBK / 2; ++ks) {
                const int c = ks & 1, nx = c ^ 1;
                if (ks + 1 < kBK / 2) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[nx][mi] = as[(2 * ks + 2) * LA + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) b[nx][ni] = bs[(2 * ks + 2) * LB + ni * 32];
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][mi], b[c][ni], acc[mi][ni], 0, 0, 0);
            }
            if (more) {                                    // the other stage: nobody reads it now
                const bool full = kt + 2 * kBK <= k_hi;
                store_op(As[cur ^ 1], LA, a_kfast, a_vec, full, ra, RA{}, NA{});
                store_op(Bs[cur ^ 1], LB, b_kfast, b_vec, full, rb, RB{}, NB{});
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    gemm_epilogue<WN, EPI, FAST, DROP>(op, acc, m0, n0, wm, wn, lane, bz);
}

template <int WN, int EPI, bool FAST, bool DROP>
__global__ __launch_bounds__(kBlock) void k_gemm(GemmOp op) {
    gemm_f32_tile<WN, EPI, FAST, DROP>(op, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Two independent products in ONE launch (round 6): the user side's and the item side's table products of a NeuMF step are
// each ~100 workgroups of 16 dependent k steps - latency-bound, 29-34 us per launch whatever the loader.  Side by side
// (workgroups [0, ax) take `a`, the rest `b`; k slices beyond a product's own count leave at once) the pair costs what one did.
template <int WN, int EPI>
__global__ __launch_bounds__(kBlock) void k_gemm_pair(GemmOp a, GemmOp b, unsigned ax, unsigned az, unsigned bz_n) {
    if (blockIdx.x < ax) { if (blockIdx.z < az) gemm_f32_tile<WN, EPI, false, false>(a, blockIdx.x, blockIdx.y, blockIdx.z); }
    else if (blockIdx.z < bz_n) gemm_f32_tile<WN, EPI, false, false>(b, blockIdx.x - ax, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------
// bf16-input variant (throughput mode; BASELINE configs[3] names it): operands stay fp32 in HBM, are
// rounded to bf16 (nearest-even) on their way into LDS and multiplied by v_mfma_f32_32x32x16_bf16
// (fp32 accumulate, 16x the fp32 MFMA rate).  LDS tiles are row-major with k contiguous - a lane's
// fragment is 8 consecutive k = one 16-byte read - at an 80-byte row pitch (odd multiple of 16 B:
// conflict free).  Interior, aligned tiles only (launch_gemm falls back to the fp32 kernel otherwise).
// ---------------------------------------------------------------------------------------------
constexpr int kBK16 = 32, kLdk16 = kBK16 + 8;


template <int WN, int EPI, bool DROP>
__global__ __launch_bounds__(kBlock) void k_gemm_bf16(GemmOp op) {
    constexpr int BM = kGemmBM, BN = 64 * WN;
    constexpr int QA = BM * kBK16 / kBlock / 4, QB = BN * kBK16 / kBlock / 4;      // float4 loads per thread per tile
    __shared__ __attribute__((aligned(16))) uint16_t As[2][BM * kLdk16];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][BN * kLdk16];
    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    const int wm = wave / 2, wn = wave % 2;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int64_t k_lo = (int64_t)blockIdx.z * op.k_chunk;
    const int64_t k_hi = (k_lo + op.k_chunk < op.K) ? (k_lo + op.k_chunk) : op.K;
    const bool a_kfast = (op.sak == 1), b_kfast = (op.sbk == 1);

    floatx16 acc[2][WN];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    float ra[4 * QA], rb[4 * QB];
    // k-contiguous operand: float4 along k (8 lanes cover the 32 k of a row).  Operand contiguous along
    // its rows (the weight-gradient GEMM's two operands, W in the d-input GEMM): a thread takes ONE row
    // and 4*Q consecutive k with scalar loads - each wave instruction still reads 64 consecutive rows of
    // one k, 256 contiguous bytes - so that its bf16 pack is k-contiguous and lands in LDS as 16-byte
    // writes (a float4 along the rows would have to be scattered with 2-byte stores).
    auto load_op = [&](const float *__restrict__ P, int64_t srow, int64_t sk, bool kfast, int64_t row0, int64_t kt,
                       auto &r, auto rows_c, auto q_c) {
        constexpr int ROWS = decltype(rows_c)::value, Q = decltype(q_c)::value;
        if (kfast) {
            const float *src = P + (row0 + tid / 8) * srow + kt + (tid % 8) * 4;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)q * (kBlock / 8) * srow);
                r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
            }
        } else {
            const float *src = P + row0 + (tid % ROWS) + (kt + (int64_t)(tid / ROWS) * (4 * Q)) * sk;
#pragma unroll
            for (int q = 0; q < 4 * Q; ++q) r[q] = src[(int64_t)q * sk];
        }
    };
    auto pack2 = [](float lo, float hi) { return bf16_pack2(lo, hi); };
    auto store_op = [&](uint16_t *__restrict__ S, bool kfast, auto &r, auto rows_c, auto q_c) {
        constexpr int ROWS = decltype(rows_c)::value, Q = decltype(q_c)::value;
        if (kfast) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int row = tid / 8 + q * (kBlock / 8), k = (tid % 8) * 4;
                *reinterpret_cast<uint2 *>(S + row * kLdk16 + k) =
                    make_uint2(pack2(r[4 * q], r[4 * q + 1]), pack2(r[4 * q + 2], r[4 * q + 3]));
            }
        } else {
            uint16_t *dst = S + (tid % ROWS) * kLdk16 + (tid / ROWS) * (4 * Q);
#pragma unroll
            for (int q = 0; q < Q / 2; ++q)
                *reinterpret_cast<uint4 *>(dst + 8 * q) =
                    make_uint4(pack2(r[8 * q], r[8 * q + 1]), pack2(r[8 * q + 2], r[8 * q + 3]),
                               pack2(r[8 * q + 4], r[8 * q + 5]), pack2(r[8 * q + 6], r[8 * q + 7]));
        }
    };
    using RA = std::integral_constant<int, BM>; using RB = std::integral_constant<int, BN>;
    using NA = std::integral_constant<int, QA>; using NB = std::integral_constant<int, QB>;

    if (k_lo < k_hi) {
        load_op(op.A, op.sam, op.sak, a_kfast, m0, k_lo, ra, RA{}, NA{});
        load_op(op.B, op.sbn, op.sbk, b_kfast, (int64_t)n0, k_lo, rb, RB{}, NB{});
        store_op(As[0], a_kfast, ra, RA{}, NA{});
        store_op(Bs[0], b_kfast, rb, RB{}, NB{});
        __syncthreads();
        int cur = 0;
        for (int64_t kt = k_lo; kt < k_hi; kt += kBK16) {
            const bool more = kt + kBK16 < k_hi;
            if (more) {
                load_op(op.A, op.sam, op.sak, a_kfast, m0, kt + kBK16, ra, RA{}, NA{});
                load_op(op.B, op.sbn, op.sbk, b_kfast, (int64_t)n0, kt + kBK16, rb, RB{}, NB{});
            }
            const uint16_t *as = As[cur] + (wm * 64 + lane % 32) * kLdk16 + (lane / 32) * 8;
            const uint16_t *bs = Bs[cur] + (wn * 32 * WN + lane % 32) * kLdk16 + (lane / 32) * 8;
#pragma unroll
            for (int ks = 0; ks < kBK16 / 16; ++ks) {
                bf16x8 a[2], b[WN];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a[mi] = *reinterpret_cast<const bf16x8 *>(as + mi * 32 * kLdk16 + ks * 16);
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    b[ni] = *reinterpret_cast<const bf16x8 *>(bs + ni * 32 * kLdk16 + ks * 16);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
            if (more) {
                store_op(As[cur ^ 1], a_kfast, ra, RA{}, NA{});
                store_op(Bs[cur ^ 1], b_kfast, rb, RB{}, NB{});
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    gemm_epilogue<WN, EPI, true, DROP>(op, acc, m0, n0, wm, wn, lane, blockIdx.z);
}

// bf16-STORAGE variant (precision level 2): both operands are bf16 in HBM, so a tile row of 32 k is 64 bytes -
// 4 lanes x 16 bytes, copied to LDS as they are (no conversion, half the operand bytes of the fp32-storage
// kernels above, which is what bounded them).  Operands that are contiguous along their rows instead of k (the
// weight-gradient GEMM) keep that layout in LDS and are transposed by the fragment read (lds_frag_tr below).
// Epilogues: bias + ReLU (+dropout) or gate with bf16 output (round to nearest even), or fp32 atomics (split-K).

// Which tile a workgroup computes.  Workgroups go to the 8 XCDs round-robin by their linear id (observed, used for
// speed only), and each XCD has its own L2: tiles that read the same operand panel are given to workgroups that
// land on ONE XCD next to each other in time, so the panel comes from HBM once and from that L2 afterwards.
//   one k range (forward, input gradient): the column tiles of one 128-row panel of A share it;
//   split-K (weight gradient): all tiles of one k chunk share the chunk's two panels.
struct TileId { unsigned x, y, z; };
__device__ __forceinline__ TileId tile_of_block() {
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    TileId t{blockIdx.x, blockIdx.y, blockIdx.z};
    const unsigned lin = t.x + gx * (t.y + gy * t.z), xcd = lin % 8, slot = lin / 8;
    if (gz > 1) {
        if (gz % 8 == 0) {
            const unsigned per = gx * gy, r = slot % per;
            t.z = (slot / per) * 8 + xcd; t.x = r % gx; t.y = r / gx;
        }
    } else if (gx % 8 == 0) {
        t.y = slot % gy; t.x = (slot / gy) * 8 + xcd;
    }
    return t;
}

constexpr int kBKH = DAISY_BKH, kLdkH = kBKH;          // k depth of a tile (64: 587 vs 610 TFLOP/s on the forward shape, step equal -
                                                // 64 KB of LDS per workgroup halve the resident workgroups).  k-contiguous tiles
                                                // are unpadded (64-byte rows); the four 16-byte chunks of row r sit at
                                                // position chunk ^ ((r / 4) % 4): the fragment reads (ds_read_b128: 16 rows
                                                // per LDS cycle) and the tile writes (two rows per 8-lane group) are then
                                                // both conflict-free; the first version's 80-byte pitch left 31 % of the LDS
                                                // cycles in bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
// (32-deep tiles: 4 chunks per 64-byte row, chunk ^ ((r / 4) % 4); 64-deep: 8 chunks per 128-byte row, chunk ^ ((r / 2) % 8))
__device__ __forceinline__ int swz_of_row(int row) { return kBKH == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); }
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ swz_of_row(row); }
// An operand that is contiguous along its ROWS instead of k (both operands of the weight-gradient GEMM: dZ^T and
// X^T with k = the batch row) is copied to LDS as it lies in memory - [k][row] tiles, 16-byte loads along the rows -
// and the MFMA fragment (8 consecutive k of one row per lane) comes out of gfx950's transposing LDS read:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of the [4 k][16 rows] block whose 16 four-element
// pieces the lanes address (measured: result[i][j] = piece[4j + i/4][i%4]), two of them per fragment.  The first
// version read such operands with 2-byte global loads and packed them in registers: 265 TFLOP/s on the weight
// gradients against 430-600 on the k-contiguous GEMMs.  Pitch rows + 32 halfwords: the 8 k rows one instruction
// touches fall on 4 distinct 16-bank offsets, twice - the two LDS cycles its 512 bytes need anyway.
constexpr int kPadT = 32;

template <int WN, int EPI, bool DROP, bool AK, bool BK>      // AK / BK: operand A / B is contiguous along k (else along its rows)
__global__ __launch_bounds__(kBlock) void k_gemm_h(GemmOp op) {
    constexpr int BM = kGemmBM, BN = 64 * WN;
    constexpr int LPT = kBKH / 8;                               // k-contiguous: lanes per tile row (16 bytes each)
    constexpr int RPP = kBlock / LPT;                           //               tile rows per pass of the workgroup
    constexpr int PTA = BM + kPadT, PTB = BN + kPadT;           // row-contiguous: halfwords per k row of the LDS tile
    constexpr int kTileA = AK ? BM * kLdkH : kBKH * PTA, kTileB = BK ? BN * kLdkH : kBKH * PTB;
    // one LDS block: two stages of the A and B tiles; the output tile of the bf16 epilogues reuses it afterwards
    constexpr int kStage = 2 * (kTileA + kTileB), kOut = BM * (BN + 8);
    __shared__ __attribute__((aligned(16))) uint16_t smem[kStage > kOut ? kStage : kOut];
    uint16_t *const As0 = smem, *const Bs0 = smem + 2 * kTileA;
    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    const int wm = wave / 2, wn = wave % 2;
    const TileId tile = tile_of_block();
    const int64_t m0 = (int64_t)tile.x * BM;
    const int n0 = tile.y * BN;
    const int64_t k_lo = (int64_t)tile.z * op.k_chunk;
    const int64_t k_hi = (k_lo + op.k_chunk < op.K) ? (k_lo + op.k_chunk) : op.K;

    floatx16 acc[2][WN];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    // registers of one tile: 16 bytes per load in both layouts (BM*32*2 / 256 threads = 2 loads for 128 rows)
    // (clang's own vector type: arrays of HIP's uint4 class are not split into registers and went through scratch)
    constexpr int NQ = BM * kBKH / 8 / kBlock;
    u32x4 ra[NQ], rb[NQ];
    auto load_op = [&](const uint16_t *__restrict__ P, int64_t srow, int64_t sk, auto kfast_c, int64_t row0, int64_t kt,
                       auto &r, auto rows_c) {
        constexpr int ROWS = decltype(rows_c)::value;
        if constexpr (decltype(kfast_c)::value) {
            const uint16_t *src = P + (row0 + tid / LPT) * srow + kt + (tid % LPT) * 8;
#pragma unroll
            for (int q = 0; q < ROWS / RPP; ++q) r[q] = *reinterpret_cast<const u32x4 *>(src + (int64_t)q * RPP * srow);
        } else {
            constexpr int VPR = ROWS / 8, KPP = kBlock / VPR;      // 16-byte vectors per k row; k rows per pass
            const uint16_t *src = P + row0 + (tid % VPR) * 8 + (kt + tid / VPR) * sk;
#pragma unroll
            for (int q = 0; q < kBKH / KPP; ++q) r[q] = *reinterpret_cast<const u32x4 *>(src + (int64_t)q * KPP * sk);
        }
    };
    auto store_op = [&](uint16_t *__restrict__ S, auto kfast_c, const auto &r, auto rows_c) {
        constexpr int ROWS = decltype(rows_c)::value;
        if constexpr (decltype(kfast_c)::value) {
#pragma unroll
            for (int q = 0; q < ROWS / RPP; ++q) {
                const int row = tid / LPT + q * RPP;
                *reinterpret_cast<u32x4 *>(S + row * kLdkH + swz_chunk(row, tid % LPT) * 8) = r[q];
            }
        } else {
            constexpr int VPR = ROWS / 8, KPP = kBlock / VPR, PT = ROWS + kPadT;
#pragma unroll
            for (int q = 0; q < kBKH / KPP; ++q)
                *reinterpret_cast<u32x4 *>(S + (tid / VPR + q * KPP) * PT + (tid % VPR) * 8) = r[q];
        }
    };
    using RA = std::integral_constant<int, BM>; using RB = std::integral_constant<int, BN>;
    using KA = std::integral_constant<bool, AK>; using KB = std::integral_constant<bool, BK>;
    static_assert(BN <= BM, "tile registers are sized by the A tile");

    if (k_lo < k_hi) {
        load_op(op.A16, op.sam, op.sak, KA{}, m0, k_lo, ra, RA{});
        load_op(op.B16, op.sbn, op.sbk, KB{}, (int64_t)n0, k_lo, rb, RB{});
        store_op(As0, KA{}, ra, RA{});
        store_op(Bs0, KB{}, rb, RB{});
        __syncthreads();
        int cur = 0;
        for (int64_t kt = k_lo; kt < k_hi; kt += kBKH) {
            const bool more = kt + kBKH < k_hi;
            if (more) {
                load_op(op.A16, op.sam, op.sak, KA{}, m0, kt + kBKH, ra, RA{});
                load_op(op.B16, op.sbn, op.sbk, KB{}, (int64_t)n0, kt + kBKH, rb, RB{});
            }
            const uint16_t *At = As0 + cur * kTileA, *Bt = Bs0 + cur * kTileB;
            // k-contiguous tile: lane -> row lane%32, k half lane/32.  [k][row] tile: the address of this lane's piece
            // of the transposing read (k row 8*(lane/32) + (lane%16)/4, rows 16*((lane%32)/16) + 4*(lane%4) ...)
            // (the row offsets wm*64 + mi*32 and wn*32*WN + ni*32 are multiples of 32: the swizzle of a lane's row
            // depends on lane % 32 only)
            const int sw = swz_of_row(lane % 32), half = lane / 32;
            const uint16_t *as = AK ? At + (wm * 64 + lane % 32) * kLdkH
                                    : At + (8 * (lane / 32) + (lane % 16) / 4) * PTA + wm * 64 + 16 * ((lane % 32) / 16) + 4 * (lane % 4);
            const uint16_t *bs = BK ? Bt + (wn * 32 * WN + lane % 32) * kLdkH
                                    : Bt + (8 * (lane / 32) + (lane % 16) / 4) * PTB + wn * 32 * WN + 16 * ((lane % 32) / 16) + 4 * (lane % 4);
#pragma unroll
            for (int ks = 0; ks < kBKH / 16; ++ks) {
                bf16x8 a[2], b[WN];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    if constexpr (AK) a[mi] = *reinterpret_cast<const bf16x8 *>(as + mi * 32 * kLdkH + ((2 * ks + half) ^ sw) * 8);
                    else a[mi] = lds_frag_tr(as + ks * 16 * PTA + mi * 32, PTA);
                }
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    if constexpr (BK) b[ni] = *reinterpret_cast<const bf16x8 *>(bs + ni * 32 * kLdkH + ((2 * ks + half) ^ sw) * 8);
                    else b[ni] = lds_frag_tr(bs + ks * 16 * PTB + ni * 32, PTB);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
            if (more) {
                store_op(As0 + (cur ^ 1) * kTileA, KA{}, ra, RA{});
                store_op(Bs0 + (cur ^ 1) * kTileB, KB{}, rb, RB{});
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    if (EPI == EPI_ATOMIC || op.C16 == nullptr) {      // fp32 result: split-K atomics, or the tower's input gradient
        gemm_epilogue<WN, EPI, true, DROP>(op, acc, m0, n0, wm, wn, lane, tile.z);
    } else {        // bf16 output (interior tiles only: launch_gemm_h checks)
        // The MFMA result layout gives a lane ONE column and 32 scattered rows: written directly that is 64 two-byte
        // stores per lane.  So the tile takes a detour through LDS (the operand stages are dead by now) and
        // leaves as 16-byte stores along its rows.
        constexpr int LDT = BN + 8;                                       // halfwords per staged row (16-byte multiple)
        uint16_t *Ts = smem;              // (the last k iteration ended with a barrier: every fragment read is done)
        const uint16_t *__restrict__ Gt = (EPI == EPI_GATE && op.G16) ? op.G16 + m0 * op.ldg + n0 : nullptr;
        const int ldg = (int)op.ldg;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int nl = wn * 32 * WN + ni * 32 + lane % 32;
                float bias = 0.f;
                if constexpr (EPI == EPI_BIAS_RELU) bias = op.bias[n0 + nl];
#pragma unroll
                for (int i = 0; i < 16; i += 2) {                       // rows ml, ml + 1 of column nl: one packed conversion
                    const int ml = wm * 64 + mi * 32 + (i / 4) * 8 + (lane / 32) * 4 + (i % 4);
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        v[h] = acc[mi][ni][i + h];
                        if constexpr (EPI == EPI_BIAS_RELU) v[h] = fmaxf(v[h] + bias, 0.f);
                        if constexpr (DROP)
                            if (op.drop_thresh)
                                v[h] = drop_keep(op.drop_seed, op.drop_stream,
                                                 (uint64_t)(m0 + ml + h) * (uint64_t)op.N + (uint64_t)(n0 + nl), op.drop_thresh)
                                           ? v[h] * op.drop_scale : 0.f;
                    }
                    const uint32_t pk = bf16_pack2(v[0], v[1]);
                    Ts[ml * LDT + nl] = (uint16_t)pk;
                    Ts[(ml + 1) * LDT + nl] = (uint16_t)(pk >> 16);
                }
            }
        __syncthreads();
        constexpr int VPR = BN / 8;                                       // 16-byte vectors per tile row
        uint16_t *__restrict__ Ct = op.C16 + m0 * op.ldc + n0;
        for (int e = tid; e < BM * VPR; e += kBlock) {
            const int row = e / VPR, c8 = (e % VPR) * 8;
            uint4 v = *reinterpret_cast<const uint4 *>(Ts + row * LDT + c8);
            if constexpr (EPI == EPI_GATE) {
                if (Gt) {                                                 // gate: x > 0 of the layer input, 8 columns at a time
                    const uint4 gq = *reinterpret_cast<const uint4 *>(Gt + (int64_t)row * ldg + c8);
                    const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
                    uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint16_t g0 = (uint16_t)gw[q], g1 = (uint16_t)(gw[q] >> 16);
                        float lo = bf16_positive(g0) ? bf16_to_f32((uint16_t)vw[q]) * op.gate_scale : 0.f;
                        float hi = bf16_positive(g1) ? bf16_to_f32((uint16_t)(vw[q] >> 16)) * op.gate_scale : 0.f;
                        vw[q] = bf16_pack2(lo, hi);
                    }
                    v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                }
            }
            *reinterpret_cast<uint4 *>(Ct + (int64_t)row * op.ldc + c8) = v;
        }
    }
}

// shapes the bf16-storage kernel takes: whole tiles, k ranges in multiples of 32, 16-byte aligned rows
static bool gemm_h_ok(const GemmOp &op) {
    const int bn = (op.N > 64) ? 128 : 64;
    const int64_t splits = (op.k_chunk < op.K) ? (op.K + op.k_chunk - 1) / op.k_chunk : 1;
    auto al = [](const uint16_t *p, int64_t srow, int64_t sk) {
        if (((uintptr_t)p & 15) != 0) return false;
        return sk == 1 ? (srow % 8 == 0) : (srow == 1 && sk % 8 == 0);     // 16-byte loads along k / along the rows
    };
    if (op.sak != 1 && op.sbk == 1) return false;          // (A along rows, B along k) is not a layout of the tower
    return op.M % kGemmBM == 0 && op.N % bn == 0 && op.K % kBKH == 0 && (splits == 1 || op.k_chunk % kBKH == 0) &&
           al(op.A16, op.sam, op.sak) && al(op.B16, op.sbn, op.sbk);
}

template <int EPI>
static void launch_gemm_h(GemmOp op, hipStream_t s) {
    const int64_t splits = (op.k_chunk < op.K) ? (op.K + op.k_chunk - 1) / op.k_chunk : 1;
    if (op.k_chunk >= op.K) op.k_chunk = op.K;
    const int bn = (op.N > 64) ? 128 : 64;
    dim3 grid((unsigned)(op.M / kGemmBM), (unsigned)(op.N / bn), (unsigned)splits);
    constexpr bool can_drop = (EPI == EPI_BIAS_RELU || EPI == EPI_GATE);
    const bool drop = can_drop && op.drop_thresh != 0;
    const bool ak = op.sak == 1, bk = op.sbk == 1;
    auto go = [&](auto wn_c, auto drop_c, auto ak_c, auto bk_c) {
        hipLaunchKernelGGL((k_gemm_h<decltype(wn_c)::value, EPI, decltype(drop_c)::value, decltype(ak_c)::value,
                                     decltype(bk_c)::value>), grid, dim3(kBlock), 0, s, op);
    };
    using T = std::true_type; using F = std::false_type;
    using W1 = std::integral_constant<int, 1>; using W2 = std::integral_constant<int, 2>;
    auto go2 = [&](auto wn_c, auto drop_c) {          // the three operand layouts the tower uses
        if (ak && bk) go(wn_c, drop_c, T{}, T{});         // forward:          X (k) x W (k)
        else if (ak) go(wn_c, drop_c, T{}, F{});          // input gradient:   dZ (k) x W^T (rows)
        else go(wn_c, drop_c, F{}, F{});                  // weight gradient:  dZ^T (rows) x X^T (rows)
    };
    if (op.N > 64) {
        if constexpr (can_drop) { if (drop) go2(W2{}, T{}); else go2(W2{}, F{}); } else go2(W2{}, F{});
    } else {
        if constexpr (can_drop) { if (drop) go2(W1{}, T{}); else go2(W1{}, F{}); } else go2(W1{}, F{});
    }
}

// fp32 -> bf16 (round to nearest even): the per-step copy of the MLP weights, [rows][cols] as stored and (yt)
// transposed, so that the forward GEMM and the input-gradient GEMM both read W along their k
__global__ void k_to_bf16(const float *__restrict__ x, int64_t n, int cols, uint16_t *__restrict__ y,
                          uint16_t *__restrict__ yt) {
    const int64_t rows = n / cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint16_t h = (uint16_t)bf16_rne(x[e]);
        y[e] = h;
        yt[(e % cols) * rows + e / cols] = h;
    }
}
// float4 path preconditions: unit stride along one dimension, the other stride and the base 16-byte aligned
static bool vec_ok(const float *p, int64_t s_row, int64_t s_k, int64_t k_chunk) {
    if (((uintptr_t)p & 15) != 0) return false;
    if (s_k == 1) return s_row % 4 == 0 && k_chunk % 4 == 0;
    if (s_row == 1) return s_k % 4 == 0;
    return false;
}

template <int EPI>
static void launch_gemm(GemmOp op, hipStream_t s) {
    const int64_t splits = (op.k_chunk < op.K) ? (op.K + op.k_chunk - 1) / op.k_chunk : 1;
    if (op.k_chunk >= op.K) op.k_chunk = op.K;
    op.vec_a = vec_ok(op.A, op.sam, op.sak, splits > 1 ? op.k_chunk : 4);
    op.vec_b = vec_ok(op.B, op.sbn, op.sbk, splits > 1 ? op.k_chunk : 4);
    const int bn = (op.N > 64) ? 128 : 64;
    const bool fast = op.vec_a && op.vec_b && op.M % kGemmBM == 0 && op.N % bn == 0 && op.K % kBK == 0 &&
                      (splits == 1 || op.k_chunk % kBK == 0);
    dim3 grid((unsigned)((op.M + kGemmBM - 1) / kGemmBM), (unsigned)((op.N + bn - 1) / bn), (unsigned)splits);
    const bool bf16 = op.bf16 && fast && op.K % kBK16 == 0 && (splits == 1 || op.k_chunk % kBK16 == 0);
    auto go = [&](auto wn_c, auto fast_c, auto drop_c) {
        if (bf16)
            hipLaunchKernelGGL((k_gemm_bf16<decltype(wn_c)::value, EPI, decltype(drop_c)::value>), grid, dim3(kBlock), 0,
                               s, op);
        else
            hipLaunchKernelGGL((k_gemm<decltype(wn_c)::value, EPI, decltype(fast_c)::value, decltype(drop_c)::value>),
                               grid, dim3(kBlock), 0, s, op);
    };
    using T = std::true_type; using F = std::false_type;
    using W1 = std::integral_constant<int, 1>; using W2 = std::integral_constant<int, 2>;
    constexpr bool can_drop = (EPI == EPI_BIAS_RELU || EPI == EPI_GATE);
    const bool drop = can_drop && op.drop_thresh != 0;
    if (op.N > 64) {
        if (fast) { if constexpr (can_drop) { if (drop) go(W2{}, T{}, T{}); else go(W2{}, T{}, F{}); } else go(W2{}, T{}, F{}); }
        else      { if constexpr (can_drop) { if (drop) go(W2{}, F{}, T{}); else go(W2{}, F{}, F{}); } else go(W2{}, F{}, F{}); }
    } else {
        if (fast) { if constexpr (can_drop) { if (drop) go(W1{}, T{}, T{}); else go(W1{}, T{}, F{}); } else go(W1{}, T{}, F{}); }
        else      { if constexpr (can_drop) { if (drop) go(W1{}, F{}, T{}); else go(W1{}, F{}, F{}); } else go(W1{}, F{}, F{}); }
    }
}


// two products of the same (N, tile width) in one launch (k_gemm_pair): guarded-loader kernels, fp32
template <int EPI>
static void launch_gemm_pair(GemmOp a, GemmOp b, hipStream_t s) {
    auto prep = [](GemmOp &op) -> unsigned {
        const int64_t splits = (op.k_chunk < op.K) ? (op.K + op.k_chunk - 1) / op.k_chunk : 1;
        if (op.k_chunk >= op.K) op.k_chunk = op.K;
        op.vec_a = vec_ok(op.A, op.sam, op.sak, splits > 1 ? op.k_chunk : 4);
        op.vec_b = vec_ok(op.B, op.sbn, op.sbk, splits > 1 ? op.k_chunk : 4);
        return (unsigned)splits;
    };
    const unsigned az = prep(a), bz = prep(b);
    const int bn = (a.N > 64) ? 128 : 64;
    const unsigned ax = (unsigned)((a.M + kGemmBM - 1) / kGemmBM), bx = (unsigned)((b.M + kGemmBM - 1) / kGemmBM);
    const dim3 grid(ax + bx, (unsigned)((a.N + bn - 1) / bn), az > bz ? az : bz);
    if (a.N > 64) hipLaunchKernelGGL((k_gemm_pair<2, EPI>), grid, dim3(kBlock), 0, s, a, b, ax, az, bz);
    else hipLaunchKernelGGL((k_gemm_pair<1, EPI>), grid, dim3(kBlock), 0, s, a, b, ax, az, bz);
}

// ---------------------------------------------------------------------------------------------
// the three pair layouts of daisy_neumf_scores plus the training batch
// ---------------------------------------------------------------------------------------------

struct L16 { static constexpr int LPR = 16; };

// x0[r] = [uM[user] | iM[item]] (* dropout), g[r] = uG[user]*iG[item]; TRAIN: the ten regulariser sums.
// H: the activations live as bf16 in HBM (precision level 2)
// FACT (round 5, "the first layer through the tables"): the MLP's first layer is linear in the concatenated embedding
// rows, z1[r] = W1[:, :dm] uM[user] + W1[:, dm:] iM[item] + b1, and a step of R rows meets only U + I DISTINCT table rows
// (ml-1m: 9746 against 524 288).  So T_u = uM W1[:, :dm]^T and T_i = iM W1[:, dm:]^T are two small GEMMs over the tables
// (Fact::tu, Fact::ti: fp32 [rows][n1]) and x1[r] = relu(T_u[user] + T_i[item] + b1) is a gather: x0 - 1 KB per row in bf16 -
// is never formed, the largest GEMM of the tower and its 2 x 512-column operand stream are gone.  The backward pass
// mirrors it (neumf_scatter_owner).  Needs dropout = 0 (a mask on x0's elements would not factor).
struct Fact {
    const uint16_t *tu, *ti; const float *b1; uint16_t *x1; int n1; const float2 *nu, *ni;     // nu / ni: k_nmf_row_norms
    const float *tu32, *ti32; float *x1_32;      // round 6, the fp32 (parity) mode: the products and x1 stay fp32
};
// (the products are handed to the gather as bf16: it is bound by reading them - 2 x 1 KB per row in fp32 from beyond L2 -
// and the plain path rounds x0 and W1 to bf16 BEFORE the product)
__global__ void k_f32_to_bf16(const float *__restrict__ x, int64_t n, uint16_t *__restrict__ y) {
    for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 2; e < n; e += (int64_t)gridDim.x * blockDim.x * 2)
        *reinterpret_cast<uint32_t *>(y + e) = bf16_pack2(x[e], x[e + 1]);
}

template <bool TRAIN, bool H = false, bool FACT = false>
__global__ __launch_bounds__(kBlock) void k_nmf_gather(daisy_neumf_params p, PairSrc src, int64_t R, int d,
                                                       int dm, int pointwise, float *__restrict__ X0,
                                                       float *__restrict__ G, uint32_t thresh,
                                                       float scale, uint64_t seed,
                                                       double *__restrict__ stats, Fact fact = Fact{}) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    float s1[5] = {0, 0, 0, 0, 0}, s2[5] = {0, 0, 0, 0, 0};
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
        int64_t user, item;
        pair_ids(src, r, user, item);
        const bool first = !TRAIN || r < src.B;            // rows r >= B repeat the users with the negatives
        // a lane takes 4 consecutive columns: 16-byte table reads, 16-byte (fp32) or 8-byte (bf16) stores - the
        // first version moved one element per lane and stored bf16 two bytes at a time (363 us per 524 288 rows)
        const float *um = p.uM + user * dm, *im = p.iM + item * dm;
        float *x = X0 + r * (int64_t)(2 * dm);
        uint16_t *xh = reinterpret_cast<uint16_t *>(X0) + r * (int64_t)(2 * dm);
        if constexpr (FACT && !H) {
            // fp32 mode: x1 = relu((T_u[user] + T_i[item]) + b1) in fp32 - the reference's first layer up to the association of
            // its 2 dm-term dot product (two dm-term products, then two adds)
            const float *tu = fact.tu32 + user * fact.n1, *ti = fact.ti32 + item * fact.n1;
            float *x1 = fact.x1_32 + r * (int64_t)fact.n1;
            for (int c = 4 * lane; c < fact.n1; c += 64) {
                const float4 a4 = *reinterpret_cast<const float4 *>(tu + c), b4 = *reinterpret_cast<const float4 *>(ti + c);
                const float4 c4 = *reinterpret_cast<const float4 *>(fact.b1 + c);
                *reinterpret_cast<float4 *>(x1 + c) = make_float4(fmaxf((a4.x + b4.x) + c4.x, 0.f), fmaxf((a4.y + b4.y) + c4.y, 0.f),
                                                                  fmaxf((a4.z + b4.z) + c4.z, 0.f), fmaxf((a4.w + b4.w) + c4.w, 0.f));
            }
        }
        if constexpr (FACT && H) {
            // x1 from the two table products; the embedding rows themselves are read only where the regulariser counts
            // them (the positive half of the rows: NeuMFRecommender.py:149-167)
            const uint16_t *tu = fact.tu + user * fact.n1, *ti = fact.ti + item * fact.n1;
            uint16_t *x1 = fact.x1 + r * (int64_t)fact.n1;
            for (int c = 4 * lane; c < fact.n1; c += 64) {
                const uint2 ah = *reinterpret_cast<const uint2 *>(tu + c), bh = *reinterpret_cast<const uint2 *>(ti + c);
                const float4 a4 = make_float4(__uint_as_float(ah.x << 16), __uint_as_float(ah.x & 0xFFFF0000u),
                                              __uint_as_float(ah.y << 16), __uint_as_float(ah.y & 0xFFFF0000u));
                const float4 b4 = make_float4(__uint_as_float(bh.x << 16), __uint_as_float(bh.x & 0xFFFF0000u),
                                              __uint_as_float(bh.y << 16), __uint_as_float(bh.y & 0xFFFF0000u));
                const float4 c4 = *reinterpret_cast<const float4 *>(fact.b1 + c);
                const float z0 = fmaxf((a4.x + b4.x) + c4.x, 0.f), z1 = fmaxf((a4.y + b4.y) + c4.y, 0.f);
                const float z2 = fmaxf((a4.z + b4.z) + c4.z, 0.f), z3 = fmaxf((a4.w + b4.w) + c4.w, 0.f);
                *reinterpret_cast<uint2 *>(x1 + c) = make_uint2(bf16_pack2(z0, z1), bf16_pack2(z2, z3));
            }
        }
        if constexpr (FACT && TRAIN) {
            if (first && lane == 0) {              // the MLP rows' share of the regulariser sums, from the per-row table
                const float2 a = fact.nu[user], b = fact.ni[item];
                s1[1] += a.x; s2[1] += a.y; s1[3] += b.x; s2[3] += b.y;
            }
        }
        for (int c = 4 * lane; c < dm && !FACT; c += 64) {
            const float4 a4 = *reinterpret_cast<const float4 *>(um + c), b4 = *reinterpret_cast<const float4 *>(im + c);
            float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
            if (TRAIN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (first) {
                        s1[1] += fabsf(a[k]); s2[1] = fmaf(a[k], a[k], s2[1]);
                        s1[3] += fabsf(b[k]); s2[3] = fmaf(b[k], b[k], s2[3]);
                    }
                    if (thresh) {
                        a[k] = drop_keep(seed, 1, (uint64_t)r * (2 * dm) + c + k, thresh) ? a[k] * scale : 0.f;
                        b[k] = drop_keep(seed, 1, (uint64_t)r * (2 * dm) + dm + c + k, thresh) ? b[k] * scale : 0.f;
                    }
                }
            }
            if constexpr (FACT) {
                // (no x0)
            } else if constexpr (H) {
                *reinterpret_cast<uint2 *>(xh + c) =
                    make_uint2(bf16_pack2(a[0], a[1]), bf16_pack2(a[2], a[3]));
                *reinterpret_cast<uint2 *>(xh + dm + c) =
                    make_uint2(bf16_pack2(b[0], b[1]), bf16_pack2(b[2], b[3]));
            } else {
                *reinterpret_cast<float4 *>(x + c) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4 *>(x + dm + c) = make_float4(b[0], b[1], b[2], b[3]);
            }
        }
        const float *ug = p.uG + user * d, *ig = p.iG + item * d;
        for (int c = 4 * lane; c < d; c += 64) {
            const float4 a4 = *reinterpret_cast<const float4 *>(ug + c), b4 = *reinterpret_cast<const float4 *>(ig + c);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
            *reinterpret_cast<float4 *>(G + r * (int64_t)d + c) = make_float4(a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]);
            if (TRAIN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (first) { s1[0] += fabsf(a[k]); s2[0] = fmaf(a[k], a[k], s2[0]); s1[2] += fabsf(b[k]); s2[2] = fmaf(b[k], b[k], s2[2]); }
                    else if (!pointwise) { s1[4] += fabsf(b[k]); s2[4] = fmaf(b[k], b[k], s2[4]); }
                }
            }
        }
    }
    if constexpr (TRAIN) {
        __shared__ double sm[kBlock / kWave][10];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double a = wave_sum_f64((double)s1[k]), b = wave_sum_f64((double)s2[k]);
            if (threadIdx.x % kWave == 0) { sm[threadIdx.x / kWave][k] = a; sm[threadIdx.x / kWave][5 + k] = b; }
        }
        __syncthreads();
        if (threadIdx.x < 10) {
            double t = 0.0;
            for (int w = 0; w < kBlock / kWave; ++w) t += sm[w][threadIdx.x];
            atomicAdd(stats + DAISY_NST_L1 + threadIdx.x, t);        // L1[5] then SQ[5] are adjacent
        }
    }
}

// pred[r] = <Wp[:dg], g[r]> + <Wp[dg:], x_L[r]> + bp      (dg = 0 for model MLP, nl = 0 for model GMF)
template <bool H = false>
__global__ __launch_bounds__(kBlock) void k_nmf_predict(const float *__restrict__ G, int dg,
                                                        const float *__restrict__ XL, int nl,
                                                        const float *__restrict__ Wp,
                                                        const float *__restrict__ bp, int64_t R,
                                                        float *__restrict__ pred) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
        float s = 0.f;
        for (int c = lane; c < dg; c += 16) s = fmaf(Wp[c], G[r * (int64_t)dg + c], s);
        for (int c = lane; c < nl; c += 16) {
            float x;
            if constexpr (H) x = bf16_to_f32(reinterpret_cast<const uint16_t *>(XL)[r * (int64_t)nl + c]);
            else x = XL[r * (int64_t)nl + c];
            s = fmaf(Wp[dg + c], x, s);
        }
        s = group_sum<L16>(s);
        if (lane == 0) pred[r] = s + bp[0];
    }
}

// criterion epilogue (AbstractRecommender.py:79-93, daisy/utils/loss.py): d loss / d pred for both rows of a sample
__global__ __launch_bounds__(kBlock) void k_nmf_loss(const float *__restrict__ pred,
                                                     const int32_t *__restrict__ j, int64_t B,
                                                     int loss_type, float gamma, int pointwise,
                                                     float *__restrict__ dpred,
                                                     double *__restrict__ stats,
                                                     float *__restrict__ gbp_ws) {
    // gbp = sum_b (cp_b + cn_b), paired per sample: under BPR / HL every pair is exactly 0, as it is in
    // the reference's autograd (a rounding residue here would be blown up to +-lr by Adam)
    double acc = 0.0;
    float accb = 0.f;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float term, cp, cn;
        pair_coef(loss_type, pred[b], pointwise ? (float)j[b] : pred[B + b], gamma, term, cp, cn);
        dpred[b] = cp;
        if (!pointwise) dpred[B + b] = cn;
        acc += (double)term;
        accb += cp + cn;
    }
    acc = wave_sum_f64(acc);
    const double sb = wave_sum_f64((double)accb);
    __shared__ double smb[kBlock / kWave];
    if (threadIdx.x % kWave == 0) {
        atomicAdd(stats + DAISY_NST_LOSS_DATA, acc);
        smb[threadIdx.x / kWave] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                            // waves in order; the workgroups' sums are added by k_reduce_slices
        double t = 0.0;
        for (int w = 0; w < kBlock / kWave; ++w) t += smb[w];
        gbp_ws[blockIdx.x] = (float)t;
    }
}

// NeuMFRecommender.py:149-167 summed up: the negative item's GMF rows enter twice, its MLP rows never
__global__ void k_nmf_finalize(double *__restrict__ stats, float reg_1, float reg_2, int pointwise) {
    if (threadIdx.x || blockIdx.x) return;
    double l1 = 0.0, fro = 0.0;
    for (int k = 0; k < 5; ++k) {
        const double n = sqrt(stats[DAISY_NST_SQ + k]);
        stats[DAISY_NST_NORM + k] = n;
        const double w = (k == 4) ? (pointwise ? 0.0 : 2.0) : 1.0;
        l1 += w * stats[DAISY_NST_L1 + k];
        fro += w * n;
    }
    const double loss = stats[DAISY_NST_LOSS_DATA] + (double)reg_1 * l1 + (double)reg_2 * fro;
    stats[DAISY_NST_LOSS] = loss;
    stats[DAISY_NST_LOSS_SUM] += loss;
}

// out[c] += sum_s ws[s][c], s = 0 .. nslices-1 in that order: the second half of every reduction over the batch rows
// whose first half is spread over workgroups (split-K slices of the weight-gradient GEMMs, row tiles of the column
// sums, workgroups of the predict layer's backward pass).  Fixed order instead of fp32 atomics: two runs of a step
// give the same bits.
__global__ __launch_bounds__(kBlock) void k_reduce_slices(const float *__restrict__ ws, int nslices, int64_t len,
                                                          float *__restrict__ out) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < len; c += (int64_t)gridDim.x * blockDim.x) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int sidx = 0;
        for (; sidx + 3 < nslices; sidx += 4) {                          // four loads in flight, fixed association
            t0 += ws[(int64_t)sidx * len + c];
            t1 += ws[(int64_t)(sidx + 1) * len + c];
            t2 += ws[(int64_t)(sidx + 2) * len + c];
            t3 += ws[(int64_t)(sidx + 3) * len + c];
        }
        for (; sidx < nslices; ++sidx) t0 += ws[(int64_t)sidx * len + c];
        out[c] += (t0 + t1) + (t2 + t3);
    }
}

// the same into a [rows][cols] block of a wider matrix (leading dimension ldo): slices are contiguous [rows][cols]
__global__ __launch_bounds__(kBlock) void k_reduce_slices_2d(const float *__restrict__ ws, int nslices, int rows, int cols,
                                                             float *__restrict__ out, int64_t ldo,
                                                             const float *__restrict__ ws_b = nullptr, int nslices_b = 0,
                                                             float *__restrict__ out_b = nullptr) {
    if (blockIdx.y) { ws = ws_b; nslices = nslices_b; out = out_b; }      // (a second block of the same shape in the same launch)
    const int64_t len = (int64_t)rows * cols;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < len; c += (int64_t)gridDim.x * blockDim.x) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int sidx = 0;
        for (; sidx + 3 < nslices; sidx += 4) {
            t0 += ws[(int64_t)sidx * len + c];
            t1 += ws[(int64_t)(sidx + 1) * len + c];
            t2 += ws[(int64_t)(sidx + 2) * len + c];
            t3 += ws[(int64_t)(sidx + 3) * len + c];
        }
        for (; sidx < nslices; ++sidx) t0 += ws[(int64_t)sidx * len + c];
        out[(c / cols) * ldo + (c % cols)] += (t0 + t1) + (t2 + t3);
    }
}

// per table row: (sum |x|, sum x^2) - what the regulariser sums of a step need from an MLP embedding row
// (k_nmf_gather<FACT>, which does not read the rows themselves)
__global__ __launch_bounds__(kBlock) void k_nmf_row_norms(const float *__restrict__ Ta, int64_t rows_a, const float *__restrict__ Tb,
                                                          int64_t rows_b, int width, float2 *__restrict__ out_a,
                                                          float2 *__restrict__ out_b) {
    // (both tables in one launch: blockIdx.y)
    const float *__restrict__ T = blockIdx.y ? Tb : Ta;
    const int64_t rows = blockIdx.y ? rows_b : rows_a;
    float2 *__restrict__ out = blockIdx.y ? out_b : out_a;
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < rows; r += gstride) {
        float a = 0.f, b = 0.f;
        for (int c = 4 * lane; c < width; c += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(T + r * (int64_t)width + c);
            a += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
            b = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, b))));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); b += __shfl_xor(b, o, 16); }
        if (lane == 0) out[r] = make_float2(a, b);
    }
}

// dZ_L[r] = dpred[r] * Wp[dg:] gated by x_L[r] > 0;  gWp += sum_r dpred[r]*[g[r] | x_L[r]];  gbp += sum dpred
// (the column sums leave as one row of `ws` per workgroup: ws[blockIdx.x][dg + nl], summed by k_reduce_slices)
template <bool H = false>
__global__ __launch_bounds__(kBlock) void k_nmf_pred_bwd(const float *__restrict__ dpred,
                                                         const float *__restrict__ G, int dg,
                                                         const float *__restrict__ XL, int nl,
                                                         const float *__restrict__ Wp, int64_t R,
                                                         float *__restrict__ DZ, float *__restrict__ ws) {
    __shared__ float colg[kBlock / 16][128];         // one sweep of 128 columns: every lane group's partial sums
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    for (int c0 = 0; c0 < dg + nl; c0 += 16 * 8) {   // 8 column registers per lane per sweep
        float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
            const float dp = dpred[r];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = c0 + q * 16 + lane;
                if (c < dg) part[q] = fmaf(dp, G[r * (int64_t)dg + c], part[q]);
                else if (c < dg + nl) {
                    float x;
                    if constexpr (H) x = bf16_to_f32(reinterpret_cast<const uint16_t *>(XL)[r * (int64_t)nl + (c - dg)]);
                    else x = XL[r * (int64_t)nl + (c - dg)];
                    part[q] = fmaf(dp, x, part[q]);
                    const float dzv = (x > 0.f) ? dp * Wp[c] : 0.f;
                    if constexpr (H) reinterpret_cast<uint16_t *>(DZ)[r * (int64_t)nl + (c - dg)] = (uint16_t)bf16_rne(dzv);
                    else DZ[r * (int64_t)nl + (c - dg)] = dzv;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) colg[group][q * 16 + lane] = part[q];
        __syncthreads();
        if ((int)threadIdx.x < 128 && c0 + (int)threadIdx.x < dg + nl) {           // groups in order
            float t = 0.f;
#pragma unroll
            for (int gq = 0; gq < kBlock / 16; ++gq) t += colg[gq][threadIdx.x];
            ws[(int64_t)blockIdx.x * (dg + nl) + c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// the same for dg == nl == d <= 64, d % 4 == 0 (every NeuMF tower: the last layer is `factors` wide like the GMF
// product): 16 lanes per row, each lane 4 consecutive columns of g[r] and of x_L[r] with one vector load each
template <bool H>
__global__ __launch_bounds__(kBlock) void k_nmf_pred_bwd_v(const float *__restrict__ dpred,
                                                           const float *__restrict__ G, int d,
                                                           const float *__restrict__ XL,
                                                           const float *__restrict__ Wp, int64_t R,
                                                           float *__restrict__ DZ, float *__restrict__ ws) {
    __shared__ float colg[kBlock / 16][128];          // [lane group][g columns 0..63 | x columns 64..127]
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const bool on = 4 * lane < d;
    const int c4 = on ? 4 * lane : 0;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    const float4 wp = *reinterpret_cast<const float4 *>(Wp + d + c4);
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
        const float dp = dpred[r];
        const float4 g4 = *reinterpret_cast<const float4 *>(G + r * d + c4);
        float x[4];
        if constexpr (H) {
            const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(XL) + r * d + c4);
            x[0] = __uint_as_float(q.x << 16); x[1] = __uint_as_float(q.x & 0xFFFF0000u);
            x[2] = __uint_as_float(q.y << 16); x[3] = __uint_as_float(q.y & 0xFFFF0000u);
        } else {
            const float4 q = *reinterpret_cast<const float4 *>(XL + r * d + c4);
            x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
        }
        pg[0] = fmaf(dp, g4.x, pg[0]); pg[1] = fmaf(dp, g4.y, pg[1]);
        pg[2] = fmaf(dp, g4.z, pg[2]); pg[3] = fmaf(dp, g4.w, pg[3]);
        const float w4[4] = {wp.x, wp.y, wp.z, wp.w};
        float dz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            px[k] = fmaf(dp, x[k], px[k]);
            dz[k] = (x[k] > 0.f) ? dp * w4[k] : 0.f;
        }
        if (on) {
            if constexpr (H)
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(DZ) + r * d + c4) =
                    make_uint2(bf16_pack2(dz[0], dz[1]), bf16_pack2(dz[2], dz[3]));
            else
                *reinterpret_cast<float4 *>(DZ + r * d + c4) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { colg[group][4 * lane + k] = on ? pg[k] : 0.f; colg[group][64 + 4 * lane + k] = on ? px[k] : 0.f; }
    __syncthreads();
    const int t = (int)threadIdx.x;
    if (t < 128) {                                    // lane groups in order; ws row = [g columns (d) | x columns (d)]
        float acc = 0.f;
#pragma unroll
        for (int gq = 0; gq < kBlock / 16; ++gq) acc += colg[gq][t];
        if (t < d) ws[(int64_t)blockIdx.x * (2 * d) + t] = acc;
        else if (t >= 64 && t < 64 + d) ws[(int64_t)blockIdx.x * (2 * d) + d + (t - 64)] = acc;
    }
}

// ws[row tile][n] = sum over the tile's rows of X[r*ld + n]: a block takes 64 columns x kColsumRows rows (grid = column
// tiles x row tiles); k_reduce_slices adds the row tiles in order
constexpr int kColsumRows = 512;
// rows per workgroup: 512, or 64 for matrices of a few thousand rows (a step of 512 rows was ONE workgroup per 64 columns
// walking all of them: 15 us for 100 KB)
static inline int colsum_rows(int64_t R) { return R <= 8192 ? 64 : kColsumRows; }
template <bool H = false>
__global__ __launch_bounds__(kBlock) void k_colsum(const float *__restrict__ X, int64_t R, int N, int64_t ld,
                                                   float *__restrict__ out, int rows_per_block = kColsumRows) {
    auto at = [&](int64_t idx) -> float {
        if constexpr (H) return bf16_to_f32(reinterpret_cast<const uint16_t *>(X)[idx]);
        else return X[idx];
    };
    __shared__ float sm[4][64];
    const int c = threadIdx.x % 64, rr = threadIdx.x / 64;
    const int n = blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int64_t r = r0 + rr;
        for (; r + 4 < r1; r += 8) { s0 += at(r * ld + n); s1 += at((r + 4) * ld + n); }
        if (r < r1) s0 += at(r * ld + n);
    }
    sm[rr][c] = s0 + s1;
    __syncthreads();
    if (rr == 0 && n < N) out[(int64_t)blockIdx.y * N + n] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
}

// the same for a bf16 matrix whose rows are whole 16-byte vectors (N % 8 == 0, N <= 2048, ld == N): a thread owns
// 8 consecutive columns and reads them with one 16-byte load per row; a block takes kColsumRowsH rows
constexpr int kColsumRowsH = 512;
__global__ __launch_bounds__(kBlock) void k_colsum_h(const uint16_t *__restrict__ X, int64_t R, int N,
                                                     float *__restrict__ out) {
    __shared__ float sm[kBlock][9];
    const int vpr = N / 8;                            // vectors per row (a divisor of kBlock or a multiple: see the launch)
    const int cv = threadIdx.x % vpr, rr = threadIdx.x / vpr, rpp = kBlock / vpr;
    const int64_t r0 = (int64_t)blockIdx.x * kColsumRowsH;
    const int64_t r1 = (r0 + kColsumRowsH < R) ? r0 + kColsumRowsH : R;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add = [&](const uint4 &q) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __uint_as_float(w[k] << 16);
            acc[2 * k + 1] += __uint_as_float(w[k] & 0xFFFF0000u);
        }
    };
    const uint16_t *col = X + cv * 8;
    int64_t r = r0 + rr;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {          // four rows in flight per thread
        const uint4 q0 = *reinterpret_cast<const uint4 *>(col + r * N);
        const uint4 q1 = *reinterpret_cast<const uint4 *>(col + (r + rpp) * N);
        const uint4 q2 = *reinterpret_cast<const uint4 *>(col + (r + 2 * rpp) * N);
        const uint4 q3 = *reinterpret_cast<const uint4 *>(col + (r + 3 * rpp) * N);
        add(q0); add(q1); add(q2); add(q3);
    }
    for (; r < r1; r += rpp) add(*reinterpret_cast<const uint4 *>(col + r * N));
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += kBlock) {   // column c: vector c/8, element c%8, summed over the row groups
        float t = 0.f;
        for (int g = 0; g < rpp; ++g) t += sm[g * vpr + c / 8][c % 8];
        out[(int64_t)blockIdx.x * N + c] = t;          // (row tile blockIdx.x of the workspace: see k_colsum)
    }
}

// embedding gradients of one row r (dense tables, fp32 atomics) + the regulariser gradients
template <bool H = false>
__global__ __launch_bounds__(kBlock) void k_nmf_scatter(daisy_neumf_params p, daisy_neumf_params g, PairSrc src,
                                                        int64_t R, int d, int dm, int model, int pointwise,
                                                        const float *__restrict__ dpred,
                                                        const float *__restrict__ DX0,
                                                        const double *__restrict__ stats, float reg_1,
                                                        float reg_2) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    float inv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double n = stats[DAISY_NST_NORM + k];
        inv[k] = (n > 0.0) ? (float)((double)reg_2 / n) : 0.f;
    }
    const bool reg = (reg_1 != 0.f) || (reg_2 != 0.f);
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + group; r < R; r += gstride) {
        int64_t user, item;
        pair_ids(src, r, user, item);
        const bool first = r < src.B;
        const float dp = dpred[r];
        // MLP tables
        for (int c = lane; c < dm; c += 16) {
            float gu = 0.f, gi = 0.f;
            if (model != DAISY_NEUMF_GMF) {
                if constexpr (H) {
                    const uint16_t *dx = reinterpret_cast<const uint16_t *>(DX0) + r * (int64_t)(2 * dm);
                    gu = bf16_to_f32(dx[c]);
                    gi = bf16_to_f32(dx[dm + c]);
                } else {
                    gu = DX0[r * (int64_t)(2 * dm) + c];
                    gi = DX0[r * (int64_t)(2 * dm) + dm + c];
                }
            }
            if (reg && first) {
                const float a = p.uM[user * dm + c], b = p.iM[item * dm + c];
                gu += fmaf(inv[1], a, reg_1 * sgn(a));
                gi += fmaf(inv[3], b, reg_1 * sgn(b));
            }
            if (gu != 0.f) unsafeAtomicAdd(g.uM + user * dm + c, gu);
            if (gi != 0.f) unsafeAtomicAdd(g.iM + item * dm + c, gi);
        }
        // GMF tables
        for (int c = lane; c < d; c += 16) {
            const float a = p.uG[user * d + c], b = p.iG[item * d + c];
            float gu = 0.f, gi = 0.f;
            if (model != DAISY_NEUMF_MLP) {
                const float w = dp * p.Wp[c];
                gu = w * b;
                gi = w * a;
            }
            if (reg) {
                if (first) {
                    gu += fmaf(inv[0], a, reg_1 * sgn(a));
                    gi += fmaf(inv[2], b, reg_1 * sgn(b));
                } else if (!pointwise) {
                    gi += 2.f * fmaf(inv[4], b, reg_1 * sgn(b));
                }
            }
            if (gu != 0.f) unsafeAtomicAdd(g.uG + user * d + c, gu);
            if (gi != 0.f) unsafeAtomicAdd(g.iG + item * d + c, gi);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Embedding gradients without atomics (default): the R rows of a step are sorted by user and by item (two
// radix sorts of R int32 keys), and each table's gradient is a segmented reduction over the sorted list on the
// MF item pass's kernel (segsum_rows: single owner per table row, fixed summation order, bitwise reproducible).
// With ml-1m's 6040 users a batch of 524 288 rows hits every user row ~87 times: the atomic kernel serialises
// on those addresses.  The regulariser terms are count * f(row) per table row (integer counts).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Round 6: the embedding gradients of a SMALL step (at most 1024 rows: the reference's own batch of 256 samples is 512) in
// one launch - sixteen workgroups per side - instead of the ~22 launches of the owner-based scatter below (keys, two sorts, entry
// lists, four segmented reductions with their edge launches, four commits): at that size every one of them is a few
// microseconds of launch latency around almost no work, and together they were a third of the 300 us step.
// Per side: the rows' (table row, row) pairs are sorted in LDS (bitonic, one element per thread); a lane group owns each
// table row that occurs and adds its rows' contributions in ascending row order (deterministic), then the regulariser terms
// of NeuMFRecommender.py:149-167 from the run's own counts, and writes the four gradient rows.
// ---------------------------------------------------------------------------------------------
constexpr int kScatterSmallRows = 1024;       // threads of the small-step scatter kernels (and the most rows the sorting one takes)
constexpr int kScanMaxRows = 8192;            // most rows of a step the scanning kernel takes (its keys, masks and round numbers: 116 KB of LDS)
// its workgroup: 16 step rows (one 16-lane group each, four waves) up to 2048 rows, 32 beyond - every workgroup
// holds ALL keys of the step in LDS and compares its rows with them, a wave scanning for its four rows at once: the scan's
// length does not depend on the workgroup's size, so the smallest one that still gives one workgroup per CU spreads it best
static int scan_block(int64_t R) { return R <= 2048 ? 256 : 512; }      // (64 groups x 128 rounds of masks would not fit beside 8192 keys)
__global__ __launch_bounds__(kScatterSmallRows) void k_nmf_scatter_small(daisy_neumf_params p, daisy_neumf_params g, PairSrc src,
                                                                        int R, int d, int dm, int model, int pointwise,
                                                                        const float *__restrict__ dpred,
                                                                        const float *__restrict__ DX0,
                                                                        const double *__restrict__ stats, float reg_1,
                                                                        float reg_2) {
    __shared__ uint32_t comp[kScatterSmallRows];          // table row << 10 | step row; padding sorts last
    // (16 workgroups per side: every one sorts the whole list - microseconds - and owns the runs whose heads fall on its
    // share of the positions; two workgroups walked ~250 runs each through dependent loads: 90 us)
    const int side = blockIdx.x, tid = threadIdx.x;
    {
        uint32_t c = 0xFFFFFFFFu;
        if (tid < R) {
            int64_t user, item;
            pair_ids(src, tid, user, item);
            c = ((uint32_t)(side ? item : user) << 10) | (uint32_t)tid;
        }
        comp[tid] = c;
    }
    __syncthreads();
    for (int k = 2; k <= kScatterSmallRows; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int other = tid ^ j;
            if (other > tid) {
                const uint32_t a = comp[tid], b = comp[other];
                const bool up = (tid & k) == 0;
                if ((a > b) == up) { comp[tid] = b; comp[other] = a; }
            }
            __syncthreads();
        }
    auto inv = [&](int k) { const double n = stats[DAISY_NST_NORM + k]; return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f; };
    const float i_m = inv(side ? 3 : 1), i_g = inv(side ? 2 : 0), i_neg = 2.f * inv(4);
    const int lane = tid % 16, group = tid / 16;
    const float *tabM = side ? p.iM : p.uM, *tabG = side ? p.iG : p.uG, *otherG = side ? p.uG : p.iG;
    float *gM = side ? g.iM : g.uM, *gG = side ? g.iG : g.uG;
    for (int e = (int)blockIdx.y * (kScatterSmallRows / 16) + group; e < R; e += (int)gridDim.y * (kScatterSmallRows / 16)) {
        const uint32_t row = comp[e] >> 10;
        if (e > 0 && (comp[e - 1] >> 10) == row) continue;            // the head of a run owns the table row
        int run = 1;
        while (e + run < R && (comp[e + run] >> 10) == row) ++run;
        float npos = 0.f, nneg = 0.f;
        for (int q = 0; q < run; ++q) { if ((int64_t)(comp[e + q] & 1023u) < src.B) npos += 1.f; else nneg += 1.f; }
        // MLP table: the rows' input gradients (this side's half of dX0), the regulariser on the positive rows' occurrences
        for (int c = lane; c < dm; c += 16) {
            float v = 0.f;
            if (model != DAISY_NEUMF_GMF)
                for (int q = 0; q < run; ++q) v += DX0[(int64_t)(comp[e + q] & 1023u) * (2 * dm) + side * dm + c];
            if (npos > 0.f) { const float w = tabM[(int64_t)row * dm + c]; v += fmaf(npos * i_m, w, reg_1 * npos * sgn(w)); }
            if (v != 0.f) gM[(int64_t)row * dm + c] += v;
        }
        // GMF table: Wp[c] x sum of dpred[r] x the OTHER table's row; the negative item's rows count twice in the regulariser
        for (int c = lane; c < d; c += 16) {
            float v = 0.f;
            if (model != DAISY_NEUMF_MLP) {
                for (int q = 0; q < run; ++q) {
                    const int64_t r = comp[e + q] & 1023u;
                    int64_t user, item;
                    pair_ids(src, r, user, item);
                    v = fmaf(dpred[r], otherG[(side ? user : item) * d + c], v);
                }
                v *= p.Wp[c];
            }
            const float na = npos, nb = (side && !pointwise) ? nneg : 0.f;
            if (na + nb > 0.f) {
                const float w = tabG[(int64_t)row * d + c];
                v += fmaf(na * i_g + nb * i_neg, w, reg_1 * (na + 2.f * nb) * sgn(w));
            }
            if (v != 0.f) gG[(int64_t)row * d + c] += v;
        }
    }
}

// The same without the sort, and for steps of up to kScanMaxRows rows: a workgroup per 16 (32) step rows, a 16-lane group per
// row.  Every workgroup holds the step's keys in LDS; a wave compares them, 64 per round, with the keys of its four rows - a
// ballot is the round's mask of rows with the same user (item) - and keeps the rounds with a match; a row whose key occurred
// earlier in the step leaves (the first occurrence owns the table row), an owner walks its masks - the matching rows in
// ascending order, the order of the sorted list - with all of a matched row's columns in flight at once.  Same sums in the
// same order as k_nmf_scatter_small: bit-identical gradients (tests/test_gpu_neumf.py).  The scan is O(rows^2 / 64) per side:
// 3 us at 512 rows, 11 at 4096, 17 at 8192 (profiles/r06_neumf_small_steps.txt) - beyond that the counting pass below.
template <int NT>
__global__ __launch_bounds__(kScatterSmallRows) void k_nmf_scatter_scan(daisy_neumf_params p, daisy_neumf_params g, PairSrc src,
                                                                       int R, int d, int dm, int model, int pointwise,
                                                                       const float *__restrict__ dpred,
                                                                       const float *__restrict__ DX0,
                                                                       const double *__restrict__ stats, float reg_1,
                                                                       float reg_2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];
#ifdef DAISY_SCAN_PROF
    long long sprof[6] = {0, 0, 0, 0, 0, 0}, spt0 = wall_clock64();
#define SCAN_MARK(k) { const long long now_ = wall_clock64(); sprof[k] += now_ - spt0; spt0 = now_; }
#else
#define SCAN_MARK(k)
#endif
    const int side = blockIdx.x, tid = threadIdx.x;
    const int rounds = (R + 63) / 64;
    const int gpb = (int)blockDim.x / 16;              // 16-lane groups (= step rows) of this workgroup: 16, 32 or 64
    // (dynamic LDS, sized by the step: per 16-lane group and round one 64-bit mask and one round number - only the rounds with a
    // match are kept - and three words per row: 88 KB at kScanMaxRows)
    uint64_t *mask_all = reinterpret_cast<uint64_t *>(scan_lds);                      // [64 groups][rounds]
    uint32_t *key_s = reinterpret_cast<uint32_t *>(mask_all + (size_t)gpb * rounds);    // this side's table row of a step row
    uint32_t *oth_s = key_s + rounds * 64;                                            // the other side's
    float *dp_s = reinterpret_cast<float *>(oth_s + rounds * 64);
    uint16_t *rnd_all = reinterpret_cast<uint16_t *>(dp_s + rounds * 64);             // [64 groups][rounds]
    // (the norms first: their loads fly with the ids' - read after the scan they were a memory round trip of their own)
    const double nrm_m = stats[DAISY_NST_NORM + (side ? 3 : 1)], nrm_g = stats[DAISY_NST_NORM + (side ? 2 : 0)], nrm_neg = stats[DAISY_NST_NORM + 4];
    for (int t = tid; t < rounds * 64; t += (int)blockDim.x) {
        uint32_t own = 0xFFFFFFFFu, oth = 0u;
        float dp = 0.f;
        if (t < R) {
            int64_t user, item;
            pair_ids(src, t, user, item);
            own = (uint32_t)(side ? item : user);
            oth = (uint32_t)(side ? user : item);
            dp = dpred[t];
        }
        key_s[t] = own; oth_s[t] = oth; dp_s[t] = dp;
    }
    __syncthreads();
    SCAN_MARK(0)
    const int lane = tid % 16, group = tid / 16;
    // The scan: a wave compares 64 keys per round with the keys of ITS four rows - one LDS read, four compares, four ballots,
    // and a ballot IS the round's mask of matching rows; four rounds' reads are issued together (a round on its own is one LDS
    // latency: 22 us for 4096 rows).  Rounds without a match are not kept; the occurrence counts of the regulariser
    // (rows < B: positives) are taken from the masks as they pass.  A row with a match before itself is not the first occurrence
    // of its key: it owns nothing.
    __shared__ int cnt_s[kScatterSmallRows / 16], npos_s[kScatterSmallRows / 16], nneg_s[kScatterSmallRows / 16];
    {
        constexpr int Q = kWave / 16;
        // (the wave's number through readfirstlane: everything derived from it - its rows, their flags and counters - is then
        // scalar for the compiler too; as lane-derived values they were carried in VGPRs with exec-mask branches around
        // every step, ~110 instructions per round and row)
        const int lane64 = tid % kWave, wave = __builtin_amdgcn_readfirstlane(tid / kWave);
        const int g0 = wave * Q, e0 = (int)blockIdx.y * gpb + g0;
        uint32_t rowk[Q];
        int cnt[Q], np_[Q], nn_[Q];
        bool early[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { rowk[q] = key_s[(e0 + q < R) ? e0 + q : 0]; cnt[q] = 0; np_[q] = 0; nn_[q] = 0; early[q] = e0 + q >= R; }
        for (int rb = 0; rb < rounds; rb += 4) {
            uint32_t kk[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) kk[x] = (rb + x < rounds) ? key_s[(rb + x) * 64 + lane64] : 0xFFFFFFFEu;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int rd = rb + x;
                if (rd >= rounds) break;
                uint64_t mq[Q], any = 0;
#pragma unroll
                for (int q = 0; q < Q; ++q) { mq[q] = __ballot(kk[x] == rowk[q]); any |= mq[q]; }
                if (any == 0) continue;                                               // (most rounds: one branch for the four rows)
                const int64_t npos_bits = (int64_t)src.B - (int64_t)rd * 64;          // positions of this round that are positive rows
                const uint64_t posm = npos_bits >= 64 ? ~0ull : (npos_bits <= 0 ? 0ull : (((uint64_t)1 << npos_bits) - 1));
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const uint64_t m = mq[q];
                    if (m == 0 || early[q]) continue;
                    const int before = e0 + q - rd * 64;                              // positions of this round before the row itself
                    if (before >= 64 || (before > 0 && (m & (((uint64_t)1 << before) - 1)) != 0)) { early[q] = true; continue; }
                    if (lane64 == 0) { mask_all[(size_t)(g0 + q) * rounds + cnt[q]] = m; rnd_all[(size_t)(g0 + q) * rounds + cnt[q]] = (uint16_t)rd; }
                    ++cnt[q];
                    np_[q] += (int)__popcll(m & posm);
                    nn_[q] += (int)__popcll(m & ~posm);
                }
            }
        }
        if (lane64 == 0)
#pragma unroll
            for (int q = 0; q < Q; ++q) { cnt_s[g0 + q] = early[q] ? -1 : cnt[q]; npos_s[g0 + q] = np_[q]; nneg_s[g0 + q] = nn_[q]; }
    }
    // (the wave that wrote a row's masks is the wave its 16-lane group belongs to: no workgroup barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    SCAN_MARK(1)
    const int e = (int)blockIdx.y * gpb + group;
    if (e >= R) return;
    const int nent = cnt_s[group];
    if (nent < 0) return;                                   // the first occurrence owns the table row
    const uint32_t row = key_s[e];
    const uint64_t *mask_s = mask_all + (size_t)group * rounds;       // this group's kept rounds: masks and round numbers
    const uint16_t *rnd_s = rnd_all + (size_t)group * rounds;
    const float npos = (float)npos_s[group], nneg = (float)nneg_s[group];
    SCAN_MARK(2)
    auto inv = [&](double n) { return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f; };
    const float i_m = inv(nrm_m), i_g = inv(nrm_g), i_neg = 2.f * inv(nrm_neg);
    const float *tabM = side ? p.iM : p.uM, *tabG = side ? p.iG : p.uG, *otherG = side ? p.uG : p.iG;
    float *gM = side ? g.iM : g.uM, *gG = side ? g.iG : g.uG;
    // The table row's dm + d columns as float4 chunks, NT per lane, all of a step row's chunks loaded at once: one dependent
    // memory access (0.2 - 0.4 us: profiles/r06_latency_probe.txt) per matching step row plus one for the table rows and the
    // gradient rows, instead of three per 16 columns (first version: 20 us at factors 24, this one 10.5).  Per element the same operations
    // in the same order as k_nmf_scatter_small.
    const int mch = dm / 4, nch = mch + d / 4;         // chunks 0 .. mch-1: the MLP row; mch .. nch-1: the GMF row
    float4 acc[NT], tw[NT], gw[NT], wpv[NT <= 2 ? NT : 1];     // (NT = 5: 128 registers per lane at 1024 threads - Wp is read late there)
    int cidx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = lane + 16 * t;
        cidx[t] = (ch < nch) ? ch : -1;
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int cc = (ch < nch) ? ch : 0;
        const float *trow = (cc < mch) ? tabM + (int64_t)row * dm + 4 * cc : tabG + (int64_t)row * d + 4 * (cc - mch);
        const float *grow_ = (cc < mch) ? gM + (int64_t)row * dm + 4 * cc : gG + (int64_t)row * d + 4 * (cc - mch);
        tw[t] = *reinterpret_cast<const float4 *>(trow);
        gw[t] = *reinterpret_cast<const float4 *>(grow_);
        if constexpr (NT <= 2) wpv[t] = *reinterpret_cast<const float4 *>(p.Wp + ((cc < mch) ? 0 : 4 * (cc - mch)));
    }
    for (int c = 0; c < nent; ++c)
        for (uint64_t m = mask_s[c]; m; m &= m - 1) {
            const int r = (int)rnd_s[c] * 64 + (int)__builtin_ctzll(m);
            const float dp = dp_s[r];
            const float *xrow = DX0 + (int64_t)r * (2 * dm) + side * dm, *orow = otherG + (int64_t)oth_s[r] * d;
            float4 v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int cc = cidx[t] < 0 ? 0 : cidx[t];
                v[t] = *reinterpret_cast<const float4 *>((cc < mch) ? xrow + 4 * cc : orow + 4 * (cc - mch));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (cidx[t] < 0) continue;
                if (cidx[t] < mch) {
                    if (model != DAISY_NEUMF_GMF) { acc[t].x += v[t].x; acc[t].y += v[t].y; acc[t].z += v[t].z; acc[t].w += v[t].w; }
                } else if (model != DAISY_NEUMF_MLP) {
                    acc[t].x = fmaf(dp, v[t].x, acc[t].x); acc[t].y = fmaf(dp, v[t].y, acc[t].y);
                    acc[t].z = fmaf(dp, v[t].z, acc[t].z); acc[t].w = fmaf(dp, v[t].w, acc[t].w);
                }
            }
        }
    SCAN_MARK(3)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (cidx[t] < 0) continue;
        float a4[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
        const float w4[4] = {tw[t].x, tw[t].y, tw[t].z, tw[t].w};
        float o4[4] = {gw[t].x, gw[t].y, gw[t].z, gw[t].w};
        if (cidx[t] < mch) {
            // MLP table: the rows' input gradients (this side's half of dX0), the regulariser on the positive rows' occurrences
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = a4[k];
                if (npos > 0.f) v += fmaf(npos * i_m, w4[k], reg_1 * npos * sgn(w4[k]));
                if (v != 0.f) o4[k] += v;
            }
            *reinterpret_cast<float4 *>(gM + (int64_t)row * dm + 4 * cidx[t]) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        } else {
            // GMF table: Wp[c] x sum of dpred[r] x the OTHER table's row; the negative item's rows count twice in the regulariser
            const int c0 = 4 * (cidx[t] - mch);
            float4 wq;
            if constexpr (NT <= 2) wq = wpv[t]; else wq = *reinterpret_cast<const float4 *>(p.Wp + c0);
            const float wp4[4] = {wq.x, wq.y, wq.z, wq.w};
            const float na = npos, nb = (side && !pointwise) ? nneg : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = a4[k];
                if (model != DAISY_NEUMF_MLP) v *= wp4[k];
                if (na + nb > 0.f) v += fmaf(na * i_g + nb * i_neg, w4[k], reg_1 * (na + 2.f * nb) * sgn(w4[k]));
                if (v != 0.f) o4[k] += v;
            }
            *reinterpret_cast<float4 *>(gG + (int64_t)row * d + c0) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
    }
#ifdef DAISY_SCAN_PROF
    SCAN_MARK(4)
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        printf("k_nmf_scatter_scan block 0 group 0, x10 ns: ids->LDS %lld  scan %lld  counts %lld  table rows + matches %lld  commit %lld (R %d)\n",
               sprof[0], sprof[1], sprof[2], sprof[3], sprof[4], R);
#endif
}

// ---------------------------------------------------------------------------------------------
// Round 6: the step's rows grouped by user and by item with ONE stable counting pass per side instead of two radix sorts
// (rocprim: two digit passes + histogram + ~5 memsets per sort - 88 us per side at 524 288 rows, a launch chain, not
// bandwidth).  The table has a few thousand rows (ml-1m: 6040 / 3706), so a whole histogram fits a wave's share of LDS:
//   k_cs_count    each wave counts the keys of ITS contiguous range of rows (LDS atomics: counts do not depend on order)
//   k_cs_prefix   per key: exclusive prefix over the waves' counts, in wave order (= row order); the pos / neg halves' totals
//                 are the regulariser's occurrence counts (what k_nmf_sort_keys counted with global atomics)
//   k_cs_base     exclusive scan of the keys' totals
//   k_cs_entries  the slots' rows -> the segmented reductions' entry lists, with coalesced stores
//   k_cs_scatter  each wave walks its rows in order, 64 at a time: a row's slot = base[key] + the waves before + the rows of
//                 this wave before it with the same key (ballot match inside the 64, a running LDS counter across them)
// Stable by construction - rows of one key stay in ascending row order - hence the same bits as the radix sorts' output.
// Both sides (users, items) ride in the same four launches (blockIdx.y).
// ---------------------------------------------------------------------------------------------
constexpr int kCsBlocks = 128, kCsWaves = kBlock / kWave, kCsNW = kCsBlocks * kCsWaves;      // 512 wave ranges per side
constexpr int kCsMaxKeys = 9600;                                                             // 4 waves x keys x 4 B <= 150 KB of LDS

struct CsRange { int64_t lo, hi; };
// wave range gw of a step of R rows: the pos half [0, B) and the neg half [B, R) are cut separately (so that a half's counts
// are whole waves); point-wise steps have one half
__device__ __forceinline__ CsRange cs_range(int gw, int64_t R, int64_t B, int halves) {
    const int wph = kCsNW / halves, hf = gw / wph, within = gw % wph;
    const int64_t len = (halves == 2) ? ((hf == 0) ? B : R - B) : R, base = (halves == 2 && hf == 1) ? B : 0;
    const int64_t chunk = ((len + wph - 1) / wph + kWave - 1) / kWave * kWave;
    int64_t lo = base + within * chunk, hi = lo + chunk;
    if (lo > base + len) lo = base + len;
    if (hi > base + len) hi = base + len;
    return CsRange{lo, hi};
}
__device__ __forceinline__ int32_t cs_key(const PairSrc &src, int64_t r, int side) {
    int64_t user, item;
    pair_ids(src, r, user, item);
    return (int32_t)(side ? item : user);
}

__global__ __launch_bounds__(kBlock) void k_cs_count(PairSrc src, int64_t R, int halves, int Ku, int Ki, int kstride,
                                                     int32_t *__restrict__ hist) {
    extern __shared__ int32_t cs_lds[];
    const int side = blockIdx.y, K = side ? Ki : Ku;
    // (the wave's number through readfirstlane: its range and the loops over it are then scalar for the compiler too)
    const int lane = threadIdx.x % kWave, w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), gw = blockIdx.x * kCsWaves + w;
    int32_t *h = cs_lds + w * kstride;
    for (int k = lane; k < K; k += kWave) h[k] = 0;
    const CsRange rg = cs_range(gw, R, src.B, halves);
    for (int64_t r0 = rg.lo + lane; r0 < rg.hi; r0 += 4 * kWave) {          // four rows' ids in flight per lane
        int32_t key[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) key[x] = (r0 + x * kWave < rg.hi) ? cs_key(src, r0 + x * kWave, side) : -1;
#pragma unroll
        for (int x = 0; x < 4; ++x) if (key[x] >= 0) atomicAdd(&h[key[x]], 1);
    }
    // (a wave's LDS operations complete in order: no barrier between its own adds and reads)
    int32_t *out = hist + ((int64_t)side * kCsNW + gw) * kstride;
    for (int k = lane; k < K; k += kWave) out[k] = h[k];
}

__global__ __launch_bounds__(kBlock) void k_cs_prefix(int halves, int Ku, int Ki, int kstride, int32_t *__restrict__ hist,
                                                      int32_t *__restrict__ total, int32_t *__restrict__ cnt_u,
                                                      int32_t *__restrict__ cnt_i, int32_t *__restrict__ cnt_j) {
    const int side = blockIdx.y, K = side ? Ki : Ku;
    const int key = blockIdx.x * kBlock + threadIdx.x;
    if (key >= K) return;
    int32_t *col = hist + (int64_t)side * kCsNW * kstride + key;
    const int wph = kCsNW / halves;
    int32_t run = 0, first_half = 0;
    static_assert(kCsNW % 64 == 0, "the prefix walks the wave ranges 32 at a time, and a half is a whole number of such groups");
    for (int g0 = 0; g0 < kCsNW; g0 += 32) {                    // 32 independent loads in flight (8: 64 dependent round trips, 28 us), then the running sum
        int32_t cnt[32];
#pragma unroll
        for (int x = 0; x < 32; ++x) cnt[x] = col[(int64_t)(g0 + x) * kstride];
#pragma unroll
        for (int x = 0; x < 32; ++x) { col[(int64_t)(g0 + x) * kstride] = run; run += cnt[x]; }
        if (g0 + 32 == wph) first_half = run;
    }
    if (halves == 1) first_half = run;
    total[side * kstride + key] = run;
    // the regulariser's occurrence counts (NeuMFRecommender.py:149-167): users / items of the positive rows, items of the negatives
    if (side == 0) cnt_u[key] = first_half;
    else { cnt_i[key] = first_half; cnt_j[key] = run - first_half; }
}

// exclusive scan of total[side][0 .. K) in place (one workgroup per side; K <= kCsMaxKeys)
__global__ __launch_bounds__(1024) void k_cs_base(int Ku, int Ki, int kstride, int32_t *__restrict__ total) {
    __shared__ int32_t part[1024];
    const int side = blockIdx.x, K = side ? Ki : Ku, tid = threadIdx.x;
    int32_t *t = total + side * kstride;
    const int per = (K + 1023) / 1024, lo = tid * per, hi = (lo + per < K) ? lo + per : K;
    int32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += t[k];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                 // Hillis-Steele over the 1024 partial sums
        const int32_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int32_t run = part[tid] - sum;                             // exclusive
    for (int k = lo; k < hi; ++k) { const int32_t c = t[k]; t[k] = run; run += c; }
}

// The grouped rows leave as the segmented reduction's entry lists (what k_nmf_entries / k_nmf_entries_gmf wrote in four launches
// of their own): per side, entry e -> key = table row << 1; MLP list: source row = mlp_rows_per * r + half, weight 1; GMF list:
// source row = the OTHER id of row r, weight dpred[r].  An odd count is padded with a weightless copy of the last entry.
struct CsEntries { uint32_t *ekey; uint2 *esu_m; float2 *w_m; uint2 *esu_g; float2 *w_g; };
__global__ __launch_bounds__(kBlock) void k_cs_scatter(PairSrc src, int64_t R, int halves, int Ku, int Ki, int kstride,
                                                       const int32_t *__restrict__ hist, const int32_t *__restrict__ total,
                                                       CsEntries eu, CsEntries ei, int mlp_rows_per, int mlp_half_by_side,
                                                       const float *__restrict__ dpred) {
    extern __shared__ int32_t cs_lds[];
    const int side = blockIdx.y, K = side ? Ki : Ku;
    const int lane = threadIdx.x % kWave, w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), gw = blockIdx.x * kCsWaves + w;
    int32_t *off = cs_lds + w * kstride;
    const int32_t *mine = hist + ((int64_t)side * kCsNW + gw) * kstride, *base = total + side * kstride;
    for (int k = lane; k < K; k += kWave) off[k] = base[k] + mine[k];
    const CsEntries en = side ? ei : eu;
    (void)mlp_rows_per; (void)mlp_half_by_side; (void)dpred;          // (the entries themselves: k_cs_entries)
    const CsRange rg = cs_range(gw, R, src.B, halves);
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    for (int64_t rb = rg.lo; rb < rg.hi; rb += 4 * kWave) {
      int32_t keys[4];                                         // the ids of four 64-row groups in flight
#pragma unroll
      for (int x = 0; x < 4; ++x) keys[x] = (rb + x * kWave + lane < rg.hi) ? cs_key(src, rb + x * kWave + lane, side) : -1;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int64_t r = rb + x * kWave + lane;
        const bool valid = r < rg.hi;
        const int32_t key = keys[x];
        if (rb + x * kWave >= rg.hi) break;
        uint64_t peers = __ballot(valid);                      // lanes of this 64 with the same key
#pragma unroll
        for (int b = 0; b < 14; ++b) {
            const uint64_t m = __ballot((key >> b) & 1);
            peers &= ((key >> b) & 1) ? m : ~m;
        }
        if (valid) {
            // the row's number into its slot (ONE scattered 4-byte store per row; k_cs_entries turns the slots into entries with
            // coalesced stores - writing the five entry arrays from here was five scattered partial-line stores per row: 83 us)
            const int32_t slot = off[key] + (int32_t)__popcll(peers & lt);
            en.ekey[slot] = (uint32_t)r;
            if ((peers & lt) == 0) off[key] += (int32_t)__popcll(peers);      // one lane per key moves the running counter
        }
      }
    }
}

// slot e of a side (holding the step row k_cs_scatter put there) -> the segmented reductions' entries, both lists
__global__ __launch_bounds__(kBlock) void k_cs_entries(PairSrc src, int64_t R, CsEntries eu, CsEntries ei, int mlp_rows_per,
                                                       int mlp_half_by_side, const float *__restrict__ dpred) {
    const int side = blockIdx.y;
    const CsEntries en = side ? ei : eu;
    const int half = mlp_half_by_side ? side : 0;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < R; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = en.ekey[e];
        int64_t user, item;
        pair_ids(src, r, user, item);
        const uint32_t ek = (uint32_t)(side ? item : user) << 1, sm = (uint32_t)(mlp_rows_per * (int32_t)r + half),
                       sg = (uint32_t)(side ? user : item);
        const float dp = dpred ? dpred[r] : 0.f;
        en.ekey[e] = ek;
        en.esu_m[e] = make_uint2((uint32_t)e, sm); en.w_m[e] = make_float2(1.f, 0.f);
        en.esu_g[e] = make_uint2((uint32_t)e, sg); en.w_g[e] = make_float2(dp, 0.f);
        if ((R & 1) && e == R - 1) {                       // the weightless copy that makes the count even
            en.ekey[R] = ek;
            en.esu_m[R] = make_uint2((uint32_t)R, sm); en.w_m[R] = make_float2(0.f, 0.f);
            en.esu_g[R] = make_uint2((uint32_t)R, sg); en.w_g[R] = make_float2(0.f, 0.f);
        }
    }
}

__global__ void k_nmf_sort_keys(PairSrc src, int64_t R, int32_t *__restrict__ ku, int32_t *__restrict__ ki,
                                int32_t *__restrict__ val, int pointwise, int32_t *__restrict__ cnt_u,
                                int32_t *__restrict__ cnt_i, int32_t *__restrict__ cnt_j) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t user, item;
        pair_ids(src, r, user, item);
        ku[r] = (int32_t)user;
        ki[r] = (int32_t)item;
        val[r] = (int32_t)r;
        if (r < src.B) { atomicAdd(cnt_u + user, 1); atomicAdd(cnt_i + item, 1); }     // regulariser occurrences (:149-167)
        else if (!pointwise) atomicAdd(cnt_j + item, 1);
    }
}

// entry e of a sorted list -> the segmented reduction's view: key = table row << 1, source row = rows_per*r + half
__global__ void k_nmf_entries(const int32_t *__restrict__ key_sorted, const int32_t *__restrict__ val_sorted, int64_t R,
                              int64_t n_pad, int rows_per, int half, uint32_t *__restrict__ ekey,
                              uint2 *__restrict__ esu, float2 *__restrict__ w) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_pad; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = e < R ? e : R - 1;               // an odd count is padded with a weightless copy of the last entry
        ekey[e] = (uint32_t)key_sorted[q] << 1;
        esu[e] = make_uint2((uint32_t)e, (uint32_t)(rows_per * val_sorted[q] + half));
        w[e] = make_float2(e < R ? 1.f : 0.f, 0.f);
    }
}

// GMF branch: d/d uG[user] = Wp * sum over the user's rows of dpred[r] * iG[item_r]  (and the mirror image for iG).  The
// sum is a segmented reduction over the rows sorted by user whose SOURCE rows are the other table's - cache-resident - rows
// and whose weights are dpred[r]: entry e -> (key = table row << 1, source row = the other id of row r, weight dpred[r]);
// Wp multiplies the finished sum (k_nmf_table_commit).  Until round 5 the per-row products were materialised first (two
// [R, d] fp32 arrays written by a kernel of their own and read back by the reductions: 0.4 GB per step at R = 524 288).
__global__ void k_nmf_entries_gmf(const int32_t *__restrict__ key_sorted, const int32_t *__restrict__ val_sorted, int64_t R,
                                  int64_t n_pad, PairSrc src, int side, const float *__restrict__ dpred,
                                  uint32_t *__restrict__ ekey, uint2 *__restrict__ esu, float2 *__restrict__ w) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_pad; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = e < R ? e : R - 1;               // an odd count is padded with a weightless copy of the last entry
        const int64_t r = val_sorted[q];
        int64_t user, item;
        pair_ids(src, r, user, item);
        ekey[e] = (uint32_t)key_sorted[q] << 1;
        esu[e] = make_uint2((uint32_t)e, (uint32_t)(side ? user : item));
        w[e] = make_float2(e < R ? dpred[r] : 0.f, 0.f);
    }
}

// g[row] += sum[row] (clearing sum) + (ca*ia + cb*ib) * w[row] + reg_1*(ca + cb) * sign(w[row]); counts cleared
__global__ __launch_bounds__(kBlock) void k_nmf_table_commit(float *__restrict__ g, float *__restrict__ sum,
                                                             const float *__restrict__ w, int64_t rows, int width,
                                                             int32_t *__restrict__ ca, int ka, int32_t *__restrict__ cb,
                                                             int kb, float scale_b, const double *__restrict__ stats,
                                                             float reg_1, float reg_2, int clear_counts,
                                                             const float *__restrict__ colscale = nullptr) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    auto inv = [&](int k) { const double n = stats[DAISY_NST_NORM + k]; return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f; };
    const float ia = inv(ka), ib = cb ? scale_b * inv(kb) : 0.f;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 16) + group; row < rows; row += gstride) {
        const float na = (float)ca[row], nb = cb ? (float)cb[row] : 0.f;
        const float r2 = na * ia + nb * ib, r1 = reg_1 * (na + scale_b * nb);
        for (int c = lane; c < width; c += 16) {
            const int64_t x = row * (int64_t)width + c;
            float v = 0.f;
            if (sum) { v = colscale ? sum[x] * colscale[c] : sum[x]; sum[x] = 0.f; }      // (GMF tables: Wp x the summed rows)
            if (na + nb > 0.f) { const float e = w[x]; v += fmaf(r2, e, r1 * sgn(e)); }
            if (v != 0.f) g[x] += v;
        }
        if (clear_counts && lane == 0) { ca[row] = 0; if (cb) cb[row] = 0; }
    }
}

// the same with 16-byte accesses (width % 4 == 0, 16-byte aligned tables): a lane takes 4 consecutive columns - the scalar form
// above walks a 256-column row in 16 dependent trips per lane and cost 15-19 us per table for 6 MB
__global__ __launch_bounds__(kBlock) void k_nmf_table_commit_v(float *__restrict__ g, float *__restrict__ sum,
                                                               const float *__restrict__ w, int64_t rows, int width,
                                                               int32_t *__restrict__ ca, int ka, int32_t *__restrict__ cb,
                                                               int kb, float scale_b, const double *__restrict__ stats,
                                                               float reg_1, float reg_2, int clear_counts,
                                                               const float *__restrict__ colscale) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    auto inv = [&](int k) { const double n = stats[DAISY_NST_NORM + k]; return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f; };
    const float ia = inv(ka), ib = cb ? scale_b * inv(kb) : 0.f;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 16) + group; row < rows; row += gstride) {
        const float na = (float)ca[row], nb = cb ? (float)cb[row] : 0.f;
        const float r2 = na * ia + nb * ib, r1 = reg_1 * (na + scale_b * nb);
        const bool reg = na + nb > 0.f;
        if (!sum && !reg) continue;
        for (int c = 4 * lane; c < width; c += 64) {
            const int64_t x = row * (int64_t)width + c;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (sum) {
                const float4 sv = *reinterpret_cast<const float4 *>(sum + x);
                v[0] = sv.x; v[1] = sv.y; v[2] = sv.z; v[3] = sv.w;
                if (colscale) {
                    const float4 cs = *reinterpret_cast<const float4 *>(colscale + c);
                    v[0] *= cs.x; v[1] *= cs.y; v[2] *= cs.z; v[3] *= cs.w;
                }
                *reinterpret_cast<float4 *>(sum + x) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (reg) {
                const float4 ev = *reinterpret_cast<const float4 *>(w + x);
                const float e[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += fmaf(r2, e[k], r1 * sgn(e[k]));
            }
            float4 gv = *reinterpret_cast<float4 *>(g + x);
            gv.x += v[0]; gv.y += v[1]; gv.z += v[2]; gv.w += v[3];      // (+ 0 where nothing arrived: the bits of g stay)
            *reinterpret_cast<float4 *>(g + x) = gv;
        }
        if (clear_counts && lane == 0) { ca[row] = 0; if (cb) cb[row] = 0; }
    }
}

// Both tables of a side - MLP then GMF - for both sides in ONE launch (blockIdx.y: the side): the four commits of a step were
// four launches of ~6.5 us each, mostly latency.  A lane group takes a table row of its side and commits its MLP row, its GMF
// row, then clears the row's occurrence counts (both commits read them).  Per element the operations of k_nmf_table_commit_v.
struct CommitSide {
    float *gM, *sumM; const float *wM; int widthM, kM;            // MLP table: gradient, row sums (or null), weights, columns, norm slot
    float *gG, *sumG; const float *wG; int widthG, kG;            // GMF table
    int64_t rows;
    int32_t *ca, *cb;                                             // occurrences: positives; negatives (items' GMF rows only, or null)
    const float *colscale;                                        // Wp over the GMF sums (or null)
};
__global__ __launch_bounds__(kBlock) void k_nmf_table_commit_pair(CommitSide su, CommitSide si, const double *__restrict__ stats,
                                                                  float reg_1, float reg_2) {
    const CommitSide &j = blockIdx.y ? si : su;
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    const int64_t gstride = (int64_t)gridDim.x * (kBlock / 16);
    auto inv = [&](int k) { const double n = stats[DAISY_NST_NORM + k]; return (n > 0.0) ? (float)((double)reg_2 / n) : 0.f; };
    const float iM = inv(j.kM), iG = inv(j.kG), iN = j.cb ? 2.f * inv(4) : 0.f;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 16) + group; row < j.rows; row += gstride) {
        const float na = (float)j.ca[row], nb = j.cb ? (float)j.cb[row] : 0.f;
        auto commit = [&](float *g, float *sum, const float *w, int width, float r2, float r1, bool reg, const float *colscale) {
            if (!sum && !reg) return;
            for (int c = 4 * lane; c < width; c += 64) {
                const int64_t x = row * (int64_t)width + c;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (sum) {
                    const float4 sv = *reinterpret_cast<const float4 *>(sum + x);
                    v[0] = sv.x; v[1] = sv.y; v[2] = sv.z; v[3] = sv.w;
                    if (colscale) {
                        const float4 cs = *reinterpret_cast<const float4 *>(colscale + c);
                        v[0] *= cs.x; v[1] *= cs.y; v[2] *= cs.z; v[3] *= cs.w;
                    }
                    *reinterpret_cast<float4 *>(sum + x) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (reg) {
                    const float4 ev = *reinterpret_cast<const float4 *>(w + x);
                    const float e[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += fmaf(r2, e[k], r1 * sgn(e[k]));
                }
                float4 gv = *reinterpret_cast<float4 *>(g + x);
                gv.x += v[0]; gv.y += v[1]; gv.z += v[2]; gv.w += v[3];      // (+ 0 where nothing arrived: the bits of g stay)
                *reinterpret_cast<float4 *>(g + x) = gv;
            }
        };
        // (k_nmf_table_commit_v's r2 = na * ia + nb * ib, r1 = reg_1 * (na + scale_b * nb) with ib = scale_b * inv(kb))
        commit(j.gM, j.sumM, j.wM, j.widthM, na * iM + 0.f * 0.f, reg_1 * (na + 0.f * 0.f), na + 0.f > 0.f, nullptr);
        commit(j.gG, j.sumG, j.wG, j.widthG, na * iG + nb * iN, reg_1 * (na + 2.f * nb), na + nb > 0.f, j.colscale);
        if (lane == 0) { j.ca[row] = 0; if (j.cb) j.cb[row] = 0; }
    }
}

static void launch_table_commit(float *g, float *sum, const float *w, int64_t rows, int width, int32_t *ca, int ka, int32_t *cb,
                                int kb, float scale_b, const double *stats, float reg_1, float reg_2, int clear_counts,
                                const float *colscale, hipStream_t s) {
    auto al = [](const void *p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    if (width % 4 == 0 && al(g) && al(sum) && al(w) && al(colscale))
        hipLaunchKernelGGL(k_nmf_table_commit_v, dim3(grid_for(rows, kBlock / 16)), dim3(kBlock), 0, s, g, sum, w, rows, width, ca, ka,
                           cb, kb, scale_b, stats, reg_1, reg_2, clear_counts, colscale);
    else
        hipLaunchKernelGGL(k_nmf_table_commit, dim3(grid_for(rows, kBlock / 16 * 2)), dim3(kBlock), 0, s, g, sum, w, rows, width, ca,
                           ka, cb, kb, scale_b, stats, reg_1, reg_2, clear_counts, colscale);
}

__global__ __launch_bounds__(kBlock) void k_sgd_dense(float *__restrict__ W, float *__restrict__ g, int64_t n,
                                                      float lr) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        W[e] = fmaf(-lr, g[e], W[e]);
        g[e] = 0.f;
    }
}

}  // namespace daisy

using namespace daisy;

struct daisy_neumf_ctx {
    int64_t max_rows, U, I;
    int d, L, dm, model;
    int width[DAISY_NEUMF_MAX_LAYERS + 1];   // width[0] = 2*dm, width[l] = width[l-1]/2
    void *arena;
    size_t arena_bytes;
    float *X[DAISY_NEUMF_MAX_LAYERS + 1];    // X[0] = (dropped) concat input, X[l] = layer outputs
    float *G, *pred, *dpred, *DZ[2];
    uint16_t *W16[DAISY_NEUMF_MAX_LAYERS];   // bf16 copies of the MLP weights (precision level 2), refreshed per call
    uint16_t *W16T[DAISY_NEUMF_MAX_LAYERS];  // ... and their transposes [n_in][n_out]
    // scratch of the owner-based embedding scatter (allocated at its first use)
    void *sc_arena;
    int32_t *sc_ku, *sc_ki, *sc_val, *sc_ks, *sc_vs, *sc_cu, *sc_ci, *sc_cj;
    uint32_t *sc_ekey; uint2 *sc_esu; float2 *sc_w;
    float *sc_sum, *sc_sum2, *sc_sumg, *sc_sumg2, *sc_edge_vec, *sc_edge_b;      // row sums: MLP users, MLP items, GMF users, GMF items
    int32_t *sc_edge_item, *sc_edge_whole;
    void *sc_tmp; size_t sc_tmp_bytes;
    void *cs_ent;                            // counting pass: the two sides' entry lists (CsEntries)
    int32_t *cs_hist;                        // counting pass: [2 sides][kCsNW waves][key stride] counts -> prefixes, then [2][stride] totals
    int bf16;                                // daisy_neumf_ctx_set_precision: 0 fp32, 1 bf16 MFMA inputs, 2 bf16 storage
    // per-workgroup partial sums of the reductions over the batch rows (split-K slices of the weight-gradient GEMMs,
    // row tiles of the column sums ...), added in a fixed order by k_reduce_slices; allocated at the first training step
    float *det_ws;
    size_t det_ws_floats;
    float *fact_t;                           // T_u [U][n1] then T_i [I][n1]: the first layer through the tables (k_nmf_gather<FACT>)
    bool tower_aligned;                      // W2 / W3 of the current call are 16-byte aligned (the tower reads them as float4)
    bool mid_fits, mid_aligned;              // k_nmf_mid: the layers fit the LDS (ctx_create); this call's weights are 16-byte aligned
    Fact fact_cur;                           // ... as the forward pass of the current step set them up (the fused tower reads them)
};

constexpr int kWgradChunkDefault = 2048;
static int wgrad_chunk() {
    static const int v = getenv("DAISY_WGRAD_CHUNK") ? atoi(getenv("DAISY_WGRAD_CHUNK")) : kWgradChunkDefault;
    return v > 0 ? v : kWgradChunkDefault;
}

constexpr int kMidMaxRows = 8192;           // steps k_nmf_mid takes (csrc/neumf_mid.hip; = kScanMaxRows: its scatter)
static int neumf_need_det_ws(daisy_neumf_ctx *ctx) {
    if (ctx->det_ws) return DAISY_OK;
    const size_t splits = ((size_t)ctx->max_rows + wgrad_chunk() - 1) / wgrad_chunk();
    size_t layer = 1024;                               // (predict layer: <= 1024 workgroups x 512 columns, covered below)
    for (int l = 1; l <= ctx->L; ++l) {
        const size_t e = (size_t)ctx->width[l] * ctx->width[l - 1];
        if (e > layer) layer = e;
    }
    size_t n = splits * layer;
    const size_t pred = (size_t)1024 * 512;
    if (pred > n) n = pred;
    // the fused tower (csrc/neumf_tower.hip): one slab of partial sums per workgroup
    const size_t tower = (neumf_tower_ws_bytes(ctx->d, neumf_tower_blocks((ctx->max_rows + 63) / 64)) + 3) / 4;
    if (tower > n) n = tower;
    const int rows_mid = ctx->max_rows < kMidMaxRows ? (int)ctx->max_rows : kMidMaxRows;
    const size_t mid = (neumf_mid_ws_bytes(ctx->L, ctx->width, ctx->d, rows_mid) + 3) / 4;
    if (mid > n) n = mid;
    hipError_t e = hipMalloc((void **)&ctx->det_ws, n * sizeof(float));
    if (e != hipSuccess) {
        ctx->det_ws = nullptr;
        set_error("neumf: hipMalloc(%zu) for the reduction workspace failed: %s", n * sizeof(float), hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    ctx->det_ws_floats = n;
    return DAISY_OK;
}

// Many slices (the split-K weight gradients: 256 slices of up to 131 072 elements; the column sums: 1024 slices of 64-256):
// one thread per element walking all slices left most of the chip idle with four loads in flight per thread - 33.5 MB in
// 66 us, and ONE workgroup for the column sums.  Here a workgroup takes C elements and its 256 / C thread groups every
// (256 / C)-th slice each; the partial sums meet in LDS in group order: still one fixed association per element.
template <int C>
__global__ __launch_bounds__(kBlock) void k_reduce_slices_wide(const float *__restrict__ ws, int nslices, int64_t len,
                                                               float *__restrict__ out) {
    constexpr int G = kBlock / C;
    __shared__ float sm[G][C];
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    for (int64_t c0 = (int64_t)blockIdx.x * C; c0 < len; c0 += (int64_t)gridDim.x * C) {
        const int64_t col = c0 + c;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        if (col < len) {
            int sidx = g;
            for (; sidx + 3 * G < nslices; sidx += 4 * G) {
                t0 += ws[(int64_t)sidx * len + col];
                t1 += ws[(int64_t)(sidx + G) * len + col];
                t2 += ws[(int64_t)(sidx + 2 * G) * len + col];
                t3 += ws[(int64_t)(sidx + 3 * G) * len + col];
            }
            for (; sidx < nslices; sidx += G) t0 += ws[(int64_t)sidx * len + col];
        }
        sm[g][c] = (t0 + t1) + (t2 + t3);
        __syncthreads();
        if (g == 0 && col < len) {
            float t = sm[0][c];
#pragma unroll
            for (int k = 1; k < G; ++k) t += sm[k][c];
            out[col] += t;
        }
        __syncthreads();
    }
}

static void reduce_slices(const float *ws, int nslices, int64_t len, float *out, hipStream_t s) {
    if (nslices >= 32 && len >= 4096)
        hipLaunchKernelGGL((k_reduce_slices_wide<64>), dim3(grid_for(len, 64, 4096)), dim3(kBlock), 0, s, ws, nslices, len, out);
    else if (nslices >= 64)
        hipLaunchKernelGGL((k_reduce_slices_wide<16>), dim3(grid_for(len, 16, 4096)), dim3(kBlock), 0, s, ws, nslices, len, out);
    else
        hipLaunchKernelGGL(k_reduce_slices, dim3(grid_for(len, kBlock, 2048)), dim3(kBlock), 0, s, ws, nslices, len, out);
}

static inline hipStream_t NS(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static uint32_t drop_threshold(float p) {
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 4294967296.0;
    return (t >= 4294967295.0) ? 4294967295u : (uint32_t)t;
}

// precision level 2 applies when every GEMM of the call is made of whole tiles (else the call runs at level 1)
static bool neumf_use_h(const daisy_neumf_ctx *ctx, int64_t R) {
    if (ctx->bf16 != 2 || ctx->model == DAISY_NEUMF_GMF || R % kGemmBM != 0) return false;
    for (int l = 0; l <= ctx->L; ++l)
        if (ctx->width[l] % 64 != 0) return false;
    return true;
}

// small steps of the fp32 mode (the reference's own batch of 256 samples): everything between the gather and the scatter in
// one launch with the layers' weights in LDS (csrc/neumf_mid.hip).  DAISY_NMF_MID=0: the layer-by-layer kernels (A/B, tests).
static bool neumf_use_mid(const daisy_neumf_ctx *ctx, int64_t R, bool train) {
    const char *env = getenv("DAISY_NMF_MID");               // (read per call: the tests switch it)
    const int tune = env ? atoi(env) : 1;
    return tune != 0 && train && ctx->bf16 == 0 && ctx->model == DAISY_NEUMF_FULL && R <= kMidMaxRows && ctx->mid_fits &&
           ctx->mid_aligned;
}
// the first layer through the tables: bf16 storage (the throughput mode), training, no dropout, a first layer of the
// standard halving tower, and fewer distinct table rows than rows in the step.  DAISY_NMF_FACT=0 switches it off (A/B).
static bool neumf_use_fact(const daisy_neumf_ctx *ctx, int64_t R, bool train, uint32_t thresh) {
    const char *env = getenv("DAISY_NMF_FACT");              // (read per call: the tests switch it)
    const int tune = env ? atoi(env) : 1;
    if (neumf_use_mid(ctx, R, train)) return false;
    // bf16 storage (whole tiles: neumf_use_h) or, round 6, the fp32 parity mode (level 1 - bf16 MFMA inputs - keeps the
    // plain path: its first layer rounds x0 and W1, which a product over the tables would not)
    const bool mode_ok = (ctx->bf16 == 2) ? (neumf_use_h(ctx, R) && ctx->dm % 64 == 0) : (ctx->bf16 == 0);
    return tune != 0 && train && thresh == 0 && mode_ok && ctx->model != DAISY_NEUMF_GMF && ctx->L >= 1 &&
           ctx->width[1] == ctx->dm && ctx->U + ctx->I <= R;
}
// layers 2..3, the predict layer, the criterion and their backward pass in one persistent kernel (csrc/neumf_tower.hip):
// the first layer through the tables, the 4d -> 2d -> d tower at d = 64, the full model.  DAISY_NMF_TOWER=0: the
// layer-by-layer kernels (A/B, and the reference the fused kernel is tested against).
static bool neumf_use_tower(const daisy_neumf_ctx *ctx, int64_t R, bool train, uint32_t thresh) {
    const char *env = getenv("DAISY_NMF_TOWER");              // (read per call: the tests switch it)
    const int tune = env ? atoi(env) : 1;
    return tune != 0 && ctx->bf16 == 2 && neumf_use_fact(ctx, R, train, thresh) && ctx->L == 3 && ctx->d == 64 && ctx->model == DAISY_NEUMF_FULL &&
           R % 64 == 0 && ctx->tower_aligned;
}
static int neumf_need_fact(daisy_neumf_ctx *ctx) {
    if (ctx->fact_t) return DAISY_OK;
    // the products in fp32, the row norms (2 floats per row), the products again as bf16 (half a float per element)
    // (+ 32 floats: the bf16 copy starts on a 128-byte boundary - its rows are whole cache lines for the tower's gather)
    const size_t n = (size_t)(ctx->U + ctx->I) * ((size_t)ctx->width[1] + 2 + (size_t)ctx->width[1] / 2) + 32;
    if (hipMalloc((void **)&ctx->fact_t, n * sizeof(float)) != hipSuccess) {
        set_error("neumf: hipMalloc(%zu) of the first-layer table products failed", n * sizeof(float));
        ctx->fact_t = nullptr;
        return DAISY_ERR_HIP;
    }
    return DAISY_OK;
}

// x_L and pred for R pairs starting at src.base (eval: thresh == 0)
static int neumf_forward_rows(daisy_neumf_ctx *ctx, const daisy_neumf_params *p, const PairSrc &src, int64_t R,
                              bool train, int pointwise, uint32_t thresh, float scale, uint64_t seed,
                              double *stats, hipStream_t s) {
    const int d = ctx->d, dm = ctx->dm, L = ctx->L;
    const int grid = grid_for(R, kBlock / 16 * 2);
    const bool H = neumf_use_h(ctx, R);
    if (H) {          // bf16 copies of the MLP weights (a few hundred KB)
        // (the first layer's copy - the largest - has no reader when that layer runs through the tables)
        ctx->tower_aligned = L >= 3 && (((uintptr_t)p->W[1] | (uintptr_t)p->W[2]) & 15) == 0;
        // (... and none at all under the fused tower, which rounds W2 / W3 as it loads them into LDS)
        for (int l = neumf_use_tower(ctx, R, train, thresh) ? L + 1 : (neumf_use_fact(ctx, R, train, thresh) ? 2 : 1); l <= L; ++l) {
            const int64_t nw = (int64_t)ctx->width[l] * ctx->width[l - 1];
            hipLaunchKernelGGL(k_to_bf16, dim3(grid_for(nw, kBlock * 4)), dim3(kBlock), 0, s, p->W[l - 1], nw,
                               ctx->width[l - 1], ctx->W16[l - 1], ctx->W16T[l - 1]);
        }
    }
    if (neumf_use_mid(ctx, R, train)) return DAISY_OK;     // (the gather, the layers and the predict layer happen in k_nmf_mid)
    const bool fact = neumf_use_fact(ctx, R, train, thresh);
    if (fact) {
        int rc = neumf_need_fact(ctx);
        if (rc) return rc;
        const int n1 = ctx->width[1];
        float *tu = ctx->fact_t, *ti = ctx->fact_t + (size_t)ctx->U * n1;
        GemmOp top[2];
        for (int side = 0; side < 2; ++side) {          // T = table x W1[:, half]^T  (fp32: the tables are fp32)
            GemmOp op{};
            op.A = side ? p->iM : p->uM; op.sam = dm; op.sak = 1;
            op.B = p->W[0] + (side ? dm : 0); op.sbn = ctx->width[0]; op.sbk = 1;
            op.C = side ? ti : tu; op.ldc = n1;
            op.M = side ? ctx->I : ctx->U; op.N = n1; op.K = dm;
            op.k_chunk = op.K;
            // fp32 products of the fp32 tables and weights, rounded to bf16 once (k_f32_to_bf16): the rounding points do not
            // depend on whether the table's row count happens to tile (oracle/neumf_numpy.py: neumf_grad_bf16 'fact')
            op.bf16 = 0;
            top[side] = op;
        }
        launch_gemm_pair<EPI_STORE>(top[0], top[1], s);      // (both sides in one launch: k_gemm_pair)
        float2 *nu = reinterpret_cast<float2 *>(ctx->fact_t + (size_t)(ctx->U + ctx->I) * n1), *ni = nu + ctx->U;
        hipLaunchKernelGGL(k_nmf_row_norms, dim3(grid_for(ctx->U > ctx->I ? ctx->U : ctx->I, kBlock / 16), 2), dim3(kBlock), 0, s, p->uM,
                           ctx->U, p->iM, ctx->I, dm, nu, ni);
        uint16_t *t16 = reinterpret_cast<uint16_t *>(ctx->fact_t + (((size_t)(ctx->U + ctx->I) * ((size_t)n1 + 2) + 31) / 32) * 32);
        const int64_t nt = (int64_t)(ctx->U + ctx->I) * n1;
        if (H) hipLaunchKernelGGL(k_f32_to_bf16, dim3(grid_for(nt, kBlock * 2)), dim3(kBlock), 0, s, tu, nt, t16);
        const Fact f{t16, t16 + (size_t)ctx->U * n1, p->b[0], reinterpret_cast<uint16_t *>(ctx->X[1]), n1, nu, ni, tu, ti, ctx->X[1]};
        ctx->fact_cur = f;
        if (!H) {
            hipLaunchKernelGGL((k_nmf_gather<true, false, true>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, pointwise,
                               ctx->X[0], ctx->G, thresh, scale, seed, stats, f);
        } else
        if (neumf_use_tower(ctx, R, train, thresh)) {        // the gather, the layers and the predict layer happen in the tower kernel
            DAISY_LAUNCH_CHECK();
            return DAISY_OK;
        } else
        hipLaunchKernelGGL((k_nmf_gather<true, true, true>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, pointwise,
                           ctx->X[0], ctx->G, thresh, scale, seed, stats, f);
    } else if (train) {
        if (H) hipLaunchKernelGGL((k_nmf_gather<true, true>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, pointwise,
                                  ctx->X[0], ctx->G, thresh, scale, seed, stats);
        else hipLaunchKernelGGL((k_nmf_gather<true, false>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, pointwise,
                                ctx->X[0], ctx->G, thresh, scale, seed, stats);
    } else {
        if (H) hipLaunchKernelGGL((k_nmf_gather<false, true>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, 0, ctx->X[0],
                                  ctx->G, 0u, 1.f, (uint64_t)0, (double *)nullptr);
        else hipLaunchKernelGGL((k_nmf_gather<false, false>), dim3(grid), dim3(kBlock), 0, s, *p, src, R, d, dm, 0, ctx->X[0],
                                ctx->G, 0u, 1.f, (uint64_t)0, (double *)nullptr);
    }
    DAISY_LAUNCH_CHECK();
    if (ctx->model != DAISY_NEUMF_GMF) {
        for (int l = fact ? 2 : 1; l <= L; ++l) {        // (fact: x1 came out of the gather)
            GemmOp op{};
            op.A = ctx->X[l - 1]; op.sam = ctx->width[l - 1]; op.sak = 1;
            op.B = p->W[l - 1]; op.sbn = ctx->width[l - 1]; op.sbk = 1;
            op.C = ctx->X[l]; op.ldc = ctx->width[l];
            op.M = R; op.N = ctx->width[l]; op.K = ctx->width[l - 1];
            op.bias = p->b[l - 1];
            op.k_chunk = op.K;
            op.bf16 = ctx->bf16 ? 1 : 0;
            if (l < L && thresh) {       // the Dropout in front of Linear l+1 acts on this output
                op.drop_thresh = thresh; op.drop_scale = scale; op.drop_seed = seed; op.drop_stream = (uint32_t)(l + 1);
            }
            if (H) {
                op.A16 = reinterpret_cast<const uint16_t *>(ctx->X[l - 1]);
                op.B16 = ctx->W16[l - 1];
                op.C16 = reinterpret_cast<uint16_t *>(ctx->X[l]);
                if (!gemm_h_ok(op)) { set_error("neumf: layer %d does not tile for the bf16-storage GEMM", l); return DAISY_ERR_STATE; }
                launch_gemm_h<EPI_BIAS_RELU>(op, s);
            } else {
                launch_gemm<EPI_BIAS_RELU>(op, s);
            }
            DAISY_LAUNCH_CHECK();
        }
    }
    const int dg = (ctx->model == DAISY_NEUMF_MLP) ? 0 : d;
    const int nl = (ctx->model == DAISY_NEUMF_GMF) ? 0 : ctx->width[L];
    if (H) hipLaunchKernelGGL((k_nmf_predict<true>), dim3(grid), dim3(kBlock), 0, s, ctx->G, dg, ctx->X[L], nl, p->Wp, p->bp, R,
                              ctx->pred);
    else hipLaunchKernelGGL((k_nmf_predict<false>), dim3(grid), dim3(kBlock), 0, s, ctx->G, dg, ctx->X[L], nl, p->Wp, p->bp, R,
                            ctx->pred);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

static int neumf_scatter_scratch(daisy_neumf_ctx *c) {
    if (c->sc_arena) return DAISY_OK;
    const size_t R = (size_t)c->max_rows + 1, dm = (size_t)c->dm;
    const size_t rows_max = (size_t)(c->U > c->I ? c->U : c->I);
    size_t chunks = (size_t)segsum_chunks((int64_t)R + 1, c->dm);
    const size_t ch2 = (size_t)segsum_chunks((int64_t)R + 1, c->d);
    if (ch2 > chunks) chunks = ch2;
    chunks += 2;
    c->sc_tmp_bytes = sort_pairs_i32_temp_bytes((int64_t)R);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_ku = take(R * 4), o_ki = take(R * 4), o_val = take(R * 4), o_ks = take(R * 4), o_vs = take(R * 4);
    const size_t o_cu = take((size_t)c->U * 4), o_ci = take((size_t)c->I * 4), o_cj = take((size_t)c->I * 4);
    const size_t o_ek = take((R + 1) * 4), o_es = take((R + 1) * 8), o_w = take((R + 1) * 8);
    const size_t o_sum = take(rows_max * dm * 4), o_sum2 = take(rows_max * dm * 4), o_sumg = take(rows_max * (size_t)c->d * 4),
                 o_sumg2 = take(rows_max * (size_t)c->d * 4);
    const size_t o_ev = take(2 * chunks * dm * 4), o_ei = take(2 * chunks * 4), o_eb = take(2 * chunks * 4), o_ew = take(chunks * 4);
    const size_t o_tmp = take(c->sc_tmp_bytes);
    hipError_t e = hipMalloc(&c->sc_arena, off);
    if (e != hipSuccess) {
        set_error("neumf: hipMalloc(%zu) of the scatter scratch failed: %s", off, hipGetErrorString(e));
        c->sc_arena = nullptr;
    c->cs_hist = nullptr;
    c->cs_ent = nullptr;
        return DAISY_ERR_HIP;
    }
    char *b = (char *)c->sc_arena;
    c->sc_ku = (int32_t *)(b + o_ku); c->sc_ki = (int32_t *)(b + o_ki); c->sc_val = (int32_t *)(b + o_val);
    c->sc_ks = (int32_t *)(b + o_ks); c->sc_vs = (int32_t *)(b + o_vs);
    c->sc_cu = (int32_t *)(b + o_cu); c->sc_ci = (int32_t *)(b + o_ci); c->sc_cj = (int32_t *)(b + o_cj);
    c->sc_ekey = (uint32_t *)(b + o_ek); c->sc_esu = (uint2 *)(b + o_es); c->sc_w = (float2 *)(b + o_w);
    c->sc_sum = (float *)(b + o_sum); c->sc_sum2 = (float *)(b + o_sum2); c->sc_sumg = (float *)(b + o_sumg);
    c->sc_sumg2 = (float *)(b + o_sumg2);
    c->sc_edge_vec = (float *)(b + o_ev); c->sc_edge_item = (int32_t *)(b + o_ei); c->sc_edge_b = (float *)(b + o_eb);
    c->sc_edge_whole = (int32_t *)(b + o_ew);
    c->sc_tmp = b + o_tmp;
    // the counts and the row-sum table are kept all-zero between calls by the kernels that consume them
    e = hipMemset(b + o_cu, 0, o_ek - o_cu);
    if (e == hipSuccess) e = hipMemset(b + o_sum, 0, (o_sumg2 - o_sum) + rows_max * (size_t)c->d * 4);      // (the four sum tables: contiguous)
    if (e != hipSuccess) { set_error("neumf: hipMemset of the scatter scratch failed"); return DAISY_ERR_HIP; }
    return DAISY_OK;
}

// g.{uG,iG,uM,iM} += the embedding gradients of the step (DX0: fp32 [R, 2*dm] input gradient of the MLP tower)
// fact (the first layer through the tables): DX0 is dZ_1 (bf16 [R, n1]) instead; S = its segmented sums by table row
// (fp32 [rows, n1], in the row-sum table), then  g.table += S W1[:, half]  and  gW_1[:, half] += S^T table  - two GEMMs over
// the TABLE's rows where the plain path runs one over the step's R rows and a [R, 2 dm] input gradient
static CsEntries cs_entries(const daisy_neumf_ctx *c, int side) {
    const size_t per = align_up(((size_t)c->max_rows + 2) * 8);
    char *b = (char *)c->cs_ent + (size_t)side * 5 * per;
    return CsEntries{(uint32_t *)b, (uint2 *)(b + per), (float2 *)(b + 2 * per), (uint2 *)(b + 3 * per), (float2 *)(b + 4 * per)};
}

static int neumf_scatter_owner(daisy_neumf_ctx *c, const daisy_neumf_params &p, const daisy_neumf_params &g,
                               const PairSrc &src, int64_t R, int pointwise, const float *DX0, bool dx0_bf16,
                               const double *stats, float reg_1, float reg_2, hipStream_t s, bool fact = false) {
    {
        // small steps: the whole scatter in one launch (k_nmf_scatter_scan).  DAISY_NMF_SCATTER_SMALL (read per call): 0 - off,
        // 2 - the sorting kernel it replaced (k_nmf_scatter_small: A/B, and the tests' bit-for-bit cross-check)
        const char *env_sm = getenv("DAISY_NMF_SCATTER_SMALL");
        const int sm_mode = env_sm ? atoi(env_sm) : 1;
        // (the scanning kernel holds a table row's dm + d columns as 16 x 5 float4 at most; 16-byte aligned tables and gradients)
        const bool scan_ok = (c->dm + c->d) / 4 <= 80 &&
                             ((((uintptr_t)p.uM | (uintptr_t)p.iM | (uintptr_t)p.uG | (uintptr_t)p.iG | (uintptr_t)g.uM | (uintptr_t)g.iM |
                                (uintptr_t)g.uG | (uintptr_t)g.iG | (uintptr_t)DX0 | (uintptr_t)p.Wp) & 15) == 0);
        const bool sort_ok = R <= kScatterSmallRows && c->U < (1 << 22) && c->I < (1 << 22);
        if (R <= kScanMaxRows && !dx0_bf16 && !fact && sm_mode != 0 && (scan_ok || sort_ok)) {
            if ((sm_mode == 2 || !scan_ok) && sort_ok)
                hipLaunchKernelGGL(k_nmf_scatter_small, dim3(2, 16), dim3(kScatterSmallRows), 0, s, p, g, src, (int)R, c->d, c->dm, c->model,
                                   pointwise, c->dpred, DX0, stats, reg_1, reg_2);
            else if (scan_ok) {
                const int blk = scan_block(R), gpb = blk / 16;
                const dim3 grid(2, (unsigned)((R + gpb - 1) / gpb));
                const int nt = ((c->dm + c->d) / 4 + 15) / 16;          // float4 chunks of a table row's columns per lane
                const int rounds = (int)((R + 63) / 64);
                const size_t lds = (size_t)gpb * rounds * 10 + (size_t)rounds * 64 * 12;
                static bool attr_set = false;
                if (!attr_set) {
                    const int cap = (scan_block(kScanMaxRows) / 16) * (kScanMaxRows / 64) * 10 + kScanMaxRows * 12;
                    DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_nmf_scatter_scan<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                    DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_nmf_scatter_scan<5>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                    attr_set = true;
                }
                if (nt <= 2) hipLaunchKernelGGL((k_nmf_scatter_scan<2>), grid, dim3(blk), lds, s, p, g, src, (int)R, c->d, c->dm,
                                                c->model, pointwise, c->dpred, DX0, stats, reg_1, reg_2);
                else hipLaunchKernelGGL((k_nmf_scatter_scan<5>), grid, dim3(blk), lds, s, p, g, src, (int)R, c->d, c->dm,
                                        c->model, pointwise, c->dpred, DX0, stats, reg_1, reg_2);
            } else {
                set_error("neumf: no small-step scatter for this step (rows %lld)", (long long)R);
                return DAISY_ERR_STATE;
            }
            DAISY_LAUNCH_CHECK();
            return DAISY_OK;
        }
    }
    int rc = neumf_scatter_scratch(c);
    if (rc) return rc;
    const int d = c->d, dm = c->dm, model = c->model;
    const int64_t n_pad = R + (R & 1);
    // rows grouped by table row: the counting pass (tables of at most kCsMaxKeys rows - a histogram per wave fits the LDS),
    // else two radix sorts.  DAISY_NMF_COUNTING=0 (read per call): always the sorts (A/B, and the tests' cross-check)
    const char *env_cs = getenv("DAISY_NMF_COUNTING");
    const int Kmax = (int)(c->U > c->I ? c->U : c->I);
    const bool counting = (!env_cs || atoi(env_cs) != 0) && Kmax <= kCsMaxKeys && Kmax < (1 << 14) && R >= 4096;
    const int kstride = (Kmax + 63) / 64 * 64;
    if (counting) {
        if (!c->cs_hist) {
            const size_t n = (size_t)2 * (kCsNW + 1) * (size_t)((kCsMaxKeys + 63) / 64 * 64);
            if (hipMalloc((void **)&c->cs_hist, n * sizeof(int32_t)) != hipSuccess) {
                c->cs_hist = nullptr;
                set_error("neumf: hipMalloc(%zu) of the counting pass's histograms failed", n * sizeof(int32_t));
                return DAISY_ERR_HIP;
            }
            const size_t per = align_up(((size_t)c->max_rows + 2) * 8);       // one array of (rows + pad) x 8 bytes
            if (hipMalloc(&c->cs_ent, 2 * 5 * per) != hipSuccess) {
                c->cs_ent = nullptr;
                set_error("neumf: hipMalloc(%zu) of the counting pass's entry lists failed", 2 * 5 * per);
                return DAISY_ERR_HIP;
            }
            DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cs_count), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
            DAISY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cs_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        }
        const int halves = pointwise ? 1 : 2;
        int32_t *total = c->cs_hist + (size_t)2 * kCsNW * kstride;
        const size_t lds = (size_t)kCsWaves * kstride * sizeof(int32_t);
        hipLaunchKernelGGL(k_cs_count, dim3(kCsBlocks, 2), dim3(kBlock), lds, s, src, R, halves, (int)c->U, (int)c->I, kstride, c->cs_hist);
        hipLaunchKernelGGL(k_cs_prefix, dim3((Kmax + kBlock - 1) / kBlock, 2), dim3(kBlock), 0, s, halves, (int)c->U, (int)c->I, kstride,
                           c->cs_hist, total, c->sc_cu, c->sc_ci, c->sc_cj);
        hipLaunchKernelGGL(k_cs_base, dim3(2), dim3(1024), 0, s, (int)c->U, (int)c->I, kstride, total);
        hipLaunchKernelGGL(k_cs_scatter, dim3(kCsBlocks, 2), dim3(kBlock), lds, s, src, R, halves, (int)c->U, (int)c->I, kstride,
                           c->cs_hist, total, cs_entries(c, 0), cs_entries(c, 1), fact ? 1 : 2, fact ? 0 : 1,
                           (const float *)c->dpred);
        hipLaunchKernelGGL(k_cs_entries, dim3(grid_for(R, kBlock, 2048), 2), dim3(kBlock), 0, s, src, R, cs_entries(c, 0), cs_entries(c, 1),
                           fact ? 1 : 2, fact ? 0 : 1, (const float *)c->dpred);
    } else {
        hipLaunchKernelGGL(k_nmf_sort_keys, dim3(grid_for(R, kBlock * 2)), dim3(kBlock), 0, s, src, R, c->sc_ku, c->sc_ki,
                           c->sc_val, pointwise, c->sc_cu, c->sc_ci, c->sc_cj);
    }
    DAISY_LAUNCH_CHECK();
    // both sides' commits in one launch (k_nmf_table_commit_pair) when every operand takes float4 accesses; each side then has row-sum
    // tables of its own (a shared one had to be committed before the other side's reduction refilled it)
    auto al16 = [](const void *q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const bool pair_commit = dm % 4 == 0 && d % 4 == 0 && al16(g.uM) && al16(g.iM) && al16(g.uG) && al16(g.iG) && al16(p.uM) && al16(p.iM) &&
                             al16(p.uG) && al16(p.iG) && al16(p.Wp);
    CommitSide cside[2];
    for (int side = 0; side < 2; ++side) {            // 0: the user tables, 1: the item tables
        const int64_t rows = side ? c->I : c->U;
        float *sumM = side ? c->sc_sum2 : c->sc_sum, *sumG = side ? c->sc_sumg2 : c->sc_sumg;
        // the side's grouped rows: (keys, row ids) in table-row order, rows ascending inside a key
        const int32_t *g_ks = c->sc_ks, *g_vs = c->sc_vs;
        const CsEntries en = counting ? cs_entries(c, side) : CsEntries{c->sc_ekey, c->sc_esu, c->sc_w, c->sc_esu, c->sc_w};
        if (!counting) {
            rc = sort_pairs_i32(c->sc_tmp, c->sc_tmp_bytes, side ? c->sc_ki : c->sc_ku, c->sc_ks, c->sc_val, c->sc_vs, R,
                                bits_for(rows), s);
            if (rc) return rc;
        }
        const int ge = grid_for(n_pad, kBlock * 2);
        // MLP table: source row = half `side` of DX0[r]
        if (model != DAISY_NEUMF_GMF) {
            if (!counting)
                hipLaunchKernelGGL(k_nmf_entries, dim3(ge), dim3(kBlock), 0, s, g_ks, g_vs, R, n_pad, fact ? 1 : 2,
                                   fact ? 0 : side, c->sc_ekey, c->sc_esu, c->sc_w);
            rc = segsum_rows(DX0, en.w_m, en.ekey, en.esu_m, n_pad, dm, sumM, c->sc_edge_vec,
                             c->sc_edge_item, c->sc_edge_b, c->sc_edge_whole, s, dx0_bf16);
            if (rc) return rc;
        }
        float *sumM_commit = (model != DAISY_NEUMF_GMF && !fact) ? sumM : (float *)nullptr;     // (fact: S_u / S_i feed the table GEMMs below)
        if (!pair_commit)
            launch_table_commit(side ? g.iM : g.uM, sumM_commit, side ? p.iM : p.uM, rows, dm, side ? c->sc_ci : c->sc_cu, side ? 3 : 1,
                                (int32_t *)nullptr, 0, 0.f, stats, reg_1, reg_2, 0, nullptr, s);
        // GMF table: source row = the materialised per-row gradient
        if (model != DAISY_NEUMF_MLP) {          // source rows: the OTHER table's, weights dpred (k_nmf_entries_gmf)
            if (!counting)
                hipLaunchKernelGGL(k_nmf_entries_gmf, dim3(ge), dim3(kBlock), 0, s, g_ks, g_vs, R, n_pad, src, side, c->dpred,
                                   c->sc_ekey, c->sc_esu, c->sc_w);
            rc = segsum_rows(side ? p.uG : p.iG, en.w_g, en.ekey, en.esu_g, n_pad, d, sumG, c->sc_edge_vec,
                             c->sc_edge_item, c->sc_edge_b, c->sc_edge_whole, s);
            if (rc) return rc;
        }
        // (the negative item's GMF rows enter the regulariser twice, NeuMFRecommender.py:158-161)
        float *sumG_commit = (model != DAISY_NEUMF_MLP) ? sumG : (float *)nullptr;
        const float *colscale = (model != DAISY_NEUMF_MLP) ? p.Wp : (const float *)nullptr;
        if (!pair_commit)
            launch_table_commit(side ? g.iG : g.uG, sumG_commit, side ? p.iG : p.uG, rows, d, side ? c->sc_ci : c->sc_cu, side ? 2 : 0,
                                side ? c->sc_cj : (int32_t *)nullptr, 4, 2.f, stats, reg_1, reg_2, 1, colscale, s);
        cside[side] = CommitSide{side ? g.iM : g.uM, sumM_commit, side ? p.iM : p.uM, dm, side ? 3 : 1,
                                 side ? g.iG : g.uG, sumG_commit, side ? p.iG : p.uG, d, side ? 2 : 0,
                                 rows, side ? c->sc_ci : c->sc_cu, side ? c->sc_cj : (int32_t *)nullptr, colscale};
        DAISY_LAUNCH_CHECK();
    }
    if (pair_commit) {
        const int64_t rmax = c->U > c->I ? c->U : c->I;
        hipLaunchKernelGGL(k_nmf_table_commit_pair, dim3(grid_for(rmax, kBlock / 16), 2), dim3(kBlock), 0, s, cside[0], cside[1], stats,
                           reg_1, reg_2);
        DAISY_LAUNCH_CHECK();
    }
    if (fact && model != DAISY_NEUMF_GMF) {
        // the first layer through the tables, backward: S_u / S_i (the segment sums of dZ_1 by user / by item) are both in place;
        // every product runs for both sides in ONE launch (k_gemm_pair: each side alone is a latency-bound ~100 workgroups)
        const int n1 = dm, w0 = 2 * dm;
        rc = neumf_need_det_ws(c);
        if (rc) return rc;
        {   // gb_1 = sum over the step's rows of dZ_1 = sum over the USERS of their segment sums: a column sum over U table rows
            // (6 MB at ml-1m) instead of one over the R rows of dZ_1 (268 MB: 47 us)
            const int cr = colsum_rows(c->U);
            const dim3 cs((unsigned)((n1 + 63) / 64), (unsigned)((c->U + cr - 1) / cr));
            hipLaunchKernelGGL((k_colsum<false>), cs, dim3(kBlock), 0, s, c->sc_sum, c->U, n1, (int64_t)n1, c->det_ws, cr);
            reduce_slices(c->det_ws, (int)cs.y, n1, g.b[0], s);
        }
        GemmOp a[2], b[2];
        // slices of 128 table rows while the workspace holds both sides' slices; tables with more rows than that (the
        // workspace is sized by the step's rows) take proportionally longer slices - never an error mid-step
        const int64_t cap = (int64_t)(c->det_ws_floats / ((size_t)n1 * (size_t)dm));      // >= 2
        int64_t kc = 128;
        while ((c->U + kc - 1) / kc + (c->I + kc - 1) / kc > cap) kc += 128;
        int bsplits[2];
        for (int side = 0; side < 2; ++side) {
            const int64_t rows = side ? c->I : c->U;
            float *S = side ? c->sc_sum2 : c->sc_sum;
            a[side] = GemmOp{};                // g.table[rows, dm] += S[rows, n1] W1[:, half]      (k = n1)
            a[side].A = S; a[side].sam = n1; a[side].sak = 1;
            a[side].B = p.W[0] + (side ? dm : 0); a[side].sbn = 1; a[side].sbk = w0;
            a[side].C = side ? g.iM : g.uM; a[side].ldc = dm;
            a[side].M = rows; a[side].N = dm; a[side].K = n1; a[side].k_chunk = n1;
            b[side] = GemmOp{};                // gW_1[n1, half] += S^T[n1, rows] table[rows, dm]    (k = the table's rows, in slices)
            b[side].A = S; b[side].sam = 1; b[side].sak = n1;
            b[side].B = side ? p.iM : p.uM; b[side].sbn = 1; b[side].sbk = dm;
            b[side].M = n1; b[side].N = dm; b[side].K = rows; b[side].k_chunk = kc;
            bsplits[side] = (int)((rows + kc - 1) / kc);
            b[side].C = c->det_ws + (side ? (size_t)bsplits[0] * n1 * dm : 0); b[side].ldc = dm; b[side].slice_stride = (int64_t)n1 * dm;
        }
        launch_gemm_pair<EPI_ATOMIC>(a[0], a[1], s);       // (one workgroup per output tile, k in one piece: a single add per element)
        launch_gemm_pair<EPI_ATOMIC>(b[0], b[1], s);
        hipLaunchKernelGGL(k_reduce_slices_2d, dim3(grid_for((int64_t)n1 * dm, kBlock, 2048), 2), dim3(kBlock), 0, s, b[0].C, bsplits[0], n1,
                           dm, g.W[0], (int64_t)w0, (const float *)b[1].C, bsplits[1], g.W[0] + dm);
        // (both sum tables back to all-zero: they are contiguous)
        DAISY_HIP(hipMemsetAsync(c->sc_sum, 0, (size_t)((char *)c->sc_sumg - (char *)c->sc_sum), s));
        DAISY_LAUNCH_CHECK();
    }
    return DAISY_OK;
}

extern "C" {

int daisy_neumf_ctx_create(daisy_neumf_ctx **out, int64_t max_rows, int32_t factors, int32_t num_layers,
                           int32_t model, int64_t user_num, int64_t item_num) {
    DAISY_CHECK_ARG(out != nullptr, "neumf_ctx_create: out is NULL");
    *out = nullptr;
    DAISY_CHECK_ARG(max_rows > 0 && user_num > 0 && item_num > 0, "neumf_ctx_create: bad sizes");
    DAISY_CHECK_ARG(factors > 0 && factors % 4 == 0 && factors <= 256,
                    "neumf_ctx_create: factors=%d must be a multiple of 4 in 4..256", factors);
    DAISY_CHECK_ARG(num_layers >= 1 && num_layers <= DAISY_NEUMF_MAX_LAYERS, "neumf_ctx_create: num_layers=%d",
                    num_layers);
    DAISY_CHECK_ARG(model >= DAISY_NEUMF_FULL && model <= DAISY_NEUMF_MLP, "neumf_ctx_create: model=%d", model);
    daisy_neumf_ctx *c = new daisy_neumf_ctx();
    c->max_rows = max_rows; c->U = user_num; c->I = item_num;
    c->d = factors; c->L = num_layers; c->model = model;
    c->bf16 = 0;
    c->sc_arena = nullptr;
    c->det_ws = nullptr; c->det_ws_floats = 0;
    c->dm = factors << (num_layers - 1);
    c->width[0] = 2 * c->dm;
    for (int l = 1; l <= num_layers; ++l) c->width[l] = c->width[l - 1] / 2;
    c->mid_fits = neumf_mid_fits(c->L, c->width, c->d);
    c->mid_aligned = false;
    size_t off = 0, ox[DAISY_NEUMF_MAX_LAYERS + 1];
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    for (int l = 0; l <= num_layers; ++l) ox[l] = take((size_t)max_rows * c->width[l] * 4);
    const size_t og = take((size_t)max_rows * factors * 4), op = take((size_t)max_rows * 4),
                 od = take((size_t)max_rows * 4);
    const size_t oz0 = take((size_t)max_rows * c->width[0] * 4), oz1 = take((size_t)max_rows * c->width[0] * 4);
    size_t ow[DAISY_NEUMF_MAX_LAYERS];
    for (int l = 1; l <= num_layers; ++l) ow[l - 1] = take((size_t)c->width[l] * c->width[l - 1] * 2 * 2);
    c->arena_bytes = off;
    hipError_t e = hipMalloc(&c->arena, off);
    if (e != hipSuccess) {
        set_error("neumf_ctx_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
        delete c;
        return DAISY_ERR_HIP;
    }
    char *base = (char *)c->arena;
    for (int l = 0; l <= num_layers; ++l) c->X[l] = (float *)(base + ox[l]);
    c->G = (float *)(base + og); c->pred = (float *)(base + op); c->dpred = (float *)(base + od);
    c->DZ[0] = (float *)(base + oz0); c->DZ[1] = (float *)(base + oz1);
    for (int l = 1; l <= num_layers; ++l) {
        c->W16[l - 1] = (uint16_t *)(base + ow[l - 1]);
        c->W16T[l - 1] = c->W16[l - 1] + (size_t)c->width[l] * c->width[l - 1];
    }
    *out = c;
    return DAISY_OK;
}

int daisy_neumf_ctx_destroy(daisy_neumf_ctx *ctx) {
    if (!ctx) return DAISY_OK;
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->sc_arena) (void)hipFree(ctx->sc_arena);
    if (ctx->cs_hist) (void)hipFree(ctx->cs_hist);
    if (ctx->cs_ent) (void)hipFree(ctx->cs_ent);
    if (ctx->det_ws) (void)hipFree(ctx->det_ws);
    if (ctx->fact_t) (void)hipFree(ctx->fact_t);
    delete ctx;
    return DAISY_OK;
}

size_t daisy_neumf_ctx_bytes(const daisy_neumf_ctx *ctx) { return ctx ? ctx->arena_bytes : 0; }

int daisy_neumf_ctx_set_precision(daisy_neumf_ctx *ctx, int32_t bf16_gemm) {
    DAISY_CHECK_ARG(ctx != nullptr, "neumf_ctx_set_precision: NULL context");
    DAISY_CHECK_ARG(bf16_gemm >= 0 && bf16_gemm <= 2, "neumf_ctx_set_precision: level %d not in 0..2", bf16_gemm);
    ctx->bf16 = bf16_gemm;
    return DAISY_OK;
}

int daisy_neumf_scores(daisy_neumf_ctx *ctx, const daisy_neumf_params *params, const int64_t *users,
                       const int64_t *items, int64_t n, int64_t C, float *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && params && users && out && n > 0 && C >= 0, "neumf_scores: bad argument");
    DAISY_CHECK_ARG(items || C == 0, "neumf_scores: items is NULL but C != 0");
    hipStream_t s = NS(stream);
    for (int64_t base = 0; base < n; base += ctx->max_rows) {
        const int64_t R = (n - base < ctx->max_rows) ? (n - base) : ctx->max_rows;
        PairSrc src{};
        src.users = users; src.items = items; src.C = C; src.base = base;
        int rc = neumf_forward_rows(ctx, params, src, R, false, 0, 0u, 1.f, 0, nullptr, s);
        if (rc) return rc;
        DAISY_HIP(hipMemcpyAsync(out + base, ctx->pred, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
    }
    return DAISY_OK;
}

int daisy_neumf_step_grads(daisy_neumf_ctx *ctx, const daisy_neumf_params *params,
                           const daisy_neumf_params *grads, const int32_t *u, const int32_t *i,
                           const int32_t *j, int64_t B, int32_t loss_type, float gamma, float reg_1,
                           float reg_2, float dropout_p, uint64_t seed, double *stats,
                           daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && params && grads && u && i && j && stats && B > 0, "neumf_step_grads: bad argument");
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_SL, "Invalid loss type: %d", loss_type);
    DAISY_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "neumf_step_grads: dropout_p=%g not in [0,1)", dropout_p);
    const int pointwise = loss_type >= DAISY_LOSS_CL;
    const int64_t R = pointwise ? B : 2 * B;
    DAISY_CHECK_ARG(R <= ctx->max_rows, "neumf_step_grads: %lld rows exceed the context's %lld",
                    (long long)R, (long long)ctx->max_rows);
    hipStream_t s = NS(stream);
    const daisy_neumf_params &p = *params, &g = *grads;
    const int d = ctx->d, dm = ctx->dm, L = ctx->L, model = ctx->model;
    const uint32_t thresh = (model == DAISY_NEUMF_GMF) ? 0u : drop_threshold(dropout_p);
    const float scale = thresh ? 1.f / (1.f - dropout_p) : 1.f;
    {
        uintptr_t bits = (uintptr_t)p.Wp | (uintptr_t)p.uG | (uintptr_t)p.iG | (uintptr_t)p.uM | (uintptr_t)p.iM;
        for (int l = 0; l < L; ++l) bits |= (uintptr_t)p.W[l] | (uintptr_t)p.b[l];
        ctx->mid_aligned = (bits & 15) == 0;           // (k_nmf_mid reads the parameters and the tables' rows as float4)
        // (the tower's condition as well, before the first decision that depends on it - the forward pass sets it again)
        ctx->tower_aligned = L >= 3 && (((uintptr_t)p.W[1] | (uintptr_t)p.W[2]) & 15) == 0;
    }
    // (slots 0..16: DAISY_NST_LOSS_SUM runs over steps; the small-step path writes every slot itself - k_nmf_mid_reduce)
    // (... and so does the fused tower: k_nmf_tower_reduce)
    if (!neumf_use_mid(ctx, R, true) && !neumf_use_tower(ctx, R, true, thresh))
        DAISY_HIP(hipMemsetAsync(stats, 0, DAISY_NST_LOSS_SUM * sizeof(double), s));
    PairSrc src{};
    src.u = u; src.i = i; src.j = pointwise ? i : j; src.B = B;
    int rc = neumf_forward_rows(ctx, params, src, R, true, pointwise, thresh, scale, seed, stats, s);
    if (rc) return rc;
    if ((rc = neumf_need_det_ws(ctx))) return rc;
    float *ws = ctx->det_ws;
    const bool tower = neumf_use_tower(ctx, R, true, thresh);        // (the forward pass took the same decision)
    const int dg = (model == DAISY_NEUMF_MLP) ? 0 : d;
    const int nl = (model == DAISY_NEUMF_GMF) ? 0 : ctx->width[L];
    float *dz = ctx->DZ[0], *dz_next = ctx->DZ[1];
    const bool H = neumf_use_h(ctx, R);
    // 1 (default): owner-based, reproducible embedding scatter; 0: the fp32-atomics kernel (kept for A/B measurements)
    static const int tune_scatter = getenv("DAISY_NMF_SCATTER_OWNER") ? atoi(getenv("DAISY_NMF_SCATTER_OWNER")) : 1;
    const bool owner_scatter = tune_scatter != 0;
    const bool mid = neumf_use_mid(ctx, R, true);                    // (the forward pass took the same decision)
    if (mid) {
        MidArgs ma{};
        ma.uG = p.uG; ma.iG = p.iG; ma.uM = p.uM; ma.iM = p.iM; ma.u = u; ma.i = i; ma.dm = dm;
        ma.DX0 = dz; ma.pred = ctx->pred; ma.dpred = ctx->dpred;
        for (int l = 0; l < L; ++l) { ma.W[l] = p.W[l]; ma.b[l] = p.b[l]; }
        ma.Wp = p.Wp; ma.bp = p.bp;
        for (int l = 0; l <= L; ++l) ma.width[l] = ctx->width[l];
        ma.L = L; ma.d = d;
        ma.j = j; ma.B = (int)B; ma.R = (int)R; ma.pointwise = pointwise; ma.loss_type = (int)loss_type; ma.gamma = gamma;
        ma.thresh = thresh; ma.scale = scale; ma.seed = seed;
        ma.ws = ws;
        rc = neumf_mid_step(ma, g.W, g.b, g.Wp, g.bp, stats, reg_1, reg_2, s);
        if (rc) return rc;
    } else
    if (tower) {
        // x1 gathered from the table products, layers 2..3, predict, criterion, dZ3 .. dZ1, gW3, gW2, gb3, gb2, gWp, gbp and
        // the step's statistics: one persistent kernel + the fixed-order sum of its workgroups' slabs
        TowerArgs ta{};
        ta.tu = ctx->fact_cur.tu; ta.ti = ctx->fact_cur.ti; ta.nu = ctx->fact_cur.nu; ta.ni = ctx->fact_cur.ni;
        ta.b1 = p.b[0];
        ta.W2 = p.W[1]; ta.W3 = p.W[2];
        ta.b2 = p.b[1]; ta.b3 = p.b[2]; ta.Wp = p.Wp; ta.bp = p.bp;
        ta.uG = p.uG; ta.iG = p.iG;
        ta.u = u; ta.i = i; ta.j = j; ta.B = B;
        ta.pointwise = pointwise; ta.loss_type = (int)loss_type; ta.gamma = gamma;
        ta.dZ1 = reinterpret_cast<uint16_t *>(dz); ta.dpred = ctx->dpred; ta.ws = ws;
        rc = neumf_tower_step(ta, d, R, g.W[1], g.W[2], g.b[1], g.b[2], g.Wp, g.bp, stats, reg_1, reg_2, s);
        if (rc) return rc;
    } else {
    const int loss_grid = grid_for(B, kBlock * 2);
    hipLaunchKernelGGL(k_nmf_loss, dim3(loss_grid), dim3(kBlock), 0, s, ctx->pred, j, B,
                       (int)loss_type, gamma, pointwise, ctx->dpred, stats, ws);
    reduce_slices(ws, loss_grid, 1, g.bp, s);
    hipLaunchKernelGGL(k_nmf_finalize, dim3(1), dim3(64), 0, s, stats, reg_1, reg_2, pointwise);
    DAISY_LAUNCH_CHECK();
    // ---- backward
    const bool vec_pred = dg == nl && dg > 0 && dg <= 64 && dg % 4 == 0;          // NeuMF proper (not the GMF / MLP ablations)
    const int pb_grid = vec_pred ? grid_for(R, kBlock / 16 * 8, 512) : grid_for(R, kBlock / 16 * 16, 1024);
    if (vec_pred && H) hipLaunchKernelGGL((k_nmf_pred_bwd_v<true>), dim3(pb_grid), dim3(kBlock), 0, s, ctx->dpred, ctx->G, dg,
                                          ctx->X[L], p.Wp, R, dz, ws);
    else if (vec_pred) hipLaunchKernelGGL((k_nmf_pred_bwd_v<false>), dim3(pb_grid), dim3(kBlock), 0, s, ctx->dpred, ctx->G, dg,
                                          ctx->X[L], p.Wp, R, dz, ws);
    else if (H) hipLaunchKernelGGL((k_nmf_pred_bwd<true>), dim3(pb_grid), dim3(kBlock), 0, s, ctx->dpred,
                              ctx->G, dg, ctx->X[L], nl, p.Wp, R, dz, ws);
    else hipLaunchKernelGGL((k_nmf_pred_bwd<false>), dim3(pb_grid), dim3(kBlock), 0, s, ctx->dpred,
                            ctx->G, dg, ctx->X[L], nl, p.Wp, R, dz, ws);
    reduce_slices(ws, pb_grid, dg + nl, g.Wp, s);       // gWp += the workgroups' column sums, in workgroup order
    DAISY_LAUNCH_CHECK();
    }
    const bool fact = neumf_use_fact(ctx, R, true, thresh);          // (the forward pass took the same decision)
    if (model != DAISY_NEUMF_GMF && !mid) {                 // (mid: dz already holds dX0)
        for (int l = tower ? 1 : L; l >= 1; --l) {          // (tower: dz already holds dZ_1)
            const int n_out = ctx->width[l], n_in = ctx->width[l - 1];
            if (fact && l == 1) {
                // the first layer through the tables: gb_1, gW_1 and the MLP tables' gradients come out of the
                // scatter (segmented sums of dZ_1 by user and by item, then two small GEMMs each) - no [R, 2 dm] input
                // gradient, no weight-gradient GEMM over the R rows
                // (gb_1: the column sums of the users' segment sums, neumf_scatter_owner)
                DAISY_LAUNCH_CHECK();
                break;
            }
            GemmOp w{};                                   // gW_l[n_out, n_in] += dZ^T x_{l-1}
            if (n_out % kGemmBM == 0) {
                w.A = dz; w.sam = 1; w.sak = n_out;
                w.B = ctx->X[l - 1]; w.sbn = 1; w.sbk = n_in;
                w.C = g.W[l - 1]; w.ldc = n_in;
                w.M = n_out; w.N = n_in;
            } else {                                      // narrow layer: tile the wider side over M, store transposed
                w.A = ctx->X[l - 1]; w.sam = 1; w.sak = n_in;
                w.B = dz; w.sbn = 1; w.sbk = n_out;
                w.C = g.W[l - 1]; w.ldc = 1; w.scn = n_in;
                w.M = n_in; w.N = n_out;
            }
            w.K = R;
            w.k_chunk = wgrad_chunk();
            {
                // few rows (the reference's own batch: 256 samples = 512 rows) and a small layer: ONE workgroup would walk all
                // rows of the step with guarded loads - 50 us per weight gradient at factors 24, a third of that step.  Slices
                // of at least 64 rows, as many as give the chip ~256 workgroups and as the reduction workspace holds.
                const int64_t tiles = ((w.M + kGemmBM - 1) / kGemmBM) * ((w.N + ((w.N > 64) ? 128 : 64) - 1) / ((w.N > 64) ? 128 : 64));
                int64_t kc = (R * tiles / 256 + 63) / 64 * 64;
                if (kc < 64) kc = 64;
                const int64_t cap = (int64_t)(ctx->det_ws_floats / (size_t)((int64_t)n_out * n_in));
                if (cap > 0 && (R + kc - 1) / kc > cap) kc = ((R + cap - 1) / cap + 63) / 64 * 64;
                if (kc < w.k_chunk) w.k_chunk = kc;
            }
            w.bf16 = ctx->bf16 ? 1 : 0;
            // split-K slices land side by side in the workspace and are added in slice order (no fp32 atomics)
            const int64_t wlen = (int64_t)n_out * n_in;
            const int wsplits = (int)((w.k_chunk < R) ? (R + w.k_chunk - 1) / w.k_chunk : 1);
            float *wdst = w.C;
            w.C = ws;
            w.slice_stride = wlen;
            const int cr = colsum_rows(R);
            const dim3 cs_grid((unsigned)((n_out + 63) / 64), (unsigned)((R + cr - 1) / cr));
            if (H) {
                w.A16 = reinterpret_cast<const uint16_t *>(w.A);
                w.B16 = reinterpret_cast<const uint16_t *>(w.B);
                if (!gemm_h_ok(w)) { set_error("neumf: weight gradient of layer %d does not tile for the bf16-storage GEMM", l); return DAISY_ERR_STATE; }
                launch_gemm_h<EPI_ATOMIC>(w, s);
                reduce_slices(ws, wsplits, wlen, wdst, s);
                if (n_out % 8 == 0 && kBlock % (n_out / 8) == 0) {
                    const int tiles = (int)((R + kColsumRowsH - 1) / kColsumRowsH);
                    hipLaunchKernelGGL(k_colsum_h, dim3((unsigned)tiles), dim3(kBlock), 0, s,
                                       reinterpret_cast<const uint16_t *>(dz), R, n_out, ws);
                    reduce_slices(ws, tiles, n_out, g.b[l - 1], s);
                } else {
                    hipLaunchKernelGGL((k_colsum<true>), cs_grid, dim3(kBlock), 0, s, dz, R, n_out, (int64_t)n_out, ws, cr);
                    reduce_slices(ws, (int)cs_grid.y, n_out, g.b[l - 1], s);
                }
            } else {
                launch_gemm<EPI_ATOMIC>(w, s);
                reduce_slices(ws, wsplits, wlen, wdst, s);
                hipLaunchKernelGGL((k_colsum<false>), cs_grid, dim3(kBlock), 0, s, dz, R, n_out, (int64_t)n_out, ws, cr);
                reduce_slices(ws, (int)cs_grid.y, n_out, g.b[l - 1], s);
            }
            GemmOp x{};                                   // dZ_{l-1}[R, n_in] = (dZ W_l) gated
            x.A = dz; x.sam = n_out; x.sak = 1;
            x.B = p.W[l - 1]; x.sbn = 1; x.sbk = n_in;
            x.C = dz_next; x.ldc = n_in;
            x.M = R; x.N = n_in; x.K = n_out;
            x.k_chunk = x.K;
            x.bf16 = ctx->bf16 ? 1 : 0;
            if (l > 1) {                                  // ReLU (and dropout) gate of x_{l-1}
                x.gate = ctx->X[l - 1]; x.ldg = n_in; x.gate_scale = scale;
            } else if (thresh) {                          // dropout mask of the concat input
                x.drop_thresh = thresh; x.drop_scale = scale; x.drop_seed = seed; x.drop_stream = 1u;
            }
            if (H) {
                x.A16 = reinterpret_cast<const uint16_t *>(dz);
                x.B16 = ctx->W16T[l - 1]; x.sbn = n_out; x.sbk = 1;        // W^T [n_in][n_out]: both operands along k
                x.C16 = reinterpret_cast<uint16_t *>(dz_next);        // bf16 like every other stored gradient of this level
                x.G16 = (l > 1) ? reinterpret_cast<const uint16_t *>(ctx->X[l - 1]) : nullptr;
                if (!gemm_h_ok(x)) { set_error("neumf: input gradient of layer %d does not tile for the bf16-storage GEMM", l); return DAISY_ERR_STATE; }
                launch_gemm_h<EPI_GATE>(x, s);
            } else {
                launch_gemm<EPI_GATE>(x, s);
            }
            DAISY_LAUNCH_CHECK();
            float *t = dz; dz = dz_next; dz_next = t;
        }
    }
    if (owner_scatter || fact) {
        rc = neumf_scatter_owner(ctx, p, g, src, R, pointwise, dz, H, stats, reg_1, reg_2, s, fact);
        if (rc) return rc;
    } else if (H) {
        hipLaunchKernelGGL((k_nmf_scatter<true>), dim3(grid_for(R, kBlock / 16 * 2)), dim3(kBlock), 0, s, p, g, src, R, d, dm,
                           model, pointwise, ctx->dpred, dz, stats, reg_1, reg_2);
    } else {
        hipLaunchKernelGGL((k_nmf_scatter<false>), dim3(grid_for(R, kBlock / 16 * 2)), dim3(kBlock), 0, s, p, g, src, R, d, dm,
                           model, pointwise, ctx->dpred, dz, stats, reg_1, reg_2);
    }
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_neumf_fit_epoch(daisy_neumf_ctx *ctx, const daisy_neumf_params *params, const daisy_neumf_params *grads,
                          const int32_t *u, const int32_t *i, const int32_t *j, int64_t n, int64_t batch, int32_t loss_type,
                          float gamma, float reg_1, float reg_2, float dropout_p, uint64_t seed_hi, int64_t step0,
                          int32_t optimizer, float lr, float *W, float *g, float *state0, float *state1, int64_t n_flat,
                          double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && params && grads && u && i && j && stats && W && g && n > 0 && batch > 0 && n_flat > 0 && step0 >= 0,
                    "neumf_fit_epoch: bad argument");
    DAISY_CHECK_ARG(optimizer >= 0 && optimizer <= 3, "neumf_fit_epoch: optimizer=%d (0 sgd, 1 adam, 2 adagrad, 3 rmsprop)", optimizer);
    DAISY_CHECK_ARG(optimizer == 0 || state0, "neumf_fit_epoch: optimizer %d needs its state", optimizer);
    DAISY_CHECK_ARG(optimizer != 1 || state1, "neumf_fit_epoch: Adam needs both moments");
    // the reference's loop (AbstractRecommender.py:119-128) over the epoch's batches, issued from here: at 256 samples per step
    // a step is ~40 us of kernels, less than the Python of one iteration around two library calls
    int64_t step = step0;
    for (int64_t s0 = 0; s0 < n; s0 += batch) {
        const int64_t B = (n - s0 < batch) ? n - s0 : batch;
        ++step;
        int rc = daisy_neumf_step_grads(ctx, params, grads, u + s0, i + s0, j + s0, B, loss_type, gamma, reg_1, reg_2, dropout_p,
                                        seed_hi | (uint64_t)step, stats, stream);
        if (rc) return rc;
        if (optimizer == 0) rc = daisy_sgd_dense(W, g, n_flat, lr, stream);
        else if (optimizer == 1) rc = daisy_adam_dense(W, g, state0, state1, n_flat, lr, 0.9f, 0.999f, 1e-8f, step, stream);
        else if (optimizer == 2) rc = daisy_adagrad_dense(W, g, state0, n_flat, lr, 1e-10f, stream);
        else rc = daisy_rmsprop_dense(W, g, state0, n_flat, lr, 0.99f, 1e-8f, stream);
        if (rc) return rc;
    }
    return DAISY_OK;
}

int daisy_gemm_nt_bf16(const uint16_t *A, const uint16_t *B, uint16_t *C, int64_t M, int32_t N, int32_t K,
                       daisy_stream_t stream) {
    DAISY_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "gemm_nt_bf16: bad argument");
    GemmOp op{};
    op.A16 = A; op.sam = K; op.sak = 1;
    op.B16 = B; op.sbn = K; op.sbk = 1;
    op.C16 = C; op.ldc = N;
    op.M = M; op.N = N; op.K = K; op.k_chunk = K;
    DAISY_CHECK_ARG(gemm_h_ok(op), "gemm_nt_bf16: needs M %% 128 == 0, N %% 64 == 0 (128 when N > 64), K %% 32 == 0, 16-byte aligned rows");
    launch_gemm_h<EPI_GATE>(op, NS(stream));          // no gate tensor: a plain bf16 store
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_gemm_tn_bf16(const uint16_t *At, const uint16_t *Bt, float *C, int64_t M, int32_t N, int64_t K,
                       int64_t k_chunk, daisy_stream_t stream) {
    DAISY_CHECK_ARG(At && Bt && C && M > 0 && N > 0 && K > 0 && k_chunk > 0, "gemm_tn_bf16: bad argument");
    GemmOp op{};                         // the weight-gradient layout: both operands [K][rows], rows contiguous
    op.A16 = At; op.sam = 1; op.sak = M;
    op.B16 = Bt; op.sbn = 1; op.sbk = N;
    op.C = C; op.ldc = N;
    op.M = M; op.N = N; op.K = K; op.k_chunk = k_chunk;
    DAISY_CHECK_ARG(gemm_h_ok(op), "gemm_tn_bf16: needs M %% 128 == 0, N %% 64 == 0 (128 when N > 64), K and k_chunk %% 32 == 0, "
                                   "M and N %% 8 == 0, 16-byte aligned operands");
    launch_gemm_h<EPI_ATOMIC>(op, NS(stream));
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_sgd_dense(float *W, float *g, int64_t n, float lr, daisy_stream_t stream) {
    DAISY_CHECK_ARG(W && g && n > 0, "sgd_dense: bad argument");
    hipLaunchKernelGGL(k_sgd_dense, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, NS(stream), W, g, n, lr);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_gemm_nt_f32(const float *A, const float *B, float *C, int64_t M, int32_t N, int32_t K,
                      daisy_stream_t stream) {
    return daisy_gemm_nt(A, B, C, M, N, K, 0, stream);
}

int daisy_gemm_nt(const float *A, const float *B, float *C, int64_t M, int32_t N, int32_t K, int32_t bf16,
                  daisy_stream_t stream) {
    DAISY_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "gemm_nt: bad argument");
    GemmOp op{};
    op.bf16 = bf16 ? 1 : 0;
    op.A = A; op.sam = K; op.sak = 1;
    op.B = B; op.sbn = K; op.sbk = 1;
    op.C = C; op.ldc = N; op.M = M; op.N = N; op.K = K; op.k_chunk = K;
    launch_gemm<EPI_STORE>(op, NS(stream));
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // extern "C"
