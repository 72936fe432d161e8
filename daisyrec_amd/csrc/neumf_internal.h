// Shared between csrc/neumf.hip (the layer-by-layer NeuMF step) and csrc/neumf_tower.hip (the fused tower kernel):
// MFMA fragment types, bf16 conversions, the transposing LDS fragment read, the (user, item) pair layouts.
#pragma once
#include "common.h"

namespace daisy {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// two floats -> two bf16 (round to nearest even) in one dword, lo in bits 0..15: gfx950's v_cvt_pk_bf16_f32 - one
// instruction where the integer form above takes five per value (the epilogue of a 128x128 tile converts 64 values
// per lane: that was more VALU work than the tile's MFMAs at K = 128)
__device__ __forceinline__ uint32_t bf16_pack2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ bool bf16_positive(uint16_t h) { return (h & 0x8000u) == 0 && (h & 0x7FFFu) != 0; }

// An operand that is contiguous along its ROWS instead of k is copied to LDS as it lies in memory - [k][row] tiles - and
// the MFMA fragment (8 consecutive k of one row per lane) comes out of gfx950's transposing LDS read: ds_read_b64_tr_b16
// hands lane i of a 16-lane group column i of the [4 k][16 rows] block whose 16 four-element pieces the lanes address
// (measured: result[i][j] = piece[4j + i/4][i%4]), two of them per fragment (p: this lane's piece for k rows 0..3 of its
// half, the second piece 4 k rows = 4 * pitch halfwords further).
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_frag_tr(const uint16_t *p, int pitch) {
    typedef __attribute__((address_space(3))) short4v *lds_v4;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * pitch));
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// the second piece at an address of its own (swizzled layouts: the k rows 4 apart do not differ by a fixed stride)
__device__ __forceinline__ bf16x8 lds_frag_tr2(const uint16_t *p_lo, const uint16_t *p_hi) {
    typedef __attribute__((address_space(3))) short4v *lds_v4;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p_lo);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p_hi);
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// keep mask of element `idx` of dropout stream `stream` (one stream per MLP layer input)
__host__ __device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t stream, uint64_t idx,
                                                   uint32_t thresh) {
    uint32_t h = mix32((uint32_t)idx ^ (uint32_t)seed);
    h = mix32(h + (uint32_t)(idx >> 32) * 0x9E3779B9u + (uint32_t)(seed >> 32) + stream * 0x85EBCA6Bu);
    return h >= thresh;
}

// the three pair layouts of daisy_neumf_scores plus the training batch
struct PairSrc {
    const int32_t *u, *i, *j;     // training: row r < B -> (u[r], i[r]); r >= B -> (u[r-B], j[r-B])
    int64_t B;
    const int64_t *users, *items; // scoring
    int64_t C;                    // > 0: user of pair e = users[e / C];  0 with items == NULL: (users[0], e)
    int64_t base;                 // first pair of this chunk
};
__device__ __forceinline__ void pair_ids(const PairSrc &s, int64_t r, int64_t &user, int64_t &item) {
    if (s.u) {
        const int64_t b = (r < s.B) ? r : r - s.B;
        user = s.u[b];
        item = (r < s.B) ? s.i[b] : s.j[b];
    } else {
        const int64_t e = s.base + r;
        if (!s.items) { user = s.users[0]; item = e; }
        else if (s.C > 0) { user = s.users[e / s.C]; item = s.items[e]; }
        else { user = s.users[e]; item = s.items[e]; }
    }
}

// ---- the fused tower kernel (csrc/neumf_tower.hip): layers 2..3 + predict layer + criterion + their backward pass
struct TowerArgs {
    const uint16_t *tu, *ti;          // bf16 [U][4d], [I][4d]: the first layer's table products (FACT)
    const float2 *nu, *ni;            // per table row (sum |x|, sum x^2) of uM / iM
    const float *b1;                  // [4d]
    const float *W2, *W3;             // fp32 [2d][4d], [d][2d] (16-byte aligned): rounded to bf16 as the kernel loads them
    const float *b2, *b3, *Wp, *bp;   // fp32 [2d], [d], [2d], [1]
    const float *uG, *iG;             // fp32 [U][d], [I][d]
    const int32_t *u, *i, *j;         // the batch (j: negatives, or the labels of a point-wise loss)
    int64_t B;
    int pointwise, loss_type;
    float gamma;
    uint16_t *dZ1;                    // out: bf16 [R][4d], gradient wrt the first layer's pre-activation
    float *dpred;                     // out: [R]
    float *ws;                        // per-workgroup partial sums (neumf_tower_ws_bytes)
    double *wsd;                      // (set by neumf_tower_step: the doubles behind the floats of `ws`)
};
int neumf_tower_blocks(int64_t tiles);
size_t neumf_tower_ws_bytes(int d, int nblocks);
// one launch of the tower over R rows + the fixed-order reduction of the workgroups' sums into the gradients (+=) and stats
int neumf_tower_step(const TowerArgs &args, int d, int64_t R, float *gW2, float *gW3, float *gb2, float *gb3, float *gWp,
                     float *gbp, double *stats, float reg_1, float reg_2, hipStream_t s);

// ---- the small-step kernel (csrc/neumf_mid.hip): a step of at most 1024 rows whose MLP weights fit the LDS - the gather, every
// layer, the predict layer, the criterion and their backward pass, everything before the scatter, in ONE launch (fp32)
struct MidArgs {
    const float *uG, *iG, *uM, *iM;   // the embedding tables ([U][d], [I][d], [U][dm], [I][dm]); u, i, j: the batch
    const int32_t *u, *i;
    int dm;
    float *DX0;                       // out: [R][w0] gradient wrt x0 = [uM[u] | iM[item]] (dropout mask of the input applied)
    float *pred, *dpred;              // out: [R]
    const float *W[DAISY_NEUMF_MAX_LAYERS], *b[DAISY_NEUMF_MAX_LAYERS];
    const float *Wp, *bp;
    int width[DAISY_NEUMF_MAX_LAYERS + 1];
    int L, d;
    const int32_t *j;                 // the negatives, or the labels of a point-wise loss
    int B, R, pointwise, loss_type;
    float gamma;
    uint32_t thresh;                  // dropout: keep threshold (0: off), scale, seed
    float scale;
    uint64_t seed;
    float *ws;                        // per-workgroup partial sums (neumf_mid_ws_bytes)
};
bool neumf_mid_fits(int L, const int *width, int d);
size_t neumf_mid_ws_bytes(int L, const int *width, int d, int max_rows);
// the launch + the fixed-order reduction of the workgroups' sums into the gradients (+=), the loss and the norms
int neumf_mid_step(const MidArgs &args, float *const *gW, float *const *gb, float *gWp, float *gbp, double *stats, float reg_1,
                   float reg_2, hipStream_t s);

}  // namespace daisy
