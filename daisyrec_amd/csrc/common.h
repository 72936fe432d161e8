// Internal helpers shared by the HIP translation units (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/daisyrec_amd.h"

namespace daisy {

void set_error(const char *fmt, ...);

#define DAISY_CHECK_ARG(cond, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            ::daisy::set_error(__VA_ARGS__); \
            return DAISY_ERR_ARG;           \
        }                                   \
    } while (0)

#define DAISY_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::daisy::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                               __FILE__, __LINE__);                                        \
            return DAISY_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define DAISY_LAUNCH_CHECK() DAISY_HIP(hipGetLastError())

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kMaxGrid = 2048;   // 256 CUs x 8 workgroups: grid-stride beyond that
constexpr int kMaxD = 512;

constexpr int kMaxGridSparse = 65536;  // head-detect kernels: most groups exit at once, so
                                       // many short workgroups beat a long grid-stride loop
inline int grid_for(int64_t work_items, int items_per_block, int cap = kMaxGrid) {
    int64_t g = (work_items + items_per_block - 1) / items_per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// occurrences of item r from entry `pos` on (the caller sits on the head of r's run):
// the LPR lanes of the group test LPR entries per step and count with a ballot.
template <class C>
__device__ __forceinline__ void count_run(const uint32_t *__restrict__ ekey, uint32_t imask,
                                          const uint2 *__restrict__ esu, int64_t pos, int64_t n,
                                          int32_t r, int lane, float &n_pos, float &n_neg) {
    const int shift = ((threadIdx.x % kWave) / C::LPR) * C::LPR;
    const unsigned long long gmask = (C::LPR == 64) ? ~0ull : (((1ull << C::LPR) - 1) << shift);
    int cp = 0, cn = 0;
    for (int64_t base = pos;; base += C::LPR) {
        const int64_t e = base + lane;
        const bool v = (e < n) && ((int32_t)((ekey[e] & imask) >> 1) == r);
        const bool ng = v && ((esu[e].x & 0x80000000u) != 0);
        const int cv = __popcll(__ballot(v) & gmask);
        const int cg = __popcll(__ballot(ng) & gmask);
        cp += cv - cg;
        cn += cg;
        if (cv < C::LPR) break;
    }
    n_pos = (float)cp;
    n_neg = (float)cn;
}

// ----------------------------------------------------------------------------
// Row fragments.  A d-float table row is spread over LPR consecutive lanes of a
// wave; each lane owns NV chunks of VEC floats: chunk c of lane l covers
// elements [(c*LPR + l)*VEC, +VEC).  With VEC=4 a chunk is one 16-byte
// global_load_dwordx4 and LPR lanes read LPR*16 contiguous bytes: d=64 -> 16
// lanes x 16 B = one 256-B row per quarter wave, fully coalesced.
// ----------------------------------------------------------------------------
// EXACT: d == LPR*VEC*NV, so no per-chunk bounds test (each test costs an exec-mask branch
// around the load; the common d = 32/64/128/256 take this path).
template <int LPR_, int VEC_, int NV_, bool EXACT_ = false>
struct RowCfg {
    static constexpr int LPR = LPR_;
    static constexpr int VEC = VEC_;
    static constexpr int NV = NV_;
    static constexpr bool EXACT = EXACT_;
    static constexpr int NE = VEC_ * NV_;            // floats per lane
    static constexpr int GROUPS_PER_WAVE = kWave / LPR_;
    static constexpr int GROUPS_PER_BLOCK = kBlock / LPR_;
};

template <class C>
struct Row {
    float v[C::NE];

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < C::NE; ++k) v[k] = 0.f;
    }
    // lane = index of this lane inside its LPR-lane group.  Shapes that do not fill their lanes exactly (d = 100,
    // 50 ...) have two flavours: the guarded load (lanes past d skip it), and load_clamped, where those lanes
    // read the row's first elements and zero the value afterwards, so no load sits behind a branch.  Kernels that
    // unroll many rows (k_fwd) need the second: with guarded loads the compiler hoisted every block's loads at
    // once and spilled kilobytes per lane (d = 200: 9.4 -> 3.0 ms per 2M-sample step); the staged kernels
    // measured 7-15 % faster with the first.
    __device__ __forceinline__ void load(const float *__restrict__ row, int lane, int d) {
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if constexpr (C::VEC == 4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (C::EXACT || e < d) t = *reinterpret_cast<const float4 *>(row + e);
                v[c * 4 + 0] = t.x; v[c * 4 + 1] = t.y; v[c * 4 + 2] = t.z; v[c * 4 + 3] = t.w;
            } else {
                v[c] = (C::EXACT || e < d) ? row[e] : 0.f;
            }
        }
    }
    // the same fragment from a row stored as bf16 (NeuMF's back-propagated tower-input gradient at precision level 2)
    __device__ __forceinline__ void load_bf16(const uint16_t *__restrict__ row, int lane, int d) {
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if constexpr (C::VEC == 4) {
                uint2 t = make_uint2(0u, 0u);
                if (C::EXACT || e < d) t = *reinterpret_cast<const uint2 *>(row + e);
                v[c * 4 + 0] = __uint_as_float(t.x << 16); v[c * 4 + 1] = __uint_as_float(t.x & 0xFFFF0000u);
                v[c * 4 + 2] = __uint_as_float(t.y << 16); v[c * 4 + 3] = __uint_as_float(t.y & 0xFFFF0000u);
            } else {
                v[c] = (C::EXACT || e < d) ? __uint_as_float((uint32_t)row[e] << 16) : 0.f;
            }
        }
    }
    __device__ __forceinline__ void load_clamped(const float *__restrict__ row, int lane, int d) {
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            const bool in = C::EXACT || e < d;
            if constexpr (C::VEC == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(row + (in ? e : 0));
                v[c * 4 + 0] = in ? t.x : 0.f; v[c * 4 + 1] = in ? t.y : 0.f;
                v[c * 4 + 2] = in ? t.z : 0.f; v[c * 4 + 3] = in ? t.w : 0.f;
            } else {
                const float t = row[in ? e : 0];
                v[c] = in ? t : 0.f;
            }
        }
    }
    // streaming flavours (nontemporal: the line is not kept in L2 / the Infinity Cache): rows that are written once
    // and read once by a LATER kernel - the stage of the staged step - should not evict the factor tables
    __device__ __forceinline__ void load_nt(const float *__restrict__ row, int lane, int d) {
        typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if constexpr (C::VEC == 4) {
                v4f t = {0.f, 0.f, 0.f, 0.f};
                if (C::EXACT || e < d) t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(row + e));
                v[c * 4 + 0] = t.x; v[c * 4 + 1] = t.y; v[c * 4 + 2] = t.z; v[c * 4 + 3] = t.w;
            } else {
                v[c] = (C::EXACT || e < d) ? __builtin_nontemporal_load(row + e) : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store_nt(float *__restrict__ row, int lane, int d) const {
        typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if (C::EXACT || e < d) {
                if constexpr (C::VEC == 4) {
                    const v4f t = {v[c * 4 + 0], v[c * 4 + 1], v[c * 4 + 2], v[c * 4 + 3]};
                    __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(row + e));
                } else {
                    __builtin_nontemporal_store(v[c], row + e);
                }
            }
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ row, int lane, int d) const {
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if (C::EXACT || e < d) {
                if constexpr (C::VEC == 4) {
                    *reinterpret_cast<float4 *>(row + e) =
                        make_float4(v[c * 4 + 0], v[c * 4 + 1], v[c * 4 + 2], v[c * 4 + 3]);
                } else {
                    row[e] = v[c];
                }
            }
        }
    }
    // fp32 hardware atomics (global_atomic_add_f32), no return value
    __device__ __forceinline__ void atomic_add_to(float *__restrict__ row, int lane, int d) const {
#pragma unroll
        for (int c = 0; c < C::NV; ++c) {
            const int e = (c * C::LPR + lane) * C::VEC;
            if (C::EXACT || e < d) {
#pragma unroll
                for (int k = 0; k < C::VEC; ++k) unsafeAtomicAdd(row + e + k, v[c * C::VEC + k]);
            }
        }
    }
};

// ---- cross-lane moves inside a 16-lane DPP row (a lane group never spans rows for LPR <= 16).
// DPP modifiers ride on the consuming VALU instruction; ds_bpermute (what __shfl lowers to)
// costs an LDS-crossbar round trip plus address math per use.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;    // row_half_mirror (lane i <- 7-i inside 8)
constexpr int kDppMirror = 0x140;        // row_mirror      (lane i <- 15-i inside 16)
constexpr int kDppRowBcast0 = 0x150;     // row_newbcast:k  (every lane of the row <- lane k)

// value of lane (group base + x) for every lane of the group; x is uniform.  LPR == 16 -> one DPP
// broadcast (the switch folds once the caller's loop is unrolled); other shapes: ds_bpermute.
template <class C>
__device__ __forceinline__ int group_bcast(int v, int x) {
    if constexpr (C::LPR == 16) {
        switch (x & 15) {
#define DAISY_BC(k) case k: return dpp_i32<kDppRowBcast0 + k>(v);
            DAISY_BC(0) DAISY_BC(1) DAISY_BC(2) DAISY_BC(3) DAISY_BC(4) DAISY_BC(5) DAISY_BC(6) DAISY_BC(7)
            DAISY_BC(8) DAISY_BC(9) DAISY_BC(10) DAISY_BC(11) DAISY_BC(12) DAISY_BC(13) DAISY_BC(14)
            default: return dpp_i32<kDppRowBcast0 + 15>(v);
#undef DAISY_BC
        }
    } else {
        return __shfl(v, x, C::LPR);
    }
}
template <class C>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, int x) { return (uint32_t)group_bcast<C>((int)v, x); }
template <class C>
__device__ __forceinline__ float group_bcast(float v, int x) {
    return __builtin_bit_cast(float, group_bcast<C>(__builtin_bit_cast(int, v), x));
}

// sum over the lanes of a group, result in every lane (bitwise identical across lanes: each
// step adds the same two partial sums on both sides)
template <class C>
__device__ __forceinline__ float group_sum(float x) {
    if constexpr (C::LPR <= 16) {
        if constexpr (C::LPR >= 2) x += dpp_f32<kDppQuadXor1>(x);
        if constexpr (C::LPR >= 4) x += dpp_f32<kDppQuadXor2>(x);
        if constexpr (C::LPR >= 8) x += dpp_f32<kDppHalfMirror>(x);
        if constexpr (C::LPR >= 16) x += dpp_f32<kDppMirror>(x);
        return x;
    } else {
#pragma unroll
        for (int off = C::LPR / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
        return x;
    }
}

template <class C>
__device__ __forceinline__ float row_dot(const Row<C> &a, const Row<C> &b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < C::NE; ++k) s = fmaf(a.v[k], b.v[k], s);
    return group_sum<C>(s);
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}

// Sum over the 64 lanes of a wave in fp32 without touching the LDS crossbar: four DPP steps inside each 16-lane row
// (every lane of a row then holds the row's sum), then the four row sums through v_readlane.  Fixed order; the
// result is the same in every lane.  (wave_sum_f64 costs twelve ds_bpermute per sum: with seven sums in sixteen
// waves that storm was 2-3 us of the small-batch step.)
__device__ __forceinline__ float wave_sum_f32_dpp(float x) {
    x += dpp_f32<kDppQuadXor1>(x);
    x += dpp_f32<kDppQuadXor2>(x);
    x += dpp_f32<kDppHalfMirror>(x);
    x += dpp_f32<kDppMirror>(x);
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}

// The same for a double: the DPP steps move the two halves (no LDS crossbar: wave_sum_f64's twelve ds_bpermute per sum
// are a microsecond where eight sums meet on the critical path of a launch-bound step).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = dpp_i32<CTRL>((int)(b & 0xFFFFFFFFll)), hi = dpp_i32<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double x) {
    x += dpp_f64<kDppQuadXor1>(x);
    x += dpp_f64<kDppQuadXor2>(x);
    x += dpp_f64<kDppHalfMirror>(x);
    x += dpp_f64<kDppMirror>(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}

// Pick the fragment shape for a runtime d.  F is a generic lambda taking a RowCfg tag.
template <class F>
inline int dispatch_d(int d, F &&f) {
    if (d <= 0 || d > kMaxD) {
        set_error("unsupported factor count d=%d (1..%d)", d, kMaxD);
        return DAISY_ERR_ARG;
    }
#ifdef DAISY_ONLY_D64       // development builds (make dev): one row shape, a tenth of the compile time
    if (d != 64) { set_error("this development build of the library only has d=64 (got %d)", d); return DAISY_ERR_ARG; }
    return f(RowCfg<16, 4, 1, true>{});
#else
    if (d == 32) return f(RowCfg<8, 4, 1, true>{});
    if (d == 64) return f(RowCfg<16, 4, 1, true>{});
    if (d == 128) return f(RowCfg<16, 4, 2, true>{});
    if (d == 256) return f(RowCfg<16, 4, 4, true>{});
    if (d % 4 == 0) {
        if (d <= 32) return f(RowCfg<8, 4, 1>{});
        if (d <= 64) return f(RowCfg<16, 4, 1>{});
        if (d <= 128) return f(RowCfg<16, 4, 2>{});
        if (d <= 256) return f(RowCfg<16, 4, 4>{});
        return f(RowCfg<32, 4, 4>{});
    }
    if (d <= 64) return f(RowCfg<16, 1, 4>{});
    if (d <= 256) return f(RowCfg<16, 1, 16>{});
    return f(RowCfg<32, 1, 16>{});
#endif
}

// ----------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) — identical to oracle/bpr_mf_numpy.py
// ----------------------------------------------------------------------------
struct Philox4 {
    uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                          uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}
__host__ __device__ __forceinline__ uint64_t philox_u64(uint64_t seed, uint64_t stream,
                                                        uint64_t index) {
    Philox4 r = philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), (uint32_t)stream,
                              (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    return ((uint64_t)r.y << 32) | r.x;
}

// ----------------------------------------------------------------------------
// Keyed bijection of [0, n), n <= 2^30: 4-round Feistel network on ceil(log2 n) bits with cycle
// walking - the construction behind thrust::shuffle (balanced halves for an even bit count).  Round keys come from Philox on the host.
// The plan builders evaluate it ~3x per interaction per epoch and it is their ALU bound, so the
// round function is built from FULL-RATE integer instructions only: two multiply-xorshift steps
// with 24-bit multiplies (v_mul_u32_u24; a 32-bit v_mul_lo_u32 is quarter rate on CDNA) - the
// half-block is at most 15 bits wide, so 24-bit operands lose nothing.
// Identical to oracle/bpr_mf_numpy.py::feistel_positions.
// ----------------------------------------------------------------------------
constexpr int kFeistelRounds = 4;          // even: the two halves are back in their places after the last round
struct FeistelKey {
    uint32_t k[kFeistelRounds];
    int bits_l, bits_r;     // the network works on bits_l + bits_r bits: bits_l = total / 2, bits_r = total - bits_l
};
__host__ __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {   // low 32 bits of (a mod 2^24)*(b mod 2^24)
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu));
#endif
}
__host__ __device__ __forceinline__ uint32_t feistel_round(uint32_t r, uint32_t key) {
    uint32_t x = r ^ key;
    x = mul24(x, 0xCC9E2Du);
    x ^= x >> 15;
    x = mul24(x, 0x85EBCBu);
    x ^= x >> 13;
    return x;
}
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {  // murmur3 finalizer (other hashes of the library)
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// One pass through the network: a bijection of [0, 2^(bits_l + bits_r)); cycle walking repeats it until the value
// is < n.  The domain is the smallest power of two >= n: with an odd bit count the halves differ by one bit and swap
// places every round (an unbalanced / alternating Feistel network), so at least half of the domain is always in range
// (a balanced network on the next EVEN bit count would walk 2.7 passes per element at n = 100 M, 2.1 at 500 M).
__host__ __device__ __forceinline__ uint32_t feistel_once(uint32_t x, const FeistelKey &fk) {
    const int a = fk.bits_l, b = fk.bits_r;
    const uint32_t mask_a = (1u << a) - 1u, mask_b = (1u << b) - 1u;
    uint32_t L = x >> b, R = x & mask_b;                    // L: a bits, R: b bits
#pragma unroll
    for (int r = 0; r < kFeistelRounds; r += 2) {
        uint32_t t = L ^ (feistel_round(R, fk.k[r]) & mask_a);      // a bits
        L = R;                                                      // b bits
        R = t;
        t = L ^ (feistel_round(R, fk.k[r + 1]) & mask_b);           // b bits
        L = R;                                                      // a bits
        R = t;
    }
    return (L << b) | R;
}
__host__ __device__ __forceinline__ uint32_t feistel_position32(uint32_t x, uint32_t n, const FeistelKey &fk) {
    do x = feistel_once(x, fk);
    while (x >= n);
    return x;
}
__host__ __device__ __forceinline__ uint64_t feistel_position(uint64_t x, uint64_t n, const FeistelKey &fk) {
    return (uint64_t)feistel_position32((uint32_t)x, (uint32_t)n, fk);
}
inline FeistelKey make_feistel_key(uint64_t n, uint64_t seed, uint64_t epoch) {
    FeistelKey fk;
    int bits = 2;
    while (bits < 30 && ((uint64_t)1 << bits) < n) ++bits;      // smallest domain >= n; n <= 2^30
    fk.bits_l = bits / 2;
    fk.bits_r = bits - fk.bits_l;
    for (int r = 0; r < kFeistelRounds; ++r)
        fk.k[r] = (uint32_t)philox_u64(seed, epoch | ((uint64_t)1 << 61), (uint64_t)r);
    return fk;
}

// ----------------------------------------------------------------------------
// rocPRIM wrappers live in sort.hip (keeps the heavy headers in one TU)
// ----------------------------------------------------------------------------
size_t sort_pairs_i32_temp_bytes(int64_t n);
// stable LSD radix sort of (key,val) int32 pairs on bits [0,end_bit)
int sort_pairs_i32(void *temp, size_t temp_bytes, const int32_t *kin, int32_t *kout,
                   const int32_t *vin, int32_t *vout, int64_t n, int end_bit, hipStream_t s);
size_t rle_u32_temp_bytes(int64_t n);
// run-length encode: unique_out/counts_out get one entry per run, *runs_out the number of runs
int rle_u32(void *temp, size_t temp_bytes, const uint32_t *in, int64_t n, uint32_t *unique_out,
            uint32_t *counts_out, uint32_t *runs_out, hipStream_t s);
size_t rle_u64_temp_bytes(int64_t n);
int rle_u64(void *temp, size_t temp_bytes, const uint64_t *in, int64_t n, uint64_t *unique_out,
            uint32_t *counts_out, uint32_t *runs_out, hipStream_t s);
size_t exclusive_scan_u32_temp_bytes(int64_t n);
int exclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int64_t n,
                       hipStream_t s);
size_t sort_pairs_u32_u64_temp_bytes(int64_t n);
int sort_pairs_u32_u64(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                       const uint64_t *vin, uint64_t *vout, int64_t n, int begin_bit, int end_bit,
                       hipStream_t s);
size_t sort_pairs_u64_u64_temp_bytes(int64_t n);
int sort_pairs_u64_u64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const uint64_t *vin, uint64_t *vout, int64_t n, int begin_bit, int end_bit,
                       hipStream_t s);
size_t sort_pairs_u64_i32_temp_bytes(int64_t n);
int sort_pairs_u64_i32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const int32_t *vin, int32_t *vout, int64_t n, int end_bit, hipStream_t s);
size_t sort_pairs_u64_i64_temp_bytes(int64_t n);
int sort_pairs_u64_i64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const int64_t *vin, int64_t *vout, int64_t n, int end_bit, hipStream_t s);
size_t sort_keys_u64_temp_bytes(int64_t n);
int sort_keys_u64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout, int64_t n,
                  int end_bit, hipStream_t s);
size_t seg_sort_desc_f32_i64_temp_bytes(int64_t n, int64_t segs);
// per-segment stable descending sort of float keys with int64 payload; segment s = [s*C,(s+1)*C)
int seg_sort_desc_f32_i64(void *temp, size_t temp_bytes, const float *kin, float *kout,
                          const int64_t *vin, int64_t *vout, int64_t segs, int64_t C, hipStream_t s);

size_t sort_pairs_desc_f32_i64_temp_bytes(int64_t n);
// whole-array stable descending sort of float keys with int64 payload
int sort_pairs_desc_f32_i64(void *temp, size_t temp_bytes, const float *kin, float *kout,
                            const int64_t *vin, int64_t *vout, int64_t n, hipStream_t s);

inline int bits_for(int64_t n) {  // number of bits needed for values in [0, n)
    int b = 1;
    while (b < 63 && ((int64_t)1 << b) < n) ++b;
    return b;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

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
    } else if (loss_type == DAISY_LOSS_CL) {  // BCEWithLogitsLoss(sum)(pos, label); neg carries the label
        const float y = neg;
        term = fmaxf(pos, 0.f) - pos * y + log1pf(expf(-fabsf(pos)));
        cp = sigmoidf_(pos) - y;
        cn = 0.f;
    } else if (loss_type == DAISY_LOSS_SL) {  // MSELoss(sum)(pos, label)
        const float e = pos - neg;
        term = e * e;
        cp = 2.f * e;
        cn = 0.f;
    } else {  // TOP1, loss.py:30-33
        const float s1 = sigmoidf_(neg - pos);
        const float s2 = sigmoidf_(neg * neg);
        term = s1 + s2;
        const float d1 = s1 * (1.f - s1);
        cp = -d1;
        cn = d1 + 2.f * neg * s2 * (1.f - s2);
    }
}


// sparse (entries sorted by row) x dense product on the item pass's segmented-reduction kernel (bpr_train.hip);
// edge_* are scratch for segsum_chunks(n_entries, d) chunks: vec f32[2*chunks*d], item i32[2*chunks],
// b f32[2*chunks], whole i32[chunks]
int64_t segsum_chunks(int64_t n_entries, int d);
// x_bf16: X holds bf16 rows (uint16_t [rows][d]) instead of fp32
int segsum_rows(const float *X, const float2 *coef, const uint32_t *ekey, const uint2 *esu, int64_t n_entries,
                int d, float *out, float *edge_vec, int32_t *edge_item, float *edge_b, int32_t *edge_whole,
                hipStream_t s, bool x_bf16 = false);

}  // namespace daisy
