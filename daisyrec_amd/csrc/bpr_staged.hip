// The STAGED SGD step and the PARTITIONED epoch plan (gfx950 / MI355X).
//
// Reference semantics: the same batch-synchronous step as bpr_train.hip
//   loader    daisy/utils/dataset.py:5-27          forward   daisy/model/MFRecommender.py:63-68
//   loss      MFRecommender.py:70-97, loss.py:5-33 backward  AbstractRecommender.py:125
//   SGD       AbstractRecommender.py:56,126
//
// Why a second organisation of the step.  The phase kernels of bpr_train.hip move 8.3 table rows per
// interaction (k_fwd 3, item pass 2 + a random 8-byte coefficient gather per entry that costs a full
// 128-byte fabric request, user pass 3.3) and the PMC counters say every one of them already runs at the
// fabric rate for the bytes it moves (profiles/r01_*): the lever left is FEWER BYTES.  Here:
//
//   k_unorm          sum over the batch of |P[u]|^2 from a per-row cache (4 B / sample): the one batch-wide
//                    quantity the user update needs BEFORE it runs (reg_2 * p / |P[u]|_F, MFRecommender.py:94)
//   k_staged_user    forward + user update in ONE pass over the user-grouped samples: gathers p_u, q_i, q_j
//                    once, forms both scores, the loss term and c = dL/dx, accumulates the seven batch sums,
//                    updates P[u] in place (single owner per row, run/slot/edge reduction like bpr_train.hip)
//                    and writes m_s = c_s * p_u(pre-step) to stage[slot(s)]: one 256-B row per sample
//   k_staged_item    segmented reduction over the item-sorted entries: gQ[i] = sum_e (+/-) m_{s(e)} - the
//                    coefficient rides inside the staged row, so an entry costs ONE row gather and no
//                    coefficient gather; the segment's owner then applies regulariser + SGD to Q[i] IN PLACE
//                    (no gQ round trip, no run lists, no separate apply pass).  Multi-GPU: writes the data
//                    term + (n_pos, n_neg) per item instead, for a reduce-scatter.
//
// Row traffic per interaction at d=64:  user pass 2 Q rows + 0.43 P rows read, 0.43 P rows + 1 stage row
// written; item pass 2 stage rows read (+ the touched Q rows once) = ~1.5 KB instead of ~2.1 KB.
// (BPR / HL have dL/dneg = -dL/dpos, so one premultiplied row serves both entries of a sample; TOP1 keeps
// the plain row in the stage and gathers its two coefficients.)
//
// The PARTITIONED epoch plan replaces the two payload-carrying radix sorts per epoch (112 B and 80 ps per
// interaction) by two stable one-digit partitions: the training set is indexed ONCE per fit (triples in
// CSR order, item entries sorted by item: daisy_train_index), and since a stable partition of a sorted
// list by batch id leaves every batch sorted, one counting pass + one scatter pass per epoch lay the epoch
// out batch by batch.  The batch id of triple t is pos(t) / B with pos = identity, the inverse of an
// explicit permutation, or the keyed Feistel bijection of (seed, epoch) computed in registers; the stage
// slot of a sample is pos(t) - k*B, which both its sample record and its two entry records can compute
// without ever meeting.  32 B of plan per interaction.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "bpr_internal.h"

namespace daisy {

// stage rows: written once by the user pass, read by the item pass of the same step and never again - the stores
// are nontemporal so that they do not evict the factor tables from L2 / the Infinity Cache (-2.5 % per step at both
// BASELINE shapes; nontemporal LOADS of the stage measured neutral to slightly worse and stay off)
#ifndef DAISY_STAGE_NT
#define DAISY_STAGE_NT 1
#endif
#if DAISY_STAGE_NT & 1
#define DAISY_STAGE_STORE(r, p, l, d) (r).store_nt(p, l, d)
#else
#define DAISY_STAGE_STORE(r, p, l, d) (r).store(p, l, d)
#endif
#if DAISY_STAGE_NT & 2
#define DAISY_STAGE_LOAD(r, p, l, d) (r).load_nt(p, l, d)
#else
#define DAISY_STAGE_LOAD(r, p, l, d) (r).load(p, l, d)
#endif
#ifndef DAISY_ITEM_WINDOW
#define DAISY_ITEM_WINDOW 1
#endif
// development probes of the item pass (dev builds only: the results of a probe build are WRONG by design).  Bit 0: the
// gathers alone - hop 1 (metadata), hop 2 (stage rows), one add per element to keep them alive, no reduction, no commit,
// no barrier: the ceiling of this access pattern inside this launch shape; bit 1: without the LDS window of Q rows; bit 2:
// the slot of an entry computed from its index instead of loaded (no dependent hop)
#ifndef DAISY_ITEM_PROBE
#define DAISY_ITEM_PROBE 0
#endif
#ifndef DAISY_ITEM_TOUCH
#define DAISY_ITEM_TOUCH 1
#endif
#ifndef DAISY_ITEM_RUN4
#define DAISY_ITEM_RUN4 16
#endif
#ifndef DAISY_ITEM_WINF
#define DAISY_ITEM_WINF 6144
#endif
#ifndef DAISY_ITEM_WAVES
#define DAISY_ITEM_WAVES 4
#endif
constexpr size_t kStreamTableBytes = (size_t)512 << 20;   // user tables beyond this are read past the caches (k_staged_user)
constexpr int kItemWinFloats = DAISY_ITEM_WINF;     // 24 KB of Q rows per workgroup of the item pass (96 rows at d = 64); rows of two
                                          // float4 per lane get 16 KB, wider ones none (their partial-sum slots already
                                          // take the LDS that four workgroups per CU leave)

// threads per workgroup of the item pass outside the three-launch form.  Its four barriers per chunk make a workgroup as
// slow as its slowest wave's gathers; two waves per workgroup instead of four (round 5, same box): 222 -> 216 us at
// BASELINE configs[1], 296 -> 285 us at 10 M x 1 M shapes; one wave (64 threads: four times the chunks, edge records and
// finishers) loses again (227 / 307 us).  Rounds 2 / 3 had measured 128 "the same" - with the commit stalls still in.
#ifndef DAISY_ITEM_BLK
#define DAISY_ITEM_BLK 128
#endif
constexpr int kStagedUserBlock = 128, kStagedItemBlock = DAISY_ITEM_BLK;    // threads per workgroup of the two passes (measured:
                                                                 // user pass 387 -> 360 us at 128, item pass indifferent)

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// =============================================================================
// partitioned epoch plan
// =============================================================================
// DAISY_PLAN_PARK (round 5): the counting kernel of the ENTRY records parks the epoch positions its Feistel walks arrive
// at (4 B per entry, through LDS so that they leave as whole-wave stores) and the entry scatter reads them instead of
// walking again: 1.34 -> 1.22 ms per epoch at BASELINE configs[1], 2.63 -> 2.41 at 10 M x 1 M shapes.  (Not the sample
// records: their scatter is not bound by the walk - parked, it ran 348 against 361 us while its count paid 131 against 119.)  Round 4 had taken this out ("a tie"): then the
// scatter wrote 2.1 x its records in partial lines and was bound by that; with array-of-structures records every plan
// kernel runs at 84-100 % of the VALU issue rate (profiles/r05_pmc_plan.txt: 651 M wave instructions x 4 cycles over
// 1024 SIMDs = 1.06 of the 1.27 ms) while the build moves 3.3 GB in 1.27 ms - the memory system is the idle side now.
#ifndef DAISY_PLAN_PARK
#define DAISY_PLAN_PARK 1
#endif
constexpr int kPartThreads = 256;
constexpr int kPartK = 8;                              // records per thread per sub-tile: samples (16-byte records)
constexpr int kPartKE = 8;                             //   entries (8-byte records; 16 per thread measured 1.56 against
                                                       //   1.30 ms per epoch at BASELINE configs[1]: two workgroups per CU
                                                       //   less, and every lane walks twice as many positions in a row)
constexpr int kPartSub = kPartThreads * kPartK;        // 2048 records: the counting kernels' staging unit
constexpr int kPartMaxTiles = 16384;                   // most tiles (workgroups) of a partition
constexpr int kPartWaves = kPartThreads / kWave;

struct PosFn {            // position of triple t in the epoch order
    int mode;             // daisy_order_mode
    FeistelKey fk;
    const uint32_t *inv;  // DAISY_ORDER_PERM: inv[t] = p with perm[p] = t
    const uint32_t *orig; // CSR row -> row of the caller's triple array (NULL: the array was in CSR order)
    uint64_t n;
};
__device__ __forceinline__ uint32_t pos_of(const PosFn &f, uint32_t t) {
    if (f.orig) t = f.orig[t];
    if (f.mode == DAISY_ORDER_FEISTEL) return (uint32_t)feistel_position((uint64_t)t, f.n, f.fk);
    if (f.mode == DAISY_ORDER_PERM) return f.inv[t];
    return t;
}
struct BatchDiv { uint32_t B, M0; int shift; };   // batch id = p / B without a hardware divide
static BatchDiv make_batch_div(int64_t B) {
    BatchDiv bd;
    bd.B = (uint32_t)B;
    bd.M0 = (B >= 2) ? (uint32_t)(((uint64_t)1 << 32) / (uint64_t)B) : 0u;
    bd.shift = -1;
    if ((B & (B - 1)) == 0) { bd.shift = 0; while (((int64_t)1 << bd.shift) < B) ++bd.shift; }
    return bd;
}
__device__ __forceinline__ uint32_t batch_of(uint32_t p, const BatchDiv &bd) {
    if (bd.shift >= 0) return p >> bd.shift;          // power-of-two batch (uniform branch)
    uint32_t q = __umulhi(p, bd.M0);        // floor(p*floor(2^32/B)/2^32) in {q_true-1, q_true}
    uint32_t r = p - q * bd.B;
    while (r >= bd.B) { r -= bd.B; ++q; }
    return q;
}

// Records of the partitioned plan (array of structures: a bucket piece of a sub-tile leaves as ONE contiguous run of
// 16-byte / 8-byte records - round 3's three separate arrays left as ~85-record pieces of 340 / 680 / 340 B, and the
// counters showed the scatter writing 2.1x its records in partial lines):
//   sample record  uint4 {user, pos item, neg item (or label), epoch position}
//   entry record   uint2 {item << 1 | slot, epoch position of its sample}
struct PartSrc {
    const int32_t *triples; int32_t user_base;          // samples, static source (CSR-ordered triples)
    const uint32_t *ent_t;                              // entries, static source: triple index | slot << 31
    const uint32_t *ent_key;                            //                         item << 1 | slot
    const uint4 *srec;                                  // samples, record source (LSD pass >= 1)
    const uint2 *erec;                                  // entries, record source
    uint32_t *park;                                     // the kind's parked epoch positions (device shuffle, first pass)
};
struct PartDst { uint4 *srec; uint2 *erec; };

// lanes of this wave that hold the same digit (the AMD counterpart of match.any: one ballot per digit bit)
__device__ __forceinline__ uint64_t match_digit(uint32_t dgt, bool valid, int nbits) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (dgt >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// Digit histogram of every tile.  KIND 0: samples from the static source   1: entries from the static source (their
// epoch positions are computed here and again by the scatter pass - round 3 parked them in memory for it: 12 B per
// record of traffic for ALU work that hides under the scatter's memory time; measured a tie to slightly faster,
// profiles/r04_plan_variants.txt "variant 3")   2 / 3: sample / entry records of a previous LSD pass.
template <int KIND>
__global__ __launch_bounds__(kPartThreads) void k_part_count(PartSrc src, PosFn pf, BatchDiv bd, int64_t n,
                                                             int shift, int nbits, int ndig, int64_t tile_elems,
                                                             int64_t ntiles, uint32_t *__restrict__ counts) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t lds_t[KIND == 1 ? kPartSub : 1];
    // the positions of a sub-tile wait here for ONE coalesced store per thread and round: stored from inside the walk -
    // a few lanes per trip - they made 440 us of count<entries>'s 262 (profiles/r05_notes.txt)
    __shared__ uint32_t lds_p[(DAISY_PLAN_PARK && KIND == 1) ? kPartSub : 1];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * tile_elems;
    const int64_t hi = (lo + tile_elems < n) ? lo + tile_elems : n;
    if (KIND < 2 && pf.mode == DAISY_ORDER_FEISTEL) {
        // Cycle walking inside a lock-step wave costs the MAXIMUM walk length of its 64 lanes per element
        // (~3.5 network passes instead of the 1.34 average at n = 0.75 * 2^2h).  So every lane owns a strip of
        // elements and steps through it at its own pace: each trip of the loop is one useful network pass for
        // every lane, and the lanes only wait for each other at the end of the strip.
        const uint32_t nn = (uint32_t)pf.n;
        for (int64_t sub = lo; sub < hi; sub += kPartSub) {
            if constexpr (KIND == 1) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kPartK; ++k) {
                    const int64_t e = sub + k * kPartThreads + threadIdx.x;
                    lds_t[k * kPartThreads + threadIdx.x] = (e < hi) ? (src.ent_t[e] & ~kNegBit) : 0u;
                }
                __syncthreads();
            }
            int j = 0;
            int64_t e = sub + threadIdx.x;
            bool active = e < hi;
            auto first = [&]() -> uint32_t {
                uint32_t t;
                if constexpr (KIND == 1) t = lds_t[j * kPartThreads + threadIdx.x];
                else t = (uint32_t)e;
                return pf.orig ? pf.orig[t] : t;
            };
            uint32_t x = active ? first() : 0u;
            while (active) {
                x = feistel_once(x, pf.fk);
                if (x < nn) {
                    atomicAdd(&hist[(batch_of(x, bd) >> shift) & 255u], 1u);
                    if constexpr (DAISY_PLAN_PARK && KIND == 1) lds_p[j * kPartThreads + threadIdx.x] = x;
                    ++j;
                    e += kPartThreads;
                    active = (j < kPartK) && (e < hi);
                    if (active) x = first();
                }
            }
            if constexpr (DAISY_PLAN_PARK && KIND == 1) {
                // (each thread reads back what it wrote: no barrier; element sub + k * 256 + thread: coalesced)
#pragma unroll
                for (int k = 0; k < kPartK; ++k) {
                    const int64_t ek = sub + k * kPartThreads + threadIdx.x;
                    if (ek < hi) src.park[ek] = lds_p[k * kPartThreads + threadIdx.x];
                }
            }
        }
    } else {
        const uint64_t lt_mask = ((uint64_t)1 << (threadIdx.x % kWave)) - 1;
        for (int64_t base = lo; base < hi; base += kPartThreads) {
            const int64_t e = base + threadIdx.x;
            const bool valid = e < hi;
            uint32_t p = 0;
            if (valid) {
                if constexpr (KIND == 0) p = pos_of(pf, (uint32_t)e);
                else if constexpr (KIND == 1) p = pos_of(pf, src.ent_t[e] & ~kNegBit);
                else if constexpr (KIND == 2) p = src.srec[e].w;
                else p = src.erec[e].y;
            }
            const uint32_t dgt = (batch_of(p, bd) >> shift) & 255u;
            const uint64_t peers = match_digit(dgt, valid, nbits);          // one LDS atomic per digit per wave
            if (valid && (peers & lt_mask) == 0) atomicAdd(&hist[dgt], (uint32_t)__popcll(peers));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < ndig) counts[(int64_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

// Stable scatter of one LSD digit.  A tile is cut into sub-tiles of 256 * K records; a wave takes 64 * K consecutive
// records of the sub-tile in K rounds of 64 (coalesced reads, and the round order = the record order, so ranks are
// stable), the sub-tile is sorted by digit in LDS and every bucket leaves as one contiguous piece of whole records.
// STATIC: the records are formed from the static index and their epoch positions are computed here - every lane walks
// the Feistel network over its own K records (k_part_count's strip walk) into LDS while the other workgroups of the CU
// are in their memory phases.
template <bool SAMPLES, bool STATIC, int K>
__global__ __launch_bounds__(kPartThreads) void k_part_scatter(PartSrc src, PosFn pf, BatchDiv bd, int64_t n,
                                                               int shift, int nbits, int ndig,
                                                               int64_t tile_elems, int64_t ntiles,
                                                               const uint32_t *__restrict__ offsets, uint32_t off_base,
                                                               PartDst dst) {
    constexpr int SUB = kPartThreads * K;
    // the sorted records; before that (STATIC) the sub-tile's epoch positions, 4 B per record, read back by the thread
    // that wrote them and dead by the time the first record is stored (two barriers in between)
    __shared__ __attribute__((aligned(16))) uint32_t recw[SUB * (SAMPLES ? 4 : 2)];
    __shared__ uint8_t sdig[SUB];
    __shared__ uint32_t wcnt[kPartWaves][256];
    __shared__ uint32_t tstart[256], goff[256], tot[256];
    __shared__ uint32_t wsum[kPartWaves];
    uint4 *rec4 = reinterpret_cast<uint4 *>(recw);
    uint2 *rec2 = reinterpret_cast<uint2 *>(recw);
    uint32_t *lp = recw;

    const int tid = threadIdx.x, wave = tid / kWave, wl = tid % kWave;
    const uint64_t lt_mask = ((uint64_t)1 << wl) - 1;
    goff[tid] = (tid < ndig) ? offsets[(int64_t)tid * ntiles + blockIdx.x] - off_base : 0u;
    const int64_t lo = (int64_t)blockIdx.x * tile_elems;
    const int64_t hi = (lo + tile_elems < n) ? lo + tile_elems : n;

    for (int64_t sub = lo; sub < hi; sub += SUB) {
#pragma unroll
        for (int w = 0; w < kPartWaves; ++w) wcnt[w][tid] = 0;
        uint32_t r_a[K], r_b[SAMPLES ? K : 1], r_c[SAMPLES ? K : 1], r_p[K], r_rank[K], r_dig[K];
        const int64_t wbase = sub + (int64_t)wave * (kWave * K);
        // ---- the records of this thread (round r: record wbase + r*64 + wl); all loads are issued before the walk
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int64_t e = wbase + r * kWave + wl;
            const int64_t ec = (e < hi) ? e : hi - 1;
            if constexpr (!STATIC) {
                if constexpr (SAMPLES) {
                    const uint4 q = src.srec[ec];
                    r_a[r] = q.x; r_b[r] = q.y; r_c[r] = q.z; r_p[r] = q.w;
                } else {
                    const uint2 q = src.erec[ec];
                    r_a[r] = q.x; r_p[r] = q.y;
                }
            } else if constexpr (SAMPLES) {
                const int32_t *row = src.triples + 3 * ec;
                r_a[r] = (uint32_t)(row[0] - src.user_base);
                r_b[r] = (uint32_t)row[1];
                r_c[r] = (uint32_t)row[2];
                r_p[r] = (uint32_t)ec;                      // (the triple whose position is wanted)
            } else {
                r_a[r] = src.ent_key[ec];
                r_p[r] = src.ent_t[ec] & ~kNegBit;
            }
        }
        if constexpr (STATIC) {
            // slot of the record this thread holds in round r: wave*64*K + r*64 + wl
            const int xw = wave * (kWave * K) + wl;
            if (DAISY_PLAN_PARK && !SAMPLES && pf.mode == DAISY_ORDER_FEISTEL) {
#pragma unroll
                for (int r = 0; r < K; ++r) {              // parked by k_part_count: no second walk
                    const int64_t e = wbase + r * kWave + wl;
                    r_p[r] = src.park[(e < hi) ? e : hi - 1];
                }
            } else if (pf.mode == DAISY_ORDER_FEISTEL) {
                const uint32_t nn = (uint32_t)pf.n;
#pragma unroll
                for (int r = 0; r < K; ++r) lp[xw + r * kWave] = pf.orig ? pf.orig[r_p[r]] : r_p[r];
                int r = 0;
                bool active = wbase + wl < hi;
                uint32_t v = active ? lp[xw] : 0u;
                while (active) {
                    v = feistel_once(v, pf.fk);
                    if (v < nn) {
                        lp[xw + r * kWave] = v;
                        ++r;
                        active = (r < K) && (wbase + r * kWave + wl < hi);
                        if (active) v = lp[xw + r * kWave];
                    }
                }
#pragma unroll
                for (int r = 0; r < K; ++r) r_p[r] = lp[xw + r * kWave];
            } else {
#pragma unroll
                for (int r = 0; r < K; ++r) r_p[r] = pos_of(pf, r_p[r]);
            }
        }
        __syncthreads();              // (wcnt is zero; nobody is still reading the previous sub-tile's records)
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const bool valid = wbase + r * kWave + wl < hi;
            const uint32_t dgt = (batch_of(r_p[r], bd) >> shift) & 255u;
            const uint64_t peers = match_digit(dgt, valid, nbits);
            const uint32_t base = wcnt[wave][dgt];                  // all lanes read ...
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            if (valid && rank == 0) wcnt[wave][dgt] = base + (uint32_t)__popcll(peers);   // ... then one lane per digit writes
            r_rank[r] = base + rank;
            r_dig[r] = valid ? dgt : 0xFFFFFFFFu;
        }
        __syncthreads();
        // digit totals of the sub-tile, their exclusive scan, and the waves' offsets inside each digit
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kPartWaves; ++w) { const uint32_t c = wcnt[w][tid]; wcnt[w][tid] = t; t += c; }
        tot[tid] = t;
        uint32_t inc = t;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t up = __shfl_up(inc, off, kWave);
            if (wl >= off) inc += up;
        }
        if (wl == kWave - 1) wsum[wave] = inc;
        __syncthreads();
        uint32_t wprefix = 0;
#pragma unroll
        for (int w = 0; w < kPartWaves; ++w) if (w < wave) wprefix += wsum[w];
        tstart[tid] = wprefix + inc - t;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (r_dig[r] != 0xFFFFFFFFu) {
                const uint32_t x = tstart[r_dig[r]] + wcnt[wave][r_dig[r]] + r_rank[r];
                if constexpr (SAMPLES) rec4[x] = make_uint4(r_a[r], r_b[r], r_c[r], r_p[r]);
                else rec2[x] = make_uint2(r_a[r], r_p[r]);
                sdig[x] = (uint8_t)r_dig[r];
            }
        }
        __syncthreads();
        const int cnt = (int)((hi - sub < SUB) ? (hi - sub) : SUB);
        for (int x = tid; x < cnt; x += kPartThreads) {
            const uint32_t dg = sdig[x];
            const int64_t o = (int64_t)goff[dg] + (x - tstart[dg]);
            if constexpr (SAMPLES) dst.srec[o] = rec4[x];
            else dst.erec[o] = rec2[x];
        }
        __syncthreads();
        goff[tid] += tot[tid];
    }
}

// inv[perm[p]] = p  (DAISY_ORDER_PERM: perm[p] = triple served at position p)
__global__ void k_invert_perm(const int64_t *__restrict__ perm, int64_t n, uint32_t *__restrict__ inv) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = perm[p];
        if (t >= 0 && t < n) inv[t] = (uint32_t)p;
    }
}

// positions handed in by the caller (daisy_epoch_plan_build_positions): inv[t] = pos[t]; bad |= 2 outside [0, n_total)
__global__ void k_positions_u32(const int64_t *__restrict__ pos, int64_t n, int64_t n_total, uint32_t *__restrict__ inv,
                                int *__restrict__ bad) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = pos[t];
        const bool ok = p >= 0 && p < n_total;
        if (!ok) atomicOr(bad, 2);
        inv[t] = ok ? (uint32_t)p : 0u;
    }
}

// off[k] = first record of batch k in the partitioned sample records (their batch ids never decrease)
__global__ void k_batch_offsets(const uint4 *__restrict__ srec, int64_t n, BatchDiv bd, int64_t nb,
                                int64_t *__restrict__ off) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k <= nb; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n;                       // first index whose batch id >= k
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)batch_of(srec[mid].w, bd) < k) lo = mid + 1; else hi = mid;
        }
        off[k] = lo;
    }
}

// ---- static index ---------------------------------------------------------------------------------------
// bad[0] |= 1 when a (user - user_base, item, item) lies outside [0,U) x [0,I) x [0,I)
// pointwise: rows are (user, item, label) - one entry per row, the third column is not an id
__global__ void k_index_entries(const int32_t *__restrict__ triples, int64_t n, int32_t user_base, int64_t U,
                                int64_t I, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                uint32_t *__restrict__ ukey, uint32_t *__restrict__ uval, int *__restrict__ bad,
                                int pointwise) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t *row = triples + 3 * t;
        const int64_t u = (int64_t)row[0] - user_base, i = row[1], j = row[2];
        const bool ok = u >= 0 && u < U && i >= 0 && i < I && (pointwise || (j >= 0 && j < I));
        if (!ok) atomicOr(bad, 1);
        if (key && pointwise) {
            key[t] = ok ? ((uint32_t)i << 1) : 0u;
            val[t] = (uint32_t)t;
        } else if (key) {
            key[2 * t] = ok ? ((uint32_t)i << 1) : 0u;
            val[2 * t] = (uint32_t)t;
            key[2 * t + 1] = ok ? (((uint32_t)j << 1) | 1u) : 1u;
            val[2 * t + 1] = (uint32_t)t | kNegBit;
        }
        if (ukey) { ukey[t] = ok ? (uint32_t)u : 0u; uval[t] = (uint32_t)t; }
    }
}

// entries per item (once per fit): the longest segment an item pass can meet decides how its edge chains are reduced.
// The entries are sorted by item: a segment's first and last entry write their (1-based) places - plain stores, one per
// item (an atomic histogram of the sorted list put a thousand consecutive adds on every address: 8.6 ms at 100 M entries).
__global__ void k_item_bounds(const uint32_t *__restrict__ ent_key, int64_t n_ent, uint32_t *__restrict__ first,
                              uint32_t *__restrict__ last) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t it = ent_key[e] >> 1;
        if (e == 0 || (ent_key[e - 1] >> 1) != it) first[it] = (uint32_t)e + 1u;
        if (e == n_ent - 1 || (ent_key[e + 1] >> 1) != it) last[it] = (uint32_t)e + 1u;
    }
}
__global__ void k_item_max_len(const uint32_t *__restrict__ first, const uint32_t *__restrict__ last, int64_t n,
                               uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t len = last[i] ? last[i] - first[i] + 1u : 0u;
        m = len > m ? len : m;
    }
    atomicMax(out, m);
}

__global__ void k_gather_triples(const int32_t *__restrict__ triples, const uint32_t *__restrict__ order, int64_t n,
                                 int32_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t *row = triples + 3 * (int64_t)order[e];
        out[3 * e] = row[0]; out[3 * e + 1] = row[1]; out[3 * e + 2] = row[2];
    }
}

__global__ void k_read_partitioned(StreamView v, int32_t *__restrict__ u, int32_t *__restrict__ i,
                                   int32_t *__restrict__ j, int32_t *__restrict__ ent_item,
                                   uint32_t *__restrict__ ent_s, int32_t *__restrict__ ent_u) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < v.E; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < v.B) {
            const uint4 r = sv_sample(v, e);
            u[e] = (int32_t)r.x;
            i[e] = (int32_t)r.y;
            j[e] = (int32_t)r.z;
        }
        const uint32_t k = sv_key(v, e);
        if (ent_item) ent_item[e] = (int32_t)(k >> 1);
        if (ent_s) ent_s[e] = ((v.e_pos[e * v.e_stride] & ~kNegBit) - v.pos_base) | ((k & 1u) ? kNegBit : 0u);
        if (ent_u) ent_u[e] = -1;          // this layout does not carry the user with the entry
    }
}

StreamView plan_stream_view(const daisy_epoch_plan *p, int64_t k) {
    int64_t lo = k * p->batch_size;
    const int c = p->p_cur;
    StreamView v;
    v.B = (p->n - lo < p->batch_size) ? (p->n - lo) : p->batch_size;
    const uint32_t pos_base = (uint32_t)lo;           // stage slot = epoch position - k*B in both layouts
    if (p->h_off) { lo = p->h_off[k]; v.B = p->h_off[k + 1] - lo; }      // a rank's share of the epoch
    const int64_t epl = p->pointwise ? 1 : 2;          // entries per sample (point-wise rows have no negative item)
    v.E = epl * v.B;
    v.s_rec = p->p_srec[c] + lo;
    v.s_user = nullptr; v.s_ij = nullptr;
    v.e_key = reinterpret_cast<const uint32_t *>(p->p_erec[c] + epl * lo);      // record {key, pos}: both at stride 2
    v.e_pos = v.e_key + 1;
    v.e_kstride = 2;
    v.e_stride = 2;
    v.umask = v.imask = 0xFFFFFFFFu;
    v.pos_base = pos_base;
    v.halt = nullptr;
    v.pointwise = p->pointwise;
    v.p_stream = 0;
    return v;
}

int plan_read_batch_partitioned(const daisy_epoch_plan *plan, int64_t k, int32_t *u, int32_t *i, int32_t *j,
                                int32_t *ent_item, uint32_t *ent_s, int32_t *ent_u, int64_t *B_out_host,
                                hipStream_t s) {
    const StreamView v = plan_stream_view(plan, k);
    hipLaunchKernelGGL(k_read_partitioned, dim3(grid_for(v.E, kBlock)), dim3(kBlock), 0, s, v, u, i, j, ent_item,
                       ent_s, ent_u);
    DAISY_LAUNCH_CHECK();
    if (B_out_host) *B_out_host = v.B;
    return DAISY_OK;
}

// record set x of the partitioned layout lives in one allocation: sample records [n] x 16 B, entry records [2n] x 8 B
static int plan_need_partitioned(daisy_epoch_plan *p, int set) {
    void **slot = set ? &p->parena2 : &p->parena;
    if (*slot) return DAISY_OK;
    const size_t n = (size_t)p->max_triples;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_s = take(n * 16), o_e = take(n * 16);
    size_t o_cnt = 0, o_tmp = 0, o_inv = 0, o_park = 0, cnt_elems = 0;
    if (set == 0) {
        const int64_t max_tiles = (2 * (int64_t)n + kPartSub - 1) / kPartSub;
        const int64_t tiles = max_tiles < kPartMaxTiles ? max_tiles : kPartMaxTiles + 1;
        cnt_elems = (size_t)256 * (size_t)(tiles + 1);
        p->ptemp_bytes = exclusive_scan_u32_temp_bytes((int64_t)cnt_elems);
        o_cnt = take(cnt_elems * 4 * 2);   // counts, then their exclusive scan
        o_tmp = take(p->ptemp_bytes);
        o_inv = take(n * 4);
        o_park = DAISY_PLAN_PARK ? take(n * 8) : 0;
    }
    void *mem = nullptr;
    hipError_t e = hipMalloc(&mem, off);
    if (e != hipSuccess) {
        set_error("epoch_plan_build_indexed: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    *slot = mem;
    p->parena_bytes += off;
    char *b = (char *)mem;
    p->p_srec[set] = (uint4 *)(b + o_s);
    p->p_erec[set] = (uint2 *)(b + o_e);
    if (set == 0) {
        p->p_counts = (uint32_t *)(b + o_cnt);
        p->p_offsets = p->p_counts + cnt_elems;
        p->ptemp = b + o_tmp;
        p->p_inv = (uint32_t *)(b + o_inv);
        p->p_park = DAISY_PLAN_PARK ? (uint32_t *)(b + o_park) : nullptr;
    }
    return DAISY_OK;
}

// tiles (workgroups) of a partition over n records whose sub-tiles hold `sub` records: at most kPartMaxTiles (the count
// buffers hold that many; fewer, longer tiles measured slower - profiles/r04_plan_variants.txt)
static void part_tiling(int64_t n, int64_t sub, int64_t &tile_elems, int64_t &ntiles) {
    // DAISY_PART_TILES: fewer tiles than the buffers hold (read per build).  Tiles of several sub-tiles otherwise need
    // more than 67 M records: the tests use it to walk that loop on small plans
    const char *env = getenv("DAISY_PART_TILES");
    int64_t cap = env ? atoll(env) : kPartMaxTiles;
    cap = cap < 1 ? 1 : (cap > kPartMaxTiles ? kPartMaxTiles : cap);
    int64_t subs = (n + sub * cap - 1) / (sub * cap);
    if (subs < 1) subs = 1;
    tile_elems = subs * sub;
    ntiles = (n + tile_elems - 1) / tile_elems;
}

// n_total == 0: the index holds the whole epoch (n rows, positions 0..n-1 from `order_mode`).  n_total > 0: it holds
// a subset and `perm` is not a permutation but the epoch POSITION of every caller row, in [0, n_total): batch k is
// made of the held rows with position in [k*B, (k+1)*B), so the batches have different sizes (h_off).
static int plan_build_partitioned(daisy_epoch_plan *p, const daisy_train_index *ix, const int64_t *perm,
                                  int order_mode, uint64_t seed, uint64_t epoch, int64_t batch_size,
                                  int64_t n_total, hipStream_t s) {
    const int64_t n = ix->n;
    const bool subset = n_total > 0;
    const int64_t nb = ((subset ? n_total : n) + batch_size - 1) / batch_size;
    const int bbits = (nb > 1) ? bits_for(nb) : 1;
    const int passes = (bbits + 7) / 8;
    int rc = plan_need_partitioned(p, 0);
    if (rc) return rc;
    if (passes > 1 && (rc = plan_need_partitioned(p, 1))) return rc;
    PosFn pf;
    pf.mode = order_mode;
    pf.fk = make_feistel_key((uint64_t)n, seed, epoch);
    pf.inv = p->p_inv;
    pf.orig = ix->orig;
    pf.n = (uint64_t)n;
    int *bad = nullptr;
    if (subset) {
        if (p->h_off_cap < nb + 1) {
            free(p->h_off);
            if (p->d_off) (void)hipFree(p->d_off);
            p->h_off = nullptr; p->d_off = nullptr; p->h_off_cap = 0;
            p->h_off = (int64_t *)malloc((size_t)(nb + 1) * 8);
            if (!p->h_off || hipMalloc((void **)&p->d_off, (size_t)(nb + 2) * 8) != hipSuccess) {
                free(p->h_off); p->h_off = nullptr; p->d_off = nullptr;
                set_error("epoch_plan_build_positions: allocating %lld batch offsets failed", (long long)(nb + 1));
                return DAISY_ERR_HIP;
            }
            p->h_off_cap = nb + 1;
        }
        bad = (int *)(p->d_off + nb + 1);
        DAISY_HIP(hipMemsetAsync(bad, 0, 8, s));
        hipLaunchKernelGGL(k_positions_u32, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, perm, n, n_total, p->p_inv, bad);
        DAISY_LAUNCH_CHECK();
    } else if (order_mode == DAISY_ORDER_PERM) {
        hipLaunchKernelGGL(k_invert_perm, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, perm, n, p->p_inv);
        DAISY_LAUNCH_CHECK();
    }
    const BatchDiv bd = make_batch_div(batch_size);

    // (Round 5 ran the two partitions of a build - entry records, sample records - side by side on two streams, hoping
    // the VALU-bound counting kernels would hide under the scatters: every kernel took twice as long and the build as long
    // as before, 1.33 -> 1.36 ms per epoch at BASELINE configs[1] (profiles/r05_notes.txt).  The scatters are bound by
    // instruction issue like the counts - ranking by ballots, LDS sort - not by memory; deleted.)
    // per record kind (entries, then samples) and LSD digit: histogram of every tile, exclusive scan, stable scatter
    for (int what = 0; what < 2; ++what) {
        const bool entries = (what == 0);
        const int64_t m = entries ? ix->n_ent : n;
        int64_t tile_elems, ntiles;
        part_tiling(m, kPartThreads * (entries ? kPartKE : kPartK), tile_elems, ntiles);
        for (int pass = 0; pass < passes; ++pass) {
            const int shift = 8 * pass;
            const int bits_here = (bbits - shift < 8) ? (bbits - shift) : 8;
            const int64_t dig_here = (pass == passes - 1) ? ((nb - 1) >> shift) + 1 : 256;
            const int ndig = (int)(dig_here < 256 ? dig_here : 256);
            // LSD passes ping-pong between the record sets and end in set 0
            const int dset = ((passes - 1 - pass) & 1);
            const int sset = dset ^ 1;
            PartSrc src;
            memset(&src, 0, sizeof(src));
            src.triples = ix->triples; src.user_base = ix->user_base;
            src.ent_t = ix->ent_t; src.ent_key = ix->ent_key;
            src.srec = p->p_srec[sset]; src.erec = p->p_erec[sset];
            src.park = entries ? p->p_park : nullptr;
            const PartDst dst{p->p_srec[dset], p->p_erec[dset]};
            const dim3 g((unsigned)ntiles), b(kPartThreads);
#define DAISY_PART_COUNT(KIND)                                                                                      \
    hipLaunchKernelGGL((k_part_count<KIND>), g, b, 0, s, src, pf, bd, m, shift, bits_here, ndig, tile_elems, ntiles, \
                       p->p_counts)
            if (pass == 0) { if (entries) DAISY_PART_COUNT(1); else DAISY_PART_COUNT(0); }
            else { if (entries) DAISY_PART_COUNT(3); else DAISY_PART_COUNT(2); }
#undef DAISY_PART_COUNT
            DAISY_LAUNCH_CHECK();
            rc = exclusive_scan_u32(p->ptemp, p->ptemp_bytes, p->p_counts, p->p_offsets, (int64_t)ndig * ntiles, s);
            if (rc) return rc;
#define DAISY_PART_SCATTER(SAMPLES, STATIC, KK)                                                                      \
    hipLaunchKernelGGL((k_part_scatter<SAMPLES, STATIC, KK>), g, b, 0, s, src, pf, bd, m, shift, bits_here, ndig,      \
                       tile_elems, ntiles, p->p_offsets, 0u, dst)
            if (entries) { if (pass == 0) DAISY_PART_SCATTER(false, true, kPartKE); else DAISY_PART_SCATTER(false, false, kPartKE); }
            else { if (pass == 0) DAISY_PART_SCATTER(true, true, kPartK); else DAISY_PART_SCATTER(true, false, kPartK); }
#undef DAISY_PART_SCATTER
            DAISY_LAUNCH_CHECK();
        }
    }
    p->p_cur = 0;
    p->hot_item_share = ix->n_ent > 0 ? (double)ix->max_item_entries / (double)ix->n_ent : 0.0;
    p->n = n; p->batch_size = batch_size; p->num_batches = nb;
    p->pointwise = ix->pointwise;
    p->kind = 1;
    if (subset) {          // where every batch starts: one small copy and one host sync per epoch
        hipLaunchKernelGGL(k_batch_offsets, dim3(grid_for(nb + 1, kBlock)), dim3(kBlock), 0, s, p->p_srec[0], n, bd, nb,
                           p->d_off);
        DAISY_LAUNCH_CHECK();
        int bad_host[2] = {0, 0};
        if (hipMemcpyAsync(p->h_off, p->d_off, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(bad_host, bad, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            set_error("epoch_plan_build_positions: reading the batch offsets failed");
            p->built = false;
            return DAISY_ERR_HIP;
        }
        if (bad_host[0]) {
            set_error("epoch_plan_build_positions: a position lies outside [0, %lld)", (long long)n_total);
            p->built = false;
            return DAISY_ERR_ARG;
        }
    } else if (p->h_off) {
        free(p->h_off); p->h_off = nullptr;
        if (p->d_off) (void)hipFree(p->d_off);
        p->d_off = nullptr; p->h_off_cap = 0;
    }
    p->built = true;
    p->build_gen = next_plan_build_id();
    return DAISY_OK;
}

// =============================================================================
// staged step kernels
// =============================================================================
// samples per lane group per chunk in the user pass (3 row gathers each).  Measured at d=64: 4 (128 VGPRs,
// 4 waves/SIMD) beats 8 (178 VGPRs, 2 waves/SIMD) by 10 % at C2 shapes and ties at C3 shapes.
// BLK: threads per workgroup.  A workgroup's waves load together and compute together (the barriers of the
// run/slot reduction keep them in step), so smaller workgroups interleave memory and VALU phases better
// across the CU; the price is more chunk boundaries (edge records).
template <class C, int BLK = kBlock>
struct StagedUserCfg {
    static constexpr int RUN_BY_REGS = (C::NE <= 8) ? 4 : 2;
    static constexpr int RUN = RUN_BY_REGS < C::LPR ? RUN_BY_REGS : C::LPR;
    static constexpr int G = BLK / C::LPR;
    static constexpr int E = G * RUN;
};
// SPARSE: few entries per item (small batches, or huge item tables): nearly every entry ends a segment, and each end is a
// commit that needs the item's row of Q - a DEPENDENT read inside the reduction loop wherever the row is not in the LDS
// window (which holds a contiguous item range and therefore only helps dense batches).  At B = 4096 over 100 K items the
// 16 commits of a 16-entry run were 16 trips to memory one after the other: 14 us for a pass that moves 2 MB.  The sparse
// flavour takes runs of 4 entries (four times the workgroups: these batches do not fill the chip anyway) and gathers
// the Q row of EVERY entry together with its staged row, so a commit waits for nothing.
template <class C, int BLK = kBlock, bool SPARSE = false>
struct StagedItemCfg {
#ifndef DAISY_ITEM_RUN8
#define DAISY_ITEM_RUN8 8
#endif
#ifndef DAISY_ITEM_RUN16
#define DAISY_ITEM_RUN16 4
#endif
    // staged rows in flight per lane group: 64 row registers in every shape (16 x 4 floats, 8 x 8, 4 x 16)
    static constexpr int RUN_BY_REGS = SPARSE ? ((C::NE <= 8) ? 4 : 2)
                                              : ((C::NE <= 4) ? DAISY_ITEM_RUN4 : ((C::NE <= 8) ? DAISY_ITEM_RUN8 : DAISY_ITEM_RUN16));
    static constexpr int RUN = RUN_BY_REGS < C::LPR ? RUN_BY_REGS : C::LPR;
    static constexpr int G = BLK / C::LPR;
    static constexpr int E = G * RUN;
};

template <class C>
__global__ __launch_bounds__(kBlock) void k_row_sqnorm(const float *__restrict__ W, int64_t rows, int d,
                                                       float *__restrict__ out) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < rows; r += gstride) {
        Row<C> w;
        w.load(W + r * d, lane, d);
        const float s = row_dot<C>(w, w);
        if (lane == 0) out[r] = s;
    }
}

// partials[block] = sum over the block's samples of |P[u_s]|^2 (from the cache)
__global__ __launch_bounds__(kBlock) void k_unorm(const float *__restrict__ p_sqnorm, StreamView v,
                                                  double *__restrict__ partials) {
    double acc = 0.0;
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < v.B;
         s += (int64_t)gridDim.x * blockDim.x)
        acc += (double)p_sqnorm[sv_user(v, s)];
    __shared__ double sm[kBlock / kWave];
    const double w = wave_sum_f64(acc);
    if ((threadIdx.x % kWave) == 0) sm[threadIdx.x / kWave] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < kBlock / kWave; ++k) t += sm[k];
        partials[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kBlock) void k_unorm_reduce(const double *__restrict__ partials, int nblocks,
                                                         double *__restrict__ stats) {
    __shared__ double sm[kBlock];
    double t = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) t += partials[b];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[DAISY_ST_SQ_U_PRE] = sm[0];
}

// The NEXT batch's pre-norm riding on this step's item pass (single-GPU epoch loops): once the user pass and its edge
// kernel are done, P and the row-norm cache are final for this step, so the extra workgroups [first_block, gridDim.x)
// of the item-pass launch do k_unorm's work for batch k+1 and an extra workgroup of the item-edge launch adds the
// partial sums (RideReduce) - two launches less per step (k_unorm, k_unorm_reduce: ~12 us of kernels and two
// kernel boundaries of a 0.57 ms step, one of five launches below ~130 k samples).
struct RideUnorm {
    const float *p_sqnorm; const uint4 *s_rec; int64_t B;        // the sample records of the NEXT batch (partitioned plan)
    double *partials; int first_block, nblocks;      // nblocks == 0: nothing rides
};
struct RideReduce { const double *partials; int n; double *stats; };    // n == 0: nothing rides

__device__ __forceinline__ void unorm_block(const RideUnorm &r, int b) {
    double acc = 0.0;
    for (int64_t s = (int64_t)b * blockDim.x + threadIdx.x; s < r.B; s += (int64_t)r.nblocks * blockDim.x)
        acc += (double)r.p_sqnorm[r.s_rec[s].x];
    __shared__ double sm_ride[kBlock / kWave];
    const double w = wave_sum_f64(acc);
    if ((threadIdx.x % kWave) == 0) sm_ride[threadIdx.x / kWave] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)blockDim.x / kWave; ++k) t += sm_ride[k];
        r.partials[b] = t;
    }
}

// Sum |P[u_s]|^2 over the batch as the user pass needs it before its first update.  n_pre == 0: stats holds it (the
// phase API: the host has all-reduced it over the ranks).  Otherwise the single-GPU step skips k_unorm_reduce and
// every wave adds k_unorm's n_pre (<= kPreBlocks) partial sums itself - the same loads in the same order in every
// wave of every workgroup, so all of them see the same bits.
struct PreNorm { const double *partials; int n; };
__device__ __forceinline__ double prenorm_sum(const double *__restrict__ stats, PreNorm pre) {
    if (pre.n == 0) return stats[DAISY_ST_SQ_U_PRE];
    double t = 0.0;
    for (int b = threadIdx.x % kWave; b < pre.n; b += kWave) t += pre.partials[b];
    return wave_sum_f64(t);
}

// what the owner of a finished user run does with g = sum_s dL/dx_s (q_i - q_j) over the run's n samples:
// regulariser (MFRecommender.py:94-95), the optimiser's step on P[u] in place, the row-norm cache
template <class C, bool ADAM>
__device__ __forceinline__ void user_commit(float *__restrict__ P, float *__restrict__ p_sqnorm, int64_t user, Row<C> &p,
                                            const Row<C> &acc, float n, float reg_1, float rU, const RowOpt &opt, int lane,
                                            int d, bool stream_row = false, const Row<C> *m_pre = nullptr,
                                            const Row<C> *v_pre = nullptr) {
    const float w1 = reg_1 * n, w2 = rU * n;
    Row<C> g;
#pragma unroll
    for (int k = 0; k < C::NE; ++k) g.v[k] = acc.v[k] + fmaf(w2, p.v[k], w1 * sgn(p.v[k]));
    row_apply<C, ADAM>(p, g, opt, user, lane, d, m_pre, v_pre);
    if (stream_row) p.store_nt(P + user * d, lane, d);      // (a table far beyond the caches: see StreamView::p_stream)
    else p.store(P + user * d, lane, d);
    const float sq = row_dot<C>(p, p);
    if (lane == 0) p_sqnorm[user] = sq;
}

// FM: the user bias follows its user's run (u_bias[u] -= lr * sum of the run's dL/dpos + dL/dneg, or the sum itself
// for a dense optimiser)
__device__ __forceinline__ void user_bias_commit(const StagedBias &fm, int64_t user, float sb, float lr, int lane) {
    if (fm.bu && lane == 0) {
        if (fm.grad_out) fm.g_bu[user] = sb;
        else fm.bu[user] = fmaf(-lr, sb, fm.bu[user]);
    }
}

struct UserEdges {
    float *vec;        // [2*nchunks][d] partial user gradients of runs that cross a chunk boundary
    int32_t *user;     // [2*nchunks]    their user (-1: none); [2c] head edge, [2c+1] tail edge
    float *n;          // [2*nchunks][2] their sample counts and (FM) coefficient sums
    int32_t *whole;    // [nchunks]      the head edge's run also fills the whole chunk
};

// how a sample reaches the item pass:
//   kModePremul  pairwise loss with dL/dneg = -dL/dpos (BPR, HL): stage[slot] = dL/dpos * p_u serves both entries
//   kModePlain   any pairwise loss: stage[slot] = p_u, coef[slot] = (dL/dpos, dL/dneg) (TOP1; FM's pairwise runs,
//                whose item biases need the bare coefficients)
//   kModePoint   point-wise loss (CL / SL, MFRecommender.py:75-81): rows are (user, item, label), ONE entry per
//                sample, stage[slot] = dL/dpred * p_u (+ coef[slot].x = dL/dpred when there are FM biases)
enum { kModePremul = 0, kModePlain = 1, kModePoint = 2 };

// Forward + user update in one pass over the user-grouped samples (see the header comment).
//   in-run user:          its group owns P[u]: regulariser + optimiser step in place
//   run crossing groups:  partial sums parked in LDS, added by the slot's finisher in group order
//   run crossing chunks:  edge records, chained by k_staged_user_edges in chunk order
// (no atomics, fixed summation order: bitwise reproducible).
// HAS_POS: the stage slot of a sample comes from the plan (partitioned layout) instead of its position.
template <class C, int BLK, int MODE, bool HAS_POS, bool ADAM>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu((C::NE <= 4 && !ADAM) ? 4 : 2, 8))) void k_staged_user(
    float *__restrict__ P, const float *__restrict__ Q, StreamView v, int d, const double *__restrict__ stats,
    RowOpt opt, float reg_1, float reg_2, int loss_type, float gamma, float *__restrict__ stage,
    float2 *__restrict__ coef, float *__restrict__ p_sqnorm, double *__restrict__ partials, UserEdges ed,
    PreNorm pre, StagedBias fm) {
    // an earlier step of this epoch had a non-finite loss: the epoch has stopped.  The word is requested here and looked
    // at behind the first chunk's gathers (before anything is stored): a launch-bound step - a few thousand samples -
    // is a chain of dependent trips to memory, and this one now travels with the metadata instead of in front of it
    const double halt_word = v.halt ? *v.halt : 0.0;
    constexpr int G = StagedUserCfg<C, BLK>::G, RUN = StagedUserCfg<C, BLK>::RUN, E = StagedUserCfg<C, BLK>::E;
    constexpr int ROWF = C::NE * C::LPR;
    constexpr bool PAIR = MODE != kModePoint;
    constexpr bool BIAS = MODE != kModePremul;          // FM biases ride on the plain / point flavours only
    __shared__ float part_acc[2 * G * ROWF];
    __shared__ float part_n[2 * G], part_b[2 * G];
    __shared__ int part_slot[2 * G];
    // the pre-step row of P of a run that leaves its group through the TAIL: the slot's finisher commits from it.  (Until
    // round 5 the finisher loaded the row again - a dependent trip to memory behind this chunk's row and stage STORES, which
    // the vector-memory counter also counts: s_waitcnt vmcnt(0) in front of the commit waited for all of them.)
    __shared__ float part_p[G * ROWF];
    __shared__ float part_m[ADAM ? G * ROWF : 4], part_v[ADAM ? G * ROWF : 4];      // ... and (Adam) its moments
    __shared__ int slot_user[G + 1], slot_next[G + 1];
    __shared__ int run_first[G], run_last[G];

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const int64_t n = v.B;
    const int64_t nchunks = (n + E - 1) / E;
    const float rU = inv_or_zero(sqrt(prenorm_sum(stats, pre)), reg_2);
    const bool has_bias = BIAS && fm.bu != nullptr;
    const float b0 = has_bias ? fm.b0[0] : 0.f;
    float acc7[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * E;
        const int64_t t0 = c0 + (int64_t)group * RUN;
        const int64_t t1 = (t0 + RUN < n) ? (t0 + RUN) : n;
        const int cnt = (t0 < n) ? (int)(t1 - t0) : 0;

        // ---- hop 1: metadata of the run (lane x <- sample x) and its neighbours; every load is
        // unconditional on a clamped address (a branch around a load drains vmcnt first)
        const int64_t last = n - 1;
        const int64_t il = (t0 + lane < n) ? (t0 + lane) : last;
        const int64_t i_prev = (t0 > 0) ? ((t0 - 1 < n) ? t0 - 1 : last) : 0, i_next = (t1 < n) ? t1 : last;
        const int64_t i_cprev = (c0 > 0) ? c0 - 1 : 0;
        uint32_t k_me, slot_me, k_prev, k_next, k_cprev;
        int2 ij_me;
        if constexpr (HAS_POS) {            // partitioned plan: one 16-byte record per sample
            const uint4 rec = v.s_rec[il];
            k_me = rec.x; ij_me = make_int2((int)rec.y, (int)rec.z); slot_me = rec.w - v.pos_base;
            k_prev = v.s_rec[i_prev].x; k_next = v.s_rec[i_next].x; k_cprev = v.s_rec[i_cprev].x;
        } else {
            k_me = v.s_user[il] & v.umask; ij_me = v.s_ij[il]; slot_me = (uint32_t)il;
            k_prev = v.s_user[i_prev] & v.umask; k_next = v.s_user[i_next] & v.umask; k_cprev = v.s_user[i_cprev] & v.umask;
        }
        const int32_t my_user = (lane < cnt) ? (int32_t)k_me : -1;
        const int2 my_ij = (lane < cnt) ? ij_me : make_int2(0, 0);
        const int32_t user_prev = (cnt > 0 && t0 > 0) ? (int32_t)k_prev : -1;
        const int32_t user_next = (cnt > 0 && t1 < n) ? (int32_t)k_next : -1;
        const int32_t chunk_prev_user = (c0 > 0) ? (int32_t)k_cprev : -1;
        if (tid < 2 * G) part_slot[tid] = -1;
        if (tid <= G) { slot_user[tid] = -1; slot_next[tid] = 0; }
        const int32_t user_first = group_bcast<C>(my_user, 0);
        const int32_t user_last = __shfl(my_user, cnt > 0 ? cnt - 1 : 0, C::LPR);
        if (lane == 0) {
            run_first[group] = cnt > 0 ? user_first : -2;
            run_last[group] = cnt > 0 ? user_last : -2;
        }
        // FM: the biases of lane x's own sample (scalar gathers: the tables are small next to the factor rows)
        float my_bsp = 0.f, my_bsn = 0.f;
        if constexpr (BIAS) {
            if (has_bias && lane < cnt) {
                const float bu = fm.bu[k_me];
                my_bsp = (bu + fm.bi[ij_me.x]) + b0;                 // FMRecommender.py:66
                if constexpr (PAIR) my_bsn = (bu + fm.bi[ij_me.y]) + b0;
            }
        }

        // ---- hop 2: the rows of every sample of the run
        Row<C> qi[RUN], qj[PAIR ? RUN : 1], pr[RUN];
        // Adam: the moments of every sample's user row travel with the row (round 5): the run's owner then commits from
        // registers.  (Round 4 had measured exactly this as "no change" - with its first use still behind a vmcnt(0) wait
        // for the run's stage stores; see the touch below.)
        Row<C> pm[ADAM ? RUN : 1], pv[ADAM ? RUN : 1];
        auto moments = [&](int x, int64_t urow) {
            if constexpr (ADAM) {
                pm[x].load(opt.m + urow * d, lane, d);
                pv[x].load(opt.v + urow * d, lane, d);
            }
        };
        if (__all(cnt == RUN) && v.p_stream) {     // (wave-uniform: no branch between the gathers)
            // a user table far beyond the Infinity Cache: its rows come back once every few steps, so they are read and
            // written past the caches and leave them to Q (10 M x 1 M shapes: 449 -> 420 us per pass with BOTH the loads
            // and the owner's store nontemporal - either alone measured no gain; at 1 M users, where most rows return in
            // the next step, 3 % slower: the host decides)
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                qi[x].load(Q + (int64_t)group_bcast<C>(my_ij.x, x) * d, lane, d);
                if constexpr (PAIR) qj[x].load(Q + (int64_t)group_bcast<C>(my_ij.y, x) * d, lane, d);
                pr[x].load_nt(P + (int64_t)group_bcast<C>(my_user, x) * d, lane, d);
                moments(x, (int64_t)group_bcast<C>(my_user, x));
            }
        } else if (__all(cnt == RUN)) {
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                qi[x].load(Q + (int64_t)group_bcast<C>(my_ij.x, x) * d, lane, d);
                if constexpr (PAIR) qj[x].load(Q + (int64_t)group_bcast<C>(my_ij.y, x) * d, lane, d);
                pr[x].load(P + (int64_t)group_bcast<C>(my_user, x) * d, lane, d);
                moments(x, (int64_t)group_bcast<C>(my_user, x));
            }
        } else {
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const int32_t ux = group_bcast<C>(my_user, x);
                const int ix = group_bcast<C>(my_ij.x, x);
                const int jx = group_bcast<C>(my_ij.y, x);
                if (x < cnt) {
                    qi[x].load(Q + (int64_t)ix * d, lane, d);
                    if constexpr (PAIR) qj[x].load(Q + (int64_t)jx * d, lane, d);
                    pr[x].load(P + (int64_t)ux * d, lane, d);
                    moments(x, (int64_t)ux);
                } else {
                    qi[x].zero(); pr[x].zero();
                    if constexpr (PAIR) qj[x].zero();
                    if constexpr (ADAM) { pm[x].zero(); pv[x].zero(); }
                }
            }
        }
        if (halt_word > 0.0) return;           // (uniform over the grid; nothing has been written yet)
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
            // ---- forward: scores of the run's samples -> lane x; one loss epilogue per run, a sample per lane
            float my_sp = 0.f, my_sn = 0.f;
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                if (x < cnt) {
                    const float sp = row_dot<C>(pr[x], qi[x]);
                    float sn = 0.f;
                    if constexpr (PAIR) sn = row_dot<C>(pr[x], qj[x]);
                    if (lane == x) { my_sp = sp; my_sn = sn; }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) {
                        acc7[1] += fabsf(pr[x].v[k]);
                        acc7[2] += fabsf(qi[x].v[k]);
                        acc7[4] = fmaf(pr[x].v[k], pr[x].v[k], acc7[4]);
                        acc7[5] = fmaf(qi[x].v[k], qi[x].v[k], acc7[5]);
                        if constexpr (PAIR) {
                            acc7[3] += fabsf(qj[x].v[k]);
                            acc7[6] = fmaf(qj[x].v[k], qj[x].v[k], acc7[6]);
                        }
                    }
                }
            }
            float2 my_c = make_float2(0.f, 0.f);
            if (lane < cnt) {
                float term;
                if constexpr (BIAS) { my_sp += my_bsp; my_sn += my_bsn; }
                if constexpr (!PAIR) my_sn = (float)my_ij.y;               // the label (sampler.py:93-98)
                pair_coef(loss_type, my_sp, my_sn, gamma, term, my_c.x, my_c.y);
                acc7[0] += term;
                if constexpr (BIAS) acc7[7] += my_c.x + my_c.y;            // dL/d bias_ (FM)
                if (MODE == kModePlain || (MODE == kModePoint && has_bias)) coef[slot_me] = my_c;
            }

            // ---- user gradient over the runs of equal users; the staged rows leave on the way
            int32_t cur_user = user_first;
            Row<C> pcur = pr[0];                     // pre-step P row of the current run's user
            Row<C> mcur, vcur;                       // (Adam) and its moments
            if constexpr (ADAM) {
                mcur = pm[0]; vcur = pv[0];
                // the moment registers are "used" here, in front of the first store of this chunk: the compiler's wait
                // for their gathers happens now, and the commits below wait for nothing
#pragma unroll
                for (int x = 0; x < RUN; ++x)
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) { asm volatile("" : "+v"(pm[x].v[k])); asm volatile("" : "+v"(pv[x].v[k])); }
            }
            Row<C> acc;
            acc.zero();
            float cn_ = 0.f, sb_ = 0.f;
            auto finish = [&](bool ends_here, bool to_next_chunk, const Row<C> &prow) {
                if (cur_slot < 0 && ends_here) {     // this group owns P[cur_user]
                    Row<C> pn = prow;
                    user_commit<C, ADAM>(P, p_sqnorm, cur_user, pn, acc, cn_, reg_1, rU, opt, lane, d, v.p_stream != 0,
                                         ADAM ? &mcur : nullptr, ADAM ? &vcur : nullptr);
                    if constexpr (BIAS) user_bias_commit(fm, cur_user, sb_, opt.lr, lane);
                } else {
                    const int s = (cur_slot >= 0) ? cur_slot : group + 1;
                    const int q = group * 2 + ((cur_slot >= 0) ? 0 : 1);
                    float *dst = part_acc + q * ROWF;
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) dst[k * C::LPR + lane] = acc.v[k];
                    if (cur_slot < 0) {              // the run began here: its pre-step row for slot group + 1's finisher
                        float *pd = part_p + group * ROWF;
#pragma unroll
                        for (int k = 0; k < C::NE; ++k) pd[k * C::LPR + lane] = prow.v[k];
                        if constexpr (ADAM) {
                            float *md = part_m + group * ROWF, *vd = part_v + group * ROWF;
#pragma unroll
                            for (int k = 0; k < C::NE; ++k) { md[k * C::LPR + lane] = mcur.v[k]; vd[k * C::LPR + lane] = vcur.v[k]; }
                        }
                    }
                    if (lane == 0) {
                        part_slot[q] = s;
                        part_n[q] = cn_;
                        part_b[q] = sb_;
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
                    const uint32_t sl = group_bcast<C>(slot_me, x);
                    if (ux != cur_user) {
                        finish(true, false, pcur);
                        cur_user = ux;
                        cur_slot = -1;
                        pcur = pr[x];
                        if constexpr (ADAM) { mcur = pm[x]; vcur = pv[x]; }
                        acc.zero();
                        cn_ = 0.f; sb_ = 0.f;
                    }
                    if constexpr (MODE != kModePlain) {
                        Row<C> m;
#pragma unroll
                        for (int k = 0; k < C::NE; ++k) m.v[k] = cp * pr[x].v[k];
                        DAISY_STAGE_STORE(m, stage + (int64_t)sl * d, lane, d);
                    } else {
                        DAISY_STAGE_STORE(pr[x], stage + (int64_t)sl * d, lane, d);
                    }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) {
                        if constexpr (PAIR) acc.v[k] = fmaf(cp, qi[x].v[k], fmaf(cn, qj[x].v[k], acc.v[k]));
                        else acc.v[k] = fmaf(cp, qi[x].v[k], acc.v[k]);
                    }
                    cn_ += 1.f;
                    if constexpr (BIAS) sb_ += cp + cn;
                }
            }
            const bool continues = (t1 < n) && (user_next == cur_user);
            finish(!continues, continues && (group == G - 1), pcur);
        }
        __syncthreads();

        // ---- one finisher per used slot; runs shared with a neighbouring chunk go to the edges
        if (tid == 0) { ed.user[2 * chunk] = -1; ed.user[2 * chunk + 1] = -1; ed.whole[chunk] = 0; }
        __syncthreads();
        for (int s = group; s <= G; s += G) {
            const int uu = slot_user[s];
            if (uu < 0) continue;
            Row<C> g;
            g.zero();
            float ns = 0.f, sb = 0.f;
            auto add_part = [&](int q) {
                const float *src = part_acc + q * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) g.v[k] += src[k * C::LPR + lane];
                ns += part_n[q];
                sb += part_b[q];
            };
            // slot s in group order: the tail of group s - 1 (where the run began), then the heads of the groups s, s + 1,
            // ... it runs through (see k_staged_item: the same chain instead of a scan over all 2 G parked sums)
            if (s > 0 && part_slot[2 * s - 1] == s) add_part(2 * s - 1);
            for (int q = 2 * s; q < 2 * G && part_slot[q] == s; q += 2) add_part(q);
            const bool from_prev = (s == 0), to_next = slot_next[s] != 0;
            if (!from_prev && !to_next) {
                Row<C> p;                            // (s >= 1: the tail of group s - 1 parked the row)
                const float *ps = part_p + (s - 1) * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) p.v[k] = ps[k * C::LPR + lane];
                Row<C> mrow, vrow;
                if constexpr (ADAM) {
                    const float *ms = part_m + (s - 1) * ROWF, *vs = part_v + (s - 1) * ROWF;
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) { mrow.v[k] = ms[k * C::LPR + lane]; vrow.v[k] = vs[k * C::LPR + lane]; }
                }
                user_commit<C, ADAM>(P, p_sqnorm, uu, p, g, ns, reg_1, rU, opt, lane, d, false, ADAM ? &mrow : nullptr,
                                     ADAM ? &vrow : nullptr);
                if constexpr (BIAS) user_bias_commit(fm, uu, sb, opt.lr, lane);
            } else {
                const int64_t e = 2 * chunk + (from_prev ? 0 : 1);
                g.store(ed.vec + e * d, lane, d);
                if (lane == 0) {
                    ed.user[e] = uu;
                    ed.n[2 * e] = ns;
                    ed.n[2 * e + 1] = sb;
                    if (from_prev && to_next) ed.whole[chunk] = 1;
                }
            }
        }
        __syncthreads();
    }
    __shared__ double sm7[BLK / kWave][8];
    const int wave = tid / kWave;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double w = (double)wave_sum_f32_dpp(acc7[k]);     // fp32 inside the wave, fp64 across waves and workgroups
        if ((tid % kWave) == 0) sm7[wave][k] = w;
    }
    __syncthreads();
    if (tid < 8) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / kWave; ++w) t += sm7[w][tid];
        partials[(int64_t)blockIdx.x * 8 + tid] = t;           // [7]: sum of dL/dpos + dL/dneg (FM's bias_; 0 without biases)
    }
}

// chains of edge records: the chunk whose TAIL edge starts a run owns it.  A user whose run crosses a
// chunk boundary is touched by nobody else in this step, so P[u] is still the pre-step row here.
// The single-GPU step gives this launch one more workgroup (red.nblocks > 0), which does what k_reduce_partials
// would do in a launch of its own: the user pass's per-workgroup sums -> stats, norms, loss.
struct ReduceJob {
    const double *partials;
    int nblocks;                 // 0: no reduction rides on this launch
    double *stats, *epoch_acc, *step_loss;
};
template <class C, bool ADAM>
__device__ __forceinline__ void user_edges_block(unsigned bid, unsigned nb, float *__restrict__ P, int64_t nchunks, int d,
                                                 const double *__restrict__ stats, const RowOpt &opt, float reg_1,
                                                 float reg_2, const UserEdges &ed, float *__restrict__ p_sqnorm,
                                                 const PreNorm &pre, const ReduceJob &red, const StagedBias &fm) {
    const double sq_pre = prenorm_sum(stats, pre);
    if (red.nblocks > 0) {            // workgroup 0 (dispatched first: its chain of dependent loads is the longest)
        if (bid == 0) {
            reduce_partials_block(red.partials, red.nblocks, red.stats, true, reg_1, reg_2, red.epoch_acc, red.step_loss);
            if (threadIdx.x == 0 && pre.n) red.stats[DAISY_ST_SQ_U_PRE] = sq_pre;
            // FM, SGD: bias_ -= lr * sum_b (dL/dpos + dL/dneg)  (a dense optimiser reads the sum from stats instead)
            if (threadIdx.x == 0 && fm.bu && !fm.grad_out)
                fm.b0[0] = fmaf(-opt.lr, (float)red.stats[DAISY_ST_SUM_COEF], fm.b0[0]);
            return;
        }
        nb -= 1; bid -= 1;
    }
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)nb * C::GROUPS_PER_BLOCK;
    const float rU = inv_or_zero(sqrt(sq_pre), reg_2);
    for (int64_t c = (int64_t)bid * C::GROUPS_PER_BLOCK + group; c < nchunks; c += gstride) {
        const int uu = ed.user[2 * c + 1];
        if (uu < 0) continue;
        Row<C> acc, t;
        acc.load(ed.vec + (2 * c + 1) * d, lane, d);
        float ns = ed.n[2 * (2 * c + 1)], sb = ed.n[2 * (2 * c + 1) + 1];
        for (int64_t k = c + 1; k < nchunks && ed.user[2 * k] == uu; ++k) {
            t.load(ed.vec + (2 * k) * d, lane, d);
#pragma unroll
            for (int q = 0; q < C::NE; ++q) acc.v[q] += t.v[q];
            ns += ed.n[2 * (2 * k)];
            sb += ed.n[2 * (2 * k) + 1];
            if (!ed.whole[k]) break;
        }
        Row<C> p;
        p.load(P + (int64_t)uu * d, lane, d);
        user_commit<C, ADAM>(P, p_sqnorm, uu, p, acc, ns, reg_1, rU, opt, lane, d);
        user_bias_commit(fm, uu, sb, opt.lr, lane);
    }
}

template <class C, bool ADAM>
__global__ __launch_bounds__(kBlock) void k_staged_user_edges(float *__restrict__ P, int64_t nchunks, int d,
                                                              const double *__restrict__ stats, RowOpt opt,
                                                              float reg_1, float reg_2, UserEdges ed,
                                                              float *__restrict__ p_sqnorm, PreNorm pre,
                                                              ReduceJob red, const double *__restrict__ halt,
                                                              StagedBias fm) {
    if (halted(halt)) return;              // (the riding reduction too: the epoch's sums stay at the offending step)
    user_edges_block<C, ADAM>(blockIdx.x, gridDim.x, P, nchunks, d, stats, opt, reg_1, reg_2, ed, p_sqnorm, pre, red, fm);
}

// The THREE-LAUNCH form of the step (batches up to kMergeMaxBatch samples, where a step is a chain of launches of ~4.5 us
// each and not bytes): the user pass's edge chains + the reduction of its sums, and the item pass, touch disjoint data
// - the chains write rows of P and the norm cache, the item pass reads the stage and Q - once the item workgroups stop
// waiting for the reduced norms: every one of them adds the user pass's per-workgroup sums itself (partials_sums: the
// same loads in the same order as the reduction, so the same bits).  The item launch then carries the chains and the
// reduction as extra workgroups behind its own; the next batch's pre-norm, which needs the chains' rows, moves on to the
// item-edge launch.
struct MergedJob {
    float *P; int64_t u_nchunks; UserEdges ued; float *p_sqnorm; PreNorm pre; ReduceJob red;
    int n_ue;                    // workgroups for the user pass's edge chains (the reduction's workgroup not counted)
};

struct ItemEdges2 {
    float *vec;          // [2*nchunks][d]  partial gradient rows; [2c] head edge (inherited), [2c+1] tail edge
    int32_t *item;       // [2*nchunks]     their item (-1: none)
    float *cnt;          // [2*nchunks][4]  their (n_pos, n_neg, sum of the coefficients (FM), -)
    int32_t *whole;      // [nchunks]       the head edge's segment also runs on into the next chunk
};

// pre-step rows of Q staged in LDS for the commits of one chunk (k_staged_item): rows [first, first + rows)
// (the pointer is an LDS-address-space pointer on purpose: as a generic pointer the commit's row read - window or memory,
// chosen per row - was if-converted into ONE flat_load of a selected address, and a pending FLAT access makes the compiler
// wait for vmcnt(0) AND lgkmcnt(0): every commit then waited for the previous commit's row STORE to land in L2, a full
// trip to memory inside the reduction loop - round 5, profiles/r05_item_pass_counters.txt)
typedef __attribute__((address_space(3))) const float lds_cfloat;
struct QWindow { lds_cfloat *lds; int32_t first, rows; };

// what the owner of a finished segment does with  g = sum_e w_e * stage[slot(e)]  and the entry counts:
//   APPLY:  regulariser reg_1*(np+nn)*sign(q) + reg_2*(np/|Q[i]|_F + nn/|Q[j]|_F)*q  (MFRecommender.py:88-89), then the
//           optimiser's step on Q[item] in place; FM: i_bias[item] follows with the sum of the coefficients
//   else:   gQ[item] = g (data term), cnt[item] = (np, nn): what a multi-GPU step reduce-scatters (FM: g_i_bias[item] =
//           the sum of the coefficients, which the ranks all-reduce)
template <class C, bool APPLY, bool ADAM>
__device__ __forceinline__ void item_commit(float *__restrict__ Qo, float *__restrict__ cnt_out, int64_t item,
                                            const Row<C> &g, float np, float nn, float sb, int lane, int d,
                                            const RowOpt &opt, float reg_1, float rI, float rJ, const StagedBias &fm,
                                            const QWindow win = QWindow{(lds_cfloat *)nullptr, 0, 0},
                                            const Row<C> *qpre = nullptr) {
    if constexpr (APPLY) {
        Row<C> q;
        bool in_lds = false;
        if (qpre) {                                    // (compile-time at every call site) gathered with the stage rows
            q = *qpre;
            in_lds = true;
        } else if constexpr (C::VEC == 4) {
            const uint32_t off = (uint32_t)((int32_t)item - win.first);
            if (off < (uint32_t)win.rows) {            // the row waits in LDS: no dependent trip to memory
                in_lds = true;
                lds_cfloat *src = win.lds + off * (uint32_t)d;
                typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int c = 0; c < C::NV; ++c) {
                    const int e = (c * C::LPR + lane) * 4;
                    v4f t = {0.f, 0.f, 0.f, 0.f};
                    if (C::EXACT || e < d) t = *reinterpret_cast<__attribute__((address_space(3))) const v4f *>(src + e);
                    q.v[c * 4 + 0] = t.x; q.v[c * 4 + 1] = t.y; q.v[c * 4 + 2] = t.z; q.v[c * 4 + 3] = t.w;
                }
            }
        }
        if (!in_lds) {
            q.load(Qo + item * d, lane, d);
            // the wait for THIS load stays inside this branch (a "use" of its registers): behind the join the compiler
            // would wait for vmcnt(0) on every path, i.e. also where the row came from LDS and only STORES are in flight
#pragma unroll
            for (int k = 0; k < C::NE; ++k) asm volatile("" : "+v"(q.v[k]));
        }
        const float w1 = reg_1 * (np + nn), w2 = np * rI + nn * rJ;
        Row<C> gg;
#pragma unroll
        for (int k = 0; k < C::NE; ++k) gg.v[k] = g.v[k] + fmaf(w2, q.v[k], w1 * sgn(q.v[k]));
        row_apply<C, ADAM>(q, gg, opt, item, lane, d);
        q.store(Qo + item * d, lane, d);
        if (fm.bi && lane == 0) {
            if (fm.grad_out) fm.g_bi[item] = sb;
            else fm.bi[item] = fmaf(-opt.lr, sb, fm.bi[item]);
        }
    } else {
        g.store(Qo + item * d, lane, d);
        if (lane == 0) {
            cnt_out[2 * item] = np; cnt_out[2 * item + 1] = nn;
            if (fm.bi) fm.g_bi[item] = sb;       // FM under a multi-GPU step: dL/d i_bias[item] travels like gQ's rows
        }
    }
}

// Item pass over the staged rows (see the header comment): the run/slot/edge segmented reduction of
// k_item_grad_chunked<DET> (bpr_train.hip) with the coefficient inside the gathered row and the commit in
// the owner.  A workgroup takes a chunk of G*RUN consecutive entries, every lane group a run of RUN.
// MODE (see k_staged_user): premul - entry weight +/-1; plain - the sample's (dL/dpos, dL/dneg) gathered from coef;
// point - weight 1 (a negative slot, which only the sorted layout's point-wise batches have, is inert: weight 0, not
// counted).
// (Occupancy floor: four waves per SIMD; two for the Adam and three-launch forms; three for the sparse flavour of rows of
// 16 floats per lane - d > 128 -, whose two Q rows per run do not fit 128 registers.)
// MERGED (three-launch form, see MergedJob; SGD in place only): the launch also carries the user pass's edge chains and
// the reduction of its sums, as workgroups behind the item pass's own, and every item workgroup derives the norms (and
// whether this step's loss is finite) from the user pass's per-workgroup sums itself.
template <class C, int BLK, int MODE, bool APPLY, bool ADAM, bool SPARSE, bool MERGED = false>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu((ADAM || MERGED) ? 2 : ((SPARSE && C::NE > 8) ? 3 : DAISY_ITEM_WAVES), 8))) void k_staged_item(const float *__restrict__ stage,
                                                        const float2 *__restrict__ coef, StreamView v, int d,
                                                        float *__restrict__ Qo, float *__restrict__ cnt_out,
                                                        const double *__restrict__ stats, RowOpt opt,
                                                        float reg_1, float reg_2, ItemEdges2 ed,
                                                        const int64_t *__restrict__ erange, StagedBias fm,
                                                        RideUnorm ride, MergedJob mj) {
    static_assert(!MERGED || (APPLY && !ADAM && BLK == kBlock), "the three-launch form: SGD in place, 256-thread workgroups");
    // this step's loss (reduced behind the user pass) or an earlier one was not finite: nothing may be written.  Requested
    // here, looked at behind the first chunk's gathers (see k_staged_user)
    double halt_word = v.halt ? *v.halt : 0.0;
    if (ride.nblocks && (int)blockIdx.x >= ride.first_block) {      // the next batch's pre-norm rides on this launch
        if (halt_word > 0.0) return;
        unorm_block(ride, (int)blockIdx.x - ride.first_block);
        return;
    }
    int item_grid = ride.nblocks ? ride.first_block : (int)gridDim.x;
    float rI = 0.f, rJ = 0.f;
    if constexpr (MERGED) {
        item_grid = (int)gridDim.x - mj.n_ue - 1;
        if ((int)blockIdx.x >= item_grid) {                         // the user pass's edge chains and its reduction
            if (halt_word > 0.0) return;
            user_edges_block<C, false>(blockIdx.x - item_grid, mj.n_ue + 1, mj.P, mj.u_nchunks, d, stats, opt, reg_1, reg_2,
                                       mj.ued, mj.p_sqnorm, mj.pre, mj.red, fm);
            return;
        }
        // what the reduction workgroup will write to stats, formed here from the same sums in the same order
        const double (*sums)[8] = partials_sums(mj.red.partials, mj.red.nblocks);
        double nU, nI, nJ;
        const double loss = loss_from_sums(&(*sums)[0], reg_1, reg_2, nU, nI, nJ);
        // this step's loss is not finite: the epoch stops here - when the caller gave a halt word at all; a bare step
        // (no epoch accumulator) applies the whole step in the four-launch form, and so must this one: its edge launch
        // and the user pass's edge chains have no word to look at
        if (v.halt && (!(loss == loss) || isinf(loss))) halt_word = 1.0;
        rI = inv_or_zero(nI, reg_2);
        rJ = inv_or_zero(nJ, reg_2);
    }
    using Cfg = StagedItemCfg<C, BLK, SPARSE>;
    constexpr int G = Cfg::G, RUN = Cfg::RUN, E = Cfg::E;
    // erange: only the entries [erange[0], erange[1]) - an item range of the batch (the entries are sorted by item, so
    // no segment crosses the cut and the piece is reduced exactly like a whole batch).  Multi-GPU steps cut the item
    // pass into such slices so that a finished slice can be exchanged while the next one is reduced.
    if (erange) {
        const int64_t lo = erange[0];
        v.e_key += lo * v.e_kstride;
        v.e_pos += lo * v.e_stride;
        v.E = erange[1] - lo;
    }
    constexpr int ROWF = C::NE * C::LPR;
    constexpr bool BIAS = MODE != kModePremul;
    __shared__ int slot_item[G + 1], slot_shared[G + 1];
    __shared__ int run_first[G], run_last[G];
    __shared__ float part_acc[2 * G * ROWF];               // [group][head|tail] parked partial sums
    __shared__ float part_np[2 * G], part_nn[2 * G], part_b[2 * G];
    __shared__ int part_slot[2 * G];                       //      the slot each belongs to (-1: unused)
    // Q rows of the chunk's item range [first item, last item], copied by LDS-DMA while the stage rows are in flight:
    // a commit inside the reduction then costs no dependent trip to memory (at a few entries per item - 10 M x 1 M
    // shapes - 355 -> 320 us per pass).  Only for d % 4 == 0 (16-byte units); rows past the window (sparse batches: few
    // entries spread over many items) are loaded directly.
    constexpr bool WIN = APPLY && C::VEC == 4 && C::NE <= 8 && DAISY_ITEM_WINDOW && !SPARSE && !(DAISY_ITEM_PROBE & 2);
    constexpr bool QPRE = APPLY && SPARSE;                 // the Q row of every entry rides with its staged row
    // (24 KB per 256 threads: the CU's LDS is shared by 256 / BLK times as many workgroups, a chunk covers BLK / 256 of the items)
    constexpr int WINF = ((C::NE <= 4) ? kItemWinFloats : (kItemWinFloats * 2 / 3)) * BLK / kBlock;
    __shared__ __attribute__((aligned(16))) float qwin[WIN ? WINF : 4];
    // sparse flavour: the pre-step row of Q of a segment that leaves its run through the TAIL, for the slot's finisher (a
    // load there would sit behind this chunk's row stores: k_staged_user's part_p has the story)
    __shared__ float part_q[QPRE ? G * ROWF : 4];

    const int tid = threadIdx.x;
    const int lane = tid % C::LPR;
    const int group = tid / C::LPR;
    const int64_t n = v.E;
    const int64_t nchunks = (n + E - 1) / E;
    const bool has_bias = BIAS && fm.bi != nullptr;
    if constexpr (APPLY && !MERGED) {
        rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
        rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    }

    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += item_grid) {
        const int64_t c0 = chunk * E;
        const int64_t t0 = c0 + (int64_t)group * RUN;
        const int64_t t1 = (t0 + RUN < n) ? (t0 + RUN) : n;
        const int cnt = (t0 < n) ? (int)(t1 - t0) : 0;       // entries of this run

        // ---- hop 1: the run's metadata, lane x <- entry x, plus the entries around the run
        const int64_t last = n - 1;
        const int64_t il = (t0 + lane < n) ? (t0 + lane) : last;
        const uint32_t k_me = sv_key(v, il);
#if DAISY_ITEM_PROBE & 4
        const uint32_t slot_me = (uint32_t)(((uint64_t)il * 2654435761ull) % (uint64_t)v.B);
#else
        const uint32_t slot_me = (v.e_pos[il * v.e_stride] & ~kNegBit) - v.pos_base;
#endif
        const uint32_t k_prev = sv_key(v, (t0 > 0) ? ((t0 - 1 < n) ? t0 - 1 : last) : 0);
        const uint32_t k_next = sv_key(v, (t1 < n) ? t1 : last);
        const uint32_t k_cprev = sv_key(v, (c0 > 0) ? c0 - 1 : 0);
        const int32_t my_item = (lane < cnt) ? (int32_t)(k_me >> 1) : -1;
        const int32_t my_neg = (int32_t)(k_me & 1u);
        const int32_t item_prev = (cnt > 0 && t0 > 0) ? (int32_t)(k_prev >> 1) : -1;
        const int32_t item_next = (cnt > 0 && t1 < n) ? (int32_t)(k_next >> 1) : -1;
        // (workgroup-uniform values that come out of a vector load: as scalars they cost no vector register)
        const int32_t chunk_prev_item = (c0 > 0) ? __builtin_amdgcn_readfirstlane((int32_t)(k_cprev >> 1)) : -1;
        QWindow win{(lds_cfloat *)qwin, 0, 0};
        if constexpr (WIN) {
            const int64_t c1 = (c0 + E < n) ? (c0 + E) : n;
            win.first = __builtin_amdgcn_readfirstlane((int32_t)(sv_key(v, c0) >> 1));
            const int32_t item_hi = __builtin_amdgcn_readfirstlane((int32_t)(sv_key(v, c1 - 1) >> 1));
            const int32_t cap = WINF / d;
            win.rows = (item_hi - win.first + 1 < cap) ? (item_hi - win.first + 1) : cap;
            const int units = win.rows * (d >> 2);                      // 16-byte units, contiguous in Q
            const float *src = Qo + (int64_t)win.first * d;
            const int wave_base = __builtin_amdgcn_readfirstlane((tid / kWave) * kWave);
            for (int u0 = 0; u0 < units; u0 += BLK) {                   // (workgroup-uniform trip count)
                const int u = u0 + tid;
                if (u < units)
                    __builtin_amdgcn_global_load_lds(src + 4 * (int64_t)u,
                                                     (__attribute__((address_space(3))) void *)(qwin + 4 * (u0 + wave_base)),
                                                     16, 0, 0);
            }
        }
        if (tid < 2 * G) part_slot[tid] = -1;
        if (tid <= G) { slot_item[tid] = -1; slot_shared[tid] = 0; }
        const int32_t item_first = group_bcast<C>(my_item, 0);
        const int32_t item_last = __shfl(my_item, cnt > 0 ? cnt - 1 : 0, C::LPR);
        if (lane == 0) {
            run_first[group] = cnt > 0 ? item_first : -2;
            run_last[group] = cnt > 0 ? item_last : -2;
        }

        // ---- hop 2: all staged rows of the run (and, for the flavours whose coefficients are not in the row, theirs)
        float my_w, my_c = 0.f;             // weight of the entry's staged row; its bare coefficient (FM's item bias)
        if constexpr (MODE == kModePremul) {
            my_w = (lane < cnt) ? (my_neg ? -1.f : 1.f) : 0.f;
        } else if constexpr (MODE == kModePlain) {
            const float2 c2 = coef[slot_me];
            my_w = (lane < cnt) ? (my_neg ? c2.y : c2.x) : 0.f;
            my_c = my_w;
        } else {
            my_w = (lane < cnt && !my_neg) ? 1.f : 0.f;
            if (has_bias) my_c = (lane < cnt && !my_neg) ? coef[slot_me].x : 0.f;
        }
        Row<C> p[RUN], qe[QPRE ? RUN : 1];
        if (__all(cnt == RUN)) {        // wave-uniform: every run of this wave is full
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                DAISY_STAGE_LOAD(p[x], stage + (int64_t)group_bcast<C>(slot_me, x) * d, lane, d);
                if constexpr (QPRE) qe[x].load(Qo + (int64_t)group_bcast<C>(my_item, x) * d, lane, d);
            }
        } else {
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                const uint32_t sx = group_bcast<C>(slot_me, x);
                const int32_t ix = group_bcast<C>(my_item, x);
                if (x < cnt) {
                    DAISY_STAGE_LOAD(p[x], stage + (int64_t)sx * d, lane, d);
                    if constexpr (QPRE) qe[x].load(Qo + (int64_t)ix * d, lane, d);
                } else {
                    p[x].zero();
                    if constexpr (QPRE) qe[x].zero();
                }
            }
        }
        // the window has landed before anyone reads it.  (Round 4 tried the counted form - the LDS-DMA issued from an asm
        // statement so that the compiler keeps no LDS write pending on the VM counter, s_waitcnt vmcnt(RUN * NV) here,
        // the window read as a proper ds_read_b128 instead of the flat load a generic pointer yields - so that the
        // reduction starts on the first stage rows while the last are in flight: 24 B of scratch at the 128-register
        // cap, and within +-1.5 % of this form at both BASELINE shapes, same box: profiles/r04_item_plan_variants.txt.
        // The pass is not bound by this wait.)
        if constexpr (WIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if DAISY_ITEM_TOUCH
        // (the same for the window: one LDS read that may alias the LDS-DMA's destination makes the compiler's own wait for
        // the DMA happen HERE, once, instead of in front of every commit's window read)
        if constexpr (WIN) {
            float w0 = ((lds_cfloat *)qwin)[tid & 3];
            asm volatile("" : "+v"(w0));
        }
        // every gathered row register is "used" here, so the COMPILER waits for the gathers here and knows they are done:
        // with the wait hidden in the asm statement above it kept all of them pending in its scoreboard, and since the
        // vector-memory counter also counts stores, the first use of a row behind a commit's store (data-dependent
        // control flow, merged states) became s_waitcnt vmcnt(0) - the reduction waited for every row store it issued
#pragma unroll
        for (int x = 0; x < RUN; ++x) {
#pragma unroll
            for (int k = 0; k < C::NE; ++k) asm volatile("" : "+v"(p[x].v[k]));
            if constexpr (QPRE) {
#pragma unroll
                for (int k = 0; k < C::NE; ++k) asm volatile("" : "+v"(qe[x].v[k]));
            }
        }
#endif
        if (halt_word > 0.0) return;           // (uniform over the grid; nothing has been written yet)
#if DAISY_ITEM_PROBE & 1
        {
            float a = (float)(my_item + (int32_t)my_w);
#pragma unroll
            for (int x = 0; x < RUN; ++x)
#pragma unroll
                for (int k = 0; k < C::NE; ++k) a += p[x].v[k];
            if (a == 123.456f) Qo[tid] = a + qwin[tid & 3];
            continue;
        }
#endif
        __syncthreads();

        if (cnt > 0) {
            const bool cont = (t0 > 0) && (item_prev == item_first);
            int cur_slot = -1;                      // >= 0: the current segment began before this run
            if (cont) {
                if (group == 0) cur_slot = 0;       // inherited from the previous chunk
                else {
                    int gs = group - 1;
                    while (gs > 0 && run_first[gs] == item_first && run_last[gs - 1] == item_first) --gs;
                    const bool inherited = (gs == 0) && (run_first[0] == item_first) && (c0 > 0) &&
                                           (chunk_prev_item == item_first);
                    cur_slot = inherited ? 0 : gs + 1;
                }
            }
            int32_t cur_item = item_first;
            Row<C> acc;
            acc.zero();
            float np = 0.f, nn = 0.f, sb = 0.f;
            Row<C> qcur;                            // QPRE: the pre-step row of the current segment's item
            if constexpr (QPRE) qcur = qe[0];
            auto finish = [&](bool ends_here, bool to_next_chunk) {
                if (cur_slot < 0 && ends_here) {    // interior: this group owns the item's row
                    item_commit<C, APPLY, ADAM>(Qo, cnt_out, cur_item, acc, np, nn, sb, lane, d, opt, reg_1, rI, rJ, fm, win,
                                                QPRE ? &qcur : nullptr);
                } else {                            // crosses a run boundary: park it for the slot's finisher
                    const int s = (cur_slot >= 0) ? cur_slot : group + 1;
                    const int q = group * 2 + ((cur_slot >= 0) ? 0 : 1);
                    float *dst = part_acc + q * ROWF;
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) dst[k * C::LPR + lane] = acc.v[k];
                    if constexpr (QPRE) {
                        if (cur_slot < 0) {
                            float *qd = part_q + group * ROWF;
#pragma unroll
                            for (int k = 0; k < C::NE; ++k) qd[k * C::LPR + lane] = qcur.v[k];
                        }
                    }
                    if (lane == 0) {
                        part_slot[q] = s;
                        part_np[q] = np;
                        part_nn[q] = nn;
                        part_b[q] = sb;
                        slot_item[s] = cur_item;
                        if (to_next_chunk) slot_shared[s] = 1;
                    }
                }
            };
#pragma unroll
            for (int x = 0; x < RUN; ++x) {
                if (x < cnt) {
                    const int32_t it = group_bcast<C>(my_item, x);
                    const float wx = group_bcast<C>(my_w, x);
                    const float ng = (float)group_bcast<C>(my_neg, x);
                    if (it != cur_item) {           // previous segment ended inside this run
                        finish(true, false);
                        cur_item = it;
                        cur_slot = -1;
                        acc.zero();
                        np = 0.f; nn = 0.f; sb = 0.f;
                        if constexpr (QPRE) qcur = qe[x];
                    }
#pragma unroll
                    for (int k = 0; k < C::NE; ++k) acc.v[k] = fmaf(wx, p[x].v[k], acc.v[k]);
                    if constexpr (MODE == kModePoint) {
                        np += 1.f - ng;             // (an inert negative slot is not counted)
                    } else {
                        np += 1.f - ng;
                        nn += ng;
                    }
                    if constexpr (BIAS) sb += group_bcast<C>(my_c, x);
                }
            }
            const bool continues = (t1 < n) && (item_next == cur_item);
            finish(!continues, continues && (group == G - 1));
        }
        __syncthreads();

        if (tid == 0) { ed.item[2 * chunk] = -1; ed.item[2 * chunk + 1] = -1; ed.whole[chunk] = 0; }
        __syncthreads();
        // one finisher per used slot (G+1 slots over G groups): the parked partials in group order
        for (int s = group; s <= G; s += G) {
            const int r = slot_item[s];
            if (r < 0) continue;
            Row<C> g;
            g.zero();
            float sp = 0.f, sn = 0.f, sc = 0.f;
            auto add_part = [&](int q) {
                const float *src = part_acc + q * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) g.v[k] += src[k * C::LPR + lane];
                sp += part_np[q];
                sn += part_nn[q];
                sc += part_b[q];
            };
            // the partial sums of slot s, in group order: the TAIL of group s - 1 (the run the segment began in; slot 0
            // came from the previous chunk and has none), then the HEADS of the groups s, s + 1, ... it runs through -
            // consecutive by construction.  (Until round 5 a scan over all 2 G parked slots: 32 dependent LDS reads per
            // finisher where two or three are needed - 1.3 of the ~1.8 us a chunk spends behind its gathers.)
            if (s > 0 && part_slot[2 * s - 1] == s) add_part(2 * s - 1);
            for (int q = 2 * s; q < 2 * G && part_slot[q] == s; q += 2) add_part(q);
            const bool from_prev = (s == 0), to_next = slot_shared[s] != 0;
            if (from_prev || to_next) {
                const int64_t e = 2 * chunk + (from_prev ? 0 : 1);
                g.store(ed.vec + e * d, lane, d);
                if (lane == 0) {
                    ed.item[e] = r;
                    ed.cnt[4 * e] = sp;
                    ed.cnt[4 * e + 1] = sn;
                    ed.cnt[4 * e + 2] = sc;
                    if (from_prev && to_next) ed.whole[chunk] = 1;
                }
            } else if constexpr (QPRE) {             // (s >= 1 here: the tail of group s - 1 parked the pre-step row)
                Row<C> qrow;
                const float *qs = part_q + (s - 1) * ROWF;
#pragma unroll
                for (int k = 0; k < C::NE; ++k) qrow.v[k] = qs[k * C::LPR + lane];
                item_commit<C, APPLY, ADAM>(Qo, cnt_out, r, g, sp, sn, sc, lane, d, opt, reg_1, rI, rJ, fm, win, &qrow);
            } else {
                item_commit<C, APPLY, ADAM>(Qo, cnt_out, r, g, sp, sn, sc, lane, d, opt, reg_1, rI, rJ, fm, win);
            }
        }
        __syncthreads();   // the slots are reused by the next chunk
    }
}

// Long edge chains in two levels (round 5).  A segment that runs through N chunks is a chain of N head edges which
// its owner adds one after the other: ~42 entries per item at BASELINE configs[1] make chains of one or two links, but with
// a Zipf(1.0) popularity the hottest item holds 7 % of a batch's entries - a chain of 2300 links walked by ONE lane group,
// 198 us of a 0.83 ms step (profiles/r05_notes.txt).  Level 1 (this kernel, one lane group per block of kEdgeBlock
// chunks): the sum of the head edges of the chain that ENTERS the block at its first chunk, as far as it runs inside the
// block, and whether it leaves the block at the other end.  Level 2 (k_staged_item_edges): an owner walks to the next
// block boundary link by link and from there block by block.  Fixed order, single writer: reproducible like the chains.
struct EdgeBlocks { float *vec; int32_t *item; float *cnt; int32_t *through; };      // vec == NULL: chains link by link

// How many of flag[first], flag[first + 1], ... (at most maxn) are non-zero before the first zero.  One lane group: its
// lanes load C::LPR flags at a time, the wave ballot hands every group its own bits (a group that has left the loop
// contributes none).  A chain's links used to be found one dependent load after the other - ~1 us per link.
template <class C>
__device__ __forceinline__ int64_t group_leading_ones(const int32_t *__restrict__ flag, int64_t first, int64_t maxn, int lane) {
    constexpr int GPW = kWave / C::LPR;
    const int gw = (threadIdx.x % kWave) / C::LPR;
    int64_t total = 0;
    for (int64_t base = 0; base < maxn; base += C::LPR) {
        const int64_t idx = base + lane;
        const bool f = idx < maxn && flag[first + idx] != 0;
        const uint64_t m = __ballot(f);
        int ones;
        if constexpr (GPW == 1) ones = (m == ~0ull) ? 64 : __builtin_ctzll(~m);
        else ones = __builtin_ctzll(~((m >> (C::LPR * gw)) & ((1ull << C::LPR) - 1)));
        total += ones;
        if (ones < C::LPR) break;
    }
    return total < maxn ? total : maxn;
}

// acc += the `count` records first, first + step, ... of (vec, cnt), in that order, four loads in flight
template <class C>
__device__ __forceinline__ void add_edge_records(Row<C> &acc, float &sp, float &sn, float &sc, const float *__restrict__ vec,
                                                 const float *__restrict__ cnt, int64_t first, int64_t step, int64_t count,
                                                 int lane, int d) {
    if (count == 1) {                       // (the common chain at uniform ids: one link - one load, not four clamped ones)
        Row<C> t;
        t.load(vec + first * d, lane, d);
#pragma unroll
        for (int e = 0; e < C::NE; ++e) acc.v[e] += t.v[e];
        sp += cnt[4 * first]; sn += cnt[4 * first + 1]; sc += cnt[4 * first + 2];
        return;
    }
    for (int64_t j = 0; j < count; j += 4) {
        Row<C> t[4];
        float c0[4], c1[4], c2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = first + ((j + q < count) ? (j + q) : (count - 1)) * step;      // (clamped: no load behind a branch)
            t[q].load(vec + r * d, lane, d);
            c0[q] = cnt[4 * r]; c1[q] = cnt[4 * r + 1]; c2[q] = cnt[4 * r + 2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (j + q < count) {
#pragma unroll
                for (int e = 0; e < C::NE; ++e) acc.v[e] += t[q].v[e];
                sp += c0[q]; sn += c1[q]; sc += c2[q];
            }
        }
    }
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_staged_item_edge_blocks(ItemEdges2 ed, int64_t nchunks, int d,
                                                                    const int64_t *__restrict__ erange, int chunk_entries,
                                                                    const double *__restrict__ halt, EdgeBlocks eb) {
    if (halted(halt)) return;
    if (erange) nchunks = (erange[1] - erange[0] + chunk_entries - 1) / chunk_entries;
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t nblocks = (nchunks + kEdgeBlock - 1) / kEdgeBlock;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t b = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; b < nblocks; b += gstride) {
        const int64_t k0 = b * kEdgeBlock;
        const int64_t k1 = (k0 + kEdgeBlock < nchunks) ? k0 + kEdgeBlock : nchunks;
        const int it = ed.item[2 * k0];
        int through = 0;
        if (it >= 0) {
            // chunk k0's head edge is the chain's; chunk k + 1's is too iff the chain also fills chunk k and runs on
            const int64_t ones = group_leading_ones<C>(ed.whole, k0, k1 - k0, lane);
            const int64_t links = (ones + 1 < k1 - k0) ? ones + 1 : k1 - k0;
            through = (ones == k1 - k0) ? 1 : 0;           // every chunk of the block belongs to the chain, and it runs on
            Row<C> acc;
            acc.zero();
            float sp = 0.f, sn = 0.f, sc = 0.f;
            add_edge_records<C>(acc, sp, sn, sc, ed.vec, ed.cnt, 2 * k0, 2, links, lane, d);
            acc.store(eb.vec + b * d, lane, d);
            if (lane == 0) { eb.cnt[4 * b] = sp; eb.cnt[4 * b + 1] = sn; eb.cnt[4 * b + 2] = sc; }
        }
        if (lane == 0) { eb.item[b] = it; eb.through[b] = through; }
    }
}

// chains of edge records - the chunk whose TAIL edge starts a segment owns it and adds the head edges of
// the chunks it runs through, in chunk order (single writer per row, fixed order)
template <class C, bool APPLY, bool ADAM>
__global__ __launch_bounds__(kBlock) void k_staged_item_edges(ItemEdges2 ed, int64_t nchunks, int d,
                                                              float *__restrict__ Qo, float *__restrict__ cnt_out,
                                                              const double *__restrict__ stats, RowOpt opt,
                                                              float reg_1, float reg_2,
                                                              const int64_t *__restrict__ erange, int chunk_entries,
                                                              const double *__restrict__ halt, StagedBias fm,
                                                              RideReduce rr, RideUnorm ride, EdgeBlocks eb) {
    if (halted(halt)) return;
    if (ride.nblocks && (int)blockIdx.x >= ride.first_block) {     // three-launch form: the next batch's pre-norm rides HERE
        unorm_block(ride, (int)blockIdx.x - ride.first_block);     // (the user pass's edge chains ran in the item launch)
        return;
    }
    unsigned nb = ride.nblocks ? (unsigned)ride.first_block : gridDim.x;
    if (rr.n) {                            // the last workgroup adds the next batch's pre-norm partial sums (k_unorm_reduce)
        nb -= 1;
        if (blockIdx.x == nb) {
            __shared__ double sm[kBlock];
            double t = 0.0;
            for (int b = threadIdx.x; b < rr.n; b += kBlock) t += rr.partials[b];
            sm[threadIdx.x] = t;
            __syncthreads();
            for (int off = kBlock / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0) rr.stats[DAISY_ST_SQ_U_PRE] = sm[0];
            return;
        }
    }
    if (erange) nchunks = (erange[1] - erange[0] + chunk_entries - 1) / chunk_entries;      // chunks of the slice
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)nb * C::GROUPS_PER_BLOCK;
    float rI = 0.f, rJ = 0.f;
    if constexpr (APPLY) {
        rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
        rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    }
    for (int64_t c = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; c < nchunks; c += gstride) {
        const int it = ed.item[2 * c + 1];
        if (it < 0) continue;
        Row<C> acc;
        acc.load(ed.vec + (2 * c + 1) * d, lane, d);
        float sp = ed.cnt[4 * (2 * c + 1)], sn = ed.cnt[4 * (2 * c + 1) + 1], sc = ed.cnt[4 * (2 * c + 1) + 2];
        // link by link - to the end of the chain, or (two levels) to the next block boundary: chunk c + 1's head edge is
        // the chain's (this tail edge runs on into it); chunk k + 1's is iff the chain also fills chunk k
        const int64_t k = c + 1;
        int64_t reach = nchunks;
        if (eb.vec != nullptr) {
            const int64_t bnd = (k + kEdgeBlock - 1) / kEdgeBlock * kEdgeBlock;
            reach = bnd < nchunks ? bnd : nchunks;
        }
        bool open = k < nchunks && ed.item[2 * k] == it;
        if (open && reach > k) {
            const int64_t ones = group_leading_ones<C>(ed.whole, k, reach - k, lane);
            const int64_t links = (ones + 1 < reach - k) ? ones + 1 : reach - k;
            add_edge_records<C>(acc, sp, sn, sc, ed.vec, ed.cnt, 2 * k, 2, links, lane, d);
            open = ones == reach - k;
        }
        // block by block: the block sum of the chain that enters block b is this chain's; it runs on while `through`
        if (eb.vec != nullptr && open && reach < nchunks) {
            const int64_t b0 = reach / kEdgeBlock, nb = (nchunks + kEdgeBlock - 1) / kEdgeBlock;
            if (eb.item[b0] == it) {
                const int64_t ones = group_leading_ones<C>(eb.through, b0, nb - b0, lane);
                const int64_t blocks = (ones + 1 < nb - b0) ? ones + 1 : nb - b0;
                add_edge_records<C>(acc, sp, sn, sc, eb.vec, eb.cnt, b0, 1, blocks, lane, d);
            }
        }
        item_commit<C, APPLY, ADAM>(Qo, cnt_out, it, acc, sp, sn, sc, lane, d, opt, reg_1, rI, rJ, fm);
    }
}

// Q[r] -= lr*(g[r] + regulariser from the GLOBAL entry counts); g[r] = 0, cnt[r] = 0: the owner's share of a
// multi-GPU step after the reduce-scatter of (gQ, cnt)
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_apply_counts(float *__restrict__ Q, float *__restrict__ g,
                                                              float *__restrict__ cnt, int64_t rows, int d,
                                                              float lr, float reg_1, float reg_2,
                                                              const double *__restrict__ stats) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < rows; r += gstride) {
        const float np = cnt[2 * r], nn = cnt[2 * r + 1];
        if (np + nn == 0.f) continue;              // untouched by every rank: the row does not move
        Row<C> gr, z;
        gr.load(g + r * d, lane, d);
        RowOpt opt{};
        opt.lr = lr;
        item_commit<C, true, false>(Q, nullptr, r, gr, np, nn, 0.f, lane, d, opt, reg_1, rI, rJ, StagedBias{});
        z.zero();
        z.store(g + r * d, lane, d);
        if (lane == 0) { cnt[2 * r] = 0.f; cnt[2 * r + 1] = 0.f; }
    }
}

// The same owner step with torch.optim.Adam (multi-GPU staged step): DENSE over the owner's block - every row steps in
// every step like torch's optimiser (rows without an entry: zero gradient, the moments decay), at I/N rows per rank.
template <class C>
__global__ __launch_bounds__(kBlock) void k_item_apply_counts_adam(float *__restrict__ Q, float *__restrict__ g,
                                                                   float *__restrict__ cnt, float *__restrict__ m,
                                                                   float *__restrict__ vv, int64_t rows, int d,
                                                                   float reg_1, float reg_2, RowOpt opt,
                                                                   const double *__restrict__ stats) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const float rI = inv_or_zero(stats[DAISY_ST_NORM_I], reg_2);
    const float rJ = inv_or_zero(stats[DAISY_ST_NORM_J], reg_2);
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < rows; r += gstride) {
        const float np = cnt[2 * r], nn = cnt[2 * r + 1];
        Row<C> q, gr, mr, vr;
        q.load(Q + r * d, lane, d);
        mr.load(m + r * d, lane, d);
        vr.load(vv + r * d, lane, d);
        gr.zero();
        if (np + nn != 0.f) {
            gr.load(g + r * d, lane, d);
            const float w1 = reg_1 * (np + nn), w2 = np * rI + nn * rJ;
#pragma unroll
            for (int k = 0; k < C::NE; ++k) gr.v[k] += fmaf(w2, q.v[k], w1 * sgn(q.v[k]));
        }
        adam_row<C>(q, mr, vr, gr, opt.step_size, opt.bc2_sqrt, opt.beta1, opt.beta2, opt.eps);
        q.store(Q + r * d, lane, d);
        mr.store(m + r * d, lane, d);
        vr.store(vv + r * d, lane, d);
        if (np + nn != 0.f) {
            Row<C> z;
            z.zero();
            z.store(g + r * d, lane, d);
            if (lane == 0) { cnt[2 * r] = 0.f; cnt[2 * r + 1] = 0.f; }
        }
    }
}

// rng[s] = first entry of the batch whose item is >= bounds.b[s]  (s = 0..S; the entries are sorted by item)
struct SliceBounds { int32_t b[kMaxItemSlices + 1]; };
__global__ void k_slice_ranges(StreamView v, SliceBounds bounds, int S, int64_t *__restrict__ rng) {
    const int s = threadIdx.x;
    if (s > S) return;
    int64_t lo = 0, hi = v.E;
    const uint32_t want = (uint32_t)bounds.b[s];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((sv_key(v, mid) >> 1) < want) lo = mid + 1; else hi = mid;
    }
    rng[s] = lo;
}

// ---------------------------------------------------------------------------------------------------------
// host side of the staged step
// ---------------------------------------------------------------------------------------------------------
// the staged step covers every loss of the reference (loss.py:5-33, MFRecommender.py:70-97): the point-wise ones
// need a point-wise batch (rows (user, item, label)) and vice versa; FM biases ride on the plain / point flavours
bool staged_supported(const daisy_bpr_ctx *ctx, int loss_type) {
    if (loss_type < DAISY_LOSS_BPR || loss_type > DAISY_LOSS_SL) return false;
    return (loss_type >= DAISY_LOSS_CL) == (ctx->sv.pointwise != 0);
}

static int staged_check(daisy_bpr_ctx *ctx, int loss_type, const char *who) {
    if (!ctx->batch_set) { set_error("%s: no batch set", who); return DAISY_ERR_STATE; }
    if (loss_type < DAISY_LOSS_BPR || loss_type > DAISY_LOSS_SL) {
        set_error("Invalid loss type: %d", loss_type);
        return DAISY_ERR_ARG;
    }
    if (!staged_supported(ctx, loss_type)) {
        set_error("%s: loss type %d does not match the batch layout (point-wise=%d)", who, loss_type, ctx->sv.pointwise);
        return DAISY_ERR_ARG;
    }
    return DAISY_OK;
}

// reduce: also k_unorm_reduce -> stats[SQ_U_PRE] (the phase API); else the consumers add the partial sums themselves
static int staged_prenorm(daisy_bpr_ctx *ctx, const float *P, double *stats, bool reduce, int *n_pre, hipStream_t s) {
    const StreamView &v = ctx->sv;
    const int d = ctx->d;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (ctx->p_sqnorm_of != P) {   // (re)build the squared-norm cache: one dense pass over P
            hipLaunchKernelGGL((k_row_sqnorm<C>), dim3(grid_for(ctx->U, C::GROUPS_PER_BLOCK * 4)), dim3(kBlock), 0, s,
                               P, ctx->U, d, ctx->p_sqnorm);
            ctx->p_sqnorm_of = P;
        }
        return DAISY_OK;
    });
    if (rc) return rc;
    // Batches up to kPreBlocks workgroups of pre-norm work: the consumers add the partial sums themselves (one launch
    // less; what bounds a step of that size).  Larger ones: a full grid and k_unorm_reduce - every workgroup of the
    // user pass re-adding thousands of partial sums measured slower than the launch it saves.
    int gn = grid_for(v.B, kBlock * 4);
    const bool fold = !reduce && gn <= kPreBlocks / 2;
    double *pre = fold ? ctx->partials + (size_t)kMaxGrid * 8 : ctx->partials;   // fold: behind the user pass's own sums
    hipLaunchKernelGGL(k_unorm, dim3(gn), dim3(kBlock), 0, s, ctx->p_sqnorm, v, pre);
    if (!fold) hipLaunchKernelGGL(k_unorm_reduce, dim3(1), dim3(kBlock), 0, s, pre, gn, stats);
    if (n_pre) *n_pre = fold ? gn : 0;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

static inline bool premul_loss(int loss_type) { return loss_type == DAISY_LOSS_BPR || loss_type == DAISY_LOSS_HL; }

// flavour of the two passes (k_staged_user): FM's pairwise runs take the plain one (its item biases need the bare
// coefficients)
static int staged_mode(const daisy_bpr_ctx *ctx, int loss_type) {
    if (loss_type >= DAISY_LOSS_CL) return kModePoint;
    return (premul_loss(loss_type) && !ctx->bu) ? kModePremul : kModePlain;
}

static StagedBias staged_bias(const daisy_bpr_ctx *ctx, bool grad_out) {
    StagedBias fm{ctx->bu, ctx->bi, ctx->b0, ctx->g_bu, ctx->g_bi, grad_out ? 1 : 0};
    return fm;
}

// optimiser of one table's row owners; adam == nullptr: SGD
struct StagedAdam {           // the lazy Adam state of both tables for step t (daisy_bpr_staged_adam_step)
    float *mP, *vP; int32_t *lastP;
    float *mQ, *vQ; int32_t *lastQ;
    float step_size, bc2_sqrt, beta1, beta2, eps;
    int32_t t;
};
static RowOpt row_opt(float lr, const StagedAdam *a, bool user_side) {
    RowOpt o{};
    o.lr = lr;
    if (a) {
        o.m = user_side ? a->mP : a->mQ;
        o.v = user_side ? a->vP : a->vQ;
        o.last = user_side ? a->lastP : a->lastQ;
        o.step_size = a->step_size; o.bc2_sqrt = a->bc2_sqrt;
        o.beta1 = a->beta1; o.beta2 = a->beta2; o.eps = a->eps;
        o.t = a->t;
    }
    return o;
}

static StagedAdam adam_consts(float lr, float beta1, float beta2, float eps, int64_t step) {
    StagedAdam a{};
    // step `step`'s constants: the host arithmetic of daisy_adam_dense / daisy_adam_lazy_table (same bits as table[step])
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.t = (int32_t)step;
    return a;
}

// workgroup size of the user pass: 128 threads; ONE wave (its barriers are free) for rows of two float4 per
// lane (64 < d <= 128: 1.40 -> 1.27 ms per 2M-sample step at d=128; slower at d <= 64 (0.60 -> 0.71), and
// rows of four float4 per lane would need more edge records than the context holds)
template <class C>
struct StagedUserBlk {
    static constexpr int value = (C::NE == 8 && C::LPR == 16) ? 64 : ((C::LPR <= 32) ? kStagedUserBlock : kBlock);
};

// n_pre > 0: the pre-norm comes from k_unorm's partial sums, else from stats[SQ_U_PRE].  ride_reduce (single-GPU
// step): the reduction of this pass's own sums (stats, loss; into epoch_acc / step_loss) rides on the edge
// launch; else the caller reduces
static int staged_user(daisy_bpr_ctx *ctx, float *P, const float *Q, int loss_type, float gamma, float lr,
                       float reg_1, float reg_2, double *stats, int *grid_out, int n_pre, bool ride_reduce,
                       double *epoch_acc, double *step_loss, hipStream_t s, const StagedAdam *adam = nullptr,
                       bool bias_grad_out = false, MergedJob *hand_over = nullptr) {
    // hand_over != NULL (three-launch form): the edge launch is NOT issued; *hand_over describes its work for the item launch
    ctx->pre_ready = false;                // (P is about to change: a pre-norm computed ahead of this pass is stale)
    StreamView v = ctx->sv;
    const int d = ctx->d;
    const int mode = staged_mode(ctx, loss_type);
    const bool has_pos = v.s_rec != nullptr;       // (the partitioned plan: the slot comes with the record)
    v.p_stream = (ctx->p_stream_mode >= 0) ? (ctx->p_stream_mode != 0)
                                           : ((size_t)ctx->U * (size_t)d * 4 > kStreamTableBytes ? 1 : 0);
    UserEdges ed{ctx->edge_vec, ctx->edge_user, ctx->edge_n, ctx->edge_whole};
    const PreNorm pre{ctx->partials + (size_t)kMaxGrid * 8, n_pre};
    const RowOpt opt = row_opt(lr, adam, true);
    const StagedBias fm = staged_bias(ctx, bias_grad_out);
    static const int tune_ug = getenv("DAISY_STAGED_UGRID") ? atoi(getenv("DAISY_STAGED_UGRID")) : kMaxGrid;
    int64_t nchunks_out = 0;
    bool overflow = false;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        constexpr int BLK = StagedUserBlk<C>::value;
        const int64_t nchunks = (v.B + StagedUserCfg<C, BLK>::E - 1) / StagedUserCfg<C, BLK>::E;
        if (nchunks > ctx->edge_chunks) { overflow = true; return DAISY_OK; }
        const int gu = grid_for(nchunks, 1, tune_ug < kMaxGrid ? tune_ug : kMaxGrid);
        *grid_out = gu;
        nchunks_out = nchunks;
        auto launch = [&](auto mode_tag, auto hp_tag, auto adam_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool HP = decltype(hp_tag)::value, AD = decltype(adam_tag)::value;
            hipLaunchKernelGGL((k_staged_user<C, BLK, MODE, HP, AD>), dim3(gu), dim3(BLK), 0, s, P, Q, v, d, stats, opt,
                               reg_1, reg_2, loss_type, gamma, ctx->p_stage, ctx->coef, ctx->p_sqnorm, ctx->partials, ed,
                               pre, fm);
        };
        auto by_adam = [&](auto mode_tag, auto hp_tag) {
            if (adam) launch(mode_tag, hp_tag, std::true_type{});
            else launch(mode_tag, hp_tag, std::false_type{});
        };
        auto by_pos = [&](auto mode_tag) {
            if (has_pos) by_adam(mode_tag, std::true_type{});
            else by_adam(mode_tag, std::false_type{});
        };
        if (mode == kModePremul) by_pos(std::integral_constant<int, kModePremul>{});
        else if (mode == kModePlain) by_pos(std::integral_constant<int, kModePlain>{});
        else by_pos(std::integral_constant<int, kModePoint>{});
        const ReduceJob red{ctx->partials, ride_reduce ? *grid_out : 0, stats, epoch_acc, step_loss};
        const dim3 ge(grid_for(nchunks_out, C::GROUPS_PER_BLOCK) + (ride_reduce ? 1 : 0));
        if (hand_over) {
            *hand_over = MergedJob{P, nchunks_out, ed, ctx->p_sqnorm, pre, red, grid_for(nchunks_out, C::GROUPS_PER_BLOCK)};
            return DAISY_OK;
        }
        if (adam)
            hipLaunchKernelGGL((k_staged_user_edges<C, true>), ge, dim3(kBlock), 0, s, P, nchunks_out, d, stats, opt, reg_1,
                               reg_2, ed, ctx->p_sqnorm, pre, red, v.halt, fm);
        else
            hipLaunchKernelGGL((k_staged_user_edges<C, false>), ge, dim3(kBlock), 0, s, P, nchunks_out, d, stats, opt, reg_1,
                               reg_2, ed, ctx->p_sqnorm, pre, red, v.halt, fm);
        return DAISY_OK;
    });
    if (rc) return rc;
    if (overflow) { set_error("staged step: the batch needs more edge records than the context holds"); return DAISY_ERR_STATE; }
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

// the user pass's edge launch from a MergedJob that was not merged after all (SGD)
static int staged_user_edges_launch(daisy_bpr_ctx *ctx, const MergedJob &mj, float lr, float reg_1, float reg_2,
                                    const double *stats, hipStream_t s) {
    const RowOpt opt = row_opt(lr, nullptr, true);
    const StagedBias fm = staged_bias(ctx, false);
    int rc = dispatch_d(ctx->d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_staged_user_edges<C, false>), dim3(mj.n_ue + (mj.red.nblocks > 0 ? 1 : 0)), dim3(kBlock), 0, s, mj.P,
                           mj.u_nchunks, ctx->d, stats, opt, reg_1, reg_2, mj.ued, mj.p_sqnorm, mj.pre, mj.red, ctx->sv.halt, fm);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

// apply != 0: Qo = Q, updated in place; else Qo = gQ (data term), cnt_out f32[I][2]
// slice >= 0: only the entries of item slice `slice` (daisy_bpr_staged_item_slices has run for this batch)
static int staged_item(daisy_bpr_ctx *ctx, int loss_type, float *Qo, float *cnt_out, bool apply, float lr,
                       float reg_1, float reg_2, const double *stats, hipStream_t s, int slice = -1,
                       const StagedAdam *adam = nullptr, bool bias_grad_out = false,
                       RideUnorm ride = RideUnorm{nullptr, nullptr, 0, nullptr, 0, 0},
                       RideReduce rr = RideReduce{nullptr, 0, nullptr}, const MergedJob *merged = nullptr) {
    // merged != NULL (three-launch form; apply, SGD): the item launch carries the user pass's edge work, the item pass
    // keeps its edge records in the context's second set, and `ride` rides on the item-EDGE launch
    const int64_t *erange = (slice >= 0) ? ctx->slice_rng + slice : nullptr;
    const StreamView &v = ctx->sv;
    const int d = ctx->d;
    const int mode = staged_mode(ctx, loss_type);
    ItemEdges2 ed{ctx->edge_vec, ctx->edge_user, ctx->edge_cnt, ctx->edge_whole};
    if (merged) ed = ItemEdges2{ctx->edge2_vec, ctx->edge2_item, ctx->edge2_cnt, ctx->edge2_whole};
    const int64_t edge_cap = merged ? ctx->edge2_chunks : ctx->edge_chunks;
    const RowOpt opt = row_opt(lr, adam, false);
    const StagedBias fm = staged_bias(ctx, bias_grad_out);
    if (adam && !apply) { set_error("staged item pass: the Adam owner update needs the in-place form"); return DAISY_ERR_ARG; }
    static const int tune_ig = getenv("DAISY_STAGED_IGRID") ? atoi(getenv("DAISY_STAGED_IGRID")) : 0;    // 0: 16 384 x 256 threads
    // DAISY_STAGED_SPARSE (read per call: the tests switch it): 0 / 1 force the dense / sparse flavour of the SGD item pass
    const char *env_sparse = getenv("DAISY_STAGED_SPARSE");
    const int tune_sparse = env_sparse ? atoi(env_sparse) : -1;
    bool overflow = false;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
      auto with_block = [&](auto blk_tag) {
        constexpr int BLK = decltype(blk_tag)::value;        // (128-thread workgroups measured the same, r02 / r03)
        // the sparse flavour (StagedItemCfg) when a dense chunk's item range would not fit the LDS window: expected
        // items under a chunk = chunk entries x items / entries  >  rows of the window
        constexpr int E_dense = StagedItemCfg<C, BLK>::E, E_sparse = StagedItemCfg<C, BLK, true>::E;
        constexpr int win_rows = (kItemWinFloats * BLK / kBlock) / (C::NE * C::LPR);
        bool sparse = apply && !adam && (double)v.E * win_rows < (double)E_dense * (double)ctx->I;
        if (tune_sparse >= 0) sparse = apply && !adam && tune_sparse != 0;
        if (sparse && (v.E + E_sparse - 1) / E_sparse > edge_cap) {
            // (cannot happen with the context's own sizing, which counts StagedItemCfg<C, 128, true>::E; a build with
            // another DAISY_ITEM_BLK / run length says so once instead of silently running the slower flavour)
            static bool warned = false;
            if (!warned) { warned = true; fprintf(stderr, "daisyrec: staged item pass: %lld sparse chunks exceed the %lld edge records of the context - dense flavour\n", (long long)((v.E + E_sparse - 1) / E_sparse), (long long)edge_cap); }
            sparse = false;
        }
        const int chunk_e = sparse ? E_sparse : E_dense;
        const int64_t nchunks = (v.E + chunk_e - 1) / chunk_e;
        if (nchunks > edge_cap) { overflow = true; return DAISY_OK; }
        const int gi = grid_for(v.E, chunk_e, tune_ig > 0 ? tune_ig : 16384 * kBlock / BLK);
        // Edge chains in two levels when the batch can hold a segment of 16 chunks or more: the hottest item's share of
        // the set (the index counted it) x the batch's entries, with room for its fluctuation; unknown - a batch that did
        // not come from a partitioned plan - means a large batch.  DAISY_EDGE_BLOCKS (read per call): 0 never, 1 always.
        const daisy_epoch_plan *pl_ = (ctx->batch_kind == 1) ? ctx->cur_plan : nullptr;
        const double hot = pl_ ? pl_->hot_item_share * (double)v.E : -1.0;
        const char *env_eb = getenv("DAISY_EDGE_BLOCKS");
        bool two_level = !merged && nchunks / kEdgeBlock + 2 <= ctx->eb_blocks &&
                         (hot >= 0.0 ? (hot + 4.0 * sqrt(hot) >= 16.0 * chunk_e) : nchunks >= 64 * kEdgeBlock);
        if (env_eb) two_level = atoi(env_eb) != 0 && !merged && nchunks / kEdgeBlock + 2 <= ctx->eb_blocks;
        const EdgeBlocks eb = two_level ? EdgeBlocks{ctx->eb_vec, ctx->eb_item, ctx->eb_cnt, ctx->eb_through}
                                        : EdgeBlocks{nullptr, nullptr, nullptr, nullptr};
        const int ge_own = grid_for(nchunks, C::GROUPS_PER_BLOCK) + (rr.n ? 1 : 0);
        RideUnorm ru = ride, ru_edge = RideUnorm{nullptr, nullptr, 0, nullptr, 0, 0};
        MergedJob mj{};
        if (merged) {                        // the ride moves to the edge launch; the user pass's edge work joins this one
            ru_edge = ride;
            ru_edge.first_block = ge_own;
            ru.nblocks = 0;
            mj = *merged;
        }
        ru.first_block = gi;                 // the riding workgroups follow the item pass's own
        const dim3 g(gi + ru.nblocks + (merged ? mj.n_ue + 1 : 0)), b(BLK), ge(ge_own + ru_edge.nblocks);
        auto launch = [&](auto mode_tag, auto ap_tag, auto adam_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool AP = decltype(ap_tag)::value, AD = decltype(adam_tag)::value;
            if constexpr (AP && !AD) {
#define DAISY_ITEM_LAUNCH(SP, MG)                                                                                          \
    hipLaunchKernelGGL((k_staged_item<C, BLK, MODE, true, false, SP, MG>), g, b, 0, s, ctx->p_stage, ctx->coef, v, d, Qo, \
                       cnt_out, stats, opt, reg_1, reg_2, ed, erange, fm, ru, mj)
                if constexpr (BLK == kBlock) {
                    if (merged) { if (sparse) DAISY_ITEM_LAUNCH(true, true); else DAISY_ITEM_LAUNCH(false, true); }
                }
                if (!merged) { if (sparse) DAISY_ITEM_LAUNCH(true, false); else DAISY_ITEM_LAUNCH(false, false); }
#undef DAISY_ITEM_LAUNCH
            } else {
                hipLaunchKernelGGL((k_staged_item<C, BLK, MODE, AP, AD, false>), g, b, 0, s, ctx->p_stage, ctx->coef, v, d, Qo,
                                   cnt_out, stats, opt, reg_1, reg_2, ed, erange, fm, ru, mj);
            }
            if (two_level)
                hipLaunchKernelGGL((k_staged_item_edge_blocks<C>),
                                   dim3(grid_for((nchunks + kEdgeBlock - 1) / kEdgeBlock, C::GROUPS_PER_BLOCK)), dim3(kBlock), 0,
                                   s, ed, nchunks, d, erange, chunk_e, v.halt, eb);
            hipLaunchKernelGGL((k_staged_item_edges<C, AP, AD>), ge, dim3(kBlock), 0, s, ed, nchunks, d, Qo, cnt_out, stats,
                               opt, reg_1, reg_2, erange, chunk_e, v.halt, fm, rr, ru_edge, eb);
        };
        auto by_apply = [&](auto mode_tag) {
            if (apply && adam) launch(mode_tag, std::true_type{}, std::true_type{});
            else if (apply) launch(mode_tag, std::true_type{}, std::false_type{});
            else launch(mode_tag, std::false_type{}, std::false_type{});
        };
        if (mode == kModePremul) by_apply(std::integral_constant<int, kModePremul>{});
        else if (mode == kModePlain) by_apply(std::integral_constant<int, kModePlain>{});
        else by_apply(std::integral_constant<int, kModePoint>{});
        return DAISY_OK;
      };
        // (the three-launch form is written for 256-thread workgroups)
        if (merged) return with_block(std::integral_constant<int, kBlock>{});
        return with_block(std::integral_constant<int, kStagedItemBlock>{});
    });
    if (rc) return rc;
    if (overflow) { set_error("staged step: the batch needs more edge records than the context holds"); return DAISY_ERR_STATE; }
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int staged_sgd_step(daisy_bpr_ctx *ctx, float *P, float *Q, int loss_type, float gamma, float lr, float reg_1,
                    float reg_2, double *stats, double *epoch_acc, double *step_loss, hipStream_t s) {
    int rc = staged_check(ctx, loss_type, "sgd_step");
    if (rc) return rc;
    // four launches inside an epoch: user pass, its edges (+ the reduction of its sums), item pass (+ the NEXT batch's
    // pre-norm partial sums), its edges (+ their reduction); the first step of an epoch computes its own pre-norm
    // first (k_unorm, and k_unorm_reduce for large batches)
    static const int tune_ride = getenv("DAISY_STAGED_RIDE") ? atoi(getenv("DAISY_STAGED_RIDE")) : 1;
    int gu = 0, n_pre = 0;
    const bool have_pre = ctx->pre_ready && ctx->cur_plan && ctx->pre_plan == ctx->cur_plan && ctx->pre_k == ctx->cur_k &&
                          ctx->pre_gen == ctx->cur_gen && ctx->cur_gen == ctx->cur_plan->build_gen && ctx->pre_P == P &&
                          ctx->p_sqnorm_of == P && (ctx->pre_n > 0 || ctx->pre_stats == stats);
    ctx->pre_ready = false;                  // consumed, or stale
    if (have_pre) n_pre = ctx->pre_n;        // > 0: partial sums wait behind the user pass's own; 0: stats[SQ_U_PRE] holds the sum
    else if ((rc = staged_prenorm(ctx, P, stats, false, &n_pre, s))) return rc;
    // THREE launches (MergedJob) for batches whose user pass leaves at most kMergeMaxPartials rows of partial sums (every
    // item workgroup adds them itself): B <= 32768 at d = 64.  DAISY_STAGED_MERGE (read per call: the tests switch it):
    // 0 never, 1 whenever the buffers allow it
    const char *env_merge = getenv("DAISY_STAGED_MERGE");
    const int tune_merge = env_merge ? atoi(env_merge) : -1;
    // (round 5, same box, four vs three launches: B = 32 768 - 1024 rows of partial sums - 36.9 -> 33.4 us per step; B =
    // 65 536 - 2048 rows - 51.6 -> 58.6, and with the user pass capped at 1024 workgroups a tie: profiles/r05_mid_matrix.txt)
    constexpr int64_t kMergeMaxPartials = 1024;
    const daisy_epoch_plan *pl = ctx->cur_plan;
    const bool ride_next = tune_ride && pl && pl->kind == 1 && ctx->cur_gen == pl->build_gen &&
                           ctx->cur_k + 1 < pl->num_batches && daisy_epoch_plan_batch_rows(pl, ctx->cur_k + 1) > 0;
    bool merge = tune_merge != 0 && ctx->edge2_vec != nullptr && ctx->sv.B <= kMergeMaxBatch &&
                 (!ride_next || grid_for(plan_stream_view(pl, ctx->cur_k + 1).B, kBlock * 4) <= kPreBlocks / 2);
    MergedJob mj{};
    if ((rc = staged_user(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, stats, &gu, n_pre, true, epoch_acc, step_loss, s,
                          nullptr, false, merge ? &mj : nullptr))) return rc;
    if (merge && tune_merge < 0 && gu > kMergeMaxPartials) merge = false;
    if (!merge && mj.P != nullptr) {         // (decided against it after all: issue the edge launch that was held back)
        if ((rc = staged_user_edges_launch(ctx, mj, lr, reg_1, reg_2, stats, s))) return rc;
    }
    // the next batch of the same plan: its pre-norm rides on this step's item pass (three-launch form: on the item-edge launch)
    RideUnorm ride{nullptr, nullptr, 0, nullptr, 0, 0};
    RideReduce rr{nullptr, 0, nullptr};
    if (ride_next) {
        const StreamView nv = plan_stream_view(pl, ctx->cur_k + 1);
        const int gn = grid_for(nv.B, kBlock * 4);
        const bool fold = gn <= kPreBlocks / 2;              // (the same rule as staged_prenorm)
        ride.p_sqnorm = ctx->p_sqnorm; ride.s_rec = nv.s_rec; ride.B = nv.B;
        ride.partials = fold ? ctx->partials + (size_t)kMaxGrid * 8 : ctx->partials;
        ride.nblocks = gn;
        if (!fold) { rr.partials = ride.partials; rr.n = gn; rr.stats = stats; }
        ctx->pre_plan = pl; ctx->pre_k = ctx->cur_k + 1; ctx->pre_gen = ctx->cur_gen;
        ctx->pre_P = P; ctx->pre_stats = stats; ctx->pre_n = fold ? gn : 0;
    }
    if ((rc = staged_item(ctx, loss_type, Q, nullptr, true, lr, reg_1, reg_2, stats, s, -1, nullptr, false, ride, rr,
                          merge ? &mj : nullptr))) return rc;
    ctx->pre_ready = ride.nblocks > 0;
    ctx->fwd_done = false;
    return DAISY_OK;
}

// ---- lazy Adam on the staged step ------------------------------------------------------------------------
// Rows a batch references are brought to step t-1 before its forward pass (the zero-gradient replay of
// adam_claim_row, bpr_train.hip).  Samples are grouped by user and entries sorted by item, so the first record of a
// run of equal ids is the row's single owner: no atomics.  The user side also refreshes the row-norm cache.
template <class C>
__global__ __launch_bounds__(kBlock) void k_staged_adam_catchup(StreamView v, int d, float *__restrict__ P,
                                                                float *__restrict__ Q, StagedAdam a,
                                                                const float2 *__restrict__ table,
                                                                float *__restrict__ p_sqnorm, int users_only) {
    if (halted(v.halt)) return;
    const int lane = threadIdx.x % C::LPR, group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t total = users_only ? v.B : v.B + v.E;
    for (int64_t x = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; x < total; x += gstride) {
        const bool user_side = x < v.B;
        int64_t row;
        if (user_side) {
            row = (int64_t)sv_user(v, x);
            if (x > 0 && (int64_t)sv_user(v, x - 1) == row) continue;
        } else {
            const int64_t e = x - v.B;
            row = (int64_t)(sv_key(v, e) >> 1);
            if (e > 0 && (int64_t)(sv_key(v, e - 1) >> 1) == row) continue;
        }
        float *W = user_side ? P : Q, *M = user_side ? a.mP : a.mQ, *V = user_side ? a.vP : a.vQ;
        int32_t *last = user_side ? a.lastP : a.lastQ;
        const int32_t old = last[row];
        if (old >= a.t - 1) continue;
        Row<C> w, m, vv, g;
        w.load(W + row * d, lane, d); m.load(M + row * d, lane, d); vv.load(V + row * d, lane, d);
        g.zero();
        for (int32_t st = old + 1; st < a.t; ++st) {                  // zero-gradient steps old+1 .. t-1
            const float2 c = table[st];
            adam_row<C>(w, m, vv, g, c.x, c.y, a.beta1, a.beta2, a.eps);
        }
        // (Round 4, measured and rejected - profiles/r04_bench_adam_fused_catchup_rejected.txt, r04_adam_prefetch_ab.txt:
        // NOT writing the users back and replaying the same steps in the user pass, which then gathers each sample's
        // moments - three row transfers per stale user less, but the replay is ~60 VALU cycles per element and step
        // (sqrt, two divides) and doing it twice cost more than the traffic: 2.80 -> 3.62 ms per step at 10 M x 1 M.
        // Gathering the moments with the rows WITHOUT the replay, so that a run's owner commits from registers: no
        // change either way (1.293 / 1.291 and 3.032 / 3.039 ms, same box) at 180 instead of 161 VGPRs.)
        w.store(W + row * d, lane, d); m.store(M + row * d, lane, d); vv.store(V + row * d, lane, d);
        if (lane == 0) last[row] = a.t - 1;
        if (user_side) {
            const float sq = row_dot<C>(w, w);
            if (lane == 0) p_sqnorm[row] = sq;
        }
    }
}

int staged_adam_step(daisy_bpr_ctx *ctx, float *P, float *Q, int loss_type, float gamma, float reg_1, float reg_2,
                     const StagedAdam &a, const float *table, bool bias_grad_out, double *stats, double *epoch_acc,
                     double *step_loss, hipStream_t s) {
    int rc = staged_check(ctx, loss_type, "staged_adam_step");
    if (rc) return rc;
    const StreamView &v = ctx->sv;
    const int d = ctx->d;
    // the row-norm cache has to exist before the catch-up refreshes the rows it changes
    rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (ctx->p_sqnorm_of != P) {
            hipLaunchKernelGGL((k_row_sqnorm<C>), dim3(grid_for(ctx->U, C::GROUPS_PER_BLOCK * 4)), dim3(kBlock), 0, s,
                               P, ctx->U, d, ctx->p_sqnorm);
            ctx->p_sqnorm_of = P;
        }
        hipLaunchKernelGGL((k_staged_adam_catchup<C>), dim3(grid_for(v.B + v.E, C::GROUPS_PER_BLOCK, kMaxGridSparse)),
                           dim3(kBlock), 0, s, v, d, P, Q, a, reinterpret_cast<const float2 *>(table), ctx->p_sqnorm, 0);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    int gu = 0, n_pre = 0;
    if ((rc = staged_prenorm(ctx, P, stats, false, &n_pre, s))) return rc;
    if ((rc = staged_user(ctx, P, Q, loss_type, gamma, 0.f, reg_1, reg_2, stats, &gu, n_pre, true, epoch_acc, step_loss, s,
                          &a, bias_grad_out))) return rc;
    if ((rc = staged_item(ctx, loss_type, Q, nullptr, true, 0.f, reg_1, reg_2, stats, s, -1, &a, bias_grad_out))) return rc;
    ctx->fwd_done = false;
    return DAISY_OK;
}

}  // namespace daisy

using namespace daisy;

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int daisy_train_index_create(daisy_train_index **out, const int32_t *triples, int64_t n_triples, int64_t user_num,
                             int64_t item_num, int32_t user_base, int32_t flags, daisy_stream_t stream) {
    DAISY_CHECK_ARG(out && triples, "train_index_create: NULL argument");
    DAISY_CHECK_ARG(n_triples > 0 && n_triples < ((int64_t)1 << 30), "train_index_create: n_triples=%lld out of range",
                    (long long)n_triples);
    DAISY_CHECK_ARG(user_num > 0 && user_num <= INT32_MAX && item_num > 0 && item_num < ((int64_t)1 << 30),
                    "train_index_create: user_num/item_num out of range");
    hipStream_t s = S(stream);
    const int64_t n = n_triples;
    const bool sorted = (flags & DAISY_PLAN_TRIPLES_USER_SORTED) != 0;
    const int pointwise = (flags & DAISY_PLAN_POINTWISE) ? 1 : 0;
    daisy_train_index *ix = new daisy_train_index();
    memset(ix, 0, sizeof(*ix));
    ix->n = n; ix->U = user_num; ix->I = item_num; ix->user_base = user_base;
    ix->pointwise = pointwise;
    ix->n_ent = pointwise ? n : 2 * n;
    // scratch: unsorted entry pairs [2n] x2, (unsorted user pairs [n] x2 + sorted pairs [n] x2), bad flag, sort temp
    const size_t t_sort = sort_pairs_i32_temp_bytes(2 * n);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_k = take((size_t)n * 8), o_v = take((size_t)n * 8);
    const size_t o_uk = take((size_t)n * 4), o_uv = take((size_t)n * 4), o_uk2 = take((size_t)n * 4);
    const size_t o_bad = take(256), o_tmp = take(t_sort), o_hist = take((size_t)item_num * 8);
    char *scratch = nullptr;
    void *keep = nullptr;
    const size_t keep_bytes = align_up((size_t)n * 8) * 2 + (sorted ? 0 : align_up((size_t)n * 12) + align_up((size_t)n * 4));
    hipError_t e = hipMalloc((void **)&scratch, off);
    if (e == hipSuccess) e = hipMalloc(&keep, keep_bytes);
    if (e != hipSuccess) {
        set_error("train_index_create: hipMalloc failed: %s", hipGetErrorString(e));
        if (scratch) (void)hipFree(scratch);
        delete ix;
        return DAISY_ERR_HIP;
    }
    ix->ent_t = (uint32_t *)keep;
    ix->ent_key = (uint32_t *)((char *)keep + align_up((size_t)n * 8));
    ix->sorted_copy = sorted ? nullptr : (int32_t *)((char *)keep + 2 * align_up((size_t)n * 8));
    uint32_t *orig = sorted ? nullptr : (uint32_t *)((char *)keep + 2 * align_up((size_t)n * 8) + align_up((size_t)n * 12));
    ix->orig = orig;
    ix->bytes = keep_bytes;
    int *bad = (int *)(scratch + o_bad);
    uint32_t *k = (uint32_t *)(scratch + o_k), *v = (uint32_t *)(scratch + o_v);
    uint32_t *uk = (uint32_t *)(scratch + o_uk), *uv = (uint32_t *)(scratch + o_uv);
    uint32_t *uk2 = (uint32_t *)(scratch + o_uk2);
    int rc = DAISY_OK;
    auto fail = [&](int code) {
        (void)hipFree(scratch);
        (void)hipFree(keep);
        delete ix;
        return code;
    };
    if (hipMemsetAsync(bad, 0, 8, s) != hipSuccess) return fail(DAISY_ERR_HIP);
    const int32_t *src = triples;
    if (!sorted) {   // CSR order first: stable sort of the row indices by user, then one gather
        hipLaunchKernelGGL(k_index_entries, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, triples, n, user_base,
                           user_num, item_num, (uint32_t *)nullptr, (uint32_t *)nullptr, uk, uv, bad, pointwise);
        rc = sort_pairs_i32(scratch + o_tmp, t_sort, (const int32_t *)uk, (int32_t *)uk2, (const int32_t *)uv,
                            (int32_t *)orig, n, bits_for(user_num), s);
        if (rc) return fail(rc);
        hipLaunchKernelGGL(k_gather_triples, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, triples, orig, n,
                           ix->sorted_copy);
        src = ix->sorted_copy;
    }
    ix->triples = src;
    hipLaunchKernelGGL(k_index_entries, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, src, n, user_base, user_num,
                       item_num, k, v, (uint32_t *)nullptr, (uint32_t *)nullptr, bad, pointwise);
    rc = sort_pairs_i32(scratch + o_tmp, t_sort, (const int32_t *)k, (int32_t *)ix->ent_key, (const int32_t *)v,
                        (int32_t *)ix->ent_t, ix->n_ent, bits_for(item_num) + 1, s);
    if (rc) return fail(rc);
    // (bad[1]: entries of the most frequent item; out-of-range items were replaced by 0 and are reported below)
    uint32_t *hist = (uint32_t *)(scratch + o_hist);
    if (hipMemsetAsync(hist, 0, (size_t)item_num * 8, s) != hipSuccess) return fail(DAISY_ERR_HIP);
    hipLaunchKernelGGL(k_item_bounds, dim3(grid_for(ix->n_ent, kBlock * 4)), dim3(kBlock), 0, s, ix->ent_key, ix->n_ent, hist,
                       hist + item_num);
    hipLaunchKernelGGL(k_item_max_len, dim3(grid_for(item_num, kBlock * 4, 256)), dim3(kBlock), 0, s, hist, hist + item_num,
                       item_num, (uint32_t *)(bad + 1));
    int bad_host[2] = {0, 0};
    if (hipMemcpyAsync(bad_host, bad, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        set_error("train_index_create: reading the validation flag failed");
        return fail(DAISY_ERR_HIP);
    }
    ix->max_item_entries = (int64_t)(uint32_t)bad_host[1];
    (void)hipFree(scratch);
    if (bad_host[0]) {
        set_error("index out of range in the training triples: need %d <= user < %lld and 0 <= item < %lld "
                  "(the reference raises IndexError in nn.Embedding, MFRecommender.py:64-65)",
                  user_base, (long long)(user_base + user_num), (long long)item_num);
        (void)hipFree(keep);
        delete ix;
        return DAISY_ERR_ARG;
    }
    *out = ix;
    return DAISY_OK;
}

int daisy_train_index_destroy(daisy_train_index *index) {
    if (!index) return DAISY_OK;
    hipError_t e = hipFree(index->ent_t);
    delete index;
    if (e != hipSuccess) {
        set_error("train_index_destroy: hipFree failed: %s", hipGetErrorString(e));
        return DAISY_ERR_HIP;
    }
    return DAISY_OK;
}

size_t daisy_train_index_bytes(const daisy_train_index *index) { return index ? index->bytes : 0; }

int daisy_epoch_plan_build_indexed(daisy_epoch_plan *plan, const daisy_train_index *index, const int64_t *perm,
                                   int32_t order_mode, uint64_t seed, uint64_t epoch, int64_t batch_size,
                                   daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan && index, "epoch_plan_build_indexed: NULL argument");
    DAISY_CHECK_ARG(index->n <= plan->max_triples && index->U == plan->U && index->I == plan->I,
                    "epoch_plan_build_indexed: the index (n %lld, U %lld, I %lld) does not fit the plan",
                    (long long)index->n, (long long)index->U, (long long)index->I);
    DAISY_CHECK_ARG(batch_size > 0 && batch_size < ((int64_t)1 << 31), "epoch_plan_build_indexed: bad batch_size");
    DAISY_CHECK_ARG(order_mode >= DAISY_ORDER_IDENTITY && order_mode <= DAISY_ORDER_FEISTEL,
                    "epoch_plan_build_indexed: bad order_mode %d", order_mode);
    DAISY_CHECK_ARG(order_mode != DAISY_ORDER_PERM || perm != nullptr,
                    "epoch_plan_build_indexed: DAISY_ORDER_PERM needs perm");
    return plan_build_partitioned(plan, index, perm, order_mode, seed, epoch, batch_size, 0, S(stream));
}

int daisy_epoch_plan_build_positions(daisy_epoch_plan *plan, const daisy_train_index *index, const int64_t *positions,
                                     int64_t n_total, int64_t batch_size, daisy_stream_t stream) {
    DAISY_CHECK_ARG(plan && index && positions, "epoch_plan_build_positions: NULL argument");
    DAISY_CHECK_ARG(index->n <= plan->max_triples && index->U == plan->U && index->I == plan->I,
                    "epoch_plan_build_positions: the index (n %lld, U %lld, I %lld) does not fit the plan",
                    (long long)index->n, (long long)index->U, (long long)index->I);
    DAISY_CHECK_ARG(batch_size > 0 && batch_size < ((int64_t)1 << 31), "epoch_plan_build_positions: bad batch_size");
    DAISY_CHECK_ARG(n_total >= index->n && n_total < ((int64_t)1 << 32),
                    "epoch_plan_build_positions: n_total=%lld must be in [n, 2^32)", (long long)n_total);
    return plan_build_partitioned(plan, index, positions, DAISY_ORDER_PERM, 0, 0, batch_size, n_total, S(stream));
}

int64_t daisy_epoch_plan_batch_rows(const daisy_epoch_plan *plan, int64_t k) {
    if (!plan || !plan->built || k < 0 || k >= plan->num_batches) return -1;
    if (plan->h_off) return plan->h_off[k + 1] - plan->h_off[k];
    const int64_t lo = k * plan->batch_size;
    return (plan->n - lo < plan->batch_size) ? (plan->n - lo) : plan->batch_size;
}

int daisy_bpr_ctx_invalidate_cache(daisy_bpr_ctx *ctx) {
    DAISY_CHECK_ARG(ctx != nullptr, "ctx_invalidate_cache: NULL context");
    ctx->p_sqnorm_of = nullptr;
    ctx->pre_ready = false;       // the next batch's pre-norm that rode on the last item pass was summed from the old cache
    return DAISY_OK;
}

int daisy_bpr_ctx_set_p_stream(daisy_bpr_ctx *ctx, int32_t mode) {
    DAISY_CHECK_ARG(ctx != nullptr, "ctx_set_p_stream: NULL context");
    DAISY_CHECK_ARG(mode >= -1 && mode <= 1, "ctx_set_p_stream: mode %d not in -1 (automatic), 0, 1", mode);
    ctx->p_stream_mode = mode;
    return DAISY_OK;
}

int daisy_bpr_staged_prenorm(daisy_bpr_ctx *ctx, const float *P, double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && stats, "staged_prenorm: NULL argument");
    if (!ctx->batch_set) { set_error("staged_prenorm: no batch set"); return DAISY_ERR_STATE; }
    return staged_prenorm(ctx, P, stats, true, nullptr, S(stream));
}

int daisy_bpr_staged_user(daisy_bpr_ctx *ctx, float *P, const float *Q, int32_t loss_type, float gamma, float lr,
                          float reg_1, float reg_2, double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats, "staged_user: NULL argument");
    int rc = staged_check(ctx, loss_type, "staged_user");
    if (rc) return rc;
    if (ctx->p_sqnorm_of != P) {
        set_error("staged_user: daisy_bpr_staged_prenorm has not run on this table");
        return DAISY_ERR_STATE;
    }
    int gu = 0;
    if ((rc = staged_user(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, stats, &gu, 0, false, nullptr, nullptr, S(stream)))) return rc;
    return launch_reduce_partials(ctx->partials, gu, stats, false, 0.f, 0.f, nullptr, nullptr, S(stream));
}

int daisy_bpr_staged_item(daisy_bpr_ctx *ctx, int32_t loss_type, float *Q, float *gQ, float *cnt, float lr,
                          float reg_1, float reg_2, const double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && stats && ((Q && !gQ) || (!Q && gQ && cnt)),
                    "staged_item: give either Q (apply in place) or gQ + cnt (gradient output)");
    int rc = staged_check(ctx, loss_type, "staged_item");
    if (rc) return rc;
    return staged_item(ctx, loss_type, Q ? Q : gQ, cnt, Q != nullptr, lr, reg_1, reg_2, stats, S(stream));
}

int daisy_bpr_staged_item_slices(daisy_bpr_ctx *ctx, const int32_t *item_bounds, int32_t n_slices, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && item_bounds && n_slices >= 1 && n_slices <= kMaxItemSlices,
                    "staged_item_slices: 1..%d slices", kMaxItemSlices);
    if (!ctx->batch_set) { set_error("staged_item_slices: no batch set"); return DAISY_ERR_STATE; }
    SliceBounds sb;
    for (int k = 0; k <= n_slices; ++k) {
        DAISY_CHECK_ARG(item_bounds[k] >= 0 && (k == 0 || item_bounds[k] >= item_bounds[k - 1]),
                        "staged_item_slices: the item bounds must not decrease");
        sb.b[k] = item_bounds[k];
    }
    DAISY_CHECK_ARG(item_bounds[0] == 0 && item_bounds[n_slices] >= ctx->I,
                    "staged_item_slices: the slices must cover the items 0..%lld", (long long)ctx->I);
    hipLaunchKernelGGL(k_slice_ranges, dim3(1), dim3(64), 0, S(stream), ctx->sv, sb, (int)n_slices, ctx->slice_rng);
    DAISY_LAUNCH_CHECK();
    ctx->n_slices = n_slices;
    return DAISY_OK;
}

int daisy_bpr_staged_item_slice(daisy_bpr_ctx *ctx, int32_t loss_type, float *gQ, float *cnt, int32_t slice, float lr,
                                float reg_1, float reg_2, const double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && gQ && cnt && stats, "staged_item_slice: NULL argument");
    int rc = staged_check(ctx, loss_type, "staged_item_slice");
    if (rc) return rc;
    DAISY_CHECK_ARG(slice >= 0 && slice < ctx->n_slices, "staged_item_slice: slice %d of %d (daisy_bpr_staged_item_slices first)",
                    slice, ctx->n_slices);
    return staged_item(ctx, loss_type, gQ, cnt, false, lr, reg_1, reg_2, stats, S(stream), slice);
}

int daisy_bpr_staged_adam_step(daisy_bpr_ctx *ctx, float *P, float *Q, int32_t loss_type, float gamma, float lr,
                               float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP, float *mQ, float *vQ,
                               int32_t *lastQ, const float *table, float beta1, float beta2, float eps, int64_t step,
                               double *stats, double *epoch_acc, double *step_loss, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && mP && vP && lastP && mQ && vQ && lastQ && table && stats && step >= 1,
                    "staged_adam_step: bad argument");
    StagedAdam a = adam_consts(lr, beta1, beta2, eps, step);
    a.mP = mP; a.vP = vP; a.lastP = lastP; a.mQ = mQ; a.vQ = vQ; a.lastQ = lastQ;
    struct HaltScope {
        daisy_bpr_ctx *c;
        HaltScope(daisy_bpr_ctx *c_, const double *h) : c(c_) { c->v.halt = h; c->sv.halt = h; }
        ~HaltScope() { c->v.halt = nullptr; c->sv.halt = nullptr; }
    } halt_scope(ctx, epoch_acc ? epoch_acc + 1 : nullptr);
    const bool bias_grad_out = ctx->bu != nullptr;         // FM: the caller's dense optimiser steps the biases
    if (bias_grad_out && !(ctx->g_bu && ctx->g_bi)) {
        set_error("staged_adam_step: the context has biases but no g_u_bias / g_i_bias outputs");
        return DAISY_ERR_ARG;
    }
    return staged_adam_step(ctx, P, Q, loss_type, gamma, reg_1, reg_2, a, table, bias_grad_out, stats, epoch_acc,
                            step_loss, S(stream));
}

// The epoch loop of GeneralRecommender.fit (AbstractRecommender.py:118-128) with torch.optim.Adam
// (AbstractRecommender.py:54): every batch of a built plan through the staged Adam step, enqueued natively - steps
// first_step, first_step + 1, ... - and, flush != 0, the rows no batch referenced brought up to the epoch's last step
// (daisy_adam_lazy_flush on both tables), so the tables are current when the call's work has drained.
int daisy_bpr_fit_epoch_adam(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q, int32_t loss_type,
                             float gamma, float lr, float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP,
                             float *mQ, float *vQ, int32_t *lastQ, const float *table, int64_t table_steps, float beta1,
                             float beta2, float eps, int64_t first_step, int32_t flush, double *stats, double *epoch_acc,
                             double *step_losses, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && plan && P && Q && mP && vP && lastP && mQ && vQ && lastQ && table && stats && first_step >= 1,
                    "fit_epoch_adam: bad argument");
    if (!plan->built) { set_error("fit_epoch_adam: plan has not been built"); return DAISY_ERR_STATE; }
    DAISY_CHECK_ARG(loss_type >= DAISY_LOSS_BPR && loss_type <= DAISY_LOSS_SL, "Invalid loss type: %d", loss_type);
    DAISY_CHECK_ARG(first_step + plan->num_batches - 1 <= table_steps,
                    "fit_epoch_adam: the constants table holds %lld steps, the epoch ends at step %lld",
                    (long long)table_steps, (long long)(first_step + plan->num_batches - 1));
    if (ctx->bu) {          // FM: the biases step through the caller's dense optimiser between two steps
        set_error("fit_epoch_adam: contexts with FM biases are driven step by step (daisy_bpr_staged_adam_step)");
        return DAISY_ERR_ARG;
    }
    if (small_epoch_supported(ctx, plan, loss_type)) {
        // batches of a few hundred samples over the sorted plan (the reference's default batch): every step of the epoch
        // inside one persistent workgroup, like the SGD epoch (csrc/bpr_small.hip) - 48 -> ~12 us per step at B = 256
        DAISY_CHECK_ARG(plan->U == ctx->U && plan->I == ctx->I, "fit_epoch_adam: plan does not fit the context");
        SmallAdamArgs ad{mP, vP, lastP, mQ, vQ, lastQ, table, beta1, beta2, eps, first_step};
        int rc = small_fit_epoch(ctx, plan, P, Q, loss_type, gamma, lr, reg_1, reg_2, stats, epoch_acc, step_losses, S(stream), &ad);
        if (rc) return rc;
    } else {
        if (plan->kind == 0) {
            set_error("fit_epoch_adam: a plan in the sorted layout is only run by the small-batch epoch kernel "
                      "(daisy_bpr_small_epoch_supported); build the partitioned layout (daisy_epoch_plan_build_indexed)");
            return DAISY_ERR_ARG;
        }
    for (int64_t k = 0; k < plan->num_batches; ++k) {
        int rc = daisy_bpr_set_batch_from_plan(ctx, plan, k, stream);
        if (rc) return rc;
        rc = daisy_bpr_staged_adam_step(ctx, P, Q, loss_type, gamma, lr, reg_1, reg_2, mP, vP, lastP, mQ, vQ, lastQ, table,
                                        beta1, beta2, eps, first_step + k, stats, epoch_acc,
                                        step_losses ? step_losses + k : nullptr, stream);
        if (rc) return rc;
    }
    }
    if (flush && plan->num_batches > 0) {
        const int64_t t = first_step + plan->num_batches - 1;
        int rc = daisy_adam_lazy_flush(P, mP, vP, lastP, ctx->U, ctx->d, table, beta1, beta2, eps, t, stream);
        if (rc) return rc;
        if ((rc = daisy_adam_lazy_flush(Q, mQ, vQ, lastQ, ctx->I, ctx->d, table, beta1, beta2, eps, t, stream))) return rc;
        ctx->p_sqnorm_of = nullptr;          // the flush rewrote rows of P behind the row-norm cache
        ctx->pre_ready = false;
    }
    return DAISY_OK;
}

int daisy_bpr_small_epoch_supported(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int32_t loss_type) {
    if (!ctx || !plan || !plan->built) return 0;
    if (!small_epoch_supported(ctx, plan, loss_type)) return 0;
    return small_epoch_adam_pays(ctx, plan) ? 3 : 1;
}

int daisy_bpr_staged_adam_catchup_users(daisy_bpr_ctx *ctx, float *P, float *mP, float *vP, int32_t *lastP,
                                        const float *table, float beta1, float beta2, float eps, int64_t step,
                                        daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && mP && vP && lastP && table && step >= 1, "staged_adam_catchup_users: bad argument");
    if (!ctx->batch_set) { set_error("staged_adam_catchup_users: no batch set"); return DAISY_ERR_STATE; }
    StagedAdam a = adam_consts(0.f, beta1, beta2, eps, step);
    a.mP = mP; a.vP = vP; a.lastP = lastP;
    hipStream_t s = S(stream);
    const StreamView &v = ctx->sv;
    const int d = ctx->d;
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        if (ctx->p_sqnorm_of != P) {
            hipLaunchKernelGGL((k_row_sqnorm<C>), dim3(grid_for(ctx->U, C::GROUPS_PER_BLOCK * 4)), dim3(kBlock), 0, s,
                               P, ctx->U, d, ctx->p_sqnorm);
            ctx->p_sqnorm_of = P;
        }
        hipLaunchKernelGGL((k_staged_adam_catchup<C>), dim3(grid_for(v.B, C::GROUPS_PER_BLOCK, kMaxGridSparse)), dim3(kBlock),
                           0, s, v, d, P, (float *)nullptr, a, reinterpret_cast<const float2 *>(table), ctx->p_sqnorm, 1);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_bpr_staged_user_adam(daisy_bpr_ctx *ctx, float *P, const float *Q, int32_t loss_type, float gamma, float lr,
                               float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP, float beta1, float beta2,
                               float eps, int64_t step, double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(ctx && P && Q && stats && mP && vP && lastP && step >= 1, "staged_user_adam: NULL argument");
    int rc = staged_check(ctx, loss_type, "staged_user_adam");
    if (rc) return rc;
    if (ctx->p_sqnorm_of != P) {
        set_error("staged_user_adam: daisy_bpr_staged_prenorm has not run on this table");
        return DAISY_ERR_STATE;
    }
    if (ctx->bu) { set_error("staged_user_adam: contexts with FM biases are not supported by the phase form"); return DAISY_ERR_ARG; }
    StagedAdam a = adam_consts(lr, beta1, beta2, eps, step);
    a.mP = mP; a.vP = vP; a.lastP = lastP;
    int gu = 0;
    if ((rc = staged_user(ctx, P, Q, loss_type, gamma, 0.f, reg_1, reg_2, stats, &gu, 0, false, nullptr, nullptr, S(stream), &a))) return rc;
    return launch_reduce_partials(ctx->partials, gu, stats, false, 0.f, 0.f, nullptr, nullptr, S(stream));
}

int daisy_item_apply_counts_adam(float *Q, float *g, float *cnt, float *m, float *v, int64_t rows, int32_t d, float lr,
                                 float reg_1, float reg_2, float beta1, float beta2, float eps, int64_t step,
                                 const double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(Q && g && cnt && m && v && stats && rows > 0 && step >= 1, "item_apply_counts_adam: bad argument");
    const StagedAdam a = adam_consts(lr, beta1, beta2, eps, step);
    const RowOpt opt = row_opt(lr, &a, false);
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_item_apply_counts_adam<C>), dim3(grid_for(rows, C::GROUPS_PER_BLOCK * 2)), dim3(kBlock), 0,
                           S(stream), Q, g, cnt, m, v, rows, (int)d, reg_1, reg_2, opt, stats);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_item_apply_counts(float *Q, float *g, float *cnt, int64_t rows, int32_t d, float lr, float reg_1,
                            float reg_2, const double *stats, daisy_stream_t stream) {
    DAISY_CHECK_ARG(Q && g && cnt && stats && rows > 0, "item_apply_counts: bad argument");
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_item_apply_counts<C>), dim3(grid_for(rows, C::GROUPS_PER_BLOCK * 2)), dim3(kBlock), 0,
                           S(stream), Q, g, cnt, rows, (int)d, lr, reg_1, reg_2, stats);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // extern "C"
