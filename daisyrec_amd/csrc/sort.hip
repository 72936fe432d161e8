// rocPRIM (AMD's native device primitives) wrappers.  Only index preparation
// goes through these radix sorts (grouping a batch by user / by item, the
// epoch permutation, argsort of candidate scores); the gather / dot / update
// kernels of the hot path are hand written in bpr_train.hip.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace daisy {

// rocPRIM's default sends every sort of up to 1 M items to its merge sort, which ignores end_bit: ~18 launches of
// ~6 us for the 524 288 13-bit keys of a NeuMF step, where Onesweep needs a histogram and two digit passes.
// The id sorts here know their key width, so Onesweep takes over above 262 144 items (measured: merge sort still wins at 131 072 and below).
using IdSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::default_config, 262144>;

size_t sort_pairs_i32_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<IdSortConfig>(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0, 32);
    return bytes;
}

int sort_pairs_i32(void *temp, size_t temp_bytes, const int32_t *kin, int32_t *kout,
                   const int32_t *vin, int32_t *vout, int64_t n, int end_bit, hipStream_t s) {
    // keys are non-negative ids, so unsigned ordering == signed ordering
    DAISY_HIP(rocprim::radix_sort_pairs<IdSortConfig>(temp, temp_bytes, reinterpret_cast<const uint32_t *>(kin),
                                        reinterpret_cast<uint32_t *>(kout), vin, vout, (size_t)n, 0,
                                        (unsigned)end_bit, s));
    return DAISY_OK;
}

size_t rle_u32_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::run_length_encode(nullptr, bytes, (const uint32_t *)nullptr, (unsigned)n,
                                     (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    return bytes;
}

int rle_u32(void *temp, size_t temp_bytes, const uint32_t *in, int64_t n, uint32_t *unique_out,
            uint32_t *counts_out, uint32_t *runs_out, hipStream_t s) {
    DAISY_HIP(rocprim::run_length_encode(temp, temp_bytes, in, (unsigned)n, unique_out, counts_out,
                                         runs_out, s));
    return DAISY_OK;
}

size_t rle_u64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::run_length_encode(nullptr, bytes, (const uint64_t *)nullptr, (unsigned)n,
                                     (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    return bytes;
}

int rle_u64(void *temp, size_t temp_bytes, const uint64_t *in, int64_t n, uint64_t *unique_out,
            uint32_t *counts_out, uint32_t *runs_out, hipStream_t s) {
    DAISY_HIP(rocprim::run_length_encode(temp, temp_bytes, in, (unsigned)n, unique_out, counts_out,
                                         runs_out, s));
    return DAISY_OK;
}

size_t exclusive_scan_u32_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u,
                                  (size_t)n, rocprim::plus<uint32_t>());
    return bytes;
}

int exclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int64_t n,
                       hipStream_t s) {
    DAISY_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, (size_t)n,
                                      rocprim::plus<uint32_t>(), s));
    return DAISY_OK;
}

size_t sort_pairs_u32_u64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<IdSortConfig>(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n, 0, 32);
    return bytes;
}

int sort_pairs_u32_u64(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                       const uint64_t *vin, uint64_t *vout, int64_t n, int begin_bit, int end_bit,
                       hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_pairs<IdSortConfig>(temp, temp_bytes, kin, kout, vin, vout, (size_t)n,
                                        (unsigned)begin_bit, (unsigned)end_bit, s));
    return DAISY_OK;
}

size_t sort_pairs_u64_u64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<IdSortConfig>(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n, 0, 64);
    return bytes;
}

int sort_pairs_u64_u64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const uint64_t *vin, uint64_t *vout, int64_t n, int begin_bit, int end_bit,
                       hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_pairs<IdSortConfig>(temp, temp_bytes, kin, kout, vin, vout, (size_t)n,
                                        (unsigned)begin_bit, (unsigned)end_bit, s));
    return DAISY_OK;
}

size_t sort_pairs_u64_i32_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<IdSortConfig>(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0, 64);
    return bytes;
}

int sort_pairs_u64_i32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const int32_t *vin, int32_t *vout, int64_t n, int end_bit, hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_pairs<IdSortConfig>(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0,
                                        (unsigned)end_bit, s));
    return DAISY_OK;
}

size_t sort_pairs_u64_i64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<IdSortConfig>(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const int64_t *)nullptr, (int64_t *)nullptr, (size_t)n, 0, 64);
    return bytes;
}

int sort_pairs_u64_i64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                       const int64_t *vin, int64_t *vout, int64_t n, int end_bit, hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_pairs<IdSortConfig>(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0,
                                        (unsigned)end_bit, s));
    return DAISY_OK;
}

size_t sort_keys_u64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys<IdSortConfig>(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                   (size_t)n, 0, 64);
    return bytes;
}

int sort_keys_u64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout, int64_t n,
                  int end_bit, hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_keys<IdSortConfig>(temp, temp_bytes, kin, kout, (size_t)n, 0, (unsigned)end_bit,
                                       s));
    return DAISY_OK;
}

namespace {
struct MulBy {
    int64_t c;
    __host__ __device__ int64_t operator()(int64_t x) const { return x * c; }
};
using OffIt = rocprim::transform_iterator<rocprim::counting_iterator<int64_t>, MulBy, int64_t>;
inline OffIt offsets(int64_t C) {
    return rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), MulBy{C});
}
}  // namespace

size_t seg_sort_desc_f32_i64_temp_bytes(int64_t n, int64_t segs) {
    size_t bytes = 0;
    OffIt off = offsets(n / (segs > 0 ? segs : 1));
    (void)rocprim::segmented_radix_sort_pairs_desc(nullptr, bytes, (const float *)nullptr,
                                                   (float *)nullptr, (const int64_t *)nullptr,
                                                   (int64_t *)nullptr, (unsigned)n, (unsigned)segs,
                                                   off, off + 1, 0, 32);
    return bytes;
}

int seg_sort_desc_f32_i64(void *temp, size_t temp_bytes, const float *kin, float *kout,
                          const int64_t *vin, int64_t *vout, int64_t segs, int64_t C,
                          hipStream_t s) {
    OffIt off = offsets(C);
    DAISY_HIP(rocprim::segmented_radix_sort_pairs_desc(temp, temp_bytes, kin, kout, vin, vout,
                                                       (unsigned)(segs * C), (unsigned)segs, off,
                                                       off + 1, 0, 32, s));
    return DAISY_OK;
}

size_t sort_pairs_desc_f32_i64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (const float *)nullptr, (float *)nullptr,
                                         (const int64_t *)nullptr, (int64_t *)nullptr, (size_t)n, 0,
                                         32);
    return bytes;
}

int sort_pairs_desc_f32_i64(void *temp, size_t temp_bytes, const float *kin, float *kout,
                            const int64_t *vin, int64_t *vout, int64_t n, hipStream_t s) {
    DAISY_HIP(rocprim::radix_sort_pairs_desc(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, 32,
                                             s));
    return DAISY_OK;
}

}  // namespace daisy
