// Error reporting, ABI version and the memory-system micro-benchmarks.
#include <stdarg.h>

#include <atomic>

#include "common.h"

namespace daisy {

static thread_local char g_err[512] = "";

// ids of epoch-plan builds, unique in the process (daisy_epoch_plan::build_gen)
uint64_t next_plan_build_id() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// what: 0 = gather-read rows and sum, 1 = scatter fp32 atomics, 2 = scatter plain stores,
//       3 = read-modify-write rows (load, add, store)
template <class C>
__global__ __launch_bounds__(kBlock) void k_membench(int what, float *__restrict__ table, int d,
                                                     const int32_t *__restrict__ idx, int64_t n,
                                                     float *__restrict__ out) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    float acc = 0.f;
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < n; s += gstride) {
        float *row = table + (int64_t)idx[s] * d;
        Row<C> r;
        if (what == 0) {
            r.load(row, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k) acc += r.v[k];
        } else if (what == 1) {
#pragma unroll
            for (int k = 0; k < C::NE; ++k) r.v[k] = 1e-6f;
            r.atomic_add_to(row, lane, d);
        } else if (what == 2) {
#pragma unroll
            for (int k = 0; k < C::NE; ++k) r.v[k] = (float)s;
            r.store(row, lane, d);
        } else {
            r.load(row, lane, d);
#pragma unroll
            for (int k = 0; k < C::NE; ++k) r.v[k] += 1e-6f;
            r.store(row, lane, d);
        }
    }
    if (what == 0 && acc == 123.456f) out[0] = acc;  // keep the loads alive
}

}  // namespace daisy

using namespace daisy;

extern "C" {

const char *daisy_last_error(void) { return g_err; }

int daisy_abi_version(void) { return DAISY_ABI_VERSION; }

int daisy_membench(int32_t what, float *table, int64_t rows, int32_t d, const int32_t *idx, int64_t n,
                   float *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(table && idx && out && rows > 0 && n > 0 && what >= 0 && what <= 3,
                    "membench: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_membench<C>), dim3(grid_for(n, C::GROUPS_PER_BLOCK * 4)), dim3(kBlock), 0,
                           s, (int)what, table, (int)d, idx, n, out);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

}  // extern "C"
