// What one DEPENDENT global-memory read costs a kernel on MI355X, by where the line was last written: the unit in which the
// launch-bound steps of this library (NeuMF at 256 samples, MF at 16 k - 64 k samples) are best understood - a kernel whose
// work is "ids -> rows -> owner rows -> result" cannot finish before 3-4 of these have passed, whatever its bandwidth.
//   hipcc --offload-arch=gfx950 -O2 tools/latency_probe.hip -o /tmp/latency_probe && /tmp/latency_probe
// One thread of one workgroup chases a random cycle through a buffer (64 dependent loads per measurement), timed with the
// 100 MHz wall clock.  Cases: the buffer written by a PREVIOUS kernel spread over all XCDs (what a step's kernel sees of its
// predecessor's output), the same lines chased a second time (now in this XCD's L2), a 1 GB buffer (HBM + TLB), a buffer
// written by the host (hipMemcpy).  Also: kernel-argument fetch + first instruction (the clock at kernel entry against the
// end of the previous kernel is not observable from the device; the empty-kernel round trip on the host is printed instead).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void k_write_chain(const uint32_t *__restrict__ next, uint32_t *__restrict__ buf, size_t n, size_t stride_words) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        buf[i * stride_words] = next[i];
}

__global__ void k_chase(const uint32_t *buf, size_t stride_words, uint32_t start, int hops, int passes, long long *out, uint32_t *sink) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t p = start;
    for (int pass = 0; pass < passes; ++pass) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p) : : "memory");
        const long long t0 = wall_clock64();
        for (int h = 0; h < hops; ++h) p = __builtin_nontemporal_load(buf + (size_t)p * stride_words);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p) : : "memory");      // (the clock read must not move above the last load)
        out[pass] = wall_clock64() - t0;
        p = start;                      // the same lines again
    }
    *sink = p;
}

// the same with ordinary loads (the nontemporal form above bypasses nothing on the read side; both printed)
__global__ void k_chase_plain(const uint32_t *buf, size_t stride_words, uint32_t start, int hops, int passes, long long *out, uint32_t *sink) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t p = start;
    for (int pass = 0; pass < passes; ++pass) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p) : : "memory");
        const long long t0 = wall_clock64();
        for (int h = 0; h < hops; ++h) p = buf[(size_t)p * stride_words];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p) : : "memory");
        out[pass] = wall_clock64() - t0;
        p = start;
    }
    *sink = p;
}

__global__ void k_empty() {}

// Two workgroups hand a token back and forth (release store / acquire poll at agent scope, the hand-off of the small-batch Adam
// kernel's helper workgroups): workgroup `a` and workgroup `b` of the grid play, the others leave at once.  Workgroups go to
// the XCDs round robin, so (0, 1) are on different XCDs and (0, 8) on the same one.
__global__ void k_pingpong(int *flags, int a, int b, int rounds, long long *out) {
    if (threadIdx.x) return;
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    int *mine = flags + 64 * me, *theirs = flags + 64 * (1 - me);
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            __hip_atomic_store(mine, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(theirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r) {}
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r) {}
            __hip_atomic_store(mine, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) out[0] = wall_clock64() - t0;
}

static int run_case(const char *name, size_t n, size_t stride_words, bool host_written) {
    std::vector<uint32_t> perm(n), next(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < n; ++i) next[perm[i]] = perm[(i + 1) % n];       // one cycle through all slots
    uint32_t *d_next, *d_buf, *d_sink;
    long long *d_out;
    CHECK(hipMalloc(&d_next, n * 4));
    CHECK(hipMalloc(&d_buf, n * stride_words * 4));
    CHECK(hipMalloc(&d_out, 8 * sizeof(long long)));
    CHECK(hipMalloc(&d_sink, 4));
    CHECK(hipMemcpy(d_next, next.data(), n * 4, hipMemcpyHostToDevice));
    const int hops = 64;
    for (int plain = 0; plain < 2; ++plain) {
        long long best[3] = {1 << 30, 1 << 30, 1 << 30}, sum[3] = {0, 0, 0};
        const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            if (host_written) {
                std::vector<uint32_t> img(n * stride_words, 0u);
                for (size_t i = 0; i < n; ++i) img[i * stride_words] = next[i];
                CHECK(hipMemcpy(d_buf, img.data(), img.size() * 4, hipMemcpyHostToDevice));
            } else {
                hipLaunchKernelGGL(k_write_chain, dim3(2048), dim3(256), 0, 0, d_next, d_buf, n, stride_words);
            }
            if (plain) hipLaunchKernelGGL(k_chase_plain, dim3(1), dim3(64), 0, 0, d_buf, stride_words, perm[r % n], hops, 3, d_out, d_sink);
            else hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d_buf, stride_words, perm[r % n], hops, 3, d_out, d_sink);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            long long h[3];
            CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
            for (int k = 0; k < 3; ++k) { sum[k] += h[k]; if (h[k] < best[k]) best[k] = h[k]; }
        }
        printf("%-44s %s loads: first pass %6.0f ns per dependent load (best %5.0f), second pass %5.0f (best %5.0f), third %5.0f\n", name,
               plain ? "plain      " : "nontemporal", sum[0] * 10.0 / reps / hops, best[0] * 10.0 / hops, sum[1] * 10.0 / reps / hops,
               best[1] * 10.0 / hops, sum[2] * 10.0 / reps / hops);
    }
    (void)hipFree(d_next); (void)hipFree(d_buf); (void)hipFree(d_out); (void)hipFree(d_sink);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    if (run_case("4 MB buffer written by the previous kernel", 1 << 14, 64, false)) return 1;          // one slot per 256 B
    if (run_case("4 MB buffer written by the host (hipMemcpy)", 1 << 14, 64, true)) return 1;
    if (run_case("1 GB buffer written by the previous kernel", 1 << 22, 64, false)) return 1;
    {
        int *flags; long long *d_out;
        CHECK(hipMalloc(&flags, 1024)); CHECK(hipMalloc(&d_out, 64));
        const int pairs[3][2] = {{0, 1}, {0, 8}, {0, 4}};
        for (auto &pr : pairs) {
            CHECK(hipMemset(flags, 0, 1024));
            hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(64), 0, 0, flags, pr[0], pr[1], 1000, d_out);
            CHECK(hipDeviceSynchronize());
            long long t;
            CHECK(hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost));
            printf("token between workgroups %d and %d (release store / acquire poll, agent scope): %.0f ns per round trip\n", pr[0], pr[1],
                   t * 10.0 / 1000);
        }
        (void)hipFree(flags); (void)hipFree(d_out);
    }
    // host-side: launch + completion of an empty kernel, and of a chain of 100 of them (per-launch cost in a queue)
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
    CHECK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 200; ++r) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); CHECK(hipDeviceSynchronize()); }
    auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < 20; ++r) { for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); CHECK(hipDeviceSynchronize()); }
    auto t2 = std::chrono::steady_clock::now();
    printf("empty kernel: launch + synchronize %.1f us; in a queue of 100: %.2f us per kernel\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    return 0;
}
