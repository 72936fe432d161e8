// VERDICT r05 item 8 asked whether Adam's user pass at configs[2] shapes (10 M users) would gain from storing a user's row, m
// and v as ONE 768-byte record instead of three 256-byte rows in three arrays.  This probe answers it without touching the
// library: the same read-modify-write of 2 M sorted random users (a batch's stale users), once over three [N][64] float
// arrays, once over one [N][192] array, 16 lanes x float4 per 256 bytes either way; GB/s of (bytes read + written).
//   hipcc --offload-arch=gfx950 -O2 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

// PARTS arrays of rows of 64 floats (pitch = row pitch in floats; the three-array form passes three bases with pitch 64, the
// record form one base with pitch 192 and offsets 0 / 64 / 128)
template <int PARTS>
__global__ __launch_bounds__(256) void k_rmw(float *a, float *b, float *c, int pitch, int pitch_bc, const int32_t *__restrict__ ids, int64_t m) {
    const int lane = threadIdx.x % 16, group = threadIdx.x / 16;
    for (int64_t e = blockIdx.x * 16ll + group; e < m; e += (int64_t)gridDim.x * 16) {
        const int64_t r = ids[e];
        float4 *pa = reinterpret_cast<float4 *>(a + r * pitch) + lane;
        float4 *pb = reinterpret_cast<float4 *>(b + r * pitch_bc) + lane;
        float4 *pc = reinterpret_cast<float4 *>(c + r * pitch_bc) + lane;
        float4 x = *pa, y = make_float4(0, 0, 0, 0), z = y;
        if (PARTS > 1) { y = *pb; z = *pc; }
        x.x += 1e-3f * (y.x + z.x); x.y += 1e-3f * (y.y + z.y); x.z += 1e-3f * (y.z + z.z); x.w += 1e-3f * (y.w + z.w);
        *pa = x;
        if (PARTS > 1) { y.x *= 0.9f; y.y *= 0.9f; y.z *= 0.9f; y.w *= 0.9f; z.x *= 0.999f; z.y *= 0.999f; z.z *= 0.999f; z.w *= 0.999f; *pb = y; *pc = z; }
    }
}

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, int64_t n4, float *out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = a[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ a, int64_t n4) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const int64_t N = 10000000, M = 2097152;
    std::vector<int32_t> ids(M);
    std::mt19937_64 rng(1);
    for (auto &v : ids) v = (int32_t)(rng() % N);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int64_t m = (int64_t)ids.size();
    int32_t *d_ids;
    float *A, *B, *C, *Rec;
    CHECK(hipMalloc(&d_ids, m * 4));
    CHECK(hipMemcpy(d_ids, ids.data(), m * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&A, N * 64 * 4)); CHECK(hipMalloc(&B, N * 64 * 4)); CHECK(hipMalloc(&C, N * 64 * 4));
    CHECK(hipMalloc(&Rec, N * 192 * 4));
    CHECK(hipMemset(A, 0, N * 64 * 4)); CHECK(hipMemset(B, 0, N * 64 * 4)); CHECK(hipMemset(C, 0, N * 64 * 4));
    CHECK(hipMemset(Rec, 0, N * 192 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char *name, double bytes) {
        for (int k = 0; k < 3; ++k) launch();
        float best = 1e30f, sum = 0.f;
        for (int k = 0; k < 10; ++k) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-58s %8.1f us (best %8.1f)  %6.2f TB/s\n", name, sum / 10 * 1e3, best * 1e3, bytes / (sum / 10 * 1e-3) / 1e12);
    };
    printf("%lld distinct sorted random rows of %lld\n", (long long)m, (long long)N);
    const int grid = 16384;
    timeit([&] { hipLaunchKernelGGL((k_rmw<1>), dim3(grid), dim3(256), 0, 0, A, A, A, 64, 64, d_ids, m); },
           "one 256-B row per user, read + written (SGD's user rows)", m * 512.0);
    timeit([&] { hipLaunchKernelGGL((k_rmw<3>), dim3(grid), dim3(256), 0, 0, A, B, C, 64, 64, d_ids, m); },
           "three 256-B rows in three arrays (row, m, v)", m * 1536.0);
    timeit([&] { hipLaunchKernelGGL((k_rmw<3>), dim3(grid), dim3(256), 0, 0, Rec, Rec + 64, Rec + 128, 192, 192, d_ids, m); },
           "one 768-B record per user", m * 1536.0);
    timeit([&] { hipLaunchKernelGGL((k_rmw<3>), dim3(grid), dim3(256), 0, 0, A, Rec, Rec + 64, 64, 128, d_ids, m); },
           "row in its array, m and v as one 512-B pair", m * 1536.0);
    // the same kernel over CONSECUTIVE rows (what a streaming pass reaches with this access shape), and a plain device copy
    std::vector<int32_t> seq(m);
    for (int64_t k = 0; k < m; ++k) seq[k] = (int32_t)k;
    CHECK(hipMemcpy(d_ids, seq.data(), m * 4, hipMemcpyHostToDevice));
    timeit([&] { hipLaunchKernelGGL((k_rmw<1>), dim3(grid), dim3(256), 0, 0, A, A, A, 64, 64, d_ids, m); },
           "consecutive 256-B rows, read + written", m * 512.0);
    timeit([&] { hipLaunchKernelGGL((k_rmw<3>), dim3(grid), dim3(256), 0, 0, A, B, C, 64, 64, d_ids, m); },
           "consecutive rows of three arrays, read + written", m * 1536.0);
    timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4 *>(A), N * 16, C); },
           "read only, 2.56 GB streamed", (double)N * 64 * 4);
    timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float4 *>(B), N * 16); },
           "write only, 2.56 GB streamed", (double)N * 64 * 4);
    timeit([&] { (void)hipMemcpyAsync(B, A, (size_t)N * 64 * 4, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy device to device, 2.56 GB",
           2.0 * N * 64 * 4);
    return 0;
}
