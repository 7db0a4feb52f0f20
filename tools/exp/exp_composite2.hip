// Experiment harness (not part of the library): times the product's own compositor kernels (csrc/composite.hip is
// included verbatim) on 2^20 synthetic rays.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DANR_EXP_STOP=n]
//   -o build/exp_composite2 tools/exp/exp_composite2.hip
#include <stdarg.h>
#include <stdlib.h>
#include "../../anim-nerf_amd/csrc/composite.hip"
namespace anr { int fail(int code, const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); return code; } }

__global__ void fill(float* p, int64_t n, float lo, float hi, uint32_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = lo + (hi - lo) * (h & 0xffffff) / 16777216.0f; }
}
__global__ void fill_rays(float* r, int64_t R) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) { r[i * 8 + 6] = 2.0f; r[i * 8 + 7] = 4.0f; }
}
template <class F> float time_it(F launch, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int64_t R = 1 << 20;
    float *rgbs, *z, *rays, *o3, *o1, *o2, *zs, *steps, *u, *w;
    hipMalloc(&rgbs, R * 128 * 16); hipMalloc(&z, R * 128 * 4); hipMalloc(&rays, R * 32); hipMalloc(&o3, R * 12); hipMalloc(&o1, R * 4);
    hipMalloc(&o2, R * 4); hipMalloc(&zs, R * 128 * 4); hipMalloc(&steps, 64 * 4); hipMalloc(&u, 64 * 4); hipMalloc(&w, R * 64 * 4);
    fill<<<(R * 128 * 4 + 255) / 256, 256>>>(rgbs, R * 128 * 4, -30.f, 3.f, 1);
    fill_rays<<<(R + 255) / 256, 256>>>(rays, R);
    float hs[64], hu[64];
    for (int i = 0; i < 64; ++i) { hs[i] = (float)i / 64; hu[i] = (float)i / 63; }
    hipMemcpy(steps, hs, 256, hipMemcpyHostToDevice); hipMemcpy(u, hu, 256, hipMemcpyHostToDevice);
    // coarse pass produces sorted depths for the fine-pass test
    auto fused = [&] { anr_composite_sample(rgbs, nullptr, steps, rays, 8, nullptr, u, 0, R, 64, 64, 1, nullptr, o3, o1, o2, nullptr, zs, nullptr, nullptr); };
    auto fused_z = [&] { anr_composite_sample(rgbs, zs, nullptr, rays, 8, nullptr, u, 0, R, 64, 64, 1, nullptr, o3, o1, o2, nullptr, z, nullptr, nullptr); };
    fused(); hipDeviceSynchronize();
    auto fine = [&] { anr_composite(rgbs, zs, rays, 8, nullptr, R, 128, 1, nullptr, o3, o1, o2, nullptr); };
    auto coarse_w = [&] { anr_composite(rgbs, zs, rays, 8, nullptr, R, 64, 1, w, o3, o1, o2, nullptr); };
    auto merge = [&] { anr_sample_fine_merge(zs, w, u, 0, R, 64, 64, nullptr, z, nullptr, nullptr); };
    double b;
#define RUN(name, fn, bytes) { float ms = time_it(fn, 20); b = (bytes); printf("%-40s %.3f ms  %.0f GB/s\n", name, ms, b / ms / 1e6); }
    RUN("composite K=128", fine, (double)R * (128 * 20 + 28));
    RUN("composite K=64 + weights", coarse_w, (double)R * (64 * 24 + 28));
    RUN("sample_fine_merge 64+64", merge, (double)R * (64 * 8 + 128 * 4));
    RUN("composite_sample 64+64 (steps)", fused, (double)R * (64 * 16 + 28 + 128 * 4));
    RUN("composite_sample 64+64 (z array)", fused_z, (double)R * (64 * 20 + 28 + 128 * 4));
    return 0;
}
