// Experiment harness (not part of the library): do 16-byte row stores overlap with a dependent MFMA chain on the same
// wave?  8 waves x 32 rows per workgroup, one workgroup per CU, per out-tile NMMA v_mfma_f32_32x32x16_bf16 and two stores
// (tile-blocked layout), as in the training forward.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp_store_mfma tools/exp/exp_store_mfma.hip ; ./build/exp_store_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILES = 76;

// STORES: 0 none, 1 two per tile behind the 2nd and 4th MFMA, 2 two per tile after the chain
// CONV: the stored data is the packed accumulator of the previous tile (as in the kernels) instead of constants
template <int NMMA, int STORES, bool CONV>
__global__ __launch_bounds__(512, 2) void k(char* __restrict__ out, int64_t rows, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    const int64_t n_groups = rows / 256;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    f32x16 acc[2];
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t row0 = g * 256 + wave * 32;
#pragma unroll 2
        for (int t = 0; t < TILES; ++t) {
            f32x16& cur = acc[t & 1];
            const f32x16& prev = acc[(t & 1) ^ 1];
            auto st = [&](int piece) {
                uint4 v;
                if (CONV) {
                    v.x = __float_as_uint(prev[4 * piece]) >> 16 | (__float_as_uint(prev[4 * piece + 1]) & 0xffff0000u);
                    v.y = __float_as_uint(prev[4 * piece + 2]) >> 16 | (__float_as_uint(prev[4 * piece + 3]) & 0xffff0000u);
                    v.z = __float_as_uint(prev[8 + 4 * piece]) >> 16 | (__float_as_uint(prev[9 + 4 * piece]) & 0xffff0000u);
                    v.w = __float_as_uint(prev[10 + 4 * piece]) >> 16 | (__float_as_uint(prev[11 + 4 * piece]) & 0xffff0000u);
                } else {
                    v = make_uint4(t, lane, (unsigned)g, piece);
                }
                *reinterpret_cast<uint4*>(out + (int64_t)t * rows * 64 + (row0 + i) * 64 + piece * 32 + h * 16) = v;
            };
#pragma unroll
            for (int m = 0; m < NMMA; ++m) {
                cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, cur, 0, 0, 0);
                if (STORES == 1 && m == 1) { st(0); __builtin_amdgcn_sched_barrier(0); }
                if (STORES == 1 && m == 3) { st(1); __builtin_amdgcn_sched_barrier(0); }
            }
            if (STORES == 2) { st(0); st(1); }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
    if (s == 12345.678f) *sink = s;
}

template <int NMMA, int STORES, bool CONV> int run(char* buf, int64_t rows, float* sink, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NMMA, STORES, CONV>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<NMMA, STORES, CONV>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-40s mfma/tile %2d stores %d conv %d: %7.3f ms\n", name, NMMA, STORES, (int)CONV, ms);
    return 0;
}

int main() {
    const int64_t rows = 1 << 20;
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)rows * 5184));
    CK(hipMalloc(&sink, 4));
    run<16, 0, false>(buf, rows, sink, "mfma only");
    run<16, 1, false>(buf, rows, sink, "mfma + stores in the chain");
    run<16, 2, false>(buf, rows, sink, "mfma + stores after the chain");
    run<16, 1, true>(buf, rows, sink, "mfma + stores of the previous acc");
    run<16, 2, true>(buf, rows, sink, "mfma + stores of the prev acc, after");
    run<0, 2, false>(buf, rows, sink, "stores only");
    run<8, 0, false>(buf, rows, sink, "mfma only");
    run<8, 1, false>(buf, rows, sink, "mfma + stores in the chain");
    run<8, 1, true>(buf, rows, sink, "mfma + stores of the previous acc");
    return 0;
}
