// Experiment harness (not part of the library): how fast can a CU write the saved-activation rows of the training MLP
// kernels, by store pattern?  Every variant writes the same bytes (rows x 76 tiles x 64 B) from 8 waves x 32 rows per
// workgroup, one persistent workgroup per CU; only WHICH lane writes WHICH 16 bytes changes.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp_store tools/exp/exp_store_patterns.hip ; ./build/exp_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int PITCH = 5184;      // bytes per row (bf16 ACT_PITCH)
constexpr int TILES = 76;

// P = 0: row per lane, 16 B per lane, the two half-waves adjacent (product kernel after T21): 32 rows x 32 B per instruction
// P = 1: 8 lanes = one full 128-B line of one row (pairs of tiles): 8 rows x 128 B per instruction
// P = 2: 4 lanes = 64 B of one row: 16 rows x 64 B per instruction
// P = 3: tile-blocked layout [tile][row][64 B]: the 32 rows of a wave are 2 KB contiguous; lanes as in P = 0
// P = 4: fully contiguous 1 KB per instruction (ceiling)
// P = 5: tile-blocked layout, 4 lanes = one row's 64 B: 16 rows x 64 B = 1 KB contiguous per instruction
// WORK: dependent FMAs per tile (stand-in for the MFMA time between the stores)
template <int P, int WORK>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ out, int64_t rows, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    const int64_t n_groups = rows / 256;
    float acc = lane;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t row0 = g * 256 + wave * 32;
        for (int t = 0; t < TILES; t += 2) {
            uint4 v = make_uint4(t, lane, (unsigned)g, __float_as_uint(acc));
#pragma unroll
            for (int w = 0; w < WORK; ++w) acc = fmaf(acc, 1.0001f, 0.5f);
            if (P == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)      // tile t + (q >> 1), quarter pair q & 1
                    *reinterpret_cast<uint4*>(out + (row0 + i) * PITCH + (t + (q >> 1)) * 64 + (q & 1) * 32 + h * 16) = v;
            } else if (P == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(out + (row0 + 8 * q + (lane >> 3)) * PITCH + t * 64 + (lane & 7) * 16) = v;
            } else if (P == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(out + (row0 + 16 * (q & 1) + (lane >> 2)) * PITCH + (t + (q >> 1)) * 64 + (lane & 3) * 16) = v;
            } else if (P == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(out + (int64_t)(t + (q >> 1)) * rows * 64 + (row0 + i) * 64 + (q & 1) * 32 + h * 16) = v;
            } else if (P == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(out + ((g * 8 + wave) * (TILES / 2) + t / 2) * 4096 + q * 1024 + lane * 16) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(out + (int64_t)(t + (q >> 1)) * rows * 64 + (row0 + 16 * (q & 1) + (lane >> 2)) * 64 + (lane & 3) * 16) = v;
            }
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int P, int WORK> int run(char* buf, int64_t rows, float* sink, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((store_kernel<P, WORK>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((store_kernel<P, WORK>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)rows * TILES * 64;
    printf("%-58s work %4d: %7.3f ms  %7.1f GB/s  %5.1f B/clk/CU at 2.4 GHz\n", name, WORK, ms, bytes / ms / 1e6,
           bytes / ms / 1e6 * 1e9 / 256 / 2.4e9 / 1e3 * 1e3 / 1e3);
    return 0;
}

int main() {
    const int64_t rows = 1 << 20;
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)rows * PITCH));
    CK(hipMalloc(&sink, 4));
#define ALL(W) \
    run<0, W>(buf, rows, sink, "P0 row per lane, 32 rows x 32 B / instr (product)"); \
    run<1, W>(buf, rows, sink, "P1 8 lanes = a 128-B line, 8 rows / instr"); \
    run<2, W>(buf, rows, sink, "P2 4 lanes = 64 B, 16 rows / instr"); \
    run<3, W>(buf, rows, sink, "P3 tile-blocked layout, lanes as P0 (2 KB span / instr)"); \
    run<5, W>(buf, rows, sink, "P5 tile-blocked layout, 4 lanes = a row (1 KB contiguous)"); \
    run<4, W>(buf, rows, sink, "P4 contiguous 1 KB / instr (ceiling)");
    ALL(0)
    ALL(256)
    ALL(1024)
    return 0;
}
