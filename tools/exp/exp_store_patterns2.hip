// Experiment harness (not part of the library), second round: the saved-activation stores of the training MLP kernels as
// they are ISSUED in the kernels — two 16-byte stores per lane and tile, spread over the tile's MFMA time, next to an
// L2 -> CU weight stream of the same size — by layout and grouping.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp_store2 tools/exp/exp_store_patterns2.hip ; ./build/exp_store2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int PITCH = 5184;
constexpr int TILES = 76;

// V = 0: row-major rows, piece A then piece B of a tile half a tile apart (product)
// V = 1: row-major rows, A and B back to back after the tile
// V = 2: tile-blocked [tile][row][64 B], A and B half a tile apart
// V = 3: tile-blocked, A and B back to back
// V = 4: row-major rows, four stores per tile PAIR, 8 lanes = one 128-B line of a row (through an LDS transpose)
// V = 5: tile-blocked in PAIRS [pair][row][128 B], four stores per pair, 8 lanes = a row's 128 B (4 KB contiguous per wave)
// WAITN >= 0: `s_waitcnt vmcnt(WAITN)` after every tile, as the kernels wait for their weight DMA: at most WAITN stores
// per wave stay in flight
template <int V, int WORK, bool WEIGHTS, int WAITN = -1>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ out, int64_t rows, const uint4* __restrict__ wts, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    const int64_t n_groups = rows / 256;
    float acc = lane;
    unsigned x = 0;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t row0 = g * 256 + wave * 32;
        for (int t = 0; t < TILES; ++t) {
            uint4 v = make_uint4(t, lane, (unsigned)g, __float_as_uint(acc) ^ x);
            if (WEIGHTS) {                         // 20 KB per tile and workgroup from a 1.2 MB L2-resident table
                const uint4 w = wts[((t * 20 + wave * 2) * 64 + lane) % (75 * 1024)];
                const uint4 w2 = wts[((t * 20 + wave * 2 + 1) * 64 + lane) % (75 * 1024)];
                x ^= w.x ^ w2.y;
            }
            auto work = [&](int n) {
#pragma unroll 16
                for (int w = 0; w < n; ++w) acc = fmaf(acc, 1.0001f, 0.5f);
            };
            auto rowmajor = [&](int tt, int piece) { return out + (row0 + i) * PITCH + tt * 64 + piece * 32 + h * 16; };
            auto blocked = [&](int tt, int piece) { return out + (int64_t)tt * rows * 64 + (row0 + i) * 64 + piece * 32 + h * 16; };
            if (V == 0) {
                work(WORK / 2); *reinterpret_cast<uint4*>(rowmajor(t, 0)) = v;
                work(WORK / 2); *reinterpret_cast<uint4*>(rowmajor(t, 1)) = v;
            } else if (V == 1) {
                work(WORK); *reinterpret_cast<uint4*>(rowmajor(t, 0)) = v; *reinterpret_cast<uint4*>(rowmajor(t, 1)) = v;
            } else if (V == 2) {
                work(WORK / 2); *reinterpret_cast<uint4*>(blocked(t, 0)) = v;
                work(WORK / 2); *reinterpret_cast<uint4*>(blocked(t, 1)) = v;
            } else if (V == 3) {
                work(WORK); *reinterpret_cast<uint4*>(blocked(t, 0)) = v; *reinterpret_cast<uint4*>(blocked(t, 1)) = v;
            } else if (V == 4) {
                work(WORK);
                if (t & 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint4*>(out + (row0 + 8 * q + (lane >> 3)) * PITCH + (t - 1) * 64 + (lane & 7) * 16) = v;
                }
            } else {
                work(WORK);
                if (t & 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint4*>(out + (int64_t)(t >> 1) * rows * 128 + (row0 + 8 * q + (lane >> 3)) * 128 + (lane & 7) * 16) = v;
                }
            }
            if (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN < 0 ? 0 : WAITN) : "memory");
        }
    }
    if (acc == 12345.678f) *sink = acc + x;
}

template <int V, int WORK, bool WEIGHTS, int WAITN = -1> int run(char* buf, int64_t rows, const uint4* wts, float* sink, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((store_kernel<V, WORK, WEIGHTS, WAITN>), dim3(256), dim3(512), 0, 0, buf, rows, wts, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((store_kernel<V, WORK, WEIGHTS, WAITN>), dim3(256), dim3(512), 0, 0, buf, rows, wts, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)rows * TILES * 64;
    printf("%-64s work %4d weights %d wait %2d: %7.3f ms  %7.1f GB/s\n", name, WORK, (int)WEIGHTS, WAITN, ms, bytes / ms / 1e6);
    return 0;
}

int main() {
    const int64_t rows = 1 << 20;
    char* buf; float* sink; uint4* wts;
    CK(hipMalloc(&buf, (size_t)rows * PITCH));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&wts, 75 * 1024 * 16));
    CK(hipMemset(wts, 1, 75 * 1024 * 16));
#define ALL(W, WT) \
    run<0, W, WT>(buf, rows, wts, sink, "V0 row-major, pieces half a tile apart (product)"); \
    run<1, W, WT>(buf, rows, wts, sink, "V1 row-major, both pieces of a tile together"); \
    run<2, W, WT>(buf, rows, wts, sink, "V2 tile-blocked, pieces half a tile apart"); \
    run<3, W, WT>(buf, rows, wts, sink, "V3 tile-blocked, both pieces together"); \
    run<4, W, WT>(buf, rows, wts, sink, "V4 row-major, 8 lanes = a 128-B line, 4 stores per tile pair"); \
    run<5, W, WT>(buf, rows, wts, sink, "V5 pair-blocked, 8 lanes = a row's 128 B, 4 stores per pair");
    ALL(128, false)
    ALL(128, true)
    // at most WAITN stores per wave in flight (no weight loads: they would be waited for as well)
    run<2, 128, false, 2>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 128, false, 4>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 128, false, 8>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 128, false, 16>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 128, false, 32>(buf, rows, wts, sink, "V2 tile-blocked");
    run<0, 128, false, 4>(buf, rows, wts, sink, "V0 row-major");
    run<0, 128, false, 8>(buf, rows, wts, sink, "V0 row-major");
    run<0, 128, false, 16>(buf, rows, wts, sink, "V0 row-major");
    run<2, 256, false, 4>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 256, false, 8>(buf, rows, wts, sink, "V2 tile-blocked");
    run<2, 256, false, 16>(buf, rows, wts, sink, "V2 tile-blocked");
    return 0;
}
