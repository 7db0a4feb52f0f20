// Experiment harness (not part of the library), round 5: what does ONE 16-byte-per-lane activation store cost the wave
// that issues it next to its MFMA chain, as a function of the ADDRESS PATTERN of the instruction?
// 8 waves x 32 rows per workgroup, one workgroup per CU, per out-tile 16 v_mfma_f32_32x32x16_bf16 and two stores of the
// previous tile's packed accumulators, as in the training forward (mlp_core.h FragEpi, SAVE).
//   PAT 0: the round-3 layout, blocks [R][32] bf16: an instruction writes 32 rows x 32 B at a 64-B pitch (two 16-B halves of
//          a row come from lanes i and i+32)
//   PAT 1: sub-blocks [R][8] bf16: each half-wave writes 512 contiguous bytes (two spans per instruction)
//   PAT 2: 1 KiB contiguous per instruction, lane l -> byte 16 l (the data is NOT transposed: cost of the pattern alone)
//   PAT 3: PAT 2 with the real transposition through LDS (2 ds_write_b128 + 2 ds_read_b128 per tile, XOR-swizzled)
//   PAT 4: PAT 1 with nontemporal stores;  PAT 5: PAT 0 with nontemporal stores
//   hipcc --offload-arch=gfx950 -O3 -o build/exp_store_mfma2 tools/exp/exp_store_mfma2.hip ; ./build/exp_store_mfma2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILES = 76;

__device__ __forceinline__ u32x4 pack_piece(const f32x16& prev, int piece) {
    u32x4 v;
    v[0] = __float_as_uint(prev[4 * piece]) >> 16 | (__float_as_uint(prev[4 * piece + 1]) & 0xffff0000u);
    v[1] = __float_as_uint(prev[4 * piece + 2]) >> 16 | (__float_as_uint(prev[4 * piece + 3]) & 0xffff0000u);
    v[2] = __float_as_uint(prev[8 + 4 * piece]) >> 16 | (__float_as_uint(prev[9 + 4 * piece]) & 0xffff0000u);
    v[3] = __float_as_uint(prev[10 + 4 * piece]) >> 16 | (__float_as_uint(prev[11 + 4 * piece]) & 0xffff0000u);
    return v;
}

// WHEN: 1 = stores behind the 2nd and 4th MFMA of the tile (as the kernel), 2 = both behind the 4th, 0 = no stores
template <int PAT, int WHEN, int NMMA>
__global__ __launch_bounds__(512, 2) void k(char* __restrict__ out, int64_t rows, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[8 * 2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    const int64_t n_groups = rows / 256;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    f32x16 acc[2];
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    char* my = lds + wave * 2048;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t row0 = g * 256 + wave * 32;
#pragma unroll 2
        for (int t = 0; t < TILES; ++t) {
            f32x16& cur = acc[t & 1];
            const f32x16& prev = acc[(t & 1) ^ 1];
            auto st = [&](int piece) {
                const u32x4 v = pack_piece(prev, piece);
                char* p;
                if (PAT == 0 || PAT == 5) p = out + (int64_t)t * rows * 64 + (row0 + i) * 64 + piece * 32 + h * 16;
                else if (PAT == 1 || PAT == 4) p = out + ((int64_t)t * 4 + piece * 2 + h) * rows * 16 + (row0 + i) * 16;
                else p = out + (int64_t)t * rows * 64 + row0 * 64 + piece * 1024 + lane * 16;
                if (PAT == 3) {
                    // lane (h, i) holds piece 2*piece + h of row i; swizzled so that neither side conflicts
                    *reinterpret_cast<u32x4*>(my + i * 64 + (((piece * 2 + h) ^ ((i >> 1) & 3)) * 16)) = v;
                    if (piece == 1) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const int r = s * 16 + (lane >> 2), q = lane & 3;
                            const u32x4 w = *reinterpret_cast<const u32x4*>(my + r * 64 + ((q ^ ((r >> 1) & 3)) * 16));
                            *reinterpret_cast<u32x4*>(out + (int64_t)t * rows * 64 + row0 * 64 + s * 1024 + lane * 16) = w;
                        }
                    }
                } else if (PAT >= 4) {
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
                } else {
                    *reinterpret_cast<u32x4*>(p) = v;
                }
            };
#pragma unroll
            for (int m = 0; m < NMMA; ++m) {
                cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, cur, 0, 0, 0);
                if (WHEN == 1 && m == 1) { st(0); __builtin_amdgcn_sched_barrier(0); }
                if (WHEN == 1 && m == 3) { st(1); __builtin_amdgcn_sched_barrier(0); }
                if (WHEN == 2 && m == 3) { st(0); st(1); __builtin_amdgcn_sched_barrier(0); }
            }
            if (NMMA == 0 && WHEN) { st(0); st(1); }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
    if (s == 12345.678f) *sink = s;
}

template <int PAT, int WHEN, int NMMA> int run(char* buf, int64_t rows, float* sink, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<PAT, WHEN, NMMA>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<PAT, WHEN, NMMA>), dim3(256), dim3(512), 0, 0, buf, rows, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-52s pat %d when %d mfma/tile %2d: %7.3f ms\n", name, PAT, WHEN, NMMA, ms);
    return 0;
}

int main() {
    const int64_t rows = 1 << 20;
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)rows * 5184));
    CK(hipMalloc(&sink, 4));
    run<0, 0, 16>(buf, rows, sink, "mfma only");
    run<0, 1, 0>(buf, rows, sink, "stores only, blocks [R][32]");
    run<1, 1, 0>(buf, rows, sink, "stores only, sub-blocks [R][8]");
    run<2, 1, 0>(buf, rows, sink, "stores only, 1 KiB contiguous");
    run<3, 1, 0>(buf, rows, sink, "stores only, 1 KiB contiguous via LDS");
    run<4, 1, 0>(buf, rows, sink, "stores only, sub-blocks, nontemporal");
    run<0, 1, 16>(buf, rows, sink, "mfma + stores, blocks [R][32] (round 3)");
    run<1, 1, 16>(buf, rows, sink, "mfma + stores, sub-blocks [R][8]");
    run<2, 1, 16>(buf, rows, sink, "mfma + stores, 1 KiB contiguous (no transpose)");
    run<3, 1, 16>(buf, rows, sink, "mfma + stores, 1 KiB contiguous via LDS");
    run<4, 1, 16>(buf, rows, sink, "mfma + stores, sub-blocks, nontemporal");
    run<5, 1, 16>(buf, rows, sink, "mfma + stores, blocks, nontemporal");
    run<0, 2, 16>(buf, rows, sink, "mfma + store pair, blocks");
    run<1, 2, 16>(buf, rows, sink, "mfma + store pair, sub-blocks");
    run<2, 2, 16>(buf, rows, sink, "mfma + store pair, contiguous");
    run<0, 0, 24>(buf, rows, sink, "mfma only (24 per tile ~ the kernel's pace)");
    run<0, 1, 24>(buf, rows, sink, "mfma 24 + stores, blocks");
    run<1, 1, 24>(buf, rows, sink, "mfma 24 + stores, sub-blocks");
    run<2, 1, 24>(buf, rows, sink, "mfma 24 + stores, contiguous");
    run<4, 1, 24>(buf, rows, sink, "mfma 24 + stores, sub-blocks, nontemporal");
    run<3, 1, 24>(buf, rows, sink, "mfma 24 + stores, contiguous via LDS");
    return 0;
}
