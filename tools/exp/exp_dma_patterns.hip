// Experiment harness (not part of the library), round 5: does the weight-gradient kernel's L2 -> LDS slab stream
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, per-lane source addresses) care whether the 64 lanes of an
// instruction read ONE contiguous KiB (blocks [R][32] bf16: 16 rows x 64 B) or FOUR runs of 256 B (sub-blocks [R][8] bf16:
// 16 rows x 16 B from each of 4 sub-blocks)?  8 waves per workgroup, one workgroup per CU, 3-stage ring of 32 KiB stages as in
// csrc/mlp_wgrad.hip, no MFMAs: the stream alone.
//   PAT 0: piece = 16 rows x 64 B contiguous (lane l -> row l / 4, 16-byte segment l % 4)
//   PAT 1: piece = 4 sub-blocks x 16 rows x 16 B, lane l -> sub-block l % 4, row l / 4        (LDS image unchanged)
//   PAT 2: piece = 4 sub-blocks x 16 rows x 16 B, lane l -> sub-block l / 16, row (l + 4 (l / 16)) % 16 (rotated image)
//   hipcc --offload-arch=gfx950 -O3 -o build/exp_dma_patterns tools/exp/exp_dma_patterns.hip ; ./build/exp_dma_patterns
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__device__ __forceinline__ void dma16(const char* gsrc_lane, unsigned dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc_lane), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc_lane), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}

constexpr int COLS = 512, SR = 32;                 // a stage = 32 rows of 512 bf16 columns (256 of dact + 256 of act) = 32 KiB
constexpr int PIECES = COLS / 32 * (SR / 16);      // 1-KiB pieces per stage: 16 blocks x 2 row groups
constexpr int PPW = PIECES / 8;

template <int PAT, bool NT>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ src, int64_t rows, int splits, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int64_t per = rows / splits, r0 = (int64_t)blockIdx.x * per;
    const int n_stages = (int)(per / SR);
    auto issue = [&](int st) {
        const int64_t row0 = r0 + (int64_t)st * SR;
        const unsigned buf = lds0 + (unsigned)(st % 3) * (PIECES * 1024);
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int piece = wave + p * 8, blk = piece >> 1, rg = piece & 1;
            const char* s;
            if (PAT == 0) s = src + (int64_t)blk * rows * 64 + (row0 + rg * 16 + (lane >> 2)) * 64 + (lane & 3) * 16;
            else if (PAT == 1) s = src + ((int64_t)blk * 4 + (lane & 3)) * rows * 16 + (row0 + rg * 16 + (lane >> 2)) * 16;
            else s = src + ((int64_t)blk * 4 + (lane >> 4)) * rows * 16 + (row0 + rg * 16 + ((lane - 4 * (lane >> 4)) & 15)) * 16;
            dma16<NT>(s, buf + piece * 1024);
        }
    };
    float acc = 0.f;
    if (n_stages > 0) issue(0);
    if (n_stages > 1) issue(1);
    for (int st = 0; st < n_stages; ++st) {
        if (st + 1 < n_stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 2 < n_stages) issue(st + 2);
        acc += *reinterpret_cast<const float*>(lds + (st % 3) * (PIECES * 1024) + threadIdx.x * 16);
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int PAT, bool NT = false> int run(const char* buf, int64_t rows, float* sink, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 3 * PIECES * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<PAT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<PAT, NT>), dim3(256), dim3(512), lds, 0, buf, rows, 256, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<PAT, NT>), dim3(256), dim3(512), lds, 0, buf, rows, 256, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-56s pat %d nt %d: %7.3f ms  %6.2f TB/s\n", name, PAT, (int)NT, ms, (double)rows * COLS * 2 / ms / 1e9);
    return 0;
}

int main() {
    const int64_t rows = 1 << 22;                  // 4 GiB per pass: past the 256 MiB Infinity Cache
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)rows * COLS * 2));
    CK(hipMemset(buf, 1, (size_t)rows * COLS * 2));
    CK(hipMalloc(&sink, 4));
    run<0>(buf, rows, sink, "slab stream, 1 KiB contiguous per instruction");
    run<1>(buf, rows, sink, "slab stream, 4 x 256 B, lanes interleave the sub-blocks");
    run<2>(buf, rows, sink, "slab stream, 4 x 256 B, 16 lanes per sub-block (rotated)");
    run<0>(buf, rows, sink, "slab stream, 1 KiB contiguous per instruction");
    run<0, true>(buf, rows, sink, "slab stream, 1 KiB contiguous, nt");
    run<1, true>(buf, rows, sink, "slab stream, 4 x 256 B interleaved, nt");
    run<1>(buf, rows, sink, "slab stream, 4 x 256 B, lanes interleave the sub-blocks");
    return 0;
}
