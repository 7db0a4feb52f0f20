// Shared helpers for libanimnerf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/animnerf_hip.h"

namespace anr {

// thread-local last error text (anr_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr int WAVE = 64;

// Zero `bytes` (a multiple of 4) at a 4-byte aligned address: the library's own fill instead of hipMemsetAsync.  A memset
// NODE in a captured HIP graph is what made long replayed training runs end in a GPU memory fault on ROCm 7.2 (DESIGN.md
// section 4.4: replays -> device synchronise -> work on the default stream -> replay wrote to a stale address; a graph of
// framework kernels only, or of this library's kernels without the memset nodes, survives the same sequence).  A kernel node
// carries its arguments in the node itself.
static __global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t words) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    // 16-byte stores on the aligned middle, dwords on the ragged ends
    const size_t head = (4 - (((uintptr_t)p >> 2) & 3)) & 3;
    const size_t h = head < words ? head : words;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const size_t quads = (words - h) / 4;
    for (size_t i = i0; i < quads; i += step) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i0 < h) p[i0] = 0u;
    const size_t tail0 = h + 4 * quads;
    if (i0 < words - tail0) p[tail0 + i0] = 0u;
}

inline int zero_fill(void* p, size_t bytes, hipStream_t st, const char* what) {
    if (bytes == 0) return 0;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return fail(ANR_E_ALIGN, "%s: zero_fill needs 4-byte alignment", what);
    const size_t words = bytes / 4, blocks = (words / 4 + 255) / 256;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st,
                       reinterpret_cast<uint32_t*>(p), words);
    return check_launch(what);
}

// XCD-aware block remap: consecutive logical blocks land on the same XCD (private L2) instead of
// being round-robined over the 8 XCDs.  Bijective for any grid size (cdna guide, T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned NX = 8;
    unsigned xcd = bid % NX, k = bid / NX;
    unsigned q = nwg / NX, r = nwg % NX;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace anr

#define ANR_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return anr::fail((code), __VA_ARGS__); } while (0)
