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
