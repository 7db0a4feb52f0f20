// Experiment harness (not part of the library): streaming-rate variants of the per-ray compositor.
// hipcc --offload-arch=gfx950 -O3 -o build/exp_composite tools/exp/exp_composite.hip ; ./build/exp_composite
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include "../../anim-nerf_amd/csrc/composite.hip"
namespace anr { int fail(int code, const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); return code; } }
#pragma clang fp contract(off)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

namespace ex {
template <int W> __device__ __forceinline__ float seg_excl_prod(float v, int l) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < W; o <<= 1) { float t = __shfl_up(inc, o, W); if (l >= o) inc *= t; }
    float ex = __shfl_up(inc, 1, W);
    return l == 0 ? 1.0f : ex;
}
template <int W> __device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace ex
using ex::seg_excl_prod;
template <int W> __device__ __forceinline__ float seg_sum(float v) { return ex::seg_sum<W>(v); }

// A: the product kernel's body (contiguous S samples per lane)
template <int S, int LPR, int WPB>
__global__ __launch_bounds__(64 * WPB) void comp_a(const float4* __restrict__ rgbs, const float* __restrict__ z,
                                                   const float* __restrict__ rays, int64_t R, int K,
                                                   float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int64_t r_raw = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool active = r_raw < R;
    const int64_t r = active ? r_raw : R - 1;
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    float alpha[S], tr[S], zz[S]; float4 col[S]; float prod = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = l * S + s;
        alpha[s] = 0.f; tr[s] = 1.f; zz[s] = 0.f; col[s] = make_float4(0, 0, 0, 0);
        if (k < K) {
            col[s] = c[k]; zz[s] = zr[k];
            float delta = (k + 1 < K) ? (zr[k + 1] - zz[s]) : 1e10f;
            alpha[s] = 1.0f - expf(-delta * fmaxf(col[s].w, 0.0f));
            tr[s] = prod; prod = prod * (1.0f - alpha[s] + 1e-10f);
        }
    }
    const float before = seg_excl_prod<LPR>(prod, l);
    float wsum = 0, cr = 0, cg = 0, cb = 0, dep = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = l * S + s;
        if (k < K) { float w = alpha[s] * (before * tr[s]); wsum += w; cr += w * col[s].x; cg += w * col[s].y; cb += w * col[s].z; dep += w * zz[s]; }
    }
    wsum = seg_sum<LPR>(wsum); cr = seg_sum<LPR>(cr); cg = seg_sum<LPR>(cg); cb = seg_sum<LPR>(cb); dep = seg_sum<LPR>(dep);
    if (l == 0 && active) {
        float far = rays[r * 8 + 7];
        dep = dep + (1.0f - wsum) * far; cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
        rgb_out[r * 3 + 0] = cr; rgb_out[r * 3 + 1] = cg; rgb_out[r * 3 + 2] = cb; depth_out[r] = dep; acc_out[r] = wsum;
    }
}

// B: read-only upper bounds.  PATTERN 0: lane-contiguous 16*S bytes (as A); 1: coalesced (sample j*LPR + l)
template <int S, int LPR, int WPB, int PATTERN>
__global__ __launch_bounds__(64 * WPB) void read_only(const float4* __restrict__ rgbs, const float* __restrict__ z, int64_t R, int K,
                                                     float* __restrict__ acc_out) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int64_t r_raw = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int64_t r = r_raw < R ? r_raw : R - 1;
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    float acc = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = PATTERN == 0 ? l * S + s : s * LPR + l;
        if (k < K) { float4 v = c[k]; acc += v.x + v.y + v.z + v.w + zr[k]; }
    }
    acc = seg_sum<LPR>(acc);
    if (l == 0 && r_raw < R) acc_out[r] = acc;
}

// C: coalesced loads, strided ownership (sample j*LPR + l), one segmented scan per chunk of LPR samples
template <int S, int LPR, int WPB, bool RELOAD = false, bool USE_DPP = false, bool DPP_SUM = false>
__global__ __launch_bounds__(64 * WPB) void comp_c(const float4* __restrict__ rgbs, const float* __restrict__ z,
                                                   const float* __restrict__ rays, int64_t R, int K,
                                                   float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int64_t r_raw = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool active = r_raw < R;
    const int64_t r = active ? r_raw : R - 1;
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    float4 col[S]; float zz[S], zn[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = s * LPR + l;
        col[s] = make_float4(0, 0, 0, -1e5f); zz[s] = 0.f; zn[s] = 0.f;
        if (k < K) { col[s] = c[k]; zz[s] = zr[k]; }
    }
    float carry = 1.0f, wsum = 0, cr = 0, cg = 0, cb = 0, dep = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = s * LPR + l;
        // next depth: lane l+1 of this chunk, or lane 0 of the next chunk
        float nxt;
        if (RELOAD) { nxt = (k + 1 < K) ? zr[k + 1] : 0.f; }
        else {
            nxt = __shfl_down(zz[s], 1, LPR);
            float first_next = (s + 1 < S) ? __shfl(zz[s + 1], 0, LPR) : 0.f;
            if (l == LPR - 1) nxt = first_next;
        }
        float delta = (k + 1 < K) ? (nxt - zz[s]) : 1e10f;
        float alpha = (k < K) ? 1.0f - expf(-delta * fmaxf(col[s].w, 0.0f)) : 0.0f;
        float t = (k < K) ? (1.0f - alpha + 1e-10f) : 1.0f;
        // inclusive scan of t in the chunk
        float inc = t, ex;
        if (USE_DPP) {
            inc = anr::seg_incl_prod<LPR>(t);
            ex = anr::dpp<anr::DPP_WAVE_SHR1>(1.0f, inc);
        } else {
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) { float u = __shfl_up(inc, o, LPR); if (l >= o) inc *= u; }
            ex = __shfl_up(inc, 1, LPR);
        }
        ex = (l == 0) ? 1.0f : ex;
        float w = alpha * (carry * ex);
        carry = carry * (USE_DPP ? anr::seg_last<LPR>(inc, lane) : __shfl(inc, LPR - 1, LPR));
        wsum += w; cr += w * col[s].x; cg += w * col[s].y; cb += w * col[s].z; dep += w * zz[s];
    }
    if (DPP_SUM) { wsum = anr::seg_incl_sum<LPR>(wsum); cr = anr::seg_incl_sum<LPR>(cr); cg = anr::seg_incl_sum<LPR>(cg); cb = anr::seg_incl_sum<LPR>(cb); dep = anr::seg_incl_sum<LPR>(dep); }
    else { wsum = seg_sum<LPR>(wsum); cr = seg_sum<LPR>(cr); cg = seg_sum<LPR>(cg); cb = seg_sum<LPR>(cb); dep = seg_sum<LPR>(dep); }
    if (l == (DPP_SUM ? LPR - 1 : 0) && active) {
        float far = rays[r * 8 + 7];
        dep = dep + (1.0f - wsum) * far; cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
        rgb_out[r * 3 + 0] = cr; rgb_out[r * 3 + 1] = cg; rgb_out[r * 3 + 2] = cb; depth_out[r] = dep; acc_out[r] = wsum;
    }
}

__global__ void fill(float* p, int64_t n, float lo, float hi, uint32_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = lo + (hi - lo) * (h & 0xffffff) / 16777216.0f; }
}
__global__ void fill_z(float* z, int64_t R, int K) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * K) z[i] = 2.0f + 2.0f * (float)(i % K) / K;
}

template <class F> float time_it(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t R = 1 << 20; const int K = argc > 1 ? atoi(argv[1]) : 128;
    float *rgbs, *z, *rays, *o3, *o1, *o2;
    CK(hipMalloc(&rgbs, R * K * 16)); CK(hipMalloc(&z, R * K * 4)); CK(hipMalloc(&rays, R * 32));
    CK(hipMalloc(&o3, R * 12)); CK(hipMalloc(&o1, R * 4)); CK(hipMalloc(&o2, R * 4));
    fill<<<(R * K * 4 + 255) / 256, 256>>>(rgbs, R * K * 4, -3.f, 3.f, 1);
    fill<<<(R * 8 + 255) / 256, 256>>>(rays, R * 8, 3.f, 4.f, 2);
    fill_z<<<(R * K + 255) / 256, 256>>>(z, R, K);
    CK(hipDeviceSynchronize());
    const double bytes = (double)R * (K * 20 + 8 + 20);
    const float4* c4 = (const float4*)rgbs;
#define RUN(name, expr) { float ms = time_it([&] { expr; }, 20); printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); }
    if (K == 128) {
        RUN("A <4,32> wpb4 (product)", (comp_a<4, 32, 4><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("A <4,32> wpb8", (comp_a<4, 32, 8><<<(R + 15) / 16, 512>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("A <4,32> wpb1", (comp_a<4, 32, 1><<<(R + 1) / 2, 64>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("A <2,64> wpb4 (1 ray/wave)", (comp_a<2, 64, 4><<<(R + 3) / 4, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("B read-only lane-contiguous <4,32>", (read_only<4, 32, 4, 0><<<(R + 7) / 8, 256>>>(c4, z, R, K, o1)));
        RUN("B read-only coalesced <4,32>", (read_only<4, 32, 4, 1><<<(R + 7) / 8, 256>>>(c4, z, R, K, o1)));
        RUN("B read-only coalesced <2,64>", (read_only<2, 64, 4, 1><<<(R + 3) / 4, 256>>>(c4, z, R, K, o1)));
        RUN("C coalesced+chunk scans <4,32> wpb4", (comp_c<4, 32, 4><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C coalesced+chunk scans <2,64> wpb4", (comp_c<2, 64, 4><<<(R + 3) / 4, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C coalesced+chunk scans <4,32> wpb8", (comp_c<4, 32, 8><<<(R + 15) / 16, 512>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C1 reload z", (comp_c<4, 32, 4, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C2 dpp scan", (comp_c<4, 32, 4, false, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C3 dpp scan + dpp sums", (comp_c<4, 32, 4, false, true, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C4 reload + dpp scan + dpp sums", (comp_c<4, 32, 4, true, true, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("P0 product anr_composite", (anr_composite(rgbs, z, rays, 8, nullptr, R, K, 1, nullptr, o3, o1, o2, nullptr)));
    } else {
        RUN("A <2,32> wpb4 (product)", (comp_a<2, 32, 4><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("A <1,64> wpb4", (comp_a<1, 64, 4><<<(R + 3) / 4, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("B read-only lane-contiguous <2,32>", (read_only<2, 32, 4, 0><<<(R + 7) / 8, 256>>>(c4, z, R, K, o1)));
        RUN("B read-only coalesced <2,32>", (read_only<2, 32, 4, 1><<<(R + 7) / 8, 256>>>(c4, z, R, K, o1)));
        RUN("C coalesced+chunk scans <2,32> wpb4", (comp_c<2, 32, 4><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C coalesced <1,64> wpb4", (comp_c<1, 64, 4><<<(R + 3) / 4, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C3 dpp scan + dpp sums <2,32>", (comp_c<2, 32, 4, false, true, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("C4 reload + dpp <2,32>", (comp_c<2, 32, 4, true, true, true><<<(R + 7) / 8, 256>>>(c4, z, rays, R, K, o3, o1, o2)));
        RUN("P0 product anr_composite", (anr_composite(rgbs, z, rays, 8, nullptr, R, K, 1, nullptr, o3, o1, o2, nullptr)));
    }
    return 0;
}
