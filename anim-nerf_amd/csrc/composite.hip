// Per-ray kernels: alpha compositing (a13) and importance sampling + merge (a14).
// One 64-lane wavefront owns one ray; transmittance is a wavefront multiplicative scan
// (DPP/shuffle), reductions are butterfly shuffles.  HBM-bound: 20 B per sample in, 4 out.
#include "anr_common.h"

#pragma clang fp contract(off)

namespace anr {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAXS = ANR_MAX_SAMPLES / WAVE;    // samples per lane, max

// the same over segments of W lanes (W = 32: two rays per wavefront); l = lane inside the segment
template <int W>
__device__ __forceinline__ float seg_excl_prod(float v, int l) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        float t = __shfl_up(inc, o, W);
        if (l >= o) inc *= t;
    }
    float ex = __shfl_up(inc, 1, W);
    return l == 0 ? 1.0f : ex;
}
template <int W>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exclusive multiplicative scan across the wave; returns product of lanes < lane
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc *= t;
    }
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}

__device__ __forceinline__ float wave_excl_sum(float v, int lane, float* total) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    *total = __shfl(inc, 63, 64);
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0f : ex;
}

// reference: models/volume_rendering.py:122-160
// LPR lanes per ray (64: one ray per wavefront; 32: two — for K <= 128 the instruction stream, which is what bounds this
// kernel, then serves two rays), S samples per lane.
template <int S, int LPR>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ rays, int stride,
    const float* __restrict__ noise, int64_t R, int K, int white_bkgd, float* __restrict__ weights_out,
    float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out,
    const uint8_t* __restrict__ valid) {
    constexpr int RPW = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int l = lane % LPR;
    const int64_t r_raw = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool active = r_raw < R;
    if (RPW == 1 && !active) return;
    const int64_t r = active ? r_raw : R - 1;              // an idle half-wave shadows the last ray, stores nothing
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    const uint8_t* vr = valid ? valid + r * K : nullptr;

    float alpha[S], tr[S], zz[S];
    float4 col[S];
    float prod = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = l * S + s;
        alpha[s] = 0.0f; tr[s] = 1.0f; zz[s] = 0.0f; col[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            // a sample the warp found invalid is (0, 0, 0, -1e5) by definition (models/anim_nerf.py:245-290, :305):
            // its rgb-sigma row was never written and is not read
            col[s] = (vr != nullptr && vr[k] == 0) ? make_float4(0.f, 0.f, 0.f, -1e5f) : c[k];
            zz[s] = zr[k];
            float delta = (k + 1 < K) ? (zr[k + 1] - zz[s]) : 1e10f;
            float sg = col[s].w;
            if (noise != nullptr) sg = sg + noise[r * K + k];
            alpha[s] = 1.0f - expf(-delta * fmaxf(sg, 0.0f));
            tr[s] = prod;                                  // local exclusive product
            prod = prod * (1.0f - alpha[s] + 1e-10f);
        }
    }
    const float before = seg_excl_prod<LPR>(prod, l);
    float wsum = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dep = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = l * S + s;
        if (k < K) {
            float w = alpha[s] * (before * tr[s]);
            if (weights_out != nullptr && active) weights_out[r * K + k] = w;
            wsum += w; cr += w * col[s].x; cg += w * col[s].y; cb += w * col[s].z; dep += w * zz[s];
        }
    }
    wsum = seg_sum<LPR>(wsum); cr = seg_sum<LPR>(cr); cg = seg_sum<LPR>(cg); cb = seg_sum<LPR>(cb); dep = seg_sum<LPR>(dep);
    if (l == 0 && active) {
        if (white_bkgd) {
            float far = rays[r * stride + 7];
            dep = dep + (1.0f - wsum) * far;
            cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
        }
        rgb_out[r * 3 + 0] = cr; rgb_out[r * 3 + 1] = cg; rgb_out[r * 3 + 2] = cb;
        depth_out[r] = dep;
        acc_out[r] = wsum;
    }
}

// inclusive sum over lanes >= this lane (suffix), via shuffles
__device__ __forceinline__ float wave_suffix_incl_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    return inc;
}

// Backward of composite_kernel (a16, the part autograd differentiates in models/volume_rendering.py:131-160).
// With G_i = dL/dw_i = g_rgb . c_i + g_depth z_i + g_acc' (+ g_w_i), t_i = 1 - alpha_i + 1e-10:
//   dL/dc_i = w_i g_rgb ;  dL/dalpha_i = G_i T_i - (sum_{j>i} G_j w_j) / t_i ;
//   dL/dsigma_i = dL/dalpha_i * delta_i * exp(-delta_i relu(sigma_i)) * [sigma_i > 0].
// White background folds -sum(g_rgb) - g_depth * far into g_acc'.
template <int S>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_backward_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ rays, int stride,
    const float* __restrict__ noise, int64_t R, int K, int white_bkgd, const float* __restrict__ g_w,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_acc,
    float4* __restrict__ d_rgbs, float* __restrict__ d_z, float* __restrict__ d_far) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    const float gr = g_rgb[r * 3 + 0], gg = g_rgb[r * 3 + 1], gb = g_rgb[r * 3 + 2], gd = g_depth[r];
    float ga = g_acc[r];
    if (white_bkgd) ga = ga - (gr + gg + gb) - gd * rays[r * stride + 7];

    float alpha[S], tr[S], zz[S], delta[S], sg[S];
    float4 col[S];
    float prod = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = lane * S + s;
        alpha[s] = 0.0f; tr[s] = 1.0f; zz[s] = 0.0f; delta[s] = 0.f; sg[s] = 0.f; col[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            col[s] = c[k];
            zz[s] = zr[k];
            delta[s] = (k + 1 < K) ? (zr[k + 1] - zz[s]) : 1e10f;
            sg[s] = col[s].w;
            if (noise != nullptr) sg[s] = sg[s] + noise[r * K + k];
            alpha[s] = 1.0f - expf(-delta[s] * fmaxf(sg[s], 0.0f));
            tr[s] = prod;
            prod = prod * (1.0f - alpha[s] + 1e-10f);
        }
    }
    const float before = wave_excl_prod(prod, lane);
    // G_i w_i per sample, suffix sums
    float G[S], w[S], gw_local = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = lane * S + s;
        w[s] = 0.f; G[s] = 0.f;
        if (k < K) {
            tr[s] = before * tr[s];
            w[s] = alpha[s] * tr[s];
            G[s] = gr * col[s].x + gg * col[s].y + gb * col[s].z + gd * zz[s] + ga;
            if (g_w != nullptr) G[s] += g_w[r * K + k];
            gw_local += G[s] * w[s];
        }
    }
    float after = wave_suffix_incl_sum(gw_local, lane) - gw_local;       // lanes > this lane
    float ddelta[S];                                                      // dL/d(delta_k)
    float wsum = 0.f;
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
        int k = lane * S + s;
        ddelta[s] = 0.f;
        if (k < K) {
            const float t = 1.0f - alpha[s] + 1e-10f;
            const float dalpha = G[s] * tr[s] - after / t;
            const float e = expf(-delta[s] * fmaxf(sg[s], 0.0f));
            const float dsig = (sg[s] > 0.0f) ? dalpha * delta[s] * e : 0.0f;
            d_rgbs[r * K + k] = make_float4(w[s] * gr, w[s] * gg, w[s] * gb, dsig);
            if (k + 1 < K) ddelta[s] = dalpha * fmaxf(sg[s], 0.0f) * e;   // the last delta is the constant 1e10
            after += G[s] * w[s];
            wsum += w[s];
        }
    }
    if (d_z != nullptr) {
        // z_k enters depth (g_depth w_k), delta_k (minus) and delta_{k-1} (plus)
        const float prev_lane = __shfl_up(ddelta[S - 1], 1, 64);          // ddelta of sample lane*S - 1
#pragma unroll
        for (int s = 0; s < S; ++s) {
            int k = lane * S + s;
            if (k < K) {
                const float before_d = (s > 0) ? ddelta[s - 1] : (lane > 0 ? prev_lane : 0.0f);
                d_z[r * K + k] = gd * w[s] - ddelta[s] + before_d;
            }
        }
    }
    if (d_far != nullptr) {
        wsum = wave_sum(wsum);
        if (lane == 0) d_far[r] = white_bkgd ? gd * (1.0f - wsum) : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// a14: importance sampling + merge (models/volume_rendering.py:59-97, :199-207), as a wavefront-segment routine shared
// by the stand-alone kernel (weights from HBM: the training path) and the fused coarse compositor below.
//
// Per ray the segment's LDS holds  zall[KT] (coarse depths, then fine) | cdf[KT] (later: the permutation) |
// wbuf[KT] (coarse weights, later: the sorted depths).  The cdf is accumulated as torch's CPU cumsum accumulates it:
// in double, rounded to float per entry (at::acc_type<float, false> = double) — so that, given the same weights, the
// `denom < eps` branch below takes the reference's side (it flips on the last ulp of the cdf in empty bins).
template <int W>
__device__ __forceinline__ double seg_excl_sum_d(double v, int l) {
    double inc = v;
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        double t = __shfl_up(inc, o, W);
        if (l >= o) inc += t;
    }
    double ex = __shfl_up(inc, 1, W);
    return l == 0 ? 0.0 : ex;
}

template <int KT> struct RayLds {
    float zall[KT];
    float cdf[KT];
    float wbuf[KT];
};

// `sync` = a barrier that makes this segment's LDS writes visible to its other lanes.
// Returns with wbuf[0..K) = sorted depths and (PERM) ((uint8_t*)cdf)[0..K) / ((int*)cdf) = permutation.
template <int LPR, int KT, typename PermT, class Sync>
__device__ __forceinline__ void fine_and_merge(RayLds<KT>& L, int l, const float* __restrict__ u_row, int Kc, int Kf,
                                               float* __restrict__ z_fine_row, bool want_perm, Sync sync) {
    constexpr int MAXS = (KT + LPR - 1) / LPR;
    const float eps = 1e-5f;
    const int nb = Kc - 1;                                // bins and cdf entries
    const int np = Kc - 2;                                // pdf entries
    // pdf over weights[1:-1] + eps; each lane owns a contiguous run of S entries
    const int S = (np + LPR - 1) / LPR;
    float wl[MAXS];
    float loc = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        int i = l * S + s;
        wl[s] = (s < S && i < np) ? (L.wbuf[1 + i] + eps) : 0.f;
        loc += wl[s];
    }
    const float total = seg_sum<LPR>(loc);
    double ploc = 0.0;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) { wl[s] = wl[s] / total; ploc += (double)wl[s]; }
    double run = seg_excl_sum_d<LPR>(ploc, l);
    if (l == 0) L.cdf[0] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        int i = l * S + s;
        if (s < S && i < np) { run += (double)wl[s]; L.cdf[i + 1] = (float)run; }
    }
    sync();

    for (int j = l; j < Kf; j += LPR) {
        const float uu = u_row[j];
        int lo = 0, hi = nb;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (L.cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
        const int below = max(lo - 1, 0), above = min(lo, Kc - 2);
        const float c0 = L.cdf[below], c1 = L.cdf[above];
        const float b0 = 0.5f * (L.zall[below] + L.zall[below + 1]), b1 = 0.5f * (L.zall[above] + L.zall[above + 1]);
        float den = c1 - c0;
        if (den < eps) den = 1.0f;
        const float zf = b0 + (uu - c0) / den * (b1 - b0);
        L.zall[Kc + j] = zf;
        if (z_fine_row != nullptr) z_fine_row[j] = zf;
    }
    sync();

    // Stable sort of the Kc+Kf depths: rank(p) = #(y < x) + #(y == x, q < p).  Both halves are normally ascending
    // already (stratified coarse depths; fine depths from ascending u through a monotone inverse cdf), and then the
    // rank is the element's own index plus one binary search in the other half.  Anything else (random u, a 1-ulp
    // inversion at a bin edge) takes the all-pairs count, which is valid for any input.
    const int K = Kc + Kf;
    bool ordered = true;
    for (int p = l; p < K; p += LPR)
        if (p + 1 < K && p + 1 != Kc) ordered &= L.zall[p] <= L.zall[p + 1];
    // (the vote spans the wavefront: with two rays per wavefront both take the general path if either needs it)
    const bool fast = __all(ordered);
    PermT* perm = reinterpret_cast<PermT*>(L.cdf);        // the cdf is dead from here on (barrier after the sampling loop)
    for (int p = l; p < K; p += LPR) {
        const float x = L.zall[p];
        int rank;
        if (fast) {
            if (p < Kc) {                       // + fine entries strictly below x
                int lo = 0, hi = Kf;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.zall[Kc + mid] < x) lo = mid + 1; else hi = mid; }
                rank = p + lo;
            } else {                            // + coarse entries below or equal to x
                int lo = 0, hi = Kc;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.zall[mid] <= x) lo = mid + 1; else hi = mid; }
                rank = (p - Kc) + lo;
            }
        } else {
            rank = 0;
            for (int q = 0; q < K; ++q) {
                const float y = L.zall[q];
                rank += (y < x || (y == x && q < p)) ? 1 : 0;
            }
        }
        L.wbuf[rank] = x;
        if (want_perm) perm[rank] = (PermT)p;               // z_sorted[rank] = cat(z_coarse, z_fine)[p]
    }
    sync();
}

// rows of K floats (+ K permutation entries) from LDS to HBM, 16 bytes per lane where the row allows it
template <int LPR, int KT, typename PermT>
__device__ __forceinline__ void store_sorted(const RayLds<KT>& L, int l, int K, float* __restrict__ zs_row,
                                             PermT* __restrict__ perm_row) {
    if ((K & 3) == 0 && (((uintptr_t)zs_row) & 15) == 0) {
        for (int q = l; q < K / 4; q += LPR)
            reinterpret_cast<float4*>(zs_row)[q] = reinterpret_cast<const float4*>(L.wbuf)[q];
    } else {
        for (int q = l; q < K; q += LPR) zs_row[q] = L.wbuf[q];
    }
    if (perm_row != nullptr) {
        const PermT* perm = reinterpret_cast<const PermT*>(L.cdf);
        if (sizeof(PermT) == 1 && (K & 3) == 0 && (((uintptr_t)perm_row) & 3) == 0) {
            for (int q = l; q < K / 4; q += LPR)
                reinterpret_cast<uint32_t*>(perm_row)[q] = reinterpret_cast<const uint32_t*>(perm)[q];
        } else {
            for (int q = l; q < K; q += LPR) perm_row[q] = perm[q];
        }
    }
}

// Stand-alone importance sampling + merge (weights from HBM).  LPR lanes per ray.
template <int LPR, int KT, typename PermT>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void sample_fine_merge_kernel(
    const float* __restrict__ z_coarse, const float* __restrict__ weights, const float* __restrict__ u,
    int u_per_ray, int64_t R, int Kc, int Kf, float* __restrict__ z_fine_out, float* __restrict__ z_sorted_out,
    PermT* __restrict__ perm_out) {
    constexpr int RPW = WAVE / LPR;
    __shared__ __attribute__((aligned(16))) RayLds<KT> lds[WAVES_PER_BLOCK * RPW];
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int slot = (threadIdx.x >> 6) * RPW + lane / LPR;
    const int64_t r_raw = (int64_t)blockIdx.x * (WAVES_PER_BLOCK * RPW) + slot;
    const bool active = r_raw < R;                        // tail segments compute ray R-1 again, store nothing
    const int64_t r = active ? r_raw : R - 1;
    RayLds<KT>& L = lds[slot];
    for (int k = l; k < Kc; k += LPR) { L.zall[k] = z_coarse[r * Kc + k]; L.wbuf[k] = weights[r * Kc + k]; }
    auto sync = [] { __syncthreads(); };
    sync();
    const int K = Kc + Kf;
    fine_and_merge<LPR, KT, PermT>(L, l, u_per_ray ? u + r * Kf : u, Kc, Kf,
                                   (active && z_fine_out != nullptr) ? z_fine_out + r * Kf : nullptr, perm_out != nullptr, sync);
    if (active) store_sorted<LPR, KT, PermT>(L, l, K, z_sorted_out + r * K, perm_out ? perm_out + r * K : nullptr);
}

// ---------------------------------------------------------------------------------------------
// Fused coarse pass (inference): compositing of the Kc coarse samples AND the importance sampling + merge that
// consumes its weights — the weights and the coarse depths never travel through HBM (-1.0 KB per ray and one launch
// against composite + sample_fine_merge).  z_coarse == nullptr: the depths are the deterministic stratified ones,
// z_k = near (1 - steps_k) + far steps_k (models/volume_rendering.py:43-44), computed here with the same roundings as
// anr_sample_coarse.
template <int S, int LPR, int KT, typename PermT>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_sample_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ steps,
    const float* __restrict__ rays, int stride, const uint8_t* __restrict__ valid, const float* __restrict__ u,
    int u_per_ray, int64_t R, int Kc, int Kf, int white_bkgd, float* __restrict__ weights_out,
    float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out,
    float* __restrict__ z_fine_out, float* __restrict__ z_sorted_out, PermT* __restrict__ perm_out) {
    constexpr int RPW = WAVE / LPR;
    __shared__ __attribute__((aligned(16))) RayLds<KT> lds[WAVES_PER_BLOCK * RPW];
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int slot = (threadIdx.x >> 6) * RPW + lane / LPR;
    const int64_t r_raw = (int64_t)blockIdx.x * (WAVES_PER_BLOCK * RPW) + slot;
    const bool active = r_raw < R;
    const int64_t r = active ? r_raw : R - 1;
    RayLds<KT>& L = lds[slot];
    const float4* c = rgbs + r * Kc;
    const uint8_t* vr = valid ? valid + r * Kc : nullptr;
    const float near = rays[r * stride + 6], far = rays[r * stride + 7];
    auto depth = [&](int k) { if (z) return z[r * Kc + k]; const float s = steps[k]; return near * (1.0f - s) + far * s; };

    float alpha[S], tr[S], zz[S];
    float4 col[S];
    float prod = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = l * S + s;
        alpha[s] = 0.0f; tr[s] = 1.0f; zz[s] = 0.0f; col[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < Kc) {
            col[s] = (vr != nullptr && vr[k] == 0) ? make_float4(0.f, 0.f, 0.f, -1e5f) : c[k];
            zz[s] = depth(k);
            L.zall[k] = zz[s];
            const float delta = (k + 1 < Kc) ? (depth(k + 1) - zz[s]) : 1e10f;
            alpha[s] = 1.0f - expf(-delta * fmaxf(col[s].w, 0.0f));
            tr[s] = prod;
            prod = prod * (1.0f - alpha[s] + 1e-10f);
        }
    }
    const float before = seg_excl_prod<LPR>(prod, l);
    float wsum = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dep = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = l * S + s;
        if (k < Kc) {
            const float w = alpha[s] * (before * tr[s]);
            L.wbuf[k] = w;
            if (weights_out != nullptr && active) weights_out[r * Kc + k] = w;
            wsum += w; cr += w * col[s].x; cg += w * col[s].y; cb += w * col[s].z; dep += w * zz[s];
        }
    }
    wsum = seg_sum<LPR>(wsum); cr = seg_sum<LPR>(cr); cg = seg_sum<LPR>(cg); cb = seg_sum<LPR>(cb); dep = seg_sum<LPR>(dep);
    if (l == 0 && active) {
        if (white_bkgd) {
            dep = dep + (1.0f - wsum) * far;
            cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
        }
        rgb_out[r * 3 + 0] = cr; rgb_out[r * 3 + 1] = cg; rgb_out[r * 3 + 2] = cb;
        depth_out[r] = dep;
        acc_out[r] = wsum;
    }
    auto sync = [] { __syncthreads(); };
    sync();
    const int K = Kc + Kf;
    fine_and_merge<LPR, KT, PermT>(L, l, u_per_ray ? u + r * Kf : u, Kc, Kf,
                                   (active && z_fine_out != nullptr) ? z_fine_out + r * Kf : nullptr, perm_out != nullptr, sync);
    if (active) store_sorted<LPR, KT, PermT>(L, l, K, z_sorted_out + r * K, perm_out ? perm_out + r * K : nullptr);
}

}  // namespace anr

using namespace anr;

extern "C" int anr_composite(const float* rgbs, const float* z, const float* rays, int stride, const float* noise,
                             int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out,
                             float* depth_out, float* acc_out, void* stream) {
    return anr_composite_masked(rgbs, z, rays, stride, noise, nullptr, R, K, white_bkgd, weights_out, rgb_out, depth_out,
                                acc_out, stream);
}

extern "C" int anr_composite_masked(const float* rgbs, const float* z, const float* rays, int stride, const float* noise,
                                    const uint8_t* valid, int64_t R, int K, int white_bkgd, float* weights_out,
                                    float* rgb_out, float* depth_out, float* acc_out, void* stream) {
    ANR_REQUIRE(rgbs && z && rays && rgb_out && depth_out && acc_out, ANR_E_BADARG, "anr_composite: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && stride >= 8, ANR_E_BADARG, "anr_composite: R=%lld K=%d stride=%d", (long long)R, K, stride);
    ANR_REQUIRE(K <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_composite: K=%d > %d", K, ANR_MAX_SAMPLES);
    ANR_REQUIRE(((uintptr_t)rgbs & 15) == 0, ANR_E_ALIGN, "anr_composite: rgbs must be 16-B aligned");
    const float4* c = reinterpret_cast<const float4*>(rgbs);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(WAVE * WAVES_PER_BLOCK);
#define ANR_LAUNCH_COMPOSITE(SS, LPR)                                                                              \
    hipLaunchKernelGGL((composite_kernel<SS, LPR>), dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) /       \
                                                                    (WAVES_PER_BLOCK * (WAVE / LPR)))),              \
                       block, 0, st, c, z, rays, stride, noise, R, K, white_bkgd, weights_out, rgb_out, depth_out,   \
                       acc_out, valid)
    if (K <= 64)       { ANR_LAUNCH_COMPOSITE(2, 32); }         // two rays per wavefront
    else if (K <= 128) { ANR_LAUNCH_COMPOSITE(4, 32); }
    else if (K <= 192) { ANR_LAUNCH_COMPOSITE(3, 64); }
    else               { ANR_LAUNCH_COMPOSITE(4, 64); }
#undef ANR_LAUNCH_COMPOSITE
    return check_launch("anr_composite");
}

extern "C" int anr_composite_backward(const float* rgbs, const float* z, const float* rays, int stride,
                                      const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                      const float* g_rgb, const float* g_depth, const float* g_acc, float* d_rgbs,
                                      float* d_z, float* d_far, void* stream) {
    ANR_REQUIRE(rgbs && z && rays && g_rgb && g_depth && g_acc && d_rgbs, ANR_E_BADARG, "anr_composite_backward: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && stride >= 8, ANR_E_BADARG, "anr_composite_backward: R=%lld K=%d stride=%d", (long long)R, K, stride);
    ANR_REQUIRE(K <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_composite_backward: K=%d > %d", K, ANR_MAX_SAMPLES);
    ANR_REQUIRE((((uintptr_t)rgbs | (uintptr_t)d_rgbs) & 15) == 0, ANR_E_ALIGN, "anr_composite_backward: rgbs/d_rgbs must be 16-B aligned");
    dim3 grid((unsigned)((R + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), block(WAVE * WAVES_PER_BLOCK);
    const float4* c = reinterpret_cast<const float4*>(rgbs);
    float4* d = reinterpret_cast<float4*>(d_rgbs);
    hipStream_t st = (hipStream_t)stream;
#define ANR_LAUNCH_CB(SS)                                                                                      \
    hipLaunchKernelGGL(composite_backward_kernel<SS>, grid, block, 0, st, c, z, rays, stride, noise, R, K, white_bkgd, \
                       g_weights, g_rgb, g_depth, g_acc, d, d_z, d_far)
    switch ((K + 63) / 64) {
        case 1: ANR_LAUNCH_CB(1); break;
        case 2: ANR_LAUNCH_CB(2); break;
        case 3: ANR_LAUNCH_CB(3); break;
        default: ANR_LAUNCH_CB(4); break;
    }
#undef ANR_LAUNCH_CB
    return check_launch("anr_composite_backward");
}

template <typename PermT>
static int launch_merge(const float* z_coarse, const float* weights, const float* u, int u_per_ray, int64_t R, int Kc, int Kf,
                        float* z_fine_out, float* z_sorted_out, PermT* perm_out, hipStream_t st) {
    dim3 block(WAVE * WAVES_PER_BLOCK);
#define ANR_LAUNCH_MERGE(LPR, KT)                                                                                  \
    hipLaunchKernelGGL((sample_fine_merge_kernel<LPR, KT, PermT>),                                                  \
                       dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) / (WAVES_PER_BLOCK * (WAVE / LPR)))), \
                       block, 0, st, z_coarse, weights, u, u_per_ray, R, Kc, Kf, z_fine_out, z_sorted_out, perm_out)
    if (Kc <= 128) {                                      // two rays per wavefront
        if (Kc + Kf <= 128) { ANR_LAUNCH_MERGE(32, 128); } else { ANR_LAUNCH_MERGE(32, ANR_MAX_SAMPLES); }
    } else {
        ANR_LAUNCH_MERGE(64, ANR_MAX_SAMPLES);
    }
#undef ANR_LAUNCH_MERGE
    return check_launch("anr_sample_fine_merge");
}

extern "C" int anr_sample_fine_merge(const float* z_coarse, const float* weights, const float* u, int u_per_ray,
                                     int64_t R, int Kc, int Kf, float* z_fine_out, float* z_sorted_out,
                                     int32_t* perm_out, void* stream) {
    ANR_REQUIRE(z_coarse && weights && u && z_sorted_out, ANR_E_BADARG, "anr_sample_fine_merge: null pointer");
    ANR_REQUIRE(R > 0 && Kc >= 3 && Kf > 0, ANR_E_BADARG, "anr_sample_fine_merge: R=%lld Kc=%d Kf=%d", (long long)R, Kc, Kf);
    ANR_REQUIRE(Kc + Kf <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_sample_fine_merge: Kc+Kf=%d > %d", Kc + Kf, ANR_MAX_SAMPLES);
    return launch_merge<int32_t>(z_coarse, weights, u, u_per_ray, R, Kc, Kf, z_fine_out, z_sorted_out, perm_out, (hipStream_t)stream);
}

extern "C" int anr_sample_fine_merge_u8(const float* z_coarse, const float* weights, const float* u, int u_per_ray,
                                        int64_t R, int Kc, int Kf, float* z_fine_out, float* z_sorted_out,
                                        uint8_t* perm_out, void* stream) {
    ANR_REQUIRE(z_coarse && weights && u && z_sorted_out && perm_out, ANR_E_BADARG, "anr_sample_fine_merge_u8: null pointer");
    ANR_REQUIRE(R > 0 && Kc >= 3 && Kf > 0, ANR_E_BADARG, "anr_sample_fine_merge_u8: R=%lld Kc=%d Kf=%d", (long long)R, Kc, Kf);
    ANR_REQUIRE(Kc + Kf <= ANR_MAX_SAMPLES && ANR_MAX_SAMPLES <= 256, ANR_E_SHAPE, "anr_sample_fine_merge_u8: Kc+Kf=%d > %d", Kc + Kf, ANR_MAX_SAMPLES);
    return launch_merge<uint8_t>(z_coarse, weights, u, u_per_ray, R, Kc, Kf, z_fine_out, z_sorted_out, perm_out, (hipStream_t)stream);
}

extern "C" int anr_composite_sample(const float* rgbs, const float* z_coarse, const float* steps, const float* rays,
                                    int stride, const uint8_t* valid, const float* u, int u_per_ray, int64_t R, int Kc,
                                    int Kf, int white_bkgd, float* weights_out, float* rgb_out, float* depth_out,
                                    float* acc_out, float* z_fine_out, float* z_sorted_out, uint8_t* perm_out,
                                    void* stream) {
    ANR_REQUIRE(rgbs && (z_coarse || steps) && rays && u && rgb_out && depth_out && acc_out && z_sorted_out, ANR_E_BADARG,
                "anr_composite_sample: null pointer");
    ANR_REQUIRE(R > 0 && Kc >= 3 && Kf > 0 && stride >= 8, ANR_E_BADARG, "anr_composite_sample: R=%lld Kc=%d Kf=%d stride=%d",
                (long long)R, Kc, Kf, stride);
    ANR_REQUIRE(Kc + Kf <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_composite_sample: Kc+Kf=%d > %d", Kc + Kf, ANR_MAX_SAMPLES);
    ANR_REQUIRE(((uintptr_t)rgbs & 15) == 0, ANR_E_ALIGN, "anr_composite_sample: rgbs must be 16-B aligned");
    const float4* c = reinterpret_cast<const float4*>(rgbs);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(WAVE * WAVES_PER_BLOCK);
#define ANR_LAUNCH_CS(SS, LPR, KT)                                                                                   \
    hipLaunchKernelGGL((composite_sample_kernel<SS, LPR, KT, uint8_t>),                                               \
                       dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) / (WAVES_PER_BLOCK * (WAVE / LPR)))), \
                       block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd,    \
                       weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out)
    // (S, LPR) by Kc exactly as anr_composite picks them, LPR of the sampling stage by Kc as anr_sample_fine_merge does:
    // the fused launch returns the bits of the two separate ones
    if (Kc <= 64)       { if (Kc + Kf <= 128) { ANR_LAUNCH_CS(2, 32, 128); } else { ANR_LAUNCH_CS(2, 32, ANR_MAX_SAMPLES); } }
    else if (Kc <= 128) { if (Kc + Kf <= 128) { ANR_LAUNCH_CS(4, 32, 128); } else { ANR_LAUNCH_CS(4, 32, ANR_MAX_SAMPLES); } }
    else if (Kc <= 192) { ANR_LAUNCH_CS(3, 64, ANR_MAX_SAMPLES); }
    else                { ANR_LAUNCH_CS(4, 64, ANR_MAX_SAMPLES); }
#undef ANR_LAUNCH_CS
    return check_launch("anr_composite_sample");
}
