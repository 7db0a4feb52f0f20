// Per-ray kernels: alpha compositing (a13) and importance sampling + merge (a14).
// One 64-lane wavefront owns one ray; transmittance is a wavefront multiplicative scan
// (DPP/shuffle), reductions are butterfly shuffles.  HBM-bound: 20 B per sample in, 4 out.
#include "composite_core.h"

#pragma clang fp contract(off)

namespace anr {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAXS = ANR_MAX_SAMPLES / WAVE;    // samples per lane, max

// MASKED: validity bytes given; NOISY: sigma noise given (training) — compile-time, so that the S row loads of a lane are
// issued back to back instead of behind a (uniform) branch each.
// INDEXED (training, the compacted network pass): rgbs = the rows of the VALID samples only, pos[R*K] = a sample's row or -1
// (a sample the warp found invalid — (0, 0, 0, -1e5) as above): the pass's output is read where the network left it instead
// of being expanded to one row per sample first (anr_expand_rows: a launch and 2 x 16 B per sample).
template <int S, int LPR, bool MASKED, bool NOISY, bool INDEXED = false>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ rays, int stride,
    const float* __restrict__ noise, int64_t R, int K, int white_bkgd, float* __restrict__ weights_out,
    float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out,
    const uint8_t* __restrict__ valid, const int32_t* __restrict__ pos = nullptr) {
    constexpr int RPW = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int l = lane % LPR;
    const int64_t r_raw = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool active = r_raw < R;
    if (RPW == 1 && !active) return;
    const int64_t r = active ? r_raw : R - 1;              // an idle half-wave shadows the last ray, stores nothing
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    const uint8_t* vr = valid + r * K;
    const int32_t* pr = pos + r * K;
    const float* nr = noise + r * K;
    float w[S], zz[S], wsum, cr, cg, cb, dep;
    // a sample the warp found invalid is (0, 0, 0, -1e5) by definition (models/anim_nerf.py:245-290, :305): its rgb-sigma
    // row was never written and is not read
    uint8_t vb[S];
    bool empty = false;
    if (MASKED) {                                          // the validity bytes first: a wavefront whose rays miss the body
        unsigned seen = 0;                                 // altogether (most of a frame's) has nothing to composite
#pragma unroll
        for (int s = 0; s < S; ++s) { vb[s] = (s * LPR + l < K) ? vr[s * LPR + l] : (uint8_t)0; seen |= vb[s]; }
        empty = !__any(seen != 0);
    }
    if (empty) {
        // all sigma = -1e5: alpha = 1 - exp(-delta * 0) = 0, every weight 0 * T = 0 and every sum 0 — the values the general
        // path arrives at, without its loads, exponentials and scans
#pragma unroll
        for (int s = 0; s < S; ++s) w[s] = 0.0f;
        wsum = 0.f; cr = 0.f; cg = 0.f; cb = 0.f; dep = 0.f;
    } else {
        composite_ray<S, LPR>(lane, K,
                              [&](int k, int s) {
                                  if (INDEXED) { const int p = pr[k]; return p < 0 ? make_float4(0.f, 0.f, 0.f, -1e5f) : rgbs[p]; }
                                  if (MASKED) return vb[s] == 0 ? make_float4(0.f, 0.f, 0.f, -1e5f) : c[k];
                                  return c[k];
                              },
                              [&](int k) { return zr[k]; }, [&](int k) { return NOISY ? nr[k] : 0.0f; }, w, zz, wsum, cr, cg, cb, dep);
    }
    if (weights_out != nullptr && active) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s * LPR + l < K) weights_out[r * K + s * LPR + l] = w[s];
    }
    if (l == LPR - 1 && active) {
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
// pos != NULL: rgbs = the valid samples' rows, looked up through pos as in composite_kernel<INDEXED> (d_rgbs stays one row
// per sample: anr_mlp_head_grad gathers it)
template <int S>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_backward_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ rays, int stride,
    const float* __restrict__ noise, int64_t R, int K, int white_bkgd, const float* __restrict__ g_w,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_acc,
    float4* __restrict__ d_rgbs, float* __restrict__ d_z, float* __restrict__ d_far, const int32_t* __restrict__ pos,
    float4* __restrict__ g4_rows = nullptr, const int32_t* __restrict__ count = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    // g4_rows (with pos): the gradient leaves as the g operand of the network's backward — row pos[sample] of the COMPACTED
    // pass = (dL/d rgb . sigmoid', dL/d sigma), what anr_mlp_head_grad makes of d_rgbs — and nothing is written per sample;
    // the (< 64) padding rows behind the count[0] listed ones are zeroed here
    if (g4_rows != nullptr && count != nullptr && blockIdx.x == 0 && threadIdx.x < 64) {
        const int lo = count[0], hi = count[1];
        if (lo + (int)threadIdx.x < hi) g4_rows[lo + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (r >= R) return;
    const float4* c = rgbs + r * K;
    const float* zr = z + r * K;
    // a NULL upstream gradient is a zero one (autograd hands over None for outputs the loss does not use)
    const float gr = g_rgb ? g_rgb[r * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    float ga = g_acc ? g_acc[r] : 0.f;
    if (white_bkgd) ga = ga - (gr + gg + gb) - gd * rays[r * stride + 7];

    float alpha[S], tr[S], zz[S], delta[S], sg[S];
    float4 col[S];
    int prow[S];
    float prod = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int k = lane * S + s;
        alpha[s] = 0.0f; tr[s] = 1.0f; zz[s] = 0.0f; delta[s] = 0.f; sg[s] = 0.f; col[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        prow[s] = -1;
        if (k < K) {
            if (pos != nullptr) {
                const int p = pos[r * K + k];
                prow[s] = p;
                col[s] = p < 0 ? make_float4(0.f, 0.f, 0.f, -1e5f) : rgbs[p];
            } else {
                col[s] = c[k];
            }
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
            if (g4_rows != nullptr) {
                if (prow[s] >= 0) {
                    const float4 o = col[s];
                    const float ux = w[s] * gr, uy = w[s] * gg, uz = w[s] * gb;
                    g4_rows[prow[s]] = make_float4(ux * o.x * (1.0f - o.x), uy * o.y * (1.0f - o.y), uz * o.z * (1.0f - o.z), dsig);
                }
            } else {
                d_rgbs[r * K + k] = make_float4(w[s] * gr, w[s] * gg, w[s] * gb, dsig);
            }
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

// rows of K floats (+ K permutation entries) from LDS to HBM, 16 bytes per lane where the rows allow it.  `zs` / `perm_out`
// are wave-uniform bases and `row` the ray's index from there (a 32-bit lane offset on a scalar base: no 64-bit vector
// address arithmetic per store).
template <int LPR, int KT, typename PermT>
__device__ __forceinline__ void store_sorted(const RayLds<KT>& L, int l, int K, float* __restrict__ zs, PermT* __restrict__ perm_out,
                                             unsigned row) {
    const unsigned off = row * (unsigned)K;
    if ((K & 3) == 0 && (((uintptr_t)zs) & 15) == 0) {
        for (int q = l; q < K / 4; q += LPR)
            reinterpret_cast<float4*>(zs)[off / 4 + (unsigned)q] = reinterpret_cast<const float4*>(L.wbuf)[q];
    } else {
        for (int q = l; q < K; q += LPR) zs[off + (unsigned)q] = L.wbuf[q];
    }
    if (perm_out != nullptr) {
        const PermT* perm = reinterpret_cast<const PermT*>(L.cdf);
        if (sizeof(PermT) == 1 && (K & 3) == 0 && (((uintptr_t)perm_out) & 3) == 0) {
            for (int q = l; q < K / 4; q += LPR)
                reinterpret_cast<uint32_t*>(perm_out)[off / 4 + (unsigned)q] = reinterpret_cast<const uint32_t*>(perm)[q];
        } else {
            for (int q = l; q < K; q += LPR) perm_out[off + (unsigned)q] = perm[q];
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
    float uu[(KT + LPR - 1) / LPR];
    load_u<LPR, KT>(u_per_ray ? u + r * Kf : u, lane, Kf, uu);
    for (int k = l; k < Kc; k += LPR) { L.zall[k] = z_coarse[r * Kc + k]; L.wbuf[k] = weights[r * Kc + k]; }
    for (int k = l; k <= Kc; k += LPR) L.hist[k] = 0;
    // a ray's LDS segment is touched by the lanes of ONE wavefront only (LPR <= 64 lanes of it): the LDS unit executes a
    // wave's instructions in order, so the compiler fence is all the "barrier" the segment needs — no workgroup barrier
    // that makes four wavefronts wait for each other eight times per ray
    auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    sync();
    const int K = Kc + Kf;
    fine_and_merge<LPR, KT, PermT>(L, lane, uu, Kc, Kf,
                                   (active && z_fine_out != nullptr) ? z_fine_out + r * Kf : nullptr, perm_out != nullptr, sync);
    const int64_t r0 = (int64_t)blockIdx.x * (WAVES_PER_BLOCK * RPW);
    if (active) store_sorted<LPR, KT, PermT>(L, l, K, z_sorted_out + r0 * K, perm_out ? perm_out + r0 * K : nullptr, (unsigned)slot);
}

// ---------------------------------------------------------------------------------------------
// Fused coarse pass (inference): compositing of the Kc coarse samples AND the importance sampling + merge that
// consumes its weights — the weights and the coarse depths never travel through HBM (-1.0 KB per ray and one launch
// against composite + sample_fine_merge).  z_coarse == nullptr: the depths are the deterministic stratified ones,
// z_k = near (1 - steps_k) + far steps_k (models/volume_rendering.py:43-44), computed here with the same roundings as
// anr_sample_coarse.
template <int S, int LPR, int KT, typename PermT, bool MASKED, bool HAS_Z, int KC = 0, int KF = 0>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void composite_sample_kernel(
    const float4* __restrict__ rgbs, const float* __restrict__ z, const float* __restrict__ steps,
    const float* __restrict__ rays, int stride, const uint8_t* __restrict__ valid, const float* __restrict__ u,
    int u_per_ray, int64_t R, int Kc_rt, int Kf_rt, int white_bkgd, float* __restrict__ weights_out,
    float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out,
    float* __restrict__ z_fine_out, float* __restrict__ z_sorted_out, PermT* __restrict__ perm_out) {
    constexpr int RPW = WAVE / LPR;
    const int Kc = KC ? KC : Kc_rt, Kf = KF ? KF : Kf_rt;
    __shared__ __attribute__((aligned(16))) RayLds<KT> lds[WAVES_PER_BLOCK * RPW];
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int slot = (threadIdx.x >> 6) * RPW + lane / LPR;
    // The block's first ray r0 is a scalar, a lane's ray is r0 + rs with rs < 8: every array below is addressed as
    // (wave-uniform base of the block's rays)[32-bit lane offset] — scalar base + vector offset in the load / store itself
    // instead of a 64-bit vector multiply-add per access (round 5: ~40 of the kernel's 500 VALU instructions per wavefront,
    // and the kernel is bound by exactly those: 98 % VALU-busy by the counters).
    const int64_t r0 = (int64_t)blockIdx.x * (WAVES_PER_BLOCK * RPW);
    const int64_t left = R - 1 - r0;                      // >= 0: the grid is ceil(R / rays per block)
    const bool active = slot <= left;                     // tail segments compute ray R-1 again, store nothing
    const unsigned rs = active ? (unsigned)slot : (unsigned)left;
    RayLds<KT>& L = lds[slot];
    const float4* c = rgbs + r0 * Kc;
    const uint8_t* vr = valid + r0 * Kc;
    const float* zr = z + r0 * Kc;
    const float* ray0 = rays + r0 * stride;
    const unsigned ck = rs * (unsigned)Kc;
    const float near = ray0[rs * (unsigned)stride + 6], far = ray0[rs * (unsigned)stride + 7];
    float uu[(KT + LPR - 1) / LPR];
    load_u<LPR, KT, KF>(u_per_ray ? u + r0 * Kf + rs * (unsigned)Kf : u, lane, Kf, uu);
    float w[S], zz[S], wsum, cr, cg, cb, dep;
    auto depth_of = [&](int k) { if (HAS_Z) return zr[ck + (unsigned)k]; const float sk = steps[k]; return near * (1.0f - sk) + far * sk; };
    uint8_t vb[S];
    bool empty = false;
    if (MASKED) {                                          // (see composite_kernel: wavefronts whose rays miss the body)
        unsigned seen = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) { vb[s] = (s * LPR + l < Kc) ? vr[ck + (unsigned)(s * LPR + l)] : (uint8_t)0; seen |= vb[s]; }
        empty = !__any(seen != 0);
    }
    if (empty) {
#pragma unroll
        for (int s = 0; s < S; ++s) { w[s] = 0.0f; zz[s] = (s * LPR + l < Kc) ? depth_of(s * LPR + l) : 0.0f; }
        wsum = 0.f; cr = 0.f; cg = 0.f; cb = 0.f; dep = 0.f;
    } else {
        composite_ray<S, LPR>(lane, Kc,
                              [&](int k, int s) { if (MASKED) return vb[s] == 0 ? make_float4(0.f, 0.f, 0.f, -1e5f) : c[ck + (unsigned)k]; return c[ck + (unsigned)k]; },
                              depth_of, [](int) { return 0.0f; }, w, zz, wsum, cr, cg, cb, dep);
    }
    float* wo = weights_out + r0 * Kc;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = s * LPR + l;
        if (k < Kc) {
            L.zall[k] = zz[s];
            L.wbuf[k] = w[s];
            if (weights_out != nullptr && active) wo[ck + (unsigned)k] = w[s];
        }
    }
    if (KC) {                                             // (static shape: no loop, no exec-mask bookkeeping)
#pragma unroll
        for (int k0 = 0; k0 <= KC; k0 += LPR)
            if (k0 + LPR <= KC + 1 || l <= KC - k0) L.hist[k0 + l] = 0;
    } else {
        for (int k = l; k <= Kc; k += LPR) L.hist[k] = 0;
    }
    if (l == LPR - 1 && active) {
        if (white_bkgd) {
            dep = dep + (1.0f - wsum) * far;
            cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
        }
        float* ro = rgb_out + r0 * 3;
        ro[rs * 3u + 0] = cr; ro[rs * 3u + 1] = cg; ro[rs * 3u + 2] = cb;
        (depth_out + r0)[rs] = dep;
        (acc_out + r0)[rs] = wsum;
    }
    // a ray's LDS segment is touched by the lanes of ONE wavefront only (LPR <= 64 lanes of it): the LDS unit executes a
    // wave's instructions in order, so the compiler fence is all the "barrier" the segment needs — no workgroup barrier
    // that makes four wavefronts wait for each other eight times per ray
    auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    sync();
    const int K = Kc + Kf;
    fine_and_merge<LPR, KT, PermT, KC, KF>(L, lane, uu, Kc, Kf,
                                           (active && z_fine_out != nullptr) ? z_fine_out + r0 * Kf + rs * (unsigned)Kf : nullptr,
                                           perm_out != nullptr, sync);
    if (active) store_sorted<LPR, KT, PermT>(L, l, K, z_sorted_out + r0 * K, perm_out ? perm_out + r0 * K : nullptr, rs);
}

}  // namespace anr

using namespace anr;

extern "C" int anr_composite(const float* rgbs, const float* z, const float* rays, int stride, const float* noise,
                             int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out,
                             float* depth_out, float* acc_out, void* stream) {
    return anr_composite_masked(rgbs, z, rays, stride, noise, nullptr, R, K, white_bkgd, weights_out, rgb_out, depth_out,
                                acc_out, stream);
}

static int composite_any(const float* rgbs, const float* z, const float* rays, int stride, const float* noise, const uint8_t* valid,
                         const int32_t* pos, int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out, float* depth_out,
                         float* acc_out, void* stream);

extern "C" int anr_composite_masked(const float* rgbs, const float* z, const float* rays, int stride, const float* noise,
                                    const uint8_t* valid, int64_t R, int K, int white_bkgd, float* weights_out,
                                    float* rgb_out, float* depth_out, float* acc_out, void* stream) {
    return composite_any(rgbs, z, rays, stride, noise, valid, nullptr, R, K, white_bkgd, weights_out, rgb_out, depth_out, acc_out, stream);
}

extern "C" int anr_composite_indexed(const float* rows, const int32_t* pos, const float* z, const float* rays, int stride,
                                     const float* noise, int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out,
                                     float* depth_out, float* acc_out, void* stream) {
    ANR_REQUIRE(pos, ANR_E_BADARG, "anr_composite_indexed: null pos");
    return composite_any(rows, z, rays, stride, noise, nullptr, pos, R, K, white_bkgd, weights_out, rgb_out, depth_out, acc_out, stream);
}

static int composite_any(const float* rgbs, const float* z, const float* rays, int stride, const float* noise, const uint8_t* valid,
                         const int32_t* pos, int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out, float* depth_out,
                         float* acc_out, void* stream) {
    ANR_REQUIRE(rgbs && z && rays && rgb_out && depth_out && acc_out, ANR_E_BADARG, "anr_composite: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && stride >= 8, ANR_E_BADARG, "anr_composite: R=%lld K=%d stride=%d", (long long)R, K, stride);
    ANR_REQUIRE(K <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_composite: K=%d > %d", K, ANR_MAX_SAMPLES);
    ANR_REQUIRE(((uintptr_t)rgbs & 15) == 0, ANR_E_ALIGN, "anr_composite: rgbs must be 16-B aligned");
    const float4* c = reinterpret_cast<const float4*>(rgbs);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(WAVE * WAVES_PER_BLOCK);
#define ANR_LAUNCH_COMPOSITE_(SS, LPR, M, N)                                                                        \
    hipLaunchKernelGGL((composite_kernel<SS, LPR, M, N>), dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) /  \
                                                                          (WAVES_PER_BLOCK * (WAVE / LPR)))),        \
                       block, 0, st, c, z, rays, stride, noise, R, K, white_bkgd, weights_out, rgb_out, depth_out,   \
                       acc_out, valid)
#define ANR_LAUNCH_COMPOSITE_IDX(SS, LPR, N)                                                                         \
    hipLaunchKernelGGL((composite_kernel<SS, LPR, false, N, true>), dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) / \
                                                                                (WAVES_PER_BLOCK * (WAVE / LPR)))),   \
                       block, 0, st, c, z, rays, stride, noise, R, K, white_bkgd, weights_out, rgb_out, depth_out,    \
                       acc_out, valid, pos)
#define ANR_LAUNCH_COMPOSITE(SS, LPR)                                                                              \
    do {                                                                                                           \
        if (pos && noise)   { ANR_LAUNCH_COMPOSITE_IDX(SS, LPR, true); }                                           \
        else if (pos)       { ANR_LAUNCH_COMPOSITE_IDX(SS, LPR, false); }                                          \
        else if (valid && noise) { ANR_LAUNCH_COMPOSITE_(SS, LPR, true, true); }                                   \
        else if (valid)     { ANR_LAUNCH_COMPOSITE_(SS, LPR, true, false); }                                       \
        else if (noise)     { ANR_LAUNCH_COMPOSITE_(SS, LPR, false, true); }                                       \
        else                { ANR_LAUNCH_COMPOSITE_(SS, LPR, false, false); }                                      \
    } while (0)
    if (K <= 64)       { ANR_LAUNCH_COMPOSITE(2, 32); }         // two rays per wavefront
    else if (K <= 128) { ANR_LAUNCH_COMPOSITE(4, 32); }
    else if (K <= 192) { ANR_LAUNCH_COMPOSITE(3, 64); }
    else               { ANR_LAUNCH_COMPOSITE(4, 64); }
#undef ANR_LAUNCH_COMPOSITE_
#undef ANR_LAUNCH_COMPOSITE_IDX
#undef ANR_LAUNCH_COMPOSITE
    return check_launch("anr_composite");
}

extern "C" int anr_composite_backward(const float* rgbs, const float* z, const float* rays, int stride,
                                      const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                      const float* g_rgb, const float* g_depth, const float* g_acc, float* d_rgbs,
                                      float* d_z, float* d_far, void* stream) {
    return anr_composite_backward_indexed(rgbs, nullptr, z, rays, stride, noise, R, K, white_bkgd, g_weights, g_rgb, g_depth, g_acc,
                                          d_rgbs, d_z, d_far, stream);
}

static int composite_backward_any(const float* rgbs, const int32_t* pos, const float* z, const float* rays, int stride,
                                  const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                  const float* g_rgb, const float* g_depth, const float* g_acc, float* d_rgbs,
                                  float* d_z, float* d_far, float* g4_rows, const int32_t* count, void* stream);

extern "C" int anr_composite_backward_indexed(const float* rgbs, const int32_t* pos, const float* z, const float* rays, int stride,
                                              const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                              const float* g_rgb, const float* g_depth, const float* g_acc, float* d_rgbs,
                                              float* d_z, float* d_far, void* stream) {
    ANR_REQUIRE(d_rgbs, ANR_E_BADARG, "anr_composite_backward: null pointer");
    return composite_backward_any(rgbs, pos, z, rays, stride, noise, R, K, white_bkgd, g_weights, g_rgb, g_depth, g_acc, d_rgbs, d_z, d_far,
                                  nullptr, nullptr, stream);
}

extern "C" int anr_composite_backward_compact(const float* rows, const int32_t* pos, const int32_t* count, const float* z, const float* rays,
                                              int stride, const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                              const float* g_rgb, const float* g_depth, const float* g_acc, float* g4_rows_out,
                                              float* d_z, float* d_far, void* stream) {
    ANR_REQUIRE(pos && count && g4_rows_out, ANR_E_BADARG, "anr_composite_backward_compact: null pointer");
    ANR_REQUIRE(((uintptr_t)g4_rows_out & 15) == 0, ANR_E_ALIGN, "anr_composite_backward_compact: g4_rows_out must be 16-B aligned");
    return composite_backward_any(rows, pos, z, rays, stride, noise, R, K, white_bkgd, g_weights, g_rgb, g_depth, g_acc, g4_rows_out, d_z,
                                  d_far, g4_rows_out, count, stream);
}

static int composite_backward_any(const float* rgbs, const int32_t* pos, const float* z, const float* rays, int stride,
                                  const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                  const float* g_rgb, const float* g_depth, const float* g_acc, float* d_rgbs,
                                  float* d_z, float* d_far, float* g4_rows, const int32_t* count, void* stream) {
    ANR_REQUIRE(rgbs && z && rays && d_rgbs, ANR_E_BADARG, "anr_composite_backward: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && stride >= 8, ANR_E_BADARG, "anr_composite_backward: R=%lld K=%d stride=%d", (long long)R, K, stride);
    ANR_REQUIRE(K <= ANR_MAX_SAMPLES, ANR_E_SHAPE, "anr_composite_backward: K=%d > %d", K, ANR_MAX_SAMPLES);
    ANR_REQUIRE((((uintptr_t)rgbs | (uintptr_t)d_rgbs) & 15) == 0, ANR_E_ALIGN, "anr_composite_backward: rgbs/d_rgbs must be 16-B aligned");
    dim3 grid((unsigned)((R + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), block(WAVE * WAVES_PER_BLOCK);
    const float4* c = reinterpret_cast<const float4*>(rgbs);
    float4* d = reinterpret_cast<float4*>(d_rgbs);
    hipStream_t st = (hipStream_t)stream;
#define ANR_LAUNCH_CB(SS)                                                                                      \
    hipLaunchKernelGGL(composite_backward_kernel<SS>, grid, block, 0, st, c, z, rays, stride, noise, R, K, white_bkgd, \
                       g_weights, g_rgb, g_depth, g_acc, d, d_z, d_far, pos, reinterpret_cast<float4*>(g4_rows), count)
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
#define ANR_LAUNCH_CS_(SS, LPR, KT, M, Z)                                                                            \
    hipLaunchKernelGGL((composite_sample_kernel<SS, LPR, KT, uint8_t, M, Z>),                                         \
                       dim3((unsigned)((R + WAVES_PER_BLOCK * (WAVE / LPR) - 1) / (WAVES_PER_BLOCK * (WAVE / LPR)))), \
                       block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd,    \
                       weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out)
#define ANR_LAUNCH_CS(SS, LPR, KT)                                                                                   \
    do {                                                                                                             \
        if (valid && z_coarse) { ANR_LAUNCH_CS_(SS, LPR, KT, true, true); }                                          \
        else if (valid)        { ANR_LAUNCH_CS_(SS, LPR, KT, true, false); }                                         \
        else if (z_coarse)     { ANR_LAUNCH_CS_(SS, LPR, KT, false, true); }                                         \
        else                   { ANR_LAUNCH_CS_(SS, LPR, KT, false, false); }                                        \
    } while (0)
    // the shipped shapes (64 + 64: BASELINE configs[1..2]; 64 + 32: configs/people_snapshot/*.yml) with every bound a constant
#define ANR_LAUNCH_CS_STATIC(KC, KF)                                                                                 \
    do {                                                                                                             \
        const dim3 grid((unsigned)((R + WAVES_PER_BLOCK * 2 - 1) / (WAVES_PER_BLOCK * 2)));                          \
        if (valid && z_coarse)                                                                                       \
            hipLaunchKernelGGL((composite_sample_kernel<2, 32, 128, uint8_t, true, true, KC, KF>), grid, block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd, weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out); \
        else if (valid)                                                                                              \
            hipLaunchKernelGGL((composite_sample_kernel<2, 32, 128, uint8_t, true, false, KC, KF>), grid, block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd, weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out); \
        else if (z_coarse)                                                                                           \
            hipLaunchKernelGGL((composite_sample_kernel<2, 32, 128, uint8_t, false, true, KC, KF>), grid, block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd, weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out); \
        else                                                                                                         \
            hipLaunchKernelGGL((composite_sample_kernel<2, 32, 128, uint8_t, false, false, KC, KF>), grid, block, 0, st, c, z_coarse, steps, rays, stride, valid, u, u_per_ray, R, Kc, Kf, white_bkgd, weights_out, rgb_out, depth_out, acc_out, z_fine_out, z_sorted_out, perm_out); \
        return check_launch("anr_composite_sample");                                                                 \
    } while (0)
    if (Kc == 64 && Kf == 64) ANR_LAUNCH_CS_STATIC(64, 64);
    if (Kc == 64 && Kf == 32) ANR_LAUNCH_CS_STATIC(64, 32);
    // (S, LPR) by Kc exactly as anr_composite picks them, LPR of the sampling stage by Kc as anr_sample_fine_merge does:
    // the fused launch returns the bits of the two separate ones
    if (Kc <= 64)       { if (Kc + Kf <= 128) { ANR_LAUNCH_CS(2, 32, 128); } else { ANR_LAUNCH_CS(2, 32, ANR_MAX_SAMPLES); } }
    else if (Kc <= 128) { if (Kc + Kf <= 128) { ANR_LAUNCH_CS(4, 32, 128); } else { ANR_LAUNCH_CS(4, 32, ANR_MAX_SAMPLES); } }
    else if (Kc <= 192) { ANR_LAUNCH_CS(3, 64, ANR_MAX_SAMPLES); }
    else                { ANR_LAUNCH_CS(4, 64, ANR_MAX_SAMPLES); }
#undef ANR_LAUNCH_CS
#undef ANR_LAUNCH_CS_
#undef ANR_LAUNCH_CS_STATIC
    return check_launch("anr_composite_sample");
}
