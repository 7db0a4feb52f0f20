// What the explicit training step (anim_nerf_amd/fused_step.py: forward + backward of train.py:324-348 as a fixed sequence of
// this library's launches, no autograd graph) needs beside the big kernels, so that NOT ONE framework kernel sits between the
// first and the last node of the step's HIP graph:
//
//   anr_train_draws               every random number of a step in one launch: the stratified jitter (models/volume_rendering.py:
//                                 48-54), the sigma noise of both passes (:122-129), the importance sampler's uniforms (:66-70)
//                                 and the normal regulariser's two perturbations of the template vertices (train.py:289-290) —
//                                 Philox4x32-10 keyed by (seed, step counter); the counter lives on the device and a one-thread
//                                 launch behind the draws advances it, so a replayed graph draws fresh numbers
//   anr_gather_frame_params       BodyModelParams.forward (models/body_model_params.py:5-68): rows of the four embedding tables
//                                 -> betas[bs][10], pose[bs][72], transl[bs][3]
//   anr_scatter_frame_param_grads its backward: dL/d(betas | global_orient | body_pose | transl)[bs][85] -> the tables' gradients
//                                 (written whole: rows no frame touched get 0 — fixed order, no atomics)
//   anr_merge_backward2           anr_merge_backward of the SUM of two upstream gradients (warp + compositor of the fine pass), by
//                                 the byte permutation the fine pass's warp copies its coarse rows by
//   anr_sample_coarse_backward_acc  anr_sample_coarse_backward of the sum of up to three upstream gradients, ADDED into the
//                                 accumulated ray gradient together with the two compositors' dL/d far'
//   anr_zero_fill                 the library's memset (anr_common.h: zero_fill)
//   anr_zero_segments             ... of up to 24 buffers in one launch (the step's accumulators and counters, filled up front)
//   anr_add_inplace               dst += src: the normals branch's weight gradients joining the render passes'
#include "anr_common.h"

// (products and sums round separately in this file, as the framework ops they replace round them: `points + randn * scale`)
#pragma clang fp contract(off)

namespace anr {

// ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, stateless
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x, hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}
// 24 random bits -> [0, 1): what torch's uniform_ makes of a 32-bit draw
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
// Box-Muller: two 32-bit draws -> two standard normals (u1 in (0, 1]: no log(0))
__device__ __forceinline__ float2 normal2(unsigned a, unsigned b) {
    // the hardware's transcendental units directly (v_log_f32 = log2, v_sqrt_f32, v_sin_f32 / v_cos_f32 take REVOLUTIONS):
    // ~1e-6 of absolute error, irrelevant for noise; the library calls (IEEE sqrt, sincospif's reduction and polynomials)
    // made this kernel 117 us for 4.2 M values
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // sqrt(-2 ln u1)
    const float t = u01(b);
    return make_float2(r * __builtin_amdgcn_cosf(t), r * __builtin_amdgcn_sinf(t));
}

__global__ __launch_bounds__(256) void train_draws_kernel(uint64_t* __restrict__ state, anr_draw_plan p, int64_t g0, int64_t g1, int64_t g2,
                                                          int64_t g3, int64_t g4) {
    // groups of 4 values: [0, g0) jitter | [g0, g1) coarse noise | [g1, g2) fine uniforms | [g2, g3) fine noise | [g3, g4) points
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t seed = state[0], step = state[1];
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    if (g < g4) {
        const int seg = g < g0 ? 0 : g < g1 ? 1 : g < g2 ? 2 : g < g3 ? 3 : 4;
        const int64_t local = g - (seg == 0 ? 0 : seg == 1 ? g0 : seg == 2 ? g1 : seg == 3 ? g2 : g3);
        // counter = (group index lo, hi | segment << 28, step lo, step hi): disjoint streams per segment and per step
        const uint4 ctr = make_uint4((unsigned)local, (unsigned)(local >> 32) | ((unsigned)seg << 28), (unsigned)step, (unsigned)(step >> 32));
        const uint4 x = philox4x32_10(ctr, key);
        const int64_t e = local * 4;
        if (seg == 0 || seg == 2) {
            float* out = seg == 0 ? p.t_rand : p.u_fine;
            const int64_t n = seg == 0 ? p.n_t : p.n_u;
            const float sc = seg == 0 ? p.t_scale : 1.0f;
            const float v[4] = {u01(x.x) * sc, u01(x.y) * sc, u01(x.z) * sc, u01(x.w) * sc};
            if (e + 3 < n) *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int i = 0; i < 4; ++i) if (e + i < n) out[e + i] = v[i];
        } else if (seg == 1 || seg == 3) {
            float* out = seg == 1 ? p.noise_c : p.noise_f;
            const int64_t n = seg == 1 ? p.n_nc : p.n_nf;
            const float2 a = normal2(x.x, x.y), b = normal2(x.z, x.w);
            const float v[4] = {a.x * p.noise_scale, a.y * p.noise_scale, b.x * p.noise_scale, b.y * p.noise_scale};
            if (e + 3 < n) *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int i = 0; i < 4; ++i) if (e + i < n) out[e + i] = v[i];
        } else {
            // train.py:289-290: points = verts_template + N(0,1) dis_threshold / 2; neighbours = points + N(0,1) epsilon
            const uint4 y = philox4x32_10(make_uint4(ctr.x, ctr.y | (5u << 28), ctr.z, ctr.w), key);
            const float2 a0 = normal2(x.x, x.y), b0 = normal2(x.z, x.w), a1 = normal2(y.x, y.y), b1 = normal2(y.z, y.w);
            const float d0[4] = {a0.x, a0.y, b0.x, b0.y}, d1[4] = {a1.x, a1.y, b1.x, b1.y};
            for (int i = 0; i < 4; ++i)
                if (e + i < p.n_v3) {
                    // Every rounding step pinned to ONE materialised value: the normals that are stored are the normals that are
                    // used, and product and sum round separately, as the framework's `points + randn * scale` rounds them — so that
                    // a caller who recomputes the points from n0 / n1 gets these bits.
                    float a = d0[i], b = d1[i];
                    asm volatile("" : "+v"(a), "+v"(b));
                    float pa = a * p.point_scale, pb = b * p.neighbour_scale;
                    asm volatile("" : "+v"(pa), "+v"(pb));
                    float pt = p.verts_template[e + i] + pa;
                    asm volatile("" : "+v"(pt));
                    const float nb = pt + pb;
                    p.n0[e + i] = a;
                    p.n1[e + i] = b;
                    p.pair[e + i] = pt;
                    p.pair[p.n_v3 + e + i] = nb;
                    if (p.quads != nullptr) {               // component c of point pi (and of its neighbour): four rows each
                        const int64_t pi = (e + i) / 3, pj = p.n_v3 / 3 + pi;
                        const int c = (int)((e + i) - 3 * pi);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            p.quads[(4 * pi + q) * 4 + c] = pt;
                            p.quads[(4 * pj + q) * 4 + c] = nb;
                            if (c == 0) { p.quads[(4 * pi + q) * 4 + 3] = 1.0f; p.quads[(4 * pj + q) * 4 + 3] = 1.0f; }
                        }
                    }
                }
        }
    }
}

// The step counter is advanced by a launch of its own behind the draws (stream order: every thread of the draws kernel has read
// it by then).  A "last workgroup" ticket inside the draws kernel cost 100 of its 118 us: 4,096 workgroups' atomic adds on one
// cache line retire one after the other, ~25 ns each — also when spread over 64 addresses of that line.
__global__ void draws_advance_kernel(uint64_t* __restrict__ state) { state[1] += 1; }

constexpr int FP_COLS = 85;                      // betas 10 | global_orient 3 | body_pose 69 | transl 3
__global__ __launch_bounds__(128) void gather_frame_params_kernel(const int64_t* __restrict__ idx, const float* __restrict__ betas_w,
                                                                  int betas_rows, const float* __restrict__ go_w,
                                                                  const float* __restrict__ bp_w, const float* __restrict__ tr_w,
                                                                  float* __restrict__ betas, float* __restrict__ pose,
                                                                  float* __restrict__ transl) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= FP_COLS) return;
    const int64_t t = idx[b];
    if (c < 10) betas[b * 10 + c] = betas_w[(betas_rows == 1 ? 0 : t) * 10 + c];
    else if (c < 13) pose[b * 72 + (c - 10)] = go_w[t * 3 + (c - 10)];
    else if (c < 82) pose[b * 72 + 3 + (c - 13)] = bp_w[t * 69 + (c - 13)];
    else transl[b * 3 + (c - 82)] = tr_w[t * 3 + (c - 82)];
}

// one thread per table entry (row t, column c): the sum over the frames of the batch that use the row, in batch order
__global__ __launch_bounds__(128) void scatter_frame_param_grads_kernel(const int64_t* __restrict__ idx, const float* __restrict__ grads,
                                                                        int bs, int betas_rows, float* __restrict__ d_betas,
                                                                        float* __restrict__ d_go, float* __restrict__ d_bp,
                                                                        float* __restrict__ d_tr) {
    const int t = blockIdx.x, c = threadIdx.x;
    if (c >= FP_COLS) return;
    float v = 0.0f;
    if (c < 10 && betas_rows == 1) {
        if (t != 0) return;
        for (int b = 0; b < bs; ++b) v += grads[b * FP_COLS + c];
        if (d_betas) d_betas[c] = v;
        return;
    }
    for (int b = 0; b < bs; ++b)
        if (idx[b] == t) v += grads[b * FP_COLS + c];
    if (c < 10) { if (d_betas) d_betas[t * 10 + c] = v; }
    else if (c < 13) { if (d_go) d_go[t * 3 + (c - 10)] = v; }
    else if (c < 82) { if (d_bp) d_bp[t * 69 + (c - 13)] = v; }
    else if (d_tr) d_tr[t * 3 + (c - 82)] = v;
}

__global__ __launch_bounds__(256) void merge_backward2_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                              const uint8_t* __restrict__ perm, int64_t n, int K, int Kc,
                                                              float* __restrict__ d_zc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    if (p < Kc) d_zc[(i / K) * Kc + p] = ga[i] + (gb ? gb[i] : 0.0f);
}

// sample_coarse_backward_kernel (train_glue.hip) on g = ga + gb + gc, added into the accumulated ray gradient.
// One WAVEFRONT per ray (round 5): lane k <-> sample k, k + 64, ...: every load instruction reads consecutive floats of one
// ray, the two sums meet by shuffles.  (One THREAD per ray walked K samples with loads K floats apart from up to four
// arrays: 72 us at 2,048 rays — as long as a pass of the 8-layer network over the same batch.)
__global__ __launch_bounds__(256) void sample_coarse_backward_acc_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                                         const float* __restrict__ gc, const float* __restrict__ steps,
                                                                         const float* __restrict__ t_rand, const float* __restrict__ dfar_a,
                                                                         const float* __restrict__ dfar_b, int64_t R, int K,
                                                                         float* __restrict__ d_rays) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float dn = 0.0f, df = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float s = steps[k];
        float e = s;
        if (t_rand) {
            const float lo = k > 0 ? 0.5f * (s + steps[k - 1]) : s, up = k + 1 < K ? 0.5f * (steps[k + 1] + s) : s;
            e = lo + (up - lo) * t_rand[r * K + k];
        }
        const float gk = ga[r * K + k] + (gb ? gb[r * K + k] : 0.0f) + (gc ? gc[r * K + k] : 0.0f);
        dn += gk * (1.0f - e);
        df += gk * e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dn += __shfl_xor(dn, o, 64); df += __shfl_xor(df, o, 64); }
    if (lane == 0) {
        if (dfar_a) df += dfar_a[r];
        if (dfar_b) df += dfar_b[r];
        d_rays[r * 8 + 6] += dn;
        d_rays[r * 8 + 7] += df;
    }
}

}  // namespace anr

using namespace anr;

extern "C" int anr_train_draws(int64_t* state, const anr_draw_plan* plan, void* stream) {
    ANR_REQUIRE(state && plan, ANR_E_BADARG, "anr_train_draws: null pointer");
    const anr_draw_plan& p = *plan;
    ANR_REQUIRE((p.n_t == 0 || p.t_rand) && (p.n_nc == 0 || p.noise_c) && (p.n_u == 0 || p.u_fine) && (p.n_nf == 0 || p.noise_f) &&
                (p.n_v3 == 0 || (p.verts_template && p.n0 && p.n1 && p.pair)), ANR_E_BADARG, "anr_train_draws: a segment without its buffer");
    ANR_REQUIRE((((uintptr_t)p.t_rand | (uintptr_t)p.noise_c | (uintptr_t)p.u_fine | (uintptr_t)p.noise_f) & 15) == 0, ANR_E_ALIGN,
                "anr_train_draws: outputs must be 16-B aligned");
    auto groups = [](int64_t n) { return (n + 3) / 4; };
    const int64_t g0 = groups(p.n_t), g1 = g0 + groups(p.n_nc), g2 = g1 + groups(p.n_u), g3 = g2 + groups(p.n_nf), g4 = g3 + groups(p.n_v3);
    ANR_REQUIRE(g4 > 0 && g4 < ((int64_t)1 << 28) * 256, ANR_E_BADARG, "anr_train_draws: nothing to draw, or too much");
    hipLaunchKernelGGL(train_draws_kernel, dim3((unsigned)((g4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<uint64_t*>(state), p, g0, g1, g2, g3, g4);
    hipLaunchKernelGGL(draws_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<uint64_t*>(state));
    return check_launch("anr_train_draws");
}

extern "C" int anr_gather_frame_params(const int64_t* frame_idx, int bs, const float* betas_w, int betas_rows, const float* go_w,
                                       const float* bp_w, const float* tr_w, float* betas_out, float* pose_out, float* transl_out,
                                       void* stream) {
    ANR_REQUIRE(frame_idx && betas_w && go_w && bp_w && tr_w && betas_out && pose_out && transl_out, ANR_E_BADARG,
                "anr_gather_frame_params: null pointer");
    ANR_REQUIRE(bs > 0 && betas_rows >= 1, ANR_E_BADARG, "anr_gather_frame_params: bs=%d betas_rows=%d", bs, betas_rows);
    hipLaunchKernelGGL(gather_frame_params_kernel, dim3(bs), dim3(128), 0, (hipStream_t)stream, frame_idx, betas_w, betas_rows, go_w, bp_w,
                       tr_w, betas_out, pose_out, transl_out);
    return check_launch("anr_gather_frame_params");
}

extern "C" int anr_scatter_frame_param_grads(const int64_t* frame_idx, const float* grads, int bs, int table_rows, int betas_rows,
                                             float* d_betas_w, float* d_go_w, float* d_bp_w, float* d_tr_w, void* stream) {
    ANR_REQUIRE(frame_idx && grads, ANR_E_BADARG, "anr_scatter_frame_param_grads: null pointer");
    ANR_REQUIRE(bs > 0 && table_rows > 0 && betas_rows >= 1, ANR_E_BADARG, "anr_scatter_frame_param_grads: bs=%d rows=%d", bs, table_rows);
    hipLaunchKernelGGL(scatter_frame_param_grads_kernel, dim3(table_rows), dim3(128), 0, (hipStream_t)stream, frame_idx, grads, bs,
                       betas_rows, d_betas_w, d_go_w, d_bp_w, d_tr_w);
    return check_launch("anr_scatter_frame_param_grads");
}

extern "C" int anr_merge_backward2(const float* g_a, const float* g_b, const uint8_t* perm, int64_t R, int K, int Kc,
                                   float* d_z_coarse_out, void* stream) {
    ANR_REQUIRE(g_a && perm && d_z_coarse_out, ANR_E_BADARG, "anr_merge_backward2: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && K <= 256 && Kc > 0 && Kc <= K, ANR_E_BADARG, "anr_merge_backward2: R=%lld K=%d Kc=%d", (long long)R, K, Kc);
    const int64_t n = R * K;
    hipLaunchKernelGGL(merge_backward2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_a, g_b, perm, n, K,
                       Kc, d_z_coarse_out);
    return check_launch("anr_merge_backward2");
}

extern "C" int anr_sample_coarse_backward_acc(const float* g_a, const float* g_b, const float* g_c, const float* steps,
                                              const float* t_rand, const float* dfar_a, const float* dfar_b, int64_t R, int K,
                                              float* d_rays_acc, void* stream) {
    ANR_REQUIRE(g_a && steps && d_rays_acc, ANR_E_BADARG, "anr_sample_coarse_backward_acc: null pointer");
    ANR_REQUIRE(R > 0 && K > 0, ANR_E_BADARG, "anr_sample_coarse_backward_acc: R=%lld K=%d", (long long)R, K);
    hipLaunchKernelGGL(sample_coarse_backward_acc_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g_a, g_b,
                       g_c, steps, t_rand, dfar_a, dfar_b, R, K, d_rays_acc);
    return check_launch("anr_sample_coarse_backward_acc");
}

namespace anr {
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n && ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0)) {
        float4 a = *reinterpret_cast<const float4*>(dst + i);
        const float4 b = *reinterpret_cast<const float4*>(src + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        *reinterpret_cast<float4*>(dst + i) = a;
    } else {
        for (int64_t k = i; k < n && k < i + 4; ++k) dst[k] += src[k];
    }
}
}  // namespace anr

extern "C" int anr_add_inplace(float* dst, const float* src, int64_t n, void* stream) {
    ANR_REQUIRE(dst && src && n > 0, ANR_E_BADARG, "anr_add_inplace: null pointer or n=%lld", (long long)n);
    hipLaunchKernelGGL(anr::add_inplace_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    return anr::check_launch("anr_add_inplace");
}

namespace anr {
constexpr int ADD_MAX_SEGS = 24;
struct AddSegs { float* dst[ADD_MAX_SEGS]; const float* src[ADD_MAX_SEGS]; int64_t n[ADD_MAX_SEGS]; };
__global__ __launch_bounds__(256) void add_segments_kernel(AddSegs t) {
    float* dst = t.dst[blockIdx.y];
    const float* src = t.src[blockIdx.y];
    const int64_t n = t.n[blockIdx.y];
    const bool vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (vec && i + 4 <= n) {
            float4 a = *reinterpret_cast<const float4*>(dst + i);
            const float4 b = *reinterpret_cast<const float4*>(src + i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            *reinterpret_cast<float4*>(dst + i) = a;
        } else {
            for (int64_t k = i; k < n && k < i + 4; ++k) dst[k] += src[k];
        }
    }
}
}  // namespace anr

extern "C" int anr_add_segments(float* const* dst, const float* const* src, const int64_t* floats, int n_segments, void* stream) {
    ANR_REQUIRE(dst && src && floats, ANR_E_BADARG, "anr_add_segments: null pointer");
    ANR_REQUIRE(n_segments > 0 && n_segments <= anr::ADD_MAX_SEGS, ANR_E_BADARG, "anr_add_segments: n=%d segments (1..%d)", n_segments,
                anr::ADD_MAX_SEGS);
    anr::AddSegs t{};
    int64_t most = 0;
    for (int i = 0; i < n_segments; ++i) {
        ANR_REQUIRE(dst[i] && src[i] && floats[i] >= 0, ANR_E_BADARG, "anr_add_segments: segment %d: null pointer or negative size", i);
        t.dst[i] = dst[i]; t.src[i] = src[i]; t.n[i] = floats[i];
        most = floats[i] > most ? floats[i] : most;
    }
    int64_t blocks = (most + 1023) / 1024;
    blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(anr::add_segments_kernel, dim3((unsigned)blocks, (unsigned)n_segments), dim3(256), 0, (hipStream_t)stream, t);
    return anr::check_launch("anr_add_segments");
}

namespace anr {
constexpr int COPY_MAX_SEGS = 24;
struct CopySegs { const char* src[COPY_MAX_SEGS]; char* dst[COPY_MAX_SEGS]; int64_t bytes[COPY_MAX_SEGS]; };
// blockIdx.y = segment; the table rides in the kernel arguments (no upload, nothing for a graph to bake but the node itself)
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs t) {
    const char* s = t.src[blockIdx.y];
    char* d = t.dst[blockIdx.y];
    const int64_t n = t.bytes[blockIdx.y];
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        for (int64_t i = tid; i < n / 16; i += nth) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (int64_t i = (n / 16) * 16 + tid; i < n; i += nth) d[i] = s[i];
    } else {
        for (int64_t i = tid; i < n; i += nth) d[i] = s[i];
    }
}
}  // namespace anr

extern "C" int anr_copy_segments(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream) {
    ANR_REQUIRE(src && dst && bytes, ANR_E_BADARG, "anr_copy_segments: null pointer");
    ANR_REQUIRE(n > 0 && n <= anr::COPY_MAX_SEGS, ANR_E_BADARG, "anr_copy_segments: n=%d segments (1..%d)", n, anr::COPY_MAX_SEGS);
    anr::CopySegs t{};
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        ANR_REQUIRE(src[i] && dst[i] && bytes[i] >= 0, ANR_E_BADARG, "anr_copy_segments: segment %d: null pointer or negative size", i);
        t.src[i] = reinterpret_cast<const char*>(src[i]);
        t.dst[i] = reinterpret_cast<char*>(dst[i]);
        t.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    int64_t blocks = (most / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 64 ? 64 : blocks;
    hipLaunchKernelGGL(anr::copy_segments_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return anr::check_launch("anr_copy_segments");
}

namespace anr {
struct ZeroSegs { char* dst[COPY_MAX_SEGS]; int64_t bytes[COPY_MAX_SEGS]; };
__global__ __launch_bounds__(256) void zero_segments_kernel(ZeroSegs t) {
    char* d = t.dst[blockIdx.y];
    const int64_t n = t.bytes[blockIdx.y];
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if (((uintptr_t)d & 15) == 0) {
        for (int64_t i = tid; i < n / 16; i += nth) reinterpret_cast<uint4*>(d)[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t i = (n / 16) * 16 + tid; i < n; i += nth) d[i] = 0;
    } else {
        for (int64_t i = tid; i < n; i += nth) d[i] = 0;
    }
}
}  // namespace anr

extern "C" int anr_zero_segments(void* const* dst, const int64_t* bytes, int n, void* stream) {
    ANR_REQUIRE(dst && bytes, ANR_E_BADARG, "anr_zero_segments: null pointer");
    ANR_REQUIRE(n > 0 && n <= anr::COPY_MAX_SEGS, ANR_E_BADARG, "anr_zero_segments: n=%d segments (1..%d)", n, anr::COPY_MAX_SEGS);
    anr::ZeroSegs t{};
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        ANR_REQUIRE(dst[i] && bytes[i] >= 0, ANR_E_BADARG, "anr_zero_segments: segment %d: null pointer or negative size", i);
        t.dst[i] = reinterpret_cast<char*>(dst[i]);
        t.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    int64_t blocks = (most / 16 + 1023) / 1024;                // four 16-byte stores per thread at the largest segment
    blocks = blocks < 1 ? 1 : blocks > 256 ? 256 : blocks;
    hipLaunchKernelGGL(anr::zero_segments_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return anr::check_launch("anr_zero_segments");
}

extern "C" int anr_zero_fill(void* ptr, int64_t bytes, void* stream) {
    ANR_REQUIRE(ptr && bytes >= 0, ANR_E_BADARG, "anr_zero_fill: null pointer");
    return zero_fill(ptr, (size_t)bytes, (hipStream_t)stream, "anr_zero_fill");
}
