// Backward of the per-frame chain under pose refinement: what torch autograd differentiates in
//   smplx/lbs.py:152-251 (shape / pose blend shapes, Rodrigues, kinematic chain, skinning transforms),
//   models/anim_nerf.py:128-145 (rays and body state into the root-joint frame) and :147-151 (per-vertex
//   observation -> canonical transforms: 6890 affine inverses per frame)
// i.e. dL/d(betas, global_orient, body_pose, transl) from dL/d ober2cano[bs,V,4,4] and dL/d rays_body[bs,R,8].
//
// The chain has 85 scalar inputs per frame and ~110 k outputs, so the vector-Jacobian product is evaluated the other
// way round: ONE launch, one workgroup per (frame, parameter), which pushes the unit tangent of its parameter through
// the chain in forward mode — the same code as the values, on a value+derivative number type — and dots the resulting
// tangent of every output with the upstream gradient (block reduction, fixed order: deterministic).  No adjoint code to
// derive or keep in sync; ~10 GFLOP per 16-frame step instead of ~480 framework launches.
#include "anr_common.h"

namespace anr {

struct Dual {
    float v, d;
    __device__ Dual() : v(0.f), d(0.f) {}
    __device__ Dual(float a) : v(a), d(0.f) {}
    __device__ Dual(float a, float b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual dsqrt(Dual a) { const float s = sqrtf(a.v); return Dual(s, a.d / (2.0f * s)); }
__device__ __forceinline__ Dual dsin(Dual a) { return Dual(sinf(a.v), cosf(a.v) * a.d); }
__device__ __forceinline__ Dual dcos(Dual a) { return Dual(cosf(a.v), -sinf(a.v) * a.d); }

// affine 3x4 [R | t] (bottom row 0 0 0 1 implied), row-major m[r*4 + c]
struct Aff { Dual m[12]; };
__device__ __forceinline__ Aff compose(const Aff& a, const Aff& b) {        // a . b
    Aff o;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Dual s = a.m[r * 4 + 0] * b.m[0 * 4 + c] + a.m[r * 4 + 1] * b.m[1 * 4 + c] + a.m[r * 4 + 2] * b.m[2 * 4 + c];
            if (c == 3) s = s + a.m[r * 4 + 3];
            o.m[r * 4 + c] = s;
        }
    }
    return o;
}
// closed-form inverse (adjugate / determinant), as models/anim_nerf.py's differentiable form
__device__ __forceinline__ Aff inverse(const Aff& a) {
    const Dual *r0 = a.m, *r1 = a.m + 4, *r2 = a.m + 8;
    auto cross = [](const Dual* x, const Dual* y, Dual* o) {
        o[0] = x[1] * y[2] - x[2] * y[1]; o[1] = x[2] * y[0] - x[0] * y[2]; o[2] = x[0] * y[1] - x[1] * y[0];
    };
    Dual c12[3], c20[3], c01[3];
    cross(r1, r2, c12); cross(r2, r0, c20); cross(r0, r1, c01);
    const Dual det = r0[0] * c12[0] + r0[1] * c12[1] + r0[2] * c12[2];
    Aff o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                 // Rinv[i][0..2] = (c12[i], c20[i], c01[i]) / det
        o.m[i * 4 + 0] = c12[i] / det; o.m[i * 4 + 1] = c20[i] / det; o.m[i * 4 + 2] = c01[i] / det;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.m[i * 4 + 3] = -(o.m[i * 4 + 0] * a.m[3] + o.m[i * 4 + 1] * a.m[7] + o.m[i * 4 + 2] * a.m[11]);
    return o;
}

constexpr int FB_J = 24, FB_NB = 10, FB_NP = FB_NB + 3 * FB_J + 3;     // 85 parameters per frame
constexpr int FB_THREADS = 256;

// The joint chain of one (frame, parameter) in LDS, value + tangent of parameter `pi` (pi < 0: values only): joint
// transforms relative to the rest pose, the inverse of the root transform, transl, and the tangent of the pose feature of
// the parameter's joint `pj`.  Called by every thread of the workgroup (it synchronises); blockDim.x >= 24.
struct ChainLds {
    float A[FB_J][12][2];            // joint transforms relative to the rest pose (no transl): value, tangent
    float G[12][2];                  // inverse of the root transform
    float Tr[3][2];                  // transl
    float Feat[9];                   // tangent of the 9 pose-feature entries of the parameter's joint
    float Rot[FB_J][9][2], Jr[FB_J][3][2], World[FB_J][12][2];
    int desc;                        // bit j: joint j lies in the subtree of the parameter's joint
    int lvl_start[FB_J + 1];         // joints by depth in the kinematic tree: level l = lvl_joint[lvl_start[l] .. lvl_start[l + 1])
    int lvl_joint[FB_J];
    int n_levels;
    int par[FB_J], depth[FB_J];
};

__device__ __forceinline__ void run_chain(ChainLds& L, int b, int pi, int pj, const float* __restrict__ betas,
                                          const float* __restrict__ pose, const float* __restrict__ transl,
                                          const float* __restrict__ J0, const float* __restrict__ JS,
                                          const int64_t* __restrict__ parents) {
    // ---- the joint chain of this (frame, parameter), value + tangent, cooperatively and out of LDS (a single thread with
    // its 24 transforms in private memory spends milliseconds in scratch accesses)
    auto seed = [&](int idx, float val) { return Dual(val, idx == pi ? 1.0f : 0.0f); };
    if (threadIdx.x < FB_J) {
        const int j = threadIdx.x;
        // (every input of the lane first — 10 betas, its 30 regressor entries, 3 rest coordinates, 3 pose entries: one trip to
        // L2 — then the sums: a loop of load, load, multiply-add was thirty trips, most of frame_params_kernel's 28 us)
        float bet[FB_NB], js[3][FB_NB], j0[3], pv[3];
#pragma unroll
        for (int k = 0; k < FB_NB; ++k) bet[k] = betas[b * FB_NB + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < FB_NB; ++k) js[c][k] = JS[(j * 3 + c) * FB_NB + k];
            j0[c] = J0[j * 3 + c];
            pv[c] = pose[(b * FB_J + j) * 3 + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {                              // rest joints: J_regressor . (v_template + shapedirs . betas)
            Dual s(j0[c]);
#pragma unroll
            for (int k = 0; k < FB_NB; ++k) s = s + seed(k, bet[k]) * Dual(js[c][k]);
            L.Jr[j][c][0] = s.v; L.Jr[j][c][1] = s.d;
        }
        // Rodrigues, angle = |rv + 1e-8| (smplx/lbs.py:316)
        Dual rv[3];
        for (int c = 0; c < 3; ++c) rv[c] = seed(FB_NB + 3 * j + c, pv[c]);
        const Dual e0 = rv[0] + Dual(1e-8f), e1 = rv[1] + Dual(1e-8f), e2 = rv[2] + Dual(1e-8f);
        const Dual theta = dsqrt(e0 * e0 + e1 * e1 + e2 * e2);
        const Dual kx = rv[0] / theta, ky = rv[1] / theta, kz = rv[2] / theta;
        const Dual sn = dsin(theta), c1 = Dual(1.0f) - dcos(theta);
        const Dual K[9] = {Dual(0.f), -kz, ky, kz, Dual(0.f), -kx, -ky, kx, Dual(0.f)};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const Dual kk = K[r * 3 + 0] * K[0 * 3 + c] + K[r * 3 + 1] * K[1 * 3 + c] + K[r * 3 + 2] * K[2 * 3 + c];
                const Dual m = Dual(r == c ? 1.0f : 0.0f) + sn * K[r * 3 + c] + c1 * kk;
                L.Rot[j][r * 3 + c][0] = m.v; L.Rot[j][r * 3 + c][1] = m.d;
                if (j == pj) L.Feat[r * 3 + c] = m.d;              // pose feature = R - I: same tangent
            }
    }
    if (threadIdx.x < 9 && pj < 1) L.Feat[threadIdx.x] = 0.0f;      // the root joint has no pose blend shapes
    if (threadIdx.x == 0) L.desc = 0;
    if (threadIdx.x < FB_J) L.par[threadIdx.x] = (int)parents[threadIdx.x];
    if (threadIdx.x < 3) {
        const Dual t = seed(FB_NB + 3 * FB_J + threadIdx.x, transl[b * 3 + threadIdx.x]);
        L.Tr[threadIdx.x][0] = t.v; L.Tr[threadIdx.x][1] = t.d;
    }
    __syncthreads();
    // the joints by depth (a stable sort of 24 entries, every step out of LDS and in parallel): the chain below walks the tree
    // LEVEL BY LEVEL — 9 levels for SMPL's skeleton, the joints of a level side by side — instead of joint by joint (24 steps
    // of one dependent LDS round trip each: most of this 64-lane workgroup's time)
    if (threadIdx.x < FB_J) {
        int dpt = 0;
        for (int k = threadIdx.x; k > 0; k = L.par[k]) ++dpt;              // (parents[j] < j: tree order)
        L.depth[threadIdx.x] = dpt;
        // the subtree of the parameter's joint, a lane per joint out of LDS (one thread walking all 24 paths through the
        // parents array in global memory — ~100 dependent loads — was the longest thing in a pose parameter's workgroup)
        if (pj >= 1) {
            int k = threadIdx.x;
            while (k > 0 && k != pj) k = L.par[k];
            if (k == pj) atomicOr(&L.desc, 1 << threadIdx.x);
        }
    }
    __syncthreads();
    if (threadIdx.x < FB_J) {
        const int mine = L.depth[threadIdx.x];
        int rank = 0;
        for (int q = 0; q < FB_J; ++q) rank += (L.depth[q] < mine || (L.depth[q] == mine && q < (int)threadIdx.x)) ? 1 : 0;
        L.lvl_joint[rank] = threadIdx.x;
    }
    if (threadIdx.x <= FB_J) {
        int c = 0, deepest = 0;
        for (int q = 0; q < FB_J; ++q) { c += L.depth[q] < (int)threadIdx.x ? 1 : 0; deepest = max(deepest, L.depth[q]); }
        L.lvl_start[threadIdx.x] = c;
        if (threadIdx.x == 0) L.n_levels = deepest + 1;
    }
    __syncthreads();
    // world_j = world_parent . [R_j | J_j - J_parent], one element (r, c) per lane, five joints of a level per pass
    const int n_levels = L.n_levels;
#pragma unroll 1
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        const int first = L.lvl_start[lvl], n_here = L.lvl_start[lvl + 1] - first;
        for (int s0 = 0; s0 < n_here; s0 += 5) {
            const int slot = (int)threadIdx.x / 12, e = (int)threadIdx.x % 12;
            if (threadIdx.x < 60 && s0 + slot < n_here) {
                const int j = L.lvl_joint[first + s0 + slot];
                const int r = e >> 2, c = e & 3;
                const int par = j > 0 ? L.par[j] : 0;
                auto local = [&](int k, int cc) {                     // element (k, cc) of joint j's local transform
                    if (cc < 3) return Dual(L.Rot[j][k * 3 + cc][0], L.Rot[j][k * 3 + cc][1]);
                    Dual t(L.Jr[j][k][0], L.Jr[j][k][1]);
                    if (j > 0) t = t - Dual(L.Jr[par][k][0], L.Jr[par][k][1]);
                    return t;
                };
                Dual w;
                if (j == 0) {
                    w = local(r, c);
                } else {
                    w = Dual(L.World[par][r * 4 + 0][0], L.World[par][r * 4 + 0][1]) * local(0, c) +
                        Dual(L.World[par][r * 4 + 1][0], L.World[par][r * 4 + 1][1]) * local(1, c) +
                        Dual(L.World[par][r * 4 + 2][0], L.World[par][r * 4 + 2][1]) * local(2, c);
                    if (c == 3) w = w + Dual(L.World[par][r * 4 + 3][0], L.World[par][r * 4 + 3][1]);
                }
                L.World[j][e][0] = w.v; L.World[j][e][1] = w.d;
            }
        }
        __syncthreads();
    }
    // relative to the rest pose: t -= R_world . J_rest
    for (int i = threadIdx.x; i < FB_J * 12; i += (int)blockDim.x) {
        const int j = i / 12, e = i % 12, r = e >> 2, c = e & 3;
        Dual a(L.World[j][e][0], L.World[j][e][1]);
        if (c == 3)
            a = a - (Dual(L.World[j][r * 4 + 0][0], L.World[j][r * 4 + 0][1]) * Dual(L.Jr[j][0][0], L.Jr[j][0][1]) +
                     Dual(L.World[j][r * 4 + 1][0], L.World[j][r * 4 + 1][1]) * Dual(L.Jr[j][1][0], L.Jr[j][1][1]) +
                     Dual(L.World[j][r * 4 + 2][0], L.World[j][r * 4 + 2][1]) * Dual(L.Jr[j][2][0], L.Jr[j][2][1]));
        L.A[j][e][0] = a.v; L.A[j][e][1] = a.d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                        // global transform = A_0 + transl (body_models.py:373)
        Aff G;
        for (int e = 0; e < 12; ++e) G.m[e] = Dual(L.A[0][e][0], L.A[0][e][1]);
        for (int r = 0; r < 3; ++r) G.m[r * 4 + 3] = G.m[r * 4 + 3] + Dual(L.Tr[r][0], L.Tr[r][1]);
        const Aff Gi0 = inverse(G);
        for (int e = 0; e < 12; ++e) { L.G[e][0] = Gi0.m[e].v; L.G[e][1] = Gi0.m[e].d; }
    }
    __syncthreads();

}

// grads[bs][85] = (betas 10 | global_orient 3 | body_pose 69 | transl 3)
__global__ __launch_bounds__(FB_THREADS, 2) void frame_backward_kernel(
    const float* __restrict__ betas, const float* __restrict__ pose, const float* __restrict__ transl,
    const float* __restrict__ J0, const float* __restrict__ JS, const int64_t* __restrict__ parents,
    const float* __restrict__ lbs_w, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ T_templ, int64_t templ_stride, const float* __restrict__ rays_world, int ray_stride, int R,
    const float* __restrict__ d_o2c, const float* __restrict__ d_rays, int V, const int32_t* __restrict__ vjmask,
    float* __restrict__ grads) {
    const int b = blockIdx.y, pi = blockIdx.x;
    __shared__ ChainLds L;
    __shared__ float sRed[FB_THREADS / 64];
    const int pj = (pi >= FB_NB && pi < FB_NB + 3 * FB_J) ? (pi - FB_NB) / 3 : -1;      // joint of a pose parameter

    run_chain(L, b, pi, pj, betas, pose, transl, J0, JS, parents);

    Aff Gi;
#pragma unroll
    for (int e = 0; e < 12; ++e) Gi.m[e] = Dual(L.G[e][0], L.G[e][1]);
    float acc = 0.0f;
    // ---- every vertex: tangent of T_template . inverse(G^-1 . T_v) (+ offsets), dotted with dL/d ober2cano
    // A body_pose parameter (joint pj >= 1) moves only the joints of its subtree: the root transform, the translation and
    // every vertex without a skinning weight on that subtree have a zero tangent, and only the pose blend shape of the
    // vertex is left of its term — 27 loads and a 3x3 product instead of the dual-number inverse.  Exact (the skipped
    // tangents are exact zeros); vjmask[v] = bit mask of the joints vertex v has a non-zero weight on (may be NULL).
    const int desc = L.desc;
    const bool local = pj >= 1 && vjmask != nullptr;
#pragma unroll 1
    for (int v = threadIdx.x; v < V; v += FB_THREADS) {
        if (local && !(vjmask[v] & desc)) {
            if (d_o2c == nullptr) break;
            float dpo[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pd = posedirs + (int64_t)(9 * (pj - 1)) * (3 * V) + 3 * v + c;
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 9; ++e) s += L.Feat[e] * pd[(int64_t)e * 3 * V];
                dpo[c] = s;
            }
            const float* Tt = T_templ + b * templ_stride + (int64_t)v * 16;
            const float* g = d_o2c + ((int64_t)b * V + v) * 16;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                acc -= g[r * 4 + 3] * (Tt[r * 4 + 0] * dpo[0] + Tt[r * 4 + 1] * dpo[1] + Tt[r * 4 + 2] * dpo[2]);
            continue;
        }
        Aff T;
#pragma unroll
        for (int e = 0; e < 12; ++e) T.m[e] = Dual(0.f);
        const float* w = lbs_w + (int64_t)v * FB_J;
#pragma unroll 1
        for (int j = 0; j < FB_J; ++j) {
            const float wj = w[j];
            if (wj != 0.0f) {
#pragma unroll
                for (int e = 0; e < 12; ++e) { T.m[e].v += wj * L.A[j][e][0]; T.m[e].d += wj * L.A[j][e][1]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { T.m[r * 4 + 3].v += L.Tr[r][0]; T.m[r * 4 + 3].d += L.Tr[r][1]; }
        const Aff M = inverse(compose(Gi, T));
        float dM[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) dM[e] = M.m[e].d;
        // translation += (shape_off_template - shape_off) + (pose_off_template - pose_off): only the tangents matter here
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float dso = pi < FB_NB ? shapedirs[((int64_t)v * 3 + c) * FB_NB + pi] : 0.0f;
            float dpo = 0.0f;
            if (pj >= 1) {
                const float* pd = posedirs + (int64_t)(9 * (pj - 1)) * (3 * V) + 3 * v + c;
#pragma unroll
                for (int e = 0; e < 9; ++e) dpo += L.Feat[e] * pd[(int64_t)e * 3 * V];
            }
            dM[c * 4 + 3] -= dso + dpo;
        }
        if (d_o2c == nullptr) break;
        const float* Tt = T_templ + b * templ_stride + (int64_t)v * 16;
        const float* g = d_o2c + ((int64_t)b * V + v) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t = Tt[r * 4 + 0] * dM[0 * 4 + c] + Tt[r * 4 + 1] * dM[1 * 4 + c] + Tt[r * 4 + 2] * dM[2 * 4 + c];
                acc += g[r * 4 + c] * t;
            }
    }
    // ---- every ray: o' = G^-1 [o,1], d' = G^-1 [d,0], near' = max(near, |o'| - 1), far' = min(far, |o'| + 1)
    if (d_rays != nullptr && pj < 1) {                             // (a body_pose parameter does not move the root frame)
#pragma unroll 1
        for (int r = threadIdx.x; r < R; r += FB_THREADS) {
            const float* ry = rays_world + ((int64_t)b * R + r) * ray_stride;
            const float* g = d_rays + ((int64_t)b * R + r) * 8;
            Dual o[3];
            float norm2 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                o[i] = Gi.m[i * 4 + 0] * Dual(ry[0]) + Gi.m[i * 4 + 1] * Dual(ry[1]) + Gi.m[i * 4 + 2] * Dual(ry[2]) + Gi.m[i * 4 + 3];
                const float dd = Gi.m[i * 4 + 0].d * ry[3] + Gi.m[i * 4 + 1].d * ry[4] + Gi.m[i * 4 + 2].d * ry[5];
                acc += g[i] * o[i].d + g[3 + i] * dd;
                norm2 += o[i].v * o[i].v;
            }
            const float dist = sqrtf(norm2);
            const float ddist = (o[0].v * o[0].d + o[1].v * o[1].d + o[2].v * o[2].d) / dist;
            if (dist - 1.0f > ry[6]) acc += g[6] * ddist;            // torch.max / torch.min pass the gradient to the larger /
            if (dist + 1.0f < ry[7]) acc += g[7] * ddist;            // smaller argument
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < FB_THREADS / 64; ++i) s += sRed[i];
        grads[(int64_t)b * FB_NP + pi] = s;
    }
}

// =====================================================================================================================
// The same gradient the other way round: reverse mode through the 6,890 per-vertex inverses (once per frame, not once per
// parameter), forward mode only through the 24-joint chain.
//   A  frame_adjoint_kernel: one workgroup per 256 vertices (or 256 rays) of a frame.  Per vertex: T = sum_j w_j A_j + transl,
//      X = G^-1 T, M = X^-1 (+ offsets), out = T_template M; the upstream gradient of `out` is pulled back to
//        g_off (3, the adjoint of the blend-shape offsets)   -> goff[b][v][3]
//        g_T (12) -> g_A[j] += w_j g_T, g_transl += g_T[:,3]     g_Ginv (12)
//      and, per ray, o' = G^-1 [o,1], d' = G^-1 [d,0], near'/far' = clamp(|o'| -+ 1) to g_Ginv.
//      Accumulated per workgroup in LDS, then one atomic per value into acc[b][303] = (g_A 24x12 | g_Ginv 12 | g_transl 3).
//   B  frame_offsets_kernel: H[b][f] = - sum_{v,c} goff[b][v][c] posedirs[f][3v+c] (f < 207: adjoint of the pose features),
//      H[b][207+k] = - sum goff . shapedirs[.,.,k] (betas through the shape blend shapes).  Fixed-order block reductions.
//   C  frame_params_kernel: one small workgroup per (frame, parameter): the chain above in forward mode (24 joints, nothing
//      per vertex) and the dot of its tangents with acc and H.
// ~0.1 GFLOP per 16-frame step instead of ~5; checked against the forward-mode kernel
// (tests/test_gpu_training.py::test_frame_backward_adjoint_equals_forward_mode).
constexpr int FA_ACC = FB_J * 12 + 12 + 3;
constexpr int FA_H = 207 + FB_NB;

__global__ __launch_bounds__(FB_THREADS) void frame_adjoint_kernel(
    const float* __restrict__ betas, const float* __restrict__ pose, const float* __restrict__ transl,
    const float* __restrict__ J0, const float* __restrict__ JS, const int64_t* __restrict__ parents,
    const float* __restrict__ lbs_w, const float* __restrict__ T_templ, int64_t templ_stride,
    const float* __restrict__ rays_world, int ray_stride, int R, const float* __restrict__ d_o2c,
    const float* __restrict__ d_rays, int V, int n_vblocks, float* __restrict__ acc, float* __restrict__ goff,
    const float* __restrict__ A_fwd, const float* __restrict__ Ginv_fwd) {
    __shared__ ChainLds L;
    __shared__ float sAcc[FA_ACC];
    // per wavefront: the 64 vertices' skinning weights [64][24] and pulled-back transforms gT [64][12], for the contraction
    // g_A[j][e] = sum_v w[v][j] gT[v][e] below
    __shared__ float sW[FB_THREADS / 64][64][FB_J];
    __shared__ float sGT[FB_THREADS / 64][64][12];
    const int b = blockIdx.y;
    if (A_fwd != nullptr && Ginv_fwd != nullptr) {
        // the VALUES of the chain as the forward kernels left them (anr_smpl_forward's joints_transform, with transl on its
        // translation column; anr_to_root_frame's inverse root transform): no 24-step chain per workgroup (25 of this kernel's
        // 89 us at 2 frames)
        for (int i = threadIdx.x; i < FB_J * 12; i += FB_THREADS) {
            const int j = i / 12, e = i % 12;
            float a = A_fwd[((int64_t)b * FB_J + j) * 16 + e];
            if ((e & 3) == 3) a -= transl[b * 3 + (e >> 2)];
            L.A[j][e][0] = a;
        }
        if (threadIdx.x < 12) L.G[threadIdx.x][0] = Ginv_fwd[(int64_t)b * 16 + threadIdx.x];
        if (threadIdx.x < 3) L.Tr[threadIdx.x][0] = transl[b * 3 + threadIdx.x];
    } else {
        run_chain(L, b, -1, -1, betas, pose, transl, J0, JS, parents);
    }
    for (int i = threadIdx.x; i < FA_ACC; i += FB_THREADS) sAcc[i] = 0.0f;
    __syncthreads();
    float Gi[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Gi[e] = L.G[e][0];
    float gG[12], gTr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 12; ++e) gG[e] = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < n_vblocks) {
        const int v = blockIdx.x * FB_THREADS + threadIdx.x;
        float gT[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) gT[e] = 0.0f;
        if (v < V) {
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.0f;
            const float* w = lbs_w + (int64_t)v * FB_J;
            float wv[FB_J];                                      // (the 24 weights in one trip, not one trip per joint)
#pragma unroll
            for (int j = 0; j < FB_J; ++j) wv[j] = w[j];
#pragma unroll
            for (int j = 0; j < FB_J; ++j) {
                const float wj = wv[j];
                if (wj != 0.0f) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) T[e] += wj * L.A[j][e][0];
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) T[r * 4 + 3] += L.Tr[r][0];
            float X[12];                                                  // X = G^-1 . T
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    X[r * 4 + c] = Gi[r * 4 + 0] * T[c] + Gi[r * 4 + 1] * T[4 + c] + Gi[r * 4 + 2] * T[8 + c] + (c == 3 ? Gi[r * 4 + 3] : 0.0f);
            // Y = inverse of the 3x3 block (adjugate / determinant)
            float Y[9];
            {
                const float c00 = X[5] * X[10] - X[6] * X[9], c01 = X[6] * X[8] - X[4] * X[10], c02 = X[4] * X[9] - X[5] * X[8];
                const float id = 1.0f / (X[0] * c00 + X[1] * c01 + X[2] * c02);
                Y[0] = c00 * id; Y[1] = (X[2] * X[9] - X[1] * X[10]) * id; Y[2] = (X[1] * X[6] - X[2] * X[5]) * id;
                Y[3] = c01 * id; Y[4] = (X[0] * X[10] - X[2] * X[8]) * id; Y[5] = (X[2] * X[4] - X[0] * X[6]) * id;
                Y[6] = c02 * id; Y[7] = (X[1] * X[8] - X[0] * X[9]) * id; Y[8] = (X[0] * X[5] - X[1] * X[4]) * id;
            }
            const float* Tt = T_templ + b * templ_stride + (int64_t)v * 16;
            const float* g = d_o2c + ((int64_t)b * V + v) * 16;
            float gM[12];                                                 // gM[k][c] = sum_r Tt[r][k] g[r][c]
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) gM[k * 4 + c] = Tt[k] * g[c] + Tt[4 + k] * g[4 + c] + Tt[8 + k] * g[8 + c];
#pragma unroll
            for (int k = 0; k < 3; ++k) goff[((int64_t)b * V + v) * 3 + k] = gM[k * 4 + 3];
            // M_R = Y, M_t = -Y X_t + off:  GR = gM_R - gM_t (x) X_t,  gX_t = -Y^T gM_t,  gX_R = -Y^T GR Y^T
            float GR[9], gXt[3], P[9], gXR[9];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int m = 0; m < 3; ++m) GR[k * 3 + m] = gM[k * 4 + m] - gM[k * 4 + 3] * X[m * 4 + 3];
#pragma unroll
            for (int m = 0; m < 3; ++m) gXt[m] = -(Y[0 * 3 + m] * gM[3] + Y[1 * 3 + m] * gM[7] + Y[2 * 3 + m] * gM[11]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n < 3; ++n) P[k * 3 + n] = GR[k * 3 + 0] * Y[n * 3 + 0] + GR[k * 3 + 1] * Y[n * 3 + 1] + GR[k * 3 + 2] * Y[n * 3 + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int n = 0; n < 3; ++n) gXR[i * 3 + n] = -(Y[0 * 3 + i] * P[0 * 3 + n] + Y[1 * 3 + i] * P[1 * 3 + n] + Y[2 * 3 + i] * P[2 * 3 + n]);
            // X_R = G_R T_R, X_t = G_R T_t + G_t
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) gT[k * 4 + c] = Gi[0 * 4 + k] * gXR[0 * 3 + c] + Gi[1 * 4 + k] * gXR[1 * 3 + c] + Gi[2 * 4 + k] * gXR[2 * 3 + c];
                gT[k * 4 + 3] = Gi[0 * 4 + k] * gXt[0] + Gi[1 * 4 + k] * gXt[1] + Gi[2 * 4 + k] * gXt[2];
                gTr[k] = gT[k * 4 + 3];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    gG[r * 4 + k] = gXR[r * 3 + 0] * T[k * 4 + 0] + gXR[r * 3 + 1] * T[k * 4 + 1] + gXR[r * 3 + 2] * T[k * 4 + 2] + gXt[r] * T[k * 4 + 3];
                gG[r * 4 + 3] = gXt[r];
            }
        }
        // g_A[j][e] += sum over the wavefront's 64 vertices of w[v][j] gT[v][e]: both operands through this wave's LDS patch,
        // lane l owns the outputs l, l + 64, ... of the 288 (consecutive lanes: consecutive e of one j — the w reads broadcast,
        // the gT reads hit 12 consecutive words) and walks the vertices in order.  (Every lane adding its 12 products per joint
        // with LDS atomics queued up to 64 adds on ONE address 288 times per wavefront: most of this kernel's 89 us at 2 frames.)
        {
            const float* wv = lbs_w + (int64_t)(v < V ? v : V - 1) * FB_J;
#pragma unroll
            for (int j = 0; j < FB_J; ++j) sW[wave][lane][j] = v < V ? wv[j] : 0.0f;
#pragma unroll
            for (int e = 0; e < 12; ++e) sGT[wave][lane][e] = gT[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int k = 0; k < (FB_J * 12 + 63) / 64; ++k) {
            const int o = lane + 64 * k;
            if (o < FB_J * 12) {
                const int j = o / 12, e = o % 12;
                float acc_je = 0.0f;
                for (int u = 0; u < 64; ++u) acc_je += sW[wave][u][j] * sGT[wave][u][e];
                if (acc_je != 0.0f) atomicAdd(&sAcc[o], acc_je);
            }
        }
    } else {
        const int r = ((int)blockIdx.x - n_vblocks) * FB_THREADS + threadIdx.x;
        if (r < R) {
            const float* ry = rays_world + ((int64_t)b * R + r) * ray_stride;
            const float* g = d_rays + ((int64_t)b * R + r) * 8;
            float o[3], norm2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                o[i] = Gi[i * 4 + 0] * ry[0] + Gi[i * 4 + 1] * ry[1] + Gi[i * 4 + 2] * ry[2] + Gi[i * 4 + 3];
                norm2 += o[i] * o[i];
            }
            const float dist = sqrtf(norm2);
            // torch.max / torch.min pass the gradient to the larger / smaller argument
            const float gd = ((dist - 1.0f > ry[6]) ? g[6] : 0.0f) + ((dist + 1.0f < ry[7]) ? g[7] : 0.0f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float go = g[i] + gd * o[i] / dist;
#pragma unroll
                for (int k = 0; k < 3; ++k) gG[i * 4 + k] = go * ry[k] + g[3 + i] * ry[3 + k];
                gG[i * 4 + 3] = go;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const float sum = wave_sum(gG[e]);
        if ((threadIdx.x & 63) == 0 && sum != 0.0f) atomicAdd(&sAcc[FB_J * 12 + e], sum);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sum = wave_sum(gTr[c]);
        if ((threadIdx.x & 63) == 0 && sum != 0.0f) atomicAdd(&sAcc[FB_J * 12 + 12 + c], sum);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FA_ACC; i += FB_THREADS)
        if (sAcc[i] != 0.0f) atomicAdd(&acc[(int64_t)b * FA_ACC + i], sAcc[i]);
}

// (1,024 threads per dot product of length 3 V = 20,670: 21 loads per thread instead of 81 — the kernel is a latency chain,
// 42 us at 2 frames with 256; fixed-order tree: the same bits on every run)
constexpr int FO_THREADS = 1024;
// BT frames per workgroup: a blend-shape row (82 KB; 17 MB in all) is read once for BT frames — a workgroup per (row, frame)
// read the 17 MB once per frame: 70 us at 16 frames (and the frames of a row placed back to back on one XCD walked the same
// lines in lockstep, one channel at a time: 300 us).  Per (row, frame) the sum and its order are what they were.
template <int BT>
__global__ __launch_bounds__(FO_THREADS) void frame_offsets_kernel(const float* __restrict__ goff, const float* __restrict__ shapedirs,
                                                                   const float* __restrict__ posedirs, int V, float* __restrict__ H, int bs) {
    __shared__ float sRed[BT][FO_THREADS / 64];
    const int f = blockIdx.x, b0 = (int)blockIdx.y * BT;
    float s[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) s[t] = 0.0f;
    const int n = 3 * V;
    // (U iterations' loads in flight at a time — a loop of load, load, multiply-add is one trip to L2 per iteration: 21 trips)
    constexpr int U = 8;
    const float* row = f < 207 ? posedirs + (int64_t)f * n : shapedirs + (f - 207);
    const int stride = f < 207 ? 1 : FB_NB;
    for (int i0 = threadIdx.x; i0 < n; i0 += U * FO_THREADS) {
        float p[U], g[U][BT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * FO_THREADS;
            p[u] = i < n ? row[(int64_t)i * stride] : 0.0f;
#pragma unroll
            for (int t = 0; t < BT; ++t) g[u][t] = (i < n && b0 + t < bs) ? goff[(int64_t)(b0 + t) * n + i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * FO_THREADS < n) {
#pragma unroll
                for (int t = 0; t < BT; ++t)
                    if (b0 + t < bs) s[t] += g[u][t] * p[u];
            }
    }
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const float w = wave_sum(s[t]);
        if ((threadIdx.x & 63) == 0) sRed[t][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < BT && b0 + (int)threadIdx.x < bs) {
        const float* r = sRed[threadIdx.x];
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < FO_THREADS / 64; w += 4) t += (r[w] + r[w + 1]) + (r[w + 2] + r[w + 3]);
        H[(int64_t)(b0 + threadIdx.x) * FA_H + f] = -t;
    }
}

__global__ __launch_bounds__(64) void frame_params_kernel(
    const float* __restrict__ betas, const float* __restrict__ pose, const float* __restrict__ transl,
    const float* __restrict__ J0, const float* __restrict__ JS, const int64_t* __restrict__ parents,
    const float* __restrict__ acc, const float* __restrict__ H, float* __restrict__ grads) {
    __shared__ ChainLds L;
    const int b = blockIdx.y, pi = blockIdx.x;
    const int pj = (pi >= FB_NB && pi < FB_NB + 3 * FB_J) ? (pi - FB_NB) / 3 : -1;
    run_chain(L, b, pi, pj, betas, pose, transl, J0, JS, parents);
    const float* a = acc + (int64_t)b * FA_ACC;
    float s = 0.0f;
    for (int i = threadIdx.x; i < FA_ACC; i += 64) {
        const float t = i < FB_J * 12 ? L.A[i / 12][i % 12][1] : i < FB_J * 12 + 12 ? L.G[i - FB_J * 12][1] : L.Tr[i - FB_J * 12 - 12][1];
        s += a[i] * t;
    }
    if (H != nullptr) {
        const float* h = H + (int64_t)b * FA_H;
        if (pj >= 1 && threadIdx.x < 9) s += h[9 * (pj - 1) + threadIdx.x] * L.Feat[threadIdx.x];
        if (pi < FB_NB && threadIdx.x == 0) s += h[207 + pi];
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) grads[(int64_t)b * FB_NP + pi] = s;
}

}  // namespace anr

using namespace anr;

extern "C" int anr_frame_backward(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                                  const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                                  const float* posedirs, int V, const float* T_template, int template_bs,
                                  const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                                  const float* d_rays_body, const int32_t* vertex_joint_mask, float* grads_out, void* stream) {
    ANR_REQUIRE(betas && pose && transl && J0 && JS && parents && lbs_weights && shapedirs && posedirs && T_template &&
                (d_ober2cano || d_rays_body) && grads_out, ANR_E_BADARG, "anr_frame_backward: null pointer");
    ANR_REQUIRE(bs > 0 && V > 0 && (template_bs == 1 || template_bs == bs), ANR_E_BADARG, "anr_frame_backward: bs=%d V=%d template_bs=%d",
                bs, V, template_bs);
    ANR_REQUIRE(!d_rays_body || (rays_world && R > 0 && ray_stride >= 8), ANR_E_BADARG, "anr_frame_backward: rays R=%d stride=%d", R, ray_stride);
    hipLaunchKernelGGL(frame_backward_kernel, dim3(FB_NP, bs), dim3(FB_THREADS), 0, (hipStream_t)stream, betas, pose, transl,
                       J0, JS, parents, lbs_weights, shapedirs, posedirs, T_template,
                       template_bs == 1 ? (int64_t)0 : (int64_t)V * 16, rays_world, ray_stride, R, d_ober2cano, d_rays_body, V,
                       vertex_joint_mask, grads_out);
    return check_launch("anr_frame_backward");
}

extern "C" int64_t anr_frame_backward_ws_floats(int bs, int V) { return (int64_t)bs * (FA_ACC + FA_H + 3 * (int64_t)V); }
extern "C" int64_t anr_frame_backward_ws_zero_floats(int bs) { return (int64_t)bs * FA_ACC; }

extern "C" int anr_frame_backward_adjoint(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                                          const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                                          const float* posedirs, int V, const float* T_template, int template_bs,
                                          const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                                          const float* d_rays_body, float* workspace, float* grads_out, void* stream) {
    return anr_frame_backward_adjoint_values(betas, pose, transl, bs, J0, JS, parents, lbs_weights, shapedirs, posedirs, V, T_template,
                                             template_bs, rays_world, ray_stride, R, d_ober2cano, d_rays_body, nullptr, nullptr, workspace,
                                             grads_out, 0, stream);
}

extern "C" int anr_frame_backward_adjoint_values(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                                                 const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                                                 const float* posedirs, int V, const float* T_template, int template_bs,
                                                 const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                                                 const float* d_rays_body, const float* joints_transform, const float* g_inv,
                                                 float* workspace, float* grads_out, int flags, void* stream) {
    ANR_REQUIRE(betas && pose && transl && J0 && JS && parents && lbs_weights && shapedirs && posedirs && T_template &&
                (d_ober2cano || d_rays_body) && workspace && grads_out, ANR_E_BADARG, "anr_frame_backward_adjoint: null pointer");
    ANR_REQUIRE((joints_transform == nullptr) == (g_inv == nullptr), ANR_E_BADARG,
                "anr_frame_backward_adjoint_values: joints_transform and g_inv come together");
    ANR_REQUIRE(bs > 0 && V > 0 && (template_bs == 1 || template_bs == bs), ANR_E_BADARG,
                "anr_frame_backward_adjoint: bs=%d V=%d template_bs=%d", bs, V, template_bs);
    ANR_REQUIRE(!d_rays_body || (rays_world && R > 0 && ray_stride >= 8), ANR_E_BADARG, "anr_frame_backward_adjoint: rays R=%d stride=%d", R,
                ray_stride);
    hipStream_t st = (hipStream_t)stream;
    float* acc = workspace;
    float* H = acc + (int64_t)bs * FA_ACC;
    float* goff = H + (int64_t)bs * FA_H;
    if (!(flags & 1))                                           // (1: the caller zeroed the accumulators with its other fills)
        if (int rc = zero_fill(acc, sizeof(float) * (size_t)bs * FA_ACC, st, "anr_frame_backward_adjoint (zero)")) return rc;
    const int nvb = d_ober2cano ? (V + FB_THREADS - 1) / FB_THREADS : 0;
    const int nrb = d_rays_body ? (R + FB_THREADS - 1) / FB_THREADS : 0;
    hipLaunchKernelGGL(frame_adjoint_kernel, dim3(nvb + nrb, bs), dim3(FB_THREADS), 0, st, betas, pose, transl, J0, JS, parents,
                       lbs_weights, T_template, template_bs == 1 ? (int64_t)0 : (int64_t)V * 16, rays_world, ray_stride, R, d_ober2cano,
                       d_rays_body, V, nvb, acc, goff, joints_transform, g_inv);
    if (d_ober2cano)
    {
        // (frames per workgroup, 16 frames: 1 -> 153 us, 2 -> 49, 4 -> 66 next to the weight gradients' tail; 2 frames: 43 / 29)
        const int bt = bs >= 3 ? 4 : bs;
        if (bt >= 4) hipLaunchKernelGGL(frame_offsets_kernel<4>, dim3(FA_H, (bs + 3) / 4), dim3(FO_THREADS), 0, st, goff, shapedirs, posedirs, V, H, bs);
        else if (bt == 2) hipLaunchKernelGGL(frame_offsets_kernel<2>, dim3(FA_H, (bs + 1) / 2), dim3(FO_THREADS), 0, st, goff, shapedirs, posedirs, V, H, bs);
        else hipLaunchKernelGGL(frame_offsets_kernel<1>, dim3(FA_H, bs), dim3(FO_THREADS), 0, st, goff, shapedirs, posedirs, V, H, bs);
    }
    hipLaunchKernelGGL(frame_params_kernel, dim3(FB_NP, bs), dim3(64), 0, st, betas, pose, transl, J0, JS, parents, acc,
                       d_ober2cano ? H : (const float*)nullptr, grads_out);
    return check_launch("anr_frame_backward_adjoint");
}
