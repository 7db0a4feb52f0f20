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

// grads[bs][85] = (betas 10 | global_orient 3 | body_pose 69 | transl 3)
__global__ __launch_bounds__(FB_THREADS, 2) void frame_backward_kernel(
    const float* __restrict__ betas, const float* __restrict__ pose, const float* __restrict__ transl,
    const float* __restrict__ J0, const float* __restrict__ JS, const int64_t* __restrict__ parents,
    const float* __restrict__ lbs_w, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ T_templ, int64_t templ_stride, const float* __restrict__ rays_world, int ray_stride, int R,
    const float* __restrict__ d_o2c, const float* __restrict__ d_rays, int V, float* __restrict__ grads) {
    const int b = blockIdx.y, pi = blockIdx.x;
    __shared__ float sA[FB_J][12][2];            // joint transforms relative to the rest pose (no transl): value, tangent
    __shared__ float sG[12][2];                  // inverse of the root transform
    __shared__ float sTr[3][2];                  // transl
    __shared__ float sFeat[9];                   // tangent of the 9 pose-feature entries of the parameter's joint
    __shared__ float sRed[FB_THREADS / 64];
    const int pj = (pi >= FB_NB && pi < FB_NB + 3 * FB_J) ? (pi - FB_NB) / 3 : -1;      // joint of a pose parameter

    // ---- the joint chain of this (frame, parameter), value + tangent, cooperatively and out of LDS (a single thread with
    // its 24 transforms in private memory spends milliseconds in scratch accesses)
    __shared__ float sRot[FB_J][9][2], sJr[FB_J][3][2], sWorld[FB_J][12][2];
    auto seed = [&](int idx, float val) { return Dual(val, idx == pi ? 1.0f : 0.0f); };
    if (threadIdx.x < FB_J) {
        const int j = threadIdx.x;
        for (int c = 0; c < 3; ++c) {                              // rest joints: J_regressor . (v_template + shapedirs . betas)
            Dual s(J0[j * 3 + c]);
#pragma unroll 1
            for (int k = 0; k < FB_NB; ++k) s = s + seed(k, betas[b * FB_NB + k]) * Dual(JS[(j * 3 + c) * FB_NB + k]);
            sJr[j][c][0] = s.v; sJr[j][c][1] = s.d;
        }
        // Rodrigues, angle = |rv + 1e-8| (smplx/lbs.py:316)
        Dual rv[3];
        for (int c = 0; c < 3; ++c) rv[c] = seed(FB_NB + 3 * j + c, pose[(b * FB_J + j) * 3 + c]);
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
                sRot[j][r * 3 + c][0] = m.v; sRot[j][r * 3 + c][1] = m.d;
                if (j == pj) sFeat[r * 3 + c] = m.d;              // pose feature = R - I: same tangent
            }
    }
    if (threadIdx.x < 9 && pj < 1) sFeat[threadIdx.x] = 0.0f;      // the root joint has no pose blend shapes
    if (threadIdx.x < 3) {
        const Dual t = seed(FB_NB + 3 * FB_J + threadIdx.x, transl[b * 3 + threadIdx.x]);
        sTr[threadIdx.x][0] = t.v; sTr[threadIdx.x][1] = t.d;
    }
    __syncthreads();
    // world_j = world_parent . [R_j | J_j - J_parent], one element (r, c) per lane, joints in tree order
#pragma unroll 1
    for (int j = 0; j < FB_J; ++j) {
        if (threadIdx.x < 12) {
            const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
            const int par = (int)parents[j];
            auto local = [&](int k, int cc) {                     // element (k, cc) of joint j's local transform
                if (cc < 3) return Dual(sRot[j][k * 3 + cc][0], sRot[j][k * 3 + cc][1]);
                Dual t(sJr[j][k][0], sJr[j][k][1]);
                if (j > 0) t = t - Dual(sJr[par][k][0], sJr[par][k][1]);
                return t;
            };
            Dual w;
            if (j == 0) {
                w = local(r, c);
            } else {
                w = Dual(sWorld[par][r * 4 + 0][0], sWorld[par][r * 4 + 0][1]) * local(0, c) +
                    Dual(sWorld[par][r * 4 + 1][0], sWorld[par][r * 4 + 1][1]) * local(1, c) +
                    Dual(sWorld[par][r * 4 + 2][0], sWorld[par][r * 4 + 2][1]) * local(2, c);
                if (c == 3) w = w + Dual(sWorld[par][r * 4 + 3][0], sWorld[par][r * 4 + 3][1]);
            }
            sWorld[j][threadIdx.x][0] = w.v; sWorld[j][threadIdx.x][1] = w.d;
        }
        __syncthreads();
    }
    // relative to the rest pose: t -= R_world . J_rest
    for (int i = threadIdx.x; i < FB_J * 12; i += FB_THREADS) {
        const int j = i / 12, e = i % 12, r = e >> 2, c = e & 3;
        Dual a(sWorld[j][e][0], sWorld[j][e][1]);
        if (c == 3)
            a = a - (Dual(sWorld[j][r * 4 + 0][0], sWorld[j][r * 4 + 0][1]) * Dual(sJr[j][0][0], sJr[j][0][1]) +
                     Dual(sWorld[j][r * 4 + 1][0], sWorld[j][r * 4 + 1][1]) * Dual(sJr[j][1][0], sJr[j][1][1]) +
                     Dual(sWorld[j][r * 4 + 2][0], sWorld[j][r * 4 + 2][1]) * Dual(sJr[j][2][0], sJr[j][2][1]));
        sA[j][e][0] = a.v; sA[j][e][1] = a.d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                        // global transform = A_0 + transl (body_models.py:373)
        Aff G;
        for (int e = 0; e < 12; ++e) G.m[e] = Dual(sA[0][e][0], sA[0][e][1]);
        for (int r = 0; r < 3; ++r) G.m[r * 4 + 3] = G.m[r * 4 + 3] + Dual(sTr[r][0], sTr[r][1]);
        const Aff Gi0 = inverse(G);
        for (int e = 0; e < 12; ++e) { sG[e][0] = Gi0.m[e].v; sG[e][1] = Gi0.m[e].d; }
    }
    __syncthreads();

    Aff Gi;
#pragma unroll
    for (int e = 0; e < 12; ++e) Gi.m[e] = Dual(sG[e][0], sG[e][1]);
    float acc = 0.0f;
    // ---- every vertex: tangent of T_template . inverse(G^-1 . T_v) (+ offsets), dotted with dL/d ober2cano
#pragma unroll 1
    for (int v = threadIdx.x; v < V; v += FB_THREADS) {
        Aff T;
#pragma unroll
        for (int e = 0; e < 12; ++e) T.m[e] = Dual(0.f);
        const float* w = lbs_w + (int64_t)v * FB_J;
#pragma unroll 1
        for (int j = 0; j < FB_J; ++j) {
            const float wj = w[j];
            if (wj != 0.0f) {
#pragma unroll
                for (int e = 0; e < 12; ++e) { T.m[e].v += wj * sA[j][e][0]; T.m[e].d += wj * sA[j][e][1]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { T.m[r * 4 + 3].v += sTr[r][0]; T.m[r * 4 + 3].d += sTr[r][1]; }
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
                for (int e = 0; e < 9; ++e) dpo += sFeat[e] * pd[(int64_t)e * 3 * V];
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
    if (d_rays != nullptr) {
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

}  // namespace anr

using namespace anr;

extern "C" int anr_frame_backward(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                                  const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                                  const float* posedirs, int V, const float* T_template, int template_bs,
                                  const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                                  const float* d_rays_body, float* grads_out, void* stream) {
    ANR_REQUIRE(betas && pose && transl && J0 && JS && parents && lbs_weights && shapedirs && posedirs && T_template &&
                (d_ober2cano || d_rays_body) && grads_out, ANR_E_BADARG, "anr_frame_backward: null pointer");
    ANR_REQUIRE(bs > 0 && V > 0 && (template_bs == 1 || template_bs == bs), ANR_E_BADARG, "anr_frame_backward: bs=%d V=%d template_bs=%d",
                bs, V, template_bs);
    ANR_REQUIRE(!d_rays_body || (rays_world && R > 0 && ray_stride >= 8), ANR_E_BADARG, "anr_frame_backward: rays R=%d stride=%d", R, ray_stride);
    hipLaunchKernelGGL(frame_backward_kernel, dim3(FB_NP, bs), dim3(FB_THREADS), 0, (hipStream_t)stream, betas, pose, transl,
                       J0, JS, parents, lbs_weights, shapedirs, posedirs, T_template,
                       template_bs == 1 ? (int64_t)0 : (int64_t)V * 16, rays_world, ray_stride, R, d_ober2cano, d_rays_body, V,
                       grads_out);
    return check_launch("anr_frame_backward");
}
