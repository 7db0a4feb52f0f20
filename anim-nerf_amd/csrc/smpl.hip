// SMPL forward / linear blend skinning on the device (a2): smplx/body_models.py:289-387 + smplx/lbs.py:152-404.
// Per frame and per body: 6890 vertices, 24 joints — three small launches instead of ~100 framework kernels
// (einsum, 24 chained 4x4 matmuls, pads, cats).  Tables are the reference's buffers, untouched:
//   v_template[V,3], shapedirs[V,3,NB], posedirs[P=9(J-1), 3V], J_regressor[J,V], parents[J], lbs_weights[V,J].
#include "anr_common.h"

namespace anr {

constexpr int SMPL_MAX_J = 32;

// ---- 1. shape blend shapes + joint regression partials ------------------------------------------
// grid (ceil(V/256), bs): shape_off = shapedirs . betas ; v_shaped = v_template + shape_off ;
// joint_part[b, block, j] = sum over the block's vertices of J_regressor[j, v] * v_shaped[v]; the chain kernel adds the
// blocks in a fixed order (no atomics: the frame state must be reproducible bit for bit)
__global__ __launch_bounds__(256) void smpl_shape_kernel(const float* __restrict__ betas, int NB,
                                                         const float* __restrict__ v_template,
                                                         const float* __restrict__ shapedirs,
                                                         const float* __restrict__ J_regressor, int V, int J,
                                                         float* __restrict__ shape_off, float* __restrict__ v_shaped,
                                                         float* __restrict__ joints_rest) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float vs[3] = {0.f, 0.f, 0.f};
    if (v < V) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            for (int l = 0; l < NB; ++l) s += betas[b * NB + l] * shapedirs[((int64_t)v * 3 + c) * NB + l];
            shape_off[((int64_t)b * V + v) * 3 + c] = s;
            vs[c] = v_template[v * 3 + c] + s;
            v_shaped[((int64_t)b * V + v) * 3 + c] = vs[c];
        }
    }
    __shared__ float red[4][3];
    for (int j = 0; j < J; ++j) {
        const float w = (v < V) ? J_regressor[(int64_t)j * V + v] : 0.f;
        float p[3] = {w * vs[0], w * vs[1], w * vs[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = wave_sum(p[c]);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = p[0]; red[threadIdx.x >> 6][1] = p[1]; red[threadIdx.x >> 6][2] = p[2]; }
        __syncthreads();
        if (threadIdx.x < 3) {
            float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            joints_rest[(((int64_t)b * gridDim.x + blockIdx.x) * J + j) * 3 + threadIdx.x] = s;
        }
        __syncthreads();
    }
}

// ---- 2. Rodrigues + kinematic chain (one workgroup of 64 lanes per body) ------------------------------
// lbs.py:298-332 (angle = |rv + 1e-8|), :348-404.  Outputs A[J,4,4] (with transl added to its translation column,
// body_models.py:373), posed joints (+ transl), pose feature (R[1:] - I) flattened [9(J-1)].
__global__ __launch_bounds__(64) void smpl_chain_kernel(const float* __restrict__ pose, const float* __restrict__ transl,
                                                        const float* __restrict__ joints_rest, int n_part,
                                                        const int64_t* __restrict__ parents, int J,
                                                        float* __restrict__ A_out, float* __restrict__ joints_out,
                                                        float* __restrict__ feat_out) {
    const int b = blockIdx.x;
    const int j = threadIdx.x;
    __shared__ float Rm[SMPL_MAX_J][9];
    __shared__ float Jr[SMPL_MAX_J][3];
    __shared__ float Wd[SMPL_MAX_J][12];
    if (j < J) {
        const float* rv = pose + ((int64_t)b * J + j) * 3;
        const float x = rv[0], y = rv[1], z = rv[2];
        const float xe = x + 1e-8f, ye = y + 1e-8f, ze = z + 1e-8f;
        const float th = sqrtf(xe * xe + ye * ye + ze * ze);
        const float kx = x / th, ky = y / th, kz = z / th;
        const float s = sinf(th), c1 = 1.0f - cosf(th);
        // K = [[0,-kz,ky],[kz,0,-kx],[-ky,kx,0]];  R = I + s K + (1-c) K K
        const float K[9] = {0.f, -kz, ky, kz, 0.f, -kx, -ky, kx, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float kk = K[r * 3 + 0] * K[0 * 3 + c] + K[r * 3 + 1] * K[1 * 3 + c] + K[r * 3 + 2] * K[2 * 3 + c];
                Rm[j][r * 3 + c] = (r == c ? 1.0f : 0.0f) + s * K[r * 3 + c] + c1 * kk;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            for (int q = 0; q < n_part; ++q) acc += joints_rest[(((int64_t)b * n_part + q) * J + j) * 3 + c];
            Jr[j][c] = acc;
        }
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                feat_out[(int64_t)b * 9 * (J - 1) + (j - 1) * 9 + e] = Rm[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
    }
    __syncthreads();
    if (j == 0) {                                   // the chain is serial over 24 joints: one lane walks it
        for (int q = 0; q < J; ++q) {
            float loc[12];
            const int p = (q == 0) ? -1 : (int)parents[q];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                loc[r * 4 + 0] = Rm[q][r * 3 + 0]; loc[r * 4 + 1] = Rm[q][r * 3 + 1]; loc[r * 4 + 2] = Rm[q][r * 3 + 2];
                loc[r * 4 + 3] = Jr[q][r] - (p >= 0 ? Jr[p][r] : 0.0f);
            }
            if (p < 0) {
#pragma unroll
                for (int e = 0; e < 12; ++e) Wd[q][e] = loc[e];
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        Wd[q][r * 4 + c] = Wd[p][r * 4 + 0] * loc[0 * 4 + c] + Wd[p][r * 4 + 1] * loc[1 * 4 + c] +
                                           Wd[p][r * 4 + 2] * loc[2 * 4 + c] + (c == 3 ? Wd[p][r * 4 + 3] : 0.0f);
            }
        }
    }
    __syncthreads();
    if (j < J) {
        const float* tr = transl + (int64_t)b * 3;
        float* A = A_out + ((int64_t)b * J + j) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float shift = Wd[j][r * 4 + 0] * Jr[j][0] + Wd[j][r * 4 + 1] * Jr[j][1] + Wd[j][r * 4 + 2] * Jr[j][2];
            A[r * 4 + 0] = Wd[j][r * 4 + 0]; A[r * 4 + 1] = Wd[j][r * 4 + 1]; A[r * 4 + 2] = Wd[j][r * 4 + 2];
            A[r * 4 + 3] = (Wd[j][r * 4 + 3] - shift) + tr[r];
            joints_out[((int64_t)b * J + j) * 3 + r] = Wd[j][r * 4 + 3] + tr[r];
        }
        A[12] = 0.f; A[13] = 0.f; A[14] = 0.f; A[15] = 1.0f;
    }
}

// ---- 3. pose blend shapes + skinning (per vertex) ----------------------------------------------------------
// pose_off = feat . posedirs ; T = sum_j w[v,j] A_j (A already carries transl: sum_j w = 1 in SMPL tables, and the
// reference adds transl to T after the blend, body_models.py:374 — so blend the un-translated A and add transl);
// verts = T_rel v_posed + transl.
__global__ __launch_bounds__(256) void smpl_skin_kernel(const float* __restrict__ feat, int P,
                                                        const float* __restrict__ posedirs,
                                                        const float* __restrict__ lbs_weights,
                                                        const float* __restrict__ A, const float* __restrict__ transl,
                                                        const float* __restrict__ v_shaped, int V, int J,
                                                        float* __restrict__ pose_off, float* __restrict__ T_out,
                                                        float* __restrict__ verts) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float sA[SMPL_MAX_J][12];
    __shared__ float sf[9 * SMPL_MAX_J];
    const float* tr = transl + (int64_t)b * 3;
    for (int e = threadIdx.x; e < J * 12; e += blockDim.x) {
        const int j = e / 12, k = e % 12;
        float a = A[((int64_t)b * J + j) * 16 + k];
        if ((k & 3) == 3) a -= tr[k >> 2];                  // back to the un-translated relative transform
        sA[j][k] = a;
    }
    for (int e = threadIdx.x; e < P; e += blockDim.x) sf[e] = feat[(int64_t)b * P + e];
    __syncthreads();
    if (v >= V) return;
    float po[3] = {0.f, 0.f, 0.f};
    const int64_t row = (int64_t)3 * V;
    for (int p = 0; p < P; ++p) {
        const float f = sf[p];
        const float* pd = posedirs + p * row + (int64_t)v * 3;
        po[0] += f * pd[0]; po[1] += f * pd[1]; po[2] += f * pd[2];
    }
    float vp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pose_off[((int64_t)b * V + v) * 3 + c] = po[c];
        vp[c] = v_shaped[((int64_t)b * V + v) * 3 + c] + po[c];
    }
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int j = 0; j < J; ++j) {
        const float w = lbs_weights[(int64_t)v * J + j];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += w * sA[j][e];
    }
    float4* dst = reinterpret_cast<float4*>(T_out + ((int64_t)b * V + v) * 16);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float x = T[r * 4 + 0] * vp[0] + T[r * 4 + 1] * vp[1] + T[r * 4 + 2] * vp[2] + T[r * 4 + 3];
        verts[((int64_t)b * V + v) * 3 + r] = x + tr[r];
        dst[r] = make_float4(T[r * 4 + 0], T[r * 4 + 1], T[r * 4 + 2], T[r * 4 + 3] + tr[r]);
    }
    // the reference's T row 3 is sum_j w_j [0,0,0,1] = [0,0,0,sum w]
    float ws = 0.f;
    for (int j = 0; j < J; ++j) ws += lbs_weights[(int64_t)v * J + j];
    dst[3] = make_float4(0.f, 0.f, 0.f, ws);
}

}  // namespace anr

using namespace anr;

extern "C" int anr_smpl_forward(const float* betas, const float* pose, const float* transl, int bs, int NB,
                                const float* v_template, const float* shapedirs, const float* posedirs,
                                const float* J_regressor, const int64_t* parents, const float* lbs_weights, int V, int J,
                                float* verts, float* joints, float* A, float* T, float* shape_off, float* pose_off,
                                float* ws_v_shaped, float* ws_joints_rest, float* ws_feat, void* stream) {
    ANR_REQUIRE(betas && pose && transl && v_template && shapedirs && posedirs && J_regressor && parents && lbs_weights,
                ANR_E_BADARG, "anr_smpl_forward: null input");
    ANR_REQUIRE(verts && joints && A && T && shape_off && pose_off && ws_v_shaped && ws_joints_rest && ws_feat,
                ANR_E_BADARG, "anr_smpl_forward: null output/workspace");
    ANR_REQUIRE(bs > 0 && V > 0 && J > 1 && J <= SMPL_MAX_J && NB > 0, ANR_E_BADARG, "anr_smpl_forward: bs=%d V=%d J=%d NB=%d", bs, V, J, NB);
    ANR_REQUIRE(((uintptr_t)T & 15) == 0, ANR_E_ALIGN, "anr_smpl_forward: T must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    dim3 gv((V + 255) / 256, bs);
    hipLaunchKernelGGL(smpl_shape_kernel, gv, dim3(256), 0, st, betas, NB, v_template, shapedirs, J_regressor, V, J,
                       shape_off, ws_v_shaped, ws_joints_rest);
    hipLaunchKernelGGL(smpl_chain_kernel, dim3(bs), dim3(64), 0, st, pose, transl, ws_joints_rest, (int)gv.x, parents, J, A, joints,
                       ws_feat);
    hipLaunchKernelGGL(smpl_skin_kernel, gv, dim3(256), 0, st, ws_feat, 9 * (J - 1), posedirs, lbs_weights, A, transl,
                       ws_v_shaped, V, J, pose_off, T, verts);
    return check_launch("anr_smpl_forward");
}
