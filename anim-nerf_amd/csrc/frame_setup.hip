// The per-frame set-up of a step in TWO launches (round 5): what AnimNeRF.set_body_model, convert_to_body_model_space and
// clac_ober2cano_transform (models/anim_nerf.py:108-151; smplx/body_models.py:289-387, smplx/lbs.py:152-404) compute for a
// batch of frames — gather the frames' rows of the pose tables, SMPL (shape / pose blend shapes, Rodrigues, the 24-joint
// chain, skinning), body state and rays into the root-joint frame, the per-vertex observation -> canonical transforms —
// was eight launches of a few workgroups each, 122 us of pure dependency chain at the per-rank batch of the reference's
// 8-GPU run (2 frames).  Here:
//   1  frame_chain_kernel   one 64-lane workgroup per frame: table rows -> betas / pose / transl, rest joints, Rodrigues,
//                           chain, A (+ transl), the root transform's inverse, the joints in the root frame, the pose feature.
//                           The rest joints are J0 + JS . betas (J0 = J_regressor v_template, JS = J_regressor shapedirs:
//                           constants of the body model, the form anr_frame_backward differentiates) — no reduction over the
//                           6,890 vertices between the shape blend shapes and the chain, hence no launch boundary there.
//   2  frame_vertex_kernel  vertex blocks: shape / pose offsets, skinning transform, posed vertex, both into the root frame,
//                           ober2cano = T_template (G^-1 T)^-1 + offset differences; ray blocks: o', d', near', far'.
// The values agree with the separate kernels (anr_smpl_forward, anr_to_root_frame, anr_rays_to_body, anr_ober2cano) to fp32
// rounding (tests/test_gpu_parity.py::test_fused_frame_setup_matches_reference_and_the_separate_kernels); both training steps (explicit
// and autograd) go through this one, so they see the same bits.
#include "anr_common.h"

#pragma clang fp contract(off)

namespace anr {

constexpr int FS_J = 24, FS_NB = 10, FS_P = 9 * (FS_J - 1);

__device__ __forceinline__ void fs_affine_inverse12(const float* G, float (&I)[12]) {
    const float a = G[0], b = G[1], c = G[2], d = G[4], e = G[5], f = G[6], g = G[8], h = G[9], i = G[10];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const float det = a * c00 + b * c01 + c * c02;
    const float r = 1.0f / det;
    I[0] = c00 * r; I[1] = (c * h - b * i) * r; I[2] = (b * f - c * e) * r;
    I[4] = c01 * r; I[5] = (a * i - c * g) * r; I[6] = (c * d - a * f) * r;
    I[8] = c02 * r; I[9] = (b * g - a * h) * r; I[10] = (a * e - b * d) * r;
#pragma unroll
    for (int k = 0; k < 3; ++k) I[k * 4 + 3] = -(I[k * 4 + 0] * G[3] + I[k * 4 + 1] * G[7] + I[k * 4 + 2] * G[11]);
}

// frame_idx != NULL: the four arrays are TABLES (betas_w[betas_rows][10] — row min(idx, rows - 1): the shipped tables share one
// betas row —, go_w / bp_w / tr_w[table_rows][3 / 69 / 3]); NULL: per-frame arrays (go_w[bs][3], bp_w[bs][69], ...).
__global__ __launch_bounds__(64) void frame_chain_kernel(const int64_t* __restrict__ frame_idx, const float* __restrict__ betas_w,
                                                         int betas_rows, const float* __restrict__ go_w, const float* __restrict__ bp_w,
                                                         const float* __restrict__ tr_w, const float* __restrict__ J0,
                                                         const float* __restrict__ JS, const int64_t* __restrict__ parents,
                                                         float* __restrict__ betas_out, float* __restrict__ pose_out,
                                                         float* __restrict__ transl_out, float* __restrict__ A_out,
                                                         float* __restrict__ joints_root_out, float* __restrict__ feat_out,
                                                         float* __restrict__ ginv_out, float* __restrict__ groot_out,
                                                         int table_rows) {
    const int b = blockIdx.x, j = threadIdx.x;
    __shared__ float sB[FS_NB], sP[3 * FS_J], sT[3];
    // (the narrow part of the chain — 24 joints, 85 scalars per frame — accumulates in DOUBLE: it costs nothing here, and the
    // pose gradients downstream are cancelling sums over 6,890 vertices that amplify every rounding of A ~500 x)
    __shared__ float Rm[FS_J][9], sI[12];
    __shared__ double Jr[FS_J][3], Wd[FS_J][12];
    // A frame index outside the tables (nn.Embedding raises there: models/body_model_params.py:5-68) must neither read out of
    // bounds nor pass for a pose: row 0 is read instead and the frame's parameters are NaN from here on — the step's loss and
    // every gradient of that frame say so at the next host read, no synchronisation here (table_rows = 0: rows unknown, no check)
    int64_t row = frame_idx ? frame_idx[b] : b;
    const bool bad = frame_idx != nullptr && table_rows > 0 && (row < 0 || row >= table_rows);
    if (bad) row = 0;
    const float poison = bad ? __builtin_nanf("") : 0.0f;
    const int64_t brow = frame_idx ? (row < betas_rows ? row : betas_rows - 1) : b;
    if (j < FS_NB) { sB[j] = betas_w[brow * FS_NB + j] + poison; betas_out[b * FS_NB + j] = sB[j]; }
    for (int e = j; e < 3 * FS_J; e += 64) {
        const float v = (e < 3 ? go_w[row * 3 + e] : bp_w[row * 69 + (e - 3)]) + poison;
        sP[e] = v;
        pose_out[b * 3 * FS_J + e] = v;
    }
    if (j < 3) { sT[j] = tr_w[row * 3 + j] + poison; transl_out[b * 3 + j] = sT[j]; }
    __syncthreads();
    if (j < FS_J) {
#pragma clang fp contract(fast)                                          // (the SMPL part rounds as csrc/smpl.hip does: fused multiply-adds)
        const float x = sP[3 * j], y = sP[3 * j + 1], z = sP[3 * j + 2];
        const float xe = x + 1e-8f, ye = y + 1e-8f, ze = z + 1e-8f;          // lbs.py:316: angle = |rv + 1e-8|
        const float th = sqrtf(xe * xe + ye * ye + ze * ze);
        const float kx = x / th, ky = y / th, kz = z / th;
        const float s = sinf(th), c1 = 1.0f - cosf(th);
        const float K[9] = {0.f, -kz, ky, kz, 0.f, -kx, -ky, kx, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float kk = K[r * 3 + 0] * K[0 * 3 + c] + K[r * 3 + 1] * K[1 * 3 + c] + K[r * 3 + 2] * K[2 * 3 + c];
                Rm[j][r * 3 + c] = (r == c ? 1.0f : 0.0f) + s * K[r * 3 + c] + c1 * kk;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = J0[j * 3 + c];
            for (int k = 0; k < FS_NB; ++k) acc += (double)sB[k] * (double)JS[(j * 3 + c) * FS_NB + k];
            Jr[j][c] = acc;
        }
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                feat_out[(int64_t)b * FS_P + (j - 1) * 9 + e] = Rm[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
    }
    __syncthreads();
    // world_q = world_parent . [R_q | J_q - J_parent], one element (r, c) per lane, the tree LEVEL BY LEVEL (9 levels for SMPL's
    // skeleton, five joints of a level per pass) instead of joint by joint (24 dependent LDS round trips)
    __shared__ int lvl_start[FS_J + 1], lvl_joint[FS_J], n_levels_s, par_s[FS_J], depth_s[FS_J];
    if (j < FS_J) par_s[j] = (int)parents[j];
    __syncthreads();
    if (j < FS_J) {
        int dpt = 0;
        for (int k = j; k > 0; k = par_s[k]) ++dpt;                        // (parents[q] < q: tree order)
        depth_s[j] = dpt;
    }
    __syncthreads();
    if (j < FS_J) {
        const int mine = depth_s[j];
        int rank = 0;
        for (int q = 0; q < FS_J; ++q) rank += (depth_s[q] < mine || (depth_s[q] == mine && q < j)) ? 1 : 0;
        lvl_joint[rank] = j;
    }
    if (j <= FS_J) {
        int c = 0, deepest = 0;
        for (int q = 0; q < FS_J; ++q) { c += depth_s[q] < j ? 1 : 0; deepest = max(deepest, depth_s[q]); }
        lvl_start[j] = c;
        if (j == 0) n_levels_s = deepest + 1;
    }
    __syncthreads();
    const int n_levels = n_levels_s;
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        const int first = lvl_start[lvl], n_here = lvl_start[lvl + 1] - first;
        for (int s0 = 0; s0 < n_here; s0 += 5) {
            const int slot = j / 12, e = j % 12;
            if (j < 60 && s0 + slot < n_here) {
#pragma clang fp contract(fast)
                const int q = lvl_joint[first + s0 + slot];
                const int r = e >> 2, c = e & 3;
                const int p = q == 0 ? -1 : par_s[q];
                auto loc = [&](int k, int cc) { return cc < 3 ? (double)Rm[q][k * 3 + cc] : Jr[q][k] - (p >= 0 ? Jr[p][k] : 0.0); };
                double w;
                if (p < 0) w = loc(r, c);
                else w = Wd[p][r * 4 + 0] * loc(0, c) + Wd[p][r * 4 + 1] * loc(1, c) + Wd[p][r * 4 + 2] * loc(2, c) + (c == 3 ? Wd[p][r * 4 + 3] : 0.0);
                Wd[q][e] = w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    float Aj[12];
    if (j < FS_J) {
#pragma clang fp contract(fast)
        float* A = A_out + ((int64_t)b * FS_J + j) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double shift = Wd[j][r * 4 + 0] * Jr[j][0] + Wd[j][r * 4 + 1] * Jr[j][1] + Wd[j][r * 4 + 2] * Jr[j][2];
            Aj[r * 4 + 0] = (float)Wd[j][r * 4 + 0]; Aj[r * 4 + 1] = (float)Wd[j][r * 4 + 1]; Aj[r * 4 + 2] = (float)Wd[j][r * 4 + 2];
            Aj[r * 4 + 3] = (float)((Wd[j][r * 4 + 3] - shift) + (double)sT[r]);   // body_models.py:373: transl on the translation column
#pragma unroll
            for (int c = 0; c < 4; ++c) A[r * 4 + c] = Aj[r * 4 + c];
        }
        A[12] = 0.f; A[13] = 0.f; A[14] = 0.f; A[15] = 1.0f;
    }
    if (j == 0) {
        // models/anim_nerf.py:128-145: the root transform (joint 0, with transl) and its inverse; the reference keeps the
        // product G^-1 G, not the identity
        float I[12];
        fs_affine_inverse12(Aj, I);
        float* o = ginv_out + b * 16;
        float* gr = groot_out + b * 16;
#pragma unroll
        for (int k = 0; k < 12; ++k) { o[k] = I[k]; sI[k] = I[k]; }
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gr[r * 4 + c] = I[r * 4 + 0] * Aj[c] + I[r * 4 + 1] * Aj[4 + c] + I[r * 4 + 2] * Aj[8 + c] + (c == 3 ? I[r * 4 + 3] : 0.0f);
        gr[12] = 0.f; gr[13] = 0.f; gr[14] = 0.f; gr[15] = 1.f;
    }
    __syncthreads();
    if (j < FS_J) {                                          // posed joints (+ transl), then into the root frame
        float pj[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pj[r] = (float)(Wd[j][r * 4 + 3] + (double)sT[r]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            joints_root_out[((int64_t)b * FS_J + j) * 3 + r] = sI[r * 4 + 0] * pj[0] + sI[r * 4 + 1] * pj[1] + sI[r * 4 + 2] * pj[2] + sI[r * 4 + 3];
    }
}

// 256 threads = 64 vertices x 4 slices of the 207 pose features (the pose blend shapes are a 621-load chain per vertex when
// one thread walks them alone), or 256 rays.
// BT frames per vertex workgroup: a vertex's 621 blend-shape values (207 rows x 768 B per workgroup; 17 MB over the mesh) are
// loaded ONCE and applied to the features of BT frames — the pass is that stream, and a workgroup per (block, frame) read it
// once per frame: 46 us at 16 frames.  (An XCD-aware order of those workgroups — the frames of a block back to back on one
// XCD, the stream from its L2 — made the frames' workgroups walk the same lines in lockstep, on one channel at a time: 62 us.)
// Per frame the arithmetic and its order are what they were: the same bits.
template <int BT>
__global__ __launch_bounds__(256) void frame_vertex_kernel(
    const float* __restrict__ betas, const float* __restrict__ transl, const float* __restrict__ A, const float* __restrict__ feat,
    const float* __restrict__ ginv, const float* __restrict__ v_template, const float* __restrict__ shapedirs,
    const float* __restrict__ posedirs, const float* __restrict__ lbs_weights, int V, const float* __restrict__ T_templ,
    const float* __restrict__ so_templ, const float* __restrict__ po_templ, int64_t templ_stride_T, int64_t templ_stride_o,
    const float* __restrict__ rays_world, int ray_stride, int R, int n_vblocks, float* __restrict__ shape_off,
    float* __restrict__ pose_off, float* __restrict__ verts_root, float* __restrict__ T_root, float* __restrict__ o2c,
    float* __restrict__ rays_body, int bs, int n_rblocks) {
    const int n_groups = (bs + BT - 1) / BT;
    __shared__ float sA[BT][FS_J][12], sf[BT][FS_P], sB[BT][FS_NB], sI[BT][12], sT[BT][3];
    __shared__ float sPo[BT][4][64][3];
    if ((int)blockIdx.x >= n_vblocks * n_groups) {           // ---- rays: models/anim_nerf.py:128-137
        const int t = (int)blockIdx.x - n_vblocks * n_groups;
        const int b = t / n_rblocks;
        const int r = (t % n_rblocks) * 256 + threadIdx.x;
        if (r >= R) return;
        const float* G = ginv + b * 16;
        const float* s = rays_world + ((int64_t)b * R + r) * ray_stride;
        const float o[3] = {s[0], s[1], s[2]}, d[3] = {s[3], s[4], s[5]};
        float on[3], dn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            on[a] = G[a * 4 + 0] * o[0] + G[a * 4 + 1] * o[1] + G[a * 4 + 2] * o[2] + G[a * 4 + 3];
            dn[a] = G[a * 4 + 0] * d[0] + G[a * 4 + 1] * d[1] + G[a * 4 + 2] * d[2];
        }
        const float dist = sqrtf(on[0] * on[0] + on[1] * on[1] + on[2] * on[2]);
        float4* dst = reinterpret_cast<float4*>(rays_body + ((int64_t)b * R + r) * 8);
        dst[0] = make_float4(on[0], on[1], on[2], dn[0]);
        dst[1] = make_float4(dn[1], dn[2], fmaxf(s[6], dist - 1.0f), fminf(s[7], dist + 1.0f));
        return;
    }
    const int bx = (int)blockIdx.x % n_vblocks, b0 = ((int)blockIdx.x / n_vblocks) * BT;
    const int nb = min(BT, bs - b0);                           // frames of this group
    for (int e = threadIdx.x; e < nb * FS_J * 12; e += 256) {
        const int t = e / (FS_J * 12), j = (e / 12) % FS_J, k = e % 12;
        float a = A[((int64_t)(b0 + t) * FS_J + j) * 16 + k];
        if ((k & 3) == 3) a -= transl[(int64_t)(b0 + t) * 3 + (k >> 2)];      // back to the un-translated relative transform
        sA[t][j][k] = a;
    }
    for (int e = threadIdx.x; e < BT * FS_P; e += 256) {
        const int t = e / FS_P, q = e % FS_P;
        sf[t][q] = t < nb ? feat[(int64_t)(b0 + t) * FS_P + q] : 0.0f;
    }
    if (threadIdx.x < nb * FS_NB) sB[threadIdx.x / FS_NB][threadIdx.x % FS_NB] = betas[(b0 + threadIdx.x / FS_NB) * FS_NB + threadIdx.x % FS_NB];
    if (threadIdx.x < nb * 12) sI[threadIdx.x / 12][threadIdx.x % 12] = ginv[(b0 + threadIdx.x / 12) * 16 + threadIdx.x % 12];
    if (threadIdx.x < nb * 3) sT[threadIdx.x / 3][threadIdx.x % 3] = transl[(int64_t)(b0 + threadIdx.x / 3) * 3 + threadIdx.x % 3];
    __syncthreads();
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int v = bx * 64 + lane;
    const bool live = v < V;
    // pose blend shapes: slice s takes features s, s + 4, ... (lbs.py:152-251: pose_offsets = feat . posedirs)
    float pot_[BT][3];
#pragma unroll
    for (int t = 0; t < BT; ++t) { pot_[t][0] = 0.f; pot_[t][1] = 0.f; pot_[t][2] = 0.f; }
    if (live) {
#pragma clang fp contract(fast)                                          // (the SMPL part rounds as csrc/smpl.hip does)
        const int64_t row = (int64_t)3 * V;
        for (int p = slice; p < FS_P; p += 4) {
            const float* pd = posedirs + p * row + (int64_t)v * 3;
            const float d0 = pd[0], d1 = pd[1], d2 = pd[2];
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const float f = sf[t][p];
                pot_[t][0] += f * d0; pot_[t][1] += f * d1; pot_[t][2] += f * d2;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < BT; ++t) { sPo[t][slice][lane][0] = pot_[t][0]; sPo[t][slice][lane][1] = pot_[t][1]; sPo[t][slice][lane][2] = pot_[t][2]; }
    __syncthreads();
    // wavefront t finishes frame t of the group (BT <= 4 wavefronts): the frames' chains of dependent loads side by side, not
    // one behind the other in wavefront 0
    static_assert(BT <= 4, "one wavefront per frame of the group");
    if (slice >= nb || !live) return;
    {
        const int t = slice;
        const int b = b0 + t;
        float po[3], so[3], T[16], x[3], ws = 0.f;
        {
#pragma clang fp contract(fast)
            float vs[3], vp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                po[c] = (sPo[t][0][lane][c] + sPo[t][1][lane][c]) + (sPo[t][2][lane][c] + sPo[t][3][lane][c]);
                float s = 0.f;
                for (int l = 0; l < FS_NB; ++l) s += sB[t][l] * shapedirs[((int64_t)v * 3 + c) * FS_NB + l];
                so[c] = s;
                vs[c] = v_template[v * 3 + c] + s;
                vp[c] = vs[c] + po[c];
                shape_off[((int64_t)b * V + v) * 3 + c] = s;
                pose_off[((int64_t)b * V + v) * 3 + c] = po[c];
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
            for (int j = 0; j < FS_J; ++j) {
                const float w = lbs_weights[(int64_t)v * FS_J + j];
                ws += w;
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] += w * sA[t][j][e];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                x[r] = (T[r * 4 + 0] * vp[0] + T[r * 4 + 1] * vp[1] + T[r * 4 + 2] * vp[2] + T[r * 4 + 3]) + sT[t][r];
                T[r * 4 + 3] += sT[t][r];
            }
        }
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = ws;          // the reference's T row 3 is sum_j w_j [0,0,0,1]
        // into the root frame (models/anim_nerf.py:138-144): verts, and T as the full 4x4 product torch computes
        float Xr[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            verts_root[((int64_t)b * V + v) * 3 + r] = sI[t][r * 4 + 0] * x[0] + sI[t][r * 4 + 1] * x[1] + sI[t][r * 4 + 2] * x[2] + sI[t][r * 4 + 3];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                Xr[r * 4 + c] = sI[t][r * 4 + 0] * T[c] + sI[t][r * 4 + 1] * T[4 + c] + sI[t][r * 4 + 2] * T[8 + c] + sI[t][r * 4 + 3] * T[12 + c];
        }
        float4* t4 = reinterpret_cast<float4*>(T_root + ((int64_t)b * V + v) * 16);
#pragma unroll
        for (int r = 0; r < 3; ++r) t4[r] = make_float4(Xr[r * 4 + 0], Xr[r * 4 + 1], Xr[r * 4 + 2], Xr[r * 4 + 3]);
        t4[3] = make_float4(0.f, 0.f, 0.f, ws);
        // observation -> canonical (models/anim_nerf.py:147-151): T_template (T_root)^-1, offsets on the translation
        float I[12];
        fs_affine_inverse12(Xr, I);
        const float* sot = so_templ + b * templ_stride_o + (int64_t)v * 3;
        const float* pot = po_templ + b * templ_stride_o + (int64_t)v * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float tt = I[r * 4 + 3];
            tt += sot[r] - so[r];
            tt += pot[r] - po[r];
            I[r * 4 + 3] = tt;
        }
        const float4* pb = reinterpret_cast<const float4*>(T_templ + b * templ_stride_T + (int64_t)v * 16);
        float4* dst = reinterpret_cast<float4*>(o2c + ((int64_t)b * V + v) * 16);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float4 B = pb[r];
            const float o0 = B.x * I[0] + B.y * I[4] + B.z * I[8];
            const float o1 = B.x * I[1] + B.y * I[5] + B.z * I[9];
            const float o2 = B.x * I[2] + B.y * I[6] + B.z * I[10];
            const float o3 = B.x * I[3] + B.y * I[7] + B.z * I[11] + B.w;
            dst[r] = make_float4(o0, o1, o2, o3);
        }
        dst[3] = make_float4(0.f, 0.f, 0.f, 1.f);
    }
}

}  // namespace anr

using namespace anr;

extern "C" int anr_frame_setup_rows(const int64_t* frame_idx, int table_rows, const float* betas_w, int betas_rows, const float* global_orient_w,
                               const float* body_pose_w, const float* transl_w, int bs, const float* J0, const float* JS,
                               const int64_t* parents, const float* v_template, const float* shapedirs, const float* posedirs,
                               const float* lbs_weights, int V, int J, int NB, const float* T_template,
                               const float* shape_off_template, const float* pose_off_template, int template_bs,
                               const float* rays_world, int ray_stride, int R, float* betas_out, float* pose_out, float* transl_out,
                               float* A_out, float* joints_root_out, float* g_inv_out, float* g_root_out, float* shape_off_out,
                               float* pose_off_out, float* verts_root_out, float* T_root_out, float* ober2cano_out,
                               float* rays_body_out, float* ws_feat, void* stream) {
    ANR_REQUIRE(table_rows >= 0, ANR_E_BADARG, "anr_frame_setup: table_rows=%d", table_rows);
    ANR_REQUIRE(betas_w && global_orient_w && body_pose_w && transl_w && J0 && JS && parents && v_template && shapedirs && posedirs &&
                lbs_weights && T_template && shape_off_template && pose_off_template, ANR_E_BADARG, "anr_frame_setup: null input");
    ANR_REQUIRE(betas_out && pose_out && transl_out && A_out && joints_root_out && g_inv_out && g_root_out && shape_off_out &&
                pose_off_out && verts_root_out && T_root_out && ober2cano_out && ws_feat, ANR_E_BADARG, "anr_frame_setup: null output");
    ANR_REQUIRE(bs > 0 && V > 0 && J == FS_J && NB == FS_NB && betas_rows > 0 && (template_bs == 1 || template_bs == bs), ANR_E_BADARG,
                "anr_frame_setup: bs=%d V=%d J=%d (24) NB=%d (10) betas_rows=%d template_bs=%d", bs, V, J, NB, betas_rows, template_bs);
    ANR_REQUIRE(R == 0 || (rays_world && rays_body_out && ray_stride >= 8), ANR_E_BADARG, "anr_frame_setup: rays R=%d stride=%d", R, ray_stride);
    ANR_REQUIRE((((uintptr_t)T_template | (uintptr_t)T_root_out | (uintptr_t)ober2cano_out | (uintptr_t)rays_body_out) & 15) == 0, ANR_E_ALIGN,
                "anr_frame_setup: T_template / T_root / ober2cano / rays_body must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_chain_kernel, dim3(bs), dim3(64), 0, st, frame_idx, betas_w, betas_rows, global_orient_w, body_pose_w,
                       transl_w, J0, JS, parents, betas_out, pose_out, transl_out, A_out, joints_root_out, ws_feat, g_inv_out, g_root_out,
                       table_rows);
    const int nvb = (V + 63) / 64, nrb = (R + 255) / 256;
#define ANR_FRAME_VERTEX(BT)                                                                                                      \
    hipLaunchKernelGGL(frame_vertex_kernel<BT>, dim3((unsigned)(nvb * ((bs + BT - 1) / BT) + nrb * bs)), dim3(256), 0, st, betas_out,       \
                       transl_out, A_out, ws_feat, g_inv_out, v_template, shapedirs, posedirs, lbs_weights, V, T_template,                 \
                       shape_off_template, pose_off_template, template_bs == 1 ? (int64_t)0 : (int64_t)V * 16,                            \
                       template_bs == 1 ? (int64_t)0 : (int64_t)V * 3, rays_world, ray_stride, R, nvb, shape_off_out, pose_off_out,      \
                       verts_root_out, T_root_out, ober2cano_out, rays_body_out, bs, nrb)
    {
        // (frames per workgroup, 16 frames: 1 -> 66 us, 2 -> 55, 4 -> 38; 2 frames: 22 / 25)
        const int bt = bs >= 3 ? 4 : bs;
        if (bt >= 4) ANR_FRAME_VERTEX(4); else if (bt == 2) ANR_FRAME_VERTEX(2); else ANR_FRAME_VERTEX(1);
    }
#undef ANR_FRAME_VERTEX
    return check_launch("anr_frame_setup");
}

extern "C" int anr_frame_setup(const int64_t* frame_idx, const float* betas_w, int betas_rows, const float* global_orient_w,
                               const float* body_pose_w, const float* transl_w, int bs, const float* J0, const float* JS,
                               const int64_t* parents, const float* v_template, const float* shapedirs, const float* posedirs,
                               const float* lbs_weights, int V, int J, int NB, const float* T_template,
                               const float* shape_off_template, const float* pose_off_template, int template_bs,
                               const float* rays_world, int ray_stride, int R, float* betas_out, float* pose_out, float* transl_out,
                               float* A_out, float* joints_root_out, float* g_inv_out, float* g_root_out, float* shape_off_out,
                               float* pose_off_out, float* verts_root_out, float* T_root_out, float* ober2cano_out,
                               float* rays_body_out, float* ws_feat, void* stream) {
    return anr_frame_setup_rows(frame_idx, 0, betas_w, betas_rows, global_orient_w, body_pose_w, transl_w, bs, J0, JS, parents, v_template,
                                shapedirs, posedirs, lbs_weights, V, J, NB, T_template, shape_off_template, pose_off_template, template_bs,
                                rays_world, ray_stride, R, betas_out, pose_out, transl_out, A_out, joints_root_out, g_inv_out, g_root_out,
                                shape_off_out, pose_off_out, verts_root_out, T_root_out, ober2cano_out, rays_body_out, ws_feat, stream);
}
