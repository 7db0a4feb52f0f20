// Per-camera / per-frame / per-ray elementwise kernels (HBM-bound, one pass each):
//   anr_ray_gen, anr_rays_to_body, anr_ober2cano, anr_sample_coarse, anr_points_from_rays.
// Floating-point contraction is OFF in this file so that products and sums round exactly like
// the reference's separate torch ops (z and x = o + z d come out bit-identical).
#include "anr_common.h"
#include <stdarg.h>

#pragma clang fp contract(off)

namespace anr {

// ------------------------------------------------------------------ error plumbing
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------ a1 ray generation
// reference: datasets/anim_nerf_dataset.py:56-85
__global__ void ray_gen_kernel(const float* __restrict__ c2w, const float* __restrict__ focal,
                               const float* __restrict__ center, int H, int W, float near, float far,
                               float* __restrict__ rays) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    int j = (int)(p / W), i = (int)(p % W);
    float dx = ((float)i - center[0]) / focal[0];
    float dy = -((float)j - center[1]) / focal[1];
    float dz = -1.0f;
    float n = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= n; dy /= n; dz /= n;
    float o[8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = c2w[a * 4 + 3];
        o[3 + a] = dx * c2w[a * 4 + 0] + dy * c2w[a * 4 + 1] + dz * c2w[a * 4 + 2];
    }
    o[6] = near; o[7] = far;
    float4* dst = reinterpret_cast<float4*>(rays + p * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// ------------------------------------------------------------------ a4 rays -> root frame
// reference: models/anim_nerf.py:128-137
__global__ void rays_to_body_kernel(const float* __restrict__ ginv, const float* __restrict__ rin,
                                    float* __restrict__ rout, int R, int stride) {
    int b = blockIdx.y;
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* G = ginv + b * 16;
    const float* s = rin + ((int64_t)b * R + r) * stride;
    float o[3] = {s[0], s[1], s[2]}, d[3] = {s[3], s[4], s[5]};
    float on[3], dn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        on[a] = G[a * 4 + 0] * o[0] + G[a * 4 + 1] * o[1] + G[a * 4 + 2] * o[2] + G[a * 4 + 3];
        dn[a] = G[a * 4 + 0] * d[0] + G[a * 4 + 1] * d[1] + G[a * 4 + 2] * d[2];
    }
    float dist = sqrtf(on[0] * on[0] + on[1] * on[1] + on[2] * on[2]);
    float near = fmaxf(s[6], dist - 1.0f);
    float far = fminf(s[7], dist + 1.0f);
    float4* dst = reinterpret_cast<float4*>(rout + ((int64_t)b * R + r) * 8);
    dst[0] = make_float4(on[0], on[1], on[2], dn[0]);
    dst[1] = make_float4(dn[1], dn[2], near, far);
}

// ------------------------------------------------------------------ a4 body state -> root frame
// reference: models/anim_nerf.py:128-145: G^-1 (closed form: the root transform is affine), then verts, joints, the
// root transform itself and the per-vertex transforms move into the root-joint frame.  One launch instead of a LAPACK
// inverse (a host synchronisation on this stack) + four batched products of 4x4 matrices.
__device__ __forceinline__ void affine_inverse12(const float* __restrict__ G, float (&I)[12]) {
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

__global__ void to_root_frame_kernel(const float* __restrict__ G, int64_t g_stride, const float* __restrict__ verts,
                                     const float* __restrict__ joints, const float* __restrict__ T, int V, int J,
                                     float* __restrict__ ginv_out, float* __restrict__ g_root_out,
                                     float* __restrict__ verts_out, float* __restrict__ joints_out, float* __restrict__ T_out) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const float* Gb = G + b * g_stride;
    float I[12];
    affine_inverse12(Gb, I);
    if (v == 0) {
        float* o = ginv_out + b * 16;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = I[k];
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
        float* gr = g_root_out + b * 16;                     // G^-1 . G (the reference keeps the product, not the identity)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gr[r * 4 + c] = I[r * 4 + 0] * Gb[c] + I[r * 4 + 1] * Gb[4 + c] + I[r * 4 + 2] * Gb[8 + c] + (c == 3 ? I[r * 4 + 3] : 0.0f);
        gr[12] = 0.f; gr[13] = 0.f; gr[14] = 0.f; gr[15] = 1.f;
    }
    if (v < J) {
        const float* p = joints + ((int64_t)b * J + v) * 3;
        float* o = joints_out + ((int64_t)b * J + v) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = I[r * 4 + 0] * p[0] + I[r * 4 + 1] * p[1] + I[r * 4 + 2] * p[2] + I[r * 4 + 3];
    }
    if (v >= V) return;
    {
        const float* p = verts + ((int64_t)b * V + v) * 3;
        float* o = verts_out + ((int64_t)b * V + v) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = I[r * 4 + 0] * p[0] + I[r * 4 + 1] * p[1] + I[r * 4 + 2] * p[2] + I[r * 4 + 3];
    }
    const float4* t4 = reinterpret_cast<const float4*>(T + ((int64_t)b * V + v) * 16);
    const float4 r0 = t4[0], r1 = t4[1], r2 = t4[2], r3 = t4[3];
    const float M[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    float4* o4 = reinterpret_cast<float4*>(T_out + ((int64_t)b * V + v) * 16);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)      // full 4x4 product as torch computes it (the bottom row of T is 0 0 0 1 up to rounding)
            q[c] = I[r * 4 + 0] * M[c] + I[r * 4 + 1] * M[4 + c] + I[r * 4 + 2] * M[8 + c] + I[r * 4 + 3] * M[12 + c];
        o4[r] = make_float4(q[0], q[1], q[2], q[3]);
    }
    o4[3] = r3;
}

// ------------------------------------------------------------------ a5 observation -> canonical
// reference: models/anim_nerf.py:147-151.  One thread per vertex; affine closed-form inverse.
__global__ void ober2cano_kernel(const float* __restrict__ tp, const float* __restrict__ tt,
                                 const float* __restrict__ so, const float* __restrict__ sot,
                                 const float* __restrict__ po, const float* __restrict__ pot,
                                 float* __restrict__ out, int64_t n, int64_t n_template) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int64_t vt = v % n_template;                       // one template for all frames, or one per frame
    float A[12], B[12];
    const float4* pa = reinterpret_cast<const float4*>(tp + v * 16);
    const float4* pb = reinterpret_cast<const float4*>(tt + vt * 16);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float4 a = pa[r], b = pb[r];
        A[r * 4 + 0] = a.x; A[r * 4 + 1] = a.y; A[r * 4 + 2] = a.z; A[r * 4 + 3] = a.w;
        B[r * 4 + 0] = b.x; B[r * 4 + 1] = b.y; B[r * 4 + 2] = b.z; B[r * 4 + 3] = b.w;
    }
    // inverse of the 3x3 block by cofactors
    float c00 = A[5] * A[10] - A[6] * A[9];
    float c01 = A[6] * A[8] - A[4] * A[10];
    float c02 = A[4] * A[9] - A[5] * A[8];
    float det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    float id = 1.0f / det;
    float I[12];
    I[0] = c00 * id;
    I[1] = (A[2] * A[9] - A[1] * A[10]) * id;
    I[2] = (A[1] * A[6] - A[2] * A[5]) * id;
    I[4] = c01 * id;
    I[5] = (A[0] * A[10] - A[2] * A[8]) * id;
    I[6] = (A[2] * A[4] - A[0] * A[6]) * id;
    I[8] = c02 * id;
    I[9] = (A[1] * A[8] - A[0] * A[9]) * id;
    I[10] = (A[0] * A[5] - A[1] * A[4]) * id;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float t = -(I[r * 4 + 0] * A[3] + I[r * 4 + 1] * A[7] + I[r * 4 + 2] * A[11]);
        t += sot[vt * 3 + r] - so[v * 3 + r];
        t += pot[vt * 3 + r] - po[v * 3 + r];
        I[r * 4 + 3] = t;
    }
    float4* dst = reinterpret_cast<float4*>(out + v * 16);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float o0 = B[r * 4 + 0] * I[0] + B[r * 4 + 1] * I[4] + B[r * 4 + 2] * I[8];
        float o1 = B[r * 4 + 0] * I[1] + B[r * 4 + 1] * I[5] + B[r * 4 + 2] * I[9];
        float o2 = B[r * 4 + 0] * I[2] + B[r * 4 + 1] * I[6] + B[r * 4 + 2] * I[10];
        float o3 = B[r * 4 + 0] * I[3] + B[r * 4 + 1] * I[7] + B[r * 4 + 2] * I[11] + B[r * 4 + 3];
        dst[r] = make_float4(o0, o1, o2, o3);
    }
    dst[3] = make_float4(0.f, 0.f, 0.f, 1.f);
}

// ------------------------------------------------------------------ a6 coarse depths
// reference: models/volume_rendering.py:29-56
__global__ void sample_coarse_kernel(const float* __restrict__ rays, int stride,
                                     const float* __restrict__ steps, const float* __restrict__ t_rand,
                                     int64_t R, int K, float* __restrict__ z_out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * K) return;
    int64_t r = idx / K;
    int k = (int)(idx % K);
    float near = rays[r * stride + 6], far = rays[r * stride + 7];
    auto zk = [&](int kk) { float s = steps[kk]; return near * (1.0f - s) + far * s; };
    float z = zk(k);
    if (t_rand != nullptr) {
        float zl = (k > 0) ? zk(k - 1) : z, zu = (k + 1 < K) ? zk(k + 1) : z;
        float lower = (k > 0) ? 0.5f * (z + zl) : z;
        float upper = (k + 1 < K) ? 0.5f * (zu + z) : z;
        z = lower + (upper - lower) * t_rand[idx];
    }
    z_out[idx] = z;
}

// ------------------------------------------------------------------ a7 sample points, no warp
// reference: models/volume_rendering.py:117, models/anim_nerf.py:296-297
__global__ void points_from_rays_kernel(const float* __restrict__ rays, int stride,
                                        const float* __restrict__ z, int K, int64_t n,
                                        float4* __restrict__ pts) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* ry = rays + (idx / K) * stride;
    float zz = z[idx];
    pts[idx] = make_float4(ry[0] + zz * ry[3], ry[1] + zz * ry[4], ry[2] + zz * ry[5], 1.0f);
}

// ------------------------------------------------------------------ sigma-grid points (mesh extraction)
// reference: extract_mesh.py:27-35 (create_grid = np.meshgrid(x, y, z), indexing 'xy') and :152-157 (+ centre)
__global__ void grid_points_kernel(int N, double x0, double x1, double y0, double y1, double z0, double z1,
                                   const float* __restrict__ center, int64_t first, int64_t count,
                                   float4* __restrict__ pts) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int64_t n = first + t;
    const int k = (int)(n % N), i = (int)((n / N) % N), j = (int)(n / ((int64_t)N * N));
    // np.linspace: start + arange * step in fp64 (two roundings), last element = stop
    auto lin = [N](double a, double b, int m) {
        if (m == N - 1) return b;
        double step = (b - a) / (double)(N - 1);
        return __dadd_rn(__dmul_rn((double)m, step), a);
    };
    float x = (float)lin(x0, x1, i), y = (float)lin(y0, y1, j), z = (float)lin(z0, z1, k);
    pts[t] = make_float4(x + center[0], y + center[1], z + center[2], 1.0f);
}

// The same points for a LIST of 8 x 8 x 8-voxel cells (N % 8 == 0): cell id c = (cj * (N/8) + ci) * (N/8) + ck covers array
// indices j in [8 cj, 8 cj + 8) and likewise i, k.  One workgroup per listed cell; pts[l * 512 + t] and the voxel's flat grid
// index vox[l * 512 + t].  (Mesh extraction evaluates the field only where a cell can hold a valid voxel: sigma_grid.)
__global__ __launch_bounds__(512) void grid_points_cells_kernel(int N, double x0, double x1, double y0, double y1, double z0, double z1,
                                                                const float* __restrict__ center, const int32_t* __restrict__ cells,
                                                                float4* __restrict__ pts, int32_t* __restrict__ vox) {
    const int C = N / 8, c = cells[blockIdx.x], t = threadIdx.x;
    const int ck = c % C, ci = (c / C) % C, cj = c / (C * C);
    const int k = 8 * ck + (t & 7), i = 8 * ci + ((t >> 3) & 7), j = 8 * cj + (t >> 6);
    auto lin = [N](double a, double b, int m) {
        if (m == N - 1) return b;
        double step = (b - a) / (double)(N - 1);
        return __dadd_rn(__dmul_rn((double)m, step), a);
    };
    const float x = (float)lin(x0, x1, i), y = (float)lin(y0, y1, j), z = (float)lin(z0, z1, k);
    const int64_t o = (int64_t)blockIdx.x * 512 + t;
    pts[o] = make_float4(x + center[0], y + center[1], z + center[2], 1.0f);
    vox[o] = (int32_t)(((int64_t)j * N + i) * N + k);
}

// out[vox[t] - first] = max(values[t], 0) for the voxels of [first, first + count) (a rank's slab)
__global__ __launch_bounds__(256) void scatter_relu_kernel(const float* __restrict__ values, const int32_t* __restrict__ vox, int64_t n,
                                                           int64_t first, int64_t count, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t v = (int64_t)vox[t] - first;
    if (v >= 0 && v < count) out[v] = fmaxf(values[t], 0.0f);
}

}  // namespace anr

using namespace anr;

extern "C" int anr_grid_points_cells(int N, double x0, double x1, double y0, double y1, double z0, double z1, const float* center,
                                     const int32_t* cells, int64_t n_cells, float* pts_out, int32_t* vox_out, void* stream) {
    ANR_REQUIRE(center && cells && pts_out && vox_out, ANR_E_BADARG, "anr_grid_points_cells: null pointer");
    ANR_REQUIRE(N >= 8 && N % 8 == 0 && N <= 1288 && n_cells > 0 && n_cells <= (int64_t)(N / 8) * (N / 8) * (N / 8), ANR_E_BADARG,
                "anr_grid_points_cells: N=%d (a multiple of 8, <= 1288) cells=%lld", N, (long long)n_cells);
    ANR_REQUIRE(((uintptr_t)pts_out & 15) == 0, ANR_E_ALIGN, "anr_grid_points_cells: pts_out must be 16-B aligned");
    hipLaunchKernelGGL(grid_points_cells_kernel, dim3((unsigned)n_cells), dim3(512), 0, (hipStream_t)stream, N, x0, x1, y0, y1, z0, z1,
                       center, cells, reinterpret_cast<float4*>(pts_out), vox_out);
    return check_launch("anr_grid_points_cells");
}

extern "C" int anr_scatter_relu(const float* values, const int32_t* vox, int64_t n, int64_t first, int64_t count, float* out, void* stream) {
    ANR_REQUIRE(values && vox && out, ANR_E_BADARG, "anr_scatter_relu: null pointer");
    ANR_REQUIRE(n > 0 && first >= 0 && count > 0, ANR_E_BADARG, "anr_scatter_relu: n=%lld first=%lld count=%lld", (long long)n, (long long)first,
                (long long)count);
    hipLaunchKernelGGL(scatter_relu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values, vox, n, first, count, out);
    return check_launch("anr_scatter_relu");
}

extern "C" int anr_grid_points(int N, double x0, double x1, double y0, double y1, double z0, double z1,
                               const float* center, int64_t first, int64_t count, float* pts_out, void* stream) {
    ANR_REQUIRE(center && pts_out, ANR_E_BADARG, "anr_grid_points: null pointer");
    ANR_REQUIRE(N >= 2 && first >= 0 && count > 0 && first + count <= (int64_t)N * N * N, ANR_E_BADARG,
                "anr_grid_points: N=%d first=%lld count=%lld", N, (long long)first, (long long)count);
    ANR_REQUIRE(((uintptr_t)pts_out & 15) == 0, ANR_E_ALIGN, "anr_grid_points: pts_out must be 16-B aligned");
    hipLaunchKernelGGL(grid_points_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N,
                       x0, x1, y0, y1, z0, z1, center, first, count, reinterpret_cast<float4*>(pts_out));
    return check_launch("anr_grid_points");
}

extern "C" int anr_version(void) { return ANR_VERSION; }
extern "C" const char* anr_last_error(void) { return err_buf(); }

extern "C" int anr_ray_gen(const float* c2w, const float* focal, const float* center, int H, int W,
                           float near, float far, float* rays_out, void* stream) {
    ANR_REQUIRE(c2w && focal && center && rays_out, ANR_E_BADARG, "anr_ray_gen: null pointer");
    ANR_REQUIRE(H > 0 && W > 0, ANR_E_BADARG, "anr_ray_gen: H=%d W=%d", H, W);
    ANR_REQUIRE(((uintptr_t)rays_out & 15) == 0, ANR_E_ALIGN, "anr_ray_gen: rays_out must be 16-B aligned");
    int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(ray_gen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c2w, focal, center, H, W, near, far, rays_out);
    return check_launch("anr_ray_gen");
}

extern "C" int anr_rays_to_body(const float* g_inv, const float* rays_in, float* rays_out, int bs, int R,
                                int stride_in, void* stream) {
    ANR_REQUIRE(g_inv && rays_in && rays_out, ANR_E_BADARG, "anr_rays_to_body: null pointer");
    ANR_REQUIRE(bs > 0 && R > 0 && stride_in >= 8, ANR_E_BADARG, "anr_rays_to_body: bs=%d R=%d stride=%d", bs, R, stride_in);
    ANR_REQUIRE(((uintptr_t)rays_out & 15) == 0, ANR_E_ALIGN, "anr_rays_to_body: rays_out must be 16-B aligned");
    hipLaunchKernelGGL(rays_to_body_kernel, dim3((R + 255) / 256, bs), dim3(256), 0, (hipStream_t)stream,
                       g_inv, rays_in, rays_out, R, stride_in);
    return check_launch("anr_rays_to_body");
}

extern "C" int anr_to_root_frame(const float* global_transform, const float* verts, const float* joints, const float* T, int bs,
                                 int V, int J, float* g_inv_out, float* g_root_out, float* verts_out, float* joints_out,
                                 float* T_out, void* stream) {
    return anr_to_root_frame_strided(global_transform, 16, verts, joints, T, bs, V, J, g_inv_out, g_root_out, verts_out, joints_out,
                                     T_out, stream);
}

extern "C" int anr_to_root_frame_strided(const float* global_transform, int64_t g_stride, const float* verts, const float* joints,
                                         const float* T, int bs, int V, int J, float* g_inv_out, float* g_root_out, float* verts_out,
                                         float* joints_out, float* T_out, void* stream) {
    ANR_REQUIRE(global_transform && verts && joints && T && g_inv_out && g_root_out && verts_out && joints_out && T_out,
                ANR_E_BADARG, "anr_to_root_frame: null pointer");
    ANR_REQUIRE(bs > 0 && V > 0 && J > 0 && J <= V && g_stride >= 16, ANR_E_BADARG, "anr_to_root_frame: bs=%d V=%d J=%d stride=%lld", bs, V, J,
                (long long)g_stride);
    ANR_REQUIRE((((uintptr_t)T | (uintptr_t)T_out) & 15) == 0, ANR_E_ALIGN, "anr_to_root_frame: T must be 16-B aligned");
    hipLaunchKernelGGL(to_root_frame_kernel, dim3((V + 127) / 128, bs), dim3(128), 0, (hipStream_t)stream, global_transform, g_stride, verts,
                       joints, T, V, J, g_inv_out, g_root_out, verts_out, joints_out, T_out);
    return check_launch("anr_to_root_frame");
}

extern "C" int anr_ober2cano(const float* t_pose, const float* t_template, const float* shape_off,
                             const float* shape_off_t, const float* pose_off, const float* pose_off_t,
                             float* out, int64_t n, int64_t n_template, void* stream) {
    ANR_REQUIRE(t_pose && t_template && shape_off && shape_off_t && pose_off && pose_off_t && out,
                ANR_E_BADARG, "anr_ober2cano: null pointer");
    ANR_REQUIRE(n > 0 && n_template > 0 && n % n_template == 0, ANR_E_BADARG, "anr_ober2cano: n=%lld n_template=%lld", (long long)n,
                (long long)n_template);
    ANR_REQUIRE((((uintptr_t)t_pose | (uintptr_t)t_template | (uintptr_t)out) & 15) == 0, ANR_E_ALIGN,
                "anr_ober2cano: matrices must be 16-B aligned");
    hipLaunchKernelGGL(ober2cano_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       t_pose, t_template, shape_off, shape_off_t, pose_off, pose_off_t, out, n, n_template);
    return check_launch("anr_ober2cano");
}

extern "C" int anr_sample_coarse(const float* rays, int stride, const float* steps, const float* t_rand,
                                 int64_t R, int K, float* z_out, void* stream) {
    ANR_REQUIRE(rays && steps && z_out, ANR_E_BADARG, "anr_sample_coarse: null pointer");
    ANR_REQUIRE(R > 0 && K > 0 && stride >= 8, ANR_E_BADARG, "anr_sample_coarse: R=%lld K=%d stride=%d", (long long)R, K, stride);
    int64_t n = R * K;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, stride, steps, t_rand, R, K, z_out);
    return check_launch("anr_sample_coarse");
}

extern "C" int anr_points_from_rays(const float* rays, int ray_stride, const float* z, int K, int64_t n_points,
                                    float* pts_out, void* stream) {
    ANR_REQUIRE(rays && z && pts_out, ANR_E_BADARG, "anr_points_from_rays: null pointer");
    ANR_REQUIRE(n_points > 0 && K > 0 && ray_stride >= 8, ANR_E_BADARG, "anr_points_from_rays: bad sizes");
    ANR_REQUIRE(((uintptr_t)pts_out & 15) == 0, ANR_E_ALIGN, "anr_points_from_rays: pts_out must be 16-B aligned");
    hipLaunchKernelGGL(points_from_rays_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rays, ray_stride, z, K, n_points, reinterpret_cast<float4*>(pts_out));
    return check_launch("anr_points_from_rays");
}

// ------------------------------------------------------------------ encoding for the weight-gradient GEMMs (a16)
// models/embedding.py:22-39 as a row-major [n][63] matrix (the input of dW_1 and dW_5 in the backward pass), and the
// chain rule back through it.  The forward pass never materialises this (the MLP kernel encodes in registers).
namespace anr {

// COLS = 63 (the reference's matrix) or 64 (one zero column of padding: the layout anr_mlp_wgrad stages)
template <typename T, int COLS = 63>
__global__ __launch_bounds__(256) void encode_kernel(const float* __restrict__ pts, int stride, int64_t n, T* __restrict__ enc,
                                                     int tangent = 0, const int32_t* __restrict__ count = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (count && *count < n) n = *count;                    // row limit on the device (training: the compacted list's padded length)
    if (i >= n) return;
    const float x[3] = {pts[i * stride], pts[i * stride + 1], pts[i * stride + 2]};
    T* row = enc + i * COLS;
    if (COLS > 63) row[63] = (T)0.0f;
    if (tangent && (i & 3)) {                              // row 4p + t, t = 1..3: d enc / d x_{t-1} (ANR_MLP_FLAG_TANGENT)
        const int a = (int)(i & 3) - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) row[d] = (T)(d == a ? 1.0f : 0.0f);
        for (int k = 0; k < 10; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                row[3 + 6 * k + d] = (T)(d == a ? f * cosf(f * x[d]) : 0.0f);
                row[6 + 6 * k + d] = (T)(d == a ? -f * sinf(f * x[d]) : 0.0f);
            }
        }
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) row[d] = (T)x[d];
    for (int k = 0; k < 10; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            row[3 + 6 * k + d] = (T)sinf(f * x[d]);
            row[6 + 6 * k + d] = (T)cosf(f * x[d]);
        }
    }
}

// The 64-column form for anr_mlp_wgrad, one row per thread computed in registers and written as 16-byte pieces: the kernel
// above stores a row element by element — 64 two-byte stores per thread, each wave-wide store instruction touching 64 rows
// 128 B apart — and calls sinf / cosf 60 times per row: 76-105 us at the head of the fine pass's weight-gradient chain of a
// 16-frame step.  bf16: the octaves by the double-angle recurrence from a sincosf at 2^0 and another at 2^5 (the error doubles
// per octave: ~3e-6 absolute after four, three decimal orders under bf16's resolution; from a single seed it reached 1.5e-4
// at 2^9), the wavefront's 64 rows transposed through LDS (row pitch
// 144 B: conflict-free) so that every store instruction writes 1 KB of consecutive bytes.  fp32 (parity mode): sincosf per
// frequency, 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void encode64_kernel(const float* __restrict__ pts, int stride, int64_t n, T* __restrict__ enc,
                                                       int tangent, const int32_t* __restrict__ count) {
    constexpr bool BF = sizeof(T) == 2;
    __shared__ __attribute__((aligned(16))) char sh[BF ? 4 * 64 * 144 : 16];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (count && *count < n) n = *count;                    // row limit on the device (training: the compacted list's padded length)
    const bool live = i < n;
    float v[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) v[c] = 0.0f;
    if (live) {
        const float x[3] = {pts[i * stride], pts[i * stride + 1], pts[i * stride + 2]};
        const int a = (tangent && (i & 3)) ? (int)(i & 3) - 1 : -1;       // row 4p + t, t = 1..3: d enc / d x_{t-1}
        float sn[3], cs[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sincosf(x[d], &sn[d], &cs[d]);
            v[d] = a < 0 ? x[d] : (d == a ? 1.0f : 0.0f);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if ((!BF && k > 0) || k == 5) sincosf(f * x[d], &sn[d], &cs[d]);
                if (a < 0) { v[3 + 6 * k + d] = sn[d]; v[6 + 6 * k + d] = cs[d]; }
                else if (d == a) { v[3 + 6 * k + d] = f * cs[d]; v[6 + 6 * k + d] = -f * sn[d]; }
                if (BF) { const float s2 = 2.0f * sn[d] * cs[d], c2 = 1.0f - 2.0f * sn[d] * sn[d]; sn[d] = s2; cs[d] = c2; }
            }
        }
    }
    if constexpr (BF) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        char* mine = sh + wave * (64 * 144);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 lo = (__bf16)v[8 * q + 2 * e], hi = (__bf16)v[8 * q + 2 * e + 1];
                w[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
            }
            *reinterpret_cast<uint4*>(mine + lane * 144 + q * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        // (the patch is this wavefront's own: the LDS unit executes a wave's instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q * 8 + (lane >> 3), c = lane & 7;
            if (row0 + r < n)
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(enc) + (row0 + r) * 128 + c * 16) =
                    *reinterpret_cast<const uint4*>(mine + r * 144 + c * 16);
        }
    } else {
        if (live) {
            float4* row = reinterpret_cast<float4*>(enc + i * 64);
#pragma unroll
            for (int q = 0; q < 16; ++q) row[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
}

__global__ __launch_bounds__(256) void encode_backward_kernel(const float* __restrict__ pts, int stride, const float* __restrict__ d_enc,
                                                              int64_t n, float4* __restrict__ d_pts, const int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (count && *count < n) n = *count;
    if (i >= n) return;
    const float* g = d_enc + i * 63;
    float dx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float x = pts[i * stride + d];
        float a = g[d];
        for (int k = 0; k < 10; ++k) {
            const float f = (float)(1 << k);
            a += f * (cosf(f * x) * g[3 + 6 * k + d] - sinf(f * x) * g[6 + 6 * k + d]);
        }
        dx[d] = a;
    }
    d_pts[i] = make_float4(dx[0], dx[1], dx[2], 0.0f);
}

}  // namespace anr

extern "C" int anr_encode(const float* pts, int pts_stride, int64_t n, int bf16_out, void* enc_out, void* stream) {
    ANR_REQUIRE(pts && enc_out, ANR_E_BADARG, "anr_encode: null pointer");
    ANR_REQUIRE(n > 0 && pts_stride >= 3, ANR_E_BADARG, "anr_encode: n=%lld stride=%d", (long long)n, pts_stride);
    dim3 grid((unsigned)((n + 255) / 256));
    if (bf16_out)
        hipLaunchKernelGGL(anr::encode_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, pts, pts_stride, n, (__bf16*)enc_out, 0);
    else
        hipLaunchKernelGGL(anr::encode_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, pts, pts_stride, n, (float*)enc_out, 0);
    return anr::check_launch("anr_encode");
}

static int encode64(const float* pts, int pts_stride, int64_t n, const int32_t* count, int flags, void* enc_out, void* stream) {
    ANR_REQUIRE(pts && enc_out, ANR_E_BADARG, "anr_encode64: null pointer");
    ANR_REQUIRE(n > 0 && pts_stride >= 3, ANR_E_BADARG, "anr_encode64: n=%lld stride=%d", (long long)n, pts_stride);
    dim3 grid((unsigned)((n + 255) / 256));
    const int tan = (flags & ANR_MLP_FLAG_TANGENT) ? 1 : 0;
    ANR_REQUIRE(((uintptr_t)enc_out & 15) == 0, ANR_E_ALIGN, "anr_encode64: enc_out must be 16-B aligned");
    if (flags & 1)
        hipLaunchKernelGGL(anr::encode64_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, pts, pts_stride, n, (__bf16*)enc_out, tan, count);
    else
        hipLaunchKernelGGL(anr::encode64_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, pts, pts_stride, n, (float*)enc_out, tan, count);
    return anr::check_launch("anr_encode64");
}
extern "C" int anr_encode64(const float* pts, int pts_stride, int64_t n, int flags, void* enc_out, void* stream) {
    return encode64(pts, pts_stride, n, nullptr, flags, enc_out, stream);
}
extern "C" int anr_encode64_counted(const float* pts, int pts_stride, int64_t n, const int32_t* count, int flags, void* enc_out, void* stream) {
    ANR_REQUIRE(count, ANR_E_BADARG, "anr_encode64_counted: null count");
    return encode64(pts, pts_stride, n, count, flags, enc_out, stream);
}

static int encode_backward(const float* pts, int pts_stride, const float* d_enc, int64_t n, const int32_t* count, float* d_pts_out,
                           void* stream) {
    ANR_REQUIRE(pts && d_enc && d_pts_out, ANR_E_BADARG, "anr_encode_backward: null pointer");
    ANR_REQUIRE(n > 0 && pts_stride >= 3, ANR_E_BADARG, "anr_encode_backward: n=%lld stride=%d", (long long)n, pts_stride);
    ANR_REQUIRE(((uintptr_t)d_pts_out & 15) == 0, ANR_E_ALIGN, "anr_encode_backward: d_pts_out must be 16-B aligned");
    hipLaunchKernelGGL(anr::encode_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
                       pts_stride, d_enc, n, reinterpret_cast<float4*>(d_pts_out), count);
    return anr::check_launch("anr_encode_backward");
}
extern "C" int anr_encode_backward(const float* pts, int pts_stride, const float* d_enc, int64_t n, float* d_pts_out, void* stream) {
    return encode_backward(pts, pts_stride, d_enc, n, nullptr, d_pts_out, stream);
}
extern "C" int anr_encode_backward_counted(const float* pts, int pts_stride, const float* d_enc, int64_t n, const int32_t* count,
                                           float* d_pts_out, void* stream) {
    ANR_REQUIRE(count, ANR_E_BADARG, "anr_encode_backward_counted: null count");
    return encode_backward(pts, pts_stride, d_enc, n, count, d_pts_out, stream);
}
