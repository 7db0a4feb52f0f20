// Canonical-space warp: exact 4-nearest-vertex search + blend-weight confidence + blended inverse
// skinning transform (a7-a10), and the stand-alone KNN entry point that replaces knn_cuda.KNN.
//
// Layout.  One workgroup (512 threads = 8 wavefronts) serves one body: the posed vertex table
// (V x 3 fp32, 82.7 KB at V = 6890) is staged once into LDS as three SoA planes and every lane
// scans it with wave-uniform (broadcast) ds_read_b128s for PTS points held in registers, so the
// table is read from HBM/L2 once per workgroup and the V x N distance matrix the CUDA reference
// materialises in global memory (55 KB per point) never exists.  The per-vertex tables that are
// only gathered for the four winners (lbs_weights 24 floats, ober2cano 12 floats) stay in L2.
#include "anr_common.h"

namespace anr {

constexpr int WARP_THREADS = 512;
constexpr int PTS = 2;                     // points per lane
constexpr int MAX_J = 32;

struct Best4 {
    float d[4];
    int i[4];
};

__device__ __forceinline__ void best_init(Best4& b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { b.d[k] = 3.0e38f; b.i[k] = 0; }
}

// insert (c, v) keeping d ascending; strict < so the lower vertex id wins ties
__device__ __forceinline__ void best_insert(Best4& b, float c, int v) {
    if (c < b.d[3]) {
        b.d[3] = c; b.i[3] = v;
#pragma unroll
        for (int k = 3; k > 0; --k) {
            if (b.d[k] < b.d[k - 1]) {
                float td = b.d[k]; b.d[k] = b.d[k - 1]; b.d[k - 1] = td;
                int ti = b.i[k]; b.i[k] = b.i[k - 1]; b.i[k - 1] = ti;
            }
        }
    }
}

// Stage verts[V*3] (AoS) into LDS planes x[Vp], y[Vp], z[Vp]; pad with far-away points.
__device__ __forceinline__ void stage_verts(const float* __restrict__ verts, int V, int Vp, float* lds) {
    for (int e = threadIdx.x; e < V * 3; e += blockDim.x) {
        int v = e / 3, c = e - v * 3;
        lds[c * Vp + v] = verts[e];
    }
    for (int v = V + threadIdx.x; v < Vp; v += blockDim.x) {
        lds[v] = 1.0e18f; lds[Vp + v] = 1.0e18f; lds[2 * Vp + v] = 1.0e18f;
    }
    __syncthreads();
}

// brute-force scan of the LDS table for PTS points per lane
__device__ __forceinline__ void scan_table(const float* lds, int Vp, const float (&px)[PTS], const float (&py)[PTS],
                                           const float (&pz)[PTS], Best4 (&best)[PTS]) {
    const float4* X = reinterpret_cast<const float4*>(lds);
    const float4* Y = reinterpret_cast<const float4*>(lds + Vp);
    const float4* Z = reinterpret_cast<const float4*>(lds + 2 * Vp);
    const int n4 = Vp >> 2;
    for (int q = 0; q < n4; ++q) {
        float4 vx = X[q], vy = Y[q], vz = Z[q];
        const float ax[4] = {vx.x, vx.y, vx.z, vx.w};
        const float ay[4] = {vy.x, vy.y, vy.z, vy.w};
        const float az[4] = {vz.x, vz.y, vz.z, vz.w};
#pragma unroll
        for (int p = 0; p < PTS; ++p) {
            float d2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float dx = px[p] - ax[t], dy = py[p] - ay[t], dz = pz[p] - az[t];
                d2[t] = dx * dx + dy * dy + dz * dz;
            }
            float m = fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3]));
            if (m < best[p].d[3]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) best_insert(best[p], d2[t], q * 4 + t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reference: models/anim_nerf.py:153-192 (get_neighbs + unpose), volume_rendering.py:117
__global__ __launch_bounds__(WARP_THREADS) void warp_points_kernel(
    const float* __restrict__ xyz, int xyz_stride, const float* __restrict__ rays, int ray_stride,
    const float* __restrict__ z, int K, const float* __restrict__ verts, const float* __restrict__ ober2cano,
    const float* __restrict__ lbs_w, int V, int Vp, int J, int64_t N, float thr, float4* __restrict__ pts_out,
    float* __restrict__ dist_out, int32_t* __restrict__ idx_out, float* __restrict__ blended_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    stage_verts(verts + (int64_t)b * V * 3, V, Vp, lds);

    const int64_t base = (int64_t)blockIdx.x * (WARP_THREADS * PTS);
    float px[PTS], py[PTS], pz[PTS];
    int64_t n[PTS];
    Best4 best[PTS];
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        n[p] = base + p * WARP_THREADS + threadIdx.x;
        int64_t nn = n[p] < N ? n[p] : N - 1;
        if (xyz != nullptr) {
            const float* s = xyz + ((int64_t)b * N + nn) * xyz_stride;
            px[p] = s[0]; py[p] = s[1]; pz[p] = s[2];
        } else {
            const float* ry = rays + ((int64_t)b * (N / K) + nn / K) * ray_stride;
            float zz = z[(int64_t)b * N + nn];
            px[p] = __fadd_rn(ry[0], __fmul_rn(zz, ry[3]));
            py[p] = __fadd_rn(ry[1], __fmul_rn(zz, ry[4]));
            pz[p] = __fadd_rn(ry[2], __fmul_rn(zz, ry[5]));
        }
        best_init(best[p]);
    }
    scan_table(lds, Vp, px, py, pz, best);

    const float* O2C = ober2cano + (int64_t)b * V * 16;
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        if (n[p] >= N) continue;
        float dist[4], conf[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dist[k] = sqrtf(best[p].d[k]);
        // blend-weight confidence against neighbour 0 (anim_nerf.py:165-168)
        const float* w0 = lbs_w + (int64_t)best[p].i[0] * J;
        conf[0] = 1.0f;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float* wk = lbs_w + (int64_t)best[p].i[k] * J;
            float s = 0.f;
            for (int j = 0; j < J; ++j) s += fabsf(wk[j] - w0[j]);
            conf[k] = (expf(-s / 0.02f) > 0.9f) ? 1.0f : 0.0f;
        }
        float wsum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { w[k] = expf(-dist[k]) * conf[k]; wsum += w[k]; }
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        float db = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[k] = w[k] / wsum;
            const float4* M = reinterpret_cast<const float4*>(O2C + (int64_t)best[p].i[k] * 16);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float4 m = M[r];
                T[r * 4 + 0] += w[k] * m.x; T[r * 4 + 1] += w[k] * m.y;
                T[r * 4 + 2] += w[k] * m.z; T[r * 4 + 3] += w[k] * m.w;
            }
            db += w[k] * dist[k];
        }
        float cx = T[0] * px[p] + T[1] * py[p] + T[2] * pz[p] + T[3];
        float cy = T[4] * px[p] + T[5] * py[p] + T[6] * pz[p] + T[7];
        float cz = T[8] * px[p] + T[9] * py[p] + T[10] * pz[p] + T[11];
        const int64_t o = (int64_t)b * N + n[p];
        pts_out[o] = make_float4(cx, cy, cz, db < thr ? 1.0f : 0.0f);
        if (dist_out != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { dist_out[o * 4 + k] = dist[k]; idx_out[o * 4 + k] = best[p].i[k]; }
            blended_out[o] = db;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reference: models/anim_nerf.py:157-163 (KNN_CUDA call / in-repo fallback definition)
__global__ __launch_bounds__(WARP_THREADS) void knn_kernel(const float* __restrict__ verts,
                                                           const float* __restrict__ xyz, int V, int Vp, int64_t N,
                                                           float* __restrict__ dist_out,
                                                           int64_t* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    stage_verts(verts + (int64_t)b * V * 3, V, Vp, lds);
    const int64_t base = (int64_t)blockIdx.x * (WARP_THREADS * PTS);
    float px[PTS], py[PTS], pz[PTS];
    int64_t n[PTS];
    Best4 best[PTS];
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        n[p] = base + p * WARP_THREADS + threadIdx.x;
        int64_t nn = n[p] < N ? n[p] : N - 1;
        const float* s = xyz + ((int64_t)b * N + nn) * 3;
        px[p] = s[0]; py[p] = s[1]; pz[p] = s[2];
        best_init(best[p]);
    }
    scan_table(lds, Vp, px, py, pz, best);
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        if (n[p] >= N) continue;
        const int64_t o = ((int64_t)b * N + n[p]) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dist_out[o + k] = sqrtf(best[p].d[k]); idx_out[o + k] = best[p].i[k]; }
    }
}

inline int lds_bytes_for(int V, int* Vp_out) {
    int Vp = (V + 3) & ~3;
    *Vp_out = Vp;
    return Vp * 3 * (int)sizeof(float);
}

template <typename Kern>
int allow_big_lds(Kern k, int bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%d B LDS): %s", who, bytes, hipGetErrorString(e));
    return 0;
}

}  // namespace anr

using namespace anr;

extern "C" int anr_warp_points(const float* xyz, int xyz_stride, const float* rays, int ray_stride, const float* z,
                               int K, const float* verts, const float* ober2cano, const float* lbs_weights, int bs,
                               int V, int J, int64_t N, float dis_threshold, float* pts_out, float* dist_out,
                               int32_t* idx_out, float* blended_out, void* stream) {
    ANR_REQUIRE(verts && ober2cano && lbs_weights && pts_out, ANR_E_BADARG, "anr_warp_points: null pointer");
    ANR_REQUIRE((xyz != nullptr) || (rays != nullptr && z != nullptr), ANR_E_BADARG,
                "anr_warp_points: need xyz or (rays, z)");
    ANR_REQUIRE(bs > 0 && V >= 4 && N > 0 && J > 0 && J <= MAX_J, ANR_E_BADARG,
                "anr_warp_points: bs=%d V=%d N=%lld J=%d", bs, V, (long long)N, J);
    ANR_REQUIRE(xyz != nullptr ? xyz_stride >= 3 : (K > 0 && ray_stride >= 8 && N % K == 0), ANR_E_BADARG,
                "anr_warp_points: bad stride/K");
    ANR_REQUIRE((dist_out == nullptr) == (idx_out == nullptr) && (dist_out == nullptr) == (blended_out == nullptr),
                ANR_E_BADARG, "anr_warp_points: debug outputs are all-or-none");
    ANR_REQUIRE((((uintptr_t)pts_out | (uintptr_t)ober2cano) & 15) == 0, ANR_E_ALIGN,
                "anr_warp_points: pts_out / ober2cano must be 16-B aligned");
    int Vp, bytes = lds_bytes_for(V, &Vp);
    ANR_REQUIRE(bytes <= 160 * 1024, ANR_E_SHAPE, "anr_warp_points: V=%d needs %d B of LDS (>160 KiB)", V, bytes);
    if (int rc = allow_big_lds(warp_points_kernel, bytes, "anr_warp_points")) return rc;
    dim3 grid((unsigned)((N + WARP_THREADS * PTS - 1) / (WARP_THREADS * PTS)), bs);
    hipLaunchKernelGGL(warp_points_kernel, grid, dim3(WARP_THREADS), bytes, (hipStream_t)stream, xyz, xyz_stride, rays,
                       ray_stride, z, K, verts, ober2cano, lbs_weights, V, Vp, J, N, dis_threshold,
                       reinterpret_cast<float4*>(pts_out), dist_out, idx_out, blended_out);
    return check_launch("anr_warp_points");
}

extern "C" int anr_knn(const float* verts, const float* xyz, int bs, int V, int64_t N, float* dist_out,
                       int64_t* idx_out, void* stream) {
    ANR_REQUIRE(verts && xyz && dist_out && idx_out, ANR_E_BADARG, "anr_knn: null pointer");
    ANR_REQUIRE(bs > 0 && V >= 4 && N > 0, ANR_E_BADARG, "anr_knn: bs=%d V=%d N=%lld", bs, V, (long long)N);
    int Vp, bytes = lds_bytes_for(V, &Vp);
    ANR_REQUIRE(bytes <= 160 * 1024, ANR_E_SHAPE, "anr_knn: V=%d needs %d B of LDS (>160 KiB)", V, bytes);
    if (int rc = allow_big_lds(knn_kernel, bytes, "anr_knn")) return rc;
    dim3 grid((unsigned)((N + WARP_THREADS * PTS - 1) / (WARP_THREADS * PTS)), bs);
    hipLaunchKernelGGL(knn_kernel, grid, dim3(WARP_THREADS), bytes, (hipStream_t)stream, verts, xyz, V, Vp, N, dist_out,
                       idx_out);
    return check_launch("anr_knn");
}
