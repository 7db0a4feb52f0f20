// Exact k nearest SMPL vertices for k != 4 (models/anim_nerf.py:42, 153-163: `k_neigh` is a constructor argument; every
// shipped config leaves it at 4, which the pruned search of warp.hip serves).  Plain exhaustive scan: the body's vertices
// sit in LDS (V <= 13,000: 156 KB of the CU's 160), one thread per query point keeps its k best (distance, vertex id)
// pairs in registers — ascending lexicographically, so ties resolve to the lower id whatever the scan order.  VALU-bound,
// V distance evaluations per point: a served option, not a tuned one.
#include "anr_common.h"

namespace anr {

constexpr int KNNK_THREADS = 256;

template <int K>
__global__ __launch_bounds__(KNNK_THREADS) void knn_k_kernel(const float* __restrict__ verts, const float* __restrict__ xyz, int xyz_stride,
                                                             int V, int64_t N, float* __restrict__ dist_out, int64_t* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float sv[];
    const int b = blockIdx.y;
    const float* vb = verts + (int64_t)b * V * 3;
    for (int i = threadIdx.x; i < V * 3; i += KNNK_THREADS) sv[i] = vb[i];
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * KNNK_THREADS + threadIdx.x; p < N; p += (int64_t)gridDim.x * KNNK_THREADS) {
        const float* q = xyz + ((int64_t)b * N + p) * xyz_stride;
        const float px = q[0], py = q[1], pz = q[2];
        float d[K];
        int id[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { d[k] = 3.0e38f; id[k] = 0x7fffffff; }
        for (int v = 0; v < V; ++v) {
            const float dx = px - sv[3 * v], dy = py - sv[3 * v + 1], dz = pz - sv[3 * v + 2];
            float c = dx * dx + dy * dy + dz * dz;
            if (c < d[K - 1]) {                              // strictly better than the current k-th (ids ascend: ties keep the lower)
                int ci = v;
#pragma unroll
                for (int k = 0; k < K; ++k) {                // one bubble pass of the new pair through the sorted list
                    const bool lt = c < d[k] || (c == d[k] && ci < id[k]);
                    const float td = lt ? d[k] : c;
                    const int ti = lt ? id[k] : ci;
                    d[k] = lt ? c : d[k];
                    id[k] = lt ? ci : id[k];
                    c = td; ci = ti;
                }
            }
        }
        const int64_t o = ((int64_t)b * N + p) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) { dist_out[o + k] = sqrtf(d[k]); idx_out[o + k] = id[k]; }
    }
}

template <int K>
int launch_knn_k(const float* verts, const float* xyz, int xyz_stride, int bs, int V, int64_t N, float* dist, int64_t* idx, hipStream_t st) {
    const int lds = V * 3 * (int)sizeof(float);
    auto kern = knn_k_kernel<K>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail((int)e, "anr_knn_k: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int64_t blocks = (N + KNNK_THREADS - 1) / KNNK_THREADS;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, bs), dim3(KNNK_THREADS), lds, st, verts, xyz, xyz_stride, V, N, dist, idx);
    return check_launch("anr_knn_k");
}

}  // namespace anr

using namespace anr;

extern "C" int anr_knn_k(const float* verts, const float* xyz, int xyz_stride, int bs, int V, int64_t N, int k, float* dist_out,
                         int64_t* idx_out, void* stream) {
    ANR_REQUIRE(verts && xyz && dist_out && idx_out, ANR_E_BADARG, "anr_knn_k: null pointer");
    ANR_REQUIRE(bs > 0 && V >= k && N > 0 && xyz_stride >= 3, ANR_E_BADARG, "anr_knn_k: bs=%d V=%d N=%lld stride=%d", bs, V, (long long)N, xyz_stride);
    ANR_REQUIRE(k >= 1 && k <= 8, ANR_E_SHAPE, "anr_knn_k: k=%d (1..8)", k);
    ANR_REQUIRE(V <= 13000, ANR_E_SHAPE, "anr_knn_k: V=%d vertices do not fit the LDS (13000)", V);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: return launch_knn_k<1>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 2: return launch_knn_k<2>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 3: return launch_knn_k<3>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 4: return launch_knn_k<4>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 5: return launch_knn_k<5>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 6: return launch_knn_k<6>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        case 7: return launch_knn_k<7>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
        default: return launch_knn_k<8>(verts, xyz, xyz_stride, bs, V, N, dist_out, idx_out, st);
    }
}
