// Level-set extraction from the thresholded density grid (extract_mesh.py:159-173 of the reference, which calls PyMCubes):
// marching cubes in two passes over the grid, HBM-bound streaming kernels.
//   pass 1 (anr_mc_classify): per grid point, which of its three outgoing grid edges (+axis 0, 1, 2) cross the level
//            (-> that many vertices), and how many triangles the cube based at the point emits (case table);
//   scan   : exclusive prefix sums of both counts (the caller's: torch.cumsum);
//   pass 2 (anr_mc_emit): the vertices (linear interpolation along the edge, index coordinates) and the triangles (vertex id of
//            cube edge e = first vertex of the grid point that owns e + the rank of e's axis among that point's crossing edges).
// The 256-case table is generated on the host (anim_nerf_amd/mesh.py: the same face rule for both cubes sharing a face, so
// the surface has no cracks) and handed in.  inside = value < level.
#include "anr_common.h"

namespace anr {

struct McDims { int n0, n1, n2; };

__device__ __forceinline__ int64_t mc_lin(const McDims& d, int i, int j, int k) { return ((int64_t)i * d.n1 + j) * d.n2 + k; }

__global__ __launch_bounds__(256) void mc_classify_kernel(const float* __restrict__ vol, McDims d, float level,
                                                          const uint8_t* __restrict__ n_tris, uint8_t* __restrict__ vmask,
                                                          int32_t* __restrict__ vcnt, int32_t* __restrict__ tcnt) {
    const int64_t total = (int64_t)d.n0 * d.n1 * d.n2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % d.n2), j = (int)((idx / d.n2) % d.n1), i = (int)(idx / ((int64_t)d.n1 * d.n2));
    const bool hi = i + 1 < d.n0, hj = j + 1 < d.n1, hk = k + 1 < d.n2;
    // corner c = (c & 1, (c >> 1) & 1, (c >> 2) & 1) along axes (0, 1, 2)
    bool in[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int di = c & 1, dj = (c >> 1) & 1, dk = (c >> 2) & 1;
        const bool ok = (!di || hi) && (!dj || hj) && (!dk || hk);
        in[c] = ok ? vol[mc_lin(d, i + di, j + dj, k + dk)] < level : false;
    }
    unsigned m = 0;
    if (hi && in[0] != in[1]) m |= 1u;
    if (hj && in[0] != in[2]) m |= 2u;
    if (hk && in[0] != in[4]) m |= 4u;
    vmask[idx] = (uint8_t)m;
    vcnt[idx] = __popc(m);
    int nt = 0;
    if (hi && hj && hk) {
        unsigned cs = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) cs |= (in[c] ? 1u : 0u) << c;
        nt = n_tris[cs];
    }
    tcnt[idx] = nt;
}

__global__ __launch_bounds__(256) void mc_emit_kernel(const float* __restrict__ vol, McDims d, float level,
                                                      const uint8_t* __restrict__ n_tris, const int8_t* __restrict__ tris,
                                                      const uint8_t* __restrict__ vmask, const int64_t* __restrict__ vstart,
                                                      const int64_t* __restrict__ tstart, float* __restrict__ verts,
                                                      int32_t* __restrict__ faces) {
    const int64_t total = (int64_t)d.n0 * d.n1 * d.n2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % d.n2), j = (int)((idx / d.n2) % d.n1), i = (int)(idx / ((int64_t)d.n1 * d.n2));
    const unsigned m = vmask[idx];
    const float v0 = vol[idx];
    if (m) {
        int64_t o = vstart[idx];
        const int64_t step[3] = {(int64_t)d.n1 * d.n2, d.n2, 1};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!((m >> a) & 1u)) continue;
            const float v1 = vol[idx + step[a]];
            const float t = (level - v0) / (v1 - v0);
            float p[3] = {(float)i, (float)j, (float)k};
            p[a] += t;
            verts[3 * o] = p[0]; verts[3 * o + 1] = p[1]; verts[3 * o + 2] = p[2];
            ++o;
        }
    }
    if (!(i + 1 < d.n0 && j + 1 < d.n1 && k + 1 < d.n2)) return;
    unsigned cs = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        cs |= (vol[mc_lin(d, i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1))] < level ? 1u : 0u) << c;
    const int nt = n_tris[cs];
    if (!nt) return;
    // cube edge e: corners (a, b) differing in one bit, a < b -> owner point = corner a, axis = log2(a ^ b); the host
    // enumerates the edges in the same order (mesh.py: EDGES)
    int e_corner[12], e_axis[12], ne = 0;
    for (int a = 0; a < 8; ++a)
        for (int b = a + 1; b < 8; ++b) {
            const int x = a ^ b;
            if (x == 1 || x == 2 || x == 4) { e_corner[ne] = a; e_axis[ne] = x == 1 ? 0 : x == 2 ? 1 : 2; ++ne; }
        }
    int64_t o = tstart[idx];
    const int8_t* tt = tris + cs * 24;
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = tt[3 * t + q], c = e_corner[e], a = e_axis[e];
            const int64_t owner = mc_lin(d, i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1));
            const unsigned om = vmask[owner];
            faces[3 * o + q] = (int32_t)(vstart[owner] + __popc(om & ((1u << a) - 1u)));
        }
        ++o;
    }
}

}  // namespace anr

using namespace anr;

extern "C" int anr_mc_classify(const float* volume, int n0, int n1, int n2, float level, const uint8_t* n_tris, uint8_t* vmask_out,
                               int32_t* vcount_out, int32_t* tcount_out, void* stream) {
    ANR_REQUIRE(volume && n_tris && vmask_out && vcount_out && tcount_out, ANR_E_BADARG, "anr_mc_classify: null pointer");
    ANR_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, ANR_E_BADARG, "anr_mc_classify: grid %d x %d x %d", n0, n1, n2);
    const int64_t total = (int64_t)n0 * n1 * n2;
    hipLaunchKernelGGL(mc_classify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, volume,
                       McDims{n0, n1, n2}, level, n_tris, vmask_out, vcount_out, tcount_out);
    return check_launch("anr_mc_classify");
}

extern "C" int anr_mc_emit(const float* volume, int n0, int n1, int n2, float level, const uint8_t* n_tris, const int8_t* tris,
                           const uint8_t* vmask, const int64_t* vstart, const int64_t* tstart, float* verts_out, int32_t* faces_out,
                           void* stream) {
    ANR_REQUIRE(volume && n_tris && tris && vmask && vstart && tstart && verts_out && faces_out, ANR_E_BADARG, "anr_mc_emit: null pointer");
    ANR_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, ANR_E_BADARG, "anr_mc_emit: grid %d x %d x %d", n0, n1, n2);
    const int64_t total = (int64_t)n0 * n1 * n2;
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, volume,
                       McDims{n0, n1, n2}, level, n_tris, tris, vmask, vstart, tstart, verts_out, faces_out);
    return check_launch("anr_mc_emit");
}
