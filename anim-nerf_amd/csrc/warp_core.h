// (header) The exact 4-nearest-vertex search and the blend of csrc/warp.hip as device routines — index layout (IndexDims), the reach
// mask's cell arithmetic, Best4 / best_insert, the pruned tree walks (search_from, search_near), blend_and_store — shared by
// csrc/warp.hip (the stand-alone warp / KNN kernels) and csrc/ray_march.hip (the one-pass kernel with the warp, round 6): ONE
// definition, so that both return the same neighbours and the same canonical points bit for bit.  `lds` in the signatures is
// "where the index lives": LDS in the warp kernels, global memory (L2) in the one-pass kernel, whose LDS holds the weight ring.
// Reference: models/anim_nerf.py:153-192; knn_cuda.KNN at :82-83, :159 (fallback definition :161-163).
#pragma once
#include "anr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace anr {

constexpr int WARP_THREADS = 1024;
constexpr int CS = 8;                      // vertices per cluster
constexpr int SC = 8;                      // clusters per super-cluster
constexpr int TC = 8;                      // super-clusters per top
constexpr int MAX_NC = 2048;               // clusters the build kernel can hold (V <= 16384)
constexpr int MAX_J = 32;
constexpr float FAR = 1.0e18f;

struct IndexDims {
    int V, NC, NS, NT, Vp;
    __host__ __device__ int box_off() const { return 3 * Vp; }
    __host__ __device__ int sbox_off() const { return 3 * Vp + 8 * NC; }
    __host__ __device__ int tbox_off() const { return 3 * Vp + 8 * NC + 8 * NS; }
    __host__ __device__ int body_off() const { return 3 * Vp + 8 * NC + 8 * NS + 8 * NT; }
    __host__ __device__ int order_off() const { return body_off() + 8; }
    __host__ __device__ int lds_floats() const { return order_off(); }
    __host__ __device__ int reach_off() const { return order_off() + Vp; }           // the reach mask (RG^3 bits), see below
    __host__ __device__ int total_floats() const { return reach_off() + 1024; }
};
inline IndexDims index_dims(int V) {
    IndexDims d;
    d.V = V;
    d.NC = (V + CS - 1) / CS;
    d.NS = (d.NC + SC - 1) / SC;
    d.NT = (d.NS + TC - 1) / TC;
    d.Vp = d.NC * CS;
    return d;
}

// ---------------------------------------------------------------------------------------------
// The reach mask: a 32^3 grid over the body's box padded by the validity radius `thr` the index was built for (stored in the
// body box's 4th float; 0 = no mask); bit (ix, iy, iz) is set iff some vertex lies within thr of the CELL (box distance).  A
// sample can only be valid if a vertex lies within dis_threshold of it (its blended distance is a convex combination of its
// four neighbours' distances, models/anim_nerf.py:169-183) — i.e. only in a cell whose bit is set, for any dis_threshold <=
// thr: the classify pass drops the others without a search.  60 % of a training batch's samples fall into the padded box, 6.5 %
// are valid (profiles/r04/warp_small_batch_by_bodies.txt): the searches that only prove a sample invalid were most of the
// per-sample search's work.
constexpr int RG = 32;
__device__ __forceinline__ float reach_cell_size(const float* gbox, float thr) {
    const float ex = gbox[4] - gbox[0], ey = gbox[5] - gbox[1], ez = gbox[6] - gbox[2];
    return (fmaxf(fmaxf(ex, ey), ez) + 2.0f * thr) * (1.0f / (float)RG);
}
// (ix, iy, iz) of a point, or -1 outside the padded cube the grid covers (such a point is farther than thr from the box)
__device__ __forceinline__ int reach_cell(const float* gbox, float thr, float inv, float px, float py, float pz) {
    const float fx = (px - gbox[0] + thr) * inv, fy = (py - gbox[1] + thr) * inv, fz = (pz - gbox[2] + thr) * inv;
    if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)RG && fy < (float)RG && fz < (float)RG)) return -1;
    return ((int)fx * RG + (int)fy) * RG + (int)fz;
}


// ---------------------------------------------------------------------------------------------
struct Best4 {
    float d[4];
    int i[4];
};
// cap2 = squared search radius: only vertices strictly closer than sqrt(cap2) are collected; slots stay -1 otherwise
__device__ __forceinline__ void best_init(Best4& b, float cap2 = 3.0e38f) {
    // (an empty slot holds the float just below cap2 and slot -1 = 0xffffffff: under best_insert's unsigned 64-bit order
    // (distance bits : slot) a candidate enters exactly when its distance is < cap2)
    const float below = __uint_as_float(__float_as_uint(cap2) - 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k) { b.d[k] = below; b.i[k] = -1; }
}
// insert (c, v) keeping (d, slot) ascending LEXICOGRAPHICALLY: among vertices at exactly the same distance the lower
// index slot wins, whatever order the traversal visits them in — the exact search, the cell-sorted search and the
// brute-force check then agree bit for bit also on ties (about one sample in 10^7 on a 1024^2 frame, enough to break
// a bit-identity test).  Branch-free: new d[k] = median(d[k-1], d[k], c) for a sorted list, the ids follow the four
// comparisons.
__device__ __forceinline__ void best_insert(Best4& b, float c, int v) {
    // the four tests as ONE unsigned 64-bit compare each — squared distances are non-negative floats, whose bit patterns order
    // like the numbers — instead of three compares and two mask operations ((c < d) | ((c == d) & (v < i)); written with || and
    // && hipcc even built each test out of three nested branches: ~60 instructions and eight s_cbranch per insertion)
    const unsigned long long key = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned)v;
    auto at = [&](int k) { return ((unsigned long long)__float_as_uint(b.d[k]) << 32) | (unsigned)b.i[k]; };
    const bool m0 = key < at(0), m1 = key < at(1), m2 = key < at(2), m3 = key < at(3);
    b.i[3] = m3 ? (m2 ? b.i[2] : v) : b.i[3];
    b.i[2] = m2 ? (m1 ? b.i[1] : v) : b.i[2];
    b.i[1] = m1 ? (m0 ? b.i[0] : v) : b.i[1];
    b.i[0] = m0 ? v : b.i[0];
    const float d0 = b.d[0], d1 = b.d[1], d2 = b.d[2], d3 = b.d[3];
    b.d[0] = __builtin_amdgcn_fmed3f(d0, c, -1.0f);                  // = min(d0, c) for non-negative operands, one instruction
    b.d[1] = __builtin_amdgcn_fmed3f(d0, d1, c);
    b.d[2] = __builtin_amdgcn_fmed3f(d1, d2, c);
    b.d[3] = __builtin_amdgcn_fmed3f(d2, d3, c);
}

// squared distance from p to an axis-aligned box (0 inside)
__device__ __forceinline__ float box_d2(const float* bx, float px, float py, float pz) {
    const float4 lo = *reinterpret_cast<const float4*>(bx);
    const float4 hi = *reinterpret_cast<const float4*>(bx + 4);
    float dx = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.f);
    float dy = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.f);
    float dz = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

// squared distance, ONE sequence of roundings wherever it is computed (the searches must agree bit for bit):
// fma(dz, dz, fma(dx, dx, dy * dy)) — what hipcc's contraction made of dx*dx + dy*dy + dz*dz in scan_cluster, spelled out
__device__ __forceinline__ float dist2(float px, float py, float pz, float vx, float vy, float vz) {
    const float dx = __fsub_rn(px, vx), dy = __fsub_rn(py, vy), dz = __fsub_rn(pz, vz);
    return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}

__device__ __forceinline__ void scan_cluster(const float* lds, int Vp, int c, float px, float py, float pz,
                                             Best4& best) {
    const float4* X = reinterpret_cast<const float4*>(lds + c * CS);
    const float4* Y = reinterpret_cast<const float4*>(lds + Vp + c * CS);
    const float4* Z = reinterpret_cast<const float4*>(lds + 2 * Vp + c * CS);
    // the cluster's 96 bytes in six reads issued together (one wait, not one per half), the eight distances two at a time
    // (v_pk_add/mul/fma_f32: the same operations in the same order as the scalar form)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 P[3] = {f2{px, px}, f2{py, py}, f2{pz, pz}};
    float4 vx[CS / 4], vy[CS / 4], vz[CS / 4];
#pragma unroll
    for (int q = 0; q < CS / 4; ++q) { vx[q] = X[q]; vy[q] = Y[q]; vz[q] = Z[q]; }
    float d2[CS];
#pragma unroll
    for (int q = 0; q < CS / 4; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f2 ax = t ? f2{vx[q].z, vx[q].w} : f2{vx[q].x, vx[q].y}, ay = t ? f2{vy[q].z, vy[q].w} : f2{vy[q].x, vy[q].y},
                     az = t ? f2{vz[q].z, vz[q].w} : f2{vz[q].x, vz[q].y};
            const f2 dx = P[0] - ax, dy = P[1] - ay, dz = P[2] - az;
            const f2 r = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));      // == dist2()
            d2[4 * q + 2 * t] = r.x; d2[4 * q + 2 * t + 1] = r.y;
        }
    }
    float m = d2[0];
#pragma unroll
    for (int t = 1; t < CS; ++t) m = fminf(m, d2[t]);
    if (m <= best.d[3]) {                                  // (<=: a tie with the current 4th may carry a lower slot)
        // ... and inside, vertex by vertex: a wavefront comes here when ANY lane has a candidate among the eight, mostly one
        // or two of them — an insertion is ~40 instructions, the test that skips it four
#pragma unroll
        for (int t = 0; t < CS; ++t)
            if (d2[t] <= best.d[3]) best_insert(best, d2[t], c * CS + t);
    }
}

// greedy descent by box distance (per-lane LDS addresses below the top level) -> the cluster to scan first
__device__ __forceinline__ int seed_cluster(const float* lds, const IndexDims& d, float px, float py, float pz) {
    const float* boxes = lds + d.box_off();
    const float* sboxes = lds + d.sbox_off();
    const float* tboxes = lds + d.tbox_off();
    int seed_t = 0;
    float seed_v = 3.0e38f;
    for (int t = 0; t < d.NT; ++t) {
        const float v = box_d2(tboxes + t * 8, px, py, pz);
        if (v < seed_v) { seed_v = v; seed_t = t; }
    }
    int seed_s = seed_t * TC;
    seed_v = 3.0e38f;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int q = min(seed_t * TC + j, d.NS - 1);
        const float v = box_d2(sboxes + q * 8, px, py, pz);
        if (v < seed_v) { seed_v = v; seed_s = q; }
    }
    int seed_c = seed_s * SC;
    seed_v = 3.0e38f;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
        const int c = min(seed_s * SC + j, d.NC - 1);
        const float v = box_d2(boxes + c * 8, px, py, pz);
        if (v < seed_v) { seed_v = v; seed_c = c; }
    }
    return seed_c;
}

#ifdef ANR_SEARCH_PROF
// experiment builds only (tools/exp/walk_prof.py): what a wavefront of the lane-per-sample search executes, summed over the
// launches since the last read: {calls, top / super / cluster box tests, cluster scans, lanes that needed a scan, seed passes,
// active lanes}
__device__ unsigned long long anr_walk_prof[8];
#define WALK_ADD(k, v) walk[k] += (v)
#else
#define WALK_ADD(k, v)
#endif
// exact 4 nearest vertices of (px,py,pz); best.i are index SLOTS (map through order[] for vertex ids).
// Control flow is wave-uniform (ballots); per-lane work is predicated.  seed_c: the cluster each lane scans first
// (any cluster is correct; a near one makes the bound tight before the traversal starts).
__device__ __forceinline__ void search_from(const float* lds, const IndexDims& d, float px, float py, float pz, bool active,
                                            Best4& best, int seed_c) {
    const float* boxes = lds + d.box_off();
    const float* sboxes = lds + d.sbox_off();
    const float* tboxes = lds + d.tbox_off();
    if (!active) seed_c = -1;
#ifdef ANR_SEARCH_PROF
    unsigned walk[8] = {1, 0, 0, 0, 0, 0, 0, (unsigned)__popcll(__ballot(active))};
#endif
    unsigned long long rem = __ballot(active);
    while (rem) {                                   // one pass per distinct seed cluster in the wavefront
        const int first = __builtin_ctzll(rem);
        const int c = __builtin_amdgcn_readlane(seed_c, first);
        const bool mine = (seed_c == c);
        if (mine) scan_cluster(lds, d.Vp, c, px, py, pz, best);
        rem &= ~__ballot(mine);
        WALK_ADD(6, 1);
    }
    // every other cluster whose box can still beat the current 4th-best
    for (int t = 0; t < d.NT; ++t) {
        const float tv = box_d2(tboxes + t * 8, px, py, pz);
        WALK_ADD(1, 1);
        if (!__any(active && tv <= best.d[3])) continue;
        const int s_end = min((t + 1) * TC, d.NS);
        for (int q = t * TC; q < s_end; ++q) {
            const float sv = box_d2(sboxes + q * 8, px, py, pz);
            WALK_ADD(2, 1);
            if (!__any(active && sv <= best.d[3])) continue;
            const int c_end = min((q + 1) * SC, d.NC);
            for (int c = q * SC; c < c_end; ++c) {
                const float v = box_d2(boxes + c * 8, px, py, pz);
                const bool need = active && (c != seed_c) && (v <= best.d[3]);
                WALK_ADD(3, 1);
                if (__any(need)) {
                    WALK_ADD(4, 1);
                    WALK_ADD(5, (unsigned)__popcll(__ballot(need)));
                    if (need) scan_cluster(lds, d.Vp, c, px, py, pz, best);
                }
            }
        }
    }
#ifdef ANR_SEARCH_PROF
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&anr_walk_prof[k], (unsigned long long)walk[k]);
#endif
}
// wave-wide max / min in every lane: four DPP steps inside the rows of 16, then the four row results through SGPRs
template <bool MAX> __device__ __forceinline__ float wave_reduce(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));      // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));      // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));     // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));     // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return op(op(r0, r1), op(r2, r3));
}

// The same search for a wavefront whose points are NEIGHBOURS IN SPACE (the cell-sorted list: 64 consecutive entries lie in one
// or two 4-cm cells).  The walk above tests 14 top + ~42 super + ~98 cluster boxes per wavefront to find the ~23 clusters it
// scans (tools/exp/walk_prof.py on a cfg3 frame) — every lane the same boxes: two thirds of the kernel's instructions.  Here the
// LANES SPLIT THE BOXES once per wavefront: after the seed cluster, every point's four neighbours lie within
// sqrt(max over the lanes of d4) of it, hence within R = that + the half diagonal of the points' bounding box of the box's
// centre; lane j tests cluster box j, j + 64, ... against that sphere (14 tests for 862 clusters) and the ballots are the
// candidate set, a superset of what any lane needs.  The lanes then test only the candidates against their own bound.
// Any order of visits gives the same four neighbours ((distance, slot) is a total order: best_insert).
__device__ __forceinline__ void search_near(const float* lds, const IndexDims& d, float px, float py, float pz, bool active,
                                            Best4& best, int seed_c) {
    const float* boxes = lds + d.box_off();
    if (!active) seed_c = -1;
#ifdef ANR_SEARCH_PROF
    unsigned walk[8] = {1, 0, 0, 0, 0, 0, 0, (unsigned)__popcll(__ballot(active))};
#endif
    unsigned long long rem = __ballot(active);
    while (rem) {                                   // one pass per distinct seed cluster in the wavefront
        const int first = __builtin_ctzll(rem);
        const int c = __builtin_amdgcn_readlane(seed_c, first);
        const bool mine = (seed_c == c);
        if (mine) scan_cluster(lds, d.Vp, c, px, py, pz, best);
        rem &= ~__ballot(mine);
        WALK_ADD(6, 1);
    }
    constexpr float BIG = 3.0e38f;
    const float lox = wave_reduce<false>(active ? px : BIG), hix = wave_reduce<true>(active ? px : -BIG);
    const float loy = wave_reduce<false>(active ? py : BIG), hiy = wave_reduce<true>(active ? py : -BIG);
    const float loz = wave_reduce<false>(active ? pz : BIG), hiz = wave_reduce<true>(active ? pz : -BIG);
    const float cx = 0.5f * (lox + hix), cy = 0.5f * (loy + hiy), cz = 0.5f * (loz + hiz);
    const float ex = hix - cx, ey = hiy - cy, ez = hiz - cz;
    const float reach = (sqrtf(wave_reduce<true>(active ? best.d[3] : 0.f)) + sqrtf(ex * ex + ey * ey + ez * ez)) * 1.001f + 1.0e-6f;
    const float reach2 = reach * reach;
    const int lane = threadIdx.x & 63;
    // (measured and dropped: two rounds, the candidates within half the reach first and the reach taken again for the rest —
    // 20.8 scans and 48 + 28 box tests per item instead of 22.7 and 52 + 14, but 5 % slower)
    for (int c0 = 0; c0 < d.NC; c0 += 64) {
        const int cj = c0 + lane;
        const float cv = cj < d.NC ? box_d2(boxes + cj * 8, cx, cy, cz) : BIG;
        unsigned long long cand = __ballot(cv <= reach2);
        WALK_ADD(1, 1);
        while (cand) {
            const int c = c0 + __builtin_ctzll(cand);
            cand &= cand - 1;
            const float v = box_d2(boxes + c * 8, px, py, pz);
            const bool need = active && (c != seed_c) && (v <= best.d[3]);
            WALK_ADD(3, 1);
            if (__any(need)) {
                WALK_ADD(4, 1);
                WALK_ADD(5, (unsigned)__popcll(__ballot(need)));
                if (need) scan_cluster(lds, d.Vp, c, px, py, pz, best);
            }
        }
    }
#ifdef ANR_SEARCH_PROF
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&anr_walk_prof[k], (unsigned long long)walk[k]);
#endif
}
__device__ __forceinline__ void search(const float* lds, const IndexDims& d, float px, float py, float pz, bool active,
                                       Best4& best) {
    search_from(lds, d, px, py, pz, active, best, seed_cluster(lds, d, px, py, pz));
}


// Blend the four neighbours (models/anim_nerf.py:165-192) and store the canonical point; best.i are index slots.
__device__ __forceinline__ bool blend_and_store(const Best4& best, const int32_t* __restrict__ order,
                                                const float* __restrict__ lbs_w, int J, const float* __restrict__ O2C,
                                                float thr, float px, float py, float pz, int64_t o,
                                                float4* __restrict__ pts_out, float* __restrict__ dist_out,
                                                int32_t* __restrict__ idx_out, float* __restrict__ blended_out,
                                                int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_w) {
    float dist[4], conf[4], w[4];
    int vid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { dist[k] = sqrtf(best.d[k]); vid[k] = order[best.i[k]]; }
    // blend-weight confidence against neighbour 0 (anim_nerf.py:165-168); rows are read as float4 (J % 4 == 0,
    // 16-B aligned rows): one request per 16 B instead of per float on this uncoalesced gather
    conf[0] = 1.0f;
    if ((J & 3) == 0) {
        const float4* r0 = reinterpret_cast<const float4*>(lbs_w + (int64_t)vid[0] * J);
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float4* r1 = reinterpret_cast<const float4*>(lbs_w + (int64_t)vid[1] * J);
        const float4* r2 = reinterpret_cast<const float4*>(lbs_w + (int64_t)vid[2] * J);
        const float4* r3 = reinterpret_cast<const float4*>(lbs_w + (int64_t)vid[3] * J);
        for (int q = 0; q < (J >> 2); ++q) {
            const float4 a = r0[q], b1 = r1[q], b2 = r2[q], b3 = r3[q];
            s1 += fabsf(b1.x - a.x) + fabsf(b1.y - a.y) + fabsf(b1.z - a.z) + fabsf(b1.w - a.w);
            s2 += fabsf(b2.x - a.x) + fabsf(b2.y - a.y) + fabsf(b2.z - a.z) + fabsf(b2.w - a.w);
            s3 += fabsf(b3.x - a.x) + fabsf(b3.y - a.y) + fabsf(b3.z - a.z) + fabsf(b3.w - a.w);
        }
        conf[1] = (expf(-s1 / 0.02f) > 0.9f) ? 1.0f : 0.0f;
        conf[2] = (expf(-s2 / 0.02f) > 0.9f) ? 1.0f : 0.0f;
        conf[3] = (expf(-s3 / 0.02f) > 0.9f) ? 1.0f : 0.0f;
    } else {
        const float* w0 = lbs_w + (int64_t)vid[0] * J;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float* wk = lbs_w + (int64_t)vid[k] * J;
            float s = 0.f;
            for (int j = 0; j < J; ++j) s += fabsf(wk[j] - w0[j]);
            conf[k] = (expf(-s / 0.02f) > 0.9f) ? 1.0f : 0.0f;
        }
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
        const float4* M = reinterpret_cast<const float4*>(O2C + (int64_t)vid[k] * 16);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float4 m = M[r];
            T[r * 4 + 0] += w[k] * m.x; T[r * 4 + 1] += w[k] * m.y;
            T[r * 4 + 2] += w[k] * m.z; T[r * 4 + 3] += w[k] * m.w;
        }
        db += w[k] * dist[k];
    }
    float cx = T[0] * px + T[1] * py + T[2] * pz + T[3];
    float cy = T[4] * px + T[5] * py + T[6] * pz + T[7];
    float cz = T[8] * px + T[9] * py + T[10] * pz + T[11];
    pts_out[o] = make_float4(cx, cy, cz, db < thr ? 1.0f : 0.0f);
    if (nbr_w != nullptr) {                  // what the backward pass needs: blend weights and vertex ids
        reinterpret_cast<float4*>(nbr_w)[o] = make_float4(w[0], w[1], w[2], w[3]);
        reinterpret_cast<int4*>(nbr_idx)[o] = make_int4(vid[0], vid[1], vid[2], vid[3]);
    }
    if (dist_out != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { dist_out[o * 4 + k] = dist[k]; idx_out[o * 4 + k] = vid[k]; }
        blended_out[o] = db;
    }
    return db < thr;
}

}  // namespace anr
