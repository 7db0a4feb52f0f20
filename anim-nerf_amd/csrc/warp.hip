// Canonical-space warp: exact 4-nearest-vertex search + blend-weight confidence + blended inverse
// skinning transform (a7-a10), and the stand-alone KNN entry point that replaces knn_cuda.KNN.
//
// The CUDA reference materialises a V x N distance matrix in global memory (55 KB per point) and
// insertion-sorts its columns.  Here the search is exact but pruned: once per frame
// `anr_knn_index_build` lays the posed vertices out in a fixed spatial order as a three-level tree of
// bounding boxes (clusters of 8 vertices, super-clusters of 8 clusters, tops of 8 super-clusters);
// `anr_warp_points` / `anr_knn` stage that 114 KB index into LDS once per workgroup and every lane walks
// top -> super -> cluster -> vertex with wave-uniform (broadcast) LDS reads, skipping whatever cannot beat its
// current 4th-best distance.  The lanes of a wavefront hold points that are neighbours in space (see the two-pass
// renderer path below), so they agree on what to skip.  Small clusters matter for the samples that are 10-20 cm
// away from the surface (most of the valid ones): their 4th-neighbour sphere grazes a wide patch of the mesh, and
// the work is the number of vertices in the boxes it touches.
// The per-vertex tables that are only gathered for the four winners (lbs_weights 24 floats,
// ober2cano 12 floats) stay in L2.
//
// Index layout per body (floats): x[Vp] y[Vp] z[Vp] | cluster boxes NC x 8 | super boxes NS x 8 | top boxes NT x 8 |
// body box 8 | order[Vp] (int32: slot -> original vertex id) | reach mask (1,024 words).  Vp = 8 NC, NC = ceil(V/8),
// NS = ceil(NC/8), NT = ceil(NS/8).
#include "warp_core.h"

namespace anr {

// per-frame index build: one workgroup per body (blockIdx.x = 0), + one per body for the reach mask (blockIdx.x = 1, reach_thr
// > 0), next to it on another CU.  The mask in two steps, both in LDS: the occupancy of the grid (a bit per cell that holds a
// vertex), then its dilation by every cell offset whose BOX-TO-BOX distance is within thr — a superset of the cells within thr
// of a vertex by at most one cell width, at the price of ~600 word operations per (x, y) row of 32 cells instead of a
// distance test per (vertex, cell) pair (2.4 M of them: 0.2 ms per body as a first cut).
constexpr int REACH_MAX_HW = 8;                // widest stencil (cells): beyond it the grid is too fine for the radius -> no mask
constexpr int IB_THREADS = 1024;               // one cluster (8 vertices) / one row of the reach grid per thread
__global__ __launch_bounds__(IB_THREADS) void knn_index_build_kernel(const float* __restrict__ verts,
                                                              const int32_t* __restrict__ order, IndexDims d,
                                                              float* __restrict__ index, float reach_thr) {
    const int b = blockIdx.y;
    if (blockIdx.x > 0) {
        static_assert(RG == 32, "a word of the mask = the 32 z-cells of one (x, y)");
        const float* v = verts + (int64_t)b * d.V * 3;
        __shared__ float red[6][IB_THREADS / 64];
        __shared__ unsigned occ[RG * RG];
        // a thread's vertices stay in registers between the two passes (V <= 16,384: MAX_NC clusters): every load of the pass in
        // flight at once, not one trip to L2 per vertex
        constexpr int PT = MAX_NC * CS / IB_THREADS;
        float px[PT], py[PT], pz[PT];
        float lo[3] = {FAR, FAR, FAR}, hi[3] = {-FAR, -FAR, -FAR};
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int i = k * IB_THREADS + threadIdx.x;
            px[k] = py[k] = pz[k] = 0.0f;
            if (i < d.V) { px[k] = v[i * 3 + 0]; py[k] = v[i * 3 + 1]; pz[k] = v[i * 3 + 2]; }
        }
#pragma unroll
        for (int k = 0; k < PT; ++k)
            if (k * IB_THREADS + (int)threadIdx.x < d.V) {
                lo[0] = fminf(lo[0], px[k]); lo[1] = fminf(lo[1], py[k]); lo[2] = fminf(lo[2], pz[k]);
                hi[0] = fmaxf(hi[0], px[k]); hi[1] = fmaxf(hi[1], py[k]); hi[2] = fmaxf(hi[2], pz[k]);
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
            if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
        }
        for (int i = threadIdx.x; i < RG * RG; i += IB_THREADS) occ[i] = 0u;
        __syncthreads();
        float gbox[8];
#pragma unroll
        for (int a = 0; a < 3; ++a) {                          // (min / max are exact: the bits of the body box the other workgroup stores)
            float l = red[a][0], h = red[3 + a][0];
#pragma unroll
            for (int w = 1; w < IB_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            gbox[a] = l; gbox[4 + a] = h;
        }
        const float cs = reach_cell_size(gbox, reach_thr), inv = 1.0f / cs;
#pragma unroll
        for (int k = 0; k < PT; ++k)
            if (k * IB_THREADS + (int)threadIdx.x < d.V) {
                int c = reach_cell(gbox, reach_thr, inv, px[k], py[k], pz[k]);
                if (c < 0) c = 0;                               // (cannot happen for a point of the box; a NaN vertex lands here)
                atomicOr(&occ[c >> 5], 1u << (c & 31));
            }
        __syncthreads();
        // cell offsets (dx, dy, dz) with box-to-box distance cs sqrt(sum max(|d| - 1, 0)^2) <= thr (+ the roundings of
        // reach_cell on either side)
        const float rho = (reach_thr * 1.001f + 1e-5f) * inv + 1e-3f, rho2 = rho * rho;
        const int hw = (int)rho + 1;                           // |d| - 1 <= rho
        unsigned* mask = reinterpret_cast<unsigned*>(index + (int64_t)b * d.total_floats() + d.reach_off());
        for (int row = threadIdx.x; row < RG * RG; row += IB_THREADS) {
            const int ix = row >> 5, iy = row & 31;
            unsigned out = 0u;
            if (hw <= REACH_MAX_HW) {
                for (int dx = -hw; dx <= hw; ++dx) {
                    const int sx = ix + dx;
                    if (sx < 0 || sx >= RG) continue;
                    const float ax = (float)max(abs(dx) - 1, 0);
                    for (int dy = -hw; dy <= hw; ++dy) {
                        const int sy = iy + dy;
                        if (sy < 0 || sy >= RG) continue;
                        const unsigned w = occ[sx * RG + sy];
                        if (w == 0u) continue;
                        const float ay = (float)max(abs(dy) - 1, 0), rest = rho2 - ax * ax - ay * ay;
                        if (rest < 0.0f) continue;
                        const int k = (int)sqrtf(rest) + 1;    // |dz| - 1 <= sqrt(rest)
                        unsigned dil = w;
                        for (int t = 1; t <= k; ++t) dil |= (w << t) | (w >> t);
                        out |= dil;
                    }
                }
            } else {
                out = 0xffffffffu;
            }
            mask[row] = out;
        }
        return;
    }
    const float* v = verts + (int64_t)b * d.V * 3;
    float* out = index + (int64_t)b * d.total_floats();
    int32_t* ord_out = reinterpret_cast<int32_t*>(out + d.order_off());
    __shared__ float cbox[MAX_NC][6];
    __shared__ float sbox[MAX_NC / SC][6];
    __shared__ float tbox[MAX_NC / SC / TC + 1][6];
    for (int c = threadIdx.x; c < d.NC; c += blockDim.x) {
        // (the 8 slot -> vertex ids first, then the 24 coordinates: two trips to L2 per cluster, not sixteen)
        int src[CS];
        float p[CS][3];
#pragma unroll
        for (int i = 0; i < CS; ++i) {
            const int slot = c * CS + i;
            src[i] = slot < d.V ? (order ? order[slot] : slot) : -1;
        }
#pragma unroll
        for (int i = 0; i < CS; ++i) {
            p[i][0] = p[i][1] = p[i][2] = FAR;
            if (src[i] >= 0) { p[i][0] = v[src[i] * 3 + 0]; p[i][1] = v[src[i] * 3 + 1]; p[i][2] = v[src[i] * 3 + 2]; }
        }
        float lo[3] = {FAR, FAR, FAR}, hi[3] = {-FAR, -FAR, -FAR};
#pragma unroll
        for (int i = 0; i < CS; ++i) {
            const int slot = c * CS + i;
            if (src[i] >= 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[i][a]); hi[a] = fmaxf(hi[a], p[i][a]); }
            }
            out[slot] = p[i][0]; out[d.Vp + slot] = p[i][1]; out[2 * d.Vp + slot] = p[i][2];
            ord_out[slot] = src[i] < 0 ? 0 : src[i];
        }
        float* bx = out + d.box_off() + c * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[4 + a] = hi[a]; cbox[c][a] = lo[a]; cbox[c][3 + a] = hi[a]; }
        bx[3] = 0.f; bx[7] = 0.f;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < d.NS; s += blockDim.x) {
        float lo[3] = {FAR, FAR, FAR}, hi[3] = {-FAR, -FAR, -FAR};
#pragma unroll
        for (int k = 0; k < SC; ++k) {
            const int c = s * SC + k;
            if (c < d.NC) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], cbox[c][a]); hi[a] = fmaxf(hi[a], cbox[c][3 + a]); }
            }
        }
        float* bx = out + d.sbox_off() + s * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[4 + a] = hi[a]; sbox[s][a] = lo[a]; sbox[s][3 + a] = hi[a]; }
        bx[3] = 0.f; bx[7] = 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < d.NT; t += blockDim.x) {
        float lo[3] = {FAR, FAR, FAR}, hi[3] = {-FAR, -FAR, -FAR};
#pragma unroll
        for (int k = 0; k < TC; ++k) {
            const int q = t * TC + k;
            if (q < d.NS) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], sbox[q][a]); hi[a] = fmaxf(hi[a], sbox[q][3 + a]); }
            }
        }
        float* bx = out + d.tbox_off() + t * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[4 + a] = hi[a]; tbox[t][a] = lo[a]; tbox[t][3 + a] = hi[a]; }
        bx[3] = 0.f; bx[7] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                   // the body box: the union of the (<= 32) top boxes
        float lo[3] = {FAR, FAR, FAR}, hi[3] = {-FAR, -FAR, -FAR};
        for (int t = 0; t < d.NT; ++t)
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tbox[t][a]); hi[a] = fmaxf(hi[a], tbox[t][3 + a]); }
        float* bx = out + d.body_off();
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[4 + a] = hi[a]; }
        bx[3] = reach_thr > 0.0f ? reach_thr : 0.f;          // the radius the reach mask holds for (0: none)
        bx[7] = 0.f;
    }
}

__device__ __forceinline__ void stage_index(const float* __restrict__ index, int n_floats, float* lds) {
    const float4* src = reinterpret_cast<const float4*>(index);
    float4* dst = reinterpret_cast<float4*>(lds);
    // seven loads in flight per thread, then seven LDS stores (one at a time, each load waited for before the next is
    // issued, the 114 KB took 28 round trips to L2 per workgroup: ~10 % of a small-batch search)
    constexpr int U = 7;
    const int n4 = n_floats / 4, step = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < n4; i0 += U * step) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * step;
            v[u] = src[i < n4 ? i : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * step;
            if (i < n4) dst[i] = v[u];
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// reference: models/anim_nerf.py:153-192 (get_neighbs + unpose), volume_rendering.py:117
// Work item = 64 points handled by one wavefront:
//   mode RAYS : 64 neighbouring rays at one sample index k; a workgroup owns RAYS_PER_WG rays x all K samples
//   mode XYZ  : 64 consecutive explicit points; a workgroup owns PTS_PER_WG of them
// Items are handed out through an LDS counter: empty-space items cost a few instructions, surface items a full
// search, and a static split would leave most of the workgroup's 16 waves idle behind the slowest one.
constexpr int RAYS_PER_WG = 256;
constexpr int PTS_PER_WG = 16384;

template <bool FROM_RAYS>
__global__ __launch_bounds__(WARP_THREADS) void warp_points_kernel(
    const float* __restrict__ xyz, int xyz_stride, const float* __restrict__ rays, int ray_stride,
    const float* __restrict__ z, int K, const float* __restrict__ index, IndexDims d,
    const float* __restrict__ ober2cano, const float* __restrict__ lbs_w, int J, int64_t N, float thr, int skip_far,
    float4* __restrict__ pts_out, float* __restrict__ dist_out, int32_t* __restrict__ idx_out,
    float* __restrict__ blended_out, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_item;
    const int b = blockIdx.y;
    const float* my_index = index + (int64_t)b * d.total_floats();
    const int32_t* order = reinterpret_cast<const int32_t*>(my_index + d.order_off());
    const float* O2C = ober2cano + (int64_t)b * d.V * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t R = FROM_RAYS ? N / K : 0;
    const int n_items = FROM_RAYS ? (RAYS_PER_WG / 64) * K : PTS_PER_WG / 64;

    // point of this lane in work item `item`: index n, coordinates, in-range flag
    auto fetch = [&](int item, int64_t& n, float& px, float& py, float& pz) -> bool {
        bool active;
        if (FROM_RAYS) {
            const int64_t ray = (int64_t)blockIdx.x * RAYS_PER_WG + (item / K) * 64 + lane;
            const int k = item % K;
            active = ray < R;
            n = (active ? ray : 0) * K + k;
            const float* ry = rays + ((int64_t)b * R + (active ? ray : 0)) * ray_stride;
            const float zz = z[(int64_t)b * N + n];
            px = __fadd_rn(ry[0], __fmul_rn(zz, ry[3]));
            py = __fadd_rn(ry[1], __fmul_rn(zz, ry[4]));
            pz = __fadd_rn(ry[2], __fmul_rn(zz, ry[5]));
        } else {
            n = (int64_t)blockIdx.x * PTS_PER_WG + item * 64 + lane;
            active = n < N;
            if (!active) n = N - 1;
            const float* s = xyz + ((int64_t)b * N + n) * xyz_stride;
            px = s[0]; py = s[1]; pz = s[2];
        }
        return active;
    };

    // Pass 0 (renderer only): a workgroup whose samples are ALL farther than the threshold from the body's
    // bounding box writes (x, 0) and leaves without staging the index — most of the frame is empty space.
    if (threadIdx.x == 0) next_item = 0;
    if (skip_far) {
        const float* gbox = my_index + d.body_off();
        bool any_near = false;
        for (int item = wave; item < n_items; item += WARP_THREADS / 64) {
            int64_t n; float px, py, pz;
            const bool active = fetch(item, n, px, py, pz);
            const bool far = box_d2(gbox, px, py, pz) >= thr * thr;
            if (active && far) {
                pts_out[(int64_t)b * N + n] = make_float4(px, py, pz, 0.0f);
                if (nbr_w != nullptr) {
                    reinterpret_cast<float4*>(nbr_w)[(int64_t)b * N + n] = make_float4(0.f, 0.f, 0.f, 0.f);
                    reinterpret_cast<int4*>(nbr_idx)[(int64_t)b * N + n] = make_int4(0, 0, 0, 0);
                }
            }
            any_near |= active && !far;
        }
        if (!__syncthreads_or(any_near)) return;
    }
    stage_index(my_index, d.lds_floats(), lds);
    const float* body_box = lds + d.body_off();

    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&next_item, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        int64_t n;
        float px, py, pz;
        const bool active = fetch(item, n, px, py, pz);
        const int64_t o = (int64_t)b * N + n;
        // Points farther than the threshold from the whole body cannot be valid: the blended distance is a convex
        // combination of neighbour distances, all >= the distance to the body's bounding box.
        bool far = false;
        if (skip_far) far = box_d2(body_box, px, py, pz) >= thr * thr;      // already written in pass 0
        const bool go = active && !far;
        if (!__any(go)) continue;

        // Renderer mode first searches only inside the validity radius: no vertex there -> the sample is invalid
        // (blended distance >= nearest distance >= threshold) and needs no neighbours at all; four or more -> those
        // ARE the exact 4-NN.  Only the thin shell with 1-3 vertices inside the radius repeats the search unbounded.
        Best4 best;
        best_init(best, skip_far ? thr * thr * 1.0002f : 3.0e38f);
        search(lds, d, px, py, pz, go, best);
        if (skip_far) {
            const bool none = go && best.i[0] < 0;
            if (none) {
                pts_out[o] = make_float4(px, py, pz, 0.0f);
                if (nbr_w != nullptr) {
                    reinterpret_cast<float4*>(nbr_w)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                    reinterpret_cast<int4*>(nbr_idx)[o] = make_int4(0, 0, 0, 0);
                }
            }
            const bool partial = go && best.i[0] >= 0 && best.i[3] < 0;
            if (__any(partial)) {
                Best4 full;
                best_init(full);
                search(lds, d, px, py, pz, partial, full);
                if (partial) best = full;
            }
            if (none) continue;
        }
        if (!go) continue;

        blend_and_store(best, order, lbs_w, J, O2C, thr, px, py, pz, o, pts_out, dist_out, idx_out, blended_out, nbr_idx, nbr_w);
    }
}

// ---------------------------------------------------------------------------------------------
// Renderer path in two passes (skip_far with a workspace).  Most samples of a frame are empty space, and the ones
// near the body sit in a narrow depth band of some of the rays: in a wavefront of 64 neighbouring rays at one sample
// index only a few lanes have anything to search for.  So:
//   pass 1 (classify): x = o' + z d' for every sample, (x, 0) written; the samples within dis_threshold of the
//           body's bounding box are appended to a list together with the id of the grid cell they fall in
//           (64^3 cells of >= 4 cm over the bounding box), and the cell's counter is bumped;
//   pass 2 (bin):      exclusive scan of the cell counters, then a counting-sort scatter of the list by cell;
//   pass 3 (search):   persistent workgroups with the frame's index staged in LDS walk the cell-sorted list 64
//           entries per wavefront — every lane busy, and the 64 points of an item lie in one or two neighbouring
//           cells, so the lanes of the wave-uniform cluster traversal all want the same few clusters.
// The order inside a cell depends on scheduling; every lane's search is independent of its wave-mates, so the
// results do not.  Workspace: anr_warp_ws_ints(bs, N) int32.
constexpr int GRID = 64;                       // finest cell grid (full frames); small batches use a coarser one, see grid_for()
constexpr int NCELL = GRID * GRID * GRID;      // stride of the per-body cell arrays, whatever grid is in use
// One exact search per occupied cell pays off when a cell serves many samples.  A training batch (1,024 rays per body)
// occupies about as many 64^3 cells as it has near samples; a 32^3 grid has an eighth of the cells to search and scan.
__host__ inline int grid_for(int64_t samples_per_body) { return samples_per_body >= (int64_t)1 << 19 ? 64 : 32; }
// (round 6: 4 cm -> 3 cm.  The grid has 64 cells per axis over the body's box + 2 dis_threshold, so the bench's body gets
// 3.2-cm cells: more cells for the per-cell pass (coarse call 132 -> 176 us) and a tighter reach for every item of the search
// (fine call 1.84 -> 1.71 ms, coarse 0.80 -> 0.78): warp_points 4.05 -> 3.94 ms per configs[2] frame, profiles/r06/ab_min_cell.txt.
// Ordering a cell's entries by octant on top — items of one or two octants instead of a random draw from the cell — bought the
// search 3-5 % and cost more than that as its own pass: ab_suborder.txt, dropped.)
#ifndef ANR_MIN_CELL_MM
#define ANR_MIN_CELL_MM 30
#endif
constexpr float MIN_CELL = ANR_MIN_CELL_MM * 0.001f;

// Work items of a persistent kernel, handed out WITHOUT A HOT COUNTER.  One atomicAdd per item on one address was the clock of
// the search kernel: 86,760 items in 1.07 ms and 237,000 in 2.92 ms (the two passes of a cfg3 frame) are both 12.3 ns per
// item — what a same-address returning atomic costs on this part — whatever the search itself executed (cutting its box tests
// from 155 to 66 per item moved the time by 2 %).  So: the first three quarters of the items are dealt out statically
// (wavefront w of W takes w, w + W, ...: neighbours in the list go to different wavefronts), and only the last quarter — what
// evens out the unequal luck of the static part — goes through counters: DEAL_CURSORS of them, 128 bytes apart (atomics on one
// cache line queue behind each other like atomics on one address), each owning a stretch of the tail; a wavefront starts on
// cursor (its number mod DEAL_CURSORS) and moves on when a stretch is used up, looking before it queues.
constexpr int DEAL_CURSORS = 4, DEAL_STRIDE = 32, DEAL_INTS = DEAL_CURSORS * DEAL_STRIDE;
constexpr int DEAL_STATIC_16THS = 12;
struct ItemDealer {
    int n_items, n_static, n_waves, next_static, seg_len, q, tried;
    int32_t* cur;
    __device__ ItemDealer(int n_items_, int n_waves_, int my_wave, int32_t* cursors)
        : n_items(n_items_), n_waves(n_waves_), next_static(my_wave), q(my_wave % DEAL_CURSORS), tried(0), cur(cursors) {
        n_static = (int)((int64_t)n_items * DEAL_STATIC_16THS / 16) / n_waves * n_waves;
        seg_len = (n_items - n_static + DEAL_CURSORS - 1) / DEAL_CURSORS;
    }
    // the wavefront's next item (wave-uniform), -1 when there is none left
    __device__ __forceinline__ int next(int lane) {
        if (next_static < n_static) {
            const int item = next_static;
            next_static += n_waves;
            return item;
        }
        while (tried < DEAL_CURSORS) {
            int got = seg_len;
            if (lane == 0) {
                int32_t* c = cur + q * DEAL_STRIDE;
                if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg_len) got = atomicAdd(c, 1);
            }
            got = __builtin_amdgcn_readfirstlane(got);
            const int item = n_static + q * seg_len + got;
            if (got < seg_len && item < n_items) return item;
            q = (q + 1) % DEAL_CURSORS;
            ++tried;
        }
        return -1;
    }
};

struct WarpWs {
    int32_t *list, *cells, *sorted, *count, *cursor, *live, *occ_count, *occ_cursor, *cell_count, *cell_start, *occ_list, *cell_seed;
    float* cell_cap2;
    // [list | cells | sorted](bs N each) | count, live, occ_count (bs each, padded to a cache line) | cursor, occ_cursor
    // (DEAL_INTS per body each: ItemDealer) | cell_count | cell_start | cell_cap2 | occ_list | cell_seed (bs NCELL each)
    __host__ static int64_t up32(int64_t v) { return (v + 31) / 32 * 32; }
    __host__ static int64_t ints(int bs, int64_t N) { return up32(3 * (int64_t)bs * N) + up32(3 * bs) + 2 * (int64_t)bs * DEAL_INTS + 5 * (int64_t)bs * NCELL; }
    // from `count` on; a small batch (no cells) needs the counters and the cursors only
    __host__ static int64_t zeroed_ints(int bs, bool cells = true) { return up32(3 * bs) + 2 * (int64_t)bs * DEAL_INTS + (cells ? (int64_t)bs * NCELL : 0); }
    __host__ static int64_t count_off(int bs, int64_t N) { return up32(3 * (int64_t)bs * N); }
    __host__ WarpWs(int32_t* ws, int bs, int64_t N) {
        list = ws; cells = list + (int64_t)bs * N; sorted = cells + (int64_t)bs * N; count = list + up32(3 * (int64_t)bs * N);
        live = count + bs; occ_count = live + bs;
        cursor = count + up32(3 * bs); occ_cursor = cursor + (int64_t)bs * DEAL_INTS;
        cell_count = occ_cursor + (int64_t)bs * DEAL_INTS; cell_start = cell_count + (int64_t)bs * NCELL;
        cell_cap2 = reinterpret_cast<float*>(cell_start + (int64_t)bs * NCELL);
        occ_list = cell_start + 2 * (int64_t)bs * NCELL;
        cell_seed = cell_start + 3 * (int64_t)bs * NCELL;
    }
};

__device__ __forceinline__ float cell_size(const float* gbox, float thr, int G) {
    const float ex = gbox[4] - gbox[0], ey = gbox[5] - gbox[1], ez = gbox[6] - gbox[2];
    return fmaxf(MIN_CELL, (fmaxf(fmaxf(ex, ey), ez) + 2.0f * thr) * (1.0f / (float)G));
}
// (inv = 1.0f / cell_size(gbox, thr, G): the same for every sample of a body — callers in a loop compute it once)
__device__ __forceinline__ int cell_of_inv(const float* gbox, float thr, int G, float inv, float px, float py, float pz) {
    const int ix = min(max((int)((px - gbox[0] + thr) * inv), 0), G - 1);
    const int iy = min(max((int)((py - gbox[1] + thr) * inv), 0), G - 1);
    const int iz = min(max((int)((pz - gbox[2] + thr) * inv), 0), G - 1);
    return (ix * G + iy) * G + iz;
}
__device__ __forceinline__ int cell_of(const float* gbox, float thr, int G, float px, float py, float pz) {
    return cell_of_inv(gbox, thr, G, 1.0f / cell_size(gbox, thr, G), px, py, pz);
}

// Per-workgroup aggregation of the cell counters: neighbouring rays and consecutive samples fall into the same few
// cells, and one global atomic per sample on those hot counters costs more than the search it prepares.  A workgroup
// counts in an LDS hash table (cell -> count) and touches the global counter once per distinct cell.
constexpr int HN = 2048;                       // hash slots per workgroup
__device__ __forceinline__ int hash_slot(int* keys, int cell) {
    unsigned s = ((unsigned)cell * 2654435761u) >> 21;
#pragma unroll 1
    for (int t = 0; t < 16; ++t) {
        const int prev = atomicCAS(&keys[s], -1, cell);
        if (prev == -1 || prev == cell) return (int)s;
        s = (s + 1) & (HN - 1);
    }
    return -1;                                 // table crowded: the caller falls back to the global counter
}

constexpr int CLS_ITERS = 8;                   // samples per classify workgroup = 8 x 1024 (16: 0.57 / 1.08 ms per call instead of
                                               // 0.45 / 0.96; 32: 0.43 / 1.10)
// VEC4 (rays mode, K % 4 == 0): a thread takes FOUR CONSECUTIVE samples of one ray per step — one 16-byte load of depths, one
// dword of merge permutation in, one dword of validity bytes out — instead of four byte-wide accesses 1,024 samples apart:
// the pass is a stream over z / perm / mask (11 B per sample) and ran at 1.4 TB/s on byte traffic.
// CELLS = false (a small batch per body — training: the near list goes straight to the per-sample search, nobody reads the
// cells): no cell of a near sample, no hash table, no per-cell counts.
template <bool FROM_RAYS, bool VEC4, bool CELLS = true>
__global__ __launch_bounds__(WARP_THREADS) void warp_classify_kernel(
    const float* __restrict__ xyz, int xyz_stride, const float* __restrict__ rays, int ray_stride,
    const float* __restrict__ z, int K, const float* __restrict__ index, IndexDims d, int64_t N, float thr,
    float4* __restrict__ pts_out, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_w, int32_t* __restrict__ list,
    int32_t* __restrict__ cells, int32_t* __restrict__ count, int32_t* __restrict__ cell_count,
    uint8_t* __restrict__ valid_mask, const float4* __restrict__ reuse_pts, const uint8_t* __restrict__ reuse_mask,
    const uint8_t* __restrict__ perm, int reuse_K, int G, const int4* __restrict__ reuse_nbr_idx = nullptr,
    const float4* __restrict__ reuse_nbr_w = nullptr) {
    static_assert(!VEC4 || FROM_RAYS, "four samples per thread: rays mode");
    __shared__ int wave_cnt[WARP_THREADS / 64];
    __shared__ int block_base;
    __shared__ int hkeys[HN], hcnt[HN];
    const int b = blockIdx.y;
    const float* gbox = index + (int64_t)b * d.total_floats() + d.body_off();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (CELLS) {
        for (int s = threadIdx.x; s < HN; s += WARP_THREADS) { hkeys[s] = -1; hcnt[s] = 0; }
        __syncthreads();
    }
    // each thread classifies CLS_ITERS samples and remembers its near ones; the list is then written with ONE
    // block-wide compaction (one scan, one global atomic, no barrier per iteration)
    constexpr int VS = VEC4 ? 4 : 1, STEPS = CLS_ITERS / VS;
    int my_cell[CLS_ITERS];
    unsigned near_bits = 0;
    const uint32_t R32 = (uint32_t)(N / (K > 0 ? K : 1));
    const float cell_inv = 1.0f / cell_size(gbox, thr, G);
    // the reach mask (knn_index_build_kernel), if the index carries one for a radius >= thr
    const float reach_thr = gbox[3];
    const bool masked = reach_thr >= thr;
    const float reach_inv = masked ? 1.0f / reach_cell_size(gbox, reach_thr) : 0.0f;
    const unsigned* __restrict__ reach = reinterpret_cast<const unsigned*>(index + (int64_t)b * d.total_floats() + d.reach_off());
    // flat index of the thread's sample (step, v)
    auto sample_of = [&](int step, int v) { return (((int64_t)blockIdx.x * STEPS + step) * WARP_THREADS + threadIdx.x) * VS + v; };
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int64_t n0 = sample_of(step, 0);
        float zz[VS];
        unsigned pm = 0, mask_out = 0;
        if (VEC4 && n0 < N) {                                   // (N % 4 == 0: the four samples are in range together)
            const float4 z4 = *reinterpret_cast<const float4*>(z + (int64_t)b * N + n0);
            zz[0] = z4.x; zz[VS > 1 ? 1 : 0] = z4.y; zz[VS > 2 ? 2 : 0] = z4.z; zz[VS > 3 ? 3 : 0] = z4.w;
            if (perm != nullptr) pm = *reinterpret_cast<const unsigned*>(perm + (int64_t)b * N + n0);
        }
        // VEC4: the four samples belong to one ray (K % 4 == 0, n0 % 4 == 0): its index and its six floats once per step
        uint32_t ray4 = 0;
        float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 0.f};
        if (VEC4 && n0 < N) {
            ray4 = (uint32_t)n0 / (uint32_t)K;                                         // (N < 2^31 on this path)
            const float* ry = rays + ((int64_t)b * R32 + ray4) * ray_stride;
            ro[0] = ry[0]; ro[1] = ry[1]; ro[2] = ry[2]; rd[0] = ry[3]; rd[1] = ry[4]; rd[2] = ry[5];
        }
#pragma unroll
        for (int v = 0; v < VS; ++v) {
            const int it = step * VS + v;
            const int64_t n = n0 + v;
            bool near = false;
            int cell = 0;
            if (n < N) {
                float px, py, pz;
                const uint32_t ray = VEC4 ? ray4 : FROM_RAYS ? (uint32_t)n / (uint32_t)K : 0;
                if (VEC4) {
                    px = __fadd_rn(ro[0], __fmul_rn(zz[v], rd[0]));
                    py = __fadd_rn(ro[1], __fmul_rn(zz[v], rd[1]));
                    pz = __fadd_rn(ro[2], __fmul_rn(zz[v], rd[2]));
                } else if (FROM_RAYS) {
                    const float* ry = rays + ((int64_t)b * R32 + ray) * ray_stride;
                    const float zv = z[(int64_t)b * N + n];
                    px = __fadd_rn(ry[0], __fmul_rn(zv, ry[3]));
                    py = __fadd_rn(ry[1], __fmul_rn(zv, ry[4]));
                    pz = __fadd_rn(ry[2], __fmul_rn(zv, ry[5]));
                } else {
                    const float* sp = xyz + ((int64_t)b * N + n) * xyz_stride;
                    px = sp[0]; py = sp[1]; pz = sp[2];
                }
                const int64_t o = (int64_t)b * N + n;
                // farther than the threshold from the whole body -> cannot be valid (see warp_points_kernel)
                near = box_d2(gbox, px, py, pz) < thr * thr;
                if (masked && near) {                       // ... and from every vertex, unless its cell's bit says otherwise
                    const int rc = reach_cell(gbox, reach_thr, reach_inv, px, py, pz);
                    near = rc >= 0 && ((reach[rc >> 5] >> (rc & 31)) & 1u) != 0u;
                }
                bool reused = false;
                if (FROM_RAYS && perm != nullptr) {
                    // fine pass: this sorted sample IS coarse sample p of the same ray (z_sorted[j] = cat(z_coarse, z_fine)
                    // [perm[j]]) -> its canonical point and validity were computed in the coarse pass: copy, do not search
                    const int pj = VEC4 ? (int)((pm >> (8 * v)) & 0xffu) : (int)perm[o];
                    if (pj < reuse_K) {
                        const int64_t src = ((int64_t)b * R32 + ray) * reuse_K + pj;
                        if (reuse_mask != nullptr) {
                            const uint8_t m = reuse_mask[src];
                            if (VEC4) mask_out |= (unsigned)m << (8 * v);
                            else valid_mask[o] = m;
                            if (m) pts_out[o] = reuse_pts[src];
                        } else {
                            // training (no validity bytes): the coarse call's whole row — (x_c, 1) or (x, 0) — and, for the
                            // backward pass, its neighbour ids and blend weights
                            pts_out[o] = reuse_pts[src];
                            if (nbr_w != nullptr) {
                                reinterpret_cast<float4*>(nbr_w)[o] = reuse_nbr_w[src];
                                reinterpret_cast<int4*>(nbr_idx)[o] = reuse_nbr_idx[src];
                            }
                        }
                        near = false;
                        reused = true;
                    }
                }
                if (!reused) {
                    // lean mode (validity bytes requested): consumers look at the byte, not at the point, so the 16-B point
                    // of a far sample is not written at all
                    if (!VEC4 && valid_mask != nullptr) valid_mask[o] = 0;
                    if (valid_mask == nullptr || near) pts_out[o] = make_float4(px, py, pz, 0.0f);
                    if (nbr_w != nullptr) {
                        reinterpret_cast<float4*>(nbr_w)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                        reinterpret_cast<int4*>(nbr_idx)[o] = make_int4(0, 0, 0, 0);
                    }
                    if (CELLS && near) {
                        cell = cell_of_inv(gbox, thr, G, cell_inv, px, py, pz);
                        const int slot = hash_slot(hkeys, cell);
                        if (slot >= 0) atomicAdd(&hcnt[slot], 1);
                        else atomicAdd(cell_count + (int64_t)b * NCELL + cell, 1);
                    }
                }
            }
            my_cell[it] = cell;
            near_bits |= (near ? 1u : 0u) << it;
        }
        if (VEC4 && valid_mask != nullptr && n0 < N) *reinterpret_cast<unsigned*>(valid_mask + (int64_t)b * N + n0) = mask_out;
    }
    {
        const int mine = __popc(near_bits);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
#pragma unroll
            for (int w = 0; w < WARP_THREADS / 64; ++w) { int c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
            block_base = tot ? atomicAdd(count + b, tot) : 0;
        }
        __syncthreads();
        int64_t pos = (int64_t)b * N + block_base + wave_cnt[wave] + incl - mine;
#pragma unroll
        for (int it = 0; it < CLS_ITERS; ++it) {
            if ((near_bits >> it) & 1u) {
                list[pos] = (int32_t)sample_of(it / VS, it % VS);
                if (CELLS) cells[pos] = my_cell[it];
                ++pos;
            }
        }
    }
    if (!CELLS) return;
    __syncthreads();
    for (int s = threadIdx.x; s < HN; s += WARP_THREADS)
        if (hkeys[s] >= 0) atomicAdd(cell_count + (int64_t)b * NCELL + hkeys[s], hcnt[s]);
}

// The renderer's lean pass (validity bytes out, no neighbour outputs; rays mode, four consecutive samples per thread) written in
// PHASES, so that every load of a thread's eight samples is in flight before the first one is used.  In the generic kernel
// above a sample's chain — merge permutation -> validity byte of the coarse sample it copies -> that sample's point -> store —
// ran once per sample behind the previous sample's (round 4's counters: 70 % of the wave-cycles waiting on memory at 30 % VALU
// issue, 2 TB/s): here the two 16-byte depth loads, the two permutation dwords and the rays go first, then the eight validity
// bytes and reach-mask words, then the (up to) eight points, then the stores.  Same outputs.
// Shape of this pass: 512 threads x 4 samples, not the 1,024 x 8 of the other classify kernel.  At eight samples per thread the
// kernel needs 108 VGPRs — four waves per SIMD, i.e. ONE 1,024-thread workgroup per CU — and its phases end in two workgroup
// barriers around a global atomic (the list range): with one workgroup resident the CU idles through every one of those waits.
// Per configs[2] frame, by the kernel trace on one box (profiles/r05/ab_lean_threads.txt): 1,024 x 8: 1.17 ms; 256 x 8: 1.00;
// 256 x 4: 1.00; 1,024 x 4: 1.09; **512 x 4: 0.89**.
#ifndef ANR_LEAN_THREADS
#define ANR_LEAN_THREADS 512
#endif
constexpr int LEAN_THREADS = ANR_LEAN_THREADS;
#ifndef ANR_LEAN_ITERS
#define ANR_LEAN_ITERS 4
#endif
constexpr int LEAN_ITERS = ANR_LEAN_ITERS;       // samples per thread of the lean pass (a multiple of 4)
template <bool CELLS>
__global__ __launch_bounds__(LEAN_THREADS) void warp_classify_lean_kernel(
    const float* __restrict__ rays, int ray_stride, const float* __restrict__ z, int K, const float* __restrict__ index, IndexDims d,
    int64_t N, float thr, float4* __restrict__ pts_out, int32_t* __restrict__ list, int32_t* __restrict__ cells,
    int32_t* __restrict__ count, int32_t* __restrict__ cell_count, uint8_t* __restrict__ valid_mask,
    const float4* __restrict__ reuse_pts, const uint8_t* __restrict__ reuse_mask, const uint8_t* __restrict__ perm, int reuse_K,
    int G, int z_steps) {
    // z_steps (round 6): `z` is the step table s[K] of the deterministic stratified depths, z_k = near' (1 - s_k) + far' s_k
    // (models/volume_rendering.py:43-44) with the roundings of anr_sample_coarse — the coarse depth array of an inference
    // frame (4 B per sample written, then read here and by the fused coarse pass) is not made at all.
    __shared__ int wave_cnt[LEAN_THREADS / 64];
    __shared__ int block_base;
    __shared__ int hkeys[HN], hcnt[HN];
    __shared__ float4 near_stage[LEAN_THREADS / 64][64];
    const int b = blockIdx.y;
    const float* gbox = index + (int64_t)b * d.total_floats() + d.body_off();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int VS = 4, STEPS = LEAN_ITERS / VS;
    const uint32_t R32 = (uint32_t)(N / K);
    auto sample_of = [&](int step, int v) { return (((int64_t)blockIdx.x * STEPS + step) * LEAN_THREADS + threadIdx.x) * VS + v; };
    // ---- A0 (round 6): the rays, and whether a ray's line comes within thr of the body's box AT ALL (three slabs
    // of the box padded by a little more than thr: a superset of the points with box_d2 < thr^2, margins far above the roundings).
    // Most rays of a frame pass the body by: their samples are neither near nor copies of a valid coarse sample (those were
    // classified by the same test), so a workgroup that holds
    // no other ray writes its zero bytes and ends before the first barrier, and the other phases skip them.
    uint32_t ray4[STEPS];
    bool in[STEPS], act[STEPS];
    // (the depths and permutation bytes are issued WITH the rays, not after the test: a ray that misses wastes 20 bytes per
    // sample of a pass that is not short of bandwidth, a ray that hits saves a memory trip — fine call 0.63 -> 0.60 ms)
    // (sample -> ray: K is 64 or 128 on every shipped shape — a shift; the general 32-bit division is ~25 instructions of the ~700
    // a wavefront of this pass executes)
    const int k_shift = (K & (K - 1)) == 0 ? 31 - __builtin_clz((unsigned)K) : -1;
    auto ray_of = [&](uint32_t n) { return k_shift >= 0 ? n >> k_shift : n / (uint32_t)K; };
    float4 z4e[STEPS];
    unsigned pme[STEPS];
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int64_t n0 = sample_of(step, 0);
        z4e[step] = make_float4(0.f, 0.f, 0.f, 0.f);
        pme[step] = 0u;
        if (n0 < N) {
            // (K % 4 == 0: the four samples are consecutive entries of one ray's step table)
            z4e[step] = z_steps ? *reinterpret_cast<const float4*>(z + ((uint32_t)n0 - ray_of((uint32_t)n0) * (uint32_t)K))
                                : *reinterpret_cast<const float4*>(z + (int64_t)b * N + n0);
            if (perm != nullptr) pme[step] = *reinterpret_cast<const unsigned*>(perm + (int64_t)b * N + n0);
        }
    }
    float ro[STEPS][3], rd[STEPS][3], nr[STEPS], fr[STEPS];
    bool any_act = false;
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int64_t n0 = sample_of(step, 0);
        in[step] = n0 < N;                                  // (N % 4 == 0: the four samples are in range together)
        ray4[step] = in[step] ? ray_of((uint32_t)n0) : 0u;                            // (N < 2^31 on this path)
        const float* ry = rays + ((int64_t)b * R32 + ray4[step]) * ray_stride;
#pragma unroll
        for (int a = 0; a < 3; ++a) { ro[step][a] = in[step] ? ry[a] : 0.0f; rd[step][a] = in[step] ? ry[3 + a] : 0.0f; }
        nr[step] = in[step] ? ry[6] : 0.0f; fr[step] = in[step] ? ry[7] : 0.0f;
        // (the whole LINE, not the segment: true for any depth a caller passes)
        const float pad = thr * 1.01f + 1.0e-4f;
        float t0 = -3.0e38f, t1 = 3.0e38f;
        bool hit = in[step];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = gbox[a] - pad - ro[step][a], hi = gbox[4 + a] + pad - ro[step][a];
            if (fabsf(rd[step][a]) > 1.0e-12f) {
                const float inv = __builtin_amdgcn_rcpf(rd[step][a]);      // (1 ulp: the pad's margin is five orders above it)
                const float ta = lo * inv, tb = hi * inv;
                t0 = fmaxf(t0, fminf(ta, tb));
                t1 = fminf(t1, fmaxf(ta, tb));
            } else if (lo > 0.0f || hi < 0.0f) {
                hit = false;
            }
        }
        act[step] = hit && t0 <= t1;
        any_act |= act[step];
    }
    if (!__syncthreads_or(any_act ? 1 : 0)) {
#pragma unroll
        for (int step = 0; step < STEPS; ++step)
            if (in[step]) *reinterpret_cast<unsigned*>(valid_mask + (int64_t)b * N + sample_of(step, 0)) = 0u;
        return;
    }
    if (CELLS) {
        for (int s = threadIdx.x; s < HN; s += LEAN_THREADS) { hkeys[s] = -1; hcnt[s] = 0; }
        __syncthreads();
    }
    const float cell_inv = 1.0f / cell_size(gbox, thr, G);
    const float reach_thr = gbox[3];
    const bool masked = reach_thr >= thr;
    const float reach_inv = masked ? 1.0f / reach_cell_size(gbox, reach_thr) : 0.0f;
    const unsigned* __restrict__ reach = reinterpret_cast<const unsigned*>(index + (int64_t)b * d.total_floats() + d.reach_off());
    // ---- A: depths, permutation bytes (loaded above)
    float zz[LEAN_ITERS];
    unsigned pm[STEPS];
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const float4 z4 = act[step] ? z4e[step] : make_float4(0.f, 0.f, 0.f, 0.f);
        pm[step] = act[step] ? pme[step] : 0u;
        zz[step * VS + 0] = z4.x; zz[step * VS + 1] = z4.y; zz[step * VS + 2] = z4.z; zz[step * VS + 3] = z4.w;
    }
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        if (z_steps) {
#pragma clang fp contract(off)                                     // (anr_sample_coarse rounds the two products and the sum separately;
            // HIP's __fmul_rn / __fadd_rn are plain operators, which this file's default would contract into an fma)
            const float near = nr[step], far = fr[step];
#pragma unroll
            for (int v = 0; v < VS; ++v) {
                const float sk = zz[step * VS + v];
                const float lo = near * (1.0f - sk), hi = far * sk;
                zz[step * VS + v] = lo + hi;
            }
        }
    }
    // ---- B: positions, the box test, the coarse samples' validity bytes, the reach-mask words.  Addresses first, then the eight
    // loads UNCONDITIONALLY (a lane without a byte or a word to fetch reads element 0) and together: written as `if (...) load`
    // per sample the compiler ended every sample's branch in a wait, four trips to L2 one behind the other.
    float px[LEAN_ITERS], py[LEAN_ITERS], pz[LEAN_ITERS];
    int64_t src[LEAN_ITERS];
    unsigned m[LEAN_ITERS], rw[LEAN_ITERS];
    unsigned near_bits = 0u, reused_bits = 0u;
    int rbit[LEAN_ITERS], rword[LEAN_ITERS];
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it) {
        const int step = it / VS, v = it % VS;
        px[it] = __fadd_rn(ro[step][0], __fmul_rn(zz[it], rd[step][0]));
        py[it] = __fadd_rn(ro[step][1], __fmul_rn(zz[it], rd[step][1]));
        pz[it] = __fadd_rn(ro[step][2], __fmul_rn(zz[it], rd[step][2]));
        bool near = act[step] && box_d2(gbox, px[it], py[it], pz[it]) < thr * thr;
        rbit[it] = 0; rword[it] = -1; src[it] = 0;
        bool reused = false;
        if (act[step] && perm != nullptr) {
            // this sorted sample IS coarse sample pj of the same ray: its canonical point and validity were computed in the
            // coarse pass
            const int pj = (int)((pm[step] >> (8 * v)) & 0xffu);
            if (pj < reuse_K) {
                src[it] = ((int64_t)b * R32 + ray4[step]) * reuse_K + pj;
                reused = true;
                near = false;
            }
        }
        if (masked && near) {
            const int rc = reach_cell(gbox, reach_thr, reach_inv, px[it], py[it], pz[it]);
            if (rc >= 0) { rword[it] = rc >> 5; rbit[it] = rc & 31; } else { near = false; }
        }
        near_bits |= (near ? 1u : 0u) << it;
        reused_bits |= (reused ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it) m[it] = 0u;
    if (perm != nullptr) {                                   // (uniform: the fine call)
#pragma unroll
        for (int it = 0; it < LEAN_ITERS; ++it) m[it] = reuse_mask[src[it]];
    }
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it) rw[it] = reach[max(rword[it], 0)];
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it) {
        if (!((reused_bits >> it) & 1u)) m[it] = 0u;
        if (rword[it] < 0) rw[it] = 0xffffffffu;
    }
    // ---- B2: the reach bits settle which samples are listed; the workgroup's list range is reserved NOW, and the atomic's trip
    // runs next to phase C's loads instead of after them (the pass is bound by the length of a thread's chain of dependent memory
    // trips — rays / depths, validity bytes, points, the counter — times the threads a CU holds, not by bytes or instructions)
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it)
        if (((near_bits >> it) & 1u) && ((rw[it] >> rbit[it]) & 1u) == 0u) near_bits &= ~(1u << it);      // no vertex can reach this cell
    const int mine = __popc(near_bits);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    // ---- C: the points of the valid coarse samples
    float4 rp[LEAN_ITERS];
#pragma unroll
    for (int it = 0; it < LEAN_ITERS; ++it) {
        rp[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m[it]) rp[it] = reuse_pts[src[it]];
    }
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < LEAN_THREADS / 64; ++w) { int c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
        block_base = tot ? atomicAdd(count + b, tot) : 0;
    }
    __syncthreads();
    // ---- D: stores.  The copied samples' bytes and points per thread, as laid out ...
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        unsigned mask_out = 0u;
#pragma unroll
        for (int v = 0; v < VS; ++v) {
            const int it = step * VS + v;
            if ((reused_bits >> it) & 1u) {
                mask_out |= m[it] << (8 * v);
                if (m[it]) pts_out[(int64_t)b * N + sample_of(step, v)] = rp[it];
            }
        }
        if (in[step]) *reinterpret_cast<unsigned*>(valid_mask + (int64_t)b * N + sample_of(step, 0)) = mask_out;
    }
    // ... and the NEAR samples (9 % of a frame's, but some in almost every wavefront) on DENSE lanes: a near sample's work — its
    // cell, the hash table, its point, its list entry and its cell entry: ~60 instructions — ran once per sample slot of the
    // thread, four times per wavefront with a few lanes active each time.  The wave's near samples already have consecutive list
    // positions (the scan above), so they are handed over through 1 KB of the wave's LDS, 64 at a time, to the lanes in list
    // order: one pass of full lanes for most wavefronts, and list / cells stores that are consecutive dwords.
    {
        const int wave_first = incl - mine;                                   // this thread's first near sample among the wave's
        const int wave_total = __builtin_amdgcn_readlane(incl, 63);
        const int64_t list_base = (int64_t)b * N + block_base + wave_cnt[wave];
        float4* mine_stage = near_stage[wave];
        for (int r0 = 0; r0 < wave_total; r0 += 64) {
            int k = wave_first;
#pragma unroll
            for (int it = 0; it < LEAN_ITERS; ++it) {
                if ((near_bits >> it) & 1u) {
                    if (k >= r0 && k < r0 + 64)
                        mine_stage[k - r0] = make_float4(px[it], py[it], pz[it], __int_as_float((int)sample_of(it / VS, it % VS)));
                    ++k;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (r0 + lane < wave_total) {
                const float4 e = mine_stage[lane];
                const int sid = __float_as_int(e.w);
                pts_out[(int64_t)b * N + sid] = make_float4(e.x, e.y, e.z, 0.0f);
                list[list_base + r0 + lane] = sid;
                if (CELLS) {
                    const int cell = cell_of_inv(gbox, thr, G, cell_inv, e.x, e.y, e.z);
                    cells[list_base + r0 + lane] = cell;
                    const int slot = hash_slot(hkeys, cell);
                    if (slot >= 0) atomicAdd(&hcnt[slot], 1);
                    else atomicAdd(cell_count + (int64_t)b * NCELL + cell, 1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (the next round overwrites what was just read)
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!CELLS) return;
    __syncthreads();
    for (int s = threadIdx.x; s < HN; s += LEAN_THREADS)
        if (hkeys[s] >= 0) atomicAdd(cell_count + (int64_t)b * NCELL + hkeys[s], hcnt[s]);
}

// Per occupied cell, one exact search from the cell's centre c (r = half diagonal):
//   * nearest vertex farther than dis_threshold + r  ->  every point of the cell is farther than dis_threshold from
//     every vertex, its blended distance (a convex combination of neighbour distances) too: the cell is dead, its
//     samples stay (x, 0) and are dropped from the list;
//   * otherwise (d4(c) + r)^2 bounds the 4th-neighbour distance of every point of the cell: searching inside that
//     radius finds the exact four neighbours in one go (no unbounded retry).
// cell_cap2[cell] = that squared radius, or -1 for a dead cell; defined for cells with a non-zero count only.
__global__ __launch_bounds__(WARP_THREADS) void warp_cell_list_kernel(const int32_t* __restrict__ cell_count,
                                                                      int32_t* __restrict__ occ_list,
                                                                      int32_t* __restrict__ occ_count,
                                                                      const float* __restrict__ prev_cap2,
                                                                      const int32_t* __restrict__ prev_seed,
                                                                      float* __restrict__ cell_cap2,
                                                                      int32_t* __restrict__ cell_seed) {
    // ids of the occupied cells, compacted (order preserved inside blocks of 4096 cells): the per-cell searches below
    // then run with full wavefronts instead of one wavefront per 64 consecutive cells of which a few are occupied.
    // Round 6: a cell's radius and seed depend on the body, the grid and dis_threshold, not on the samples — the fine pass of a
    // frame occupies the cells its coarse pass occupied (the same rays, sampled more densely where the surface is).  With
    // prev_cap2 / prev_seed (the arrays of the earlier call's workspace: anr_warp_points_cells) a cell that call searched is
    // copied and left off the list; every call leaves 0 = "not searched" in the cells it did not occupy, for a later one.
    __shared__ int wave_cnt[WARP_THREADS / 64];
    __shared__ int block_base;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = (blockIdx.x * WARP_THREADS + threadIdx.x) * 4;
    const int4 v = *reinterpret_cast<const int4*>(cell_count + (int64_t)b * NCELL + first);
    float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 sd = make_int4(0, 0, 0, 0);
    if (prev_cap2 != nullptr) {
        cp = *reinterpret_cast<const float4*>(prev_cap2 + (int64_t)b * NCELL + first);
        sd = *reinterpret_cast<const int4*>(prev_seed + (int64_t)b * NCELL + first);
    }
    if (v.x <= 0) cp.x = 0.f;
    if (v.y <= 0) cp.y = 0.f;
    if (v.z <= 0) cp.z = 0.f;
    if (v.w <= 0) cp.w = 0.f;
    const bool lx = v.x > 0 && cp.x == 0.f, ly = v.y > 0 && cp.y == 0.f, lz = v.z > 0 && cp.z == 0.f, lw = v.w > 0 && cp.w == 0.f;
    *reinterpret_cast<float4*>(cell_cap2 + (int64_t)b * NCELL + first) = cp;
    if (prev_cap2 != nullptr) *reinterpret_cast<int4*>(cell_seed + (int64_t)b * NCELL + first) = sd;
    const int mine = (int)lx + (int)ly + (int)lz + (int)lw;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < WARP_THREADS / 64; ++w) { int c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
        block_base = tot ? atomicAdd(occ_count + b, tot) : 0;
    }
    __syncthreads();
    int pos = block_base + wave_cnt[wave] + incl - mine;
    int32_t* out = occ_list + (int64_t)b * NCELL;
    if (lx) out[pos++] = first;
    if (ly) out[pos++] = first + 1;
    if (lz) out[pos++] = first + 2;
    if (lw) out[pos++] = first + 3;
}

// lanes per work item of the two search kernels: 64 when there is plenty of work, down to 8 when there is not
__device__ __forceinline__ int lanes_per_item(int n, int wave_slots) {
    int lpi = 64;
    while (lpi > 8 && (n + lpi - 1) / lpi < 2 * wave_slots) lpi >>= 1;
    return lpi;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// k-th smallest (k = 1..4 -> out[k-1]) of the union of the lanes' ascending lists a[4]: four rounds of "take the wave's
// minimum, the lowest lane that holds it moves on to its next entry"
__device__ __forceinline__ void wave_smallest4(const float (&a)[4], float (&out)[4], int& first_lane) {
    int p = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float cur = p == 0 ? a[0] : p == 1 ? a[1] : p == 2 ? a[2] : p == 3 ? a[3] : 3.0e38f;
        const float m = wave_min(cur);
        const int who = __builtin_ctzll(__ballot(cur == m));
        if (k == 0) first_lane = who;
        if ((int)(threadIdx.x & 63) == who) ++p;
        out[k] = m;
    }
}

// ONE WAVEFRONT PER OCCUPIED CELL, the lanes splitting the index: a lane owns the clusters lane, lane + 64, ...  (A lane per
// cell — the first version — made a wavefront walk the union of what 64 cells scattered over the body need, ~10^5 dependent
// instructions per item at one instruction per ~12 cycles, and a training batch has only ~10^3 such items for 4,096 wave
// slots: 0.58 ms per call at 9 % VALU issue.)
//   1. every lane: box distance of each of its clusters, the nearest one remembered; the wave's minimum is a LOWER bound on
//      the distance to the nearest vertex: at dis_threshold + r or more the cell is dead, done (most occupied cells lie
//      inside the body's bounding box but far from its surface);
//   2. every lane scans its nearest cluster; the 4th smallest distance over the wave bounds d4 from above;
//   3. every lane scans those of its other clusters whose box is inside that bound (and its own 4th best);
//   4. d1, d4 and the nearest vertex = the wave's 1st / 4th smallest over the lanes' best lists.
// What the search pass needs from here is a radius that contains the four neighbours of every point of the cell and a
// cluster to start from; neither depends on the order of the scan.
__global__ __launch_bounds__(WARP_THREADS) void warp_cells_kernel(const float* __restrict__ index, IndexDims d, float thr,
                                                                  const int32_t* __restrict__ occ_list,
                                                                  const int32_t* __restrict__ occ_count,
                                                                  int32_t* __restrict__ occ_cursor,
                                                                  float* __restrict__ cell_cap2,
                                                                  int32_t* __restrict__ cell_seed, int G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int n_occ = occ_count[b];
    if ((int)blockIdx.x * (WARP_THREADS / 64) >= n_occ) return;          // not even one cell per wavefront left
    const int32_t* occ = occ_list + (int64_t)b * NCELL;
    float* cap = cell_cap2 + (int64_t)b * NCELL;
    stage_index(index + (int64_t)b * d.total_floats(), d.lds_floats(), lds);
    const float* gbox = lds + d.body_off();
    const float* boxes = lds + d.box_off();
    const float cs = cell_size(gbox, thr, G);
    const float r = cs * 0.8662f;                        // sqrt(3)/2, rounded up
    const float lim2 = (thr + r) * (thr + r);
    const int lane = threadIdx.x & 63;
    // (cells are handed out by the counter, eight per trip: dead cells cost a few instructions and live ones a search — dealt out
    // statically (ItemDealer) the pass took 0.33 ms instead of 0.25; its ~20,000 trips are not what bounds it)
    constexpr int BATCH = 8;                             // cells per trip (4: 0.25 / 0.19 ms per call, 8: 0.19 / 0.16, 16: 0.21 / 0.19, 32: 0.29 / 0.28)
    // (round 6: the list split over 4 or 2 counters 128 bytes apart, 8 / 4 / 2 cells per trip — no faster than this, fewer cells
    // per trip slower: profiles/r06/ab_cells_cursors.txt.  It is the trip's latency a wavefront waits through, not the counter's rate)
    for (;;) {
        int first = 0;
        if (lane == 0) first = atomicAdd(occ_cursor + (int64_t)b * DEAL_INTS, BATCH);
        first = __builtin_amdgcn_readfirstlane(first);
        if (first >= n_occ) break;
        for (int i = first; i < min(first + BATCH, n_occ); ++i) {
            const int cell = occ[i];
            const int ix = cell / (G * G), iy = (cell / G) % G, iz = cell % G;
            const float cx = gbox[0] - thr + ((float)ix + 0.5f) * cs;
            const float cy = gbox[1] - thr + ((float)iy + 0.5f) * cs;
            const float cz = gbox[2] - thr + ((float)iz + 0.5f) * cs;
            int c_near = -1;
            float b_near = 3.0e38f;
            // (SMPL's 862 clusters are 14 per lane: the first sweep's box distances stay in registers for the second — a box test is
            // ~14 instructions, the two sweeps were half of what the pass executes; larger meshes take the loops that test twice)
            constexpr int BOX_ROUNDS = 14;
            const bool cached = d.NC <= 64 * BOX_ROUNDS;
            float bd[BOX_ROUNDS];
            if (cached) {
#pragma unroll
                for (int q = 0; q < BOX_ROUNDS; ++q) {
                    const int c = q * 64 + lane;
                    const float v = box_d2(boxes + min(c, d.NC - 1) * 8, cx, cy, cz);
                    bd[q] = c < d.NC ? v : 3.0e38f;
                    if (bd[q] < b_near) { b_near = bd[q]; c_near = c; }
                }
            } else {
                for (int c = lane; c < d.NC; c += 64) {
                    const float v = box_d2(boxes + c * 8, cx, cy, cz);
                    if (v < b_near) { b_near = v; c_near = c; }
                }
            }
            if (wave_min(b_near) >= lim2) {
                if (lane == 0) cap[cell] = -1.0f;
                continue;
            }
            Best4 best;
            best_init(best);
            if (c_near >= 0) scan_cluster(lds, d.Vp, c_near, cx, cy, cz, best);
            float sm[4];
            int who;
            wave_smallest4(best.d, sm, who);
            const float bound = sm[3];                   // >= d4^2: the 4th smallest of a subset of the vertices
            if (cached) {
                unsigned cand = 0u;                      // this lane's clusters inside the bound, one bit per round
#pragma unroll
                for (int q = 0; q < BOX_ROUNDS; ++q)
                    if (q * 64 + lane != c_near && bd[q] <= bound) cand |= 1u << q;
                while (__any(cand != 0u)) {
                    if (cand != 0u) {
                        const int c = __builtin_ctz(cand) * 64 + lane;
                        cand &= cand - 1u;
                        if (box_d2(boxes + c * 8, cx, cy, cz) <= best.d[3]) scan_cluster(lds, d.Vp, c, cx, cy, cz, best);
                    }
                }
            } else {
                for (int c = lane; c < d.NC; c += 64) {
                    if (c == c_near) continue;
                    const float v = box_d2(boxes + c * 8, cx, cy, cz);
                    if (v <= bound && v <= best.d[3]) scan_cluster(lds, d.Vp, c, cx, cy, cz, best);
                }
            }
            wave_smallest4(best.d, sm, who);
            const int nearest = __shfl(best.i[0], who, 64);
            if (lane == 0) {
                const float d1 = sqrtf(sm[0]), d4 = sqrtf(sm[3]);
                const float reach = d4 + r;
                cap[cell] = (d1 - r >= thr) ? -1.0f : reach * reach * 1.001f;
                cell_seed[(int64_t)b * NCELL + cell] = nearest / CS;      // cluster of the centre's nearest vertex
            }
        }
    }
}

// cell_start = exclusive scan of the live cells' counts, one workgroup per 4,096 cells: it first adds up the live counts of
// every cell in front of its chunk (the arrays are 1 MB per body and sit in L2; a single workgroup walking the 64 chunks one
// after the other took 0.12 ms per call), then scans its own.  cell_fill (the scatter's fill counters) is zeroed here;
// live[b] = number of list entries in live cells, written by the last chunk.
__global__ __launch_bounds__(1024) void warp_cell_scan_kernel(const int32_t* __restrict__ cell_count,
                                                              int32_t* __restrict__ cell_start,
                                                              const float* __restrict__ cell_cap2,
                                                              int32_t* __restrict__ cell_fill,
                                                              int32_t* __restrict__ live, int G) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int32_t* cnt = cell_count + (int64_t)blockIdx.y * NCELL;
    const float* cap = cell_cap2 + (int64_t)blockIdx.y * NCELL;
    int32_t* start = cell_start + (int64_t)blockIdx.y * NCELL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto live_counts = [&](int base) {                   // 4 consecutive cells per thread; dead cells take no room
        int4 v = reinterpret_cast<const int4*>(cnt + base)[threadIdx.x];
        const float4 cp = reinterpret_cast<const float4*>(cap + base)[threadIdx.x];
        if (v.x > 0 && cp.x < 0.f) v.x = 0;
        if (v.y > 0 && cp.y < 0.f) v.y = 0;
        if (v.z > 0 && cp.z < 0.f) v.z = 0;
        if (v.w > 0 && cp.w < 0.f) v.w = 0;
        return v;
    };
    const int my_base = (int)blockIdx.x * 4096;
    int before = 0;
#pragma unroll 4
    for (int base = 0; base < my_base; base += 4096) {
        const int4 v = live_counts(base);
        before += v.x + v.y + v.z + v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) wave_tot[wave] = before;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_tot[w];
        carry = t;
    }
    __syncthreads();
    const int4 v = live_counts(my_base);
    reinterpret_cast<int4*>(cell_fill + (int64_t)blockIdx.y * NCELL + my_base)[threadIdx.x] = make_int4(0, 0, 0, 0);
    const int mine = v.x + v.y + v.z + v.w;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();                                     // (wave_tot is reused)
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    const int ex = off + incl - mine;
    reinterpret_cast<int4*>(start + my_base)[threadIdx.x] = make_int4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
    if (my_base + 4096 >= G * G * G && threadIdx.x == 1023) live[blockIdx.y] = off + incl;
}

// counting-sort scatter of the near list by cell: a workgroup ranks 4096 consecutive list entries per cell in its
// LDS hash table, reserves one range per distinct cell with a single global atomic, and places the entries
template <int T, int E>
__global__ __launch_bounds__(T) void warp_cell_scatter_kernel(const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ cells,
                                                              const int32_t* __restrict__ count, int64_t N,
                                                              const int32_t* __restrict__ cell_start,
                                                              int32_t* __restrict__ cell_fill,
                                                              const float* __restrict__ cell_cap2,
                                                              int32_t* __restrict__ sorted) {
    __shared__ int hkeys[HN], hcnt[HN];
    const int b = blockIdx.y;
    const int cnt = count[b];
    const int32_t* my_cells = cells + (int64_t)b * N;
    const int32_t* my_list = list + (int64_t)b * N;
    const int32_t* start = cell_start + (int64_t)b * NCELL;
    int32_t* fill = cell_fill + (int64_t)b * NCELL;
    const float* cap = cell_cap2 + (int64_t)b * NCELL;
    for (int base = blockIdx.x * E * T; base < cnt; base += gridDim.x * E * T) {
        // (every load of a trip first: the entries' cells and the entries themselves, then the cells' dead flags)
        int cell[E], entry[E], slot[E], rank[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int i = base + j * T + threadIdx.x;
            cell[j] = i < cnt ? my_cells[i] : -1;
            entry[j] = i < cnt ? my_list[i] : 0;
        }
        for (int s = threadIdx.x; s < HN; s += T) { hkeys[s] = -1; hcnt[s] = 0; }
        float cp[E];
#pragma unroll
        for (int j = 0; j < E; ++j) cp[j] = cell[j] >= 0 ? cap[cell[j]] : -1.0f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            slot[j] = -1; rank[j] = 0;
            if (cp[j] < 0.f) cell[j] = -1;                  // entries of dead cells are dropped here
            if (cell[j] >= 0) {
                slot[j] = hash_slot(hkeys, cell[j]);
                if (slot[j] >= 0) rank[j] = atomicAdd(&hcnt[slot[j]], 1);
            }
        }
        __syncthreads();
        {
            // count -> base position, one global atomic per distinct cell of the trip (all of a thread's slots in flight together)
            constexpr int SL = HN / T;
            int key[SL], st[SL], got[SL];
#pragma unroll
            for (int q = 0; q < SL; ++q) {
                key[q] = hkeys[q * T + threadIdx.x];
                st[q] = key[q] >= 0 ? start[key[q]] : 0;
                got[q] = key[q] >= 0 ? atomicAdd(fill + key[q], hcnt[q * T + threadIdx.x]) : 0;
            }
#pragma unroll
            for (int q = 0; q < SL; ++q)
                if (key[q] >= 0) hcnt[q * T + threadIdx.x] = st[q] + got[q];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (cell[j] >= 0) {
                const int pos = slot[j] >= 0 ? hcnt[slot[j]] + rank[j] : start[cell[j]] + atomicAdd(fill + cell[j], 1);
                sorted[(int64_t)b * N + pos] = entry[j];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(WARP_THREADS) void warp_search_kernel(
    const float* __restrict__ index, IndexDims d, const float* __restrict__ ober2cano, const float* __restrict__ lbs_w,
    int J, int64_t N, float thr, float4* __restrict__ pts_out, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_w,
    const int32_t* __restrict__ list, const int32_t* __restrict__ count, int32_t* __restrict__ cursor,
    const float* __restrict__ cell_cap2, uint8_t* __restrict__ valid_mask, const int32_t* __restrict__ cell_seed, int G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int cnt = count[b];
    if ((int)blockIdx.x * (WARP_THREADS / 64) * 8 >= cnt) return;        // nothing left for this workgroup
    const int lpi = lanes_per_item(cnt, (int)gridDim.x * (WARP_THREADS / 64));      // (see warp_cells_kernel)
    const int n_items = (cnt + lpi - 1) / lpi;
    const float* my_index = index + (int64_t)b * d.total_floats();
    const int32_t* order = reinterpret_cast<const int32_t*>(my_index + d.order_off());
    const float* O2C = ober2cano + (int64_t)b * d.V * 16;
    const int32_t* my_list = list + (int64_t)b * N;
    // DIRECT (cell_cap2 == NULL): no cell pass ran — a small batch has about as many occupied cells as near samples, so a
    // search per cell buys nothing per sample.  The list is the classify pass's own (sample order: the lanes of an item are
    // consecutive samples of a ray), and each sample searches inside the validity radius first: nothing there -> it is
    // invalid (the blended distance is a convex combination of neighbour distances >= dis_threshold) and stays (x, 0); four
    // or more -> those ARE the exact neighbours; one to three -> the search is repeated without a bound.
    const bool direct = cell_cap2 == nullptr;
    const float* cap = direct ? nullptr : cell_cap2 + (int64_t)b * NCELL;
    const int lane = threadIdx.x & 63;
    stage_index(my_index, d.lds_floats(), lds);
    const float* gbox = lds + d.body_off();
    ItemDealer deal(n_items, (int)gridDim.x * (WARP_THREADS / 64), (int)(threadIdx.x >> 6) * (int)gridDim.x + (int)blockIdx.x,
                    cursor + (int64_t)b * DEAL_INTS);
    for (;;) {
        const int item = deal.next(lane);
        if (item < 0) break;
        const int i = item * lpi + lane;
        const bool go = lane < lpi && i < cnt;
        const int64_t o = (int64_t)b * N + (go ? my_list[i] : 0);
        const float4 p = go ? pts_out[o] : make_float4(0.f, 0.f, 0.f, 0.f);
        // inside the cell's radius the exact four neighbours are guaranteed to be found (warp_cells_kernel)
        // ... and the first cluster to scan is the one the cell's own search found nearest: no descent per point
        Best4 best;
        if (direct) {
            best_init(best, thr * thr * 1.0002f);
            search(lds, d, p.x, p.y, p.z, go, best);
            const bool partial = go && best.i[0] >= 0 && best.i[3] < 0;
            if (__any(partial)) {
                Best4 full;
                best_init(full);
                search(lds, d, p.x, p.y, p.z, partial, full);
                if (partial) best = full;
            }
            if (!go || best.i[0] < 0) continue;
        } else {
            const int cell = cell_of(gbox, thr, G, p.x, p.y, p.z);
            best_init(best, cap[cell]);
            search_near(lds, d, p.x, p.y, p.z, go, best, go ? cell_seed[(int64_t)b * NCELL + cell] : 0);
            if (!go) continue;
        }
        const bool ok = blend_and_store(best, order, lbs_w, J, O2C, thr, p.x, p.y, p.z, o, pts_out, nullptr, nullptr, nullptr,
                                        nbr_idx, nbr_w);
        if (valid_mask != nullptr && ok) valid_mask[o] = 1;     // lean mode: the byte the compositor and the MLP's list go by
    }
}

// ---------------------------------------------------------------------------------------------
// SMALL BATCHES (training: 1,024 rays per body): EIGHT LANES PER SAMPLE.
// A lane per sample makes a wavefront walk the union of what its samples need, and the near samples of a training batch are
// ~10^2 times sparser than a frame's: 32 consecutive list entries span a metre of a ray and want different parts of the body
// (0.68 ms per call for 4 x 10^5 near samples, most of the time spent in clusters that one or two lanes asked for).  The
// tree has fan-out 8 at every level, so here the 8 lanes of a GROUP take the 8 children of one node — 8 box tests or 8 vertex
// distances per instruction — and the 8 groups of a wavefront each walk their OWN sample's tree: every step a group either
// scans the nearest pending cluster, or opens the nearest pending super-cluster / top (nearest first: the bound is tight
// after the first cluster).  What is pending lives in the lanes (the box distance a lane computed for its child, +inf once
// taken); the choice is a DPP minimum over the group of (distance bits | lane), no LDS traffic, no divergence between groups
// other than in the number of steps.  The 4 best are kept REPLICATED in the group's lanes: candidates of a cluster are
// inserted in ascending order, one per round, while any still beats the 4th (4 rounds for the first cluster, ~0 later).
// Groups are persistent: a finished group takes the next sample from a small pool the wavefront prefetched (index and
// point), and the blend of a finished sample is queued in LDS until 64 of them can run with every lane busy.
// Same results as the lane-per-sample search bit for bit: dist2(), best_insert()'s (distance, slot) order, and the same
// bounded-then-unbounded rule for samples with 1-3 vertices inside the validity radius.
constexpr unsigned PICK_NONE = 0xffffffffu;
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// minimum over the 8 lanes of a group (lanes 8g..8g+7), in every lane of the group
__device__ __forceinline__ unsigned group8_min(unsigned v) {
    // three v_min_u32 with the DPP modifier on one operand (through the builtin hipcc emits mov + mov_dpp + min per step);
    // s_nop 1: the two wait states between a VALU write and a DPP read of the same register
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
// sort key of a pending child: its (non-negative) box distance with the low bits replaced by who holds it
__device__ __forceinline__ unsigned pick_key(float dist, float bound, unsigned tag, unsigned tag_mask) {
    return dist <= bound ? ((__float_as_uint(dist) & ~tag_mask) | tag) : PICK_NONE;
}
#ifndef ANR_GROUP_SCAN
#define ANR_GROUP_SCAN 2
#endif
constexpr int GROUP_SCAN = ANR_GROUP_SCAN;     // clusters a group scans per step (see the scan below): 1 -> 2: 0.147 -> 0.126-0.130 ms
                                               // per call at 2 bodies; 3: the same; 4: 0.132 (profiles/r05/ab_group_scan.txt)
constexpr int GQ_ENTRIES = 64;                 // blend queue per wavefront: {sample, slots 0|1, slots 2|3}
constexpr int GQ_BYTES = GQ_ENTRIES * 12;

#ifdef ANR_SEARCH_PROF
// experiment builds only (tools/exp/search_prof.py): per wavefront {clocks total, in trips to the cursors, in index staging, trips,
// traversal steps, samples started, unbounded repeats, flushes}
__device__ long long anr_search_prof[8192 * 8];
__device__ long long anr_search_events[64 * 256];          // wavefronts 0..63 of the launch: up to 128 {tag, time} pairs each
#define PROF_T0(v) const long long v = wall_clock64()
#define PROF_ADD(slot, v) prof[slot] += (v)
#define PROF_EV(tag) do { if (ev_w >= 0 && ev_n < 127 && lane == 0) { anr_search_events[ev_w * 256 + 2 * ev_n] = (tag); \
    anr_search_events[ev_w * 256 + 2 * ev_n + 1] = wall_clock64(); } ++ev_n; } while (0)
#else
#define PROF_T0(v)
#define PROF_ADD(slot, v)
#define PROF_EV(tag)
#endif
__global__ __launch_bounds__(WARP_THREADS) void warp_search_groups_kernel(
    const float* __restrict__ index, IndexDims d, const float* __restrict__ ober2cano, const float* __restrict__ lbs_w,
    int J, int64_t N, float thr, float4* __restrict__ pts_out, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_w,
    const int32_t* __restrict__ list, const int32_t* __restrict__ count, uint8_t* __restrict__ valid_mask) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 7, gbase = lane & 56;
#ifdef ANR_SEARCH_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long prof_begin = wall_clock64(), prof_cyc0 = clock64();
    int ev_n = 0;
    const int ev_gw = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * (WARP_THREADS / 64) + wave;
    const int ev_w = ev_gw < 64 ? ev_gw : -1;
    PROF_EV(1);
#endif
    // The body's list is dealt out STATICALLY and entry by entry: wavefront w of the body's W wavefronts takes the entries w,
    // w + W, w + 2 W, ... — neighbouring entries (one stretch of a ray: all deep inside the body, or all in the empty space around
    // it) go to different wavefronts, which balances the deep searches against the trivial ones without a single atomic (dealt
    // out in blocks of 8 consecutive entries the slowest wavefront took 5 x the median).  (Round 3 handed blocks out through 8 cursors per body and let a finished workgroup help the other bodies: at 16
    // wavefronts per workgroup that is ~45,000 atomic adds on 16 to 128 addresses per call — most of them the probes that find a
    // segment exhausted — and a same-address atomic retires in ~0.1 us: the kernel took 0.33 ms whether the batch held 2 bodies
    // or 16, i.e. however little there was to search.  Per-wavefront event log: tools/exp/search_prof.py.)
    const int b = (int)blockIdx.y, hop = 0;
    (void)hop;
    const int cnt = count[b];
    const int n_waves = (int)gridDim.x * (WARP_THREADS / 64), my_wave = (int)blockIdx.x * (WARP_THREADS / 64) + wave;
    if ((int)blockIdx.x * (WARP_THREADS / 64) >= cnt) return;            // not even one entry for this workgroup
    const float* my_index = index + (int64_t)b * d.total_floats();
    const int32_t* order = reinterpret_cast<const int32_t*>(my_index + d.order_off());
    const float* O2C = ober2cano + (int64_t)b * d.V * 16;
    const int32_t* my_list = list + (int64_t)b * N;
    PROF_EV(20 + hop);
    stage_index(my_index, d.lds_floats(), lds);
    PROF_EV(2);
    const float* boxes = lds + d.box_off();
    const float* sboxes = lds + d.sbox_off();
    const float* tboxes = lds + d.tbox_off();
    int32_t* queue = reinterpret_cast<int32_t*>(lds + d.lds_floats()) + wave * (GQ_BYTES / 4);
    const float INF = __builtin_inff();
    const float cap2 = thr * thr * 1.0002f;

    // blend the queued samples, one per lane
    int q_n = 0;
    auto flush = [&]() {

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < q_n) {
            const int32_t smp = queue[lane];
            const unsigned s01 = (unsigned)queue[GQ_ENTRIES + lane], s23 = (unsigned)queue[2 * GQ_ENTRIES + lane];
            const int64_t o = (int64_t)b * N + smp;
            const float4 p = pts_out[o];
            Best4 r;
            r.i[0] = (int)(s01 & 0xffffu); r.i[1] = (int)(s01 >> 16); r.i[2] = (int)(s23 & 0xffffu); r.i[3] = (int)(s23 >> 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.d[k] = dist2(p.x, p.y, p.z, lds[r.i[k]], lds[d.Vp + r.i[k]], lds[2 * d.Vp + r.i[k]]);
            const bool ok = blend_and_store(r, order, lbs_w, J, O2C, thr, p.x, p.y, p.z, o, pts_out, nullptr, nullptr, nullptr,
                                            nbr_idx, nbr_w);
            if (valid_mask != nullptr && ok) valid_mask[o] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        q_n = 0;
    };

    // the pool: up to 64 list entries and their points, one per lane
    int trip = 0;
    int pool_smp = 0, pool_n = 0, pool_next = 0;
    float pool_x = 0.f, pool_y = 0.f, pool_z = 0.f;
    bool list_done = false;

    // group state (the same in the 8 lanes of a group, except the pending distances)
    bool idle = true;                          // no sample
    bool second = false;                       // the unbounded repeat
    int smp = 0, s_base = 0, c_base = 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    float td[4] = {INF, INF, INF, INF}, sd = INF, cd = INF;
    Best4 best;
    best_init(best, cap2);

    auto open_tops = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = j + 8 * r;
            td[r] = t < d.NT ? box_d2(tboxes + min(t, d.NT - 1) * 8, px, py, pz) : INF;
        }
        sd = INF; cd = INF;
    };

    for (;;) {
        // ---- hand samples to the idle groups (wave-uniform control flow: every lane takes part in the permutes)
        unsigned long long idle_groups = __ballot(idle && j == 0);
        while (idle_groups) {
            PROF_ADD(3, 1);
            if (pool_next >= pool_n) {
                if (list_done) break;
                PROF_T0(t_trip);
                PROF_EV(3);
                // entries my_wave + k n_waves, k = 64 trip + lane: consecutive list entries (a stretch of one ray: similar cost) go
                // to different wavefronts
                const int first = my_wave + trip * 64 * n_waves;
                if (first >= cnt) { list_done = true; break; }
                ++trip;
                pool_n = min(64, (cnt - first + n_waves - 1) / n_waves);
                pool_next = 0;
                if (lane < pool_n) {
                    pool_smp = my_list[first + lane * n_waves];
                    const float4 p = pts_out[(int64_t)b * N + pool_smp];
                    pool_x = p.x; pool_y = p.y; pool_z = p.z;
                }
#ifdef ANR_SEARCH_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                prof[1] += wall_clock64() - t_trip;
                PROF_EV(4);
#endif
            }
            const int rank = __popcll(idle_groups & ((1ull << gbase) - 1ull));
            const int src = pool_next + rank;
            const bool take = idle && src < pool_n;
            const int addr = min(src, 63) * 4;
            const int t_smp = __builtin_amdgcn_ds_bpermute(addr, pool_smp);
            const float t_x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(pool_x)));
            const float t_y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(pool_y)));
            const float t_z = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(pool_z)));
            pool_next = min(pool_n, pool_next + __popcll(idle_groups));
            if (take) {
                smp = t_smp; px = t_x; py = t_y; pz = t_z;
                idle = false; second = false;

                best_init(best, cap2);
                open_tops();
            }
            idle_groups = __ballot(idle && j == 0);
        }
        PROF_ADD(6, 1);
        if (!__any(!idle)) break;
        PROF_EV(5);

        // ---- one step of every group's traversal
        bool done = idle;
        if (!idle) {
            unsigned cm = group8_min(pick_key(cd, best.d[3], j, 7));
            if (cm == PICK_NONE) {                               // no cluster pending: open the nearest pending super-cluster
                unsigned sm = group8_min(pick_key(sd, best.d[3], j, 7));
                if (sm == PICK_NONE) {                           // ... after opening the nearest pending top
                    unsigned tk = PICK_NONE;
#pragma unroll
                    for (int r = 0; r < 4; ++r) tk = min(tk, pick_key(td[r], best.d[3], j + 8 * r, 31));
                    const unsigned tm = group8_min(tk);
                    if (tm == PICK_NONE) {
                        done = true;
                    } else {
                        const int t = (int)(tm & 31u);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (t == j + 8 * r) td[r] = INF;
                        s_base = t * TC;
                        const int q = s_base + j;
                        sd = q < d.NS ? box_d2(sboxes + min(q, d.NS - 1) * 8, px, py, pz) : INF;
                        sm = group8_min(pick_key(sd, best.d[3], j, 7));
                    }
                }
                if (sm != PICK_NONE) {
                    const int k = (int)(sm & 7u);
                    if (j == k) sd = INF;
                    c_base = (s_base + k) * SC;
                    const int c = c_base + j;
                    cd = c < d.NC ? box_d2(boxes + min(c, d.NC - 1) * 8, px, py, pz) : INF;
                    cm = group8_min(pick_key(cd, best.d[3], j, 7));
                }
            }
            if (cm != PICK_NONE) {                               // scan the nearest pending cluster: a vertex per lane
                const int k = (int)(cm & 7u);
                if (j == k) cd = INF;
                const int slot0 = (c_base + k) * CS;
                // ... and the next pending one of the same super-cluster with it (round 5): a launch lasts as long as its longest
                // sample chain — 100-180 steps for a sample whose 4th-neighbour sphere grazes that many clusters — and a step's
                // fixed part (the picks, the hand-out and finish votes) is most of a step that inserts nothing.  The second
                // clusters are picked against the bound BEFORE the first one's scan: at worst eight distances each for nothing.
                // The (distance, slot) order of best_insert makes the result independent of the order of the candidates.
                int slots[GROUP_SCAN];
                float d2s[GROUP_SCAN];
                unsigned pend = 0u;                               // bit c: this lane's vertex of cluster c still beats the 4th
                slots[0] = slot0;
#pragma unroll
                for (int c = 1; c < GROUP_SCAN; ++c) {
                    const unsigned cmx = group8_min(pick_key(cd, best.d[3], j, 7));
                    const bool more = cmx != PICK_NONE;
                    const int kx = (int)(cmx & 7u);
                    if (more && j == kx) cd = INF;
                    slots[c] = more ? (c_base + kx) * CS : -1;
                }
#pragma unroll
                for (int c = 0; c < GROUP_SCAN; ++c) {
                    const int sl = slots[c] < 0 ? slot0 : slots[c];
                    d2s[c] = dist2(px, py, pz, lds[sl + j], lds[d.Vp + sl + j], lds[2 * d.Vp + sl + j]);
                    // (<=: a tie with the current 4th may carry a lower slot)
                    pend |= (slots[c] >= 0 && d2s[c] <= best.d[3]) ? 1u << c : 0u;
                }
                while (__any(pend != 0u)) {
                    unsigned u = PICK_NONE;
#pragma unroll
                    for (int c = 0; c < GROUP_SCAN; ++c) u = min(u, ((pend >> c) & 1u) ? __float_as_uint(d2s[c]) : PICK_NONE);
                    const unsigned m = group8_min(u);
                    int sv = 0x7fffffff, cv = 0;                  // equal distances in several clusters: the lowest slot first
#pragma unroll
                    for (int c = 0; c < GROUP_SCAN; ++c) {
                        const unsigned long long bal = __ballot(((pend >> c) & 1u) && __float_as_uint(d2s[c]) == m);
                        const unsigned mine = (unsigned)(bal >> gbase) & 0xffu;
                        const int sc = mine ? slots[c] + __builtin_ctz(mine) : 0x7fffffff;
                        if (sc < sv) { sv = sc; cv = c; }
                    }
                    if (sv != 0x7fffffff) {
                        best_insert(best, __uint_as_float(m), sv);
#pragma unroll
                        for (int c = 0; c < GROUP_SCAN; ++c)
                            if (c == cv && j == sv - slots[c]) pend &= ~(1u << c);
                    }
#pragma unroll
                    for (int c = 0; c < GROUP_SCAN; ++c)
                        if (!(d2s[c] <= best.d[3])) pend &= ~(1u << c);
                }
            }
        }

        // ---- finished groups: repeat unbounded (1-3 vertices inside the radius), or queue the blend; then idle
        if (__any(done && !idle)) {
            const bool fin = done && !idle;
            const bool partial = fin && !second && best.i[0] >= 0 && best.i[3] < 0;
            if (partial) {
                second = true;

                best_init(best);
                open_tops();
            }
            const bool live = fin && !partial && best.i[0] >= 0;
            const unsigned long long live_groups = __ballot(live && j == 0);
            if (q_n + __popcll(live_groups) > GQ_ENTRIES) flush();
            if (live && j == 0) {
                const int at = q_n + __popcll(live_groups & ((1ull << gbase) - 1ull));
                queue[at] = smp;
                queue[GQ_ENTRIES + at] = (int32_t)((unsigned)best.i[0] | ((unsigned)best.i[1] << 16));
                queue[2 * GQ_ENTRIES + at] = (int32_t)((unsigned)best.i[2] | ((unsigned)best.i[3] << 16));
            }
            q_n += __popcll(live_groups);
            if (fin && !partial) idle = true;
        }
    }
    PROF_EV(6);
    flush();
    PROF_EV(7);
#ifdef ANR_SEARCH_PROF
    prof[0] = wall_clock64() - prof_begin;
    prof[7] = clock64() - prof_cyc0;                          // shader cycles over the same interval: the clock the kernel ran at
    {   // per-lane counters (samples, repeats) summed over the wavefront; the rest is wave-uniform
        long long s5 = prof[5], s6 = prof[6];
        const int w = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * (WARP_THREADS / 64) + wave;
        if (lane == 0 && w < 8192) {
            long long* o = anr_search_prof + (long long)w * 8;
            o[0] = prof[0]; o[1] = prof[1]; o[2] = prof[2]; o[3] = prof[3]; o[4] = prof[4]; o[5] = s5; o[6] = s6; o[7] = prof[7];
        }
    }
#endif
}

#ifdef ANR_SEARCH_PROF
extern "C" int anr_search_prof_read(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(anr_search_prof), sizeof(long long) * 8192 * 8);
}
extern "C" int anr_walk_prof_read(unsigned long long* host_out) {                  // reads the counters and clears them
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(anr_walk_prof), sizeof(unsigned long long) * 8);
    if (e != hipSuccess) return (int)e;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(anr_walk_prof), zero, sizeof zero);
}
extern "C" int anr_search_events_read(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(anr_search_events), sizeof(long long) * 64 * 256);
}
#endif

// lean mode: validity bytes -> list of the valid samples' flat positions for anr_mlp_forward_indexed, in sample order inside
// every 64-KB span of the byte array (the MLP's gather of points and scatter of results then walk memory the way the rays
// were laid out).  ONE TRIP TO THE COUNTER PER 64 KB: a workgroup's four wavefronts each hold 16 one-KB pieces in registers
// (16 bytes per lane and piece, all sixteen loads in flight together), scan the lanes' counts of the pieces that hold anything
// (most of a frame's pieces end at the ballot), add up, and thread 0 reserves the workgroup's range with one atomic; then every
// non-empty piece goes through the wave's own 4 KB of LDS — positions written in order, copied out 64 consecutive dwords per
// store.  A same-address returning atomic retires in ~10 ns on this part, whoever issues it: round 5's form (a 1,024-thread
// workgroup, 4 bytes per thread, two barriers around thread 0's atomic per 4 KB) took 63 us per call on a configs[2] frame, a
// wavefront per 1-KB piece with its own atomic and no barrier at all 175 us, this one 37 us (profiles/r06/ab_valid_list.txt).
// (the last 16 bytes of an array whose length is not a multiple of 16, or an unaligned array: byte by byte, out of line)
__device__ __noinline__ uint4 bytes16_guarded(const uint8_t* __restrict__ mask, int64_t i0, int64_t n) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int k = 0; k < 16; ++k)
        if (i0 + k < n) w[k >> 2] |= (unsigned)mask[i0 + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
constexpr int VL_THREADS = 256;
constexpr int VL_PIECES = 16;                                  // 1-KB pieces per wavefront: a workgroup takes 64 KB of the byte array
constexpr int VL_SPAN = VL_PIECES * 1024 * (VL_THREADS / 64);
__global__ __launch_bounds__(VL_THREADS) void warp_valid_list_kernel(const uint8_t* __restrict__ mask, int64_t n,
                                                                     int32_t* __restrict__ valid_index,
                                                                     int32_t* __restrict__ valid_count) {
    __shared__ int32_t piece[VL_THREADS / 64][1024];
    __shared__ int wave_tot[VL_THREADS / 64];
    __shared__ int block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* mine_lds = piece[wave];
    const bool aligned = (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
    const int64_t w0 = (int64_t)blockIdx.x * VL_SPAN + (int64_t)wave * (VL_PIECES * 1024);
    uint4 v[VL_PIECES];
#pragma unroll
    for (int p = 0; p < VL_PIECES; ++p) {
        const int64_t i0 = w0 + p * 1024 + lane * 16;
        v[p] = make_uint4(0u, 0u, 0u, 0u);
        if (aligned && i0 + 15 < n) v[p] = *reinterpret_cast<const uint4*>(mask + i0);
        else if (i0 < n) v[p] = bytes16_guarded(mask, i0, n);
    }
    int before[VL_PIECES];                                     // entries of the piece in front of this lane's 16 bytes
    int tot[VL_PIECES];                                        // (wave-uniform) entries of the piece
    int wave_sum = 0;
#pragma unroll
    for (int p = 0; p < VL_PIECES; ++p) {
        v[p].x &= 0x01010101u; v[p].y &= 0x01010101u; v[p].z &= 0x01010101u; v[p].w &= 0x01010101u;
        const int mine = __popc(v[p].x) + __popc(v[p].y) + __popc(v[p].z) + __popc(v[p].w);
        before[p] = 0; tot[p] = 0;
        if (__any(mine != 0)) {
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            before[p] = incl - mine;
            tot[p] = __builtin_amdgcn_readlane(incl, 63);
            wave_sum += tot[p];
        }
    }
    if (lane == 0) wave_tot[wave] = wave_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < VL_THREADS / 64; ++w) { const int c = wave_tot[w]; wave_tot[w] = t; t += c; }
        block_base = t ? atomicAdd(valid_count, t) : 0;
    }
    __syncthreads();
    if (wave_sum == 0) return;
    int base = block_base + wave_tot[wave];
#pragma unroll
    for (int p = 0; p < VL_PIECES; ++p) {
        if (tot[p] == 0) continue;
        const int64_t i0 = w0 + p * 1024 + lane * 16;
        int pos = before[p];
        const unsigned w[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if ((w[k >> 2] >> (8 * (k & 3))) & 1u) mine_lds[pos++] = (int32_t)(i0 + k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < tot[p]; e += 64) valid_index[base + e] = mine_lds[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (the next piece overwrites what was just read)
        __builtin_amdgcn_wave_barrier();
        base += tot[p];
    }
}

// ---------------------------------------------------------------------------------------------
// reference: models/anim_nerf.py:157-163 (KNN_CUDA call / in-repo fallback definition)
__global__ __launch_bounds__(WARP_THREADS) void knn_kernel(const float* __restrict__ index, IndexDims d,
                                                           const float* __restrict__ xyz, int64_t N,
                                                           float* __restrict__ dist_out,
                                                           int64_t* __restrict__ idx_out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const float* my_index = index + (int64_t)b * d.total_floats();
    stage_index(my_index, d.lds_floats(), lds);
    const int32_t* order = reinterpret_cast<const int32_t*>(my_index + d.order_off());
    for (int it = 0; it < iters; ++it) {
        int64_t n = ((int64_t)blockIdx.x * iters + it) * WARP_THREADS + threadIdx.x;
        const bool active = n < N;
        if (!active) n = N - 1;
        const float* s = xyz + ((int64_t)b * N + n) * 3;
        const float px = s[0], py = s[1], pz = s[2];
        if (!__any(active)) continue;
        Best4 best;
        best_init(best);
        search(lds, d, px, py, pz, active, best);
        if (!active) continue;
        const int64_t o = ((int64_t)b * N + n) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dist_out[o + k] = sqrtf(best.d[k]); idx_out[o + k] = order[best.i[k]]; }
    }
}

// d1[n] = distance of xyz[n] to its nearest vertex if that is below `radius`, +inf otherwise: the same search started from
// the bound radius^2 — a query far from the body is settled by the 14 top boxes (mesh extraction's empty-cell test, sigma_grid)
__global__ __launch_bounds__(WARP_THREADS) void knn_within_kernel(const float* __restrict__ index, IndexDims d,
                                                                  const float* __restrict__ xyz, int64_t N, float radius,
                                                                  float* __restrict__ d1_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    stage_index(index + (int64_t)b * d.total_floats(), d.lds_floats(), lds);
    int64_t n = (int64_t)blockIdx.x * WARP_THREADS + threadIdx.x;
    const bool active = n < N;
    if (!active) n = N - 1;
    const float* s = xyz + ((int64_t)b * N + n) * 3;
    Best4 best;
    best_init(best, radius * radius * 1.0002f);
    search(lds, d, s[0], s[1], s[2], active, best);
    if (active) d1_out[(int64_t)b * N + n] = best.i[0] >= 0 ? sqrtf(best.d[0]) : __builtin_inff();
}

// ---------------------------------------------------------------------------------------------
// Backward of the warp w.r.t. its differentiable inputs (a16, pose refinement): x_c = T_b [x,1], T_b = sum_k w_k M_k,
// x = o' + z d'.  The neighbour ids and weights are constants (KNN is no_grad in the reference, anim_nerf.py:158).
//   dM_{v_k}[r][c] += w_k * dxc[r] * [x,1][c]   (atomics: many samples share a vertex)
//   dx = R_b^T dxc  ->  d o' += dx, d d' += z dx (atomics per ray), dz = dx . d'
// dM: the samples of a workgroup (1,024 consecutive ones: ten to sixteen neighbouring rays) share their neighbour vertices —
// ~300 live samples x 4 neighbours land on ~150 vertices — so the 12 entries per (sample, neighbour) are added into an LDS
// table keyed by vertex (ds_add_f32, no return) and each distinct vertex goes to global memory once per workgroup: 48 global
// float atomics per live sample became ~6.  A vertex that finds no slot within 16 probes takes the direct route.
constexpr int WB_THREADS = 1024;
constexpr int WB_SLOTS = 1024;
__device__ __forceinline__ int wb_slot(int* keys, int v) {
    unsigned s = ((unsigned)v * 2654435761u) >> 22;
#pragma unroll 1
    for (int t = 0; t < 16; ++t) {
        const int prev = atomicCAS(&keys[s], -1, v);
        if (prev == -1 || prev == v) return (int)s;
        s = (s + 1) & (WB_SLOTS - 1);
    }
    return -1;
}

__global__ __launch_bounds__(WB_THREADS) void warp_backward_kernel(
    const float4* __restrict__ d_pts, const float* __restrict__ rays, int ray_stride, const float* __restrict__ z, int K,
    const float* __restrict__ ober2cano, const int4* __restrict__ nbr_idx, const float4* __restrict__ nbr_w, int V,
    int64_t N, float* __restrict__ d_o2c, float* __restrict__ d_rays, float* __restrict__ d_z, const int32_t* __restrict__ pos) {
    __shared__ int hkey[WB_SLOTS];
    __shared__ float hval[WB_SLOTS][12];
    for (int i = threadIdx.x; i < WB_SLOTS; i += WB_THREADS) {
        hkey[i] = -1;
#pragma unroll
        for (int e = 0; e < 12; ++e) hval[i][e] = 0.f;
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = n < N;
    const int64_t o = (int64_t)b * N + (in_range ? n : 0);
    // pos (anr_warp_backward_compact): d_pts holds the rows of the COMPACTED list of valid samples; pos[sample] = its row or -1
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_range) {
        if (pos) { const int r = pos[o]; if (r >= 0) g = d_pts[r]; }
        else g = d_pts[o];
    }
    const float4 w4 = in_range ? nbr_w[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
    const bool live = (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f) && (g.x != 0.f || g.y != 0.f || g.z != 0.f);
    const int64_t R = N / K;
    const int64_t ray = (in_range ? n : N - 1) / K;
    float dzv = 0.f;
    float dray[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* ry = rays + ((int64_t)b * R + ray) * ray_stride;
        const float zz = z[o];
        const float x[4] = {ry[0] + zz * ry[3], ry[1] + zz * ry[4], ry[2] + zz * ry[5], 1.0f};
        const int4 i4 = nbr_idx[o];
        const int vid[4] = {i4.x, i4.y, i4.z, i4.w};
        const float gv[3] = {g.x, g.y, g.z};
        float Rb[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (w[k] == 0.f) continue;
            const float* M = ober2cano + ((int64_t)b * V + vid[k]) * 16;
            const int slot = wb_slot(hkey, vid[k]);
            float* dM = slot >= 0 ? &hval[slot][0] : d_o2c + ((int64_t)b * V + vid[k]) * 16;
            const int pitch = slot >= 0 ? 4 : 4;                 // (rows of 4 in both places: [3][4] of the table, [4][4] of M)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Rb[r * 3 + c] += w[k] * M[r * 4 + c];
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(dM + r * pitch + c, w[k] * gv[r] * x[c]);
            }
        }
        float dx[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dx[c] = Rb[c] * gv[0] + Rb[3 + c] * gv[1] + Rb[6 + c] * gv[2];
            dray[c] = dx[c];
            dray[3 + c] = zz * dx[c];
        }
        dzv = dx[0] * ry[3] + dx[1] * ry[4] + dx[2] * ry[5];
    }
    // d o', d d': the lanes of a wavefront are consecutive samples — of ONE ray, or of two or three — and a per-lane atomic
    // sends 64 adds to the same six addresses one behind the other.  Segmented sum over the lanes of a ray first (the rays of
    // a wavefront are contiguous runs of lanes), one set of atomics per run.
    const int lane = threadIdx.x & 63;
    if (__any(live)) {
        const int ray_lo = (int)(ray & 0x7fffffff);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int other = __shfl_down(ray_lo, off, 64);
            const bool same = lane + off < 64 && other == ray_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float v = __shfl_down(dray[c], off, 64);
                if (same) dray[c] += v;
            }
        }
        const int prev = __shfl_up(ray_lo, 1, 64);
        const bool head = lane == 0 || prev != ray_lo;
        if (head && in_range) {
            float* dr = d_rays + ((int64_t)b * R + ray) * 8;
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (dray[c] != 0.f) atomicAdd(dr + c, dray[c]);
        }
    }
    if (in_range) d_z[o] = dzv;
    // the table's vertices -> global memory, once each
    __syncthreads();
    for (int i = threadIdx.x; i < WB_SLOTS; i += WB_THREADS) {
        const int v = hkey[i];
        if (v < 0) continue;
        float* dM = d_o2c + ((int64_t)b * V + v) * 16;
#pragma unroll
        for (int e = 0; e < 12; ++e)
            if (hval[i][e] != 0.f) atomicAdd(dM + e, hval[i][e]);
    }
}

template <typename Kern>
int allow_big_lds(Kern k, int bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%d B LDS): %s", who, bytes, hipGetErrorString(e));
    return 0;
}

}  // namespace anr

using namespace anr;

extern "C" int64_t anr_knn_index_bytes(int V) {
    if (V < 4) return ANR_E_BADARG;
    return (int64_t)index_dims(V).total_floats() * 4;
}

extern "C" int anr_knn_index_build(const float* verts, const int32_t* order, int bs, int V, void* index_out,
                                   void* stream) {
    return anr_knn_index_build_reach(verts, order, bs, V, 0.0f, index_out, stream);
}

extern "C" int anr_knn_index_build_reach(const float* verts, const int32_t* order, int bs, int V, float dis_threshold, void* index_out,
                                         void* stream) {
    ANR_REQUIRE(verts && index_out, ANR_E_BADARG, "anr_knn_index_build: null pointer");
    ANR_REQUIRE(bs > 0 && V >= 4 && dis_threshold >= 0.0f, ANR_E_BADARG, "anr_knn_index_build: bs=%d V=%d dis_threshold=%g", bs, V,
                dis_threshold);
    ANR_REQUIRE(((uintptr_t)index_out & 15) == 0, ANR_E_ALIGN, "anr_knn_index_build: index_out must be 16-B aligned");
    IndexDims d = index_dims(V);
    ANR_REQUIRE(d.NC <= MAX_NC, ANR_E_SHAPE, "anr_knn_index_build: V=%d too large (max %d)", V, MAX_NC * CS);
    hipLaunchKernelGGL(knn_index_build_kernel, dim3(dis_threshold > 0.0f ? 2 : 1, bs), dim3(IB_THREADS), 0, (hipStream_t)stream, verts, order, d,
                       reinterpret_cast<float*>(index_out), dis_threshold);
    return check_launch("anr_knn_index_build");
}

extern "C" int64_t anr_warp_ws_ints(int bs, int64_t N) {
    if (bs <= 0 || N <= 0) return ANR_E_BADARG;
    return WarpWs::ints(bs, N);
}

extern "C" int anr_warp_ws_zero_range(int bs, int64_t N, int64_t* first_int_out, int64_t* ints_out) {
    ANR_REQUIRE(bs > 0 && N > 0 && first_int_out && ints_out, ANR_E_BADARG, "anr_warp_ws_zero_range: bs=%d N=%lld", bs, (long long)N);
    *first_int_out = WarpWs::count_off(bs, N);
    *ints_out = WarpWs::zeroed_ints(bs, !(N < (int64_t)1 << 19 && !getenv("ANR_WARP_CELLS_ALWAYS")));
    return 0;
}

extern "C" int anr_warp_points(const float* xyz, int xyz_stride, const float* rays, int ray_stride, const float* z,
                               int K, const void* knn_index, const float* ober2cano, const float* lbs_weights, int bs,
                               int V, int J, int64_t N, float dis_threshold, int skip_far, float* pts_out,
                               float* dist_out, int32_t* idx_out, float* blended_out, int32_t* nbr_idx_out,
                               float* nbr_w_out, int32_t* ws, void* stream) {
    return anr_warp_points_lean(xyz, xyz_stride, rays, ray_stride, z, K, knn_index, ober2cano, lbs_weights, bs, V, J, N,
                                dis_threshold, skip_far, pts_out, dist_out, idx_out, blended_out, nbr_idx_out, nbr_w_out, ws,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int anr_warp_points_lean(const float* xyz, int xyz_stride, const float* rays, int ray_stride, const float* z,
                                    int K, const void* knn_index, const float* ober2cano, const float* lbs_weights, int bs,
                                    int V, int J, int64_t N, float dis_threshold, int skip_far, float* pts_out,
                                    float* dist_out, int32_t* idx_out, float* blended_out, int32_t* nbr_idx_out,
                                    float* nbr_w_out, int32_t* ws, uint8_t* valid_mask_out, int32_t* valid_index_out,
                                    int32_t* valid_count_out, const float* reuse_pts, const uint8_t* reuse_mask,
                                    const uint8_t* reuse_perm, int reuse_K, void* stream) {
    ANR_REQUIRE((reuse_pts != nullptr) == (reuse_mask != nullptr), ANR_E_BADARG,
                "anr_warp_points_lean: reuse_pts / reuse_mask / reuse_perm go together");
    return anr_warp_points_reuse(xyz, xyz_stride, rays, ray_stride, z, K, knn_index, ober2cano, lbs_weights, bs, V, J, N, dis_threshold,
                                 skip_far, pts_out, dist_out, idx_out, blended_out, nbr_idx_out, nbr_w_out, ws, valid_mask_out,
                                 valid_index_out, valid_count_out, reuse_pts, reuse_mask, reuse_perm, reuse_K, nullptr, nullptr, stream);
}

extern "C" int anr_warp_points_reuse(const float* xyz, int xyz_stride, const float* rays, int ray_stride, const float* z,
                                     int K, const void* knn_index, const float* ober2cano, const float* lbs_weights, int bs,
                                     int V, int J, int64_t N, float dis_threshold, int skip_far, float* pts_out,
                                     float* dist_out, int32_t* idx_out, float* blended_out, int32_t* nbr_idx_out,
                                     float* nbr_w_out, int32_t* ws, uint8_t* valid_mask_out, int32_t* valid_index_out,
                                     int32_t* valid_count_out, const float* reuse_pts, const uint8_t* reuse_mask,
                                     const uint8_t* reuse_perm, int reuse_K, const int32_t* reuse_nbr_idx, const float* reuse_nbr_w,
                                     void* stream) {
    return anr_warp_points_cells(xyz, xyz_stride, rays, ray_stride, z, K, knn_index, ober2cano, lbs_weights, bs, V, J, N, dis_threshold,
                                 skip_far, pts_out, dist_out, idx_out, blended_out, nbr_idx_out, nbr_w_out, ws, valid_mask_out,
                                 valid_index_out, valid_count_out, reuse_pts, reuse_mask, reuse_perm, reuse_K, reuse_nbr_idx,
                                 reuse_nbr_w, nullptr, 0, stream);
}

extern "C" int anr_warp_points_cells(const float* xyz, int xyz_stride, const float* rays, int ray_stride, const float* z,
                                     int K, const void* knn_index, const float* ober2cano, const float* lbs_weights, int bs,
                                     int V, int J, int64_t N, float dis_threshold, int skip_far, float* pts_out,
                                     float* dist_out, int32_t* idx_out, float* blended_out, int32_t* nbr_idx_out,
                                     float* nbr_w_out, int32_t* ws, uint8_t* valid_mask_out, int32_t* valid_index_out,
                                     int32_t* valid_count_out, const float* reuse_pts, const uint8_t* reuse_mask,
                                     const uint8_t* reuse_perm, int reuse_K, const int32_t* reuse_nbr_idx, const float* reuse_nbr_w,
                                     const int32_t* prev_ws, int64_t prev_N, void* stream) {
    const bool lean = valid_mask_out != nullptr;
    ANR_REQUIRE(prev_ws == nullptr || (skip_far && ws != nullptr && prev_ws != ws && prev_N > 0), ANR_E_BADARG,
                "anr_warp_points_cells: prev_ws is ANOTHER call's workspace (its N = prev_N > 0) and needs skip_far with a workspace");
    ANR_REQUIRE((reuse_pts != nullptr) == (reuse_perm != nullptr) && (reuse_mask == nullptr || reuse_pts != nullptr),
                ANR_E_BADARG, "anr_warp_points: reuse_pts / reuse_perm (/ reuse_mask) go together");
    ANR_REQUIRE(reuse_pts == nullptr || (skip_far && ws != nullptr && xyz == nullptr && reuse_K > 0 && reuse_K <= K &&
                                         ((uintptr_t)reuse_pts & 15) == 0),
                ANR_E_BADARG, "anr_warp_points: reuse needs skip_far with a workspace, rays mode, 0 < reuse_K <= K");
    // with the validity outputs the coarse call's validity BYTES are copied; without them its whole rows (training)
    ANR_REQUIRE(reuse_pts == nullptr || (lean == (reuse_mask != nullptr)), ANR_E_BADARG,
                "anr_warp_points: reuse_mask goes with the validity outputs, and only with them");
    ANR_REQUIRE((reuse_nbr_idx != nullptr) == (reuse_nbr_w != nullptr) &&
                (reuse_nbr_idx == nullptr || (reuse_pts != nullptr && !lean && nbr_idx_out != nullptr)) &&
                (reuse_pts == nullptr || lean || nbr_idx_out == nullptr || reuse_nbr_idx != nullptr) &&
                (((uintptr_t)reuse_nbr_idx | (uintptr_t)reuse_nbr_w) & 15) == 0,
                ANR_E_BADARG, "anr_warp_points: reuse_nbr_idx / reuse_nbr_w accompany reuse_pts exactly when the neighbour outputs are asked for");
    ANR_REQUIRE((valid_mask_out != nullptr) == (valid_index_out != nullptr) && (valid_mask_out != nullptr) == (valid_count_out != nullptr),
                ANR_E_BADARG, "anr_warp_points_lean: valid_mask_out / valid_index_out / valid_count_out go together");
    ANR_REQUIRE(!lean || (skip_far && ws != nullptr), ANR_E_BADARG, "anr_warp_points_lean: the validity outputs need skip_far and ws");
    ANR_REQUIRE(!lean || (int64_t)bs * N < (int64_t)1 << 31, ANR_E_BADARG, "anr_warp_points_lean: bs*N does not fit int32");
    ANR_REQUIRE(knn_index && ober2cano && lbs_weights && pts_out, ANR_E_BADARG, "anr_warp_points: null pointer");
    ANR_REQUIRE((xyz != nullptr) || (rays != nullptr && z != nullptr), ANR_E_BADARG,
                "anr_warp_points: need xyz or (rays, z)");
    ANR_REQUIRE(bs > 0 && V >= 4 && N > 0 && J > 0 && J <= MAX_J, ANR_E_BADARG,
                "anr_warp_points: bs=%d V=%d N=%lld J=%d", bs, V, (long long)N, J);
    ANR_REQUIRE(xyz != nullptr ? xyz_stride >= 3 : (K > 0 && ray_stride >= 8 && N % K == 0), ANR_E_BADARG,
                "anr_warp_points: bad stride/K");
    ANR_REQUIRE((dist_out == nullptr) == (idx_out == nullptr) && (dist_out == nullptr) == (blended_out == nullptr),
                ANR_E_BADARG, "anr_warp_points: debug outputs are all-or-none");
    ANR_REQUIRE((nbr_idx_out == nullptr) == (nbr_w_out == nullptr), ANR_E_BADARG,
                "anr_warp_points: nbr_idx_out and nbr_w_out go together");
    ANR_REQUIRE((((uintptr_t)nbr_idx_out | (uintptr_t)nbr_w_out) & 15) == 0, ANR_E_ALIGN,
                "anr_warp_points: neighbour outputs must be 16-B aligned");
    ANR_REQUIRE((((uintptr_t)pts_out | (uintptr_t)ober2cano | (uintptr_t)knn_index) & 15) == 0, ANR_E_ALIGN,
                "anr_warp_points: pts_out / ober2cano / knn_index must be 16-B aligned");
    IndexDims d = index_dims(V);
    const int bytes = d.lds_floats() * 4;
    ANR_REQUIRE(bytes <= 160 * 1024, ANR_E_SHAPE, "anr_warp_points: V=%d needs %d B of LDS (>160 KiB)", V, bytes);
    hipStream_t st = (hipStream_t)stream;
    const float* index = reinterpret_cast<const float*>(knn_index);
    // skip_far & 4: `z` is the step table s[K] of the deterministic stratified depths (the lean renderer pass only)
    const bool z_steps = (skip_far & 4) != 0;
    ANR_REQUIRE(!z_steps || (lean && xyz == nullptr && ws != nullptr), ANR_E_BADARG,
                "anr_warp_points: skip_far & 4 (z = step table) needs rays mode, a workspace and the validity outputs");
    if (skip_far && ws != nullptr) {
        // two passes: classify + compact, then search the compacted list (see warp_classify_kernel)
        ANR_REQUIRE(dist_out == nullptr, ANR_E_BADARG, "anr_warp_points: debug outputs need skip_far = 0");
        ANR_REQUIRE(N < (int64_t)1 << 31, ANR_E_BADARG, "anr_warp_points: N=%lld does not fit the int32 list", (long long)N);
        WarpWs w(ws, bs, N);
        const int G = grid_for(N);
        const int cells = G * G * G;
        // a small batch per body (training: 1,024 rays) goes straight to the per-sample search (see warp_search_kernel)
        const bool small = N < (int64_t)1 << 19 && !getenv("ANR_WARP_CELLS_ALWAYS");
        // (skip_far & 2: the caller has zeroed the counters — anr_warp_ws_zero_range — e.g. with the other fills of its step)
        if (!(skip_far & 2))
            if (int rc = zero_fill(w.count, sizeof(int32_t) * WarpWs::zeroed_ints(bs, !small), st, "anr_warp_points (zero)")) return rc;
        dim3 g1((unsigned)((N + CLS_ITERS * WARP_THREADS - 1) / (CLS_ITERS * WARP_THREADS)), bs);
        // four consecutive samples per thread where the shapes allow dword / 16-byte accesses (every shipped shape)
        // ... and the pass is a stream over the 4-byte-per-sample arrays: the renderer's lean mode.  With the neighbour outputs
        // (training) every sample also gets 48 bytes of 16-byte rows (point, neighbour ids, blend weights — zeros, or the coarse
        // call's rows): there a thread per sample makes a wavefront's store 1 KB of consecutive rows, where four samples per
        // thread make it 64 rows 64 bytes apart.
        const bool vec4 = xyz == nullptr && K % 4 == 0 && N % 4 == 0 && ((uintptr_t)z & 15) == 0 && ((uintptr_t)valid_mask_out & 3) == 0 &&
                          ((uintptr_t)reuse_perm & 3) == 0 && (nbr_w_out == nullptr || getenv("ANR_WARP_CLASSIFY_VEC4"));
#define ANR_CLASSIFY(FR, V4, CL)                                                                                              \
        hipLaunchKernelGGL((warp_classify_kernel<FR, V4, CL>), g1, dim3(WARP_THREADS), 0, st, xyz, xyz_stride, rays, ray_stride, z, \
                           K, index, d, N, dis_threshold, reinterpret_cast<float4*>(pts_out), nbr_idx_out, nbr_w_out,         \
                           w.list, w.cells, w.count, w.cell_count, valid_mask_out,                                            \
                           reinterpret_cast<const float4*>(reuse_pts), reuse_mask, reuse_perm, reuse_K, G,                    \
                           reinterpret_cast<const int4*>(reuse_nbr_idx), reinterpret_cast<const float4*>(reuse_nbr_w))
        if (vec4 && valid_mask_out != nullptr && nbr_w_out == nullptr && (reuse_pts == nullptr || reuse_mask != nullptr) &&
            !getenv("ANR_WARP_CLASSIFY_GENERIC")) {
            // the renderer's lean pass: its own kernel, loads in phases (warp_classify_lean_kernel)
            dim3 gl((unsigned)((N + LEAN_ITERS * LEAN_THREADS - 1) / (LEAN_ITERS * LEAN_THREADS)), bs);
#define ANR_CLASSIFY_LEAN(CL)                                                                                                  \
            hipLaunchKernelGGL((warp_classify_lean_kernel<CL>), gl, dim3(LEAN_THREADS), 0, st, rays, ray_stride, z, K, index, d, N,  \
                               dis_threshold, reinterpret_cast<float4*>(pts_out), w.list, w.cells, w.count, w.cell_count,          \
                               valid_mask_out, reinterpret_cast<const float4*>(reuse_pts), reuse_mask, reuse_perm, reuse_K, G,      \
                               z_steps ? 1 : 0)
            if (small) ANR_CLASSIFY_LEAN(false); else ANR_CLASSIFY_LEAN(true);
#undef ANR_CLASSIFY_LEAN
        }
        else if (z_steps) return fail(ANR_E_BADARG, "anr_warp_points: skip_far & 4 (z = step table) is the lean renderer pass's "
                                      "(validity outputs, K %% 4 == 0, 16-B aligned table, no neighbour outputs)");
        else if (vec4)           { if (small) ANR_CLASSIFY(true, true, false);   else ANR_CLASSIFY(true, true, true); }
        else if (xyz == nullptr) { if (small) ANR_CLASSIFY(true, false, false);  else ANR_CLASSIFY(true, false, true); }
        else                     { if (small) ANR_CLASSIFY(false, false, false); else ANR_CLASSIFY(false, false, true); }
#undef ANR_CLASSIFY
        if (int rc = check_launch("anr_warp_points (classify)")) return rc;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int64_t gx = (cus + bs - 1) / bs;                                   // one persistent workgroup per CU in total
        if (small) {
            const int64_t max_wg = (N + 64 * (WARP_THREADS / 64) - 1) / (64 * (WARP_THREADS / 64));
            if (gx > max_wg) gx = max_wg;
            const int group_bytes = bytes + (WARP_THREADS / 64) * GQ_BYTES;
            if (group_bytes <= 160 * 1024 && d.NT <= 32 && !getenv("ANR_WARP_LANE_PER_SAMPLE")) {
                // eight lanes per sample (warp_search_groups_kernel); the index + the blend queues fit for V <= ~9,000
                if (int rc = allow_big_lds(warp_search_groups_kernel, group_bytes, "anr_warp_points")) return rc;
                hipLaunchKernelGGL(warp_search_groups_kernel, dim3((unsigned)gx, bs), dim3(WARP_THREADS), group_bytes, st, index, d,
                                   ober2cano, lbs_weights, J, N, dis_threshold, reinterpret_cast<float4*>(pts_out), nbr_idx_out,
                                   nbr_w_out, w.list, w.count, valid_mask_out);
            } else {
                if (int rc = allow_big_lds(warp_search_kernel, bytes, "anr_warp_points")) return rc;
                hipLaunchKernelGGL(warp_search_kernel, dim3((unsigned)gx, bs), dim3(WARP_THREADS), bytes, st, index, d, ober2cano,
                                   lbs_weights, J, N, dis_threshold, reinterpret_cast<float4*>(pts_out), nbr_idx_out, nbr_w_out,
                                   w.list, w.count, w.cursor, nullptr, valid_mask_out, nullptr, G);
            }
            if (int rc = check_launch("anr_warp_points (search)")) return rc;
            if (lean) {
                if (int rc = zero_fill(valid_count_out, sizeof(int32_t), st, "anr_warp_points_lean (zero)")) return rc;
                const int64_t total = (int64_t)bs * N;
                const int64_t vb = (total + VL_SPAN - 1) / VL_SPAN;
                hipLaunchKernelGGL(warp_valid_list_kernel, dim3((unsigned)vb), dim3(VL_THREADS), 0, st,
                                   valid_mask_out, total, valid_index_out, valid_count_out);
                return check_launch("anr_warp_points_lean (valid list)");
            }
            return 0;
        }
        if (int rc = allow_big_lds(warp_cells_kernel, bytes, "anr_warp_points")) return rc;
        // (prev_ws: the cells an earlier call on this body / grid / threshold searched are copied, not searched again)
        const float* prev_cap2 = nullptr;
        const int32_t* prev_seed = nullptr;
        if (prev_ws != nullptr && !(prev_N < (int64_t)1 << 19 && !getenv("ANR_WARP_CELLS_ALWAYS")) && grid_for(prev_N) == G &&
            !getenv("ANR_WARP_NO_PREV_CELLS")) {
            const WarpWs pw(const_cast<int32_t*>(prev_ws), bs, prev_N);
            prev_cap2 = pw.cell_cap2;
            prev_seed = pw.cell_seed;
        }
        hipLaunchKernelGGL(warp_cell_list_kernel, dim3(cells / (4 * WARP_THREADS), bs), dim3(WARP_THREADS), 0, st, w.cell_count,
                           w.occ_list, w.occ_count, prev_cap2, prev_seed, w.cell_cap2, w.cell_seed);
        hipLaunchKernelGGL(warp_cells_kernel, dim3((unsigned)(gx < cells / WARP_THREADS ? gx : cells / WARP_THREADS), bs),
                           dim3(WARP_THREADS), bytes, st, index, d, dis_threshold, w.occ_list, w.occ_count, w.occ_cursor,
                           w.cell_cap2, w.cell_seed, G);
        // (occ_list is dead once the cells kernel has run: it becomes the scatter's fill counters)
        hipLaunchKernelGGL(warp_cell_scan_kernel, dim3(cells / 4096, bs), dim3(1024), 0, st, w.cell_count, w.cell_start, w.cell_cap2,
                           w.occ_list, w.live, G);
        // (threads x entries per trip: 512 x 4, 512 x 8, 1,024 x 8, twice the workgroups — all within 3 % of this by the kernel
        // trace, profiles/r06/ab_scatter_shapes.txt; issuing a trip's loads first is what took the pass from 77 / 138 to 68 / 120 us)
        const int64_t sc_blocks = (N + 4 * WARP_THREADS - 1) / (4 * WARP_THREADS);
        hipLaunchKernelGGL((warp_cell_scatter_kernel<WARP_THREADS, 4>), dim3((unsigned)(sc_blocks < 1024 ? sc_blocks : 1024), bs),
                           dim3(WARP_THREADS), 0, st, w.list, w.cells, w.count, N, w.cell_start, w.occ_list, w.cell_cap2, w.sorted);
        if (int rc = check_launch("anr_warp_points (bin)")) return rc;
        if (int rc = allow_big_lds(warp_search_kernel, bytes, "anr_warp_points")) return rc;
        const int64_t max_wg = (N + 64 * (WARP_THREADS / 64) - 1) / (64 * (WARP_THREADS / 64));
        if (gx > max_wg) gx = max_wg;
        hipLaunchKernelGGL(warp_search_kernel, dim3((unsigned)gx, bs), dim3(WARP_THREADS), bytes, st, index, d, ober2cano,
                           lbs_weights, J, N, dis_threshold, reinterpret_cast<float4*>(pts_out), nbr_idx_out, nbr_w_out,
                           w.sorted, w.live, w.cursor, w.cell_cap2, valid_mask_out, w.cell_seed, G);
        if (int rc = check_launch("anr_warp_points (search)")) return rc;
        if (lean) {
            if (int rc = zero_fill(valid_count_out, sizeof(int32_t), st, "anr_warp_points_lean (zero)")) return rc;
            const int64_t total = (int64_t)bs * N;
            const int64_t vb = (total + VL_SPAN - 1) / VL_SPAN;
            hipLaunchKernelGGL(warp_valid_list_kernel, dim3((unsigned)vb), dim3(VL_THREADS), 0, st,
                               valid_mask_out, total, valid_index_out, valid_count_out);
            return check_launch("anr_warp_points_lean (valid list)");
        }
        return 0;
    }
    if (xyz == nullptr) {
        if (int rc = allow_big_lds(warp_points_kernel<true>, bytes, "anr_warp_points")) return rc;
        const int64_t R = N / K;
        dim3 grid((unsigned)((R + RAYS_PER_WG - 1) / RAYS_PER_WG), bs);
        hipLaunchKernelGGL(warp_points_kernel<true>, grid, dim3(WARP_THREADS), bytes, st, xyz, xyz_stride, rays,
                           ray_stride, z, K, index, d, ober2cano, lbs_weights, J, N, dis_threshold, skip_far,
                           reinterpret_cast<float4*>(pts_out), dist_out, idx_out, blended_out, nbr_idx_out, nbr_w_out);
    } else {
        if (int rc = allow_big_lds(warp_points_kernel<false>, bytes, "anr_warp_points")) return rc;
        dim3 grid((unsigned)((N + PTS_PER_WG - 1) / PTS_PER_WG), bs);
        hipLaunchKernelGGL(warp_points_kernel<false>, grid, dim3(WARP_THREADS), bytes, st, xyz, xyz_stride, rays,
                           ray_stride, z, K, index, d, ober2cano, lbs_weights, J, N, dis_threshold, skip_far,
                           reinterpret_cast<float4*>(pts_out), dist_out, idx_out, blended_out, nbr_idx_out, nbr_w_out);
    }
    return check_launch("anr_warp_points");
}

extern "C" int anr_warp_backward(const float* d_pts, const float* rays, int ray_stride, const float* z, int K,
                                 const float* ober2cano, const int32_t* nbr_idx, const float* nbr_w, int bs, int V,
                                 int64_t N, float* d_ober2cano, float* d_rays, float* d_z, void* stream) {
    ANR_REQUIRE(d_pts && rays && z && ober2cano && nbr_idx && nbr_w && d_ober2cano && d_rays && d_z, ANR_E_BADARG,
                "anr_warp_backward: null pointer");
    ANR_REQUIRE(bs > 0 && V > 0 && N > 0 && K > 0 && N % K == 0 && ray_stride >= 8, ANR_E_BADARG,
                "anr_warp_backward: bs=%d V=%d N=%lld K=%d", bs, V, (long long)N, K);
    ANR_REQUIRE((((uintptr_t)d_pts | (uintptr_t)nbr_idx | (uintptr_t)nbr_w) & 15) == 0, ANR_E_ALIGN,
                "anr_warp_backward: d_pts / nbr_idx / nbr_w must be 16-B aligned");
    dim3 grid((unsigned)((N + WB_THREADS - 1) / WB_THREADS), bs);
    hipLaunchKernelGGL(warp_backward_kernel, grid, dim3(WB_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(d_pts), rays, ray_stride, z, K, ober2cano,
                       reinterpret_cast<const int4*>(nbr_idx), reinterpret_cast<const float4*>(nbr_w), V, N,
                       d_ober2cano, d_rays, d_z, nullptr);
    return check_launch("anr_warp_backward");
}

extern "C" int anr_warp_backward_compact(const float* d_pts_rows, const int32_t* pos, const float* rays, int ray_stride, const float* z,
                                         int K, const float* ober2cano, const int32_t* nbr_idx, const float* nbr_w, int bs, int V,
                                         int64_t N, float* d_ober2cano, float* d_rays, float* d_z, void* stream) {
    ANR_REQUIRE(d_pts_rows && pos && rays && z && ober2cano && nbr_idx && nbr_w && d_ober2cano && d_rays && d_z, ANR_E_BADARG,
                "anr_warp_backward_compact: null pointer");
    ANR_REQUIRE(bs > 0 && V > 0 && N > 0 && K > 0 && N % K == 0 && ray_stride >= 8, ANR_E_BADARG,
                "anr_warp_backward_compact: bs=%d V=%d N=%lld K=%d", bs, V, (long long)N, K);
    ANR_REQUIRE((((uintptr_t)d_pts_rows | (uintptr_t)nbr_idx | (uintptr_t)nbr_w) & 15) == 0, ANR_E_ALIGN,
                "anr_warp_backward_compact: d_pts_rows / nbr_idx / nbr_w must be 16-B aligned");
    dim3 grid((unsigned)((N + WB_THREADS - 1) / WB_THREADS), bs);
    hipLaunchKernelGGL(warp_backward_kernel, grid, dim3(WB_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(d_pts_rows), rays, ray_stride, z, K, ober2cano,
                       reinterpret_cast<const int4*>(nbr_idx), reinterpret_cast<const float4*>(nbr_w), V, N,
                       d_ober2cano, d_rays, d_z, pos);
    return check_launch("anr_warp_backward_compact");
}

extern "C" int anr_knn(const void* knn_index, const float* xyz, int bs, int V, int64_t N, float* dist_out,
                       int64_t* idx_out, void* stream) {
    ANR_REQUIRE(knn_index && xyz && dist_out && idx_out, ANR_E_BADARG, "anr_knn: null pointer");
    ANR_REQUIRE(bs > 0 && V >= 4 && N > 0, ANR_E_BADARG, "anr_knn: bs=%d V=%d N=%lld", bs, V, (long long)N);
    ANR_REQUIRE(((uintptr_t)knn_index & 15) == 0, ANR_E_ALIGN, "anr_knn: knn_index must be 16-B aligned");
    IndexDims d = index_dims(V);
    const int bytes = d.lds_floats() * 4;
    ANR_REQUIRE(bytes <= 160 * 1024, ANR_E_SHAPE, "anr_knn: V=%d needs %d B of LDS (>160 KiB)", V, bytes);
    if (int rc = allow_big_lds(knn_kernel, bytes, "anr_knn")) return rc;
    // points per workgroup: the staged index (114 KB) is worth amortising over 4 x 1,024 points only when that still leaves a
    // workgroup per CU
    const int iters = N >= (int64_t)256 * 4096 / bs ? 4 : 1;
    dim3 grid((unsigned)((N + (int64_t)iters * WARP_THREADS - 1) / ((int64_t)iters * WARP_THREADS)), bs);
    hipLaunchKernelGGL(knn_kernel, grid, dim3(WARP_THREADS), bytes, (hipStream_t)stream,
                       reinterpret_cast<const float*>(knn_index), d, xyz, N, dist_out, idx_out, iters);
    return check_launch("anr_knn");
}

extern "C" int anr_knn_within(const void* knn_index, const float* xyz, int bs, int V, int64_t N, float radius, float* d1_out,
                              void* stream) {
    ANR_REQUIRE(knn_index && xyz && d1_out, ANR_E_BADARG, "anr_knn_within: null pointer");
    ANR_REQUIRE(bs > 0 && V >= 4 && N > 0 && radius > 0.0f, ANR_E_BADARG, "anr_knn_within: bs=%d V=%d N=%lld radius=%g", bs, V, (long long)N,
                radius);
    ANR_REQUIRE(((uintptr_t)knn_index & 15) == 0, ANR_E_ALIGN, "anr_knn_within: knn_index must be 16-B aligned");
    IndexDims d = index_dims(V);
    const int bytes = d.lds_floats() * 4;
    ANR_REQUIRE(bytes <= 160 * 1024, ANR_E_SHAPE, "anr_knn_within: V=%d needs %d B of LDS (>160 KiB)", V, bytes);
    if (int rc = allow_big_lds(knn_within_kernel, bytes, "anr_knn_within")) return rc;
    hipLaunchKernelGGL(knn_within_kernel, dim3((unsigned)((N + WARP_THREADS - 1) / WARP_THREADS), bs), dim3(WARP_THREADS), bytes,
                       (hipStream_t)stream, reinterpret_cast<const float*>(knn_index), d, xyz, N, radius, d1_out);
    return check_launch("anr_knn_within");
}
