// The one-pass ray-march kernel (round 6): a ray comes in (32 bytes), its pixel goes out (coarse and fine colour, depth,
// opacity) — stratified coarse samples, point generation, Fourier encoding, the coarse network, alpha compositing, importance
// sampling + sorted merge, the fine network on the Kc + Kf sorted samples and the final compositing in ONE launch, with nothing
// per sample ever written to HBM.  This is BASELINE.json's "fused ray-march kernel that samples points, ..., Fourier-encodes, and
// alpha-composites in one pass" — anr_ray_march for the configuration without the warp (use_unpose=False: configs[1], the
// headline workload), anr_ray_march_warp (template flag WARP, below) with every sample inverse-skinned inside the pass (configs[2]);
// reference: models/volume_rendering.py:29-56 (sample_coarse), :113-160 (composite), :59-97 + :199-207 (sample_fine, merge),
// :163-232 (forward); models/nerf.py:129-175; models/embedding.py:22-39.
//
// Built from the parts of the staged path, not next to them: the network is Mlp<..., FUSED> of mlp_core.h — the same tile
// schedule, LDS-DMA weight ring and register-resident activations as anr_mlp_forward_rays[_steps]; the per-ray routines are
// composite_ray / fine_and_merge of composite_core.h — the same instruction sequences as anr_composite_sample / anr_composite.
// A point's value does not depend on the tile it sits in, so the kernel returns the staged path's BITS
// (tests/test_gpu_parity.py::test_one_pass_ray_march_equals_the_staged_path).
//
// A persistent workgroup (one per CU) owns GROUPS of G rays = one coarse point tile: bf16 8 waves x 32 points = 4 rays x 64
// samples, fp32 4 x 32 = 2 rays.  Per group: pass C (coarse network, 1 tile) -> the lanes leave (rgb, sigma) rows in LDS ->
// phase X: G / 2 wavefronts composite two rays each and run the importance sampler + merge on them (sorted depths stay in the
// rays' LDS segments) -> passes F0, F1 (fine network, 2 tiles: the points are o + z_sorted d) -> phase Y: the final compositing.
// The weight ring never drains: the last tile of a pass prefetches the first fragments and bias of the NEXT pass's network
// (coarse -> fine -> fine -> coarse ...: Mlp::FUSED).  LDS: two bias tables (20 KB) + the ring (120 KB) + rows (8 KB) + ray
// segments (8 KB) + rays / step table / u (1 KB).
//
// What it costs and saves against the staged path is in DESIGN.md section 4.5: the exchange through HBM (36 B per sample,
// ~0.9 ms of a 172 ms frame) against the matrix pipe idling through phases X and Y (two wavefronts busy, six waiting).
#include "mlp_core.h"
#undef ANR_SEARCH_PROF                      // (the search's experiment counters belong to warp.hip's builds)
#include "warp_core.h"                      // (before composite_core.h: that one switches fp contraction off for what follows)
#include "composite_core.h"

namespace anr {

constexpr int RM_KC = 64, RM_KF = 64, RM_K = RM_KC + RM_KF, RM_KT = 128;

// WARP (anr_ray_march_warp): every sample is inverse-skinned inside the pass — models/anim_nerf.py:153-192: exact 4 nearest
// posed vertices, blend-weight confidence, blended ober2cano transform, validity by the blended distance — before it is encoded;
// the network then runs on EVERY sample, as the reference's does, and sigma is -1e5 where the sample is not valid (:305).  The
// index is read where anr_knn_index_build left it (global memory, L2-resident: this kernel's LDS is the weight ring's), with the
// routines of warp_core.h — the neighbours, canonical points and validity bits of anr_warp_points, hence the staged path's image
// bit for bit.  Samples farther than dis_threshold from the body's box, or in a cell no vertex can reach (the reach mask), skip
// the search as in the staged classify pass.  This is the north star's single pass WITH the warp; it is the dense evaluation
// (the staged path's compaction of a whole frame's valid samples into full tiles is what a four-ray group cannot do) and is
// measured next to it in DESIGN.md section 4.5.
struct RmWarp {
    const float* index;          // anr_knn_index_build's output, bs bodies
    IndexDims d;
    const float* o2c;            // ober2cano[bs][V][16]
    const float* lbs_w;          // lbs_weights[V][J]
    int J;
    float thr;                   // dis_threshold
    int64_t rays_per_body;       // ray r belongs to body r / rays_per_body
};

template <int MODE, bool WARP = false>
struct RayMarch : Mlp<MODE, true, false, false, false, false, false, false, true> {
    using Base = Mlp<MODE, true, false, false, false, false, false, false, true>;
    using C = typename Base::C;
    using Frag = typename Base::Frag;
    static constexpr int NT = Base::NT, EF = Base::EF, HF = Base::HF, DF = Base::DF, WAVES = Base::WAVES, THREADS = Base::THREADS;
    static_assert(NT == 1, "one 32-point column tile per wavefront");
    static constexpr int TILE = WAVES * 32;                 // points per pass
    static constexpr int G = TILE / RM_KC;                  // rays per group: 4 (bf16) / 2 (fp32)
    static constexpr int NF_T = G * RM_K / TILE;            // fine passes per group: 2
    static_assert(G * RM_KC == TILE && NF_T * TILE == G * RM_K && (G % 2) == 0, "group shape");
    static constexpr int SLOT = Base::SLOT;
    // LDS map (bytes)
    static constexpr int OFF_BIAS_C = 0, OFF_BIAS_F = BIAS_BYTES, OFF_RING = 2 * BIAS_BYTES, OFF_ROWS = OFF_RING + 3 * SLOT,
                         OFF_SEG = OFF_ROWS + G * RM_K * 16, OFF_RAYS = OFF_SEG + G * (int)sizeof(RayLds<RM_KT>),
                         OFF_STEPS = OFF_RAYS + 2 * G * 8 * 4, OFF_U = OFF_STEPS + RM_KC * 4, LDS_BYTES = OFF_U + RM_KF * 4;

    using Base::wave; using Base::lane; using Base::half; using Base::lds_base; using Base::lds_bias; using Base::lds_bias_next;
    using Base::slot_cur; using Base::slot_nxt; using Base::slot_stage; using Base::gnext; using Base::gbase; using Base::more;
    using Base::c; using Base::acc; using Base::bias_c; using Base::w0; using Base::spend_nf; using Base::spend_slot;

    // One pass of the network over this wavefront's 32 points: the tile schedule of Mlp::run (mlp_core.h), outputs in registers.
    __device__ __forceinline__ float4 pass(const float4& p) {
        c = 0;
        Frag E[NT][EF];
        const float xs[3] = {p.x, p.y, p.z};
        this->encode_panel(xs, E[0]);
        Frag A[NT][HF], B[NT][HF];
        float sigma[NT];
        using NoEpi = typename Base::NoEpi;
        using SigmaEpi = typename Base::SigmaEpi;
        this->template layer<0, 8, EF, 0, true, HF, HF>(E, B, A, NoEpi{});
        this->template layer<8, 8, 0, HF, true, HF, HF>(E, A, B, this->template last_of<0, 8, true>(A));
        this->template layer<16, 8, 0, HF, true, HF, HF>(E, B, A, this->template last_of<8, 8, true>(B));
        this->template layer<24, 8, 0, HF, true, HF, HF>(E, A, B, this->template last_of<16, 8, true>(A));
        this->template layer<32, 8, EF, HF, true, HF, HF>(E, B, A, this->template last_of<24, 8, true>(B));
        this->template layer<40, 8, 0, HF, true, HF, HF>(E, A, B, this->template last_of<32, 8, true>(A));
        this->template layer<48, 8, 0, HF, true, HF, HF>(E, B, A, this->template last_of<40, 8, true>(B));
        this->template layer<56, 8, 0, HF, true, HF, HF>(E, A, B, this->template last_of<48, 8, true>(A));
        this->template tile<64, 0, HF, HF>(E, B, this->template last_of<56, 8, true>(B));
        this->template layer<65, 8, 0, HF, false, HF, HF>(E, B, A, SigmaEpi{acc[64 & 1], sigma});
        Frag Gd[NT][DF];
        this->template layer<73, 4, 0, HF, true, HF, DF>(E, A, Gd, this->template last_of<65, 8, false>(A));
        this->template tile<77, 0, DF, DF>(E, Gd, this->template last_of<73, 4, true>(Gd));
        const f32x16& r = acc[77 & 1][0];
        // (the lower half-wave holds rows 0..2 of the colour tile: the expressions of Mlp::run's output stage)
        const float cr = 1.0f / (1.0f + expf(-r[0]));
        const float cg = 1.0f / (1.0f + expf(-r[1]));
        const float cb = 1.0f / (1.0f + expf(-r[2]));
        return make_float4(cr, cg, cb, sigma[0]);
    }

    // x = o + z d with the product and the sum rounded separately (anr_points_from_rays; Mlp::run's fetch_pts)
    static __device__ __forceinline__ float4 point_at(const float* ry, float zz) {
        float m[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) asm("v_mul_f32_e32 %0, %1, %2" : "=v"(m[a]) : "v"(zz), "v"(ry[3 + a]));
        return make_float4(ry[0] + m[0], ry[1] + m[1], ry[2] + m[2], 1.0f);
    }

    // (x_canonical, valid) of the sample at depth zz of the ray ry[8] of body (ray / rays_per_body): the staged classify pass's
    // tests (box within dis_threshold, reach mask), then the exact search and the blend of warp_core.h on the lower half-wave —
    // the upper one holds the same 32 points for the other half of the encoding and takes the result over
    __device__ __forceinline__ float4 warped_point(const RmWarp& wp, const float* ry, float zz, int64_t ray) const {
        // (a wavefront's 32 samples belong to ONE ray, hence one body: the index base is a scalar.  The tree walk's box reads —
        // every lane the same box — still compile to vector loads: hipcc keeps uniform loads of plain global memory off the
        // scalar unit in a kernel that stores.  A walk written on s_load through the constant cache is the lever section 4.5 names.)
        const int b = __builtin_amdgcn_readfirstlane((int)(ray / wp.rays_per_body));
        const float* my_index = wp.index + (int64_t)b * wp.d.total_floats();
        const float* gbox = my_index + wp.d.body_off();
        const float px = __fadd_rn(ry[0], __fmul_rn(zz, ry[3]));
        const float py = __fadd_rn(ry[1], __fmul_rn(zz, ry[4]));
        const float pz = __fadd_rn(ry[2], __fmul_rn(zz, ry[5]));
        bool near = half == 0 && box_d2(gbox, px, py, pz) < wp.thr * wp.thr;
        const float reach_thr = gbox[3];
        if (reach_thr >= wp.thr && near) {
            const float inv = 1.0f / reach_cell_size(gbox, reach_thr);
            const int rc = reach_cell(gbox, reach_thr, inv, px, py, pz);
            const unsigned* reach = reinterpret_cast<const unsigned*>(my_index + wp.d.reach_off());
            near = rc >= 0 && ((reach[rc >> 5] >> (rc & 31)) & 1u);
        }
        float4 res = make_float4(px, py, pz, 0.0f);
        if (__any(near)) {
            Best4 best;
            best_init(best);
            search(my_index, wp.d, px, py, pz, near, best);
            if (near) {
                const int32_t* order = reinterpret_cast<const int32_t*>(my_index + wp.d.order_off());
                blend_and_store(best, order, wp.lbs_w, wp.J, wp.o2c + (int64_t)b * wp.d.V * 16, wp.thr, px, py, pz, 0, &res, nullptr,
                                nullptr, nullptr, nullptr, nullptr);
            }
        }
        const int src = (lane & 31) * 4;
        res.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(res.x)));
        res.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(res.y)));
        res.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(res.z)));
        res.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(res.w)));
        return res;
    }

    __device__ __forceinline__ void run(const char* __restrict__ pack_c, const char* __restrict__ pack_f, const float* __restrict__ rays,
                                        int stride, int64_t R, const float* __restrict__ steps_g, const float* __restrict__ u_g,
                                        int white_bkgd, float* __restrict__ rgb_c, float* __restrict__ dep_c, float* __restrict__ acc_c,
                                        float* __restrict__ rgb_f, float* __restrict__ dep_f, float* __restrict__ acc_f, char* lds,
                                        const RmWarp& wp) {
        const int64_t n_groups = (R + G - 1) / G;
        if ((int64_t)blockIdx.x >= n_groups) return;          // (before anything is in flight into LDS)
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        half = lane >> 5;
        const int l = lane & 31;
        lds_base = lds;
        float4* rows = reinterpret_cast<float4*>(lds + OFF_ROWS);
        RayLds<RM_KT>* seg = reinterpret_cast<RayLds<RM_KT>*>(lds + OFF_SEG);
        float* raybufs = reinterpret_cast<float*>(lds + OFF_RAYS);      // [2][G][8]: this group's rays and the next one's
        float* steps = reinterpret_cast<float*>(lds + OFF_STEPS);
        float* utab = reinterpret_cast<float*>(lds + OFF_U);
        char* bias_tab_c = lds + OFF_BIAS_C;
        char* bias_tab_f = lds + OFF_BIAS_F;
        slot_cur = OFF_RING;
        slot_nxt = slot_cur + SLOT;
        slot_stage = slot_nxt + SLOT;
        const char* first_c = pack_c + BIAS_BYTES;
        const char* first_f = pack_f + BIAS_BYTES;
        for (int i = threadIdx.x; i < BIAS_BYTES / 16; i += THREADS) {
            reinterpret_cast<uint4*>(bias_tab_c)[i] = reinterpret_cast<const uint4*>(pack_c)[i];
            reinterpret_cast<uint4*>(bias_tab_f)[i] = reinterpret_cast<const uint4*>(pack_f)[i];
        }
        // a group's rays (a ray past the end shadows ray R - 1 and stores nothing): lanes 0 .. 8 G - 1 of the LAST wavefront
        auto ray_word = [&](int64_t grp_i) {
            const int64_t rr = grp_i * G + (lane >> 3);
            return rays[(rr < R ? rr : R - 1) * stride + (lane & 7)];
        };
        const bool ray_loader = wave == WAVES - 1 && lane < G * 8;
        if (ray_loader) raybufs[lane] = ray_word(blockIdx.x);
        if (threadIdx.x < RM_KC) steps[threadIdx.x] = steps_g[threadIdx.x];
        if (threadIdx.x < RM_KF) utab[threadIdx.x] = u_g[threadIdx.x];
        gnext = first_c;
        spend_nf = 0; spend_slot = 0;
        stage_chunk<true, WAVES>(gnext, lds_base, slot_cur, chunk_frags<C, false>(0), wave, lane);
        gnext += chunk_frags<C, false>(0) * FRAG_BYTES;
        stage_chunk<true, WAVES>(gnext, lds_base, slot_nxt, chunk_frags<C, false>(1), wave, lane);
        gnext += chunk_frags<C, false>(1) * FRAG_BYTES;
        lds_bias = bias_tab_c;
        dma_wait();
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) w0[q] = (reinterpret_cast<const Frag*>(lds_base + slot_cur) + lane)[q * 64];
        bias_c = this->read_bias(0);

        // a ray's LDS segment is touched by the lanes of ONE wavefront only in phases X and Y (composite.hip)
        auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };

        int parity = 0;
        for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x, parity ^= 1) {
            const bool last_group = grp + gridDim.x >= n_groups;
            const int64_t r0 = grp * G;
            // ---- the group's rays are in LDS (loaded during the group before); the next group's are fetched now — the load is in
            // flight through pass C — and stored in the phase-X window
            float* raybuf = raybufs + (parity ? G * 8 : 0);
            float* raynext = raybufs + (parity ? 0 : G * 8);
            float next_word = 0.0f;
            if (ray_loader && !last_group) next_word = ray_word(grp + gridDim.x);
            // ---- three passes per group through ONE call site (ps = 0: the coarse network on the stratified samples; 1, 2: the
            // fine network on the two tiles of sorted samples), phases X and Y behind passes 0 and 2
#pragma unroll 1
            for (int ps = 0; ps <= NF_T; ++ps) {
                // point i of the group's coarse (fine) samples = sample k of ray g
                const int i = (ps ? (ps - 1) * TILE : 0) + wave * 32 + l;
                const int g = ps ? i / RM_K : i / RM_KC, k = ps ? i % RM_K : i % RM_KC;
                const float* ry = raybuf + g * 8;
                float zz;
                if (ps == 0) {
                    const float sk = steps[k];
                    float one_minus, lo, hi;                   // z = near' (1 - s) + far' s with anr_sample_coarse's roundings
                    asm("v_sub_f32_e32 %0, 1.0, %1" : "=v"(one_minus) : "v"(sk));
                    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(lo) : "v"(ry[6]), "v"(one_minus));
                    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(hi) : "v"(ry[7]), "v"(sk));
                    zz = lo + hi;
                } else {
                    zz = seg[g].wbuf[k];
                }
                float4 p;
                if constexpr (WARP) p = warped_point(wp, ry, zz, r0 + g < R ? r0 + g : R - 1);
                else p = point_at(ry, zz);
                const bool last_pass = ps == NF_T;
                lds_bias = ps ? bias_tab_f : bias_tab_c;
                lds_bias_next = last_pass ? bias_tab_c : bias_tab_f;
                gbase = last_pass ? first_c : first_f;
                more = !(last_pass && last_group);
                float4 o = pass(p);
                if constexpr (WARP) o.w = (p.w < 1.0f) ? -1e5f : o.w;          // models/anim_nerf.py:305
                if (half == 0) rows[i] = o;
                if (ps != 0 && ps != NF_T) continue;           // (the fine tiles' rows are read after the second one)
                __syncthreads();
                if (ps == 0 && ray_loader) raynext[lane] = next_word;
                if (ps == 0) {
                    // ---- phase X: coarse compositing + importance sampling + merge, two rays per wavefront (composite_sample_kernel)
                    if (wave < G / 2) {
                        const int gx = 2 * wave + half;
                        RayLds<RM_KT>& L = seg[gx];
                        const float* rx = raybuf + gx * 8;
                        const float near = rx[6], far = rx[7];
                        const bool active = r0 + gx < R;
                        float uu[(RM_KT + 31) / 32];
                        load_u<32, RM_KT, RM_KF>(utab, lane, RM_KF, uu);
                        float w[2], zs[2], wsum, cr, cg, cb, dep;
                        const float4* crow = rows + gx * RM_KC;
                        composite_ray<2, 32>(lane, RM_KC, [&](int kk, int) { return crow[kk]; },
                                             [&](int kk) { const float sk = steps[kk]; return near * (1.0f - sk) + far * sk; },
                                             [](int) { return 0.0f; }, w, zs, wsum, cr, cg, cb, dep);
#pragma unroll
                        for (int s = 0; s < 2; ++s) { L.zall[s * 32 + l] = zs[s]; L.wbuf[s * 32 + l] = w[s]; }
#pragma unroll
                        for (int k0 = 0; k0 <= RM_KC; k0 += 32)
                            if (k0 + 32 <= RM_KC + 1 || l <= RM_KC - k0) L.hist[k0 + l] = 0;
                        if (l == 31 && active) {
                            if (white_bkgd) {
                                dep = dep + (1.0f - wsum) * far;
                                cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
                            }
                            const int64_t r = r0 + gx;
                            rgb_c[r * 3 + 0] = cr; rgb_c[r * 3 + 1] = cg; rgb_c[r * 3 + 2] = cb;
                            dep_c[r] = dep;
                            acc_c[r] = wsum;
                        }
                        sync();
                        fine_and_merge<32, RM_KT, uint8_t, RM_KC, RM_KF>(L, lane, uu, RM_KC, RM_KF, nullptr, false, sync);
                    }
                } else {
                    // ---- phase Y: the final compositing (composite_kernel<4, 32>), two rays per wavefront
                    if (wave < G / 2) {
                        const int gx = 2 * wave + half;
                        const RayLds<RM_KT>& L = seg[gx];
                        const bool active = r0 + gx < R;
                        float w[4], zs[4], wsum, cr, cg, cb, dep;
                        const float4* crow = rows + gx * RM_K;
                        composite_ray<4, 32>(lane, RM_K, [&](int kk, int) { return crow[kk]; }, [&](int kk) { return L.wbuf[kk]; },
                                             [](int) { return 0.0f; }, w, zs, wsum, cr, cg, cb, dep);
                        if (l == 31 && active) {
                            if (white_bkgd) {
                                const float far = raybuf[gx * 8 + 7];
                                dep = dep + (1.0f - wsum) * far;
                                cr = cr + 1.0f - wsum; cg = cg + 1.0f - wsum; cb = cb + 1.0f - wsum;
                            }
                            const int64_t r = r0 + gx;
                            rgb_f[r * 3 + 0] = cr; rgb_f[r * 3 + 1] = cg; rgb_f[r * 3 + 2] = cb;
                            dep_f[r] = dep;
                            acc_f[r] = wsum;
                        }
                    }
                }
                if (ps == 0) __syncthreads();                  // (the sorted depths are the fine passes' input)
            }
            // (no barrier here: the next group writes `rows` at the END of its pass C — 39 chunk barriers after every wavefront
            // has left phase Y —, the segments and the other ray buffer behind the barrier that follows that pass)
        }
    }
};

template <int MODE, bool WARP>
__global__ __launch_bounds__(Cfg<MODE>::WAVES * 64, Cfg<MODE>::WAVES / 4) void ray_march_kernel(
    const char* __restrict__ pack_c, const char* __restrict__ pack_f, const float* __restrict__ rays, int stride, int64_t R,
    const float* __restrict__ steps, const float* __restrict__ u, int white_bkgd, float* __restrict__ rgb_c, float* __restrict__ dep_c,
    float* __restrict__ acc_c, float* __restrict__ rgb_f, float* __restrict__ dep_f, float* __restrict__ acc_f, RmWarp wp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    RayMarch<MODE, WARP> m;
    m.run(pack_c, pack_f, rays, stride, R, steps, u, white_bkgd, rgb_c, dep_c, acc_c, rgb_f, dep_f, acc_f, lds, wp);
}

template <int MODE, bool WARP>
static int launch_ray_march(const void* pack_c, const void* pack_f, const float* rays, int stride, int64_t R, const float* steps,
                            const float* u, int white_bkgd, float* rgb_c, float* dep_c, float* acc_c, float* rgb_f, float* dep_f,
                            float* acc_f, hipStream_t st, const RmWarp& wp) {
    using RM = RayMarch<MODE, WARP>;
    auto kern = ray_march_kernel<MODE, WARP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RM::LDS_BYTES);
    if (e != hipSuccess) return fail((int)e, "anr_ray_march: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int64_t n_groups = (R + RM::G - 1) / RM::G;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    dim3 grid((unsigned)(n_groups < cus ? n_groups : cus));   // one persistent workgroup per CU (LDS-limited)
    hipLaunchKernelGGL(kern, grid, dim3(RM::THREADS), RM::LDS_BYTES, st, reinterpret_cast<const char*>(pack_c),
                       reinterpret_cast<const char*>(pack_f), rays, stride, R, steps, u, white_bkgd, rgb_c, dep_c, acc_c, rgb_f, dep_f,
                       acc_f, wp);
    return check_launch("anr_ray_march");
}

}  // namespace anr

using namespace anr;

static int ray_march_any(const void* pack_coarse, const void* pack_fine, int mode, const float* rays, int ray_stride, int64_t R,
                         const float* steps, int Kc, const float* u, int Kf, int white_bkgd, float* rgb_coarse, float* depth_coarse,
                         float* acc_coarse, float* rgb_fine, float* depth_fine, float* acc_fine, const RmWarp* wp, void* stream) {
    ANR_REQUIRE(pack_coarse && pack_fine && rays && steps && u && rgb_coarse && depth_coarse && acc_coarse && rgb_fine && depth_fine &&
                acc_fine, ANR_E_BADARG, "anr_ray_march: null pointer");
    ANR_REQUIRE(R > 0 && ray_stride >= 8, ANR_E_BADARG, "anr_ray_march: R=%lld stride=%d", (long long)R, ray_stride);
    ANR_REQUIRE(Kc == RM_KC && Kf == RM_KF, ANR_E_SHAPE, "anr_ray_march: built for %d + %d samples per ray (got %d + %d)", RM_KC, RM_KF,
                Kc, Kf);
    ANR_REQUIRE((((uintptr_t)pack_coarse | (uintptr_t)pack_fine) & 15) == 0, ANR_E_ALIGN, "anr_ray_march: packs must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const RmWarp none{};
#define ANR_RM(M)                                                                                                                    \
    (wp ? launch_ray_march<M, true>(pack_coarse, pack_fine, rays, ray_stride, R, steps, u, white_bkgd, rgb_coarse, depth_coarse,    \
                                    acc_coarse, rgb_fine, depth_fine, acc_fine, st, *wp)                                            \
        : launch_ray_march<M, false>(pack_coarse, pack_fine, rays, ray_stride, R, steps, u, white_bkgd, rgb_coarse, depth_coarse,   \
                                     acc_coarse, rgb_fine, depth_fine, acc_fine, st, none))
    switch (mode & 0xff) {
        case ANR_MLP_F32:  return ANR_RM(ANR_MLP_F32);
        case ANR_MLP_BF16: return ANR_RM(ANR_MLP_BF16_W8);
        default:           return fail(ANR_E_BADARG, "anr_ray_march: unknown mode %d", mode);
    }
#undef ANR_RM
}

extern "C" int anr_ray_march(const void* pack_coarse, const void* pack_fine, int mode, const float* rays, int ray_stride, int64_t R,
                             const float* steps, int Kc, const float* u, int Kf, int white_bkgd, float* rgb_coarse,
                             float* depth_coarse, float* acc_coarse, float* rgb_fine, float* depth_fine, float* acc_fine,
                             void* stream) {
    return ray_march_any(pack_coarse, pack_fine, mode, rays, ray_stride, R, steps, Kc, u, Kf, white_bkgd, rgb_coarse, depth_coarse,
                         acc_coarse, rgb_fine, depth_fine, acc_fine, nullptr, stream);
}

extern "C" int anr_ray_march_warp(const void* pack_coarse, const void* pack_fine, int mode, const float* rays, int ray_stride, int bs,
                                  int64_t rays_per_body, const float* steps, int Kc, const float* u, int Kf, int white_bkgd,
                                  const void* knn_index, const float* ober2cano, const float* lbs_weights, int V, int J,
                                  float dis_threshold, float* rgb_coarse, float* depth_coarse, float* acc_coarse, float* rgb_fine,
                                  float* depth_fine, float* acc_fine, void* stream) {
    ANR_REQUIRE(knn_index && ober2cano && lbs_weights, ANR_E_BADARG, "anr_ray_march_warp: null pointer");
    ANR_REQUIRE(bs > 0 && rays_per_body > 0 && V >= 4 && J > 0 && J <= MAX_J && dis_threshold > 0.0f, ANR_E_BADARG,
                "anr_ray_march_warp: bs=%d rays_per_body=%lld V=%d J=%d dis_threshold=%g", bs, (long long)rays_per_body, V, J, dis_threshold);
    ANR_REQUIRE((((uintptr_t)knn_index | (uintptr_t)ober2cano) & 15) == 0, ANR_E_ALIGN, "anr_ray_march_warp: knn_index / ober2cano must be 16-B aligned");
    RmWarp wp;
    wp.index = reinterpret_cast<const float*>(knn_index);
    wp.d = index_dims(V);
    wp.o2c = ober2cano; wp.lbs_w = lbs_weights; wp.J = J; wp.thr = dis_threshold; wp.rays_per_body = rays_per_body;
    return ray_march_any(pack_coarse, pack_fine, mode, rays, ray_stride, (int64_t)bs * rays_per_body, steps, Kc, u, Kf, white_bkgd,
                         rgb_coarse, depth_coarse, acc_coarse, rgb_fine, depth_fine, acc_fine, &wp, stream);
}
