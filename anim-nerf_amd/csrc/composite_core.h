// (header) The per-ray routines of the compositing and importance-sampling kernels (a13, a14) — the cross-lane primitives on DPP,
// composite_ray, the importance sampler + merge on a ray's LDS segment — shared by csrc/composite.hip (the stand-alone kernels)
// and csrc/ray_march.hip (the one-pass ray-march kernel, round 6): ONE definition, so that both produce the same bits.
// Reference: models/volume_rendering.py:59-97, :122-160, :199-207.
#pragma once
#include "anr_common.h"

#pragma clang fp contract(off)

namespace anr {

// ---- cross-lane primitives on DPP (data-parallel-primitive modifiers of VALU instructions: no LDS crossbar, no
// ds_bpermute latency).  Rows are 16 lanes; row_bcast:15 / row_bcast:31 carry a row's (two rows') last lane into the next
// row(s), which is exactly the carry of a 32- or 64-lane segment.
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float v) {     // lanes that receive nothing keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
// inclusive product / sum over each segment of W lanes (W = 32: two rays per wavefront, W = 64: one)
// (v_mul_f32_dpp with the accumulator as destination: a lane that receives nothing is left as it is, i.e. multiplied by 1.
// The compiler fuses `v += dpp(0, v)` into v_add_f32_dpp by itself but turns `v *= dpp(1, v)` into v_mov 1.0 / v_mov_dpp /
// v_mul — three issue slots per step of a kernel that is bound by them.  s_nop 1: a DPP operand written by the previous VALU
// instruction needs two wait states, and the compiler does not look into the asm.)
template <int W>
__device__ __forceinline__ float seg_incl_prod(float v) {
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    if (W == 64) asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}
template <int W>
__device__ __forceinline__ float seg_incl_sum(float v) {
    v += dpp<DPP_ROW_SHR1>(0.0f, v);
    v += dpp<DPP_ROW_SHR2>(0.0f, v);
    v += dpp<DPP_ROW_SHR4>(0.0f, v);
    v += dpp<DPP_ROW_SHR8>(0.0f, v);
    v += dpp<DPP_ROW_BCAST15, 0xA>(0.0f, v);
    if (W == 64) v += dpp<DPP_ROW_BCAST31, 0xC>(0.0f, v);
    return v;
}
// the value of the segment's last lane, in every lane of the segment
template <int W>
__device__ __forceinline__ float seg_last(float v, int lane) {
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    if (W == 64) return hi;
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    return lane < 32 ? lo : hi;
}
// sum over the segment, in every lane (order: the DPP tree above — every caller of a given W gets the same bits)
template <int W>
__device__ __forceinline__ float seg_sum(float v, int lane) { return seg_last<W>(seg_incl_sum<W>(v), lane); }

// exclusive multiplicative scan across the wave; returns product of lanes < lane
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc *= t;
    }
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}

// reference: models/volume_rendering.py:122-160
// LPR lanes per ray (64: one ray per wavefront; 32: two), S samples per lane.  Lane l owns samples l, l + LPR, ...: every
// load instruction of a segment reads LPR consecutive 16-byte rows (a lane-contiguous split of the ray streams at 4.3 TB/s,
// this one at 6.3: tools/exp/exp_composite.hip); the transmittance is one DPP scan per chunk of LPR samples with a carry.
// Totals (wsum, colour, depth) come back in the segment's LAST lane.
template <int S, int LPR, class Col, class Depth, class Noise>
__device__ __forceinline__ void composite_ray(int lane, int K, Col col_of, Depth depth_of, Noise noise_of, float (&w)[S],
                                              float (&zz)[S], float& wsum, float& cr, float& cg, float& cb, float& dep) {
    const int l = lane % LPR;
    float4 col[S];
    float alpha[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {                          // all loads first: S x LPR x 16 B in flight per ray
        const int k = s * LPR + l;
        col[s] = make_float4(0.f, 0.f, 0.f, 0.f); zz[s] = 0.0f;
        if (k < K) { col[s] = col_of(k, s); zz[s] = depth_of(k); }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = s * LPR + l;
        // the next depth lives one lane up, or in the segment's first lane of the next chunk (re-loading z[k + 1] costs
        // 15 % of the kernel: misaligned rows)
        float nxt = dpp<DPP_WAVE_SHL1>(0.0f, zz[s]);
        if (s + 1 < S) {
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zz[s + 1 < S ? s + 1 : s]), 0));
            const float hi = LPR == 64 ? lo : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zz[s + 1 < S ? s + 1 : s]), 32));
            if (l == LPR - 1) nxt = (lane < 32) ? lo : hi;
        }
        alpha[s] = 0.0f;
        if (k < K) {
            const float delta = (k + 1 < K) ? (nxt - zz[s]) : 1e10f;
            const float sg = col[s].w + noise_of(k);
            alpha[s] = 1.0f - expf(-delta * fmaxf(sg, 0.0f));
        }
    }
    float carry = 1.0f;
    wsum = 0.f; cr = 0.f; cg = 0.f; cb = 0.f; dep = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float inc = seg_incl_prod<LPR>(1.0f - alpha[s] + 1e-10f);      // (alpha = 0 beyond K: factor 1 + 1e-10 = 1)
        float ex = dpp<DPP_WAVE_SHR1>(1.0f, inc);
        ex = (l == 0) ? 1.0f : ex;
        w[s] = alpha[s] * (carry * ex);
        if (s + 1 < S) carry = carry * seg_last<LPR>(inc, lane);
        wsum += w[s]; cr += w[s] * col[s].x; cg += w[s] * col[s].y; cb += w[s] * col[s].z; dep += w[s] * zz[s];
    }
    wsum = seg_incl_sum<LPR>(wsum); cr = seg_incl_sum<LPR>(cr); cg = seg_incl_sum<LPR>(cg); cb = seg_incl_sum<LPR>(cb);
    dep = seg_incl_sum<LPR>(dep);
}


// ---------------------------------------------------------------------------------------------
// a14: importance sampling + merge (models/volume_rendering.py:59-97, :199-207), as a wavefront-segment routine shared
// by the stand-alone kernel (weights from HBM: the training path) and the fused coarse compositor below.
//
// Per ray the segment's LDS holds  zall[KT] (coarse depths, then fine) | cdf[KT] (later: the permutation) |
// wbuf[KT] (coarse weights, later: the sorted depths).  The cdf is accumulated as torch's CPU cumsum accumulates it:
// in double, rounded to float per entry (at::acc_type<float, false> = double) — so that, given the same weights, the
// `denom < eps` branch below takes the reference's side (it flips on the last ulp of the cdf in empty bins).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double old, double v) {
    const uint64_t o = __builtin_bit_cast(uint64_t, old), x = __builtin_bit_cast(uint64_t, v);
    // every caller passes old = 0: with all rows taking part, bound_ctrl writes that zero itself (no v_mov 0 per half)
    constexpr bool BC = ROW_MASK == 0xf;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)x, CTRL, ROW_MASK, 0xf, BC);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, ROW_MASK, 0xf, BC);
    return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
template <int W>
__device__ __forceinline__ double seg_excl_sum_d(double v, int l) {
    v += dpp_d<DPP_ROW_SHR1>(0.0, v);
    v += dpp_d<DPP_ROW_SHR2>(0.0, v);
    v += dpp_d<DPP_ROW_SHR4>(0.0, v);
    v += dpp_d<DPP_ROW_SHR8>(0.0, v);
    v += dpp_d<DPP_ROW_BCAST15, 0xA>(0.0, v);
    if (W == 64) v += dpp_d<DPP_ROW_BCAST31, 0xC>(0.0, v);
    const double ex = dpp_d<DPP_WAVE_SHR1>(0.0, v);
    return l == 0 ? 0.0 : ex;
}

template <int KT> struct RayLds {
    float zall[KT];
    float cdf[KT];
    float wbuf[KT];
    int hist[KT + 4];                                     // hist[c] = number of fine samples with exactly c coarse depths <= them
};

// `sync` = a barrier that makes this segment's LDS writes visible to its other lanes.  On entry zall[0..Kc) = coarse
// depths, wbuf[0..Kc) = coarse weights, hist[0..Kc] = 0, all visible.
// Returns with wbuf[0..K) = sorted depths and (want_perm) ((PermT*)cdf)[0..K) = permutation.
// KC / KF > 0: the sample counts as compile-time constants (the shipped shapes): every bound check below folds away.
// The lane's u values (fine sample j = f * LPR + l), loaded by the CALLER before its own first wait on memory: behind the
// compositing they were a round trip to HBM in the middle of the wavefront's chain, one per sample (round 5).
template <int LPR, int KT, int KF = 0>
__device__ __forceinline__ void load_u(const float* __restrict__ u_row, int lane, int Kf_rt, float (&uu)[(KT + LPR - 1) / LPR]) {
    const int Kf = KF ? KF : Kf_rt, l = lane % LPR;
#pragma unroll
    for (int f = 0; f < (KT + LPR - 1) / LPR; ++f) uu[f] = (f * LPR + l < Kf) ? u_row[f * LPR + l] : 0.0f;
}

template <int LPR, int KT, typename PermT, int KC = 0, int KF = 0, class Sync>
__device__ __forceinline__ void fine_and_merge(RayLds<KT>& L, int lane, const float (&uu)[(KT + LPR - 1) / LPR], int Kc_rt, int Kf_rt,
                                               float* __restrict__ z_fine_row, bool want_perm, Sync sync) {
    constexpr int MAXS = (KT + LPR - 1) / LPR;
    constexpr int NS = KC ? (KC - 2 + LPR - 1) / LPR : MAXS;       // pdf entries per lane (a 0 / total beyond them is not folded away)
    constexpr int NF = KF ? (KF + LPR - 1) / LPR : MAXS;           // fine samples per lane
    const int Kc = KC ? KC : Kc_rt, Kf = KF ? KF : Kf_rt;
    const int l = lane % LPR;
    const float eps = 1e-5f;
    const int nb = Kc - 1;                                // bins and cdf entries
    const int np = Kc - 2;                                // pdf entries
    // pdf over weights[1:-1] + eps; each lane owns a contiguous run of S entries
    const int S = (np + LPR - 1) / LPR;
    float wl[NS];
    float loc = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int i = l * S + s;
        wl[s] = (s < S && i < np) ? (L.wbuf[1 + i] + eps) : 0.f;
        loc += wl[s];
    }
    const float total = seg_sum<LPR>(loc, lane);
    double ploc = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) { wl[s] = wl[s] / total; ploc += (double)wl[s]; }
    double run = seg_excl_sum_d<LPR>(ploc, l);
    if (l == 0) L.cdf[0] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int i = l * S + s;
        if (s < S && i < np) { run += (double)wl[s]; L.cdf[i + 1] = (float)run; }
    }
    sync();                                               // cdf complete; the weights in wbuf are dead

    // inverse cdf (models/volume_rendering.py:76-96): inds = searchsorted(cdf, u, right=True) = #{i : cdf[i] <= u}, by a
    // branch-free binary search (fixed trip count: the lane's searches interleave).  The sample lands in
    // [bins[below], bins[above]], bins = mid-points of the coarse depths, so the number of coarse depths <= it is
    // below + 1 or below + 2 — that is its rank in the merge, no second search.
    int top = 1;
    while (top * 2 <= nb) top *= 2;                       // (wave-uniform; a constant for the static shapes)
    float zfv[MAXS];
    int fpos[MAXS];
    bool ok = true;
    // The lane's searches go probe by probe TOGETHER: every probe is a round trip to LDS, and one search after the other
    // made a wavefront's chain twice as many of them as it needs.
    int lo[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) lo[f] = 0;
    constexpr bool FULL = KC > 1 && (KC & (KC - 1)) == 0;  // nb = 2^m - 1: lo + step never leaves the cdf, nothing to clamp
    const float* at[NF];                                   // FULL: &cdf[lo] itself is the search state (add, compare, select per probe)
#pragma unroll
    for (int f = 0; f < NF; ++f) at[f] = L.cdf;
    auto probe = [&](int f, int step) {
        const int idx = lo[f] + step;
        if (FULL) {
            at[f] = (at[f][step - 1] <= uu[f]) ? at[f] + step : at[f];
        } else {
            const float v = L.cdf[min(idx, nb) - 1];
            lo[f] = (idx <= nb && v <= uu[f]) ? idx : lo[f];
        }
    };
    if (KC) {                                             // constant trip count: unrolled
#pragma unroll
        for (int step = 128; step > 0; step >>= 1)
            if (step <= (KC ? KC - 1 : 1)) {
#pragma unroll
                for (int f = 0; f < NF; ++f) probe(f, step);
            }
    } else {
        for (int step = top; step > 0; step >>= 1) {
#pragma unroll
            for (int f = 0; f < NF; ++f) probe(f, step);
        }
    }
    if (FULL) {
#pragma unroll
        for (int f = 0; f < NF; ++f) lo[f] = (int)(at[f] - L.cdf);
    }
#pragma unroll
    for (int f = 0; f < MAXS; ++f) { zfv[f] = 0.f; fpos[f] = 0; }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int j = f * LPR + l;
        if (j < Kf) {
            const int below = max(lo[f] - 1, 0), above = min(lo[f], Kc - 2);
            const float c0 = L.cdf[below], c1 = L.cdf[above];
            const float zb = L.zall[below], zb1 = L.zall[below + 1];
            const float b0 = 0.5f * (zb + zb1), b1 = 0.5f * (L.zall[above] + L.zall[above + 1]);
            float den = c1 - c0;
            if (den < eps) den = 1.0f;
            const float zf = b0 + (uu[f] - c0) / den * (b1 - b0);
            const int cnt = below + 1 + (zb1 <= zf ? 1 : 0);
            // (what the shortcut assumes: zall[cnt - 1] <= zf < zall[cnt]; anything else takes the general path below)
            const float znext = L.zall[min(cnt, Kc - 1)];
            ok &= (zb <= zf) & ((cnt >= Kc) | (zf < znext));
            zfv[f] = zf; fpos[f] = j + cnt;
            L.zall[Kc + j] = zf;
            L.wbuf[j + cnt] = zf;                         // its place in the sorted row, if the row is regular
            atomicAdd(&L.hist[cnt], 1);
            if (z_fine_row != nullptr) z_fine_row[j] = zf;
        }
    }
    sync();

    // Stable sort of the Kc+Kf depths: rank(p) = #(y < x) + #(y == x, q < p).  The coarse half is normally ascending
    // (stratified depths) and the shortcut above gave every fine sample its number of coarse depths <= it: then a coarse
    // depth's rank is its index + the number of fine samples whose count does not exceed it — a prefix sum of the
    // histogram — and a fine sample's is its count + its rank AMONG THE FINE SAMPLES: its index when they ascend too
    // (ascending u through a monotone inverse cdf: inference), a Kf x Kf count otherwise (random u: training — 1/16 of
    // the all-pairs count at 64 + 32).  Anything else (a 1-ulp inversion at a bin edge, coinciding coarse depths) takes
    // the all-pairs count, which is valid for any input.
    const int K = Kc + Kf;
    bool okf = true;
    for (int p = l; p < K; p += LPR)
        if (p + 1 < K && p + 1 != Kc) {
            const bool asc = L.zall[p] <= L.zall[p + 1];
            if (p < Kc) ok &= asc; else okf &= asc;
        }
    // (the votes span the wavefront: with two rays per wavefront both take the slower path if either needs it)
    const bool regular = __all(ok);
    const bool fast = regular && __all(okf);
    PermT* perm = reinterpret_cast<PermT*>(L.cdf);        // the cdf is dead from here on (barrier after the sampling loop)
    if (regular) {
        if (!fast) {
#pragma unroll
            for (int f = 0; f < MAXS; ++f) {
                const int j = f * LPR + l;
                if (j < Kf) {
                    const float x = zfv[f];
                    int rank = 0;
                    for (int q = 0; q < Kf; ++q) {        // (every lane reads the same word: a broadcast)
                        const float y = L.zall[Kc + q];
                        rank += (y < x || (y == x && q < j)) ? 1 : 0;
                    }
                    fpos[f] = fpos[f] - j + rank;
                }
            }
        }
        const int SC = (Kc + LPR - 1) / LPR;              // coarse entries p = l*SC + s: a run per lane, then a DPP scan
        int h[MAXS], mine = 0;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int p = l * SC + s;
            h[s] = (s < SC && p < Kc) ? L.hist[p] : 0;
            mine += h[s];
        }
        int inc = mine;
        inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR1, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR2, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR4, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR8, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_BCAST15, 0xA, 0xf, false);
        if (LPR == 64) inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_BCAST31, 0xC, 0xf, false);
        int below_me = inc - mine;                        // fine samples counted by the lanes before this one
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int p = l * SC + s;
            if (s < SC && p < Kc) {
                below_me += h[s];
                L.wbuf[p + below_me] = L.zall[p];
                if (want_perm) perm[p + below_me] = (PermT)p;
            }
        }
        if (!fast) {                                      // (the sampling loop's stores assumed ascending fine depths)
#pragma unroll
            for (int f = 0; f < MAXS; ++f)
                if (f * LPR + l < Kf) L.wbuf[fpos[f]] = zfv[f];
        }
        if (want_perm) {
#pragma unroll
            for (int f = 0; f < MAXS; ++f)
                if (f * LPR + l < Kf) perm[fpos[f]] = (PermT)(Kc + f * LPR + l);
        }
    } else {                                              // (every slot of wbuf is rewritten)
        for (int p = l; p < K; p += LPR) {
            const float x = L.zall[p];
            int rank = 0;
            for (int q = 0; q < K; ++q) {
                const float y = L.zall[q];
                rank += (y < x || (y == x && q < p)) ? 1 : 0;
            }
            L.wbuf[rank] = x;
            if (want_perm) perm[rank] = (PermT)p;           // z_sorted[rank] = cat(z_coarse, z_fine)[p]
        }
    }
    sync();
}

}  // namespace anr
