// The small data-movement and loss steps of the training step (train.py:228-322) that sit between the big kernels:
// each replaces a chain of 5-40 framework launches of a few microseconds each by one launch.  All HBM-bound and tiny
// (tens of KB to a few MB); what they buy is launch count, not bandwidth.
//
//   anr_compact_ordered   inside_inds of models/anim_nerf.py:253 in sample ORDER (three launches: count, scan, gather) —
//                         replaces the ticket compaction + a radix sort of the index for the training path, where the row
//                         order fixes the summation order of every split-K weight gradient
//   anr_expand_rows       out[i] = src[pos[i]] or the fill row: the scatter back to all samples (and of dL/dx on the way back)
//   anr_mlp_head_grad     upstream gradient of (rgb, sigma) -> the g[n][4] operand of anr_mlp_backward (sigmoid', validity,
//                         gather to the compacted rows, zero padding rows)
//   anr_tangent_quads     xyz[n][3] -> the quads of rows the tangent-mode MLP kernels take (ANR_MLP_FLAG_TANGENT)
//   anr_sample_coarse_backward, anr_merge_backward   the depths' way back to near'/far' and through the sort (pose refinement)
//   anr_train_loss        every loss term of train.py:228-309 and the weighted total in one launch
//   anr_train_loss_backward  ... and its gradient w.r.t. every rendered / queried value in one launch
//   anr_adam_step         Adam (train.py:216-226: torch.optim.Adam, eps 1e-8, no weight decay) over every parameter tensor of
//                         the step in ONE launch, through a table of chunks; the step counter stays on the device
#include "anr_common.h"

namespace anr {

constexpr int CB = 1024;   // samples per compaction block

// riders: n_r = rows (n_fg + n_bg) extra points that count as samples n .. n + n_r - 1 with valid = 1 (the prior points of
// train.py:262-286 evaluated with the step's ray samples) — listed behind the samples, never dropped.  Rider j is point
// j % (n_fg + n_bg) of frame j / (n_fg + n_bg): the frame's foreground points fg[rows][n_fg][3], then its background points
// bg[rows][n_bg][3] (either may be absent) — the order anr_train_loss reads their sigmas in, without a concatenated copy.
struct Riders {
    const float *fg, *bg;
    int n_fg, n_bg, rows;
    __host__ __device__ int64_t count() const { return (int64_t)rows * (n_fg + n_bg); }
    __device__ float4 point(int64_t j) const {
        const int per = n_fg + n_bg, b = (int)(j / per), w = (int)(j % per);
        const float* p = w < n_fg ? fg + ((int64_t)b * n_fg + w) * 3 : bg + ((int64_t)b * n_bg + (w - n_fg)) * 3;
        return make_float4(p[0], p[1], p[2], 1.0f);
    }
};
__global__ __launch_bounds__(CB) void compact_count_kernel(const float4* __restrict__ pts, int64_t n, int64_t n_r,
                                                           int32_t* __restrict__ block_cnt) {
    __shared__ int wave_cnt[CB / WAVE];
    const int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x;
    const bool keep = (i < n && !(pts[i].w < 1.0f)) || (i >= n && i < n + n_r);
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < CB / WAVE; ++w) tot += wave_cnt[w];
        block_cnt[blockIdx.x] = tot;
    }
}

// exclusive scan of the block counts in place (one workgroup), total -> *count
__global__ __launch_bounds__(1024) void compact_scan_kernel(int32_t* __restrict__ block_cnt, int n_blocks, int32_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? block_cnt[i] : 0;
        int s = v;                                           // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < n_blocks) block_cnt[i] = off + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[0] = carry;
        count[1] = carry > 0 ? (carry + 63) / 64 * 64 : 64;      // ... and the row count the MLP kernels work on (padding rows: valid = 0)
    }
}

__global__ __launch_bounds__(CB) void compact_gather_kernel(const float4* __restrict__ pts, int64_t n, Riders riders,
                                                            int64_t n_r, const int32_t* __restrict__ block_base,
                                                            const int32_t* __restrict__ count, int32_t* __restrict__ index,
                                                            int32_t* __restrict__ pos, float4* __restrict__ pts_out) {
    __shared__ int wave_cnt[CB / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) p = pts[i];
    else if (i < n + n_r) p = riders.point(i - n);
    const bool keep = (i < n && !(p.w < 1.0f)) || (i >= n && i < n + n_r);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int r = off + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) { index[r] = (int32_t)i; pts_out[r] = p; }
    if (i < n + n_r) pos[i] = keep ? r : -1;
    if (blockIdx.x == 0 && threadIdx.x < 64) {               // padding rows up to the next multiple of 64: valid = 0
        const int c = *count, pad = (c + 63) / 64 * 64;
        if (c + (int)threadIdx.x < (pad > 0 ? pad : 64)) pts_out[c + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- the same list in ONE launch (round 5; the explicit training step calls it twice per step): a chained scan with
// decoupled look-back (Merrill & Garland 2016) over blocks of CB samples.  state[b] = (flag << 32) | value — flag 1: the
// block's own count, flag 2: the inclusive prefix up to and including it — is ONE 8-byte granule written by a relaxed
// agent-scope atomic store and read by relaxed agent-scope atomic loads (a valid cross-workgroup hand-off on gfx950 without
// fences: MI355X_MICROARCH.md, "8-B agent atomics both sides").  Blocks take their number from a ticket, so every
// predecessor a block waits for is already running.  The LAST block to finish puts state, tickets and all back to zero: the
// buffer must be zero before its first use and is left zero by every call (a replayed graph needs no fill launch).
// CSI samples per thread (block = CSI x CB consecutive samples, thread t takes t, t + CB, ...): the blocks' ticket — one
// same-address atomic each, ~25-50 ns apiece one behind the other — was the kernel's clock with a block per 1,024 samples
// (1,536 blocks: 47 us; 128: 12 us; the look-back, by one thread or by a wavefront, did not show).
constexpr int CSI = 4;
__global__ __launch_bounds__(CB) void compact_single_kernel(const float4* __restrict__ pts, int64_t n, Riders riders, int64_t n_r,
                                                            unsigned long long* __restrict__ state, int nb, int32_t* __restrict__ count,
                                                            int32_t* __restrict__ index, int32_t* __restrict__ pos,
                                                            float4* __restrict__ pts_out) {
    __shared__ int wave_cnt[CSI][CB / WAVE];
    __shared__ int sh_bid, sh_excl, sh_total;
    unsigned* tickets = reinterpret_cast<unsigned*>(state + nb);         // [0]: block numbers, [1]: blocks done
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sh_bid = (int)atomicAdd(&tickets[0], 1u);
    __syncthreads();
    const int bid = sh_bid;
    float4 p[CSI];
    bool keep[CSI];
    unsigned long long m[CSI];
#pragma unroll
    for (int it = 0; it < CSI; ++it) {                       // (every load of the thread first)
        const int64_t i = ((int64_t)bid * CSI + it) * CB + threadIdx.x;
        p[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) p[it] = pts[i];
        else if (i < n + n_r) p[it] = riders.point(i - n);
    }
#pragma unroll
    for (int it = 0; it < CSI; ++it) {
        const int64_t i = ((int64_t)bid * CSI + it) * CB + threadIdx.x;
        keep[it] = (i < n && !(p[it].w < 1.0f)) || (i >= n && i < n + n_r);
        m[it] = __ballot(keep[it]);
        if (lane == 0) wave_cnt[it][wave] = __popcll(m[it]);
    }
    __syncthreads();
    if (wave == 0) {
        // the look-back by a wavefront: lane l reads predecessor bid - 1 - l (spinning until it has published), the lanes up to
        // the first inclusive prefix add up
        int tot = 0;
#pragma unroll
        for (int it = 0; it < CSI; ++it)
#pragma unroll
            for (int w = 0; w < CB / WAVE; ++w) tot += wave_cnt[it][w];
        if (lane == 0 && bid > 0)
            __hip_atomic_store(&state[bid], (1ull << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int excl = 0;
        for (int base = bid - 1; base >= 0; base -= 64) {
            const int j = base - lane;
            unsigned long long sj = 0ull;
            if (j >= 0) {
                do { sj = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((sj >> 32) == 0);
            }
            const unsigned long long prefixes = __ballot(j >= 0 && (sj >> 32) == 2);
            const int first = prefixes ? __builtin_ctzll(prefixes) : 64;      // the nearest predecessor that knows its prefix
            int v = (j >= 0 && lane <= first) ? (int)(unsigned)sj : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            excl += v;
            if (prefixes) break;
        }
        if (lane == 0) {
            __hip_atomic_store(&state[bid], (2ull << 32) | (unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_excl = excl;
            sh_total = excl + tot;
            if (bid == nb - 1) {
                count[0] = excl + tot;
                count[1] = excl + tot > 0 ? (excl + tot + 63) / 64 * 64 : 64;   // the row count the MLP kernels work on (padding rows: valid = 0)
            }
        }
    }
    __syncthreads();
    int off = sh_excl;
#pragma unroll
    for (int it = 0; it < CSI; ++it) {                       // sample order = (it, wave, lane) order inside the block
        const int64_t i = ((int64_t)bid * CSI + it) * CB + threadIdx.x;
        int mine = off;
        for (int w = 0; w < wave; ++w) mine += wave_cnt[it][w];
        const int r = mine + __popcll(m[it] & ((1ull << lane) - 1ull));
        if (keep[it]) { index[r] = (int32_t)i; pts_out[r] = p[it]; }
        if (i < n + n_r) pos[i] = keep[it] ? r : -1;
#pragma unroll
        for (int w = 0; w < CB / WAVE; ++w) off += wave_cnt[it][w];
    }
    if (bid == nb - 1 && threadIdx.x < 64) {                 // padding rows up to the next multiple of 64: valid = 0
        const int c = sh_total, pad = (c + 63) / 64 * 64;
        if (c + (int)threadIdx.x < (pad > 0 ? pad : 64)) pts_out[c + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // every block has finished its look-back once it counts itself done: the last one clears the state for the next call
    __shared__ bool last;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&tickets[1], 1u) == (unsigned)nb - 1;
    __syncthreads();
    if (last) {
        for (int j = threadIdx.x; j <= nb; j += CB) __hip_atomic_store(&state[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int COLS>
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ pos, int64_t n,
                                                          float fill, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = pos[i];
    if (COLS == 4) {
        reinterpret_cast<float4*>(out)[i] = r >= 0 ? reinterpret_cast<const float4*>(src)[r] : make_float4(0.f, 0.f, 0.f, fill);
    } else {
        out[i] = r >= 0 ? src[r] : fill;
    }
}

__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ g, const int32_t* __restrict__ index,
                                                        const float4* __restrict__ out, const float4* __restrict__ pts, int64_t rows,
                                                        int64_t n_pad, int sigma_only, float4* __restrict__ g4,
                                                        const int32_t* __restrict__ count) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (count) {                                             // rows on the device: count[0] listed rows, count[1] padded
        rows = count[0];
        n_pad = count[1] < n_pad ? count[1] : n_pad;
    }
    if (r >= n_pad) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
        const int64_t s = index ? index[r] : r;
        const float valid = pts[r].w >= 1.0f ? 1.0f : 0.0f;      // sigma is the constant -1e5 where the sample is invalid
        if (sigma_only) v.w = g[s] * valid;
        else {
            const float4 u = reinterpret_cast<const float4*>(g)[s];
            const float4 o = out[r];
            v = make_float4(u.x * o.x * (1.0f - o.x), u.y * o.y * (1.0f - o.y), u.z * o.z * (1.0f - o.z), u.w * valid);   // sigmoid'
        }
    }
    g4[r] = v;
}

__global__ __launch_bounds__(256) void tangent_quads_kernel(const float* __restrict__ xyz, int64_t n, int64_t n_pad, float4* __restrict__ pts4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one row of a quad
    if (i >= 4 * n_pad) return;
    const int64_t p = i >> 2;
    pts4[i] = p < n ? make_float4(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// z_k = near' + (far' - near') e_k with e_k = the (jittered) step: d near' = sum g_k (1 - e_k), d far' = sum g_k e_k
// (models/volume_rendering.py:29-56; pose refinement moves near'/far' with the root transform).  One wavefront per ray:
// lane k <-> sample k, k + 64, ... (coalesced loads), the sums meet by shuffles.
__global__ __launch_bounds__(256) void sample_coarse_backward_kernel(const float* __restrict__ g, const float* __restrict__ steps,
                                                                     const float* __restrict__ t_rand, int64_t R, int K,
                                                                     float* __restrict__ d_rays) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float dn = 0.0f, df = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float s = steps[k];
        float e = s;
        if (t_rand) {
            const float lo = k > 0 ? 0.5f * (s + steps[k - 1]) : s, up = k + 1 < K ? 0.5f * (steps[k + 1] + s) : s;
            e = lo + (up - lo) * t_rand[r * K + k];
        }
        const float gk = g[r * K + k];
        dn += gk * (1.0f - e);
        df += gk * e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dn += __shfl_xor(dn, o, 64); df += __shfl_xor(df, o, 64); }
    if (lane == 0) {
        float4* o = reinterpret_cast<float4*>(d_rays + r * 8);
        o[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        o[1] = make_float4(0.f, 0.f, dn, df);
    }
}

// z_sorted[j] = cat(z_coarse, z_fine)[perm[j]], z_fine detached (models/volume_rendering.py:199-207): the gradient of the
// sorted depths goes back to the coarse ones through the permutation.
__global__ __launch_bounds__(256) void merge_backward_kernel(const float* __restrict__ g, const int32_t* __restrict__ perm, int64_t n, int K,
                                                             int Kc, float* __restrict__ d_zc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    if (p < Kc) d_zc[(i / K) * Kc + p] = g[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// losses
constexpr int LOSS_BLOCKS = 64, LOSS_TERMS = 10;

struct Normal { float x, y, z, r; };
// models/nerf.py:177-190 from a tangent quad (sigma, d sigma/dx, dy, dz): d alpha / d xyz = delta exp(-delta sigma) grad sigma
// where sigma > 0, then train.py:303: n / (|n| + 1e-5)
__device__ __forceinline__ Normal unit_normal(const float4 q, float delta, float& scale) {
    scale = q.x > 0.0f ? delta * expf(-delta * q.x) : 0.0f;
    Normal a{scale * q.y, scale * q.z, scale * q.w, 0.f};
    a.r = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return a;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {     // 256 threads; result valid in thread 0
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void train_loss_kernel(anr_loss_args a, float* __restrict__ partials, unsigned* __restrict__ ticket,
                                                         float* __restrict__ vals) {
    __shared__ float sh[4];
    __shared__ bool last;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)LOSS_BLOCKS * 256;
    float acc[LOSS_TERMS];
#pragma unroll
    for (int t = 0; t < LOSS_TERMS; ++t) acc[t] = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        const float* rgb = pass ? a.rgb_fine : a.rgb;
        const float* al = pass ? a.acc_fine : a.acc;
        const float* s = pass ? a.s_fine : a.s;
        if (s) {                                             // (the prior points' rows at the end of a compacted pass)
            const int32_t* sc = pass ? a.s_count_fine : a.s_count;
            if (sc) s += (int64_t)(a.s_stride > 0 ? a.s_stride : 1) * (sc[0] - a.prior_rows * (a.n_fg + a.n_bg));
        }
        const float4* q = reinterpret_cast<const float4*>(pass ? a.quads_fine : a.quads);
        float s_rgb = 0.f, s_al = 0.f, s_fg = 0.f, s_bg = 0.f, s_n = 0.f;
        if (rgb)
            for (int64_t i = tid; i < a.R * 3; i += nth) { const float d = rgb[i] - a.target_rgb[i]; s_rgb += d * d; }
        if (al)
            for (int64_t i = tid; i < a.R; i += nth) s_al += fabsf(al[i] - a.target_alpha[i]);
        if (s) {
            const int per = a.n_fg + a.n_bg;
            const int ss = a.s_stride > 0 ? a.s_stride : 1;
            for (int64_t i = tid; i < a.prior_rows * per; i += nth) {
                const float e = expf(a.k * fmaxf(s[i * ss], 0.0f));
                if ((int)(i % per) < a.n_fg) s_fg += e; else s_bg += 1.0f - e;
            }
        }
        if (q)
            for (int64_t p = tid; p < a.normal_sets * a.nv; p += nth) {
                const int64_t ia = (p / a.nv) * 2 * a.nv + p % a.nv;
                float sa, sb;
                const Normal u = unit_normal(q[ia], a.delta, sa), v = unit_normal(q[ia + a.nv], a.delta, sb);
                const float iu = 1.0f / (u.r + 1e-5f), iv = 1.0f / (v.r + 1e-5f);
                const float dx = u.x * iu - v.x * iv, dy = u.y * iu - v.y * iv, dz = u.z * iu - v.z * iv;
                s_n += dx * dx + dy * dy + dz * dz;
            }
        acc[0 + pass] = s_rgb; acc[2 + pass] = s_al; acc[4 + 2 * pass] = s_fg; acc[5 + 2 * pass] = s_bg; acc[8 + pass] = s_n;
    }
    // range of the target colours: train/psnr is torchmetrics' peak_signal_noise_ratio WITHOUT data_range (train.py:339-344),
    // i.e. data_range = target.max() - target.min() of the batch
    float t_hi = -INFINITY, t_lo = INFINITY;
    for (int64_t i = tid; i < a.R * 3; i += nth) { const float t = a.target_rgb[i]; t_hi = fmaxf(t_hi, t); t_lo = fminf(t_lo, t); }
#pragma unroll
    for (int t = 0; t < LOSS_TERMS; ++t) {
        const float v = block_sum(acc[t], sh);
        if (threadIdx.x == 0) partials[blockIdx.x * LOSS_TERMS + t] = v;
    }
    {
        __shared__ float sh_hi[4], sh_lo[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_hi = fmaxf(t_hi, __shfl_xor(t_hi, o, 64)); t_lo = fminf(t_lo, __shfl_xor(t_lo, o, 64)); }
        if ((threadIdx.x & 63) == 0) { sh_hi[threadIdx.x >> 6] = t_hi; sh_lo[threadIdx.x >> 6] = t_lo; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[LOSS_BLOCKS * LOSS_TERMS + 2 * blockIdx.x] = fmaxf(fmaxf(sh_hi[0], sh_hi[1]), fmaxf(sh_hi[2], sh_hi[3]));
            partials[LOSS_BLOCKS * LOSS_TERMS + 2 * blockIdx.x + 1] = fminf(fminf(sh_lo[0], sh_lo[1]), fminf(sh_lo[2], sh_lo[3]));
        }
    }
    // the last workgroup to finish adds the partial sums up (a fixed tree: same bits on every run) and resets the ticket.
    // Thread 0 wrote every partial of this workgroup: it alone releases them (agent scope, then the drained vmcnt hipcc may drop:
    // MI355X_MICROARCH.md) before it takes its ticket — 256 threads x 64 workgroups each issuing __threadfence() were a third of
    // this kernel's 33 us; the last workgroup acquires once, through one lane.
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = atomicAdd(ticket, 1u) == LOSS_BLOCKS - 1;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    // (lane b of a wavefront takes workgroup b's partial — LOSS_BLOCKS = 64 — and the lanes meet in a fixed butterfly: the same
    // bits on every run; one thread per term walking 64 partials one load behind the other took 40 of this kernel's 46 us)
    static_assert(LOSS_BLOCKS == 64, "one lane per workgroup's partial");
    __shared__ float s_val[LOSS_TERMS], s_hi, s_lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < LOSS_TERMS + 2; t += 4) {
        if (t < LOSS_TERMS) {
            float v = __builtin_nontemporal_load(&partials[lane * LOSS_TERMS + t]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const float cnt = t < 2 ? (float)(a.R * 3) : t < 4 ? (float)a.R
                            : t < 8 ? (float)(a.prior_rows * ((t & 1) ? a.n_bg : a.n_fg)) : (float)(a.normal_sets * a.nv * 3);
            if (lane == 0) s_val[t] = cnt > 0.0f ? v / cnt : 0.0f;
        } else {
            // range of the targets: train/psnr is torchmetrics' peak_signal_noise_ratio without data_range (train.py:339-344)
            float v = __builtin_nontemporal_load(&partials[LOSS_BLOCKS * LOSS_TERMS + 2 * lane + (t - LOSS_TERMS)]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_xor(v, o, 64); v = t == LOSS_TERMS ? fmaxf(v, u) : fminf(v, u); }
            if (lane == 0) { if (t == LOSS_TERMS) s_hi = v; else s_lo = v; }
        }
    }
    __syncthreads();
    if (threadIdx.x < LOSS_TERMS) vals[threadIdx.x] = s_val[threadIdx.x];
    if (threadIdx.x == 0) {
        const float w[LOSS_TERMS] = {1.f, 1.f, a.lambda_alphas, a.lambda_alphas, a.lambda_foreground, a.lambda_background,
                                     a.lambda_foreground, a.lambda_background, a.lambda_normals, a.lambda_normals};
        float tot = 0.0f;
        for (int t = 0; t < LOSS_TERMS; ++t) tot += w[t] * s_val[t];
        vals[LOSS_TERMS] = tot;
        // PSNR of the rendered batch (fine pass if there is one): 10 log10((max(target) - min(target))^2 / mse)
        const float range = s_hi - s_lo;
        vals[LOSS_TERMS + 1] = 10.0f * log10f(range * range / (a.rgb_fine ? s_val[1] : s_val[0]));
        *ticket = 0u;
    }
}

__global__ __launch_bounds__(256) void train_loss_backward_kernel(anr_loss_args a, const float* __restrict__ g_total, anr_loss_grads d) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    const float g = *g_total;
    for (int pass = 0; pass < 2; ++pass) {
        const float* rgb = pass ? a.rgb_fine : a.rgb;
        const float* al = pass ? a.acc_fine : a.acc;
        const float* s = pass ? a.s_fine : a.s;
        if (s) {
            const int32_t* sc = pass ? a.s_count_fine : a.s_count;
            if (sc) s += (int64_t)(a.s_stride > 0 ? a.s_stride : 1) * (sc[0] - a.prior_rows * (a.n_fg + a.n_bg));
        }
        const float4* q = reinterpret_cast<const float4*>(pass ? a.quads_fine : a.quads);
        float* d_rgb = pass ? d.rgb_fine : d.rgb;
        float* d_al = pass ? d.acc_fine : d.acc;
        float* d_s = pass ? d.s_fine : d.s;
        float4* d_q = reinterpret_cast<float4*>(pass ? d.quads_fine : d.quads);
        if (rgb && d_rgb) {
            const float c = g * 2.0f / (float)(a.R * 3);
            for (int64_t i = tid; i < a.R * 3; i += nth) d_rgb[i] = c * (rgb[i] - a.target_rgb[i]);
        }
        if (al && d_al) {
            const float c = g * a.lambda_alphas / (float)a.R;
            for (int64_t i = tid; i < a.R; i += nth) {
                const float e = al[i] - a.target_alpha[i];
                d_al[i] = e > 0.0f ? c : e < 0.0f ? -c : 0.0f;
            }
        }
        if (s && d_s) {
            const int per = a.n_fg + a.n_bg;
            const float cf = a.n_fg ? g * a.lambda_foreground * a.k / (float)(a.prior_rows * a.n_fg) : 0.0f;
            const float cb = a.n_bg ? -g * a.lambda_background * a.k / (float)(a.prior_rows * a.n_bg) : 0.0f;
            const int ss = a.s_stride > 0 ? a.s_stride : 1;
            const int32_t* sc = pass ? a.s_count_fine : a.s_count;
            const int64_t row_shift = (a.s_grad_rows && sc) ? (int64_t)sc[0] - a.prior_rows * per : 0;
            for (int64_t i = tid; i < a.prior_rows * per; i += nth) {
                const float v = s[i * ss];
                const float dv = v > 0.0f ? ((int)(i % per) < a.n_fg ? cf : cb) * expf(a.k * v) : 0.0f;
                // s_stride 4: the sigmas are column 3 of (r, g, b, sigma) rows and so are their gradients (d_s points at row 0's r)
                // (s_grad_rows: the gradient rows are the prior points' rows of the compacted pass, like the sigmas themselves)
                if (ss == 4) reinterpret_cast<float4*>(d_s)[i + row_shift] = make_float4(0.f, 0.f, 0.f, dv);
                else d_s[(i + row_shift) * ss] = dv;
            }
        }
        if (q && d_q) {
            const float c = g * a.lambda_normals * 2.0f / (float)(a.normal_sets * a.nv * 3);
            for (int64_t p = tid; p < a.normal_sets * a.nv; p += nth) {
                const int64_t ia = (p / a.nv) * 2 * a.nv + p % a.nv, ib = ia + a.nv;
                const float4 qa = q[ia], qb = q[ib];
                float sa, sb;
                const Normal u = unit_normal(qa, a.delta, sa), v = unit_normal(qb, a.delta, sb);
                const float iu = 1.0f / (u.r + 1e-5f), iv = 1.0f / (v.r + 1e-5f);
                const float dx = c * (u.x * iu - v.x * iv), dy = c * (u.y * iu - v.y * iv), dz = c * (u.z * iu - v.z * iv);
                // d (n / (|n| + eps)) : I / (|n| + eps) - n n^T / (|n| (|n| + eps)^2); the norm's subgradient at 0 is 0
                auto through = [&](const Normal& w, float iw, float sgn, float scale, const float4 qq) -> float4 {
                    const float dot = w.x * dx + w.y * dy + w.z * dz;
                    const float k2 = w.r > 0.0f ? dot * iw * iw / w.r : 0.0f;
                    const float gx = sgn * (dx * iw - w.x * k2), gy = sgn * (dy * iw - w.y * k2), gz = sgn * (dz * iw - w.z * k2);
                    // n = scale(sigma) * grad sigma;  d scale / d sigma = -delta * scale
                    return make_float4(-a.delta * scale * (gx * qq.y + gy * qq.z + gz * qq.w), scale * gx, scale * gy, scale * gz);
                };
                const float4 ga = through(u, iu, 1.0f, sa, qa), gb = through(v, iv, -1.0f, sb, qb);
                if (a.quad_grad_rows) {
                    // ... as the g operand of the tangent-mode backward: row 4 p + q = (0, 0, 0, d (sigma | d/dx | d/dy | d/dz))
                    // — what anr_mlp_head_grad (sigma only) makes of the quad's gradient, without that launch
                    d_q[4 * ia + 0] = make_float4(0.f, 0.f, 0.f, ga.x); d_q[4 * ia + 1] = make_float4(0.f, 0.f, 0.f, ga.y);
                    d_q[4 * ia + 2] = make_float4(0.f, 0.f, 0.f, ga.z); d_q[4 * ia + 3] = make_float4(0.f, 0.f, 0.f, ga.w);
                    d_q[4 * ib + 0] = make_float4(0.f, 0.f, 0.f, gb.x); d_q[4 * ib + 1] = make_float4(0.f, 0.f, 0.f, gb.y);
                    d_q[4 * ib + 2] = make_float4(0.f, 0.f, 0.f, gb.z); d_q[4 * ib + 3] = make_float4(0.f, 0.f, 0.f, gb.w);
                } else {
                    d_q[ia] = ga;
                    d_q[ib] = gb;
                }
            }
            const int64_t per = a.quad_grad_rows ? 4 : 1;
            for (int64_t p = per * a.normal_sets * 2 * a.nv + tid; p < per * a.quad_rows; p += nth) d_q[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

}  // namespace anr

using namespace anr;

extern "C" int64_t anr_compact_ws_ints(int64_t n) { return (n + CB - 1) / CB; }

extern "C" int anr_compact_ordered_riders(const float* pts, int64_t n, const float* fg, int n_fg, const float* bg, int n_bg, int rows,
                                          int32_t* index_out, int32_t* pos_out, float* pts_out, int32_t* count_out, int32_t* workspace,
                                          void* stream) {
    ANR_REQUIRE(pts && index_out && pos_out && pts_out && count_out && workspace, ANR_E_BADARG, "anr_compact_ordered: null pointer");
    ANR_REQUIRE(n_fg >= 0 && n_bg >= 0 && rows >= 0 && (n_fg == 0 || fg) && (n_bg == 0 || bg), ANR_E_BADARG,
                "anr_compact_ordered: rider points without their array");
    const Riders riders{fg, bg, n_fg, n_bg, rows};
    const int64_t n_r = riders.count();
    ANR_REQUIRE(n > 0 && n + n_r < (int64_t)1 << 31, ANR_E_BADARG, "anr_compact_ordered: n=%lld riders=%lld", (long long)n, (long long)n_r);
    ANR_REQUIRE((((uintptr_t)pts | (uintptr_t)pts_out) & 15) == 0, ANR_E_ALIGN, "anr_compact_ordered: pts / pts_out must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((n + n_r + CB - 1) / CB);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(CB), 0, st, reinterpret_cast<const float4*>(pts), n, n_r, workspace);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, workspace, nb, count_out);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(nb), dim3(CB), 0, st, reinterpret_cast<const float4*>(pts), n, riders, n_r, workspace,
                       count_out, index_out, pos_out, reinterpret_cast<float4*>(pts_out));
    return check_launch("anr_compact_ordered");
}

extern "C" int64_t anr_compact_state_words(int64_t n) { return (n + CB - 1) / CB + 1; }

extern "C" int anr_compact_ordered_single(const float* pts, int64_t n, const float* fg, int n_fg, const float* bg, int n_bg, int rows,
                                          int32_t* index_out, int32_t* pos_out, float* pts_out, int32_t* count_out, int64_t* state,
                                          void* stream) {
    ANR_REQUIRE(pts && index_out && pos_out && pts_out && count_out && state, ANR_E_BADARG, "anr_compact_ordered_single: null pointer");
    ANR_REQUIRE(n_fg >= 0 && n_bg >= 0 && rows >= 0 && (n_fg == 0 || fg) && (n_bg == 0 || bg), ANR_E_BADARG,
                "anr_compact_ordered_single: rider points without their array");
    const Riders riders{fg, bg, n_fg, n_bg, rows};
    const int64_t n_r = riders.count();
    ANR_REQUIRE(n > 0 && n + n_r < (int64_t)1 << 31, ANR_E_BADARG, "anr_compact_ordered_single: n=%lld riders=%lld", (long long)n, (long long)n_r);
    ANR_REQUIRE((((uintptr_t)pts | (uintptr_t)pts_out) & 15) == 0 && ((uintptr_t)state & 7) == 0, ANR_E_ALIGN,
                "anr_compact_ordered_single: pts / pts_out must be 16-B aligned, state 8-B");
    const int nb = (int)((n + n_r + (int64_t)CSI * CB - 1) / ((int64_t)CSI * CB));
    hipLaunchKernelGGL(compact_single_kernel, dim3(nb), dim3(CB), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(pts), n, riders, n_r,
                       reinterpret_cast<unsigned long long*>(state), nb, count_out, index_out, pos_out, reinterpret_cast<float4*>(pts_out));
    return check_launch("anr_compact_ordered_single");
}

extern "C" int anr_compact_ordered(const float* pts, int64_t n, int32_t* index_out, int32_t* pos_out, float* pts_out,
                                   int32_t* count_out, int32_t* workspace, void* stream) {
    return anr_compact_ordered_riders(pts, n, nullptr, 0, nullptr, 0, 0, index_out, pos_out, pts_out, count_out, workspace, stream);
}

extern "C" int anr_expand_rows(const float* src, const int32_t* pos, int64_t n, int cols, float fill, float* out, void* stream) {
    ANR_REQUIRE(src && pos && out, ANR_E_BADARG, "anr_expand_rows: null pointer");
    ANR_REQUIRE(n > 0 && (cols == 1 || cols == 4), ANR_E_BADARG, "anr_expand_rows: n=%lld cols=%d (1 or 4)", (long long)n, cols);
    ANR_REQUIRE(cols == 1 || (((uintptr_t)src | (uintptr_t)out) & 15) == 0, ANR_E_ALIGN, "anr_expand_rows: src/out must be 16-B aligned");
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (cols == 4) hipLaunchKernelGGL(expand_rows_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, pos, n, fill, out);
    else hipLaunchKernelGGL(expand_rows_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, pos, n, fill, out);
    return check_launch("anr_expand_rows");
}

static int head_grad(const float* g, const int32_t* index, const float* out, const float* pts, int64_t rows, int64_t n_pad,
                     int sigma_only, float* g4_out, const int32_t* count, void* stream) {
    ANR_REQUIRE(g && pts && g4_out && (sigma_only || out), ANR_E_BADARG, "anr_mlp_head_grad: null pointer");
    ANR_REQUIRE(rows >= 0 && n_pad >= rows && n_pad > 0, ANR_E_BADARG, "anr_mlp_head_grad: rows=%lld n_pad=%lld", (long long)rows, (long long)n_pad);
    ANR_REQUIRE((((uintptr_t)g4_out | (uintptr_t)pts | (uintptr_t)out | (sigma_only ? 0 : (uintptr_t)g)) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_head_grad: 16-B alignment");
    hipLaunchKernelGGL(head_grad_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, index,
                       reinterpret_cast<const float4*>(out), reinterpret_cast<const float4*>(pts), rows, n_pad, sigma_only,
                       reinterpret_cast<float4*>(g4_out), count);
    return check_launch("anr_mlp_head_grad");
}
extern "C" int anr_mlp_head_grad(const float* g, const int32_t* index, const float* out, const float* pts, int64_t rows, int64_t n_pad,
                                 int sigma_only, float* g4_out, void* stream) {
    return head_grad(g, index, out, pts, rows, n_pad, sigma_only, g4_out, nullptr, stream);
}
extern "C" int anr_mlp_head_grad_counted(const float* g, const int32_t* index, const float* out, const float* pts, const int32_t* count,
                                         int64_t n_alloc, int sigma_only, float* g4_out, void* stream) {
    ANR_REQUIRE(count, ANR_E_BADARG, "anr_mlp_head_grad_counted: null count");
    return head_grad(g, index, out, pts, 0, n_alloc, sigma_only, g4_out, count, stream);
}

extern "C" int anr_tangent_quads(const float* xyz, int64_t n, int64_t n_pad, float* pts4_out, void* stream) {
    ANR_REQUIRE(xyz && pts4_out, ANR_E_BADARG, "anr_tangent_quads: null pointer");
    ANR_REQUIRE(n > 0 && n_pad >= n, ANR_E_BADARG, "anr_tangent_quads: n=%lld n_pad=%lld", (long long)n, (long long)n_pad);
    hipLaunchKernelGGL(tangent_quads_kernel, dim3((unsigned)((4 * n_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n, n_pad,
                       reinterpret_cast<float4*>(pts4_out));
    return check_launch("anr_tangent_quads");
}

extern "C" int64_t anr_train_loss_ws_floats(void) { return LOSS_BLOCKS * (LOSS_TERMS + 2) + 4; }

static int check_loss_args(const anr_loss_args* a, const char* who) {
    ANR_REQUIRE(a, ANR_E_BADARG, "%s: null args", who);
    ANR_REQUIRE(a->R >= 0 && a->prior_rows >= 0 && a->n_fg >= 0 && a->n_bg >= 0 && a->nv >= 0 && a->normal_sets >= 0, ANR_E_BADARG,
                "%s: negative size", who);
    ANR_REQUIRE(!(a->rgb || a->acc || a->rgb_fine || a->acc_fine) || (a->target_rgb && a->target_alpha && a->R > 0), ANR_E_BADARG,
                "%s: rendered values without targets", who);
    ANR_REQUIRE(!(a->s || a->s_fine) || (a->prior_rows > 0 && a->n_fg + a->n_bg > 0), ANR_E_BADARG, "%s: prior sizes", who);
    ANR_REQUIRE(!(a->quads || a->quads_fine) || (a->nv > 0 && a->normal_sets > 0 && a->quad_rows >= 2 * a->nv * a->normal_sets),
                ANR_E_BADARG, "%s: normal sizes", who);
    ANR_REQUIRE((((uintptr_t)a->quads | (uintptr_t)a->quads_fine) & 15) == 0, ANR_E_ALIGN, "%s: quads must be 16-B aligned", who);
    return 0;
}

extern "C" int anr_train_loss(const anr_loss_args* a, float* workspace, float* vals_out, void* stream) {
    if (int rc = check_loss_args(a, "anr_train_loss")) return rc;
    ANR_REQUIRE(workspace && vals_out, ANR_E_BADARG, "anr_train_loss: null pointer");
    hipLaunchKernelGGL(train_loss_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, *a, workspace + 4,
                       reinterpret_cast<unsigned*>(workspace), vals_out);
    return check_launch("anr_train_loss");
}

extern "C" int anr_train_loss_backward(const anr_loss_args* a, const float* g_total, const anr_loss_grads* d, void* stream) {
    if (int rc = check_loss_args(a, "anr_train_loss_backward")) return rc;
    ANR_REQUIRE(g_total && d, ANR_E_BADARG, "anr_train_loss_backward: null pointer");
    ANR_REQUIRE((((uintptr_t)d->quads | (uintptr_t)d->quads_fine) & 15) == 0, ANR_E_ALIGN, "anr_train_loss_backward: quads must be 16-B aligned");
    hipLaunchKernelGGL(train_loss_backward_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, *a, g_total, *d);
    return check_launch("anr_train_loss_backward");
}

extern "C" int anr_sample_coarse_backward(const float* g_z, const float* steps, const float* t_rand, int64_t R, int K, float* d_rays_out,
                                          void* stream) {
    ANR_REQUIRE(g_z && steps && d_rays_out, ANR_E_BADARG, "anr_sample_coarse_backward: null pointer");
    ANR_REQUIRE(R > 0 && K > 0, ANR_E_BADARG, "anr_sample_coarse_backward: R=%lld K=%d", (long long)R, K);
    ANR_REQUIRE(((uintptr_t)d_rays_out & 15) == 0, ANR_E_ALIGN, "anr_sample_coarse_backward: d_rays_out must be 16-B aligned");
    hipLaunchKernelGGL(sample_coarse_backward_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g_z, steps,
                       t_rand, R, K, d_rays_out);
    return check_launch("anr_sample_coarse_backward");
}

extern "C" int anr_merge_backward(const float* g_sorted, const int32_t* perm, int64_t R, int K, int Kc, float* d_z_coarse_out, void* stream) {
    ANR_REQUIRE(g_sorted && perm && d_z_coarse_out, ANR_E_BADARG, "anr_merge_backward: null pointer");
    ANR_REQUIRE(R > 0 && Kc > 0 && K >= Kc, ANR_E_BADARG, "anr_merge_backward: R=%lld K=%d Kc=%d", (long long)R, K, Kc);
    const int64_t n = R * K;
    hipLaunchKernelGGL(merge_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_sorted, perm, n, K, Kc,
                       d_z_coarse_out);
    return check_launch("anr_merge_backward");
}

// ---------------------------------------------------------------------------------------------
// Adam over a table of chunks: chunk c = {param, grad, exp_avg, exp_avg_sq, count <= ADAM_CHUNK, group | tensor << 8}.  46 tensors of a
// few hundred to 80 k floats each are one launch of ~300 workgroups instead of a multi-tensor apply per parameter group
// (0.21 ms per step -> 0.02).  The arithmetic is torch.optim.Adam's (amsgrad off, maximize off, weight_decay 0):
//   m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t = steps[tensor], one counter per tensor as in torch (a tensor that gets its first gradient late starts at 1), read from
// the device (the caller increments the counters before the launch: a captured step replays with the right counts).
namespace anr {
constexpr int ADAM_CHUNK = 4096;
struct AdamChunk { float* p; const float* g; float* m; float* v; int32_t count; int32_t group; };
struct AdamHyper { float lr[4]; double beta1, beta2; float eps; };

// active / ticket (anr_adam_step_counting): `step` holds the counts BEFORE this update — the kernel works with step + 1 and the
// last workgroup to finish adds active[i] to every counter (every workgroup has read its counter by then), so the step needs
// no separate increment launch
__global__ __launch_bounds__(256) void adam_kernel(const AdamChunk* __restrict__ chunks, float* __restrict__ step, AdamHyper h,
                                                   const float* __restrict__ active, int n_tensors, unsigned* __restrict__ ticket) {
    const AdamChunk c = chunks[blockIdx.x];
    const float t = step[c.group >> 8] + (active ? 1.0f : 0.0f);
    // (1 - beta and the bias corrections in double, as the host-side arithmetic of torch.optim.Adam: 1 - 0.999f is off by 1e-5)
    const float omb1 = (float)(1.0 - h.beta1), omb2 = (float)(1.0 - h.beta2), b2 = (float)h.beta2;
    const float bc1 = (float)(1.0 - pow(h.beta1, (double)t)), bc2_sqrt = (float)sqrt(1.0 - pow(h.beta2, (double)t));
    const float step_size = h.lr[c.group & 0xff] / bc1;
    for (int i = threadIdx.x * 4; i < c.count; i += 256 * 4) {
        if (i + 4 <= c.count && ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0)) {
            float4 p = *reinterpret_cast<const float4*>(c.p + i), m = *reinterpret_cast<const float4*>(c.m + i);
            float4 v = *reinterpret_cast<const float4*>(c.v + i);
            const float4 g = *reinterpret_cast<const float4*>(c.g + i);
            float* pp = &p.x; float* mm = &m.x; float* vv = &v.x; const float* gg = &g.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[k] = mm[k] + (gg[k] - mm[k]) * omb1;
                vv[k] = b2 * vv[k] + omb2 * gg[k] * gg[k];
                pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) / bc2_sqrt + h.eps);
            }
            *reinterpret_cast<float4*>(c.p + i) = p;
            *reinterpret_cast<float4*>(c.m + i) = m;
            *reinterpret_cast<float4*>(c.v + i) = v;
        } else {
            for (int k = i; k < min(i + 4, c.count); ++k) {
                const float g = c.g[k];
                const float m = c.m[k] + (g - c.m[k]) * omb1;
                const float v = b2 * c.v[k] + omb2 * g * g;
                c.m[k] = m; c.v[k] = v;
                c.p[k] -= step_size * m / (sqrtf(v) / bc2_sqrt + h.eps);
            }
        }
    }
    if (active == nullptr) return;
    __shared__ bool last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No fence in front of the ticket (round 6): the last workgroup reads nothing the others wrote — it only bumps the step
        // counters, which every workgroup has read AND consumed (its bias corrections fed the stores above) before its thread 0
        // gets here.  __threadfence() made each of the ~290 workgroups write the L2 back and invalidate it (buffer_wbl2 /
        // buffer_inv sc1): most of the launch's 30 us.
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    for (int i = threadIdx.x; i < n_tensors; i += 256) step[i] += active[i];
    if (threadIdx.x == 0) *ticket = 0u;
}
}  // namespace anr

extern "C" int anr_adam_chunk_floats(void) { return anr::ADAM_CHUNK; }
extern "C" int anr_adam_chunk_bytes(void) { return (int)sizeof(anr::AdamChunk); }

extern "C" int anr_adam_step(const void* chunks, int n_chunks, const float* step, const float* lr, int n_groups, double beta1, double beta2,
                             double eps, void* stream) {
    ANR_REQUIRE(chunks && step && lr, ANR_E_BADARG, "anr_adam_step: null pointer");
    ANR_REQUIRE(n_chunks > 0 && n_groups >= 1 && n_groups <= 4, ANR_E_BADARG, "anr_adam_step: n_chunks=%d n_groups=%d (1..4)", n_chunks, n_groups);
    ANR_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, ANR_E_BADARG, "anr_adam_step: betas=(%g, %g) eps=%g",
                beta1, beta2, eps);
    anr::AdamHyper h;
    for (int g = 0; g < 4; ++g) h.lr[g] = g < n_groups ? lr[g] : 0.f;
    h.beta1 = beta1; h.beta2 = beta2; h.eps = (float)eps;
    hipLaunchKernelGGL(anr::adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const anr::AdamChunk*>(chunks), const_cast<float*>(step), h, (const float*)nullptr, 0, (unsigned*)nullptr);
    return anr::check_launch("anr_adam_step");
}

extern "C" int anr_adam_step_counting(const void* chunks, int n_chunks, float* step, const float* active, int n_tensors, int32_t* ticket,
                                      const float* lr, int n_groups, double beta1, double beta2, double eps, void* stream) {
    ANR_REQUIRE(chunks && step && lr && active && ticket, ANR_E_BADARG, "anr_adam_step_counting: null pointer");
    ANR_REQUIRE(n_chunks > 0 && n_tensors > 0 && n_groups >= 1 && n_groups <= 4, ANR_E_BADARG, "anr_adam_step_counting: n_chunks=%d n_groups=%d (1..4)",
                n_chunks, n_groups);
    ANR_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, ANR_E_BADARG, "anr_adam_step_counting: betas=(%g, %g) eps=%g",
                beta1, beta2, eps);
    anr::AdamHyper h;
    for (int g = 0; g < 4; ++g) h.lr[g] = g < n_groups ? lr[g] : 0.f;
    h.beta1 = beta1; h.beta2 = beta2; h.eps = (float)eps;
    hipLaunchKernelGGL(anr::adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const anr::AdamChunk*>(chunks), step, h, active, n_tensors, reinterpret_cast<unsigned*>(ticket));
    return anr::check_launch("anr_adam_step_counting");
}
