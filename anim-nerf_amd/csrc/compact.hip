// Index of the samples the field has to be evaluated at (valid >= 1) — the `inside_inds` of the reference's
// query_canonical_space_inside (models/anim_nerf.py:245-290): with use_unpose=True only samples within dis_threshold of
// the body carry a density, everything else is sigma = -1e5 (models/anim_nerf.py:305) and gets composited with weight
// exactly 0.  On the synthetic frames 6-7 % of the coarse samples are valid, so the MLP runs on the compacted list.
//
// HBM-bound: reads 16 B per sample, writes 4 B per valid sample and (optionally) 16 B or 4 B per invalid one.
#include "anr_common.h"

namespace anr {

constexpr int COMPACT_THREADS = 1024;
constexpr int COMPACT_WAVES = COMPACT_THREADS / WAVE;

// Order inside a 4,096-sample block is preserved; blocks claim their output range with one atomic, so the order of the
// blocks in `index` depends on scheduling.  Every MLP column is computed independently of its neighbours, so the
// rendered result does not depend on that order (tests/test_gpu_parity.py::test_compaction_is_bit_identical).
// Four consecutive samples per thread and trip: with one, the pass made a trip to its counter per 1,024 samples — 11,400
// same-address atomics at ~12 ns each were the 0.14 ms it took on the 11.7 M live voxels of a 512^3 grid.
constexpr int COMPACT_PER = 4;
__global__ __launch_bounds__(COMPACT_THREADS) void compact_valid_kernel(const float4* __restrict__ pts, int64_t n,
                                                                        int32_t* __restrict__ index,
                                                                        int32_t* __restrict__ count,
                                                                        float* __restrict__ fill, int fill_cols) {
    __shared__ int wave_cnt[COMPACT_WAVES];
    __shared__ int block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_blocks = (n + COMPACT_PER * COMPACT_THREADS - 1) / (COMPACT_PER * COMPACT_THREADS);
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t i0 = (b * COMPACT_THREADS + threadIdx.x) * COMPACT_PER;
        float w[COMPACT_PER];
#pragma unroll
        for (int k = 0; k < COMPACT_PER; ++k) w[k] = i0 + k < n ? pts[i0 + k].w : 0.0f;
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < COMPACT_PER; ++k) {
            if (i0 + k >= n) continue;
            if (!(w[k] < 1.0f)) {
                keep |= 1u << k;
            } else if (fill) {
                if (fill_cols == 4) reinterpret_cast<float4*>(fill)[i0 + k] = make_float4(0.f, 0.f, 0.f, -1e5f);
                else fill[i0 + k] = -1e5f;
            }
        }
        const int mine = __popc(keep);
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
            for (int v = 0; v < COMPACT_WAVES; ++v) { int c = wave_cnt[v]; wave_cnt[v] = tot; tot += c; }
            block_base = tot ? atomicAdd(count, tot) : 0;
        }
        __syncthreads();
        int pos = block_base + wave_cnt[wave] + incl - mine;
#pragma unroll
        for (int k = 0; k < COMPACT_PER; ++k)
            if ((keep >> k) & 1u) index[pos++] = (int32_t)(i0 + k);
        __syncthreads();
    }
}

}  // namespace anr

using namespace anr;

extern "C" int anr_compact_valid(const float* pts, int64_t n, int32_t* index_out, int32_t* count_out, float* fill_out,
                                 int fill_cols, void* stream) {
    ANR_REQUIRE(pts && index_out && count_out, ANR_E_BADARG, "anr_compact_valid: null pointer");
    ANR_REQUIRE(n > 0 && n < (int64_t)1 << 31, ANR_E_BADARG, "anr_compact_valid: n=%lld", (long long)n);
    ANR_REQUIRE(!fill_out || fill_cols == 4 || fill_cols == 1, ANR_E_BADARG, "anr_compact_valid: fill_cols=%d (1 or 4)", fill_cols);
    ANR_REQUIRE((((uintptr_t)pts | (uintptr_t)fill_out) & 15) == 0, ANR_E_ALIGN, "anr_compact_valid: pts/fill_out must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = zero_fill(count_out, sizeof(int32_t), st, "anr_compact_valid (zero)")) return rc;
    const int64_t n_blocks = (n + COMPACT_PER * COMPACT_THREADS - 1) / (COMPACT_PER * COMPACT_THREADS);
    const unsigned grid = (unsigned)(n_blocks < 2048 ? n_blocks : 2048);
    hipLaunchKernelGGL(compact_valid_kernel, dim3(grid), dim3(COMPACT_THREADS), 0, st,
                       reinterpret_cast<const float4*>(pts), n, index_out, count_out, fill_out, fill_cols);
    return check_launch("anr_compact_valid");
}
