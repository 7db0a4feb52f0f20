// Weight gradients of the fused MLP (training, a16): what autograd computes for every nn.Linear of
// models/nerf.py:129-175 —  dW_l = dact_l^T . in_l,  db_l = column sums of dact_l — from the two blocked buffers
// (76 blocks of [points][32 features], mlp_core.h) the training forward (anr_mlp_forward_save: activations) and the
// activation-gradient kernel (anr_mlp_backward: pre-activation gradients) leave behind, plus the encoding matrix
// enc[points][64].
//
// The contraction runs over the POINTS, i.e. over the slow index of both operands, so an MFMA fragment (one feature,
// 8 consecutive points per lane) is a column walk.  gfx950's transposing LDS read does it in hardware: for a slab of
// rows, each 32-feature block goes L2 -> LDS in 1-KiB pieces of 16 rows x its four 16-byte piece arrays (LDS-DMA, 16 B per lane
// with per-lane source addresses: the blocked buffers of mlp_core.h; inside a piece array a's copy of row r sits at position
// 16 a + (r + 4 a) mod 16, so the 4 rows x 4 arrays one ds_read_b64_tr_b16 touches per half-wave fill the 64 banks exactly), and
// each fragment is two ds_read_b64_tr_b16 (lane s of a 16-lane group points at row k0 + s/4, features f0 + 4 (s%4); it
// receives feature f0 + s of rows k0 .. k0+3).  fp32 (parity mode): v_mfma_f32_32x32x2_f32 takes one point per
// half-wave and its fragment is a plain row read.
//
// Work split: a task = (one GEMM, one slice of the points); 8 wavefronts share the staged slabs and own a
// (TM x TN) block of 32x32 accumulator tiles each (256x256: 2x4 per wave, 128 accumulator registers).  Partial products
// go to a workspace and a second kernel adds the slices in a fixed order (deterministic, no float atomics) while
// scattering into the PyTorch [out][in] layouts; the same kernel pair produces the bias gradients and the two
// skinny heads (sigma 1x256, rgb 3x128) as weighted column sums.
#include "mlp_core.h"

namespace anr {

constexpr int WG_WAVES = 8;
constexpr int WG_THREADS = WG_WAVES * 64;
constexpr int WG_NBUF = 3;
constexpr int WG_MAX_GEMMS = 8;

struct WgradGemm {
    int a_col;        // first column of dact (the M = out-feature side)
    int b_src;        // 0: act, 1: enc
    int b_col;        // first column of the source (the N = in-feature side)
    int out_off;      // float offset of this GEMM's [S][M*N + M] block in the partial workspace
    int want_bias;    // also emit the column sums of the M side (the bias gradient) behind each [M][N] slice
};
struct WgradArgs {
    WgradGemm g[WG_MAX_GEMMS];
    int n_gemms;
    int splits;                   // S
    int rows_per_split;           // multiple of the stage rows
    int tangent;                  // ANR_MLP_FLAG_TANGENT: bias sums over the primal rows (row % 4 == 0) only
    const int32_t* count;         // device row count (a multiple of 64; NULL: all n rows): the slices are cut on the device
};

template <bool BF16> struct WgCfg;
template <> struct WgCfg<true> {
    using T = __bf16;
    static constexpr int SR = 32;                 // rows (points) per LDS stage
    static constexpr int KSTEP = 16;              // rows per MFMA
};
template <> struct WgCfg<false> {
    using T = float;
    static constexpr int SR = 16;
    static constexpr int KSTEP = 2;
};

// `block` = the workgroup's index among this GEMM shape's (gemm, slice) tasks; `lds` = the workgroup's dynamic LDS
template <bool BF16, int TM, int TN, int WM>
__device__ __forceinline__ void wgrad_body(const char* __restrict__ dact, const char* __restrict__ act,
                                           const char* __restrict__ enc, int64_t n, const WgradArgs& args,
                                           float* __restrict__ partial, int block, char* lds) {
    using C = WgCfg<BF16>;
    using T = typename C::T;
    constexpr int WN = WG_WAVES / WM;
    constexpr int M = TM * 32 * WM, N = TN * 32 * WN;
    constexpr int ESZ = sizeof(T);
    constexpr int SR = C::SR;
    // LDS image of a stage: the M / 32 blocks of the A side, then the N / 32 of the B side, each [SR rows][32 features]
    constexpr int BLK = SR * 32 * ESZ;                               // 2 KiB in either mode
    constexpr int RPP = 1024 / (32 * ESZ);                           // rows per 1-KiB DMA piece (16 / 8)
    constexpr int LPR = 64 / RPP;                                    // lanes per row of a piece (4 / 8)
    constexpr int STAGE_A = (M / 32) * BLK, STAGE_B = (N / 32) * BLK;
    constexpr int PIECES = (STAGE_A + STAGE_B) / 1024;
    constexpr int PPW = (PIECES + WG_WAVES - 1) / WG_WAVES;           // DMA pieces per wave and stage (padded: uniform waits)
    constexpr int STAGE = PPW * WG_WAVES * 1024;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gi = block / args.splits, sp = block % args.splits;
    const WgradGemm gm = args.g[gi];
    int64_t n_rows = n, per = args.rows_per_split;          // n stays the buffers' row count (the block stride)
    if (args.count) {
        const int64_t cnt = *args.count;
        n_rows = cnt < n ? cnt : n;
        per = ((n_rows + args.splits - 1) / args.splits + SR - 1) / SR * SR;
    }
    const int64_t r0 = (int64_t)sp * per;
    int64_t r1 = r0 + per;
    if (r1 > n_rows) r1 = n_rows;
    const int n_stages = r1 > r0 ? (int)((r1 - r0) / SR) : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // this lane's 16 bytes of a piece: row (inside the piece) and byte inside the block row
    // A piece = RPP rows x LPR piece arrays.  Lane l fetches array psub = l / RPP, row (l % RPP - ROT psub) mod RPP: RPP
    // consecutive lanes read RPP consecutive rows of ONE array — 256 (128) contiguous bytes, whole lines (with the lanes
    // interleaving the arrays, 64 bytes of four lines per 16 lanes, the kernel lost 17-26 %) — and its 16 bytes land at LDS
    // position l of the piece = psub RPP + (row + ROT psub) mod RPP: the rotation spreads the LPR arrays' copies of the same
    // rows over different bank windows for the transposing reads below.
    constexpr int ROT = BF16 ? 4 : 1;
    const int psub = lane / RPP, prow = ((lane % RPP) - ROT * psub) & (RPP - 1), pseg = psub * 16;

    auto issue = [&](int st) {                                  // slab of stage st -> ring buffer st % NBUF
        const int64_t row0 = r0 + (int64_t)st * SR;
        const unsigned buf = lds0 + (unsigned)(st % WG_NBUF) * STAGE;
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int piece = wave + p * WG_WAVES;               // wave-uniform
            const int blk = piece / (BLK / 1024), row = (piece % (BLK / 1024)) * RPP + prow;
            const char* src;
            // (a lane's 16 bytes = piece `psub` of the block's row: its own array [n][16 B] in the blocked buffers, mlp_core.h)
            if (piece < STAGE_A / 1024) {
                src = dact + act_piece_off(n, ESZ, gm.a_col / 32 + blk, psub) + (row0 + row) * 16;
            } else if (piece < PIECES) {
                const int bb = blk - M / 32;
                if (gm.b_src) src = enc + (row0 + row) * (64 * ESZ) + bb * (32 * ESZ) + pseg;        // enc[points][64], row-major
                else          src = act + act_piece_off(n, ESZ, gm.b_col / 32 + bb, psub) + (row0 + row) * 16;
            } else {
                src = dact + (row0 + row) * 16;                     // padding piece: lands behind the images, never read
            }
            dma16_nt(src, buf + piece * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int wm = wave % WM, wn = wave / WM;
    const int m_base = wm * TM * 32, n_base = wn * TN * 32;
    // bias gradient = column sums of the M side = (M side)^T . ones: one more MFMA per row tile against a constant
    // all-ones fragment, in the waves of the first column block (exact fp32 accumulation, no extra loads)
    const bool do_bias = gm.want_bias && wn == 0;
    f32x16 bacc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) bacc[a][e] = 0.0f;
    if (n_stages > 0) issue(0);
    if (n_stages > 1) issue(1);
    for (int st = 0; st < n_stages; ++st) {
        if (st + 1 < n_stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // stage st landed everywhere; stage st-1 is no longer read
        if (st + 2 < n_stages) issue(st + 2);
        const char* A = lds + (st % WG_NBUF) * STAGE;
        const char* B = A + STAGE_A;
        if constexpr (BF16) {
            const int grp = lane >> 4, s = lane & 15, h = lane >> 5;
            // lane's row / feature inside a [16 rows][32 features] fragment source block (one DMA piece): row frow (and frow + 8
            // for the second read), features fcol .. fcol + 3 = half (s & 1) of piece array fsub, at the piece's LDS position
            // fsub 16 + (row + 4 fsub) mod 16 (issue() above)
            const int frow = 4 * h + (s >> 2), fsub = 2 * (grp & 1) + ((s & 3) >> 1);
            const unsigned foff0 = (unsigned)((fsub * 16 + ((frow + 4 * fsub) & 15)) * 16 + 8 * (s & 1));
            const unsigned foff1 = (unsigned)((fsub * 16 + ((frow + 8 + 4 * fsub) & 15)) * 16 + 8 * (s & 1));
#pragma unroll
            for (int ks = 0; ks < SR / 16; ++ks) {
                bf16x8 fa[TM], fb[TN];
                uint2 alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)A +
                                        (m_base / 32 + a) * BLK + ks * 1024;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"
                                 : "=&v"(alo[a]), "=&v"(ahi[a]) : "v"(ad + foff0), "v"(ad + foff1) : "memory");
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)B +
                                        (n_base / 32 + b) * BLK + ks * 1024;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"
                                 : "=&v"(blo[b]), "=&v"(bhi[b]) : "v"(ad + foff0), "v"(ad + foff1) : "memory");
                }
                // one wait for the whole batch; the operands tie every fragment to it (the reads are opaque to hipcc)
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[a]), "+v"(ahi[a])::"memory");
                    fa[a] = __builtin_bit_cast(bf16x8, uint4{alo[a].x, alo[a].y, ahi[a].x, ahi[a].y});
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[b]), "+v"(bhi[b])::"memory");
                    fb[b] = __builtin_bit_cast(bf16x8, uint4{blo[b].x, blo[b].y, bhi[b].x, bhi[b].y});
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
                if (do_bias) {
                    // fragment element e <-> row 8 (e >> 2) + 4 h + (e & 3) of the K-step: primal rows are e & 3 == 0
                    bf16x8 ones;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)((args.tangent && (e & 3)) ? 0.0f : 1.0f);
#pragma unroll
                    for (int a = 0; a < TM; ++a) bacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], ones, bacc[a], 0, 0, 0);
                }
            }
        } else {
            const int i = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < SR / 2; ++ks) {
                float fa[TM], fb[TN];
                // row ks 2 + h of the stage = row r of DMA piece (ks 2 + h) / 8; feature i = element i % 4 of piece array
                // i / 4, at the piece's LDS position (i / 4) 8 + (r + i / 4) mod 8 (issue() above)
                const int r = (ks * 2 + h) & 7, pc = (ks * 2 + h) >> 3;
                const int foff = pc * 1024 + ((i >> 2) * 8 + ((r + (i >> 2)) & 7)) * 16 + (i & 3) * 4;
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[a] = *reinterpret_cast<const float*>(A + (m_base / 32 + a) * BLK + foff);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[b] = *reinterpret_cast<const float*>(B + (n_base / 32 + b) * BLK + foff);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        bacc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], (args.tangent && ((ks * 2 + h) & 3)) ? 0.0f : 1.0f, bacc[a], 0, 0, 0);
                }
            }
        }
    }
    // partial[gemm][split][M][N]; accumulator register e of lane: row (e&3) + 8 (e>>2) + 4 (lane>>5), column lane&31
    float* out = partial + gm.out_off + (int64_t)sp * (M * N + M);
    const int col = lane & 31, rofs = 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                out[(int64_t)(m_base + a * 32 + (e & 3) + 8 * (e >> 2) + rofs) * N + n_base + b * 32 + col] = acc[a][b][e];
    if (do_bias && col == 0) {                               // every column of bacc holds the same sums
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) out[M * N + m_base + a * 32 + (e & 3) + 8 * (e >> 2) + rofs] = bacc[a][e];
    }
}

template <bool BF16, int TM, int TN, int WM>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_kernel(const char* __restrict__ dact, const char* __restrict__ act,
                                                              const char* __restrict__ enc, int64_t n, WgradArgs args,
                                                              float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    wgrad_body<BF16, TM, TN, WM>(dact, act, enc, n, args, partial, (int)blockIdx.x, lds);
}

// ---- the two skinny heads as weighted column sums (their M side is g, 4 fp32 columns):
//   d sigma.weight[c] = sum_p g[p][3] h8[p][c]     d rgb.weight[j][c] = sum_p g[p][j] G[p][c]     biases: sum_p g[p][j]
// One workgroup per slice of rows.  A wave reads a whole row of h8 (256 columns, four per lane: one 8- / 16-byte load) or
// two rows of G (128 columns) per instruction, four rows in flight; its four waves interleave the rows and meet in LDS;
// slices are added by the reduction kernel.  (First version: one thread per output column with 2-byte loads, 72 us per
// call at any row count; this one moves the same bytes in a quarter of the load instructions.)
constexpr int CS_COLS = 256 + 3 * 128 + 4;                  // sigma.weight | rgb.weight | rgb.bias (3), sigma.bias

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    } else {
        const float4 f = *reinterpret_cast<const float4*>(p);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
}

// `block` of `n_blocks` slices; four wavefronts (threads 0..255) and 4 x CS_COLS floats of LDS
template <typename T>
__device__ __forceinline__ void heads_body(const T* __restrict__ act, const float* __restrict__ g4, int64_t n, int rows_per_slice,
                                           int sigma_only, int tangent, float* __restrict__ partial, const int32_t* __restrict__ count,
                                           int block, int n_blocks, float (*sh)[CS_COLS]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t n_rows = n;                                      // (n stays the buffer's row count: the block stride)
    if (count) {
        const int64_t cnt = *count;
        n_rows = cnt < n ? cnt : n;
        rows_per_slice = (int)((n_rows + n_blocks - 1) / n_blocks);
    }
    // four consecutive columns per lane: block (first column / 32 + lane / 8), features 4 (lane % 8) of row r
    // (blocked by 32-feature tile and by 16-byte piece inside it, mlp_core.h: rows of a piece array are EPP elements apart)
    constexpr int EPP = 16 / (int)sizeof(T);
    const T* h8 = reinterpret_cast<const T*>(reinterpret_cast<const char*>(act) + act_elem_off(n, sizeof(T), 1792 / 32 + (lane >> 3), 4 * (lane & 7), 0));
    const T* gh = reinterpret_cast<const T*>(reinterpret_cast<const char*>(act) + act_elem_off(n, sizeof(T), 2304 / 32 + ((lane & 31) >> 3), 4 * (lane & 7), 0));
    const int64_t r0 = (int64_t)block * rows_per_slice;
    int64_t r1 = r0 + rows_per_slice;
    if (r1 > n_rows) r1 = n_rows;
    constexpr int RIF = 4;
    // sigma.weight (+ all four bias sums, lane 0): wave w takes rows r0 + w, r0 + w + 4, ...
    float sw[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    {
        int64_t r = r0 + wave;
        for (; r + 4 * (RIF - 1) < r1; r += 4 * RIF) {
            float h[RIF][4];
            float4 g[RIF];
#pragma unroll
            for (int q = 0; q < RIF; ++q) {
                load4(h8 + (r + 4 * q) * EPP, h[q]);
                g[q] = reinterpret_cast<const float4*>(g4)[r + 4 * q];
            }
#pragma unroll
            for (int q = 0; q < RIF; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sw[i] += g[q].w * h[q][i];
                const float one = (tangent && ((r + 4 * q) & 3)) ? 0.0f : 1.0f;
                sb[0] += g[q].x * one; sb[1] += g[q].y * one; sb[2] += g[q].z * one; sb[3] += g[q].w * one;
            }
        }
        for (; r < r1; r += 4) {
            float h[4];
            load4(h8 + r * EPP, h);
            const float4 g = reinterpret_cast<const float4*>(g4)[r];
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] += g.w * h[i];
            const float one = (tangent && (r & 3)) ? 0.0f : 1.0f;
            sb[0] += g.x * one; sb[1] += g.y * one; sb[2] += g.z * one; sb[3] += g.w * one;
        }
    }
    // rgb.weight: 128 columns = 32 lanes x 4; the two half-waves take alternate rows
    float sr[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (!sigma_only) {
        int64_t r = r0 + 2 * wave + (lane >> 5);
        for (; r + 8 * (RIF - 1) < r1; r += 8 * RIF) {
            float h[RIF][4];
            float4 g[RIF];
#pragma unroll
            for (int q = 0; q < RIF; ++q) {
                load4(gh + (r + 8 * q) * EPP, h[q]);
                g[q] = reinterpret_cast<const float4*>(g4)[r + 8 * q];
            }
#pragma unroll
            for (int q = 0; q < RIF; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { sr[0][i] += g[q].x * h[q][i]; sr[1][i] += g[q].y * h[q][i]; sr[2][i] += g[q].z * h[q][i]; }
        }
        for (; r < r1; r += 8) {
            float h[4];
            load4(gh + r * EPP, h);
            const float4 g = reinterpret_cast<const float4*>(g4)[r];
#pragma unroll
            for (int i = 0; i < 4; ++i) { sr[0][i] += g.x * h[i]; sr[1][i] += g.y * h[i]; sr[2][i] += g.z * h[i]; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) sr[j][i] += __shfl_xor(sr[j][i], 32, 64);      // the two half-waves' rows
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sh[wave][4 * lane + i] = sw[i];
    if (lane < 32)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) sh[wave][256 + 128 * j + 4 * lane + i] = sr[j][i];
    if (lane == 0) { sh[wave][640] = sb[0]; sh[wave][641] = sb[1]; sh[wave][642] = sb[2]; sh[wave][643] = sb[3]; }
    __syncthreads();
    for (int c = threadIdx.x; c < CS_COLS; c += 256)
        partial[(int64_t)block * CS_COLS + c] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}

template <typename T>
__global__ __launch_bounds__(256) void heads_kernel(const T* __restrict__ act, const float* __restrict__ g4, int64_t n,
                                                    int rows_per_slice, int sigma_only, int tangent, float* __restrict__ partial,
                                                    const int32_t* __restrict__ count) {
    __shared__ float sh[4][CS_COLS];
    heads_body<T>(act, g4, n, rows_per_slice, sigma_only, tangent, partial, count, (int)blockIdx.x, (int)gridDim.x, sh);
}

// The three GEMM shapes of one weight-gradient call (256 x 256: trunk + xyz_encoding_final; 256 x 64: the encoding columns of
// layers 1 and 5; 128 x 256: dir_encoding) AND the skinny heads in ONE launch (round 6): a workgroup takes its shape from its
// index.  At the per-rank batch of the reference's 8-GPU run a call's five launches were 110 + 50 + 10 + 15 + 22 us of mostly
// latency, four calls per step queued on two streams — what the step ended on (profiles/r06/train_step_timeline_f2_first_box.txt);
// together the shapes' 150-200 workgroups fit the chip at once.  Same tasks, same partial sums, same reduction: same bits.
struct HeadsArgs { const float* g4; float* partial; int rows_per_slice, sigma_only, tangent, slices; };
template <bool BF16>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_shapes_kernel(const char* __restrict__ dact, const char* __restrict__ act,
                                                                     const char* __restrict__ enc, int64_t n, WgradArgs a0, WgradArgs a1,
                                                                     WgradArgs a2, float* __restrict__ partial, HeadsArgs hd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int n0 = a0.n_gemms * a0.splits, n1 = a1.n_gemms * a1.splits, n2 = a2.n_gemms * a2.splits;
    const int b = (int)blockIdx.x;
    if (b < n0) wgrad_body<BF16, 2, 4, 4>(dact, act, enc, n, a0, partial, b, lds);
    else if (b < n0 + n1) wgrad_body<BF16, 1, 2, 8>(dact, act, enc, n, a1, partial, b - n0, lds);
    else if (b < n0 + n1 + n2) wgrad_body<BF16, 2, 2, 2>(dact, act, enc, n, a2, partial, b - n0 - n1, lds);
    else {
        // the heads' slices: the stand-alone kernel's four wavefronts, sums and order (the other four leave)
        if (threadIdx.x >= 256) return;
        using T = typename WgCfg<BF16>::T;
        heads_body<T>(reinterpret_cast<const T*>(act), hd.g4, n, hd.rows_per_slice, hd.sigma_only, hd.tangent, hd.partial, a0.count,
                      b - n0 - n1 - n2, hd.slices, reinterpret_cast<float (*)[CS_COLS]>(lds));
    }
}

// ---- reduction over the slices + scatter into the 22 gradient tensors (flat, in the order of anr_mlp_wgrad_layout)
struct WgradSeg {
    int dst;          // first float of the segment in the flat gradient
    int rows, cols;   // segment shape in the destination (a column window of a [rows][dst_pitch] matrix)
    int dst_pitch;
    int src_off;      // float offset in the workspace
    int src_pitch;    // floats per source row
    int src_slice;    // floats between consecutive slices
    int slices;
};
struct WgradSegs { WgradSeg s[32]; int n; };

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradSegs segs, const float* __restrict__ ws, float* __restrict__ grads, int accumulate) {
    const WgradSeg sg = segs.s[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sg.rows * sg.cols) return;
    const int r = i / sg.cols, c = i % sg.cols;
    const float* src = ws + sg.src_off + (int64_t)r * sg.src_pitch + c;
    float s = 0.0f;                                          // fixed order; eight loads in flight
    int k = 0;
    for (; k + 8 <= sg.slices; k += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(k + q) * sg.src_slice];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; k < sg.slices; ++k) s += src[(int64_t)k * sg.src_slice];
    float* dst = grads + sg.dst + (int64_t)r * sg.dst_pitch + c;
    *dst = accumulate ? *dst + s : s;
}

}  // namespace anr

using namespace anr;

// flat gradient layout = the reference's parameter order used throughout (autograd.PARAM_KEYS):
// W1[256x63] b1 W2 b2 W3 b3 W4 b4 W5[256x319] b5 W6 b6 W7 b7 W8 b8 | sigma.w[256] sigma.b[1] | final.w final.b | dir.w[128x256] dir.b | rgb.w[3x128] rgb.b[3]
namespace {
struct Layout {
    int w[8], b[8], sw, sb, fw, fb, dw, db, rw, rb, total;
    Layout() {
        int o = 0;
        for (int l = 0; l < 8; ++l) {
            const int in = l == 0 ? 63 : l == 4 ? 319 : 256;
            w[l] = o; o += 256 * in;
            b[l] = o; o += 256;
        }
        sw = o; o += 256; sb = o; o += 1;
        fw = o; o += 65536; fb = o; o += 256;
        dw = o; o += 128 * 256; db = o; o += 128;
        rw = o; o += 384; rb = o; o += 3;
        total = o;
    }
};
const Layout& layout() { static Layout L; return L; }

int splits_for(int64_t n, int stage_rows, int gemms, int background) {
    // one workgroup per CU (the LDS ring admits one) over the GEMMs of one shape, at least 4 stages per task
    int64_t want = (256 + gemms - 1) / gemms;
    int64_t most = n / (4 * stage_rows);
    int64_t s = want < most ? want : most;
    // ANR_MLP_FLAG_BACKGROUND: half of them.  256 workgroups of 96 KB LDS and 110 MB of partial products per call leave the
    // launches this one runs next to nowhere to go: with half, the training step (its weight gradients on a stream of their
    // own behind the backward chain) goes from 3.80 to 3.67 ms at 16 frames and from 1.85 to 1.56 at 2 frames per rank; a
    // quarter: 3.91 / 1.58.  (A lone call on 131,072 rows is 22 % slower with half the slices: not the default.)
    if (background) s /= 2;
    return (int)(s < 1 ? 1 : s);
}
}  // namespace

extern "C" int64_t anr_mlp_wgrad_floats(void) { return layout().total; }
extern "C" int64_t anr_mlp_wgrad_sigma_floats(void) { return layout().fw; }

extern "C" int64_t anr_mlp_wgrad_ws_floats(int64_t n) {
    // upper bound for any n: three GEMM shapes at their maximum split counts + the column-sum slices
    (void)n;
    return (int64_t)8 * 37 * (65536 + 256) + (int64_t)2 * 64 * (256 * 64 + 256) + (int64_t)256 * (128 * 256 + 128) + (int64_t)1024 * CS_COLS + 1024;
}

template <bool BF16>
static int wgrad_launch(const void* act, const void* dact, const void* enc, const float* g4, int64_t n, int sigma_only,
                        int tangent, int accumulate, float* ws, float* grads, hipStream_t st, const int32_t* count, int background) {
    using C = WgCfg<BF16>;
    const Layout& L = layout();
    WgradSegs segs{};
    auto seg = [&](int dst, int rows, int cols, int dst_pitch, int src_off, int src_pitch, int src_slice, int slices) {
        segs.s[segs.n++] = WgradSeg{dst, rows, cols, dst_pitch, src_off, src_pitch, src_slice, slices};
    };
    int ws_off = 0;
    const char* A = reinterpret_cast<const char*>(act);
    const char* D = reinterpret_cast<const char*>(dact);
    const char* E = reinterpret_cast<const char*>(enc);
    auto lds_bytes = [](int M, int N) {
        const int stage = C::SR * (M + N) * (int)sizeof(typename C::T);
        const int pieces = stage / 1024, ppw = (pieces + WG_WAVES - 1) / WG_WAVES;
        return WG_NBUF * ppw * WG_WAVES * 1024;
    };
    WgradArgs a0{}, a1{}, a2{};
    // ---- 256 x 256: trunk layers 2..8 (hidden part of layer 5) and xyz_encoding_final
    {
        WgradArgs& a = a0;
        a.tangent = tangent;
        a.count = count;
        const int n_g = sigma_only ? 7 : 8;
        a.splits = splits_for(n, C::SR, n_g, background);
        a.rows_per_split = (int)(((n + a.splits - 1) / a.splits + C::SR - 1) / C::SR * C::SR);
        constexpr int BLK = 65536 + 256;                       // [M][N] + the M column sums
        for (int l = 2; l <= 8; ++l) {
            a.g[a.n_gemms] = WgradGemm{256 * (l - 1), 0, 256 * (l - 2), ws_off, 1};
            seg(L.w[l - 1] + (l == 5 ? 63 : 0), 256, 256, l == 5 ? 319 : 256, ws_off, 256, BLK, a.splits);
            seg(L.b[l - 1], 1, 256, 256, ws_off + 65536, BLK, BLK, a.splits);
            ws_off += a.splits * BLK;
            ++a.n_gemms;
        }
        if (!sigma_only) {
            a.g[a.n_gemms] = WgradGemm{2048, 0, 1792, ws_off, 1};
            seg(L.fw, 256, 256, 256, ws_off, 256, BLK, a.splits);
            seg(L.fb, 1, 256, 256, ws_off + 65536, BLK, BLK, a.splits);
            ws_off += a.splits * BLK;
            ++a.n_gemms;
        }
    }
    // ---- 256 x 64: the encoding columns of layers 1 and 5
    {
        WgradArgs& a = a1;
        a.tangent = tangent;
        a.count = count;
        a.splits = splits_for(n, C::SR, 2, background);
        if (a.splits > (background ? 32 : 64)) a.splits = background ? 32 : 64;
        a.rows_per_split = (int)(((n + a.splits - 1) / a.splits + C::SR - 1) / C::SR * C::SR);
        constexpr int BLK = 256 * 64 + 256;
        for (int l : {1, 5}) {
            a.g[a.n_gemms] = WgradGemm{256 * (l - 1), 1, 0, ws_off, l == 1};
            seg(L.w[l - 1], 256, 63, l == 5 ? 319 : 63, ws_off, 64, BLK, a.splits);
            if (l == 1) seg(L.b[0], 1, 256, 256, ws_off + 256 * 64, BLK, BLK, a.splits);
            ws_off += a.splits * BLK;
            ++a.n_gemms;
        }
    }
    // ---- 128 x 256: dir_encoding
    a2.splits = 1;
    if (!sigma_only) {
        WgradArgs& a = a2;
        a.tangent = tangent;
        a.count = count;
        a.splits = splits_for(n, C::SR, 1, background);
        a.rows_per_split = (int)(((n + a.splits - 1) / a.splits + C::SR - 1) / C::SR * C::SR);
        constexpr int BLK = 128 * 256 + 128;
        a.g[0] = WgradGemm{2304, 0, 2048, ws_off, 1};
        a.n_gemms = 1;
        seg(L.dw, 128, 256, 256, ws_off, 256, BLK, a.splits);
        seg(L.db, 1, 128, 128, ws_off + 128 * 256, BLK, BLK, a.splits);
        ws_off += a.splits * BLK;
    }
    // ---- the skinny heads (sigma, rgb) and their biases: slices of rows, summed by the reduction like the GEMMs' partials
    // (256 slices at most: the reduction below adds a segment's slices one after the other per element, and with 1,024
    // slices of the heads' column sums it was that chain — not the 32 slices of the big GEMMs — that set its 50-60 us)
    int slices = (int)((n + 127) / 128);
    if (slices > 256) slices = 256;
    const int rps = (int)((n + slices - 1) / slices);
    slices = (int)((n + rps - 1) / rps);
    float* heads_ws = ws + ws_off;
    seg(L.sw, 1, 256, 256, ws_off, CS_COLS, CS_COLS, slices);
    seg(L.sb, 1, 1, 1, ws_off + 643, CS_COLS, CS_COLS, slices);
    if (!sigma_only) {
        seg(L.rw, 1, 384, 384, ws_off + 256, CS_COLS, CS_COLS, slices);
        seg(L.rb, 1, 3, 3, ws_off + 640, CS_COLS, CS_COLS, slices);
    }
    ws_off += slices * CS_COLS;
    using T = typename C::T;
    // (ANR_WGRAD_SEPARATE=1: a launch per shape and one for the heads, as before round 6 — the A/B switch)
    static const bool separate = getenv("ANR_WGRAD_SEPARATE") != nullptr;
    if (!separate) {
        auto k = wgrad_shapes_kernel<BF16>;
        const int lds = lds_bytes(256, 256);                    // the largest of the three images (the heads' 10 KB fit it)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail((int)e, "anr_mlp_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        const HeadsArgs hd{g4, heads_ws, rps, sigma_only, tangent, slices};
        hipLaunchKernelGGL(k, dim3(a0.n_gemms * a0.splits + a1.n_gemms * a1.splits + a2.n_gemms * a2.splits + slices), dim3(WG_THREADS), lds,
                           st, D, A, E, n, a0, a1, a2, ws, hd);
    } else {
        {
            auto k = wgrad_kernel<BF16, 2, 4, 4>;
            const int lds = lds_bytes(256, 256);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return fail((int)e, "anr_mlp_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(k, dim3(a0.n_gemms * a0.splits), dim3(WG_THREADS), lds, st, D, A, E, n, a0, ws);
        }
        {
            auto k = wgrad_kernel<BF16, 1, 2, 8>;
            const int lds = lds_bytes(256, 64);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return fail((int)e, "anr_mlp_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(k, dim3(a1.n_gemms * a1.splits), dim3(WG_THREADS), lds, st, D, A, E, n, a1, ws);
        }
        if (!sigma_only) {
            auto k = wgrad_kernel<BF16, 2, 2, 2>;
            const int lds = lds_bytes(128, 256);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return fail((int)e, "anr_mlp_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(k, dim3(a2.n_gemms * a2.splits), dim3(WG_THREADS), lds, st, D, A, E, n, a2, ws);
        }
        hipLaunchKernelGGL(heads_kernel<T>, dim3(slices), dim3(256), 0, st, reinterpret_cast<const T*>(act), g4, n, rps, sigma_only, tangent,
                           heads_ws, count);
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(256, segs.n), dim3(256), 0, st, segs, ws, grads, accumulate);
    return check_launch("anr_mlp_wgrad");
}

extern "C" int anr_mlp_wgrad(int mode, const void* act, const void* dact, const void* enc, const float* g4, int64_t n,
                             float* workspace, float* grads_out, void* stream) {
    return anr_mlp_wgrad_counted(mode, act, dact, enc, g4, n, nullptr, workspace, grads_out, stream);
}

extern "C" int anr_mlp_wgrad_counted(int mode, const void* act, const void* dact, const void* enc, const float* g4, int64_t n,
                                     const int32_t* count, float* workspace, float* grads_out, void* stream) {
    ANR_REQUIRE(act && dact && enc && g4 && workspace && grads_out, ANR_E_BADARG, "anr_mlp_wgrad: null pointer");
    ANR_REQUIRE(n > 0 && n % 64 == 0, ANR_E_BADARG, "anr_mlp_wgrad: n=%lld must be a positive multiple of 64", (long long)n);
    ANR_REQUIRE((((uintptr_t)act | (uintptr_t)dact | (uintptr_t)enc | (uintptr_t)workspace) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_wgrad: act/dact/enc/workspace must be 16-B aligned");
    const int so = (mode & ANR_MLP_FLAG_SIGMA_ONLY) ? 1 : 0;
    const int tan = (mode & ANR_MLP_FLAG_TANGENT) ? 1 : 0;
    ANR_REQUIRE(!tan || so, ANR_E_BADARG, "anr_mlp_wgrad: tangent mode = sigma only");
    hipStream_t st = (hipStream_t)stream;
    const int accumulate = (mode & ANR_MLP_FLAG_ACCUMULATE) ? 1 : 0;
    const int bg = (mode & ANR_MLP_FLAG_BACKGROUND) ? 1 : 0;
    if (so && !accumulate && !(mode & ANR_MLP_FLAG_NO_FILL)) {       // tensors this call does not produce: zeros
        const Layout& L = layout();
        if (int rc = zero_fill(grads_out + L.fw, sizeof(float) * (L.total - L.fw), st, "anr_mlp_wgrad (zero)")) return rc;
    }
    switch (mode & 0xff) {
        case ANR_MLP_BF16: return wgrad_launch<true>(act, dact, enc, g4, n, so, tan, accumulate, workspace, grads_out, st, count, bg);
        case ANR_MLP_F32:  return wgrad_launch<false>(act, dact, enc, g4, n, so, tan, accumulate, workspace, grads_out, st, count, bg);
        default: return fail(ANR_E_BADARG, "anr_mlp_wgrad: unknown mode %d", mode);
    }
}

// =====================================================================================================================
// dL/d enc: the gradient that leaves the MLP through its two encoding inputs (layers 1 and 5) on its way to the sample
// positions — pose refinement only (train.py:141-144):  d_enc[p][c] = dact_1[p] . W1[:, c] + dact_5[p] . W5[:, c],  c < 63.
// Points are the MFMA row dimension here, so the A fragments (2 x 4 consecutive out-features of one point: halves of two
// 16-byte pieces of its row) come straight from global memory; the two 256 x 64 weight panels are converted to B-fragment
// order in LDS once per workgroup.  One wave per 32 points, persistent workgroups.
namespace anr {

template <bool BF16>
__global__ __launch_bounds__(256) void denc_kernel(const char* __restrict__ dact, const float* __restrict__ W1,
                                                   const float* __restrict__ W5, int64_t n, float* __restrict__ d_enc,
                                                   const int32_t* __restrict__ count) {
    using T = typename WgCfg<BF16>::T;
    constexpr int KF = BF16 ? 16 : 128;                      // K fragments per layer (16 / 2 out-features each)
    constexpr int EPL = BF16 ? 8 : 1;                        // elements per lane and fragment
    extern __shared__ __attribute__((aligned(16))) char lds[];
    T* panel = reinterpret_cast<T*>(lds);                    // [layer 2][kf][ntile 2][lane 64][EPL]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * KF * 2 * 64; i += 256) {
        const int l = i & 63, nt = (i >> 6) & 1, kf = (i >> 7) % KF, layer = (i >> 7) / KF;
        const int j = (l & 31) + 32 * nt, h = l >> 5;
        const float* W = layer ? W5 : W1;
        const int ld = layer ? 319 : 63;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int k = BF16 ? 16 * kf + 8 * (e >> 2) + 4 * h + (e & 3) : 2 * kf + h;
            panel[(int64_t)i * EPL + e] = (T)(j < 63 ? W[(int64_t)k * ld + j] : 0.0f);
        }
    }
    __syncthreads();
    int64_t n_rows = n;                                      // (n stays the buffer's row count: the block stride)
    if (count) { const int64_t cnt = *count; n_rows = cnt < n ? cnt : n; }
    const int64_t n_tiles = (n_rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p = tile * 32 + (lane & 31);
        const int64_t row = p < n_rows ? p : n_rows - 1;
        const char* arow = dact + row * 16;                              // this point's row inside a piece array of dact
        const int h = lane >> 5;
        f32x16 acc[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const char* a0 = arow + act_block_off(n, sizeof(T), layer ? 32 : 0);    // dact_5 (columns 1024..) / dact_1
#pragma unroll 4
            for (int kf = 0; kf < KF; ++kf) {
                const T* b0 = panel + ((int64_t)((layer * KF + kf) * 2 + 0) * 64 + lane) * EPL;
                const T* b1 = panel + ((int64_t)((layer * KF + kf) * 2 + 1) * 64 + lane) * EPL;
                if constexpr (BF16) {
                    // features 16 (kf & 1) + 4 h .. + 3 and the same + 8: half h of pieces 2 (kf & 1) and 2 (kf & 1) + 1
                    const char* ab = a0 + act_piece_off(n, 2, kf >> 1, 2 * (kf & 1)) + 8 * h;
                    const uint2 lo = *reinterpret_cast<const uint2*>(ab);
                    const uint2 hi = *reinterpret_cast<const uint2*>(ab + n * 16);
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(b0), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(b1), acc[1], 0, 0, 0);
                } else {
                    const float fa = *reinterpret_cast<const float*>(a0 + act_elem_off(n, 4, kf >> 4, 2 * (kf & 15) + h, 0));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, *b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, *b1, acc[1], 0, 0, 0);
                }
            }
        }
        // D[row = point (e&3) + 8 (e>>2) + 4 h][col = channel lane&31 (+32)]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t pr = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int ch = (lane & 31) + 32 * nt;
                if (pr < n_rows && ch < 63) d_enc[pr * 63 + ch] = acc[nt][e];
            }
    }
}

}  // namespace anr

// =====================================================================================================================
// dL/d x of a compacted pass in ONE launch (round 5): anr_mlp_denc + anr_encode_backward — d_enc never travels through HBM
// (252 B written + read per row), the encoding's derivative runs on whole wavefronts, and the weight panels come pre-packed
// in B-fragment order from the backward weight pack (mlp_core.h: denc_panel_off) instead of being converted from fp32 by every
// workgroup.  A wavefront takes TP points (64 bf16 / 32 fp32): MFMA tiles as in denc_kernel, accumulators -> this wave's LDS
// patch [point][65 floats] (pitch 65: the column walk below is conflict-free), then lane p owns point p: its 63 channel
// gradients meet the derivative of the Fourier features,
//   d x_d = g[d] + sum_k 2^k (cos(2^k x_d) g[3 + 6k + d] - sin(2^k x_d) g[6 + 6k + d])        (models/embedding.py:22-39)
// 86 + 59 us -> one launch at the per-rank batch of the reference's 8-GPU run (~70 k rows).
namespace anr {

// NTILE = 32-point tiles per wavefront and trip: 2 for bf16 where the rows are many; 1 when the buffer is small (a 2-frame
// training batch: ~20 k listed rows are 300 64-point tiles for 1,024 wavefronts — half the chain per wavefront with 32)
template <bool BF16, int NTILE>
__global__ __launch_bounds__(BF16 ? 256 : 128) void dpoints_kernel(const char* __restrict__ dact, const char* __restrict__ panel_g,
                                                                   const float4* __restrict__ pts, int64_t n,
                                                                   float4* __restrict__ d_pts, const int32_t* __restrict__ count) {
    using T = typename WgCfg<BF16>::T;
    constexpr int KF = BF16 ? 16 : 128, EPL = BF16 ? 8 : 1;
    constexpr int WAVES = BF16 ? 4 : 2, TP = 32 * NTILE;
    constexpr int PANEL_BYTES = DENC_PANEL_ELEMS * (int)sizeof(T);
    constexpr int PITCH = 65;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const T* panel = reinterpret_cast<const T*>(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    float* patch = reinterpret_cast<float*>(lds + PANEL_BYTES) + wave * (TP * PITCH);
    {   // the panel: every 16-byte piece of a thread in flight before the first LDS store (a copy loop of load, store is one
        // trip to L2 per iteration: 16 trips, ~30 us of a 40 us launch at a 2-frame batch)
        constexpr int PIECES = PANEL_BYTES / 16 / (WAVES * 64);
        static_assert(PANEL_BYTES % (16 * WAVES * 64) == 0, "whole pieces per thread");
        uint4 tmp[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) tmp[i] = reinterpret_cast<const uint4*>(panel_g)[i * (WAVES * 64) + threadIdx.x];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) reinterpret_cast<uint4*>(lds)[i * (WAVES * 64) + threadIdx.x] = tmp[i];
    }
    __syncthreads();
    int64_t n_rows = n;                                      // (n stays the buffer's row count: the piece arrays' stride)
    if (count) { const int64_t cnt = *count; n_rows = cnt < n ? cnt : n; }
    const int64_t n_tiles = (n_rows + TP - 1) / TP;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        f32x16 acc[NTILE][2];
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[t][0][e] = 0.f; acc[t][1][e] = 0.f; }
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const int64_t p = tile * TP + t * 32 + (lane & 31);
            const int64_t row = p < n_rows ? p : n_rows - 1;
            const char* arow = dact + row * 16;
#pragma unroll
            for (int layer = 0; layer < 2; ++layer) {
                const char* a0 = arow + act_block_off(n, sizeof(T), layer ? 32 : 0);    // dact_5 (columns 1024..) / dact_1
                // (bf16: the 32 eight-byte loads of a layer in flight at once — the pass is a chain of trips to L2)
#pragma unroll 16
                for (int kf = 0; kf < KF; ++kf) {
                    const T* b0 = panel + ((int64_t)((layer * KF + kf) * 2 + 0) * 64 + lane) * EPL;
                    const T* b1 = panel + ((int64_t)((layer * KF + kf) * 2 + 1) * 64 + lane) * EPL;
                    if constexpr (BF16) {
                        const char* ab = a0 + act_piece_off(n, 2, kf >> 1, 2 * (kf & 1)) + 8 * h;
                        const uint2 lo = *reinterpret_cast<const uint2*>(ab);
                        const uint2 hi = *reinterpret_cast<const uint2*>(ab + n * 16);
                        const bf16x8 fa = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
                        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(b0), acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, *reinterpret_cast<const bf16x8*>(b1), acc[t][1], 0, 0, 0);
                    } else {
                        const float fa = *reinterpret_cast<const float*>(a0 + act_elem_off(n, 4, kf >> 4, 2 * (kf & 15) + h, 0));
                        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, *b0, acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, *b1, acc[t][1], 0, 0, 0);
                    }
                }
            }
            // D[row = point (e&3) + 8 (e>>2) + 4 h][col = channel lane&31 (+32)] -> patch[point][channel]
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    patch[(t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * PITCH + (lane & 31) + 32 * nt] = acc[t][nt][e];
        }
        // (the patch is this wavefront's own: the LDS unit executes a wave's instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int64_t p = tile * TP + lane;
        if (lane < TP && p < n_rows) {
            const float4 x4 = pts[p];
            const float x[3] = {x4.x, x4.y, x4.z};
            const float* g = patch + lane * PITCH;
            float dx[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float a = g[d];
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    const float f = (float)(1 << k), arg = f * x[d];
                    a += f * (sin_or_cos(arg, 1) * g[3 + 6 * k + d] - sin_or_cos(arg, 0) * g[6 + 6 * k + d]);
                }
                dx[d] = a;
            }
            d_pts[p] = make_float4(dx[0], dx[1], dx[2], 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

}  // namespace anr

extern "C" int anr_mlp_dpoints(const void* bwd_pack, int mode, const void* dact, const float* pts, int64_t n, const int32_t* count,
                               float* d_pts_out, void* stream) {
    ANR_REQUIRE(bwd_pack && dact && pts && d_pts_out, ANR_E_BADARG, "anr_mlp_dpoints: null pointer");
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_dpoints: n=%lld", (long long)n);
    ANR_REQUIRE((((uintptr_t)bwd_pack | (uintptr_t)dact | (uintptr_t)pts | (uintptr_t)d_pts_out) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_dpoints: bwd_pack/dact/pts/d_pts_out must be 16-B aligned");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipStream_t st = (hipStream_t)stream;
    const char* pack = reinterpret_cast<const char*>(bwd_pack);
    const float4* p4 = reinterpret_cast<const float4*>(pts);
    float4* o4 = reinterpret_cast<float4*>(d_pts_out);
    if ((mode & 0xff) == ANR_MLP_BF16) {
        const int lds = DENC_PANEL_ELEMS * 2 + 4 * 64 * 65 * 4;
        const bool small = n <= (int64_t)1 << 18;
        const int64_t wgs = (n + (small ? 127 : 255)) / (small ? 128 : 256);
        auto k = small ? dpoints_kernel<true, 1> : dpoints_kernel<true, 2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail((int)e, "anr_mlp_dpoints: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3((unsigned)(wgs < cus ? wgs : cus)), dim3(256), lds, st, reinterpret_cast<const char*>(dact),
                           pack + denc_panel_off<Cfg<ANR_MLP_BF16>>(), p4, n, o4, count);
    } else if ((mode & 0xff) == ANR_MLP_F32) {
        const int lds = DENC_PANEL_ELEMS * 4 + 2 * 32 * 65 * 4;
        const int64_t wgs = (n + 63) / 64;
        auto k = dpoints_kernel<false, 1>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail((int)e, "anr_mlp_dpoints: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3((unsigned)(wgs < cus ? wgs : cus)), dim3(128), lds, st, reinterpret_cast<const char*>(dact),
                           pack + denc_panel_off<Cfg<ANR_MLP_F32>>(), p4, n, o4, count);
    } else {
        return fail(ANR_E_BADARG, "anr_mlp_dpoints: unknown mode %d", mode);
    }
    return check_launch("anr_mlp_dpoints");
}

extern "C" int anr_mlp_denc(int mode, const void* dact, const float* w1, const float* w5, int64_t n, float* d_enc_out,
                            void* stream) {
    return anr_mlp_denc_counted(mode, dact, w1, w5, n, nullptr, d_enc_out, stream);
}

extern "C" int anr_mlp_denc_counted(int mode, const void* dact, const float* w1, const float* w5, int64_t n, const int32_t* count,
                                    float* d_enc_out, void* stream) {
    ANR_REQUIRE(dact && w1 && w5 && d_enc_out, ANR_E_BADARG, "anr_mlp_denc: null pointer");
    ANR_REQUIRE(n > 0 && ((uintptr_t)dact & 15) == 0, ANR_E_BADARG, "anr_mlp_denc: n=%lld / alignment", (long long)n);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t tiles = (n + 127) / 128;
    dim3 grid((unsigned)(tiles < 2 * cus ? tiles : 2 * cus));
    hipStream_t st = (hipStream_t)stream;
    if ((mode & 0xff) == ANR_MLP_BF16) {
        const int lds = 2 * 16 * 2 * 64 * 8 * 2;
        auto k = denc_kernel<true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail((int)e, "anr_mlp_denc: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, reinterpret_cast<const char*>(dact), w1, w5, n, d_enc_out, count);
    } else if ((mode & 0xff) == ANR_MLP_F32) {
        const int lds = 2 * 128 * 2 * 64 * 4;
        auto k = denc_kernel<false>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail((int)e, "anr_mlp_denc: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, reinterpret_cast<const char*>(dact), w1, w5, n, d_enc_out, count);
    } else {
        return fail(ANR_E_BADARG, "anr_mlp_denc: unknown mode %d", mode);
    }
    return check_launch("anr_mlp_denc");
}
