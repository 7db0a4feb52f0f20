// (header) Fused Fourier encoding + 8x256 NeRF MLP (a11 + a12) on CDNA4 matrix cores.
//
// Reference: models/embedding.py:22-39, models/nerf.py:129-175 (use_view = False).
//
// Shape of the computation.  The GEMMs are evaluated TRANSPOSED: out-features are the MFMA row
// dimension (A operand = a 32-row weight tile), sample points are the column dimension
// (B operand = activations), so D^T[feature][point] lands with each lane holding 16 features of
// ONE point.  With the K index of the next layer's weights permuted to match (done once by
// anr_mlp_pack), those 16 accumulators ARE the next layer's B fragments: activations never
// leave registers for the whole 11-GEMM chain, there is no LDS/HBM round trip and no cross-lane
// shuffle between layers.  Weights stream L2 -> LDS (chunks of 32-row tiles through a 3-slot ring, LDS-DMA) and are
// shared by the workgroup's wavefronts; each wavefront owns NT x 32 points; workgroups are persistent (one per CU).
//
//   mode BF16 (default shape, Cfg<ANR_MLP_BF16_W8>): v_mfma_f32_32x32x16_bf16, 8 waves x 32 points = 256 / workgroup
//   mode BF16 (flag 0x200, Cfg<ANR_MLP_BF16>):        4 waves x 64 points (NT = 2), half the LDS reads per flop
//   mode F32 : v_mfma_f32_32x32x2_f32, 4 waves x 32 points = 128 / workgroup — exact fp32 fmaf chains, the parity mode.
//
// Slot algebra (h = lane>>5, i = lane&31; "frag" = 16 bytes per lane = 1 KiB per wave):
//   accumulator reg (g = reg>>2, r = reg&3) of out-tile t  <->  out feature 32t + 8g + 4h + r
//   BF16 frag b, elem e  <->  hidden feature 16b + 8(e>>2) + 4h + (e&3)   => frags 2t, 2t+1 = regs 0-7, 8-15
//   F32  frag s, elem e  <->  hidden feature  8s + 4h + e                  => frags 4t..4t+3 = regs 4f..4f+3
//   encoding panel: slot j in [0,32) (BF16: j = 8b+e, F32: j = 4s+e), half h:
//       j < 30 : k = j/3, d = j%3 -> h ? cos(2^k x_d) : sin(2^k x_d)   (reference channel 3 + 6k + 3h + d)
//       j = 30 : h ? x_2 : x_0 ;  j = 31 : h ? <pad, weight 0> : x_1
//   so the two half-waves run one instruction stream (sin vs cos is a quadrant offset).
#pragma once
#include "anr_common.h"
#include <type_traits>
#include <utility>

namespace anr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef ANR_HALF_TILES
#define ANR_HALF_TILES 1            // 0: an A/B build without the small-batch half tiles (tools/bench_mlp_small.py)
#endif
constexpr int N_TILES_TOTAL = 78;              // 8*8 trunk + 1 (sigma) + 8 (final) + 4 (dir) + 1 (rgb)
constexpr int BIAS_BYTES = 10240;              // 78 tiles x 2 halves x 16 floats, padded
constexpr int FRAG_BYTES = 1024;

// MODE = arithmetic (ANR_MLP_F32 / ANR_MLP_BF16); VAR = workgroup shape variant
constexpr int ANR_MLP_BF16_W8 = 2;   // internal: bf16 with 8 waves x 32 points (2 waves per SIMD)
template <int MODE> struct Cfg;
template <> struct Cfg<ANR_MLP_BF16> {
    using Frag = bf16x8;
    static constexpr bool IS_BF16 = true;
    static constexpr int EPF = 8;    // elements per frag per lane
    static constexpr int NT = 2;     // 32-point column tiles per wave
    static constexpr int WAVES = 4;  // wavefronts per workgroup
    static constexpr int TPC = 2;    // out-tiles per staged weight chunk (one workgroup barrier per chunk)
    static constexpr int HF = 16;    // frags per 256 hidden features
    static constexpr int EF = 4;     // frags of the 64-slot encoding panel
    static constexpr int DF = 8;     // frags per 128 features (rgb head input)
};
template <> struct Cfg<ANR_MLP_BF16_W8> : Cfg<ANR_MLP_BF16> {
    static constexpr int NT = 1;
    static constexpr int WAVES = 8;
};
template <> struct Cfg<ANR_MLP_F32> {
    using Frag = f32x4;
    static constexpr bool IS_BF16 = false;
    static constexpr int EPF = 4;
    static constexpr int NT = 1;
    static constexpr int WAVES = 4;
    static constexpr int TPC = 1;
    static constexpr int HF = 32;
    static constexpr int EF = 8;
    static constexpr int DF = 16;
};

// frags of out-tile t of the flat 78-tile schedule, and of staged chunk c (= TPC consecutive tiles)
// VIEW (use_view=True, models/nerf.py:141-153): the four dir_encoding tiles (73..76) take the Fourier panel of the view
// direction in front of the 256-wide feature, as the skip layer takes the position's
template <class C, bool VIEW = false> __host__ __device__ constexpr int tile_frags(int t) {
    return t < 8 ? C::EF : t < 32 ? C::HF : t < 40 ? C::EF + C::HF : t < 73 ? C::HF : t < 77 ? (VIEW ? C::EF + C::HF : C::HF) : t < 78 ? C::DF : 0;
}
template <class C, bool VIEW = false> __host__ __device__ constexpr int chunk_frags(int c) {
    int n = 0;
    for (int i = 0; i < C::TPC; ++i) n += tile_frags<C, VIEW>(c * C::TPC + i);
    return n;
}
template <class C, bool VIEW = false> constexpr int total_frags() {
    return 8 * C::EF + 24 * C::HF + 8 * (C::EF + C::HF) + 24 * C::HF + 9 * C::HF + 4 * (C::HF + (VIEW ? C::EF : 0)) + C::DF;
}
template <class C> constexpr int slot_bytes() { return C::TPC * (C::EF + C::HF) * FRAG_BYTES; }

__device__ __forceinline__ void mma(const bf16x8& w, const bf16x8& x, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(const f32x4& w, const f32x4& x, f32x16& acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], x[e], acc, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mma_c(const bf16x8& w, const bf16x8& x, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma_c(const f32x4& w, const f32x4& x, const f32x16& c) {
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[0], c, 0, 0, 0);
#pragma unroll
    for (int e = 1; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], x[e], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ void put(bf16x8& f, int e, float v) { f[e] = (__bf16)v; }
__device__ __forceinline__ void put(f32x4& f, int e, float v) { f[e] = v; }

// Materialise a fragment in VGPRs HERE: keeps hipcc from carrying a whole layer of un-converted fp32
// accumulators (8 tiles x 32 registers) and doing the activation/convert pass at the end of the layer.
__device__ __forceinline__ void pin(bf16x8& f) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t = __builtin_bit_cast(u32x4, f);
    asm volatile("" : "+v"(t));
    f = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void pin(f32x4& f) { asm volatile("" : "+v"(f)); }

// sin(a) for q_off = 0, cos(a) for q_off = 1; Cody-Waite reduction by pi/2 + Cephes minimax polynomials (~1 ulp).
__device__ __forceinline__ float sin_or_cos(float a, int q_off) {
    const float n = rintf(a * 0.6366197466850281f);
    float r = fmaf(-n, 1.5707963705062866f, a);
    r = fmaf(-n, -4.371138828673793e-08f, r);
    r = fmaf(-n, -1.7151245100058819e-15f, r);
    const int q = (int)n + q_off;
    const float z = r * r;
    const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                         fmaf(-0.5f, z, 1.0f));
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

// ---------------------------------------------------------------------------------------------
// weight staging: chunk = nf frags of 1 KiB; wave w moves pieces w, w+WAVES, ...
//
// The LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B straight from L2 into LDS at M0 + lane*16) is issued
// from inline asm on purpose: when hipcc sees the builtin it can no longer prove which ds_reads the DMA may
// alias and degrades EVERY LDS wait in the kernel to `s_waitcnt lgkmcnt(0)`, which exposes the full LDS latency
// in front of each group of MFMAs.  Hidden in asm, the fragment reads keep their counted lgkmcnt(N) waits; the
// DMA's own completion is waited for by hand (`vmcnt(0)` in front of the workgroup barrier, dma_wait()).
__device__ __forceinline__ void dma16(const char* gsrc_lane, unsigned dst /* LDS byte address, wave-uniform */) {
    unsigned keep;
#define ANR_DMA_MOD ""
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ANR_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(__builtin_amdgcn_readfirstlane(dst))
                 : "memory");
}
// ... nontemporal: for a stream every byte of which is read ONCE by ONE workgroup (the weight-gradient kernel's slabs of saved
// activations: tools/exp/exp_dma_patterns.hip: 6.06 -> 6.86 TB/s on the bare stream); NOT for the weight chunks above, which
// every workgroup re-reads from L2
__device__ __forceinline__ void dma16_nt(const char* gsrc_lane, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(__builtin_amdgcn_readfirstlane(dst))
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// `slot` = byte offset of the ring slot inside the workgroup's dynamic LDS array `lds`
template <bool DMA, int WAVES>
__device__ __forceinline__ void stage_chunk(const char* __restrict__ g, char* lds, unsigned slot, int nf, int wave, int lane) {
    if constexpr (DMA) {
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + slot;
        for (int p = wave; p < nf; p += WAVES) dma16(g + p * FRAG_BYTES + lane * 16, base + p * FRAG_BYTES);
    } else {
        for (int p = wave; p < nf; p += WAVES) {
            uint4 v = *reinterpret_cast<const uint4*>(g + p * FRAG_BYTES + lane * 16);
            *reinterpret_cast<uint4*>(lds + slot + p * FRAG_BYTES + lane * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int V> struct IC { static constexpr int value = V; constexpr operator int() const { return V; } };
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- saved activations / activation gradients (training): a buffer of R rows (points), BLOCKED by 32-feature tile and,
// inside a block, by 16-BYTE PIECE of a row (8 bf16 / 4 fp32 features):
//   column c of row r  ->  byte ((c / 32 * PPB + (c % 32) / EPP) * R + r) * 16 + (c % EPP) * esz,   EPP = 16 / esz, PPB = 32 / EPP
//   76 blocks: h1..h8 (8 x 8) | final (8) | dir hidden (4), each PPB piece arrays [R][16 bytes]
// What a lane holds when it stores is one 16-byte piece of its row (mlp_core.h FragEpi / mlp_bwd.hip MaskEpi), and the 32
// lanes of a half-wave hold the SAME piece of 32 consecutive rows: with this layout a store instruction writes two spans of
// 512 contiguous bytes — eight full 128-byte lines, nothing partial — and may therefore be issued nontemporal.  Round 3's
// layout ([R][32 features] per block: the two 16-byte halves a store wrote of each 64-byte row came from lanes i and i + 32)
// made every instruction touch 16 lines with 32-byte partial writes: 1.31 ms per 2^20 rows next to the MFMA chain in
// tools/exp/exp_store_mfma2.hip where this one takes 1.16, and 1.09 nontemporal (round 2's row-major rows of 5 KB: 2.0).
// The consumers gather: the weight-gradient kernel's LDS-DMA reads 16 rows x 16 B from each of a block's PPB pieces per
// instruction (per-lane source addresses; the stream runs at the same 5.6-5.8 TB/s as on contiguous KiBs:
// tools/exp/exp_dma_patterns.hip) into the SAME LDS image as before.
// Behind the blocks, at element ACT_COLS * R, the SIGN BITS of the ReLU'd columns (bit
// 4Q+i of the uint16 of (block w, half-wave h) <-> feature 32w + 8Q + 4h + i is > 0), grouped the way the kernels produce
// and consume them: per trunk layer g a block [R][half-wave][16 B] at byte g * 32 R (the 8 uint16 of the layer's blocks),
// the colour head's 4 blocks as [R][half-wave][8 B] at byte 288 R.  The activation-gradient kernel gates with these instead
// of re-reading the activations.  A buffer is R * ACT_PITCH elements of either dtype (anr_mlp_act_cols()).
constexpr int ACT_COLS = 2432;
constexpr int ACT_PITCH = 2592;
__host__ __device__ constexpr int act_col(int T) { return T < 64 ? 32 * T : T < 73 ? 2048 + 32 * (T - 65) : 2304 + 32 * (T - 73); }
// byte offsets inside a buffer of R rows of ESZ-byte elements
__host__ __device__ constexpr int64_t act_block_off(int64_t R, int esz, int blk) { return (int64_t)blk * R * (32 * esz); }
__host__ __device__ constexpr int act_ppb(int esz) { return 2 * esz; }                       // 16-byte pieces per block row
// byte offset of piece `sub` (features sub * EPP ..) of block `blk`, row 0; rows are 16 bytes apart
__host__ __device__ constexpr int64_t act_piece_off(int64_t R, int esz, int blk, int sub) {
    return ((int64_t)blk * act_ppb(esz) + sub) * R * 16;
}
// byte offset of feature `f` (0..31) of block `blk`, row `r`
__host__ __device__ constexpr int64_t act_elem_off(int64_t R, int esz, int blk, int f, int64_t r) {
    return act_piece_off(R, esz, blk, f * esz / 16) + r * 16 + (f * esz) % 16;
}
#ifndef ANR_ACT_STORE_NT
#define ANR_ACT_STORE_NT 1          // the saved activations / activation gradients leave as nontemporal stores (full lines)
#endif
template <class V, class P> __device__ __forceinline__ void act_store(P* p, const V& v) {
#if ANR_ACT_STORE_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__host__ __device__ constexpr int64_t act_bits_off(int64_t R, int esz, int w) {
    return (int64_t)ACT_COLS * R * esz + (w < 72 ? (int64_t)(w / 8) * 32 * R : (int64_t)288 * R);
}
// A wave-uniform pointer that walks the blocks of such a buffer in the static order of a kernel's tile schedule.  Every
// block base is a loop invariant of the persistent point-tile loop: written as base + blk * stride, hipcc hoists all 76 of
// them out of the loop and, short of scalar registers, spills them as VGPR pairs — a scratch reload + vmcnt(0) in front of
// every store.  step() goes through an empty asm statement, so the pointer is recomputed where it is used (two SALU adds).
// (The pointer is typed as global memory: behind the asm hipcc can no longer infer that, and a generic pointer means
// flat_store, which also counts as an LDS operation.)
typedef __attribute__((address_space(1))) char gchar;
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u32x4n g_uint4;      // (native vectors: HIP's uint4 class has no operator= there)
typedef __attribute__((address_space(1))) u32x2n g_uint2;
typedef __attribute__((address_space(1))) f32x4 g_f32x4;
struct BlockWalk {
    gchar* p;
    int64_t stride;                                   // bytes per step: a block (the bits), or the two piece arrays one store instruction of a tile covers (the data)
    __device__ __forceinline__ void reset(char* base, int64_t first_block) {
        p = (gchar*)base + first_block * stride;
        asm volatile("" : "+s"(p));
    }
    template <int D> __device__ __forceinline__ void step() {
        if constexpr (D != 0) { p += (int64_t)D * stride; asm volatile("" : "+s"(p)); }
    }
};
// bit of feature 8Q + 4h + i (Q = accumulator quarter, i = 0..3) inside the uint16 of its block and half-wave
__host__ __device__ constexpr int act_sign_bit(int Q, int i) { return ((i & 1) ? 15 : 7) - (2 * Q + (i >> 1)); }
// ... plus, per lane: row * 32 * esz (+ the feature inside the block) for the data, row * 32 + 16 * half for a layer's
// bits, row * 16 + 8 * half for the head's

// PRE: the input is the 63-channel Fourier embedding itself, emb[n][63] fp32 (models/mlp.py:268-297 takes it that way):
// the encoder is skipped, the panel slots are loaded from the row.
// TAN (forward-mode normals, models/nerf.py:177-190 without autograd-of-autograd): the points come in quads — column
// 4p is point p itself, columns 4p+1..3 carry the tangents d/dx, d/dy, d/dz through the same layers: their encoding is the
// derivative of the encoding, they get no bias, and their ReLU gate is the PRIMAL column's (one DPP quad broadcast).
// sigma of a tangent column is then d sigma / d x_d.
// VIEW: the colour head takes the view direction (viewdir[n][stride], handed in where the rays-mode kernels take `rays`):
// its Fourier panel is computed like the position's and multiplied in front of the feature in the dir_encoding tiles.
// BITS_ONLY (with SAVE): only the ReLU sign bits are kept — what the activation-gradient kernel gates with — not the
// activations themselves, which only the WEIGHT gradients read: the `_refine` stage of the shipped configs trains the poses
// with the networks frozen (train.py:433-437, configs/people_snapshot/*_refine.yaml: pretrained_model_requires_grad False).
// FUSED (round 6, csrc/ray_march.hip): the tile machinery driven by the one-pass ray-march kernel — consecutive point-tile
// passes may belong to DIFFERENT networks (coarse, fine, fine, coarse, ...): the ring wraps to `gbase` = the NEXT pass's pack
// (set by the caller before each pass) and the last tile of a pass prefetches the next pass's first bias from `lds_bias_next`.
template <int MODE, bool DMA, bool SIGMA_ONLY = false, bool SAVE = false, bool PRE = false, bool TAN = false, bool VIEW = false,
          bool BITS_ONLY = false, bool FUSED = false>
struct Mlp {
    static_assert(!BITS_ONLY || SAVE, "BITS_ONLY is a variant of SAVE");
    static_assert(!FUSED || (DMA && !SIGMA_ONLY && !SAVE && !PRE && !TAN && !VIEW), "FUSED: the plain inference network");
    static_assert(!VIEW || (!SIGMA_ONLY && !SAVE && !PRE && !TAN), "the fused view-dependent head: inference, full network");
    using C = Cfg<MODE>;
    using Frag = typename C::Frag;
    static constexpr int NT = C::NT, EPF = C::EPF, HF = C::HF, EF = C::EF, DF = C::DF, WAVES = C::WAVES, TPC = C::TPC;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int FPT = 16 / EPF;          // next-layer frags produced per out-tile
    static constexpr int SLOT = slot_bytes<C>();
    using ActT = std::conditional_t<C::IS_BF16, __bf16, float>;    // saved activations: the dtype the next layer consumed
    // HALF TILES (round 6): a training pass at the per-rank batch of the reference's 8-GPU run (2 frames: 17-25 k listed rows)
    // is 66-100 point tiles of 256 rows — one per CU on a third of the chip, ~59 us each whatever their number, with two
    // wavefronts per SIMD taking turns on the matrix pipe.  When the listed rows fit HALF tiles of 128 rows on the launch's
    // workgroups (count on the device: ceil(n / 128) <= gridDim.x) a workgroup takes 128 rows instead: wavefronts 0-3, one
    // per SIMD, run the layer chain on 32 rows each, wavefronts 4-7 do nothing but stream the weight chunks (LDS-DMA issue,
    // barriers) — twice the CUs, half the multiplies per CU, the loads' issue off the multiplying wavefronts.  The rows'
    // values do not depend on which wavefront holds them: same bits.  Training-forward variants of the 8-wave shape only
    // (the inference instantiations keep their code).
    static constexpr bool HALFABLE = ANR_HALF_TILES && SAVE && !PRE && !TAN && !VIEW && DMA && C::WAVES == 8 && C::NT == 1;
    static constexpr int HALF_WAVES = 4;
    bool half_mode;          // HALFABLE: this launch runs in half tiles
    static constexpr int LAST_TILE = SIGMA_ONLY ? 64 : 77;         // sigma-only stops after the sigma row (tile 64)
    static constexpr int LAST_CHUNK = LAST_TILE / TPC;
    static constexpr int NCHUNK = LAST_CHUNK + 1;                  // chunks per pass (the slot pointers rotate at run time)

    // Per-wave pipeline state.  Weight chunks (TPC 32-row out-tiles each) flow through a 3-slot LDS ring:
    // while tile c is being multiplied, chunk c+1 is already resident (its first fragment group and its bias
    // are pulled into registers before tile c ends) and chunk c+2 is in flight on the LDS-DMA engine.
    // The activation/convert epilogue of tile c-1 is issued in the shadow of tile c's first MFMAs, so the
    // matrix pipe never waits for VALU work: accumulators ping-pong between acc[0] and acc[1].
    const char* gnext;       // global address of the next chunk to stage (chunk c+2)
    const char* gbase;       // first chunk of the pack (the ring wraps to it between point tiles)
    bool more;               // persistent loop: another point tile follows this one
    char* lds_base;          // the workgroup's dynamic LDS: [bias table | 3 ring slots]
    char* lds_bias;
    char* lds_bias_next;     // FUSED: the bias table of the NEXT pass's network
    unsigned slot_cur;       // byte offset (in lds_base) of the slot of chunk c
    unsigned slot_nxt;       // ... of chunk c+1
    unsigned slot_stage;     // ... of the slot chunk c+2 is staged into
    int c;                   // chunk counter
    int wave, lane, half;
    f32x16 acc[2][NT];       // accumulators of tile c (parity c&1) and of tile c-1 (epilogue pending)
    f32x16 bias_c;           // bias of tile c: C operand of its first MFMA
    // SAVE (training forward: the register file is full with the store staging): no bias registers — the bias of tile T+1
    // is read from LDS straight into the accumulators tile T+1 will use (free since tile T-1's epilogue, which ran behind
    // tile T's first MFMAs) and the tile accumulates onto it; 32 registers less, no spills in front of the stores
    static constexpr bool BIAS_IN_ACC = SAVE;
    Frag w0[4];              // first fragment group of tile c
    char* act_base;          // SAVE: the activation buffer (blocked layout above) and its row count
    int64_t act_rows;
    BlockWalk act_blk;       // SAVE: the data block / the layer's bits block the next epilogue stores into
    BlockWalk bits_blk;
    unsigned act_off[NT];    // SAVE: byte offset of this lane's row inside a piece array + its half-wave's array (row * 16 + half * 16 R); rows past the end alias the last row
    unsigned bits_off[NT];   // SAVE: ... inside a layer's bits block (row * 32 + 16 * half)
    char* lds_bits[NT];      // SAVE: this lane's 16 bytes of LDS where a layer's sign bits collect (one global store per layer)
    unsigned savebits[NT];   // SAVE: sign flags of the tile whose epilogue is pending

    // SAVE: stores this wave issues in the epilogue of tile V (per column tile): the 16-byte row pieces, and the layer's
    // sign bits behind its last tile
    static __host__ __device__ constexpr int epi_stores(int V) {
        if (V < 0 || V == 64 || V >= 77) return 0;
        const bool relu = V < 64 || V >= 73;
        const bool layer_end = relu && (V < 64 ? V % 8 == 7 : V == 76);
        return (BITS_ONLY ? 0 : (C::IS_BF16 ? 2 : 4)) + (layer_end ? 1 : 0);
    }
    // ... and between the LDS-DMA of chunk c+1 (issued by the advance() in front of tile T - TPC) and the advance() in
    // front of tile T, which needs that chunk: the epilogues that ran in between.  The vector-memory counter retires in
    // issue order, so `vmcnt(that many)` waits for the DMA and for nothing younger — the activation stores of the last
    // chunk stay in flight across the barrier instead of being drained (vmcnt(0)) every two tiles.  A lower bound is always
    // safe (T = 0 leaves out the output stores and the point prefetch of the previous point tile).
    static __host__ __device__ constexpr int stores_since_dma(int T) {
        if (!SAVE) return 0;
        const int first = T == 0 ? LAST_CHUNK * TPC : T - TPC, last = T == 0 ? LAST_TILE : T - 1;
        int n = 0;
        for (int U = first; U <= last; ++U) n += epi_stores(U - 1);
        return n * NT;
    }

    // barrier: chunk c+1 has landed everywhere and nobody reads chunk c-1 any more -> stage chunk c+2 over it
    static constexpr int MAXP = (TPC * (HF + EF) + WAVES - 1) / WAVES;   // register-staged pieces per wave (DMA off)
    uint4 sreg[DMA ? 1 : MAXP];
    unsigned spend_slot; int spend_nf;
    template <int T> __device__ __forceinline__ void advance() {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(stores_since_dma(T)) : "memory");
        else {
            // DMA off: the chunk loaded into registers one chunk ago goes to its ring slot now
#pragma unroll
            for (int i = 0; i < MAXP; ++i)
                if (wave + i * WAVES < spend_nf)
                    *reinterpret_cast<uint4*>(lds_base + spend_slot + (wave + i * WAVES) * FRAG_BYTES + lane * 16) = sreg[i];
            spend_nf = 0;
        }
        __syncthreads();
        // chunk c+2 of this pass, or — the workgroup is persistent — chunk 0/1 of the NEXT point tile: the weight
        // stream never drains between tiles.  Nothing may be in flight into LDS when the workgroup ends.
        if (c + 2 < NCHUNK || more) {
            if (c + 2 == NCHUNK) gnext = gbase;
            const int nf = chunk_frags<C, VIEW>((c + 2) % NCHUNK);
            if constexpr (HALFABLE) {
                if (!half_mode) stage_chunk<DMA, WAVES>(gnext, lds_base, slot_stage, nf, wave, lane);     // (half tiles: the helpers')
            } else if constexpr (DMA) stage_chunk<DMA, WAVES>(gnext, lds_base, slot_stage, nf, wave, lane);
            else {
#pragma unroll
                for (int i = 0; i < MAXP; ++i)
                    if (wave + i * WAVES < nf)
                        sreg[i] = *reinterpret_cast<const uint4*>(gnext + (wave + i * WAVES) * FRAG_BYTES + lane * 16);
                spend_slot = slot_stage; spend_nf = nf;
            }
            gnext += nf * FRAG_BYTES;
        }
    }
    __device__ __forceinline__ void rotate() {
        unsigned t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_stage; slot_stage = t;
        ++c;
    }
    __device__ __forceinline__ f32x16 read_bias(int tile_idx, bool next_pass = false) {
        const char* table = (FUSED && next_pass) ? lds_bias_next : lds_bias;
        const f32x4* b = reinterpret_cast<const f32x4*>(table + tile_idx * 128 + half * 64);
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t = b[q];
            v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
        }
        if constexpr (TAN) {                               // tangent columns are linear in the tangent: no bias
            const float keep = (lane & 3) ? 0.0f : 1.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] *= keep;
        }
        return v;
    }

    // ---- pending epilogues: what is still to be done with the accumulators of the PREVIOUS tile.
    // part<Q>() handles accumulator registers 4Q..4Q+3 of every column tile (Q = 0..3).
    struct NoEpi {
        template <int Q> __device__ __forceinline__ void part() const {}
    };
    // activation + conversion into fragments [TB, TB+FPT) of the next layer's input
    // TG = global tile index (selects the column block when activations are saved for the backward pass)
    template <bool RELU, int YF, int TB, int TG>
    struct FragEpi {
        const f32x16 (&a)[NT];
        Frag (&Y)[NT][YF];
        BlockWalk& ab;                   // SAVE: data / bits block of this tile, the lane's offsets inside them
        BlockWalk& bb;
        const unsigned (&ao)[NT];
        const unsigned (&bo)[NT];
        char* const (&lb)[NT];           // SAVE: the lane's LDS slot for the sign bits of the layer in progress
        unsigned (&sb)[NT];              // SAVE: the tile's sign flags, collected across its four parts
        int half;
        static constexpr int ESZ = sizeof(ActT);
        // sign bits of this tile (16: four quarters x four features) -> the lane's LDS slot; behind the layer's last tile the
        // slot leaves as ONE 16-byte store (the colour head's four blocks: 8 bytes) instead of a 2-byte store per tile
        __device__ __forceinline__ void put_bits(int n, unsigned bits) const {
            constexpr int w = act_col(TG) / 32, j = w < 72 ? w % 8 : w - 72;
            *reinterpret_cast<uint16_t*>(lb[n] + 2 * j) = (uint16_t)bits;
            if constexpr (w < 72 && j == 7) {
                *reinterpret_cast<g_uint4*>(bb.p + bo[n]) = *reinterpret_cast<const u32x4n*>(lb[n]);
                if (n == NT - 1) bb.template step<(w == 63) ? 2 : 1>();      // (the group of xyz_encoding_final holds no bits)
            } else if constexpr (w == 75) {
                *reinterpret_cast<g_uint2*>(bb.p + (bo[n] >> 1)) = *reinterpret_cast<const u32x2n*>(lb[n]);
            }
        }
        template <int Q> __device__ __forceinline__ void part() const {
            parts<Q>();
            // the tiles that store walk the piece arrays in order: two per store instruction (bf16: after quarters 1 and 3,
            // fp32: after every quarter)
            if constexpr (SAVE && !BITS_ONLY && (!C::IS_BF16 || (Q & 1))) ab.template step<1>();
        }
        template <int Q> __device__ __forceinline__ void parts() const {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (C::IS_BF16) {
                    // round the pairs to bf16 first, then ReLU both halves of a dword with one packed integer max
                    // (sign bit set <=> negative as int16): 2 + 2 instructions per 4 values instead of 4 + 2
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    typedef short s16x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    unsigned pk[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x2 v2 = {a[n][4 * Q + 2 * i], a[n][4 * Q + 2 * i + 1]};
                        const bf16x2 b2 = __builtin_convertvector(v2, bf16x2);
                        s16x2 s2 = __builtin_bit_cast(s16x2, b2);
                        if (RELU) {
                            if constexpr (TAN) {
                                // gate of the whole quad = the primal column's sign (lane 4q): keep where its ReLU'd value
                                // is > 0 — (0 - h) >> 15 per 16-bit half is 0xffff there, 0 for +0
                                const s16x2 prim = __builtin_bit_cast(s16x2, __builtin_amdgcn_update_dpp(
                                    0, __builtin_bit_cast(int, s2), 0x00 /* quad_perm:[0,0,0,0] */, 0xf, 0xf, false));
                                const s16x2 hp = __builtin_elementwise_max(prim, s16x2{0, 0});
                                s2 = s2 & ((s16x2{0, 0} - hp) >> 15);
                            } else {
                                s2 = __builtin_elementwise_max(s2, s16x2{0, 0});
                            }
                        }
                        pk[i] = __builtin_bit_cast(unsigned, s2);
                    }
                    Frag& dst = Y[n][TB + (4 * Q) / EPF];
                    u32x4 d4 = __builtin_bit_cast(u32x4, dst);
                    d4[((4 * Q) % EPF) / 2] = pk[0];
                    d4[((4 * Q) % EPF) / 2 + 1] = pk[1];
                    dst = __builtin_bit_cast(Frag, d4);
                    if ((4 * Q + 4) % EPF == 0) pin(dst);
                    // features 32t + 8Q + 4h + (0..3) of this lane's point: 8 contiguous bytes of its row
                    // (the two half-waves hold alternate 8-byte halves of a 16-byte piece: one v_permlane32_swap per dword hands
                    // the lower half-wave both halves of the even quarter's piece, the upper one both of the odd quarter's ->
                    // 16-byte stores, a whole piece per lane, 512 contiguous bytes per half-wave)
                    if constexpr (SAVE && !BITS_ONLY && (Q & 1)) {
                        const u32x4 prev = __builtin_bit_cast(u32x4, Y[n][TB + (4 * (Q - 1)) / EPF]);
                        constexpr int pd = ((4 * (Q - 1)) % EPF) / 2;
                        const auto s0 = __builtin_amdgcn_permlane32_swap(prev[pd], pk[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(prev[pd + 1], pk[1], false, false);
                        // (piece 2 (Q >> 1) + half of the block: ab walks two piece arrays per store, ao = row * 16 + half * R * 16)
                        act_store(reinterpret_cast<g_uint4*>(ab.p + ao[n]), u32x4n{s0[0], s1[0], s0[1], s1[1]});
                    }
                    if constexpr (SAVE && RELU) {
                        // after the ReLU a bf16 is > 0 exactly when its bit pattern is non-zero: min(x, 1) per 16-bit half is
                        // the flag, and (flags so far << 1) | flag collects pair k at bits 7 - k / 23 - k
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            unsigned fl;                  // (as asm: hipcc lowers the vector min to compares and selects)
                            asm("v_pk_min_u16 %0, %1, %2" : "=v"(fl) : "v"(pk[i]), "s"(0x00010001u));
                            sb[n] = (Q == 0 && i == 0) ? fl : ((sb[n] << 1) | fl);
                        }
                        if constexpr (Q == 3) put_bits(n, __builtin_amdgcn_perm(0u, sb[n], 0x0c0c0200u));     // bytes 0 and 2
                    }
                } else {
                    f32x4 keep;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = a[n][4 * Q + i];
                        if (RELU) {
                            if constexpr (TAN) {
                                const float prim = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, false));
                                v = prim > 0.0f ? v : 0.0f;
                            } else {
                                v = __int_as_float(max(__float_as_int(v), 0));   // relu as one v_max_i32
                            }
                        }
                        put(Y[n][TB + (4 * Q + i) / EPF], (4 * Q + i) % EPF, v);
                        keep[i] = v;
                    }
                    if ((4 * Q + 4) % EPF == 0) pin(Y[n][TB + (4 * Q) / EPF]);
                    if constexpr (SAVE && !BITS_ONLY) act_store(reinterpret_cast<g_f32x4*>(ab.p + ao[n]), keep);      // piece 2 Q + half
                    if constexpr (SAVE && RELU) {
                        unsigned nib = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) nib |= (keep[i] > 0.0f ? 1u : 0u) << act_sign_bit(Q, i);
                        sb[n] = Q == 0 ? nib : (sb[n] | nib);
                        if constexpr (Q == 3) put_bits(n, sb[n]);
                    }
                }
            }
        }
    };
    // sigma head: row 0 of its tile = accumulator register 0 of the lower half-wave
    struct SigmaEpi {
        const f32x16 (&a)[NT];
        float (&sigma)[NT];
        template <int Q> __device__ __forceinline__ void part() const {
            if (Q == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n) sigma[n] = a[n][0];
            }
        }
    };

    // One out-tile (parity PAR): acc[PAR] = bias + W_tile . [E (NFE frags), X (NFH frags)].
    // Weight fragments move LDS -> registers in groups of 4, double-buffered: group j+1 (or, in the last group,
    // the first group and the bias of the NEXT tile, already resident in the ring) loads while group j feeds the
    // matrix cores; the previous tile's epilogue (`pending`) is spread behind the first four MFMAs.
    template <int T, int NFE, int NFH, int XF, class Pending>
    __device__ __forceinline__ void tile(const Frag (&E)[NT][EF], const Frag (&X)[NT][XF], const Pending& pending) {
        static_assert((NFE + NFH) % 4 == 0, "fragment groups of 4");
        static_assert(NFE + NFH == tile_frags<C, VIEW>(T), "tile schedule mismatch");
        constexpr int NG = (NFE + NFH) / 4;
        constexpr int PAR = T & 1;
        constexpr int POS = T % TPC;                       // position of this tile inside its chunk
        constexpr int OFF = (POS == 0) ? 0 : tile_frags<C, VIEW>(T - 1);      // TPC <= 2
        constexpr bool END = (T == LAST_TILE);             // the next tile is tile 0 of the next point tile
        if constexpr (POS == 0) advance<T>();
        const Frag* cur = reinterpret_cast<const Frag*>(lds_base + slot_cur) + OFF * 64 + lane;
        const Frag* nxt = (POS + 1 < TPC && !END) ? cur + (NFE + NFH) * 64
                                                  : reinterpret_cast<const Frag*>(lds_base + slot_nxt) + lane;
        Frag wa[4], wb[4];
        f32x16 bias_n;
#pragma unroll
        for (int q = 0; q < 4; ++q) wa[q] = w0[q];
        static_for<NG>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            Frag (&use)[4] = (j & 1) ? wb : wa;
            Frag (&ld)[4] = (j & 1) ? wa : wb;
#pragma unroll
            for (int q = 0; q < 4; ++q) ld[q] = (j + 1 < NG) ? cur[((j + 1) * 4 + q) * 64] : nxt[q * 64];
            if constexpr (j + 1 == NG && !BIAS_IN_ACC) bias_n = read_bias(END ? 0 : T + 1, END);
            __builtin_amdgcn_sched_barrier(0);          // the loads above are issued before this group's MFMAs
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int f = 4 * j + q;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const Frag& x = (f < NFE) ? E[n][f < NFE ? f : 0] : X[n][f >= NFE ? f - NFE : 0];
                    if (f == 0 && !BIAS_IN_ACC) acc[PAR][n] = mma_c(use[q], x, bias_c);
                    else                        mma(use[q], x, acc[PAR][n]);
                }
                // The four quarters of the previous tile's epilogue ride behind MFMAs of the first TWO groups (one
                // quarter per two MFMA steps): packed into one group they need more issue slots than its MFMAs leave.
                // (A two-group tile consumes the fragments the epilogue produces in its second group: keep one group.)
                if constexpr (NG >= 3) {
                    if (j < 2 && (q & 1)) {
                        pending.template part<2 * (j < 2 ? j : 0) + (q >> 1)>();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (j == 0) {
                    pending.template part<q>();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // BIAS_IN_ACC: the other accumulator set is free once the pending epilogue has read it (groups 0 and 1;
                // all of it in group 0 of a short tile): the next tile's bias goes there under the last group's MFMAs
                if constexpr (BIAS_IN_ACC && !END && j + 1 == NG && q == (NG == 1 ? 3 : 0)) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[PAR ^ 1][n] = read_bias(T + 1);
                }
            });
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) w0[q] = (NG & 1) ? wb[q] : wa[q];
        if constexpr (!BIAS_IN_ACC) bias_c = bias_n;
        if constexpr (POS + 1 == TPC || END) rotate();
    }

    // a full layer of NTILES out-tiles starting at global tile index T0; `first` is the epilogue still pending
    // from the tile before T0.  The epilogue of this layer's LAST tile is left pending for the caller.
    template <int T0, int NTILES, int NFE, int NFH, bool RELU, int XF, int YF, class Pending>
    __device__ __forceinline__ void layer(const Frag (&E)[NT][EF], const Frag (&X)[NT][XF], Frag (&Y)[NT][YF],
                                          const Pending& first) {
        static_for<NTILES>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int PAR = (T0 + t) & 1;
            if constexpr (t == 0) {
                tile<T0 + t, NFE, NFH, XF>(E, X, first);
            } else {
                tile<T0 + t, NFE, NFH, XF>(E, X, FragEpi<RELU, YF, (t - 1) * FPT, T0 + t - 1>{acc[PAR ^ 1], Y, act_blk, bits_blk, act_off, bits_off, lds_bits, savebits, half});
            }
        });
    }
    template <int T0, int NTILES, bool RELU, int YF>
    __device__ __forceinline__ auto last_of(Frag (&Y)[NT][YF]) {
        return FragEpi<RELU, YF, (NTILES - 1) * FPT, T0 + NTILES - 1>{acc[(T0 + NTILES - 1) & 1], Y, act_blk, bits_blk, act_off, bits_off, lds_bits, savebits, half};
    }

    // The Fourier panel of a 3-vector as B fragments (slot map in the header comment): the position's, and with VIEW the view
    // direction's (its weights are zero in the slots of the octaves encoding_dir does not have).
    __device__ __forceinline__ void encode_panel(const float (&xs)[3], Frag (&dst)[EF]) const {
        if constexpr (C::IS_BF16) {
            // bf16 mode: exact sin/cos of the base band, then angle doubling (error doubles per octave:
            // 2^9 * 1e-7 << bf16's 2^-9) — 6 polynomial evaluations instead of 30 per lane
            float sn[3], cs[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { sn[d] = sin_or_cos(xs[d], 0); cs[d] = sin_or_cos(xs[d], 1); }
#pragma unroll
            for (int k = 0; k < 10; ++k) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int j = 3 * k + d;
                    put(dst[j / EPF], j % EPF, half ? cs[d] : sn[d]);
                    const float s2 = 2.0f * sn[d] * cs[d], c2 = 1.0f - 2.0f * sn[d] * sn[d];
                    sn[d] = s2; cs[d] = c2;
                }
            }
            put(dst[30 / EPF], 30 % EPF, half ? xs[2] : xs[0]);
            put(dst[31 / EPF], 31 % EPF, half ? 0.0f : xs[1]);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float v;
                if (j < 30) {
                    const int k = j / 3, d = j % 3;
                    v = sin_or_cos(xs[d] * (float)(1 << k), half);
                } else if (j == 30) {
                    v = half ? xs[2] : xs[0];
                } else {
                    v = half ? 0.0f : xs[1];
                }
                put(dst[j / EPF], j % EPF, v);
            }
        }
    }

    // Half tiles: what wavefronts 4-7 do — the chunk sequence of ONE point tile (a half-tile launch has one per workgroup), in
    // step with the workers' barriers: wait for the own pieces of chunk c+1, barrier, stage chunk c+2 over chunk c-1.
    __device__ __forceinline__ void stream_chunks() {
        dma_wait();
        __syncthreads();                                   // (the workers' `first` barrier)
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_base;
        for (int cc = 0; cc < NCHUNK; ++cc) {
            dma_wait();
            __syncthreads();                               // (advance() of the chunk's first tile)
            if (cc + 2 < NCHUNK) {
                const int nf = chunk_frags<C, VIEW>(cc + 2);
                for (int p = wave - HALF_WAVES; p < nf; p += WAVES - HALF_WAVES)
                    dma16(gnext + p * FRAG_BYTES + lane * 16, base + slot_stage + p * FRAG_BYTES);
                gnext += nf * FRAG_BYTES;
            }
            const unsigned t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_stage; slot_stage = t;
        }
        dma_wait();
    }

    // the row a lane past the end of the list works on instead (tangent mode: the row of the last quad with its own role)
    static __device__ __forceinline__ int64_t clamp_row(int64_t idx, int64_t n) {
        return idx < n ? idx : (TAN ? n - 4 + (idx & 3) : n - 1);
    }

    // index/count (both optional): evaluate pts[index[i]] for i < min(n_pts, *count) and write out[index[i]] — the
    // compacted list of valid samples (anr_compact_valid); saved activations are rows i of the compacted order.
    __device__ __forceinline__ void run(const char* __restrict__ pack, const float4* __restrict__ pts, int64_t n_pts,
                                        void* __restrict__ out_v, float* __restrict__ act, char* lds,
                                        const int32_t* __restrict__ index, const int32_t* __restrict__ count,
                                        const float* __restrict__ rays, int ray_stride, int K) {
        act_rows = n_pts;                                  // the buffer's row count (blocks are [n][32]), not the listed count
        if (count) {
            const int64_t cnt = *count;
            n_pts = cnt < n_pts ? cnt : n_pts;
        }
        half_mode = false;
        if constexpr (HALFABLE) half_mode = (n_pts + HALF_WAVES * 32 - 1) / (HALF_WAVES * 32) <= (int64_t)gridDim.x;
        // wavefronts that hold rows of a point tile
        const int wpt = HALFABLE ? (half_mode ? HALF_WAVES : WAVES) : WAVES;
        const int64_t n_tiles = (n_pts + wpt * NT * 32 - 1) / (wpt * NT * 32);
        if ((int64_t)blockIdx.x >= n_tiles) return;       // (before anything is in flight into LDS)
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        half = lane >> 5;
        lds_base = lds;
        lds_bias = lds;
        act_base = reinterpret_cast<char*>(act);
        slot_cur = BIAS_BYTES;
        slot_nxt = slot_cur + SLOT;
        slot_stage = slot_nxt + SLOT;
        gbase = pack + BIAS_BYTES;

        // resident bias table + the first two chunks
        for (int i = threadIdx.x; i < BIAS_BYTES / 16; i += THREADS)
            reinterpret_cast<uint4*>(lds_bias)[i] = reinterpret_cast<const uint4*>(pack)[i];
        gnext = gbase;
        spend_nf = 0; spend_slot = 0;
        stage_chunk<DMA, WAVES>(gnext, lds_base, slot_cur, chunk_frags<C, VIEW>(0), wave, lane);
        gnext += chunk_frags<C, VIEW>(0) * FRAG_BYTES;
        stage_chunk<DMA, WAVES>(gnext, lds_base, slot_nxt, chunk_frags<C, VIEW>(1), wave, lane);
        gnext += chunk_frags<C, VIEW>(1) * FRAG_BYTES;

        // Persistent workgroup: point tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The weight ring, the bias table,
        // w0 and bias_c carry over from one tile to the next.
        bool first = true;
        auto fetch_pts = [&](int64_t tile_idx, float4 (&dst)[NT]) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                int64_t idx = (tile_idx * wpt + wave) * (NT * 32) + n * 32 + (lane & 31);
                idx = clamp_row(idx, n_pts);
                if (index) {
                    // every listed sample is valid: the w lane carries its position in pts/out instead
                    const int32_t id = index[idx];
                    dst[n] = pts[id];
                    dst[n].w = __int_as_float(id);
                } else if (!VIEW && rays) {
                    // no warp (use_unpose=False): the sample point is generated here, x = o + z d with the product and
                    // the sum rounded separately like anr_points_from_rays; `pts` is then the depth array z[n]
                    // K < 0: no depth array either — `pts` is the step table s[|K|] of the deterministic stratified
                    // depths, z = near' (1 - s) + far' s with the roundings of anr_sample_coarse
                    const uint32_t Ka = (uint32_t)(K < 0 ? -K : K);
                    const uint32_t ray = (uint32_t)idx / Ka;
                    const float* ry = rays + (int64_t)ray * ray_stride;
                    float zz;
                    if (K < 0) {
                        const float sk = reinterpret_cast<const float*>(pts)[(uint32_t)idx - ray * Ka];
                        float one_minus, lo, hi;
                        asm("v_sub_f32_e32 %0, 1.0, %1" : "=v"(one_minus) : "v"(sk));
                        asm("v_mul_f32_e32 %0, %1, %2" : "=v"(lo) : "v"(ry[6]), "v"(one_minus));
                        asm("v_mul_f32_e32 %0, %1, %2" : "=v"(hi) : "v"(ry[7]), "v"(sk));
                        zz = lo + hi;
                    } else {
                        zz = reinterpret_cast<const float*>(pts)[idx];
                    }
                    // (the product goes through an asm statement: hipcc would contract it into an fma otherwise)
                    float m[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) asm("v_mul_f32_e32 %0, %1, %2" : "=v"(m[a]) : "v"(zz), "v"(ry[3 + a]));
                    dst[n] = make_float4(ry[0] + m[0], ry[1] + m[1], ry[2] + m[2], 1.0f);
                } else {
                    dst[n] = pts[idx];
                }
            }
        };
        if constexpr (HALFABLE) {
            if (half_mode && wave >= HALF_WAVES) { stream_chunks(); return; }
        }
        float4 p_cur[NT], p_nxt[NT];
        if constexpr (PRE) {
#pragma unroll
            for (int n = 0; n < NT; ++n) p_nxt[n] = make_float4(0.f, 0.f, 0.f, 1.f);
        } else {
            fetch_pts(blockIdx.x, p_nxt);
        }
        for (int64_t pt = blockIdx.x; pt < n_tiles; pt += gridDim.x) {
        more = pt + gridDim.x < n_tiles;
        c = 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) p_cur[n] = p_nxt[n];
        if constexpr (!PRE)
            if (more) fetch_pts(pt + gridDim.x, p_nxt);      // the next tile's points arrive under this tile's MFMAs
        // this wave's points, Fourier-encoded straight into B fragments
        const int64_t wave_base = (pt * wpt + wave) * (NT * 32);
        float valid[NT];
        Frag E[NT][EF];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            int64_t idx = wave_base + n * 32 + (lane & 31);
            const float4 p = p_cur[n];
            valid[n] = p.w;
            // (a lane past the end recomputes the row it aliases — clamp_row — and stores the same bytes there: every wave
            // issues the same number of stores whatever n is, which the counted waits of advance() rely on)
            if constexpr (SAVE) {
                const unsigned row = (unsigned)clamp_row(idx, n_pts);
                act_off[n] = row * 16u + (unsigned)half * (unsigned)(act_rows * 16);
                bits_off[n] = row * 32 + 16 * half;
            }
            if constexpr (SAVE) lds_bits[n] = lds + BIAS_BYTES + 3 * SLOT + ((n * WAVES + wave) * 64 + lane) * 16;
            const float xs[3] = {p.x, p.y, p.z};
            if constexpr (PRE) {
                int64_t row_i = clamp_row(idx, n_pts);
                const float* row = reinterpret_cast<const float*>(pts) + row_i * 63;
                valid[n] = 1.0f;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    // slot j, half h <-> reference channel (header): 3 + 6k + 3h + d for j = 3k + d < 30; x0/x2; x1/pad
                    float v;
                    if (j < 30) v = row[3 + 6 * (j / 3) + (j % 3) + 3 * half];
                    else if (j == 30) v = row[2 * half];
                    else v = half ? 0.0f : row[1];
                    put(E[n][j / EPF], j % EPF, v);
                }
            } else if constexpr (TAN) {
                // column t = lane & 3: 0 = the point, t = 1..3 = d/dx_{t-1} of the encoding:
                //   d sin(f x_d) = f cos(f x_d),  d cos(f x_d) = -f sin(f x_d),  d x_d = 1;  0 for the other axes
                const int t = lane & 3;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    float v;
                    if (j < 30) {
                        const int k = j / 3, d = j % 3;
                        const float f = (float)(1 << k), a = xs[d] * f;
                        const float prim = sin_or_cos(a, half);
                        const float tang = half ? -f * sin_or_cos(a, 0) : f * sin_or_cos(a, 1);
                        v = t == 0 ? prim : (t - 1 == d ? tang : 0.0f);
                    } else {
                        const int d = (j == 30) ? (half ? 2 : 0) : (half ? -1 : 1);
                        v = d < 0 ? 0.0f : (t == 0 ? xs[d < 0 ? 0 : d] : (t - 1 == d ? 1.0f : 0.0f));
                    }
                    put(E[n][j / EPF], j % EPF, v);
                }
            } else {
                encode_panel(xs, E[n]);
            }
        }

        if (first) {
            // chunk 0 resident -> its first fragment group and bias into registers
            if constexpr (DMA) dma_wait();
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) w0[q] = (reinterpret_cast<const Frag*>(lds_base + slot_cur) + lane)[q * 64];
            if constexpr (!BIAS_IN_ACC) bias_c = read_bias(0);
            first = false;
        }
        if constexpr (SAVE) {
            act_blk.stride = 2 * 16 * act_rows;               // two piece arrays per store instruction
            bits_blk.stride = 32 * act_rows;
            act_blk.reset(act_base, 0);
            bits_blk.reset(act_base + act_bits_off(act_rows, sizeof(ActT), 0), 0);
        }
        if constexpr (BIAS_IN_ACC) {                        // tile 0 accumulates onto its bias (every point tile: the last tile
#pragma unroll                                              // of the pass before may still have been using acc[0])
            for (int n = 0; n < NT; ++n) acc[0][n] = read_bias(0);
        }

        Frag A[NT][HF], B[NT][HF];
        float sigma[NT];
        // trunk: 8 layers of 8 tiles, activations ping-pong between A and B (global tile index in <>)
        layer<0, 8, EF, 0, true, HF, HF>(E, B, A, NoEpi{});                                      // 1: enc -> A
        layer<8, 8, 0, HF, true, HF, HF>(E, A, B, last_of<0, 8, true>(A));                       // 2: A -> B
        layer<16, 8, 0, HF, true, HF, HF>(E, B, A, last_of<8, 8, true>(B));                      // 3: B -> A
        layer<24, 8, 0, HF, true, HF, HF>(E, A, B, last_of<16, 8, true>(A));                     // 4: A -> B
        layer<32, 8, EF, HF, true, HF, HF>(E, B, A, last_of<24, 8, true>(B));                    // 5: [enc, B] -> A
        layer<40, 8, 0, HF, true, HF, HF>(E, A, B, last_of<32, 8, true>(A));                     // 6: A -> B
        layer<48, 8, 0, HF, true, HF, HF>(E, B, A, last_of<40, 8, true>(B));                     // 7: B -> A
        layer<56, 8, 0, HF, true, HF, HF>(E, A, B, last_of<48, 8, true>(A));                     // 8: A -> B
        // sigma row: one tile on h8 (= B).  models/anim_nerf.py:305 masks it where the warp was invalid.
        tile<64, 0, HF, HF>(E, B, last_of<56, 8, true>(B));
        if constexpr (SIGMA_ONLY) {
            float* out = reinterpret_cast<float*>(out_v);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                int64_t idx = wave_base + n * 32 + (lane & 31);
                if (half == 0 && idx < n_pts) {
                    if (index) out[__float_as_int(valid[n])] = acc[64 & 1][n][0];
                    else out[idx] = (valid[n] < 1.0f) ? -1e5f : acc[64 & 1][n][0];
                }
            }
        } else {
            float4* out = reinterpret_cast<float4*>(out_v);
            // xyz_encoding_final (no activation): B -> A
            layer<65, 8, 0, HF, false, HF, HF>(E, B, A, SigmaEpi{acc[64 & 1], sigma});
            // dir_encoding: A -> G (256 -> 128, relu); rgb: G -> 3, sigmoid
            Frag G[NT][DF];
            if constexpr (VIEW) {
                // [feature, encoding_dir(viewdir)] -> 128: the direction's panel in front, like the skip layer's (the pack puts
                // dir_encoding's last columns there).  Computed here, not at the top: 16 registers live for four tiles only.
                Frag Ed[NT][EF];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int64_t idx = wave_base + n * 32 + (lane & 31);
                    const int64_t row = index ? (int64_t)__float_as_int(valid[n]) : clamp_row(idx, n_pts);
                    const float* vd = rays + row * ray_stride;
                    const float dv[3] = {vd[0], vd[1], vd[2]};
                    encode_panel(dv, Ed[n]);
                }
                layer<73, 4, EF, HF, true, HF, DF>(Ed, A, G, last_of<65, 8, false>(A));
            } else {
                layer<73, 4, 0, HF, true, HF, DF>(E, A, G, last_of<65, 8, false>(A));
            }
            tile<77, 0, DF, DF>(E, G, last_of<73, 4, true>(G));
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                int64_t idx = wave_base + n * 32 + (lane & 31);
                const f32x16& r = acc[77 & 1][n];
                if (half == 0 && idx < n_pts) {
                    float cr = 1.0f / (1.0f + expf(-r[0]));
                    float cg = 1.0f / (1.0f + expf(-r[1]));
                    float cb = 1.0f / (1.0f + expf(-r[2]));
                    if (index) {
                        out[__float_as_int(valid[n])] = make_float4(cr, cg, cb, sigma[n]);
                    } else {
                        float s = (valid[n] < 1.0f) ? -1e5f : sigma[n];   // models/anim_nerf.py:305
                        out[idx] = make_float4(cr, cg, cb, s);
                    }
                }
            }
        }
        }   // persistent loop over point tiles
    }
};

// ---- layout of the BACKWARD weight pack (anr_mlp_bwd_pack, csrc/mlp_bwd.hip): [w_sigma table | W^T fragments of the 76
// out-tiles of the activation-gradient kernel | encoding panels]
constexpr int BWD_TABLE_BYTES = 1024;          // w_sigma, 256 fp32, in accumulator-register order per tile
constexpr int BWD_TILES = 76;
template <class C> __host__ __device__ constexpr int btile_frags(int t) { return t < 4 ? 4 : t < 12 ? C::DF : C::HF; }
template <class C> __host__ __device__ constexpr int bfrag_offset(int t) {
    int n = 0;
    for (int i = 0; i < t; ++i) n += btile_frags<C>(i);
    return n;
}
template <class C> constexpr int btotal_frags() { return bfrag_offset<C>(BWD_TILES); }
// Behind the W^T fragments: the two 256 x 64 ENCODING panels of layers 1 and 5 (xyz_encoding_1 [256,63], the first 63
// columns of xyz_encoding_5 [256,319]) as MFMA B fragments [layer][k-fragment][column tile 2][lane 64][EPL elements] — the
// operand of dL/d enc = dact_1 W1[:, :63] + dact_5 W5[:, :63] (anr_mlp_dpoints, csrc/mlp_wgrad.hip), made once per
// optimiser step by the pack kernel instead of once per workgroup by its consumer.  32,768 elements in either mode.
constexpr int DENC_PANEL_ELEMS = 32768;
template <class C> __host__ __device__ constexpr int64_t denc_panel_off() { return BWD_TABLE_BYTES + (int64_t)btotal_frags<C>() * FRAG_BYTES; }
template <class C> __host__ __device__ constexpr int64_t denc_panel_bytes() { return (int64_t)DENC_PANEL_ELEMS * (C::IS_BF16 ? 2 : 4); }

template <int MODE, bool DMA, bool SIGMA_ONLY, bool SAVE, bool PRE = false, bool TAN = false, bool VIEW = false, bool BITS_ONLY = false>
__global__ __launch_bounds__(Cfg<MODE>::WAVES * 64, Cfg<MODE>::WAVES / 4) void mlp_kernel(const char* __restrict__ pack,
                                                             const float4* __restrict__ pts, int64_t n_pts,
                                                             void* __restrict__ out, float* __restrict__ act,
                                                             const int32_t* __restrict__ index,
                                                             const int32_t* __restrict__ count,
                                                             const float* __restrict__ rays, int ray_stride, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Mlp<MODE, DMA, SIGMA_ONLY, SAVE, PRE, TAN, VIEW, BITS_ONLY> m;
    m.run(pack, pts, n_pts, out, act, lds, index, count, rays, ray_stride, K);
}

template <int MODE, bool DMA, bool SIGMA_ONLY, bool SAVE, bool PRE = false, bool TAN = false, bool VIEW = false, bool BITS_ONLY = false>
int launch_mlp(const void* pack, const float* pts, int64_t n, float* out, hipStream_t st, float* act,
               const int32_t* index = nullptr, const int32_t* count = nullptr, const float* rays = nullptr,
               int ray_stride = 0, int K = 1) {
    using C = Cfg<MODE>;
    const int lds = BIAS_BYTES + 3 * slot_bytes<C>() + (SAVE ? C::NT * C::WAVES * 64 * 16 : 0);     // + the sign-bit slots
    auto kern = mlp_kernel<MODE, DMA, SIGMA_ONLY, SAVE, PRE, TAN, VIEW, BITS_ONLY>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail((int)e, "anr_mlp_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
    // (the training-forward variants of the 8-wave shape may run in half tiles of 128 rows: Mlp::HALFABLE — the grid must
    // hold those when the buffer itself is that small; the listed rows are counted on the device)
    constexpr bool halfable = Mlp<MODE, DMA, SIGMA_ONLY, SAVE, PRE, TAN, VIEW, BITS_ONLY>::HALFABLE;
    const int pts_per_wg = halfable ? 128 : C::WAVES * C::NT * 32;
    const int64_t n_tiles = (n + pts_per_wg - 1) / pts_per_wg;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    dim3 grid((unsigned)(n_tiles < cus ? n_tiles : cus));          // one persistent workgroup per CU (LDS-limited)
    hipLaunchKernelGGL(kern, grid, dim3(C::WAVES * 64), lds, st, reinterpret_cast<const char*>(pack),
                       reinterpret_cast<const float4*>(pts), n, reinterpret_cast<void*>(out), act, index, count, rays, ray_stride, K);
    return check_launch("anr_mlp_forward");
}


}  // namespace anr
