// Backward of the fused MLP w.r.t. its activations (training, a16): the chain
//     dG = (Wr^T d_rgb') . 1[G>0]      dF = Wd^T dG      dh8' = (Wf^T dF + w_sigma d_sigma) . 1[h8>0]
//     dh'_{l-1} = (W_l^T dh'_l) . 1[h_{l-1}>0]    l = 8..2          (models/nerf.py:129-175 differentiated)
// as ONE kernel built like the forward one (mlp_core.h): transposed GEMMs on the matrix cores with the points as
// columns, the gradient fragments never leaving registers between the 10 GEMMs, W^T tiles streamed L2 -> LDS through
// the same 3-slot LDS-DMA ring, persistent workgroups.  The ReLU gates are the sign bits the training forward left behind
// each row (anr_mlp_forward_save: 16 bytes per row half and layer); a wave gathers them for the NEXT stage with one more
// LDS-DMA instruction (per-lane source address, its own 1 KiB of LDS) issued next to the weight chunk, so that no ordinary
// load — whose wait would drain the kernel's stores — sits in the steady state.  Every pre-activation gradient is written
// in the same blocked layout as the activations (dact: mlp_core.h), which is what the weight-gradient kernel
// (dW_l = dact_l^T act_{l-1}) consumes.
// The stores are never waited for: the wait in front of a barrier is `vmcnt(stores issued since the DMA it needs)`.
//
// Tile schedule (76 out-tiles of 32 rows): 0-3 rgb^T (K = 3, padded to one fragment group), 4-11 dir^T (K = 128),
// 12-19 final^T (+ the rank-1 sigma term as the C operand of the first MFMA), 20-75 trunk layers 8..2 transposed.
// sigma-only queries (foreground / background priors) start at tile 20 from dh8' = w_sigma d_sigma . 1[h8>0].
#include "mlp_core.h"

namespace anr {

// column (block = column / 32 of the blocked buffers) of the activation whose sign gates out-tile t, and where its result is stored
__host__ __device__ constexpr int bcol(int t) {
    return t < 4 ? 2304 + 32 * t : t < 12 ? 2048 + 32 * (t - 4) : 256 * (7 - (t - 12) / 8) + 32 * ((t - 12) % 8);
}
__host__ __device__ constexpr bool bmasked(int t) { return !(t >= 4 && t < 12); }      // xyz_encoding_final has no ReLU

// START: the first tile of the chain = where the upstream gradient enters: 0 (d rgb' and d sigma), 12 (FEATURE: d
// xyz_encoding_final[256] and d sigma — the view-dependent colour head, use_view=True, lives outside the kernels and hands
// over the gradient of its input), 20 (sigma only).
// ENC_ONLY (ANR_MLP_FLAG_ENC_ONLY): the networks are frozen and only the gradient towards the sample points is wanted —
// dact is written where anr_mlp_denc reads it (the gradients of layers 1 and 5: tiles 68..75 and 36..43) and nowhere else.
__host__ __device__ constexpr bool bstored(int t, bool enc_only) { return !enc_only || (t >= 36 && t < 44) || (t >= 68 && t < 76); }

template <int MODE, int START, bool ENC_ONLY = false>
struct MlpBwd {
    static constexpr bool SIGMA_ONLY = START == 20, FEATURE = START == 12;
    using C = Cfg<MODE>;
    using Frag = typename C::Frag;
    static constexpr int NT = C::NT, EPF = C::EPF, HF = C::HF, DF = C::DF, WAVES = C::WAVES, TPC = C::TPC;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int FPT = 16 / EPF;
    static constexpr int SLOT = TPC * HF * FRAG_BYTES;
    static constexpr int FIRST = START;
    static constexpr int LAST = BWD_TILES - 1;
    static constexpr int NCHUNK = (BWD_TILES - FIRST) / TPC;
    using ActT = std::conditional_t<C::IS_BF16, __bf16, float>;
    using MaskT = std::conditional_t<C::IS_BF16, uint2, f32x4>;      // four saved activations (one accumulator quarter)

    static __host__ __device__ constexpr int chunk_frags(int c) {
        int n = 0;
        for (int i = 0; i < TPC; ++i) n += btile_frags<C>(FIRST + c * TPC + i);
        return n;
    }

    // half tiles of 128 rows for small batches (mlp_core.h: Mlp::HALFABLE): wavefronts 0-3 run the chain, 4-7 stream the chunks
    static constexpr bool HALFABLE = ANR_HALF_TILES && C::WAVES == 8 && C::NT == 1;
    static constexpr int HALF_WAVES = 4;
    bool half_mode;
    const char* gnext;
    const char* gbase;
    bool more;
    char* lds_base;
    unsigned slot_cur, slot_nxt, slot_stage;
    int c;
    int wave, lane, half;
    f32x16 acc[2][NT];
    Frag w0[4];
    unsigned mk[2][NT];          // sign bits (16: four quarters x four rows) of tile c (parity) and of tile c-1 (pending epilogue)
    unsigned mkg[NT][4];         // ... of the 8 (4) blocks of the current stage: one 16-byte LDS read per lane and stage
    const char* act_base;        // the saved activations (only their sign bits are read here) and the gradient buffer: R rows
    char* dact_base;
    int64_t R;
    BlockWalk dact_blk;          // the block of dact the next epilogue stores into (mlp_core.h)
    unsigned gate_off[NT];       // row * 32 + 16 * half of the row whose sign bits gate this lane's column, and of the next point tile's
    unsigned gate_off_next[NT];
    unsigned dact_off[NT];       // byte offset of this lane's row inside a piece array of dact + its half-wave's array (row * 16 +
                                 // half * 16 R; rows past the end alias the last row: same values, same bytes)
    char* lds_bits;              // this wave's [2 buffers][NT][64 lanes x 16 B] of gathered sign bits
    static constexpr int ESZ = sizeof(ActT);

    // ---- the sign-bit groups ("bit stages") one point tile consumes, in order: START = 0: the colour head's (block 72, tile
    // 0), then h8 (blocks 56.., tile 12), h7 (tile 20) ... h1 (tile 68); START = 12: h8 (tile 12) ...; START = 20: h8 (the
    // prologue), h7 (tile 20) ...  Stage k lives in buffer k & 1.
    static constexpr int NBITS = START == 0 ? 9 : 8;
    static constexpr int LAST_ADV = BWD_TILES - TPC;                    // the tile in front of which the last advance() runs
    static __host__ __device__ constexpr int bits_w(int k) { return START == 0 ? (k == 0 ? 72 : 64 - 8 * k) : 56 - 8 * k; }
    static __host__ __device__ constexpr int bits_tile(int k) {
        return START == 0 ? (k == 0 ? 0 : 4 + 8 * k) : START == 12 ? 12 + 8 * k : (k == 0 ? -1 : 12 + 8 * k);
    }
    static __host__ __device__ constexpr int bits_stage_of_w(int w) { return START == 0 ? (w == 72 ? 0 : (64 - w) / 8) : (56 - w) / 8; }
    // a gather issued by the advance() in front of tile X has landed once the NEXT advance() has waited (it is older than
    // the weight chunk that one waits for): stage k is issued one chunk ahead of the tile that reads it, wrapping into the
    // previous point tile (then for the next tile's rows) for the stages read at the start
    static __host__ __device__ constexpr bool bits_wrapped(int k) { return bits_tile(k) - TPC < FIRST; }
    static __host__ __device__ constexpr int bits_issue_tile(int k) {
        if (bits_tile(k) < 0) return LAST_ADV - TPC;                    // the prologue reads it before the first advance()
        return bits_wrapped(k) ? bits_tile(k) - TPC + (BWD_TILES - FIRST) : bits_tile(k) - TPC;
    }
    template <int K> __device__ __forceinline__ void issue_bits(const unsigned (&rows)[NT]) {
        constexpr int W = bits_w(K);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            // the head's four blocks are 8 bytes per half-wave: both halves fetch the row's 16 and pick their own on the way out
            const char* src = act_base + act_bits_off(R, ESZ, W) + (W < 72 ? rows[n] : (rows[n] >> 1) - 8 * half);
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_bits + (((K & 1) * NT + n) << 10);
            dma16(src, dst);
        }
    }
    template <int K> __device__ __forceinline__ void read_bits(int n) {
        const uint4 v = *reinterpret_cast<const uint4*>(lds_bits + (((K & 1) * NT + n) << 10) + lane * 16);
        if constexpr (bits_w(K) < 72) {
            mkg[n][0] = v.x; mkg[n][1] = v.y; mkg[n][2] = v.z; mkg[n][3] = v.w;
        } else {
            mkg[n][0] = half ? v.z : v.x; mkg[n][1] = half ? v.w : v.y;
        }
    }
    // the uint16 of block J of the stage in mkg; bf16: its two bytes spread to bytes 0 and 2 (what MaskEpi shifts)
    template <int J> __device__ __forceinline__ unsigned block_mask(int n) const {
        if constexpr (C::IS_BF16) return __builtin_amdgcn_perm(0u, mkg[n][J / 2], (J & 1) ? 0x0c030c02u : 0x0c010c00u);
        else return (mkg[n][J / 2] >> (16 * (J & 1))) & 0xffffu;
    }
    // stores issued between the DMA of the chunk the advance() in front of tile T waits for and that wait: the epilogues
    // of the chunk before (the vector-memory counter retires in issue order; a lower bound is always safe)
    static __host__ __device__ constexpr int stores_since_dma(int T) {
        const int first = T == FIRST ? BWD_TILES - TPC : T - TPC, last = T == FIRST ? BWD_TILES - 1 : T - 1;
        int n = 0;
        for (int U = first; U <= last; ++U) n += (U - 1 >= FIRST && bstored(U - 1, ENC_ONLY)) ? (C::IS_BF16 ? 2 : 4) : 0;
        return n * NT;
    }
    template <int T> __device__ __forceinline__ void advance() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(stores_since_dma(T)) : "memory");
        __syncthreads();
        if (c + 2 < NCHUNK || more) {
            if (c + 2 == NCHUNK) gnext = gbase;
            const int nf = chunk_frags((c + 2) % NCHUNK);
            if constexpr (HALFABLE) {
                if (!half_mode) stage_chunk<true, WAVES>(gnext, lds_base, slot_stage, nf, wave, lane);     // (half tiles: the helpers')
            } else stage_chunk<true, WAVES>(gnext, lds_base, slot_stage, nf, wave, lane);
            gnext += nf * FRAG_BYTES;
        }
        static_for<NBITS>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (bits_issue_tile(k) == T) {
                if constexpr (bits_wrapped(k)) { if (more) issue_bits<k>(gate_off_next); }
                else issue_bits<k>(gate_off);
            }
        });
    }
    __device__ __forceinline__ void rotate() {
        unsigned t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_stage; slot_stage = t;
        ++c;
    }
    // half tiles: wavefronts 4-7 — the chunk sequence of the workgroup's ONE point tile in step with the workers' barriers
    __device__ __forceinline__ void stream_chunks() {
        dma_wait();
        __syncthreads();                                   // (the workers' `first` barrier)
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_base;
        for (int cc = 0; cc < NCHUNK; ++cc) {
            dma_wait();
            __syncthreads();                               // (advance() of the chunk's first tile)
            if (cc + 2 < NCHUNK) {
                const int nf = chunk_frags(cc + 2);
                for (int p = wave - HALF_WAVES; p < nf; p += WAVES - HALF_WAVES)
                    dma16(gnext + p * FRAG_BYTES + lane * 16, base + slot_stage + p * FRAG_BYTES);
                gnext += nf * FRAG_BYTES;
            }
            const unsigned t = slot_cur; slot_cur = slot_nxt; slot_nxt = slot_stage; slot_stage = t;
        }
        dma_wait();
    }

    struct NoEpi {
        template <int Q> __device__ __forceinline__ void part() const {}
    };
    // mask + conversion into fragments [TB, TB+FPT) of the next stage's input, and the store into dact
    template <int YF, int TB, int TG>
    struct MaskEpi {
        const f32x16 (&a)[NT];
        Frag (&Y)[NT][YF];
        const unsigned (&m)[NT];
        BlockWalk& db;                   // the block of dact this tile's result goes to, the lane's offset inside a block
        const unsigned (&dof)[NT];
        int half;
        template <int Q> __device__ __forceinline__ void part() const {
            parts<Q>();
            // db walks the piece arrays (mlp_core.h: two per store instruction, UPB steps per block) in the static order of the
            // schedule: on inside the block after a store, and behind the tile's last one to the block of the next tile of the chain
            constexpr int UPB = C::IS_BF16 ? 2 : 4;
            if constexpr (Q == 3) {
                if constexpr (TG + 1 < BWD_TILES) db.template step<(bcol(TG + 1) / 32 - bcol(TG) / 32) * UPB - (UPB - 1)>();
            } else if constexpr (!C::IS_BF16 || (Q & 1)) {
                db.template step<1>();
            }
        }
        template <int Q> __device__ __forceinline__ void parts() const {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (C::IS_BF16) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    typedef short s16x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    unsigned pk[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x2 v2 = {a[n][4 * Q + 2 * i], a[n][4 * Q + 2 * i + 1]};
                        pk[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
                        if (bmasked(TG)) {
                            // sign flags saved by the forward, spread by tile(): pair k = 2Q + i has its two flags at bits
                            // 7 - k and 23 - k; << (8 + k) puts them on top of the 16-bit halves, >> 15 (arithmetic, per
                            // half) makes them the masks
                            const s16x2 top = __builtin_bit_cast(s16x2, m[n] << (8 + 2 * Q + i));
                            pk[i] &= __builtin_bit_cast(unsigned, top >> 15);
                        }
                    }
                    Frag& dst = Y[n][TB + (4 * Q) / EPF];
                    u32x4 d4 = __builtin_bit_cast(u32x4, dst);
                    d4[((4 * Q) % EPF) / 2] = pk[0];
                    d4[((4 * Q) % EPF) / 2 + 1] = pk[1];
                    dst = __builtin_bit_cast(Frag, d4);
                    if ((4 * Q + 4) % EPF == 0) pin(dst);
                    // the two half-waves hold alternate 8-byte halves of a 16-byte piece of the row; one v_permlane32_swap per dword
                    // hands the lower half-wave both halves of the even quarter's piece and the upper one both of the odd
                    // quarter's: 16-byte stores, a whole piece per lane, 512 contiguous bytes per half-wave (mlp_core.h)
                    if constexpr ((Q & 1) && bstored(TG, ENC_ONLY)) {
                        const u32x4 prev = __builtin_bit_cast(u32x4, Y[n][TB + (4 * (Q - 1)) / EPF]);
                        constexpr int pd = ((4 * (Q - 1)) % EPF) / 2;
                        const auto s0 = __builtin_amdgcn_permlane32_swap(prev[pd], pk[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(prev[pd + 1], pk[1], false, false);
                        act_store(reinterpret_cast<g_uint4*>(db.p + dof[n]), u32x4n{s0[0], s1[0], s0[1], s1[1]});    // piece 2 (Q >> 1) + half
                    }
                } else {
                    f32x4 keep;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = a[n][4 * Q + i];
                        if (bmasked(TG)) v = ((m[n] >> act_sign_bit(Q, i)) & 1u) ? v : 0.0f;
                        put(Y[n][TB + (4 * Q + i) / EPF], (4 * Q + i) % EPF, v);
                        keep[i] = v;
                    }
                    if ((4 * Q + 4) % EPF == 0) pin(Y[n][TB + (4 * Q) / EPF]);
                    if constexpr (bstored(TG, ENC_ONLY)) act_store(reinterpret_cast<g_f32x4*>(db.p + dof[n]), keep);   // piece 2 Q + half
                }
            }
        }
    };

    // One out-tile: acc[PAR] = cinit + W^T_tile . X (NF fragments).  Same software pipeline as the forward tile().
    template <int T, int NF, int XF, class Pending>
    __device__ __forceinline__ void tile(const Frag (&X)[NT][XF], const Pending& pending, const float (&dsig)[NT]) {
        static_assert(NF % 4 == 0 && NF == btile_frags<C>(T), "tile schedule mismatch");
        constexpr int NG = NF / 4;
        constexpr int PAR = T & 1;
        constexpr int POS = (T - FIRST) % TPC;
        constexpr int OFF = (POS == 0) ? 0 : btile_frags<C>(T - 1);
        constexpr bool END = (T == LAST);
        if constexpr (POS == 0) advance<T>();
        const Frag* cur = reinterpret_cast<const Frag*>(lds_base + slot_cur) + OFF * 64 + lane;
        const Frag* nxt = (POS + 1 < TPC && !END) ? cur + NF * 64 : reinterpret_cast<const Frag*>(lds_base + slot_nxt) + lane;
        // the saved activations that gate THIS tile's result (used by its epilogue, one tile later)
        if constexpr (bmasked(T)) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                constexpr int w = bcol(T) / 32, j = w < 72 ? w % 8 : w - 72;
                if constexpr (j == 0) read_bits<bits_stage_of_w(w)>(n);
                mk[PAR][n] = block_mask<j>(n);
            }
        }
        // C operand: zero, except the rank-1 sigma term on the final^T tiles
        f32x16 cinit[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if constexpr (T >= 12 && T < 20) {
                const f32x4* tb = reinterpret_cast<const f32x4*>(lds_base + ((T - 12) * 32 + half * 16) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = tb[q];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cinit[n][4 * q + i] = w4[i] * dsig[n];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) cinit[n][i] = 0.0f;
            }
        }
        Frag wa[4], wb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wa[q] = w0[q];
        static_for<NG>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            Frag (&use)[4] = (j & 1) ? wb : wa;
            Frag (&ld)[4] = (j & 1) ? wa : wb;
#pragma unroll
            for (int q = 0; q < 4; ++q) ld[q] = (j + 1 < NG) ? cur[((j + 1) * 4 + q) * 64] : nxt[q * 64];
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int f = 4 * j + q;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (f == 0) acc[PAR][n] = mma_c(use[q], X[n][f], cinit[n]);
                    else        mma(use[q], X[n][f], acc[PAR][n]);
                }
                if constexpr (NG >= 3) {
                    if (j < 2 && (q & 1)) {
                        pending.template part<2 * (j < 2 ? j : 0) + (q >> 1)>();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (j == 0) {
                    pending.template part<q>();
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) w0[q] = (NG & 1) ? wb[q] : wa[q];
        if constexpr (POS + 1 == TPC || END) rotate();
    }

    template <int T0, int NTILES, int NF, int XF, int YF, class Pending>
    __device__ __forceinline__ void layer(const Frag (&X)[NT][XF], Frag (&Y)[NT][YF], const Pending& first,
                                          const float (&dsig)[NT]) {
        static_for<NTILES>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int PAR = (T0 + t) & 1;
            if constexpr (t == 0) {
                tile<T0 + t, NF, XF>(X, first, dsig);
            } else {
                tile<T0 + t, NF, XF>(X, MaskEpi<YF, (t - 1) * FPT, T0 + t - 1>{acc[PAR ^ 1], Y, mk[PAR ^ 1], dact_blk, dact_off, half}, dsig);
            }
        });
    }
    template <int T0, int NTILES, int YF>
    __device__ __forceinline__ auto last_of(Frag (&Y)[NT][YF]) {
        constexpr int T = T0 + NTILES - 1;
        return MaskEpi<YF, (NTILES - 1) * FPT, T>{acc[T & 1], Y, mk[T & 1], dact_blk, dact_off, half};
    }

    __device__ __forceinline__ void run(const char* __restrict__ pack, const float4* __restrict__ g, const ActT* __restrict__ act,
                                        ActT* __restrict__ dact, int64_t n_pts, char* lds, int tangent = 0,
                                        const float* __restrict__ dfeat = nullptr, const int32_t* __restrict__ count = nullptr) {
        R = n_pts;                                         // the buffers' row count (block stride); the rows worked on may be fewer:
        if (count) {                                       // the padded length of a compacted list, known on the device only
            const int64_t cnt = *count;
            n_pts = cnt < n_pts ? cnt : n_pts;
        }
        half_mode = false;
        if constexpr (HALFABLE) half_mode = !tangent && (n_pts + HALF_WAVES * 32 - 1) / (HALF_WAVES * 32) <= (int64_t)gridDim.x;
        const int wpt = HALFABLE ? (half_mode ? HALF_WAVES : WAVES) : WAVES;      // wavefronts that hold rows of a point tile
        const int64_t n_tiles = (n_pts + wpt * NT * 32 - 1) / (wpt * NT * 32);
        if ((int64_t)blockIdx.x >= n_tiles) return;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        half = lane >> 5;
        lds_base = lds;
        slot_cur = BWD_TABLE_BYTES;
        slot_nxt = slot_cur + SLOT;
        slot_stage = slot_nxt + SLOT;
        gbase = pack + BWD_TABLE_BYTES + bfrag_offset<C>(FIRST) * FRAG_BYTES;
        for (int i = threadIdx.x; i < BWD_TABLE_BYTES / 16; i += THREADS)
            reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(pack)[i];
        lds_bits = lds + BWD_TABLE_BYTES + 3 * SLOT + wave * (2 * NT * 1024);
        gnext = gbase;
        stage_chunk<true, WAVES>(gnext, lds_base, slot_cur, chunk_frags(0), wave, lane);
        gnext += chunk_frags(0) * FRAG_BYTES;
        stage_chunk<true, WAVES>(gnext, lds_base, slot_nxt, chunk_frags(1), wave, lane);
        gnext += chunk_frags(1) * FRAG_BYTES;
        if constexpr (HALFABLE) {
            if (half_mode && wave >= HALF_WAVES) { stream_chunks(); return; }
        }
        bool first = true;
        // rows of point tile `t`: a lane past the end aliases the last row (it recomputes and rewrites that row's values:
        // every wave issues the same stores whatever n is — advance() counts them)
        auto rows_of = [&](int64_t t, int n) {
            const int64_t idx = (t * wpt + wave) * (NT * 32) + n * 32 + (lane & 31);
            return idx < n_pts ? idx : n_pts - 1;
        };
        // tangent mode: the ReLU gates of a quad's four columns are the primal column's (row 4p)
        auto gate_row = [&](int64_t cl) { return (unsigned)(tangent ? (cl & ~(int64_t)3) : cl) * 32u + 16u * half; };
        act_base = reinterpret_cast<const char*>(act);
        dact_base = reinterpret_cast<char*>(dact);
        float4 g_next[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int64_t cl = rows_of(blockIdx.x, n);
            g_next[n] = g[cl];
            gate_off_next[n] = gate_row(cl);
        }
        // the first point tile's own opening bit stages (later ones are gathered from inside the tile before)
        static_for<NBITS>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (bits_wrapped(k)) issue_bits<k>(gate_off_next);
        });

        for (int64_t pt = blockIdx.x; pt < n_tiles; pt += gridDim.x) {
            more = pt + gridDim.x < n_tiles;
            c = 0;
            float dsig[NT];
            float4 gin[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                gin[n] = g_next[n];
                dsig[n] = gin[n].w;
                gate_off[n] = gate_off_next[n];
                dact_off[n] = (unsigned)rows_of(pt, n) * 16u + (unsigned)half * (unsigned)(R * 16);
                if (more) {                                   // the next tile's upstream gradient arrives under this tile's MFMAs
                    const int64_t cl = rows_of(pt + gridDim.x, n);
                    g_next[n] = g[cl];
                    gate_off_next[n] = gate_row(cl);
                }
            }
            if (first) {
                dma_wait();
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) w0[q] = (reinterpret_cast<const Frag*>(lds_base + slot_cur) + lane)[q * 64];
                first = false;
            }
            // the first epilogue of the chain: tile 0 (rgb^T), 4 (FEATURE: d feature enters through dir^T's), 12 (sigma only)
            dact_blk.stride = 2 * 16 * R;                     // two piece arrays per store instruction
            dact_blk.reset(dact_base, bcol(FEATURE ? 4 : SIGMA_ONLY ? 12 : 0) / 32 * (C::IS_BF16 ? 2 : 4));
            Frag A[NT][HF], B[NT][HF];
            if constexpr (FEATURE) {
                // dF comes from outside (fp32 [n][256]): through the epilogue of the stage that would have produced it —
                // conversion into the fragments of A, store into dact (the weight-gradient GEMM of xyz_encoding_final reads it
                // there) — and zeros into the dact columns of the colour head this pass does not run
                static_for<8>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float* src = dfeat + rows_of(pt, n) * 256 + 32 * j + 4 * half;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 8 * q);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[0][n][4 * q + i] = v[i];
                        }
                    }
                    MaskEpi<HF, j * FPT, 4 + j> epi{acc[0], A, mk[0], dact_blk, dact_off, half};
                    epi.template part<0>(); epi.template part<1>(); epi.template part<2>(); epi.template part<3>();
                });
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                // features 8 q + 4 half .. + 3 of the row: bf16 half a piece (piece q), fp32 piece 2 q + half
                                const int64_t row16 = (int64_t)rows_of(pt, n) * 16;
                                if constexpr (C::IS_BF16)
                                    *reinterpret_cast<uint2*>(dact_base + act_piece_off(R, ESZ, bcol(t) / 32, q) + row16 + 8 * half) = make_uint2(0u, 0u);
                                else
                                    *reinterpret_cast<f32x4*>(dact_base + act_piece_off(R, ESZ, bcol(t) / 32, 2 * q + half) + row16) = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                layer<12, 8, HF, HF, HF>(A, B, NoEpi{}, dsig);                              // final^T: dF -> dh8' (B)
                layer<20, 8, HF, HF, HF>(B, A, last_of<12, 8, HF>(B), dsig);                // W8^T   : dh8' -> dh7' (A)
            } else if constexpr (!SIGMA_ONLY) {
                // stage R input: d_rgb' (3 values) sits in slots 0..2 of fragment 0 of the lower half-wave
                Frag X0[NT][4];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int e = 0; e < EPF; ++e) put(X0[n][f], e, 0.0f);
                    put(X0[n][0], 0, half ? 0.0f : gin[n].x);
                    put(X0[n][0], 1, half ? 0.0f : gin[n].y);
                    put(X0[n][0], 2, half ? 0.0f : gin[n].z);
                }
                Frag (&Gd)[NT][HF] = B;                                                    // dG lives in B[.][0..DF)
                layer<0, 4, 4, 4, HF>(X0, Gd, NoEpi{}, dsig);                               // rgb^T  : d_rgb' -> dG
                layer<4, 8, DF, HF, HF>(B, A, last_of<0, 4, HF>(B), dsig);                  // dir^T  : dG -> dF (A)
                layer<12, 8, HF, HF, HF>(A, B, last_of<4, 8, HF>(A), dsig);                 // final^T: dF -> dh8' (B)
                layer<20, 8, HF, HF, HF>(B, A, last_of<12, 8, HF>(B), dsig);                // W8^T   : dh8' -> dh7' (A)
            } else {
                // dh8' = w_sigma d_sigma . 1[h8 > 0], straight into fragments (B) and into dact
                static_for<8>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const f32x4* tb = reinterpret_cast<const f32x4*>(lds_base + (j * 32 + half * 16) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 w4 = tb[q];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[0][n][4 * q + i] = w4[i] * dsig[n];
                        }
                        if constexpr (j == 0) read_bits<0>(n);
                        mk[0][n] = block_mask<j>(n);
                    }
                    MaskEpi<HF, j * FPT, 12 + j> epi{acc[0], B, mk[0], dact_blk, dact_off, half};
                    epi.template part<0>(); epi.template part<1>(); epi.template part<2>(); epi.template part<3>();
                });
                layer<20, 8, HF, HF, HF>(B, A, NoEpi{}, dsig);                              // W8^T   : dh8' -> dh7' (A)
            }
            layer<28, 8, HF, HF, HF>(A, B, last_of<20, 8, HF>(A), dsig);                    // W7^T
            layer<36, 8, HF, HF, HF>(B, A, last_of<28, 8, HF>(B), dsig);                    // W6^T
            layer<44, 8, HF, HF, HF>(A, B, last_of<36, 8, HF>(A), dsig);                    // W5^T (hidden part)
            layer<52, 8, HF, HF, HF>(B, A, last_of<44, 8, HF>(B), dsig);                    // W4^T
            layer<60, 8, HF, HF, HF>(A, B, last_of<52, 8, HF>(A), dsig);                    // W3^T
            layer<68, 8, HF, HF, HF>(B, A, last_of<60, 8, HF>(B), dsig);                    // W2^T   : -> dh1' (stored)
            {
                auto fin = last_of<68, 8, HF>(A);
                fin.template part<0>(); fin.template part<1>(); fin.template part<2>(); fin.template part<3>();
            }
        }
    }
};

template <int MODE, int START, bool ENC_ONLY = false>
__global__ __launch_bounds__(Cfg<MODE>::WAVES * 64, Cfg<MODE>::WAVES / 4) void mlp_bwd_kernel(
    const char* __restrict__ pack, const float4* __restrict__ g, const void* __restrict__ act, void* __restrict__ dact,
    int64_t n_pts, int tangent, const float* __restrict__ dfeat, const int32_t* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using M = MlpBwd<MODE, START, ENC_ONLY>;
    M m;
    m.run(pack, g, reinterpret_cast<const typename M::ActT*>(act), reinterpret_cast<typename M::ActT*>(dact), n_pts, lds, tangent,
          dfeat, count);
}

template <int MODE, int START, bool ENC_ONLY = false>
int launch_mlp_bwd(const void* pack, const float* g, const void* act, void* dact, int64_t n, hipStream_t st, int tangent = 0,
                   const float* dfeat = nullptr, const int32_t* count = nullptr) {
    using C = Cfg<MODE>;
    const int lds = BWD_TABLE_BYTES + 3 * MlpBwd<MODE, START, ENC_ONLY>::SLOT + C::WAVES * 2 * C::NT * 1024;    // + the gathered sign bits
    auto kern = mlp_bwd_kernel<MODE, START, ENC_ONLY>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail((int)e, "anr_mlp_backward: hipFuncSetAttribute: %s", hipGetErrorString(e));
    // (the grid holds HALF tiles of 128 rows when the buffer itself is that small: MlpBwd::HALFABLE)
    const int pts_per_wg = MlpBwd<MODE, START, ENC_ONLY>::HALFABLE ? 128 : C::WAVES * C::NT * 32;
    const int64_t n_tiles = (n + pts_per_wg - 1) / pts_per_wg;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    dim3 grid((unsigned)(n_tiles < cus ? n_tiles : cus));
    hipLaunchKernelGGL(kern, grid, dim3(C::WAVES * 64), lds, st, reinterpret_cast<const char*>(pack),
                       reinterpret_cast<const float4*>(g), act, dact, n, tangent, dfeat, count);
    return check_launch("anr_mlp_backward");
}

// ---------------------------------------------------------------------------------------------
// W^T in fragment order.  Fragment (tile t, k-fragment kf), lane (i = lane & 31, h = lane >> 5), element e:
//   row  = input feature 32 t' + i of the forward layer (t' = tile inside its stage)
//   k    = output feature of the forward layer in the slot order of the gradient fragments (mlp_core.h header)
struct BwdStage { const float* W; int ld, in_off, in_dim, out_dim, tile0, nf; };
struct BwdPlan { BwdStage s[10]; const float* w_sigma; const float* w1; const float* w5; };


// blockIdx.y = which of (up to) two networks
template <int MODE>
__global__ void mlp_bwd_pack_kernel(BwdPlan plan_a, char* __restrict__ pack_a, BwdPlan plan_b, char* __restrict__ pack_b) {
    using C = Cfg<MODE>;
    const BwdPlan& plan = blockIdx.y ? plan_b : plan_a;
    char* __restrict__ pack = blockIdx.y ? pack_b : pack_a;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_pieces = (int64_t)btotal_frags<C>() * 64;
    if (gid < n_pieces) {
        const int frag = (int)(gid >> 6), lane = (int)(gid & 63);
        const int i = lane & 31, h = lane >> 5;
        int t = 0, base = 0;
        while (base + btile_frags<C>(t) <= frag) { base += btile_frags<C>(t); ++t; }
        const int kf = frag - base;
        int si = 0;
        while (si + 1 < 10 && plan.s[si + 1].tile0 <= t) ++si;
        const BwdStage& st = plan.s[si];
        const int row = 32 * (t - st.tile0) + i;
        float v[8];
#pragma unroll
        for (int e = 0; e < C::EPF; ++e) {
            const int feat = C::IS_BF16 ? 16 * kf + 8 * (e >> 2) + 4 * h + (e & 3) : 8 * kf + 4 * h + e;
            v[e] = (row < st.in_dim && feat < st.out_dim) ? st.W[(int64_t)feat * st.ld + st.in_off + row] : 0.0f;
        }
        char* dst = pack + BWD_TABLE_BYTES + (int64_t)frag * FRAG_BYTES + lane * 16;
        if constexpr (C::IS_BF16) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else if (gid - n_pieces < BWD_TABLE_BYTES / 4) {
        const int64_t bi = gid - n_pieces;                 // w_sigma table: [tile j][half][reg]
        const int j = (int)(bi / 32), h = (int)((bi % 32) / 16), reg = (int)(bi % 16);
        reinterpret_cast<float*>(pack)[bi] = plan.w_sigma[32 * j + 8 * (reg >> 2) + 4 * h + (reg & 3)];
    } else {
        const int64_t pe = gid - n_pieces - BWD_TABLE_BYTES / 4;      // one element of the encoding panels
        if (pe >= DENC_PANEL_ELEMS) return;
        constexpr int EPL = C::IS_BF16 ? 8 : 1, KF = C::IS_BF16 ? 16 : 128;
        const int e = (int)(pe % EPL), i = (int)(pe / EPL);
        const int l = i & 63, nt = (i >> 6) & 1, kf = (i >> 7) % KF, layer = (i >> 7) / KF;
        const int j = (l & 31) + 32 * nt, h = l >> 5;
        const int k = C::IS_BF16 ? 16 * kf + 8 * (e >> 2) + 4 * h + (e & 3) : 2 * kf + h;
        const float v = j < 63 ? (layer ? plan.w5[(int64_t)k * 319 + j] : plan.w1[(int64_t)k * 63 + j]) : 0.0f;
        char* dst = pack + denc_panel_off<C>();
        if constexpr (C::IS_BF16) reinterpret_cast<__bf16*>(dst)[pe] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[pe] = v;
    }
}

}  // namespace anr

using namespace anr;

extern "C" int64_t anr_mlp_bwd_pack_bytes(int mode) {
    switch (mode & 0xff) {
        case ANR_MLP_F32:  return denc_panel_off<Cfg<ANR_MLP_F32>>() + denc_panel_bytes<Cfg<ANR_MLP_F32>>();
        case ANR_MLP_BF16: return denc_panel_off<Cfg<ANR_MLP_BF16>>() + denc_panel_bytes<Cfg<ANR_MLP_BF16>>();
        default: return ANR_E_BADARG;
    }
}

static int bwd_plan(const anr_mlp_params* p, const void* pack_out, BwdPlan& plan) {
    ANR_REQUIRE(p && pack_out, ANR_E_BADARG, "anr_mlp_bwd_pack: null pointer");
    for (int l = 1; l < 8; ++l) ANR_REQUIRE(p->w_trunk[l], ANR_E_BADARG, "anr_mlp_bwd_pack: null trunk tensor %d", l);
    ANR_REQUIRE(p->w_sigma && p->w_final && p->w_dir && p->w_rgb, ANR_E_BADARG, "anr_mlp_bwd_pack: null head tensor");
    ANR_REQUIRE(((uintptr_t)pack_out & 15) == 0, ANR_E_ALIGN, "anr_mlp_bwd_pack: pack_out must be 16-B aligned");
    plan = BwdPlan{};
    plan.s[0] = BwdStage{p->w_rgb, 128, 0, 128, 3, 0, 4};                 // rgb.0.weight [3,128]
    plan.s[1] = BwdStage{p->w_dir, 256, 0, 256, 128, 4, 0};               // dir_encoding.0.weight [128,256]
    plan.s[2] = BwdStage{p->w_final, 256, 0, 256, 256, 12, 0};            // xyz_encoding_final.weight [256,256]
    for (int s = 0; s < 7; ++s) {                                         // xyz_encoding_{8..2}: [256, 256] ([256,319] for 5)
        const int l = 7 - s;                                              // index into w_trunk (layer l+1)
        plan.s[3 + s] = BwdStage{p->w_trunk[l], l == 4 ? 319 : 256, l == 4 ? 63 : 0, 256, 256, 20 + 8 * s, 0};
    }
    plan.w_sigma = p->w_sigma;
    ANR_REQUIRE(p->w_trunk[0], ANR_E_BADARG, "anr_mlp_bwd_pack: null trunk tensor 0");
    plan.w1 = p->w_trunk[0];
    plan.w5 = p->w_trunk[4];
    return 0;
}

static int bwd_pack_networks(const anr_mlp_params* p, const anr_mlp_params* p_b, int mode, void* pack_out, void* pack_b, void* stream) {
    BwdPlan plan, plan_b;
    if (int rc = bwd_plan(p, pack_out, plan)) return rc;
    plan_b = plan;
    if (p_b != nullptr)
        if (int rc = bwd_plan(p_b, pack_b, plan_b)) return rc;
    const unsigned ny = p_b != nullptr ? 2u : 1u;
    hipStream_t st = (hipStream_t)stream;
    if ((mode & 0xff) == ANR_MLP_F32) {
        int64_t n = (int64_t)btotal_frags<Cfg<ANR_MLP_F32>>() * 64 + BWD_TABLE_BYTES / 4 + DENC_PANEL_ELEMS;
        hipLaunchKernelGGL(mlp_bwd_pack_kernel<ANR_MLP_F32>, dim3((unsigned)((n + 255) / 256), ny), dim3(256), 0, st, plan, (char*)pack_out,
                           plan_b, (char*)pack_b);
    } else if ((mode & 0xff) == ANR_MLP_BF16) {
        int64_t n = (int64_t)btotal_frags<Cfg<ANR_MLP_BF16>>() * 64 + BWD_TABLE_BYTES / 4 + DENC_PANEL_ELEMS;
        hipLaunchKernelGGL(mlp_bwd_pack_kernel<ANR_MLP_BF16>, dim3((unsigned)((n + 255) / 256), ny), dim3(256), 0, st, plan, (char*)pack_out,
                           plan_b, (char*)pack_b);
    } else {
        return fail(ANR_E_BADARG, "anr_mlp_bwd_pack: unknown mode %d", mode);
    }
    return check_launch("anr_mlp_bwd_pack");
}

extern "C" int anr_mlp_bwd_pack(const anr_mlp_params* p, int mode, void* pack_out, void* stream) {
    return bwd_pack_networks(p, nullptr, mode, pack_out, nullptr, stream);
}

extern "C" int anr_mlp_bwd_pack_pair(const anr_mlp_params* p_a, const anr_mlp_params* p_b, int mode, void* pack_a_out, void* pack_b_out,
                                     void* stream) {
    ANR_REQUIRE(p_b && pack_b_out, ANR_E_BADARG, "anr_mlp_bwd_pack_pair: null pointer");
    return bwd_pack_networks(p_a, p_b, mode, pack_a_out, pack_b_out, stream);
}

extern "C" int anr_mlp_backward(const void* bwd_pack, int mode, const float* g, const void* act, void* dact, int64_t n,
                                void* stream) {
    return anr_mlp_backward_counted(bwd_pack, mode, g, act, dact, n, nullptr, stream);
}

extern "C" int anr_mlp_backward_counted(const void* bwd_pack, int mode, const float* g, const void* act, void* dact, int64_t n,
                                        const int32_t* count, void* stream) {
    ANR_REQUIRE(bwd_pack && g && act && dact, ANR_E_BADARG, "anr_mlp_backward: null pointer");
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_backward: n=%lld", (long long)n);
    ANR_REQUIRE(n <= (((mode & 0xff) == ANR_MLP_F32) ? (int64_t)1 << 25 : (int64_t)1 << 26), ANR_E_BADARG,
                "anr_mlp_backward: n=%lld rows per call exceed the blocked layout (2^25 fp32 / 2^26 bf16): chunk the call", (long long)n);
    ANR_REQUIRE((((uintptr_t)bwd_pack | (uintptr_t)g | (uintptr_t)act | (uintptr_t)dact) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_backward: bwd_pack/g/act/dact must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool so = (mode & ANR_MLP_FLAG_SIGMA_ONLY) != 0;
    const int tan = (mode & ANR_MLP_FLAG_TANGENT) ? 1 : 0;
    ANR_REQUIRE(!tan || (so && n % 4 == 0), ANR_E_BADARG, "anr_mlp_backward: tangent mode = sigma only, points in quads");
    if (mode & ANR_MLP_FLAG_ENC_ONLY) {
        ANR_REQUIRE(!so && !tan, ANR_E_BADARG, "anr_mlp_backward: ANR_MLP_FLAG_ENC_ONLY serves the full network's render passes");
        switch (mode & 0xff) {
            case ANR_MLP_F32:  return launch_mlp_bwd<ANR_MLP_F32, 0, true>(bwd_pack, g, act, dact, n, st, 0, nullptr, count);
            case ANR_MLP_BF16: return launch_mlp_bwd<ANR_MLP_BF16_W8, 0, true>(bwd_pack, g, act, dact, n, st, 0, nullptr, count);
            default: return fail(ANR_E_BADARG, "anr_mlp_backward: unknown mode %d", mode);
        }
    }
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return so ? launch_mlp_bwd<ANR_MLP_F32, 20>(bwd_pack, g, act, dact, n, st, tan, nullptr, count)
                      : launch_mlp_bwd<ANR_MLP_F32, 0>(bwd_pack, g, act, dact, n, st, 0, nullptr, count);
        case ANR_MLP_BF16:
            return so ? launch_mlp_bwd<ANR_MLP_BF16_W8, 20>(bwd_pack, g, act, dact, n, st, tan, nullptr, count)
                      : launch_mlp_bwd<ANR_MLP_BF16_W8, 0>(bwd_pack, g, act, dact, n, st, 0, nullptr, count);
        default: return fail(ANR_E_BADARG, "anr_mlp_backward: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_backward_feature(const void* bwd_pack, int mode, const float* g, const float* d_feature, const void* act,
                                        void* dact, int64_t n, void* stream) {
    ANR_REQUIRE(bwd_pack && g && d_feature && act && dact, ANR_E_BADARG, "anr_mlp_backward_feature: null pointer");
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_backward_feature: n=%lld", (long long)n);
    ANR_REQUIRE((((uintptr_t)bwd_pack | (uintptr_t)g | (uintptr_t)d_feature | (uintptr_t)act | (uintptr_t)dact) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_backward_feature: bwd_pack/g/d_feature/act/dact must be 16-B aligned");
    ANR_REQUIRE(!(mode & (ANR_MLP_FLAG_SIGMA_ONLY | ANR_MLP_FLAG_TANGENT)), ANR_E_BADARG, "anr_mlp_backward_feature: no sigma-only / tangent mode");
    hipStream_t st = (hipStream_t)stream;
    switch (mode & 0xff) {
        case ANR_MLP_F32:  return launch_mlp_bwd<ANR_MLP_F32, 12>(bwd_pack, g, act, dact, n, st, 0, d_feature);
        case ANR_MLP_BF16: return launch_mlp_bwd<ANR_MLP_BF16_W8, 12>(bwd_pack, g, act, dact, n, st, 0, d_feature);
        default: return fail(ANR_E_BADARG, "anr_mlp_backward_feature: unknown mode %d", mode);
    }
}
