// C ABI of the fused MLP (anr_mlp_pack / anr_mlp_forward / anr_mlp_forward_save) and the weight-pack kernel.
// The kernel template lives in mlp_core.h; its instantiations are compiled in mlp_inst_*.hip (parallel builds).
#include "mlp_core.h"

namespace anr {

// instantiated in mlp_inst_*.hip
#define ANR_MLP_EXTERN(M, D, S, V) \
    extern template int launch_mlp<M, D, S, V>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
ANR_MLP_EXTERN(ANR_MLP_F32, true, false, false)  ANR_MLP_EXTERN(ANR_MLP_F32, false, false, false)
ANR_MLP_EXTERN(ANR_MLP_F32, true, true, false)   ANR_MLP_EXTERN(ANR_MLP_F32, true, false, true)
ANR_MLP_EXTERN(ANR_MLP_F32, true, true, true)
ANR_MLP_EXTERN(ANR_MLP_BF16_W8, true, false, false)  ANR_MLP_EXTERN(ANR_MLP_BF16_W8, false, false, false)
ANR_MLP_EXTERN(ANR_MLP_BF16_W8, true, true, false)   ANR_MLP_EXTERN(ANR_MLP_BF16_W8, true, false, true)
ANR_MLP_EXTERN(ANR_MLP_BF16_W8, true, true, true)
ANR_MLP_EXTERN(ANR_MLP_BF16, true, false, false)  ANR_MLP_EXTERN(ANR_MLP_BF16, false, false, false)
#undef ANR_MLP_EXTERN
extern template int launch_mlp<ANR_MLP_F32, true, true, true, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
extern template int launch_mlp<ANR_MLP_BF16_W8, true, true, true, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
#define ANR_PRE(M, S) extern template int launch_mlp<M, true, S, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
ANR_PRE(ANR_MLP_F32, false) ANR_PRE(ANR_MLP_F32, true) ANR_PRE(ANR_MLP_BF16_W8, false) ANR_PRE(ANR_MLP_BF16_W8, true)
#undef ANR_PRE
extern template int launch_mlp<ANR_MLP_F32, true, false, true, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
extern template int launch_mlp<ANR_MLP_BF16_W8, true, false, true, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);

extern template int launch_mlp<ANR_MLP_F32, true, false, true, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);
extern template int launch_mlp<ANR_MLP_BF16_W8, true, false, true, false, false, false, true>(const void*, const float*, int64_t, float*, hipStream_t, float*, const int32_t*, const int32_t*, const float*, int, int);

// ---------------------------------------------------------------------------------------------
// weight packing: one thread per (frag, lane) 16-byte piece; tail threads fill the bias table
struct PackStage {
    const float* W; const float* B;
    int out_dim, in_dim, enc_cols, n_tiles, nf_enc, nf_hid, frag0, tile0;
    int enc_col0, hid_col0;     // first column of the encoding channels / of the hidden features in W's rows
};
struct PackPlan { PackStage s[12]; int n_stages; int total_frags; };

__device__ __forceinline__ int enc_channel(int j, int h) {       // -1 = pad
    if (j < 30) return 3 + 6 * (j / 3) + 3 * h + (j % 3);
    if (j == 30) return h ? 2 : 0;
    return h ? -1 : 1;
}

// blockIdx.y = which of (up to) two networks: a training step packs the coarse and the fine network in ONE launch
template <int MODE>
__global__ void mlp_pack_kernel(PackPlan plan_a, char* __restrict__ pack_a, PackPlan plan_b, char* __restrict__ pack_b) {
    using C = Cfg<MODE>;
    const PackPlan& plan = blockIdx.y ? plan_b : plan_a;
    char* __restrict__ pack = blockIdx.y ? pack_b : pack_a;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_pieces = (int64_t)plan.total_frags * 64;
    if (gid < n_pieces) {
        const int frag = (int)(gid >> 6), lane = (int)(gid & 63);
        const int i = lane & 31, h = lane >> 5;
        int si = 0;
        while (si + 1 < plan.n_stages && plan.s[si + 1].frag0 <= frag) ++si;
        const PackStage& st = plan.s[si];
        const int per_tile = st.nf_enc + st.nf_hid;
        const int t = (frag - st.frag0) / per_tile, kf = (frag - st.frag0) % per_tile;
        const int row = 32 * t + i;
        float v[8];
#pragma unroll
        for (int e = 0; e < C::EPF; ++e) {
            int col;
            if (kf < st.nf_enc) {
                int ch = enc_channel(C::EPF * kf + e, h);
                col = (ch >= 0 && ch < st.enc_cols) ? st.enc_col0 + ch : -1;
            } else {
                int f = kf - st.nf_enc;
                int feat = C::IS_BF16 ? 16 * f + 8 * (e >> 2) + 4 * h + (e & 3) : 8 * f + 4 * h + e;
                col = st.hid_col0 + feat;
            }
            v[e] = (row < st.out_dim && col >= 0 && col < st.in_dim) ? st.W[(int64_t)row * st.in_dim + col] : 0.0f;
        }
        char* dst = pack + BIAS_BYTES + (int64_t)frag * FRAG_BYTES + lane * 16;
        if constexpr (C::IS_BF16) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else {
        const int64_t bi = gid - n_pieces;                 // float index into the bias table
        if (bi >= BIAS_BYTES / 4) return;
        const int T = (int)(bi / 32), h = (int)((bi % 32) / 16), reg = (int)(bi % 16);
        float val = 0.0f;
        if (T < N_TILES_TOTAL) {
            int si = 0;
            while (si + 1 < plan.n_stages && plan.s[si + 1].tile0 <= T) ++si;
            const PackStage& st = plan.s[si];
            const int row = 32 * (T - st.tile0) + 8 * (reg >> 2) + 4 * h + (reg & 3);
            if (row < st.out_dim) val = st.B[row];
        }
        reinterpret_cast<float*>(pack)[bi] = val;
    }
}

// dir_channels > 0: the view-dependent colour head (models/nerf.py:141-153): dir_encoding.0.weight is [128][256 + dir_channels],
// its last dir_channels columns = the Fourier encoding of the view direction -> the panel fragments of tiles 73..76
template <int MODE>
PackPlan make_plan(const anr_mlp_params* p, int dir_channels = 0) {
    using C = Cfg<MODE>;
    PackPlan plan{};
    int frag = 0, tile = 0, n = 0;
    auto add = [&](const float* W, const float* B, int out_dim, int in_dim, int enc_cols, int n_tiles, int nfe, int nfh,
                   int enc_col0 = 0, int hid_col0 = -1) {
        plan.s[n++] = PackStage{W, B, out_dim, in_dim, enc_cols, n_tiles, nfe, nfh, frag, tile, enc_col0, hid_col0 < 0 ? enc_cols : hid_col0};
        frag += n_tiles * (nfe + nfh);
        tile += n_tiles;
    };
    add(p->w_trunk[0], p->b_trunk[0], 256, 63, 63, 8, C::EF, 0);
    for (int l = 1; l < 8; ++l) {
        if (l == 4) add(p->w_trunk[l], p->b_trunk[l], 256, 319, 63, 8, C::EF, C::HF);
        else        add(p->w_trunk[l], p->b_trunk[l], 256, 256, 0, 8, 0, C::HF);
    }
    add(p->w_sigma, p->b_sigma, 1, 256, 0, 1, 0, C::HF);          // tile 64: sigma row on h8
    add(p->w_final, p->b_final, 256, 256, 0, 8, 0, C::HF);        // tiles 65..72
    if (dir_channels > 0) add(p->w_dir, p->b_dir, 128, 256 + dir_channels, dir_channels, 4, C::EF, C::HF, 256, 0);
    else                  add(p->w_dir, p->b_dir, 128, 256, 0, 4, 0, C::HF);
    add(p->w_rgb, p->b_rgb, 3, 128, 0, 1, 0, C::DF);
    plan.n_stages = n;
    plan.total_frags = frag;
    return plan;
}

}  // namespace anr

using namespace anr;

#define ANR_MLP_FLAG_NO_DMA 0x100      /* debug: stage weights through registers instead of the LDS-DMA engine */
/* ANR_MLP_FLAG_VIEW (0x2000) is public: include/animnerf_hip.h */
#define ANR_MLP_FLAG_W4     0x200      /* bf16 only: 4 waves x 64 points per workgroup instead of 8 waves x 32 */
/* ANR_MLP_FLAG_SIGMA_ONLY (0x400) is public: include/animnerf_hip.h */

extern "C" int64_t anr_mlp_pack_bytes(int mode) {
    const bool view = (mode & ANR_MLP_FLAG_VIEW) != 0;
    switch (mode & 0xff) {
        case ANR_MLP_F32:  return BIAS_BYTES + (int64_t)(view ? total_frags<Cfg<ANR_MLP_F32>, true>() : total_frags<Cfg<ANR_MLP_F32>>()) * FRAG_BYTES;
        case ANR_MLP_BF16: return BIAS_BYTES + (int64_t)(view ? total_frags<Cfg<ANR_MLP_BF16>, true>() : total_frags<Cfg<ANR_MLP_BF16>>()) * FRAG_BYTES;
        default: return ANR_E_BADARG;
    }
}

extern "C" int anr_mlp_pack(const anr_mlp_params* p, int mode, void* pack_out, void* stream) {
    return anr_mlp_pack_view(p, mode, 0, pack_out, stream);
}

static int check_pack_params(const anr_mlp_params* p, const void* pack_out) {
    ANR_REQUIRE(p && pack_out, ANR_E_BADARG, "anr_mlp_pack: null pointer");
    for (int l = 0; l < 8; ++l)
        ANR_REQUIRE(p->w_trunk[l] && p->b_trunk[l], ANR_E_BADARG, "anr_mlp_pack: null trunk tensor %d", l);
    ANR_REQUIRE(p->w_sigma && p->b_sigma && p->w_final && p->b_final && p->w_dir && p->b_dir && p->w_rgb && p->b_rgb,
                ANR_E_BADARG, "anr_mlp_pack: null head tensor");
    ANR_REQUIRE(((uintptr_t)pack_out & 15) == 0, ANR_E_ALIGN, "anr_mlp_pack: pack_out must be 16-B aligned");
    return 0;
}

// p_b / pack_b NULL: one network
static int pack_networks(const anr_mlp_params* p, const anr_mlp_params* p_b, int mode, int dir_channels, void* pack_out, void* pack_b,
                         void* stream) {
    if (int rc = check_pack_params(p, pack_out)) return rc;
    if (p_b != nullptr)
        if (int rc = check_pack_params(p_b, pack_b)) return rc;
    ANR_REQUIRE(dir_channels >= 0 && dir_channels <= 63 && (dir_channels == 0 || dir_channels % 6 == 3), ANR_E_BADARG,
                "anr_mlp_pack_view: dir_channels=%d (0, or 3 + 6 freqs_dir <= 63)", dir_channels);
    hipStream_t st = (hipStream_t)stream;
    const unsigned ny = p_b != nullptr ? 2u : 1u;
    if ((mode & 0xff) == ANR_MLP_F32) {
        PackPlan plan = make_plan<ANR_MLP_F32>(p, dir_channels);
        PackPlan plan_b = p_b != nullptr ? make_plan<ANR_MLP_F32>(p_b, dir_channels) : plan;
        int64_t n = (int64_t)plan.total_frags * 64 + BIAS_BYTES / 4;
        hipLaunchKernelGGL(mlp_pack_kernel<ANR_MLP_F32>, dim3((unsigned)((n + 255) / 256), ny), dim3(256), 0, st, plan, (char*)pack_out, plan_b,
                           (char*)pack_b);
    } else if ((mode & 0xff) == ANR_MLP_BF16) {
        PackPlan plan = make_plan<ANR_MLP_BF16>(p, dir_channels);
        PackPlan plan_b = p_b != nullptr ? make_plan<ANR_MLP_BF16>(p_b, dir_channels) : plan;
        int64_t n = (int64_t)plan.total_frags * 64 + BIAS_BYTES / 4;
        hipLaunchKernelGGL(mlp_pack_kernel<ANR_MLP_BF16>, dim3((unsigned)((n + 255) / 256), ny), dim3(256), 0, st, plan, (char*)pack_out, plan_b,
                           (char*)pack_b);
    } else {
        return fail(ANR_E_BADARG, "anr_mlp_pack: unknown mode %d", mode);
    }
    return check_launch("anr_mlp_pack");
}

extern "C" int anr_mlp_pack_view(const anr_mlp_params* p, int mode, int dir_channels, void* pack_out, void* stream) {
    return pack_networks(p, nullptr, mode, dir_channels, pack_out, nullptr, stream);
}

extern "C" int anr_mlp_pack_pair(const anr_mlp_params* p_a, const anr_mlp_params* p_b, int mode, void* pack_a_out, void* pack_b_out,
                                 void* stream) {
    ANR_REQUIRE(p_b && pack_b_out, ANR_E_BADARG, "anr_mlp_pack_pair: null pointer");
    return pack_networks(p_a, p_b, mode, 0, pack_a_out, pack_b_out, stream);
}

extern "C" int anr_mlp_act_cols(void) { return ACT_PITCH; }

extern "C" int anr_mlp_forward_save(const void* pack, int mode, const float* pts, int64_t n, float* out, void* act_v,
                                    void* stream) {
    return anr_mlp_forward_save_indexed(pack, mode, pts, nullptr, nullptr, n, out, act_v, stream);
}

extern "C" int anr_mlp_forward_save_indexed(const void* pack, int mode, const float* pts, const int32_t* index,
                                            const int32_t* count, int64_t n, float* out, void* act_v, void* stream) {
    float* act = reinterpret_cast<float*>(act_v);
    ANR_REQUIRE(pack && pts && out && act, ANR_E_BADARG, "anr_mlp_forward_save: null pointer");
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_forward_save: n=%lld", (long long)n);
    // (a lane addresses its row inside a 32-feature block with a 32-bit byte offset)
    ANR_REQUIRE(n <= (((mode & 0xff) == ANR_MLP_F32) ? (int64_t)1 << 25 : (int64_t)1 << 26), ANR_E_BADARG,
                "anr_mlp_forward_save: n=%lld rows per call exceed the saved-activation layout (2^25 fp32 / 2^26 bf16): chunk the call",
                (long long)n);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)pts | (uintptr_t)out | (uintptr_t)act) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_forward_save: pack/pts/out/act must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool so = (mode & ANR_MLP_FLAG_SIGMA_ONLY) != 0;
    if (mode & ANR_MLP_FLAG_TANGENT) {
        ANR_REQUIRE(so && !index && n % 4 == 0, ANR_E_BADARG,
                    "anr_mlp_forward_save: tangent mode = sigma only, no index list, points in quads (n %% 4 == 0)");
        switch (mode & 0xff) {
            case ANR_MLP_F32:  return launch_mlp<ANR_MLP_F32, true, true, true, false, true>(pack, pts, n, out, st, act);
            case ANR_MLP_BF16: return launch_mlp<ANR_MLP_BF16_W8, true, true, true, false, true>(pack, pts, n, out, st, act);
            default: return fail(ANR_E_BADARG, "anr_mlp_forward_save: unknown mode %d", mode);
        }
    }
    if (mode & ANR_MLP_FLAG_BITS_ONLY) {
        ANR_REQUIRE(!so, ANR_E_BADARG, "anr_mlp_forward_save: ANR_MLP_FLAG_BITS_ONLY serves the full network (rgb + sigma)");
        switch (mode & 0xff) {
            case ANR_MLP_F32:  return launch_mlp<ANR_MLP_F32, true, false, true, false, false, false, true>(pack, pts, n, out, st, act, index, count);
            case ANR_MLP_BF16: return launch_mlp<ANR_MLP_BF16_W8, true, false, true, false, false, false, true>(pack, pts, n, out, st, act, index, count);
            default: return fail(ANR_E_BADARG, "anr_mlp_forward_save: unknown mode %d", mode);
        }
    }
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return so ? launch_mlp<ANR_MLP_F32, true, true, true>(pack, pts, n, out, st, act, index, count)
                      : launch_mlp<ANR_MLP_F32, true, false, true>(pack, pts, n, out, st, act, index, count);
        case ANR_MLP_BF16:
            return so ? launch_mlp<ANR_MLP_BF16_W8, true, true, true>(pack, pts, n, out, st, act, index, count)
                      : launch_mlp<ANR_MLP_BF16_W8, true, false, true>(pack, pts, n, out, st, act, index, count);
        default: return fail(ANR_E_BADARG, "anr_mlp_forward_save: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_forward(const void* pack, int mode, const float* pts, int64_t n, float* out, void* stream) {
    return anr_mlp_forward_indexed(pack, mode, pts, nullptr, nullptr, n, out, stream);
}

extern "C" int anr_mlp_forward_rays(const void* pack, int mode, const float* rays, int ray_stride, const float* z, int K,
                                    int64_t n, float* out, void* stream) {
    ANR_REQUIRE(pack && rays && z && out, ANR_E_BADARG, "anr_mlp_forward_rays: null pointer");
    ANR_REQUIRE(n > 0 && n < (int64_t)1 << 32 && K > 0 && n % K == 0 && ray_stride >= 8, ANR_E_BADARG,
                "anr_mlp_forward_rays: n=%lld K=%d stride=%d", (long long)n, K, ray_stride);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)out) & 15) == 0, ANR_E_ALIGN, "anr_mlp_forward_rays: pack/out must be 16-B aligned");
    ANR_REQUIRE(!(mode & ANR_MLP_FLAG_SIGMA_ONLY), ANR_E_BADARG, "anr_mlp_forward_rays: rgb + sigma only");
    hipStream_t st = (hipStream_t)stream;
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return launch_mlp<ANR_MLP_F32, true, false, false>(pack, z, n, out, st, nullptr, nullptr, nullptr, rays, ray_stride, K);
        case ANR_MLP_BF16:
            return launch_mlp<ANR_MLP_BF16_W8, true, false, false>(pack, z, n, out, st, nullptr, nullptr, nullptr, rays, ray_stride, K);
        default: return fail(ANR_E_BADARG, "anr_mlp_forward_rays: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_forward_embedded(const void* pack, int mode, const float* emb, int64_t n, float* out, void* act_v,
                                        void* stream) {
    ANR_REQUIRE(pack && emb && out, ANR_E_BADARG, "anr_mlp_forward_embedded: null pointer");
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_forward_embedded: n=%lld", (long long)n);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)out) & 15) == 0 && ((uintptr_t)emb & 3) == 0, ANR_E_ALIGN,
                "anr_mlp_forward_embedded: pack/out must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool so = (mode & ANR_MLP_FLAG_SIGMA_ONLY) != 0;
    if (act_v != nullptr) {
        ANR_REQUIRE(!so && ((uintptr_t)act_v & 15) == 0, ANR_E_BADARG, "anr_mlp_forward_embedded: act needs the full network and 16-B alignment");
        float* act = reinterpret_cast<float*>(act_v);
        switch (mode & 0xff) {
            case ANR_MLP_F32:  return launch_mlp<ANR_MLP_F32, true, false, true, true>(pack, emb, n, out, st, act);
            case ANR_MLP_BF16: return launch_mlp<ANR_MLP_BF16_W8, true, false, true, true>(pack, emb, n, out, st, act);
            default: return fail(ANR_E_BADARG, "anr_mlp_forward_embedded: unknown mode %d", mode);
        }
    }
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return so ? launch_mlp<ANR_MLP_F32, true, true, false, true>(pack, emb, n, out, st, nullptr)
                      : launch_mlp<ANR_MLP_F32, true, false, false, true>(pack, emb, n, out, st, nullptr);
        case ANR_MLP_BF16:
            return so ? launch_mlp<ANR_MLP_BF16_W8, true, true, false, true>(pack, emb, n, out, st, nullptr)
                      : launch_mlp<ANR_MLP_BF16_W8, true, false, false, true>(pack, emb, n, out, st, nullptr);
        default: return fail(ANR_E_BADARG, "anr_mlp_forward_embedded: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_forward_rays_steps(const void* pack, int mode, const float* rays, int ray_stride, const float* steps,
                                          int K, int64_t n, float* out, void* stream) {
    ANR_REQUIRE(pack && rays && steps && out, ANR_E_BADARG, "anr_mlp_forward_rays_steps: null pointer");
    ANR_REQUIRE(n > 0 && n < (int64_t)1 << 32 && K > 0 && n % K == 0 && ray_stride >= 8, ANR_E_BADARG,
                "anr_mlp_forward_rays_steps: n=%lld K=%d stride=%d", (long long)n, K, ray_stride);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)out) & 15) == 0, ANR_E_ALIGN, "anr_mlp_forward_rays_steps: pack/out must be 16-B aligned");
    ANR_REQUIRE(!(mode & ANR_MLP_FLAG_SIGMA_ONLY), ANR_E_BADARG, "anr_mlp_forward_rays_steps: rgb + sigma only");
    hipStream_t st = (hipStream_t)stream;
    switch (mode & 0xff) {                              // (K < 0 tells the kernel that `pts` is the step table)
        case ANR_MLP_F32:
            return launch_mlp<ANR_MLP_F32, true, false, false>(pack, steps, n, out, st, nullptr, nullptr, nullptr, rays, ray_stride, -K);
        case ANR_MLP_BF16:
            return launch_mlp<ANR_MLP_BF16_W8, true, false, false>(pack, steps, n, out, st, nullptr, nullptr, nullptr, rays, ray_stride, -K);
        default: return fail(ANR_E_BADARG, "anr_mlp_forward_rays_steps: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_forward_indexed(const void* pack, int mode, const float* pts, const int32_t* index,
                                       const int32_t* count, int64_t n, float* out, void* stream) {
    ANR_REQUIRE(pack && pts && out, ANR_E_BADARG, "anr_mlp_forward: null pointer");
    ANR_REQUIRE(index || !count, ANR_E_BADARG, "anr_mlp_forward_indexed: count without index");
    ANR_REQUIRE(!index || n < (int64_t)1 << 31, ANR_E_BADARG, "anr_mlp_forward_indexed: n=%lld does not fit int32", (long long)n);
    ANR_REQUIRE(n > 0, ANR_E_BADARG, "anr_mlp_forward: n=%lld", (long long)n);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)pts | (uintptr_t)out) & 15) == 0, ANR_E_ALIGN,
                "anr_mlp_forward: pack/pts/out must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool dma = !(mode & ANR_MLP_FLAG_NO_DMA);
    if (mode & ANR_MLP_FLAG_SIGMA_ONLY) {              // out = float[n]: sigma only (trunk + sigma row, 83 % of the FLOPs)
        switch (mode & 0xff) {
            case ANR_MLP_F32:  return launch_mlp<ANR_MLP_F32, true, true, false>(pack, pts, n, out, st, nullptr, index, count);
            case ANR_MLP_BF16: return launch_mlp<ANR_MLP_BF16_W8, true, true, false>(pack, pts, n, out, st, nullptr, index, count);
            default: return fail(ANR_E_BADARG, "anr_mlp_forward: unknown mode %d", mode);
        }
    }
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return dma ? launch_mlp<ANR_MLP_F32, true, false, false>(pack, pts, n, out, st, nullptr, index, count) : launch_mlp<ANR_MLP_F32, false, false, false>(pack, pts, n, out, st, nullptr, index, count);
        case ANR_MLP_BF16:
            if (mode & ANR_MLP_FLAG_W4)
                return dma ? launch_mlp<ANR_MLP_BF16, true, false, false>(pack, pts, n, out, st, nullptr, index, count) : launch_mlp<ANR_MLP_BF16, false, false, false>(pack, pts, n, out, st, nullptr, index, count);
            return dma ? launch_mlp<ANR_MLP_BF16_W8, true, false, false>(pack, pts, n, out, st, nullptr, index, count) : launch_mlp<ANR_MLP_BF16_W8, false, false, false>(pack, pts, n, out, st, nullptr, index, count);
        default:
            return fail(ANR_E_BADARG, "anr_mlp_forward: unknown mode %d", mode);
    }
}

extern "C" int anr_mlp_forward_view(const void* pack, int mode, const float* pts, const float* viewdir, int view_stride,
                                    const int32_t* index, const int32_t* count, int64_t n, float* out, void* stream) {
    ANR_REQUIRE(pack && pts && viewdir && out, ANR_E_BADARG, "anr_mlp_forward_view: null pointer");
    ANR_REQUIRE(index || !count, ANR_E_BADARG, "anr_mlp_forward_view: count without index");
    ANR_REQUIRE(n > 0 && view_stride >= 3 && (!index || n < (int64_t)1 << 31), ANR_E_BADARG, "anr_mlp_forward_view: n=%lld stride=%d",
                (long long)n, view_stride);
    ANR_REQUIRE((((uintptr_t)pack | (uintptr_t)pts | (uintptr_t)out) & 15) == 0 && ((uintptr_t)viewdir & 3) == 0, ANR_E_ALIGN,
                "anr_mlp_forward_view: pack/pts/out must be 16-B aligned");
    ANR_REQUIRE(!(mode & (ANR_MLP_FLAG_SIGMA_ONLY | ANR_MLP_FLAG_TANGENT)), ANR_E_BADARG, "anr_mlp_forward_view: full network only");
    hipStream_t st = (hipStream_t)stream;
    switch (mode & 0xff) {
        case ANR_MLP_F32:
            return launch_mlp<ANR_MLP_F32, true, false, false, false, false, true>(pack, pts, n, out, st, nullptr, index, count, viewdir, view_stride, 1);
        case ANR_MLP_BF16:
            return launch_mlp<ANR_MLP_BF16_W8, true, false, false, false, false, true>(pack, pts, n, out, st, nullptr, index, count, viewdir, view_stride, 1);
        default: return fail(ANR_E_BADARG, "anr_mlp_forward_view: unknown mode %d", mode);
    }
}
