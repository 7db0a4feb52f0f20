/*
 * animnerf_hip.h — C ABI of libanimnerf_hip.so, the MI355X (gfx950) implementation
 * of Anim-NeRF's per-ray rendering hot path.
 *
 * The reference (JanaldoChen/Anim-NeRF) has no FFI for this path: it sits behind
 * Python nn.Module methods.  Each entry point below therefore
 * cites the reference Python lines it replaces.  The only native code the reference
 * calls on this path is the external KNN_CUDA wheel (`knn_cuda.KNN(k, transpose_mode=True)`,
 * call site models/anim_nerf.py:82-83,159); `anr_knn` is its drop-in.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the device that owns `stream`
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream);
 *   - the caller allocates every input, output and workspace; the library never
 *     allocates or frees device memory and keeps no pointer after return;
 *   - calls only enqueue work (asynchronous w.r.t. the host);
 *   - return value: 0 = ok, >0 = hipError_t, <0 = ANR_E_* ; anr_last_error() returns a
 *     thread-local message for the last non-zero return;
 *   - re-entrant, no global mutable state: safe from several host threads on
 *     different devices/streams;
 *   - all floating-point tensors are fp32 unless stated; index tensors are int64
 *     where the reference returns int64 (KNN), int32 otherwise.
 */
#ifndef ANIMNERF_HIP_H
#define ANIMNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANR_VERSION 100           /* major*10000 + minor*100 + patch */

#define ANR_E_BADARG   (-1)       /* null pointer / non-positive size */
#define ANR_E_SHAPE    (-2)       /* unsupported shape (e.g. K > ANR_MAX_SAMPLES) */
#define ANR_E_ALIGN    (-3)       /* pointer not aligned as documented */
#define ANR_E_UNSUPP   (-4)       /* feature not built into this library */

#define ANR_MAX_SAMPLES 256       /* samples per ray handled by composite / merge */
#define ANR_KNN_K       4         /* neighbours (reference configs: k_neigh = 4) */

/* MLP arithmetic modes (anr_mlp_*).  F32 is bit-for-bit an fp32 fmaf chain on the
 * f32-input MFMA (parity mode, 1e-4 gate); BF16 rounds weights and activations to
 * bf16 with fp32 accumulation (throughput mode, judged by PSNR). */
#define ANR_MLP_F32   0
#define ANR_MLP_BF16  1
/* OR-ed into `mode` of anr_mlp_forward: evaluate the trunk and the sigma row only (NeRF.get_sigma(only_sigma=True),
 * models/nerf.py:155-170; what extract_mesh.py:49-61 keeps); `out` is then float[n]. */
#define ANR_MLP_FLAG_SIGMA_ONLY 0x400
/* forward-mode normals (models/nerf.py:177-190, train.py:288-309): points come in quads, row 4p = point p, rows 4p+1..3 =
 * its tangents d/dx, d/dy, d/dz riding through the same layers (no bias, ReLU gate of the primal row).  Accepted, together
 * with ANR_MLP_FLAG_SIGMA_ONLY, by anr_mlp_forward_save (sigma of a tangent row = d sigma / d x_d), anr_mlp_backward,
 * anr_mlp_wgrad (bias gradients from the primal rows only) and anr_encode64 (rows 4p+1..3 = d enc / d x_d). */
#define ANR_MLP_FLAG_TANGENT 0x800
/* anr_mlp_wgrad: grads_out += (gradients accumulate over several calls, as autograd's .grad does; tensors a
 * sigma-only call does not produce are left alone instead of zero-filled). */
#define ANR_MLP_FLAG_ACCUMULATE 0x1000
/* anr_mlp_pack_bytes: the pack of a network with the view-dependent colour head fused (anr_mlp_pack_view / anr_mlp_forward_view) */
#define ANR_MLP_FLAG_VIEW 0x2000
/* anr_mlp_wgrad: the call runs NEXT TO other launches (a side stream, a parallel branch of a captured graph): half the
 * split-K slices — half the workgroups and half the partial products.  A lone call is ~20 % slower that way; next to the
 * backward chain of a training step it leaves the chain's launches room to run. */
#define ANR_MLP_FLAG_BACKGROUND 0x4000
/* The `_refine` stage of the shipped configs (configs/people_snapshot, the four <subject>_refine.yaml files: pretrained_model_requires_grad False,
 * train.py:433-437): the networks are loaded and FROZEN and only the poses train, so nothing ever reads the saved
 * activations or most of the activation gradients (they are the weight-gradient GEMMs' operands).
 * anr_mlp_forward_save[_indexed]: keep only the ReLU sign bits (304 B per row instead of 5.2 KB; the data area of `act` is
 * left untouched).  anr_mlp_backward[_counted]: write only the pre-activation gradients of layers 1 and 5 (columns 0..255 and
 * 1024..1279 of `dact`) — what anr_mlp_denc reads on the way to the sample points. */
/* anr_mlp_wgrad with ANR_MLP_FLAG_SIGMA_ONLY: the tensors behind sigma.bias — floats [anr_mlp_wgrad_sigma_floats(),
 * anr_mlp_wgrad_floats()) of grads_out — are left alone instead of zero-filled (a caller that adds only the leading
 * floats to its gradient buffer: one launch fewer) */
#define ANR_MLP_FLAG_NO_FILL 0x10000
#define ANR_MLP_FLAG_BITS_ONLY 0x8000
#define ANR_MLP_FLAG_ENC_ONLY  0x8000

int         anr_version(void);
const char* anr_last_error(void);

/* ---- a1: ray generation --------------------------------------------------------
 * datasets/anim_nerf_dataset.py:56-85 (gen_ray_directions + gen_rays; the dead twins
 * utils/ray_utils.py:74-121 are the same with scalar focal, centre (W/2,H/2)).
 * c2w[12] row-major 3x4; focal[2]; center[2] (all device fp32).
 * rays_out[H*W*8] = [o(3), d(3), near, far], row-major over (row j, col i). */
int anr_ray_gen(const float* c2w, const float* focal, const float* center,
                int H, int W, float near, float far, float* rays_out, void* stream);

/* ---- a2: SMPL forward / linear blend skinning ------------------------------------------------------
 * smplx/body_models.py:289-387 + smplx/lbs.py:152-404 for model_type 'smpl'.
 * betas[bs*NB], pose[bs*J*3] (global_orient then body_pose, axis-angle), transl[bs*3];
 * tables as the reference registers them: v_template[V*3], shapedirs[V*3*NB], posedirs[9(J-1) * 3V] (row-major),
 * J_regressor[J*V], parents[J] (int64, parents[0] = -1), lbs_weights[V*J].
 * Outputs: verts[bs*V*3], joints[bs*J*3] (posed, + transl; the 21 extra vertex-joints are a gather the caller does),
 * A[bs*J*16] and T[bs*V*16] (both with transl added to the translation column, body_models.py:373-374),
 * shape_off[bs*V*3], pose_off[bs*V*3].  Workspaces: ws_v_shaped[bs*V*3], ws_joints_rest[bs*ceil(V/256)*J*3]
 * (per-block partial sums of the joint regression, added in a fixed order: no float atomics), ws_feat[bs*9(J-1)]. */
int anr_smpl_forward(const float* betas, const float* pose, const float* transl, int bs, int NB,
                     const float* v_template, const float* shapedirs, const float* posedirs,
                     const float* J_regressor, const int64_t* parents, const float* lbs_weights, int V, int J,
                     float* verts, float* joints, float* A, float* T, float* shape_off, float* pose_off,
                     float* ws_v_shaped, float* ws_joints_rest, float* ws_feat, void* stream);

/* ---- a4: rays into the root-joint frame -----------------------------------------
 * models/anim_nerf.py:128-137: o' = Ginv [o,1], d' = Ginv [d,0],
 * near' = max(near, |o'|-1), far' = min(far, |o'|+1).
 * g_inv[bs*16] row-major 4x4 (inverse of global_transform); rays_in/out [bs*R*stride]
 * (stride >= 8 floats; only the first 8 are read/written). */
int anr_rays_to_body(const float* g_inv, const float* rays_in, float* rays_out,
                     int bs, int R, int stride_in, void* stream);
/* ... and the body state: G^-1 in closed form (the root transform is affine) and verts[bs*V*3], joints[bs*J*3],
 * vertices_transform T[bs*V*16] and G itself moved into the root frame (models/anim_nerf.py:138-144), one launch.
 * g_inv_out[bs*16] is what anr_rays_to_body takes; g_root_out[bs*16] = G^-1 G. */
int anr_to_root_frame(const float* global_transform, const float* verts, const float* joints, const float* T, int bs,
                      int V, int J, float* g_inv_out, float* g_root_out, float* verts_out, float* joints_out,
                      float* T_out, void* stream);

/* ---- a5: observation -> canonical per-vertex transform -----------------------------
 * models/anim_nerf.py:147-151 (clac_ober2cano_transform):
 *   M = inverse(T_pose); M[:3,3] += (shape_t - shape) + (pose_t - pose); out = T_template @ M.
 * Matrices are affine (last row 0 0 0 1): the inverse is the closed-form 3x3 adjugate.
 * t_pose, out: [n*16] row-major 4x4, shape_off / pose_off: [n*3]; the template's three arrays hold n_template entries
 * (n_template = n: one template per frame; n_template = V: one template shared by all frames, n % n_template == 0). */
int anr_ober2cano(const float* t_pose, const float* t_template,
                  const float* shape_off, const float* shape_off_t,
                  const float* pose_off, const float* pose_off_t,
                  float* out, int64_t n, int64_t n_template, void* stream);

/* ---- a8: exact k=4 nearest SMPL vertices (replaces knn_cuda.KNN) ----------------------
 * models/anim_nerf.py:157-163.  The search is exact but spatially pruned, so it runs against a per-frame
 * index of the posed vertices instead of the raw table:
 *   anr_knn_index_build(verts[bs*V*3], order[V] or NULL, ...) -> index_out[bs * anr_knn_index_bytes(V)]
 * `order` (int32, slot -> vertex id) is any fixed permutation that keeps spatial neighbours together (the host
 * uses the Morton order of the rest-pose template, computed once per model); NULL = identity (still exact,
 * just less pruning).  Results do not depend on `order` except in the choice between exactly tied distances.
 *   anr_knn: xyz[bs*N*3] -> dist[bs*N*4] (Euclidean, ascending), idx[bs*N*4] (int64, 0-based vertex id). */
int64_t anr_knn_index_bytes(int V);
int anr_knn_index_build(const float* verts, const int32_t* order, int bs, int V, void* index_out, void* stream);
/* ... with the reach mask for validity radii up to dis_threshold (> 0): 32^3 bits over the body's box padded by dis_threshold,
 * set where some vertex lies within dis_threshold of the cell.  anr_warp_points (skip_far with a workspace) called with a
 * dis_threshold <= this one then drops the samples of unset cells in its classify pass — they cannot be valid
 * (models/anim_nerf.py:169-183: the blended distance is a convex combination of the four neighbours' distances) — instead of
 * searching their neighbours to find that out.  Same outputs bit for bit. */
int anr_knn_index_build_reach(const float* verts, const int32_t* order, int bs, int V, float dis_threshold, void* index_out,
                              void* stream);
int anr_knn(const void* knn_index, const float* xyz, int bs, int V, int64_t N,
            float* dist_out, int64_t* idx_out, void* stream);
/* d1_out[bs*N] = distance of xyz[bs*N*3] to the nearest vertex where that is below `radius`, +inf elsewhere: the same exact
 * search started from the bound radius — a query far from the body is settled by a handful of box tests.  (The empty-cell
 * test of the sigma grid: a cell whose centre is farther than dis_threshold + its half diagonal from every vertex holds no
 * valid voxel, models/anim_nerf.py:183.) */
int anr_knn_within(const void* knn_index, const float* xyz, int bs, int V, int64_t N, float radius, float* d1_out, void* stream);

/* ... and for k_neigh != 4 (a constructor argument of the reference, models/anim_nerf.py:42; every shipped config: 4): an
 * exhaustive exact search, k = 1..8, V <= 13000.  verts[bs*V*3], xyz[bs*N*xyz_stride] (xyz first) ->
 * dist[bs*N*k] (ascending; ties: lower vertex id first), idx[bs*N*k] int64. */
int anr_knn_k(const float* verts, const float* xyz, int xyz_stride, int bs, int V, int64_t N, int k, float* dist_out,
              int64_t* idx_out, void* stream);

/* ---- a6+a7: coarse depths and sample points ------------------------------------------
 * models/volume_rendering.py:29-46 (lindisp=True branch) and :117:
 *   z[r,k] = near (1 - s[k]) + far s[k];  if t_rand != NULL the stratified jitter of
 *   :48-54 is applied with the caller's uniform numbers t_rand[R*K] (already multiplied by `perturb`).
 * rays[R*stride]; steps[K] = torch.linspace(0, 1-1/K, K) supplied by the caller.
 * z_out[R*K]. */
int anr_sample_coarse(const float* rays, int stride, const float* steps, const float* t_rand,
                      int64_t R, int K, float* z_out, void* stream);

/* ---- a7-a10: points -> canonical space ----------------------------------------------------
 * models/volume_rendering.py:117 + models/anim_nerf.py:153-192 (get_neighbs, unpose).
 * Points are either given (xyz != NULL, [bs*N*xyz_stride]) or generated from rays and depths
 * (xyz == NULL: point n = ray n / K, sample n % K, x = o + z d).
 * Per body b: knn_index[b] (anr_knn_index_build), ober2cano[b][V*16]; lbs_weights[V*J] shared.
 * pts_out[bs*N*4] = (x_c, y_c, z_c, valid) with valid = 1.0 iff blended distance < dis_threshold.
 * skip_far != 0: points whose distance to the body's bounding box is >= dis_threshold (hence provably
 *   invalid: sigma = -1e5, zero compositing weight) get (x, y, z, 0) without a search.  Rendered outputs are
 *   unchanged; per-point rgb of such points differs from the reference's, so the point-query API passes 0.
 * Optional debug outputs (may be NULL; need skip_far = 0): dist_out[bs*N*4], idx_out[bs*N*4] (int32),
 *   blended_out[bs*N].
 * Optional training outputs (both or neither): nbr_idx_out[bs*N*4] (int32 vertex ids) and nbr_w_out[bs*N*4]
 *   (normalised blend weights, anim_nerf.py:169-171; zeros for samples skipped by skip_far) — what the backward
 *   pass needs to route dL/dx_c into ober2cano and into the sample position.
 * ws (may be NULL; used with skip_far != 0): int32 workspace of anr_warp_ws_ints(bs, N) elements.  With it the call
 *   runs in passes — classify every sample against the body's bounding box and list the near ones, counting-sort
 *   the list by 64^3 grid cell, then search the sorted list 64 entries per wavefront (all lanes busy, all in one or
 *   two neighbouring cells) — instead of one pass in which a wavefront of 64 neighbouring rays searches for its few
 *   near samples.  Same results bit for bit. */
int64_t anr_warp_ws_ints(int bs, int64_t N);
/* skip_far & 2 (with ws): the caller has ALREADY zeroed the workspace's counters — ints [first, first + count) of ws, from
 * anr_warp_ws_zero_range — e.g. together with the other fills of a training step (anr_zero_segments); the call then
 * launches no fill of its own.  The range depends on (bs, N) only. */
int anr_warp_ws_zero_range(int bs, int64_t N, int64_t* first_int_out, int64_t* ints_out);
int anr_warp_points(const float* xyz, int xyz_stride,
                    const float* rays, int ray_stride, const float* z, int K,
                    const void* knn_index, const float* ober2cano, const float* lbs_weights,
                    int bs, int V, int J, int64_t N, float dis_threshold, int skip_far,
                    float* pts_out, float* dist_out, int32_t* idx_out, float* blended_out,
                    int32_t* nbr_idx_out, float* nbr_w_out, int32_t* ws, void* stream);

/* Renderer variant with the validity handed over the cheap way (needs skip_far != 0 and ws): valid_mask_out[bs*N]
 * (1 byte per sample: 1 iff valid), valid_index_out[bs*N] / valid_count_out[1] (DEVICE counter) = the flat positions
 * b*N + n of the valid samples, ready for anr_mlp_forward_indexed.  pts_out rows of samples outside the body's
 * bounding box + dis_threshold are then NOT written (nothing reads them: the MLP goes by the list, the compositor by
 * the byte, anr_composite_masked).  All three NULL = anr_warp_points.
 * reuse_* (all or none; rays mode): the fine pass re-visits the coarse samples — sorted sample j of a ray is its
 *   coarse sample p = reuse_perm[j] when p < reuse_K (anr_sample_fine_merge_u8's perm_out) — so their canonical points
 *   reuse_pts[bs*R*reuse_K*4] and validity bytes reuse_mask[bs*R*reuse_K] from the coarse call are copied, not searched
 *   for again. */
int anr_warp_points_lean(const float* xyz, int xyz_stride,
                         const float* rays, int ray_stride, const float* z, int K,
                         const void* knn_index, const float* ober2cano, const float* lbs_weights,
                         int bs, int V, int J, int64_t N, float dis_threshold, int skip_far,
                         float* pts_out, float* dist_out, int32_t* idx_out, float* blended_out,
                         int32_t* nbr_idx_out, float* nbr_w_out, int32_t* ws,
                         uint8_t* valid_mask_out, int32_t* valid_index_out, int32_t* valid_count_out,
                         const float* reuse_pts, const uint8_t* reuse_mask, const uint8_t* reuse_perm, int reuse_K,
                         void* stream);
/* ... and the same reuse for the TRAINING step (models/volume_rendering.py:199-207 + models/anim_nerf.py:153-192 under autograd:
 * the fine pass of a step warps Kc + Kf sorted samples per ray, Kc of which the coarse pass has just warped).  Without the
 * validity outputs (valid_mask_out = NULL) and with reuse_mask = NULL, sorted sample j of a ray with p = reuse_perm[j] < reuse_K
 * takes the coarse call's WHOLE row reuse_pts[(ray, p)] = (x_c, valid) and — when nbr_idx_out / nbr_w_out are asked for — its
 * neighbour ids and blend weights reuse_nbr_idx / reuse_nbr_w[bs*R*reuse_K*4] (what anr_warp_backward needs); only the Kf
 * new samples are classified and searched.  Same bits as searching them again (z_sorted[j] IS z_coarse[p]).  With the validity
 * outputs it is anr_warp_points_lean (reuse_nbr_* = NULL). */
int anr_warp_points_reuse(const float* xyz, int xyz_stride,
                          const float* rays, int ray_stride, const float* z, int K,
                          const void* knn_index, const float* ober2cano, const float* lbs_weights,
                          int bs, int V, int J, int64_t N, float dis_threshold, int skip_far,
                          float* pts_out, float* dist_out, int32_t* idx_out, float* blended_out,
                          int32_t* nbr_idx_out, float* nbr_w_out, int32_t* ws,
                          uint8_t* valid_mask_out, int32_t* valid_index_out, int32_t* valid_count_out,
                          const float* reuse_pts, const uint8_t* reuse_mask, const uint8_t* reuse_perm, int reuse_K,
                          const int32_t* reuse_nbr_idx, const float* reuse_nbr_w, void* stream);

/* anr_warp_points_reuse + the per-cell results of an EARLIER call (round 6).  The cell-sorted search of a large call first
 * searches once per occupied cell of a grid over the body (radius that holds every point's four neighbours, first cluster to
 * scan, or "dead": no point of the cell can be valid); those results depend on the body (knn_index), the grid (64^3 from
 * 2^19 samples per body on, else 32^3) and dis_threshold — not on the samples.  prev_ws = the workspace of an earlier call with
 * the same knn_index, bs and dis_threshold (its N = prev_N; still intact: nothing has written to it since), e.g. the coarse
 * pass of the frame whose fine pass this is: cells that call searched are copied instead of searched again (the fine samples
 * of a ray fall into the cells its coarse samples fell into).  Ignored (a plain anr_warp_points_reuse) when either call is
 * too small for the cell pass or their grids differ.  prev_ws = NULL: anr_warp_points_reuse.  Same outputs bit for bit. */
int anr_warp_points_cells(const float* xyz, int xyz_stride,
                          const float* rays, int ray_stride, const float* z, int K,
                          const void* knn_index, const float* ober2cano, const float* lbs_weights,
                          int bs, int V, int J, int64_t N, float dis_threshold, int skip_far,
                          float* pts_out, float* dist_out, int32_t* idx_out, float* blended_out,
                          int32_t* nbr_idx_out, float* nbr_w_out, int32_t* ws,
                          uint8_t* valid_mask_out, int32_t* valid_index_out, int32_t* valid_count_out,
                          const float* reuse_pts, const uint8_t* reuse_mask, const uint8_t* reuse_perm, int reuse_K,
                          const int32_t* reuse_nbr_idx, const float* reuse_nbr_w,
                          const int32_t* prev_ws, int64_t prev_N, void* stream);

/* Backward of anr_warp_points (rays mode) for pose refinement (a16): d_pts[bs*N*4] (w component ignored) ->
 * d_ober2cano[bs*V*16] and d_rays[bs*R*8] (ACCUMULATED with atomics: zero them first), d_z[bs*N] (written).
 * nbr_idx / nbr_w are the training outputs of the forward. */
int anr_warp_backward(const float* d_pts, const float* rays, int ray_stride, const float* z, int K,
                      const float* ober2cano, const int32_t* nbr_idx, const float* nbr_w, int bs, int V, int64_t N,
                      float* d_ober2cano, float* d_rays, float* d_z, void* stream);

/* anr_warp_backward fed from the COMPACTED list of valid samples (the explicit training step): d_pts_rows[rows*4] holds the
 * gradient of the listed samples only (anr_mlp_dpoints' output), pos[bs*N] maps a sample to its row or to -1 (the inverse
 * map anr_compact_ordered writes); a sample without a row has no gradient.  Saves the expansion to bs*N dense rows. */
int anr_warp_backward_compact(const float* d_pts_rows, const int32_t* pos, const float* rays, int ray_stride, const float* z,
                              int K, const float* ober2cano, const int32_t* nbr_idx, const float* nbr_w, int bs, int V,
                              int64_t N, float* d_ober2cano, float* d_rays, float* d_z, void* stream);

/* Same output layout without the warp (use_unpose=False, models/anim_nerf.py:296-297):
 * pts_out = (x, y, z, 1). */
int anr_points_from_rays(const float* rays, int ray_stride, const float* z, int K,
                         int64_t n_points, float* pts_out, void* stream);

/* ---- a11+a12: Fourier encoding + 8x256 MLP ---------------------------------------------------
 * models/embedding.py:22-39, models/nerf.py:129-175 (use_view=False, no latent codes).
 * Weights are handed over pre-packed: anr_mlp_pack_bytes(mode) bytes produced by
 * anr_mlp_pack() from the module's 11 weight/bias tensors in PyTorch [out,in] layout.
 * pts[n*4] = (x,y,z,valid) -> out[n*4] = (r,g,b,sigma), sigma = -1e5 where valid < 1
 * (models/anim_nerf.py:305). */
int64_t anr_mlp_pack_bytes(int mode);

typedef struct anr_mlp_params {
    const float* w_trunk[8];  const float* b_trunk[8];   /* xyz_encoding_{1..8}.0: [256,63],[256,256]x3,[256,319],[256,256]x3 */
    const float* w_sigma;     const float* b_sigma;      /* sigma: [1,256] */
    const float* w_final;     const float* b_final;      /* xyz_encoding_final: [256,256] */
    const float* w_dir;       const float* b_dir;        /* dir_encoding.0: [128,256] */
    const float* w_rgb;       const float* b_rgb;        /* rgb.0: [3,128] */
} anr_mlp_params;

int anr_mlp_pack(const anr_mlp_params* p, int mode, void* pack_out, void* stream);
/* two networks (models/anim_nerf.py: nerf and nerf_fine, repacked after every optimiser step) in ONE launch */
int anr_mlp_pack_pair(const anr_mlp_params* p_a, const anr_mlp_params* p_b, int mode, void* pack_a_out, void* pack_b_out, void* stream);
/* use_view=True (the reference's class default, models/nerf.py:60-153; no shipped config selects it): the colour head reads
 * [xyz_encoding_final (256), Embedding(viewdir) (dir_channels = 3 + 6 freqs_dir)] -> 128 -> 3.  anr_mlp_pack_view takes
 * p->w_dir as [128][256 + dir_channels] (pack_out: anr_mlp_pack_bytes(mode | ANR_MLP_FLAG_VIEW) bytes) and
 * anr_mlp_forward_view evaluates the whole network in the fused kernel: the direction's Fourier panel is computed in registers
 * like the position's and enters the four dir_encoding tiles in front of the feature, as the skip layer takes the position's.
 * viewdir[n * view_stride] (xyz first): one direction per point of pts (per listed point pts[index[i]] reads viewdir[index[i]]).
 * index / count as in anr_mlp_forward_indexed (NULL: all n points).  Inference; under autograd the head runs as framework ops
 * on the kernel's feature (anr_mlp_backward_feature). */
int anr_mlp_pack_view(const anr_mlp_params* p, int mode, int dir_channels, void* pack_out, void* stream);
int anr_mlp_forward_view(const void* pack, int mode, const float* pts, const float* viewdir, int view_stride,
                         const int32_t* index, const int32_t* count, int64_t n, float* out, void* stream);

int anr_mlp_forward(const void* pack, int mode, const float* pts, int64_t n,
                    float* out, void* stream);

/* Evaluate the field only where it is consumed.  With use_unpose=True most samples lie farther than dis_threshold
 * from the body (93 % of the coarse samples on the synthetic frames): their sigma is the constant -1e5
 * (models/anim_nerf.py:305), compositing gives them weight exactly 0, and their MLP output never reaches a result.
 * The reference has the same idea as `query_inside` / `inside_inds` (models/anim_nerf.py:245-290: evaluate
 * xyz[inside_inds], rgb = 0 and sigma = -1e5 elsewhere).
 *
 * anr_compact_valid: index_out[0..*count_out) = positions i with pts[4i+3] >= 1 (order preserved inside blocks of
 * 1024 samples, block order unspecified); *count_out is a DEVICE int32 — no host synchronisation.  fill_out (may be
 * NULL): rows of the invalid samples are set to (0,0,0,-1e5) (fill_cols = 4) or -1e5 (fill_cols = 1, sigma-only).
 *
 * anr_mlp_forward_indexed: for i < min(n, *count) (count may be NULL: all n): out[index[i]] = NeRF(pts[index[i]]);
 * other rows of out are not touched.  index = NULL is anr_mlp_forward.  n bounds the launch (n < 2^31). */
/* No warp (use_unpose=False, models/anim_nerf.py:296-297): the kernel generates the sample points itself,
 * point i = ray i / K, depth z[i], x = o + z d (rounded like anr_points_from_rays), valid = 1 — the 16-B point array
 * of anr_points_from_rays + anr_mlp_forward is never written or read.  Same output bits.  n < 2^32, n % K == 0. */
int anr_mlp_forward_rays(const void* pack, int mode, const float* rays, int ray_stride, const float* z, int K,
                         int64_t n, float* out, void* stream);
/* ... and no depth array either: the deterministic stratified depths of models/volume_rendering.py:43-44,
 * z = near' (1 - steps[k]) + far' steps[k] (k = i % K; roundings of anr_sample_coarse), are computed in the kernel too. */
int anr_mlp_forward_rays_steps(const void* pack, int mode, const float* rays, int ray_stride, const float* steps, int K,
                               int64_t n, float* out, void* stream);
/* The network on an already embedded input, emb[n*63] = (x, sin 2^k x, cos 2^k x)_k in the reference's channel order:
 * models/mlp.py:268-297 (NeRF.forward(input_xyz, ...), the pre-embedded twin of models/nerf.py) — the encoder is skipped.
 * out[n*4] = (r,g,b,sigma), or sigma[n] with ANR_MLP_FLAG_SIGMA_ONLY (only_sigma=True, :283-284).
 * act (may be NULL): the saved activations of anr_mlp_forward_save (blocked layout, below) — columns 2048..2303 are
 * xyz_encoding_final, the input of a view-dependent colour head (in_channels_dir > 0) that the caller evaluates. */
int anr_mlp_forward_embedded(const void* pack, int mode, const float* emb, int64_t n, float* out, void* act,
                             void* stream);
int anr_compact_valid(const float* pts, int64_t n, int32_t* index_out, int32_t* count_out,
                      float* fill_out, int fill_cols, void* stream);
int anr_mlp_forward_indexed(const void* pack, int mode, const float* pts, const int32_t* index,
                            const int32_t* count, int64_t n, float* out, void* stream);

/* Training forward: the same kernel, additionally storing each layer's post-activation output for the backward
 * pass (what autograd keeps alive in the reference, models/nerf.py:163-175): act = n * anr_mlp_act_cols() elements,
 * fp32 in mode ANR_MLP_F32 and bf16 in mode ANR_MLP_BF16 (the value the next layer consumed).
 * Columns: [h1..h8 (8 x 256, post-ReLU) | xyz_encoding_final (256) | dir hidden (128, post-ReLU)] = 2432; with
 * ANR_MLP_FLAG_SIGMA_ONLY only h1..h8 (and their bits) are written.
 * LAYOUT — blocked by 32-column tile, so that what a wavefront stores per out-tile (32 points x 32 features) is one
 * contiguous 2 KiB: column c of row r is element (c / 32 * n + r) * 32 + c % 32, i.e. 76 blocks [n][32].  Behind them, at
 * element 2432 * n, the SIGN BITS of the ReLU'd columns (one uint16 per block of 32 features and half-wave: bit 4Q+i <->
 * feature 32w + 8Q + 4h + i is > 0): per trunk layer l a block [n][2 half-waves][8 uint16] at byte 32 n (l-1), the colour
 * head's as [n][2][4 uint16] at byte 288 n — what anr_mlp_backward gates with, instead of re-reading the activations.
 * anr_mlp_act_cols() = 2592 elements per row in either dtype is the ALLOCATION size per row (2432 + bits + padding), not a
 * row pitch; `n` must be the same in every call that touches a buffer (it is the block stride).  dact uses the same blocks;
 * its tail is unused. */
int anr_mlp_act_cols(void);
int anr_mlp_forward_save(const void* pack, int mode, const float* pts, int64_t n,
                         float* out, void* act, void* stream);
/* ... on a compacted list (see anr_mlp_forward_indexed): activation row i belongs to sample index[i]. */
int anr_mlp_forward_save_indexed(const void* pack, int mode, const float* pts, const int32_t* index,
                                 const int32_t* count, int64_t n, float* out, void* act, void* stream);

/* ---- a16 (part): backward of the MLP w.r.t. its activations ------------------------------------------------------
 * What autograd differentiates in models/nerf.py:129-175, as one kernel (csrc/mlp_bwd.hip): from
 *   g[n*4] = (dL/d rgb_pre (3: upstream x sigmoid'), dL/d sigma (already 0 where the sample is invalid))
 * and the activations saved by anr_mlp_forward_save (only their sign bits are read), the pre-activation gradient of every
 * layer in the SAME blocked layout and dtype as `act`: columns 256(l-1).. = layer l of the trunk, 2048.. =
 * xyz_encoding_final, 2304.. = dir_encoding.  The weight gradients are then the GEMMs dact_l^T act_{l-1}
 * (anr_mlp_wgrad).  With ANR_MLP_FLAG_SIGMA_ONLY only the trunk columns are written (g = (., ., ., dL/d sigma)).
 * The transposed weights are handed over pre-packed: anr_mlp_bwd_pack_bytes(mode) bytes from anr_mlp_bwd_pack(). */
/* models/embedding.py:22-39 as a row-major matrix enc[n*63] (fp32, or bf16 if bf16_out) — the input operand of the
 * weight-gradient GEMMs of xyz_encoding_1 / _5 — and its chain rule: d_pts[n*4] = (dL/dx, dL/dy, dL/dz, 0) from
 * d_enc[n*63] (fp32).  pts[n*pts_stride] with xyz first. */
int anr_encode(const float* pts, int pts_stride, int64_t n, int bf16_out, void* enc_out, void* stream);
int anr_encode_backward(const float* pts, int pts_stride, const float* d_enc, int64_t n, float* d_pts_out,
                        void* stream);
/* ... with a 64th, zero, column: the operand layout anr_mlp_wgrad stages (16-byte rows of 8 bf16 / 4 fp32).
 * flags: 1 = bf16 output, ANR_MLP_FLAG_TANGENT = quads (see the flag). */
int anr_encode64(const float* pts, int pts_stride, int64_t n, int flags, void* enc_out, void* stream);
int64_t anr_mlp_bwd_pack_bytes(int mode);
int anr_mlp_bwd_pack(const anr_mlp_params* p, int mode, void* pack_out, void* stream);
int anr_mlp_bwd_pack_pair(const anr_mlp_params* p_a, const anr_mlp_params* p_b, int mode, void* pack_a_out, void* pack_b_out,
                          void* stream);
int anr_mlp_backward(const void* bwd_pack, int mode, const float* g, const void* act, void* dact, int64_t n,
                     void* stream);
/* ... entered at xyz_encoding_final instead of at the colour head: use_view=True (models/nerf.py:141-153) evaluates
 * [feature, encoding(viewdir)] -> 128 -> 3 outside the kernels and hands over d_feature[n*256] (fp32) = dL/d
 * xyz_encoding_final; g[n*4] carries dL/d sigma in its 4th column (columns 0..2 are ignored).  dact columns 0..2303 are
 * written as by anr_mlp_backward (2048.. = d_feature in the activation dtype), 2304..2431 are zero-filled, so that
 * anr_mlp_wgrad on the result gives the trunk, sigma and xyz_encoding_final gradients (and zeros for the colour head). */
int anr_mlp_backward_feature(const void* bwd_pack, int mode, const float* g, const float* d_feature, const void* act,
                             void* dact, int64_t n, void* stream);

/* ---- a16 / f2: backward of the per-frame chain (pose refinement, optim_body_params) ------------------------------
 * dL/d(betas[10], global_orient[3], body_pose[69], transl[3]) per frame from dL/d ober2cano[bs*V*16] (may be NULL) and
 * dL/d rays_body[bs*R*8] (may be NULL) — everything torch autograd differentiates in smplx/lbs.py:152-251,
 * models/anim_nerf.py:128-151 — in ONE launch: a workgroup per (frame, parameter) pushes that parameter's unit tangent
 * through the chain in forward mode and dots it with the upstream gradients (csrc/frame_bwd.hip).
 * pose[bs*72] = (global_orient, body_pose); J0[24*3] = J_regressor . v_template, JS[24*3*10] = J_regressor . shapedirs
 * (constants of the body model); posedirs[207 * 3V] row-major, shapedirs[V*3*10], lbs_weights[V*24], parents[24] int64;
 * T_template[template_bs*V*16] = the template pose's per-vertex transforms (template_bs = 1: shared by all frames);
 * rays_world[bs*R*ray_stride] = the rays before convert_to_body_model_space.  grads_out[bs*85] in the order above.
 * vertex_joint_mask[V] (int32, may be NULL): bit j set where lbs_weights[v][j] != 0 — lets a body_pose parameter's
 * workgroup skip the dual-number inverse of every vertex its joint's subtree does not move (exact: zero tangents). */
int anr_frame_backward(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                       const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                       const float* posedirs, int V, const float* T_template, int template_bs,
                       const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                       const float* d_rays_body, const int32_t* vertex_joint_mask, float* grads_out, void* stream);

/* The same gradients with the work the other way round (what the training step calls): reverse mode through the per-vertex
 * inverses once per frame — g_A[24][12], g_Ginv[12], g_transl[3] and the adjoint of the blend-shape offsets —, the
 * contraction of the latter with posedirs / shapedirs, and forward mode only through the 24-joint chain, dotted with those
 * (three launches, ~50x fewer flops; atomics over the vertices, like anr_warp_backward).  anr_frame_backward above is kept
 * as the cross-check.  workspace[anr_frame_backward_ws_floats(bs, V)] fp32. */
int64_t anr_frame_backward_ws_floats(int bs, int V);
int anr_frame_backward_adjoint(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                               const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                               const float* posedirs, int V, const float* T_template, int template_bs,
                               const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                               const float* d_rays_body, float* workspace, float* grads_out, void* stream);

/* ... with the VALUES of the chain handed in as the forward kernels left them — joints_transform[bs*J*16] (anr_smpl_forward's A,
 * transl on its translation column) and g_inv[bs*16] (anr_to_root_frame's inverse root transform) — instead of being
 * recomputed by every workgroup of the per-vertex kernel (both NULL: anr_frame_backward_adjoint). */
int anr_frame_backward_adjoint_values(const float* betas, const float* pose, const float* transl, int bs, const float* J0,
                                      const float* JS, const int64_t* parents, const float* lbs_weights, const float* shapedirs,
                                      const float* posedirs, int V, const float* T_template, int template_bs,
                                      const float* rays_world, int ray_stride, int R, const float* d_ober2cano,
                                      const float* d_rays_body, const float* joints_transform, const float* g_inv,
                                      float* workspace, float* grads_out, int flags, void* stream);
/* flags & 1: the first anr_frame_backward_ws_zero_floats(bs) floats of workspace (the accumulators) are zero already — filled by
 * the caller with the other fills of its step (anr_zero_segments) — and the call launches no fill of its own. */
int64_t anr_frame_backward_ws_zero_floats(int bs);

/* ---- a16 (part): weight and bias gradients of the MLP -----------------------------------------------------------
 * What autograd computes for the 22 parameter tensors of models/nerf.py:60-127 once the activation gradients exist:
 *   dW_l = dact_l^T in_l (in_1 = enc, in_5 = [enc, h4], in_l = h_{l-1}; xyz_encoding_final, dir_encoding on h8 / the
 *   feature; sigma and rgb from g), db_l = column sums — hand-written split-K MFMA GEMMs over the points
 *   (csrc/mlp_wgrad.hip; ds_read_b64_tr_b16 fragments in bf16, v_mfma_f32_32x32x2_f32 in fp32), slices added in a fixed order.
 * act, dact: the blocked buffers of anr_mlp_forward_save / anr_mlp_backward for the same n (dtype of `mode`), enc: anr_encode64 of
 * the same points and dtype, g[n*4] as handed to anr_mlp_backward.  n % 64 == 0 (pad with rows whose g is 0).
 * grads_out[anr_mlp_wgrad_floats()] fp32, PyTorch [out][in] layouts back to back in the order
 *   xyz_encoding_1..8 {weight, bias}, sigma {w, b}, xyz_encoding_final {w, b}, dir_encoding {w, b}, rgb {w, b};
 * with ANR_MLP_FLAG_SIGMA_ONLY the tensors behind sigma.bias are zero-filled.  workspace[anr_mlp_wgrad_ws_floats(n)] fp32. */
int64_t anr_mlp_wgrad_floats(void);
int64_t anr_mlp_wgrad_sigma_floats(void);
int64_t anr_mlp_wgrad_ws_floats(int64_t n);
int anr_mlp_wgrad(int mode, const void* act, const void* dact, const void* enc, const float* g, int64_t n,
                  float* workspace, float* grads_out, void* stream);

/* ... and the gradient that leaves the network through its two encoding inputs (pose refinement):
 * d_enc_out[n*63] = dact[:, 0:256] . W1[:, 0:63] + dact[:, 1024:1280] . W5[:, 0:63]  (xyz_encoding_1 / _5 weights, fp32,
 * PyTorch [256][63] / [256][319] layouts); anr_encode_backward turns it into dL/d xyz. */
int anr_mlp_denc(int mode, const void* dact, const float* w1, const float* w5, int64_t n, float* d_enc_out, void* stream);
/* anr_mlp_denc + anr_encode_backward in one launch (what the explicit training step calls): d_pts_out[n*4] = (dL/d xyz, 0) of
 * the rows of a compacted pass from their activation gradients `dact` (layers 1 and 5) and their canonical points pts[n*4];
 * the two encoding weight panels are read, pre-packed, from the BACKWARD weight pack (anr_mlp_bwd_pack writes them behind
 * the W^T fragments).  count (optional, device): the number of listed rows.  d_enc[n][63] never exists in memory. */
int anr_mlp_dpoints(const void* bwd_pack, int mode, const void* dact, const float* pts, int64_t n, const int32_t* count,
                    float* d_pts_out, void* stream);

/* ---- a16: the same steps on a compacted list whose LENGTH IS KNOWN ON THE DEVICE ONLY ------------------------------------
 * The reference's masked assignment (models/anim_nerf.py:284-289) makes the number of valid samples data dependent; reading it
 * back to size the saved activations costs a device -> host synchronisation per network pass (4 per training step: the queue
 * drains each time).  Instead every buffer is allocated for the upper bound n (all samples; rows that are never written cost
 * address space only) and the kernels take the length from `count`, a device int32: anr_compact_ordered writes
 * count_out[0] = listed rows and count_out[1] = that rounded up to a multiple of 64 (at least 64; the padding rows are valid =
 * 0 / gradient 0) — pass count_out + 1 to the *_counted entry points and to anr_mlp_forward_save_indexed (index = NULL),
 * count_out itself to anr_mlp_head_grad_counted.  n stays the row count of the BUFFERS (the block stride of act / dact).
 * Rows past the count are neither read nor written. */
int anr_mlp_head_grad_counted(const float* g, const int32_t* index, const float* out, const float* pts, const int32_t* count,
                              int64_t n, int sigma_only, float* g4_out, void* stream);
int anr_mlp_backward_counted(const void* bwd_pack, int mode, const float* g, const void* act, void* dact, int64_t n,
                             const int32_t* count, void* stream);
int anr_encode64_counted(const float* pts, int pts_stride, int64_t n, const int32_t* count, int flags, void* enc_out, void* stream);
int anr_mlp_wgrad_counted(int mode, const void* act, const void* dact, const void* enc, const float* g, int64_t n,
                          const int32_t* count, float* workspace, float* grads_out, void* stream);
int anr_mlp_denc_counted(int mode, const void* dact, const float* w1, const float* w5, int64_t n, const int32_t* count,
                         float* d_enc_out, void* stream);
int anr_encode_backward_counted(const float* pts, int pts_stride, const float* d_enc, int64_t n, const int32_t* count,
                                float* d_pts_out, void* stream);

/* ---- f4: the level set of the density grid as a triangle mesh (extract_mesh.py:159-173: mcubes.marching_cubes(-sigmas, 0.)) ---
 * Marching cubes over volume[n0][n1][n2] (fp32, axis 2 fastest), inside = value < level, in two passes with the caller's
 * exclusive prefix sums in between (csrc/mesh.hip):
 *   anr_mc_classify: vmask_out[p] bit a = the grid edge from point p along axis a crosses the level (one vertex each),
 *     vcount_out[p] = popcount, tcount_out[p] = triangles of the cube based at p (n_tris[256]: triangles per sign case, bit c
 *     of the case = corner (c&1, (c>>1)&1, (c>>2)&1) inside);
 *   anr_mc_emit: verts_out[V*3] in index coordinates (vertex vstart[p] + rank of its axis among p's crossing edges, at the
 *     linear interpolation point), faces_out[T*3] vertex ids; tris[256*8*3] int8: the case table's triangles as cube-edge ids
 *     (edges enumerated as the corner pairs (a < b) differing in one bit, in lexicographic order; -1 padded).
 * The case table itself is generated by anim_nerf_amd/mesh.py (same face rule on both sides of a shared face: no cracks).
 * PyMCubes is absent from the image: parity-unpinned, held to properties (tests/test_mesh.py, test_gpu_parity.py). */
int anr_mc_classify(const float* volume, int n0, int n1, int n2, float level, const uint8_t* n_tris, uint8_t* vmask_out,
                    int32_t* vcount_out, int32_t* tcount_out, void* stream);
int anr_mc_emit(const float* volume, int n0, int n1, int n2, float level, const uint8_t* n_tris, const int8_t* tris,
                const uint8_t* vmask, const int64_t* vstart, const int64_t* tstart, float* verts_out, int32_t* faces_out,
                void* stream);

/* ---- a16 / f1: the steps between the big kernels of a training step, one launch each (csrc/train_glue.hip) --------
 * anr_compact_ordered: `inside_inds` of models/anim_nerf.py:253 in sample order.  index_out[0..*count) = positions i with
 *   pts[4i+3] >= 1, ascending; pos_out[i] = row of sample i in that list or -1; pts_out[r] = pts[index[r]], followed by
 *   zero rows (valid = 0) up to the next multiple of 64 (at least 64 rows): the operand of anr_mlp_forward_save.
 *   pts_out needs ((n + 63) / 64) * 64 rows; workspace anr_compact_ws_ints(n) int32; count_out is TWO DEVICE int32: the number
 *   of listed rows, and that number rounded up to a multiple of 64 (at least 64) — the rows the *_counted kernels work on.
 * anr_expand_rows: out[i] = pos[i] >= 0 ? src[pos[i]] : (0, 0, 0, fill) (cols = 4) / fill (cols = 1), i < n — what the
 *   reference's masked assignment does (models/anim_nerf.py:284-289).
 * anr_mlp_head_grad: the g[n_pad*4] operand of anr_mlp_backward / anr_mlp_wgrad from autograd's upstream gradient
 *   g_in ([n_full][4] on (rgb, sigma), or [n_full] on sigma with sigma_only): row r < rows reads g_in[index ? index[r] : r],
 *   rgb columns times sigmoid' = out (1 - out) (out[n_pad*4] from anr_mlp_forward_save), sigma column zeroed where
 *   pts[4r+3] < 1; rows >= `rows` are zero.
 * anr_tangent_quads: xyz[n*3] -> pts4[4*n_pad*4]: rows 4p..4p+3 = (x, y, z, 1) for p < n, zero rows after
 *   (ANR_MLP_FLAG_TANGENT operand: a point and its three directional derivatives). */
int64_t anr_compact_ws_ints(int64_t n);
int anr_compact_ordered(const float* pts, int64_t n, int32_t* index_out, int32_t* pos_out, float* pts_out,
                        int32_t* count_out, int32_t* workspace, void* stream);
/* ... with n_r = rows (n_fg + n_bg) RIDER points appended as samples n .. n + n_r - 1 with valid = 1 (the prior points of
 * train.py:262-286 evaluated with the step's ray samples): per frame its foreground points fg[rows][n_fg][3], then its
 * background points bg[rows][n_bg][3] (n_fg or n_bg = 0: absent) — the order anr_train_loss reads their sigmas in.
 * index_out / pos_out hold n + n_r entries, pts_out roundup64(n + n_r) rows, workspace anr_compact_ws_ints(n + n_r) ints. */
int anr_compact_ordered_riders(const float* pts, int64_t n, const float* fg, int n_fg, const float* bg, int n_bg, int rows,
                               int32_t* index_out, int32_t* pos_out, float* pts_out, int32_t* count_out, int32_t* workspace,
                               void* stream);
/* The same list in ONE launch (a chained scan with decoupled look-back instead of count / scan / gather): what the explicit
 * training step calls.  state[anr_compact_state_words(n + riders)] int64 must be ZERO before its first use and is left zero by
 * every call (a call in a replayed HIP graph needs no fill in front of it); calls sharing a state buffer must not overlap. */
int64_t anr_compact_state_words(int64_t n);
int anr_compact_ordered_single(const float* pts, int64_t n, const float* fg, int n_fg, const float* bg, int n_bg, int rows,
                               int32_t* index_out, int32_t* pos_out, float* pts_out, int32_t* count_out, int64_t* state,
                               void* stream);
int anr_expand_rows(const float* src, const int32_t* pos, int64_t n, int cols, float fill, float* out, void* stream);
int anr_mlp_head_grad(const float* g_in, const int32_t* index, const float* out, const float* pts, int64_t rows,
                      int64_t n_pad, int sigma_only, float* g_out, void* stream);
int anr_tangent_quads(const float* xyz, int64_t n, int64_t n_pad, float* pts4_out, void* stream);

/* Pose refinement moves near'/far' (models/anim_nerf.py:128-137), so the sampled depths carry a gradient to the rays:
 * anr_sample_coarse_backward: d_rays_out[R*8] = (0,0,0, 0,0,0, d near', d far') from g_z[R*K] (steps, t_rand as handed to
 *   anr_sample_coarse; models/volume_rendering.py:29-56);
 * anr_merge_backward: d_z_coarse_out[R*Kc] from g_sorted[R*K] and the permutation anr_sample_fine_merge returned
 *   (z_sorted[j] = cat(z_coarse, z_fine)[perm[j]]; z_fine is detached, models/volume_rendering.py:199-207). */
int anr_sample_coarse_backward(const float* g_z, const float* steps, const float* t_rand, int64_t R, int K,
                               float* d_rays_out, void* stream);
int anr_merge_backward(const float* g_sorted, const int32_t* perm, int64_t R, int K, int Kc, float* d_z_coarse_out,
                       void* stream);

/* Every loss term of train.py:228-309 in one launch, and their gradients in another.  Pointers of terms that are not
 * wanted are NULL (fine pass, priors, normals).  All device fp32.
 *   rgb[R*3], acc[R] (+ _fine), target_rgb[R*3], target_alpha[R]: MSE and L1 (train.py:228-246);
 *   s[prior_rows * (n_fg + n_bg)] (+ _fine): sigma at the foreground points followed by the background points of each row;
 *     mean exp(k relu(s)) / mean 1 - exp(k relu(s)), k = -2 / n_samples (train.py:262-286);
 *   quads[quad_rows*4] (+ _fine): tangent-mode sigma (sigma, d/dx, d/dy, d/dz) of normal_sets x (nv points, nv perturbed
 *     neighbours); normal = delta exp(-delta sigma) grad sigma where sigma > 0 (models/nerf.py:177-190), unit length with
 *     eps 1e-5, MSE between a point's and its neighbour's (train.py:288-309).
 * vals_out[12] = loss_rgb, loss_rgb_fine, loss_alphas, loss_alphas_fine, loss_foreground, loss_background,
 *   loss_foreground_fine, loss_background_fine, loss_normals, loss_normals_fine, total (weighted by the lambdas), and the
 *   batch's PSNR = 10 log10((max target_rgb - min target_rgb)^2 / loss_rgb_fine, or loss_rgb without a fine pass): what
 *   train.py:339-344 logs as train/psnr — torchmetrics' peak_signal_noise_ratio called without data_range.
 * workspace[anr_train_loss_ws_floats()], zero before its first use (the kernel leaves it ready for the next call). */
typedef struct {
    const float *rgb, *acc, *rgb_fine, *acc_fine, *target_rgb, *target_alpha;
    const float *s, *s_fine;
    const float *quads, *quads_fine;
    int64_t R, prior_rows, nv, normal_sets, quad_rows;
    int32_t n_fg, n_bg;
    float k, delta, lambda_alphas, lambda_foreground, lambda_background, lambda_normals;
    int32_t s_stride;   /* floats between consecutive prior sigmas in s / s_fine: 0 or 1 = packed; 4 = column 3 of (r, g, b, sigma)
                           rows (s points at row 0's sigma), and then the gradient d.s / d.s_fine (pointing at row 0's r) is written
                           as whole rows (0, 0, 0, d sigma) */
    const int32_t *s_count, *s_count_fine;   /* may be NULL.  Otherwise DEVICE counts: s / s_fine point at row 0 of a compacted pass's
                           output (its sigma with s_stride 4) and the prior points are the LAST prior_rows (n_fg + n_bg) of its first
                           s_count[0] rows (anr_compact_ordered_riders) — read where the network left them; the gradient rows d.s /
                           d.s_fine are NOT offset (they belong to the expanded layout anr_mlp_head_grad gathers from) ... */
    int32_t s_grad_rows;   /* ... unless this is 1 (with s_count): d.s / d.s_fine point at row 0 of the g operand of the network's
                           backward (rows of the compacted pass, anr_composite_backward_compact) and the prior points' rows are
                           written there, (0, 0, 0, d sigma) each */
    int32_t quad_grad_rows;   /* 1: d.quads / d.quads_fine are the g operand of the tangent-mode backward, [4 quad_rows][4]: row 4 p + q
                           = (0, 0, 0, dL/d quads[p][q]) — what anr_mlp_head_grad (sigma only) makes of the quad gradients */
} anr_loss_args;
/* gradient destinations, same shapes as the inputs (NULL: not wanted); quads: all quad_rows rows are written */
typedef struct {
    float *rgb, *acc, *rgb_fine, *acc_fine, *s, *s_fine, *quads, *quads_fine;
} anr_loss_grads;
int64_t anr_train_loss_ws_floats(void);
int anr_train_loss(const anr_loss_args* args, float* workspace, float* vals_out, void* stream);
/* g_total: DEVICE pointer to dL/d total (autograd's upstream scalar) */
int anr_train_loss_backward(const anr_loss_args* args, const float* g_total, const anr_loss_grads* grads, void* stream);

/* ---- sigma-grid points for mesh extraction -------------------------------------------------------
 * extract_mesh.py:27-35 (create_grid: np.meshgrid(x, y, z), 'xy' indexing, fp64 linspace -> fp32) and :152-157
 * (+ bounding-box centre of the posed vertices).  Flat index n = (j*N + i)*N + k -> (x[i], y[j], z[k]).
 * Writes points [first, first+count) of the N^3 grid as (x, y, z, 1); center[3] is a device pointer.
 * A rank of a voxel-sharded job asks for its own [first, first+count) slab. */
int anr_grid_points(int N, double x0, double x1, double y0, double y1, double z0, double z1,
                    const float* center, int64_t first, int64_t count, float* pts_out, void* stream);
/* The same points for a list of 8 x 8 x 8-voxel CELLS (N % 8 == 0; cell id = (cj (N/8) + ci) (N/8) + ck covers array
 * indices j in [8 cj, 8 cj + 8) and likewise i, k): pts_out[n_cells*512][4] and the voxels' flat grid indices
 * vox_out[n_cells*512] — for evaluating the field only in cells that can hold a valid voxel; anr_scatter_relu then writes
 * out[vox[t] - first] = max(values[t], 0) for the voxels inside [first, first + count) (extract_mesh.py:49-61: relu(sigma)). */
int anr_grid_points_cells(int N, double x0, double x1, double y0, double y1, double z0, double z1, const float* center,
                          const int32_t* cells, int64_t n_cells, float* pts_out, int32_t* vox_out, void* stream);
int anr_scatter_relu(const float* values, const int32_t* vox, int64_t n, int64_t first, int64_t count, float* out, void* stream);

/* ---- a13: alpha compositing ----------------------------------------------------------------------
 * models/volume_rendering.py:131-160 (far=True, delta_last = 1e10).
 * rgbs[R*K*4] = (r,g,b,sigma) per sample, z[R*K], rays[R*stride] (far' = rays[..,7]).
 * noise (may be NULL): added to sigma before the relu (:128-129, caller supplies randn*noise_std).
 * weights_out[R*K] (may be NULL), rgb_out[R*3], depth_out[R], acc_out[R]. */
int anr_composite(const float* rgbs, const float* z, const float* rays, int stride,
                  const float* noise, int64_t R, int K, int white_bkgd,
                  float* weights_out, float* rgb_out, float* depth_out, float* acc_out,
                  void* stream);
/* ... with the warp's per-sample validity bytes valid[R*K] (anr_warp_points): a sample with valid == 0 is taken as
 * (0, 0, 0, -1e5) — what the reference gives it — and its row of rgbs is not read (it need not be initialised). */
int anr_composite_masked(const float* rgbs, const float* z, const float* rays, int stride,
                         const float* noise, const uint8_t* valid, int64_t R, int K, int white_bkgd,
                         float* weights_out, float* rgb_out, float* depth_out, float* acc_out,
                         void* stream);

/* ... on the COMPACTED output of a training pass: rows[n_valid*4] = the (r, g, b, sigma) rows of the valid samples only, as
 * the network pass over anr_compact_ordered's list left them; pos[R*K] = a sample's row, or -1 for a sample the warp found
 * invalid — which counts as (0, 0, 0, -1e5), models/anim_nerf.py:245-290.  Same values as anr_composite on the expanded rows
 * (anr_expand_rows), without that launch. */
int anr_composite_indexed(const float* rows, const int32_t* pos, const float* z, const float* rays, int stride, const float* noise,
                          int64_t R, int K, int white_bkgd, float* weights_out, float* rgb_out, float* depth_out, float* acc_out,
                          void* stream);

/* ---- a16 (part): backward of anr_composite -------------------------------------------------------------
 * What autograd differentiates in models/volume_rendering.py:131-160: upstream gradients of the per-ray outputs
 * g_rgb[R*3], g_depth[R], g_acc[R] (each may be NULL = zero; and optionally of the weights, g_weights[R*K] or NULL) ->
 * d_rgbs[R*K*4] = dL/d(r, g, b, sigma) per sample; optionally also dL/dz (through the interval lengths and the
 * depth output) and dL/dfar' (white-background depth term), which pose refinement needs.
 * Same inputs as the forward (nothing is kept by the library). */
int anr_composite_backward(const float* rgbs, const float* z, const float* rays, int stride, const float* noise,
                           int64_t R, int K, int white_bkgd, const float* g_weights, const float* g_rgb,
                           const float* g_depth, const float* g_acc, float* d_rgbs,
                           float* d_z /* [R*K] or NULL */, float* d_far /* [R] or NULL */, void* stream);
/* ... with the inputs of anr_composite_indexed (pos may be NULL: anr_composite_backward); d_rgbs stays one row per sample */
/* ... and straight to the g operand of the network's backward: g4_rows_out[row pos[sample]] = (dL/d rgb . sigmoid'(rgb),
 * dL/d sigma) for the valid samples — what anr_mlp_head_grad makes of d_rgbs through the list — nothing per sample; the
 * padding rows [count[0], count[1]) (anr_compact_ordered's device counts) are zeroed.  The prior points' rows are the loss
 * kernel's (anr_loss_args.s_grad_rows). */
int anr_composite_backward_compact(const float* rows, const int32_t* pos, const int32_t* count, const float* z, const float* rays,
                                   int stride, const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights,
                                   const float* g_rgb, const float* g_depth, const float* g_acc, float* g4_rows_out, float* d_z,
                                   float* d_far, void* stream);
int anr_composite_backward_indexed(const float* rows, const int32_t* pos, const float* z, const float* rays, int stride,
                                   const float* noise, int64_t R, int K, int white_bkgd, const float* g_weights, const float* g_rgb,
                                   const float* g_depth, const float* g_acc, float* d_rgbs, float* d_z, float* d_far, void* stream);

/* ---- a14: importance sampling + merge ---------------------------------------------------------------
 * models/volume_rendering.py:59-97 and :199-207: inverse-CDF samples over the Kc-1 mid-points
 * with weights[1:-1]+1e-5, then sort(cat(z_coarse, z_fine)).
 * u[Kf] if u_per_ray == 0 (deterministic linspace, shared), else u[R*Kf] (caller's uniforms).
 * z_fine_out[R*Kf] (may be NULL), z_sorted_out[R*(Kc+Kf)], perm_out[R*(Kc+Kf)] (may be NULL):
 * z_sorted[s] = cat(z_coarse, z_fine)[perm[s]] — autograd routes gradients of sorted depths back to z_coarse. */
int anr_sample_fine_merge(const float* z_coarse, const float* weights, const float* u,
                          int u_per_ray, int64_t R, int Kc, int Kf,
                          float* z_fine_out, float* z_sorted_out, int32_t* perm_out, void* stream);
/* ... with the permutation as bytes (Kc + Kf <= 256): what anr_warp_points_lean's reuse_perm takes. */
int anr_sample_fine_merge_u8(const float* z_coarse, const float* weights, const float* u, int u_per_ray,
                             int64_t R, int Kc, int Kf,
                             float* z_fine_out, float* z_sorted_out, uint8_t* perm_out, void* stream);


/* ---- a13 + a14 fused: the coarse pass of an inference render ----------------------------------------------------
 * models/volume_rendering.py:172-178 (composite of the Kc coarse samples) followed by :199-207 (importance samples from
 * its weights, cat, sort) in ONE launch: the weights and the coarse depths stay in the wavefront / LDS instead of making
 * an HBM round trip between anr_composite and anr_sample_fine_merge (bit-identical outputs to that pair).
 * z_coarse[R*Kc], or NULL: then the coarse depths are the deterministic stratified ones of :43-44,
 *   z_k = near' (1 - steps[k]) + far' steps[k]  (steps[Kc] = linspace(0, 1 - 1/Kc, Kc), near'/far' = rays[.., 6/7]),
 *   computed in the kernel with the roundings of anr_sample_coarse (no z array exists at all).
 * valid[R*Kc] (may be NULL) as in anr_composite_masked.  u as in anr_sample_fine_merge.
 * weights_out[R*Kc], z_fine_out[R*Kf], perm_out[R*(Kc+Kf)] may be NULL. */
int anr_composite_sample(const float* rgbs, const float* z_coarse, const float* steps, const float* rays, int stride,
                         const uint8_t* valid, const float* u, int u_per_ray, int64_t R, int Kc, int Kf,
                         int white_bkgd, float* weights_out, float* rgb_out, float* depth_out, float* acc_out,
                         float* z_fine_out, float* z_sorted_out, uint8_t* perm_out, void* stream);

/* ---- optimiser step (f1: train.py:216-226, `torch.optim.Adam(lr, eps=1e-8)` over the networks + the SMPL rows) ---------------
 * Adam over every parameter tensor of the step in ONE launch.  `chunks` is a device array of n_chunks records of
 * anr_adam_chunk_bytes() bytes each: {float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32 count, int32
 * group_and_tensor}, count <= anr_adam_chunk_floats() consecutive floats of one tensor (a tensor = ceil(numel / chunk) records);
 * group_and_tensor = group | tensor << 8: group < n_groups <= 4 selects lr[group] (HOST array), tensor indexes step[] (DEVICE
 * floats): the 1-based count t of THIS update of that tensor — torch keeps one counter per tensor; the caller increments them on
 * the device before the launch, so a captured step replays with the right bias corrections.
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * (torch.optim.Adam with amsgrad = maximize = False, weight_decay = 0; 1 - beta and the bias corrections are formed in double,
 * as torch's host-side arithmetic does, then used in float). */
int anr_adam_chunk_floats(void);
int anr_adam_chunk_bytes(void);
int anr_adam_step(const void* chunks, int n_chunks, const float* step, const float* lr, int n_groups, double beta1, double beta2,
                  double eps, void* stream);
/* ... with the counters advanced by the kernel itself: step[n_tensors] holds the counts BEFORE this update (the update uses
 * step + 1), and the last workgroup to finish adds active[n_tensors] (1 where the tensor has a gradient this step) to them;
 * ticket: one int32 on the device, zero at rest.  One launch instead of an increment launch + the update. */
int anr_adam_step_counting(const void* chunks, int n_chunks, float* step, const float* active, int n_tensors, int32_t* ticket,
                           const float* lr, int n_groups, double beta1, double beta2, double eps, void* stream);

/* ---- the explicit training step (anim_nerf_amd/fused_step.py): what it needs beside the kernels above so that the step's
 * HIP graph holds this library's launches only (csrc/train_step.hip) -------------------------------------------------
 * anr_train_draws: every random number of one training step (train.py:324-348) in one launch — Philox4x32-10 keyed by
 *   state[0] = seed and state[1] = a step counter that lives ON THE DEVICE and is advanced by a one-thread launch behind
 *   the draws (state[2..34]: reserved, zero), so a replayed graph draws fresh numbers.  state: int64[ANR_DRAW_STATE_WORDS = 35],
 *   device.
 *   t_rand[n_t] = t_scale U[0,1) (the stratified jitter, models/volume_rendering.py:48-54: t_scale = perturb);
 *   noise_c[n_nc], noise_f[n_nf] = noise_scale N(0,1) (sigma noise of the two passes, :122-129); u_fine[n_u] = U[0,1) (the
 *   importance sampler's uniforms, :66-70); n0[n_v3], n1[n_v3] = N(0,1) and pair[2 n_v3] = (verts_template + point_scale n0,
 *   that + neighbour_scale n1): the normal regulariser's points and their neighbours (train.py:289-290: point_scale =
 *   dis_threshold / 2, neighbour_scale = epsilon).  A segment with n = 0 is skipped. */
typedef struct {
    float *t_rand, *noise_c, *u_fine, *noise_f;
    int64_t n_t, n_nc, n_u, n_nf;
    float t_scale, noise_scale;
    const float* verts_template;
    int64_t n_v3;
    float point_scale, neighbour_scale;
    float *n0, *n1, *pair;
    float* quads;   /* may be NULL.  Otherwise [>= 4 (2 n_v3 / 3)][4]: the `pair` points as the tangent-mode operand of the network pass
                       — four rows (x, y, z, 1) per point, what anr_tangent_quads makes of `pair` — written by the same launch (rows
                       behind the last point are the caller's to zero) */
} anr_draw_plan;
#define ANR_DRAW_STATE_WORDS 35
int anr_train_draws(int64_t* state, const anr_draw_plan* plan, void* stream);
/* BodyModelParams.forward (models/body_model_params.py:5-68): rows frame_idx[bs] (int64) of the embedding tables
 * global_orient[T][3], body_pose[T][69], transl[T][3] and betas[betas_rows][10] (betas_rows = 1: one shape for all frames) ->
 * betas_out[bs][10], pose_out[bs][72] = (global_orient | body_pose), transl_out[bs][3]: the operands of anr_smpl_forward. */
int anr_gather_frame_params(const int64_t* frame_idx, int bs, const float* betas_w, int betas_rows, const float* go_w,
                            const float* bp_w, const float* tr_w, float* betas_out, float* pose_out, float* transl_out,
                            void* stream);
/* ... and its backward: grads[bs][85] = dL/d(betas 10 | global_orient 3 | body_pose 69 | transl 3) per frame (what
 * anr_frame_backward_adjoint returns) -> the tables' gradients, WRITTEN whole (rows no frame of the batch uses get 0; a row
 * used twice gets the sum in batch order; betas_rows = 1: the sum over the batch).  A NULL table is skipped. */
int anr_scatter_frame_param_grads(const int64_t* frame_idx, const float* grads, int bs, int table_rows, int betas_rows,
                                  float* d_betas_w, float* d_go_w, float* d_bp_w, float* d_tr_w, void* stream);
/* anr_merge_backward on g_a + g_b (g_b may be NULL): the fine pass's warp and compositor both differentiate the sorted depths;
 * perm: the byte permutation of anr_sample_fine_merge_u8 (K <= 256) */
int anr_merge_backward2(const float* g_a, const float* g_b, const uint8_t* perm, int64_t R, int K, int Kc,
                        float* d_z_coarse_out, void* stream);
/* anr_sample_coarse_backward on g_a + g_b + g_c (g_b, g_c may be NULL), ADDED into columns 6, 7 of d_rays_acc[R*8] together
 * with dfar_a[R] + dfar_b[R] (the two compositors' dL/d far', NULL = none): the ray gradient the warp's backward has been
 * accumulating is complete after this launch. */
int anr_sample_coarse_backward_acc(const float* g_a, const float* g_b, const float* g_c, const float* steps,
                                   const float* t_rand, const float* dfar_a, const float* dfar_b, int64_t R, int K,
                                   float* d_rays_acc, void* stream);
/* anr_to_root_frame with the root transforms g_stride floats apart (g_stride = 16 J: joints_transform[:, 0] read in place) */
int anr_to_root_frame_strided(const float* global_transform, int64_t g_stride, const float* verts, const float* joints,
                              const float* T, int bs, int V, int J, float* g_inv_out, float* g_root_out, float* verts_out,
                              float* joints_out, float* T_out, void* stream);
/* The per-frame set-up of a batch of frames in two launches (csrc/frame_setup.hip): what AnimNeRF.set_body_model,
 * convert_to_body_model_space and clac_ober2cano_transform compute (models/anim_nerf.py:108-151 over smplx/body_models.py:289-387
 * and smplx/lbs.py:152-404), i.e. anr_gather_frame_params + anr_smpl_forward + anr_to_root_frame + anr_rays_to_body +
 * anr_ober2cano with the rest joints taken as J0 + JS . betas (J0[24*3] = J_regressor v_template, JS[24*3*10] = J_regressor
 * shapedirs).  frame_idx != NULL: betas_w / global_orient_w / body_pose_w / transl_w are the BodyModelParams tables
 * (models/body_model_params.py) and frame b reads row frame_idx[b] (betas: row min(., betas_rows - 1)); NULL: per-frame arrays.
 * template_*: the template pose's state, one body (template_bs = 1) or one per frame.  R = 0: no rays.
 * Outputs: the frames' parameters betas[bs*10], pose[bs*72], transl[bs*3]; A = joints_transform[bs*24*16] (transl on the
 * translation column); joints, verts and vertices_transform IN THE ROOT FRAME; g_inv, g_root[bs*16]; shape / pose offsets;
 * ober2cano[bs*V*16]; rays_body[bs*R*8].  ws_feat[bs*207] fp32. */
int anr_frame_setup(const int64_t* frame_idx, const float* betas_w, int betas_rows, const float* global_orient_w,
                    const float* body_pose_w, const float* transl_w, int bs, const float* J0, const float* JS,
                    const int64_t* parents, const float* v_template, const float* shapedirs, const float* posedirs,
                    const float* lbs_weights, int V, int J, int NB, const float* T_template,
                    const float* shape_off_template, const float* pose_off_template, int template_bs,
                    const float* rays_world, int ray_stride, int R, float* betas_out, float* pose_out, float* transl_out,
                    float* A_out, float* joints_root_out, float* g_inv_out, float* g_root_out, float* shape_off_out,
                    float* pose_off_out, float* verts_root_out, float* T_root_out, float* ober2cano_out,
                    float* rays_body_out, float* ws_feat, void* stream);
/* ... with the pose tables' row count: frame_idx[b] outside [0, table_rows) — where the reference's nn.Embedding raises
 * (models/body_model_params.py:5-68) — reads row 0 and makes that frame's parameters, hence its loss and gradients, NaN:
 * no out-of-bounds read, no host synchronisation.  table_rows = 0: not checked (= anr_frame_setup). */
int anr_frame_setup_rows(const int64_t* frame_idx, int table_rows, const float* betas_w, int betas_rows,
                         const float* global_orient_w, const float* body_pose_w, const float* transl_w, int bs, const float* J0,
                         const float* JS, const int64_t* parents, const float* v_template, const float* shapedirs,
                         const float* posedirs, const float* lbs_weights, int V, int J, int NB, const float* T_template,
                         const float* shape_off_template, const float* pose_off_template, int template_bs,
                         const float* rays_world, int ray_stride, int R, float* betas_out, float* pose_out, float* transl_out,
                         float* A_out, float* joints_root_out, float* g_inv_out, float* g_root_out, float* shape_off_out,
                         float* pose_off_out, float* verts_root_out, float* T_root_out, float* ober2cano_out,
                         float* rays_body_out, float* ws_feat, void* stream);
/* The one-pass ray-march kernel (csrc/ray_march.hip; models/volume_rendering.py:163-232 with a model without the warp,
 * use_unpose=False — BASELINE configs[1]): per ray, in ONE launch: Kc stratified coarse depths (perturb = 0, :29-56), x = o + z d,
 * Fourier encoding + the coarse network (models/nerf.py:129-175; pack_coarse / pack_fine from anr_mlp_pack, `mode` as there),
 * compositing (:113-160), Kf importance samples from the coarse weights (u[Kf] = linspace(0, 1, Kf), :59-97) merged in depth
 * order (:199-207), the fine network on the Kc + Kf sorted samples, compositing.  Nothing per sample goes through HBM: 32 B in
 * and 2 x 20 B out per ray.  Returns the bits of anr_mlp_forward_rays_steps + anr_composite_sample + anr_mlp_forward_rays +
 * anr_composite.  Built for Kc = Kf = 64 (ANR_E_SHAPE otherwise).  rays[R*ray_stride], steps[Kc] (the table of
 * anr_sample_coarse), outputs rgb[R*3], depth[R], acc[R] for the coarse and the fine pass. */
int anr_ray_march(const void* pack_coarse, const void* pack_fine, int mode, const float* rays, int ray_stride, int64_t R,
                  const float* steps, int Kc, const float* u, int Kf, int white_bkgd, float* rgb_coarse, float* depth_coarse,
                  float* acc_coarse, float* rgb_fine, float* depth_fine, float* acc_fine, void* stream);
/* ... WITH the inverse-LBS / exact 4-nearest-vertex warp inside the pass (models/anim_nerf.py:153-192; use_unpose=True — BASELINE
 * configs[2]): every sample is warped where it is generated — the staged classify pass's tests (within dis_threshold of the body's
 * box, the reach mask), then the exact search and the blend of anr_warp_points on the index read from global memory — encoded, and
 * the networks run on EVERY sample, sigma = -1e5 where the sample is not valid (:305): the reference's own dense evaluation in one
 * launch.  rays[bs*rays_per_body*ray_stride] in the bodies' root frames (anr_rays_to_body / anr_frame_setup), knn_index from
 * anr_knn_index_build[_reach] (bs bodies), ober2cano[bs*V*16], lbs_weights[V*J].  Returns the bits of the staged path (whose
 * sparse evaluation — the networks on the valid samples of the whole frame, compacted — is several times faster: this entry
 * point is the single-pass form of the north star, not the renderer's default). */
int anr_ray_march_warp(const void* pack_coarse, const void* pack_fine, int mode, const float* rays, int ray_stride, int bs,
                       int64_t rays_per_body, const float* steps, int Kc, const float* u, int Kf, int white_bkgd,
                       const void* knn_index, const float* ober2cano, const float* lbs_weights, int V, int J, float dis_threshold,
                       float* rgb_coarse, float* depth_coarse, float* acc_coarse, float* rgb_fine, float* depth_fine,
                       float* acc_fine, void* stream);
/* zero `bytes` (a multiple of 4) at a 4-byte aligned device address: a kernel, not a memset (a memset NODE of a captured HIP
 * graph went stale on ROCm 7.2: DESIGN.md section 4.4) */
int anr_zero_fill(void* ptr, int64_t bytes, void* stream);
/* dst[0..n) += src[0..n) (fp32): the weight gradients of the normals-regulariser branch of the explicit training step, computed
 * on a second stream next to the render passes, joining the network's flat gradient buffer — what autograd's accumulation of
 * models/nerf.py:177-190's and the render passes' contributions into one .grad does (train.py:324-348) */
int anr_add_inplace(float* dst, const float* src, int64_t n, void* stream);
/* ... for n_segments <= 24 (dst, src, floats) triples in one launch (both networks' share of the regulariser) */
int anr_add_segments(float* const* dst, const float* const* src, const int64_t* floats, int n_segments, void* stream);
/* n (<= 24) device-to-device copies dst[i][0..bytes[i]) = src[i][...] in ONE launch; src / dst / bytes are HOST arrays (the table
 * travels in the kernel arguments).  The batch of a training step (train.py:324-331: rays, rgbs, alphas, the pose rows, the
 * prior points) moving into the fixed buffers a captured step replays from — one launch instead of one copy per tensor. */
int anr_copy_segments(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream);
/* memset(dst[i], 0, bytes[i]) for n <= 24 device buffers in ONE launch (a step's accumulators and counters, all filled up front). */
int anr_zero_segments(void* const* dst, const int64_t* bytes, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANIMNERF_HIP_H */
