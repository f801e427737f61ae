/*
 * gens_hip.h -- C ABI of libgens_hip.so: the GenS hot path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * Drop-in boundary.  The reference's only native ABI is the pybind pair
 *     grad2_2d / grad2_3d                (models/modules/grid_sample_cuda/gridsample_cuda.cpp:26-56)
 * reached through cuda_gridsample.grid_sample_3d (cuda_gridsample.py:12-14, 71-123); everything else on the
 * path is PyTorch tensor code.  Each entry point below names the reference code it replaces.  All citations are
 * relative to /root/reference.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous float32 unless its comment says otherwise (host arrays of
 *     device pointers are spelled `const float* const*` and documented as host);
 *   - inputs are borrowed and never written; outputs are caller-allocated; outputs documented "accumulates" must
 *     be zero-filled by the caller (they are atomically added to);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work, they never
 *     synchronise;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative GENS_E* argument error.
 *     gens_last_error() returns a static description of the last failure on the calling thread;
 *   - volumes are (C, X, Y, Z) with world x <-> X ("planar", the reference's (1,C,D,D,D) tensor, Q1) or
 *     (X, Y, Z, 4) ("packed": one 16-byte texel per voxel, built by gens_pack_volume);
 *   - feature maps / images consumed by the samplers are NHWC with the channel count padded to a multiple of 4
 *     ("texel" layout, built by gens_pack_nchw); C_pad below always means 4*ceil(C/4);
 *   - only C = 4 channels per volume level are supported (every shipped config: confs/gens.conf:63-67).
 */
#ifndef GENS_HIP_H
#define GENS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENS_MAX_LEVELS 8
#define GENS_MAX_VIEWS 16  /* the reference's fine-tune sets hold up to 11 views (dtu_finetune.py: ref + 10 sources); K1's culled fast path serves <= 8, more views take its generic kernel */

#define GENS_EINVAL (-1)   /* bad argument (null pointer, size out of range)          */
#define GENS_ELIMIT (-2)   /* more levels / views / channels than the build supports */

#define GENS_LAYOUT_PLANAR 0
#define GENS_LAYOUT_PACKED 1

const char* gens_last_error(void);
/* 12.  History: 7 = 6 + gens_sdf_grad_f16 (the split-half value + gradient kernel).
 *   8 = round 4's additions, which shipped under the stale number 7: gens_grid_sample_{fwd,bwd,bwd2} (K20), gens_depthwise_conv2d_{fwd,dgrad,wgrad}
 *       + gens_depthwise_conv2d_wgrad_parts (K21), gens_batchnorm2d_train_{fwd,bwd} + gens_batchnorm2d_scratch_doubles (K22),
 *       gens_blend_train_bwd_acc + gens_blend_train_acc_{parts,floats}, gens_merge_upsample, gens_conv3d_wgrad_parts_strided, gens_instnorm_finish,
 *       gens_sdf_grad_stash_reset, gens_sdf_grad_f16_stash_reset, and gens_ray_points' `mid` / `sample_dist` arguments.
 *   9 = round 5: gens_composite_in gained `cos_anneal_dev` (the annealing ratio read from the device, so that a captured step can be replayed
 *       with another ratio) -- a struct-layout change: callers built against 8 must be rebuilt.
 *   10 = round 5: gens_sdf_train_bwd's w6_part has a row per SIXTEEN points (npad / 16 rows, was npad / 32): its workgroups own 16 points
 *       now, two of them to a compute unit.
 *   11 = round 6: gens_blend_train_bwd_t + gens_blend_train_t_parts + gens_blend_train_bwd_t_dump (the colour branch's backward transposed: one
 *       wave per 16 rows, nothing shared between waves but the weights in LDS).
 *   12 = round 6: gens_upsample2d_cat (the warp features in one launch), gens_volume_build_levels_bits (the volume build leaves the masks as bits
 *       too), gens_select_views (the views of a fine-tune step out of the frozen maps and their layouts in one launch),
 *       gens_lookup_volume_bwd_bricks / _bwd2_bricks + gens_lookup_scatter_bricks_scratch_bytes (K2's volume-gradient scatter brick by brick). */
int gens_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Layout helpers (no reference counterpart: the reference keeps NCHW / NCDHW everywhere).
 * ---------------------------------------------------------------------------------------------------------- */

/* src (n, C, H, W) -> dst (n, H, W, C_pad), pad channels are zero. */
int gens_pack_nchw(const float* src, float* dst, int n, int c, int h, int w, void* stream);
/* adjoint: dst (n, C, H, W) = channels 0..C-1 of src (n, H, W, C_pad)   (overwrites dst). */
int gens_unpack_nhwc(const float* src, float* dst, int n, int c, int h, int w, void* stream);
/* src (4, X, Y, Z) -> dst (X, Y, Z, 4). */
int gens_pack_volume(const float* src, float* dst, int x, int y, int z, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K1  Volume.agg_mean_var, one level per call          (models/modules/volume.py:21-61)
 *   feat    (nv, H, W, 4) texels of features[level]
 *   w2c     (nv, 4, 4) = inverse(c2ws)                   (volume.py:34)
 *   intr    (nv, 4, 4) level-0 intrinsics; rows 0-1 are multiplied by intr_scale = 0.5^level in-kernel (:24-25)
 *   volume  (8, D, D, D) = [mean(4) | var(4)], mask (D, D, D) float 0/1 = (count > min_vis_view)   (:53-58)
 * bwd: g_volume (8, D, D, D) -> g_feat (nv, H, W, 4), accumulates (gradient w.r.t. features only, :27-44).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_volume_build_fwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv, int h,
                          int w, int d, int min_vis_view, float* volume, float* mask, void* stream);
int gens_volume_build_bwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv, int h,
                          int w, int d, const float* g_volume, float* g_feat, void* stream);
/* All levels of one scene in a single launch (volume.py:21-61 is a loop over the levels): feat[l] (nv, H_l, W_l, 4) texels,
 * hw = {H_0, W_0, H_1, W_1, ...}, intr[l] (nv, 4, 4) with rows 0-1 already multiplied by 0.5^l, volumes[l] (8, D_l^3), masks[l] (D_l^3).
 * Same results as n_levels calls of gens_volume_build_fwd with intr_scale = 1 (which it falls back to for sizes the fused kernel
 * does not cover).  counts: NULL, or a HOST array of device pointers (an entry may be NULL) to (D_l^3) uint8 planes, 16-byte aligned,
 * that receive the number of views each voxel is visible in (volume.py:50 `count`) -- what gens_volume_build_bwd_levels reads back. */
int gens_volume_build_levels(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                             const float* const* intr, int nv, int min_vis_view, float* const* volumes, float* const* masks,
                             uint8_t* const* counts, void* stream);
/* The same launch, leaving the masks as BITS too (ABI 12): mask_bits: NULL, or a HOST array of device pointers (an entry may be NULL) to
 * ceil(D_l^3 / 32) uint32 words, bit (i & 31) of word (i >> 5) = masks[l][i] > 0 -- what gens_pack_mask_bits makes of the float plane and the
 * ray-point / nearest look-up kernels read with mask_bits = 1; a training step then has no packing pass over the masks. */
int gens_volume_build_levels_bits(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                  const float* const* intr, int nv, int min_vis_view, float* const* volumes, float* const* masks,
                                  uint8_t* const* counts, uint32_t* const* mask_bits, void* stream);
/* d(volumes)/d(texels) of ALL levels in one launch set (five launches whatever n_levels is) with the sum owned by the IMAGE: the (64-voxel tile,
 * view) pairs are sorted by the 64 x 60-texel image tile they project into, a workgroup per tile keeps that tile's gradient in LDS (double
 * sums) and writes every touched texel once (gens_volume_build_bwd, the wave-window kernel, is bound by its ~0.3 G global atomics at 256^3).
 * It reads what the forward pass left: the means (volumes[l], planes 0-3) and the visible-view counts (counts[l]) of gens_volume_build_levels
 * on the same inputs; nothing is re-derived per voxel except the projection into the one view a work item serves.  g_volumes[l] (8, D_l^3)
 * or NULL (no gradient for that level), g_feat[l] (nv, H_l, W_l, 4) accumulates.  Every D_l must be a multiple of 16; intrinsics pre-scaled
 * per level as for gens_volume_build_levels.  scratch: gens_volume_build_bwd_levels_scratch_bytes bytes of device memory (20 bytes per
 * (64-voxel tile, view) pair + the bins; 0 = sizes not covered: use gens_volume_build_bwd level by level), contents irrelevant before and
 * after.  Results equal gens_volume_build_bwd's up to the order of the float32 sums. */
int64_t gens_volume_build_bwd_levels_scratch_bytes(const int* hw, const int* dims, int n_levels, int nv);
int gens_volume_build_bwd_levels(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                 const float* const* intr, int nv, const float* const* volumes, const uint8_t* const* counts,
                                 const float* const* g_volumes, float* const* g_feat, void* scratch, int64_t scratch_bytes, void* stream);
/* Self-test of K1's exact-division shortcuts (RN(1/b) from v_rcp_f32 + one FMA refinement; a/b from that reciprocal + FMA
 * correction) against the IEEE division over all 2^32 float32 bit patterns: counts[0] += reciprocal mismatches,
 * counts[1] += quotient mismatches (device array of 2, zeroed by the caller).  Both stay 0. */
int gens_selftest_division(unsigned long long* counts, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K2  lookup_volume(pts, volumes, "grad"): all levels in one launch
 *     forward            projector.py:217-245 -> cuda_gridsample._GridSample3dForward (cuda_gridsample.py:71-84)
 *     backward           aten::grid_sampler_3d_backward via _GridSample3dBackward.forward (:94-108)
 *     backward-backward  gridsample_grad2.grad2_3d (gridsample_cuda.cpp:42-56, gridsample_cuda.cu:212-533, 601-666)
 *   vols     HOST array of n_levels device pointers; dims HOST int[3*n_levels] = (X,Y,Z) per level
 *   pts      (N, 3) world points in [-1,1] (no flip: x indexes X)
 *   out / g_out / gg_out   (N, 4*n_levels), level-major like torch.cat(feats, -1)
 *   g_vols / g_vols2       HOST arrays of device pointers (same layout as vols), accumulate; NULL = not wanted
 *   gg_vols                HOST array or NULL (the reference's `grad2_grad_input is None`, cuda_gridsample.py:113)
 *   g_pts / gg_pts / g_pts2  (N, 3); g_pts, g_pts2 are overwritten
 * ---------------------------------------------------------------------------------------------------------- */
int gens_lookup_volume_fwd(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                           int64_t n, float* out, void* stream);
int gens_lookup_volume_bwd(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                           const float* g_out, int64_t n, float* const* g_vols, float* g_pts, void* stream);
int gens_lookup_volume_bwd2(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                            const float* g_out, const float* gg_pts, const float* const* gg_vols, int64_t n,
                            float* gg_out, float* const* g_vols2, float* g_pts2, void* stream);
/* The same two with the scatter into the volume gradients done BRICK BY BRICK (ABI 12): the points are counted into bricks of 8^3 cells of the finest
 * level, a workgroup per brick sums its points' corner contributions in LDS (double sums) and sends every touched voxel to memory once -- for large
 * point sets whatever their order (the direct scatter of the entries above sends 32 atomics per point and level).  scratch:
 * gens_lookup_scatter_bricks_scratch_bytes(n) bytes of device memory, 16-byte aligned.  Same results up to the order of the float sums.  Volumes
 * wider than 256 cells per axis fall back to the direct scatter. */
int64_t gens_lookup_scatter_bricks_scratch_bytes(int64_t n);
int gens_lookup_volume_bwd_bricks(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                  const float* g_out, int64_t n, float* const* g_vols, float* g_pts, void* scratch,
                                  int64_t scratch_bytes, void* stream);
int gens_lookup_volume_bwd2_bricks(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                   const float* g_out, const float* gg_pts, const float* const* gg_vols, int64_t n,
                                   float* gg_out, float* const* g_vols2, float* g_pts2, void* scratch, int64_t scratch_bytes,
                                   void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K3  lookup_volume(pts, mask_volumes, "nearest").any(-1)        (projector.py:231,240; Q6, Q7)
 *   masks  HOST array of device pointers to (X,Y,Z) float masks; valid (N) uint8; vals (N, n_levels) or NULL.
 * gens_ray_points fuses the point generation of render / render_core (implicit_surface.py:160-174, 367-371):
 *   z (B, n); mid != 0 -> samples at z + 0.5*dist with the last dist = sample_dist (Q10)
 *   pts (B*n, 3), valid (B*n) uint8.
 * gens_pack_mask_bits: (n) float mask -> ceil(n/32) uint32 words, bit i%32 of word i/32 = (mask[i] > 0) -- the 256^3
 *   level becomes 2 MB and stays in L2.  gens_ray_points / gens_upsample read such words when mask_bits != 0
 *   (masks[] then point to the words); the decisions are identical to the float masks.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_lookup_mask_nearest(const float* const* masks, const int* dims, int n_levels, const float* pts, int64_t n,
                             uint8_t* valid, float* vals, void* stream);
int gens_ray_points(const float* rays_o, const float* rays_d, const float* z, int64_t n_rays, int n_samples, int mid,
                    float sample_dist, const float* const* masks, const int* dims, int n_levels, int mask_bits,
                    float* pts, uint8_t* valid, void* stream);
int gens_pack_mask_bits(const float* mask, int64_t n, uint32_t* bits, void* stream);

/* Order-preserving compaction of the valid flags written by gens_ray_points / gens_upsample: idx[0..count) = positions of
 * the non-zero flags in increasing order; if no flag is set, count = min(10, n) and idx = 0..count-1 (Q7,
 * implicit_surface.py:123-124,176-177,372-373).  idx (n) int64, count (1) int32, scratch (ceil(n/1024)+1) int32: DEVICE. */
int gens_compact_valid(const uint8_t* valid, int64_t n, int64_t* idx, int32_t* count, int32_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K4  lookup_feature + compute_angle                              (projector.py:278-349)
 *   feats   HOST array of n_levels device pointers to (nv, H_l, W_l, 4) texels, hw HOST int[2*n_levels]
 *   imgs    (nv, H_0, W_0, 4) texels of the RGB images (channel 3 = pad)
 *   w2c, intr, c2w  (nv, 4, 4); view 0 is the reference view, sources are views 1..nv-1
 *   out (N, S, 3 + 4*n_levels), ray_diff (N, S, 4), vis (N, S) uint8       S = nv - 1
 * bwd: g_out -> g_feats[l] (nv, H_l, W_l, 4), g_imgs (nv, H_0, W_0, 4); accumulate; either may be NULL.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_lookup_feature_fwd(const float* const* feats, const int* hw, int n_levels, const float* imgs,
                            const float* w2c, const float* intr, const float* c2w, int nv, const float* pts,
                            int64_t n, float* out, float* ray_diff, uint8_t* vis, void* stream);
int gens_lookup_feature_bwd(const int* hw, int n_levels, const float* w2c, const float* intr, int nv,
                            const float* pts, const float* g_out, int64_t n, float* const* g_feats, float* g_imgs,
                            void* stream);
/* ... with g_out COMPACT (row i = the i-th selected point, as gens_blend_train_bwd writes it) while the point itself is pts[index[i]];
 * only min(n, *n_device) rows exist. */
int gens_lookup_feature_bwd_idx(const int* hw, int n_levels, const float* w2c, const float* intr, int nv,
                                const float* pts, const float* g_out, const int64_t* index, int64_t n, const int32_t* n_device,
                                float* const* g_feats, float* g_imgs, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K5 + K6  ImplicitSurface.up_sample + sample_pdf(det=True)      (implicit_surface.py:14-44, 60-109)
 *   one wavefront per ray; z, sdf (B, n) with n <= 128; inv_s = 64 * 2^round; n_new <= 64
 *   z_new (B, n_new); pts_new (B*n_new, 3) and valid_new (B*n_new) uint8 feed the next SDF evaluation (:117-121).
 *   valid_in (B, n) uint8 or NULL: the mask decisions of the existing samples as written by gens_ray_points /
 *   gens_upsample and carried through gens_merge_samples (the reference looks all n of them up again every round, :66-67;
 *   the values are the same because the samples are); NULL = look them up here.
 * K7  cat_z_vals: sorted merge of (z, sdf) with (z_new, sdf_new)  (:111-133); sdf / sdf_new / sdf_out may be NULL
 *   together (the `last` round); valid / valid_new / valid_out (uint8) likewise.  n + n_new <= 128.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t n_rays, int n,
                  int n_new, float inv_s, const float* const* masks, const int* dims, int n_levels, int mask_bits,
                  const uint8_t* valid_in, float* z_new, float* pts_new, uint8_t* valid_new, void* stream);
int gens_merge_samples(const float* z, const float* sdf, const float* z_new, const float* sdf_new, const uint8_t* valid,
                       const uint8_t* valid_new, int64_t n_rays, int n, int n_new, float* z_out, float* sdf_out,
                       uint8_t* valid_out, void* stream);
/* One launch per sampling round: cat_z_vals of round i (implicit_surface.py:111-133) fused with up_sample + sample_pdf of round i + 1 (:60-109,
 * mode 0) or, after the last round, with render_core's section mid-points and their mask decisions (:163-173, mode 1 = gens_ray_points(mid)).
 * (z, sdf, valid) (n_rays, n) + (z_add, sdf_add, valid_add) (n_rays, n_add) -> merged (n_rays, n + n_add) in z_out / sdf_out / valid_out;
 * mode 0: z_new (n_rays, n_new), pts_out (n_rays n_new, 3), valid_new (n_rays n_new);  mode 1: pts_out (n_rays (n + n_add), 3), valid_new likewise
 * (sdf, valid, sdf_add, valid_add, sdf_out, valid_out, z_new may be NULL).  Results equal gens_merge_samples followed by gens_upsample (mode 0,
 * with the merged mask decisions as valid_in) / by gens_ray_points with mid = 1 (mode 1), bit for bit. */
int gens_merge_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const uint8_t* valid,
                        const float* z_add, const float* sdf_add, const uint8_t* valid_add, int64_t n_rays, int n, int n_add, int n_new,
                        float inv_s, float sample_dist, const float* const* masks, const int* dims, int n_levels, int mask_bits, int mode,
                        float* z_out, float* sdf_out, uint8_t* valid_out, float* z_new, float* pts_out, uint8_t* valid_new, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K8  render_core compositing                                     (implicit_surface.py:160-168, 202-303)
 *   per-sample inputs (B, n[, 3]); n <= 128; voxel_mask (B*n) uint8; src_vis (B*n, S) uint8 or NULL
 *   smooth may be NULL (inference).  rot = inverse(c2ws[0,:3,:3]) (3,3) HOST floats passed by value as 9 floats, or rot_dev: the same
 *   nine floats in DEVICE memory (gens_scene_setup's rot_inv; NULL = use rot) -- no device-to-host read per scene
 *   inv_s: DEVICE pointer to one float (already clipped, :206); z_max: DEVICE pointer to max(z) (:301)
 *   per-ray outputs: color (B,3) normal (B,3) depth (B) wsum (B) wmax (B) valid (B) uint8 mid_in (B)
 *                    sdf_depth (B) z_cross (B) [clamped, :300-302] cross_idx (B) int32
 *                    eik_num (B) eik_den (B) smooth_vec (B,3)
 *   per-sample outputs: weights (B,n) inside (B,n)
 * bwd: cotangents (NULL = zero) g_color (B,3) g_normal (B,3) g_depth (B) g_weights (B,n) g_wsum (B) g_eik_num (B)
 *      g_smooth_vec (B,3) g_z_cross (B)  ->  g_sdf (B,n) g_grad (B,n,3) g_col (B,n,3) g_smooth (B,n,3 or NULL)
 *      g_inv_s (B) per-ray partials (sum them).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *rays_o, *rays_d, *z, *sdf, *grad, *color, *smooth;
    const uint8_t *voxel_mask, *src_vis;
    const float *inv_s, *z_max;
    int64_t n_rays;
    int n, n_src;
    float sample_dist, cos_anneal;
    float rot[9];
    const float* rot_dev;
    const float* cos_anneal_dev;   /* NULL, or a device float that REPLACES cos_anneal (read at launch: a HIP graph of the step can be replayed
                                    * with this step's ratio, runner.py:156-157,394-398) */
} gens_composite_in;

typedef struct {
    float *color, *normal, *depth, *wsum, *wmax, *mid_in, *sdf_depth, *z_cross, *eik_num, *eik_den, *smooth_vec;
    uint8_t* valid;
    int32_t* cross_idx;
    float *weights, *inside;
    float* pts_cross;          /* (B, 3) or NULL: rays_o + rays_d * z_cross, the surface point pts_sdf0 (implicit_surface.py:304) */
} gens_composite_out;

typedef struct {
    const float *g_color, *g_normal, *g_depth, *g_weights, *g_wsum, *g_eik_num, *g_smooth_vec, *g_z_cross;
    const float* weights;      /* forward output */
    const int32_t* cross_idx;  /* forward output */
    const float* smooth_vec;   /* forward output */
    float *g_sdf, *g_grad, *g_col, *g_smooth, *g_inv_s;
    /* cotangents of the per-batch scalars of gens_composite_finish_fwd (DEVICE scalars or NULL) and its `finish` output */
    const float *g_gradient_error, *g_smooth_error, *finish;
} gens_composite_grad;

int gens_composite_fwd(const gens_composite_in* in, const gens_composite_out* out, void* stream);
int gens_composite_bwd(const gens_composite_in* in, const gens_composite_grad* g, void* stream);
/* The per-batch reductions around the compositing of a training step, one workgroup each (implicit_surface.py:248-253, 206):
 *   finish_fwd: finish (4) = {sum eik_num, sum eik_den, gradient_error = sum eik_num / (sum eik_den + 1e-5), smooth_error = mean |smooth_vec|}
 *     (smooth_vec may be NULL).  Their cotangents go into gens_composite_grad.g_gradient_error / g_smooth_error.
 *   finish_bwd: g_variance (1) = 10 inv_s [inside the clip range] sum_r g_inv_s[r] with scalars = gens_compact_points' {z_max, inv_s,
 *     1 / inv_s, in range}; rows [n_ray_pts, n_all) of the dense gradient arrays g_sdf (n_all), g_grad / g_smooth (n_all, 3) are zeroed
 *     (any of the four outputs may be NULL). */
int gens_composite_finish_fwd(const float* eik_num, const float* eik_den, const float* smooth_vec, int64_t n_rays, float* finish, void* stream);
int gens_composite_finish_bwd(const float* g_inv_s, int64_t n_rays, const float* scalars, float* g_variance, float* g_sdf, float* g_grad,
                              float* g_smooth, int64_t n_ray_pts, int64_t n_all, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K6  SDFNetwork.forward(...)[:, :1] and its first derivative, inference only   (sdf_network.py:98-146)
 *   One launch = volume look-up (K2, packed volumes) + both positional encodings + 7 layers on the fp32 matrix
 *   cores (+ reverse-mode d sdf/d x when grad_out != NULL).  Architecture: d_hidden 128, n_layers 6, skip_in [3],
 *   multires 4, feat_multires 2, Softplus(beta=100); n_levels must be 3 or 5 (feat_channels 12 / 20).
 *   wf / wb: HOST arrays of 6 device pointers: forward / transposed weights as grouped MFMA B streams
 *   [n_tile][ceil(K/8)][64 lanes][4] (lane l of group g holds W[32 n_tile + (l & 31)][8 g + 4 (l >> 5) + 0..3]), built by
 *   gens_amd.ops.SdfMlpPlan.  The bias of layer l is row K_l of the forward stream (K_0 = 27, else 128 + 20 n_levels):
 *   the kernel feeds a constant-1 input in that (padding) column.  w_last (128 + 5*4*n_levels) = row 0 of lin6.
 *   index: optional (N) int64 gather/scatter map: point i is pts[index[i]] and results go to sdf_out[index[i]],
 *   grad_out[3*index[i]..] (the masked evaluation of implicit_surface.py:125,179-191); NULL = identity.
 *   n_device: optional DEVICE int32: the number of points actually evaluated is min(n, *n_device) (written by
 *   gens_compact_valid), so the masked evaluation needs no host synchronisation; NULL = n.
 *   w_last_scaled: NULL, or the PRE-SCALED convention of the forward streams (saves the two multiplications of every softplus):
 *   with c = 100 / ln 2, wf[l] then holds W_l with the columns fed by unscaled inputs (point encoding incl. layer 3's 27 skip columns,
 *   volume-feature columns) and the bias row multiplied by c, the columns fed by hidden units as they are; w_last_scaled = w_last with
 *   its first 128 entries divided by c.  wb / w_last (reverse pass) are always the plain weights.  gens_amd.ops.SdfMlpPlan builds both.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_sdf_mlp(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                 const float* const* wb, const float* w_last, const float* w_last_scaled, float b_last, float scale,
                 const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                 void* stream);

/* gens_sdf_mlp with the output bias read from DEVICE memory (training: the weights change every step and the streams come from
 * gens_sdf_train_pack; a by-value bias would cost a device-to-host read per step). */
int gens_sdf_mlp_dev(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                     const float* const* wb, const float* w_last, const float* b_last_dev, float scale,
                     const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                     void* stream);

/* The VALUE of the same network (grad_out == NULL of gens_sdf_mlp), same exact float32 MFMA arithmetic, laid out for the value-only
 * passes (k6t_sdf_value.hip): the hierarchical sampling of implicit_surface.py:125,329-352 and the lattice of :407-427.  A wavefront owns
 * 32 points and all 128 hidden units; the activated accumulators are the next layer's matrix operands (the weights are packed in the
 * order the accumulators come out in), so nothing goes through LDS and there are no barriers.
 *   wstream: DEVICE, 16-byte aligned, (gens_sdf_value_groups(n_levels) + 1) x 4 KB: per group of four feature pairs
 *   [4 output tiles][64 lanes][4 floats] in the pair order of gens_amd.ops._value_pairs, pre-scaled as gens_sdf_mlp's w_last_scaled
 *   convention; the trailing group is zero (read ahead, never used).
 *   w_out: DEVICE (2, 64 + 4 * GC) float32: row 0 of lin6 in the accumulator / slot order of each lane half (hidden part / (100/ln 2)). */
int gens_sdf_value(const float* const* vols_packed, const int* dims, int n_levels, const float* wstream, const float* w_out,
                   float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                   float* sdf_out, void* stream);
/* number of 4 KB groups in the weight stream of gens_sdf_value, without the trailing zero group (125 for 3 levels, 0 = unsupported) */
int gens_sdf_value_groups(int n_levels);

/* Value and gradient (gens_sdf_mlp with grad_out) in the transposed dataflow of gens_sdf_value (k6g_sdf_grad.hip): one wavefront per 32
 * points runs the forward chain and the reverse chain G_{l-1} = (W_l^T G_l) * softplus' entirely in registers (softplus' of layers 0 and 1
 * waits in LDS); the gradients of the volume features and of the point encoding accumulate in tiles whose rows are ordered per lane, so
 * the chain rule to x is lane-local.  Three or five volume levels.
 *   wstream: DEVICE, 16-byte aligned, (gens_sdf_grad_groups(n_levels) + 2) x 4 KB in the order of gens_amd.ops._pack_grad_stream: the
 *   forward groups of gens_sdf_value, then the reverse pass on the plain transposed matrices; the two trailing groups are zero
 *   (the kernel requests weights two groups ahead).
 *   w_out: DEVICE (2, 64 + 16 * TC) float32: row 0 of lin6 per lane half (hidden part / (100 / ln 2), then the conditioning slots).
 *   stash: DEVICE, 16-byte aligned, gens_sdf_grad_stash_bytes() bytes, ZEROED ONCE by the caller and then left alone (it may be shared by all
 *   calls on a device, concurrent ones included): a 16 KB slot + lock word per SIMD where softplus' of layer 2 waits for the reverse pass. */
int gens_sdf_grad(const float* const* vols_packed, const int* dims, int n_levels, const float* wstream, const float* w_out,
                  float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                  float* sdf_out, float* grad_out, void* stash, void* stream);
int64_t gens_sdf_grad_stash_bytes(void);
/* Re-zero the stash (its lock words) on `stream`: after a launch that died holding a slot -- a wave that finds its slot taken waits a
 * bounded time (seconds) and then traps, so that the host sees a failed launch instead of a hang. */
int gens_sdf_grad_stash_reset(void* stash, void* stream);
/* number of 4 KB groups in the weight stream of gens_sdf_grad, without the trailing zero groups (0 = unsupported level count) */
int gens_sdf_grad_groups(int n_levels);

/* The VALUE of the same network (no gradient) on the f16 matrix cores with SPLIT operands -- every float32 operand an (hi, lo) pair of halfs,
 * every product hi*hi + hi*lo + lo*hi with float32 accumulation (~1e-6 relative error); overflow_flag: DEVICE int, OR-ed with 1 when an
 * activation or volume feature leaves the half range (the caller then redoes the batch with gens_sdf_value) -- laid out for throughput
 * (k6v_sdf_value_f16.hip): the 512^3 lattice of extract_geometry (implicit_surface.py:407-427) and the value-only passes of the opt-in
 * "f16x2" arithmetic.  A wavefront owns 32 points and all 128 hidden units; activations stay in registers between the layers because
 * the weights are packed in the order the accumulators come out in; one weight stream per 128 points goes through LDS.
 *   units: DEVICE, 16-byte aligned, gens_sdf_value_f16_units(n_levels) x 8192 bytes: per 16-deep K block of the six layers
 *   [4 feature tiles][hi, lo][64 lanes][8 halfs] in the slot order of gens_amd.ops._value_slots, pre-scaled by 100 / ln 2 on the
 *   columns fed by unscaled inputs, zero padded to whole chunks of four blocks.
 *   w_out: DEVICE (2, 64 + 8 * NC) float32: row 0 of lin6 in the accumulator / slot order of each lane half. */
int gens_sdf_value_f16(const float* const* vols_packed, const int* dims, int n_levels, const void* units, const float* w_out,
                       float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                       float* sdf_out, int* overflow_flag, void* stream);
/* number of 8 KB K blocks in the weight stream of gens_sdf_value_f16 (64 for 3 levels, 80 for 5; 0 = unsupported level count) */
int gens_sdf_value_f16_units(int n_levels);

/* Value AND gradient with SPLIT operands on the f16 matrix cores (k6gh_sdf_grad_f16.hip): gens_sdf_grad's dataflow and outputs,
 * gens_sdf_value_f16's arithmetic contract (hi*hi + hi*lo + lo*hi, float32 accumulation, softplus' kept in float32) -- the value +
 * gradient pass of render_core (implicit_surface.py:179-191 -> sdf_network.py:98-154) under the opt-in "f16x2" arithmetic.  Three or
 * five volume levels (BASELINE config[1] / confs/gens.conf:63-67); GENS_ELIMIT otherwise.
 *   pieces: DEVICE, 16-byte aligned, gens_sdf_grad_f16_pieces(n_levels) x 1024 bytes in the order of gens_amd.ops._pack_grad_pieces:
 *   [64 lanes][8 halfs] per (K block, output tile, hi | lo).
 *   w_out: the output row of gens_sdf_grad (same layout).   g_scale: a power of two the gradients travel multiplied by (lo parts stay
 *   normal halfs); the result is divided by it.
 *   stash: DEVICE, 16-byte aligned, gens_sdf_grad_f16_stash_bytes() bytes, ZEROED ONCE by the caller and then left alone.
 *   overflow_flag: DEVICE int, OR-ed with 1 when an input, a hidden unit or a gradient leaves the half range or is not a number: the
 *   caller redoes the batch with gens_sdf_grad. */
int gens_sdf_grad_f16(const float* const* vols_packed, const int* dims, int n_levels, const void* pieces, const float* w_out,
                      float b_last, float scale, float g_scale, const float* pts, const int64_t* index, int64_t n,
                      const int32_t* n_device, float* sdf_out, float* grad_out, void* stash, int* overflow_flag, void* stream);
int64_t gens_sdf_grad_f16_stash_bytes(void);
int gens_sdf_grad_f16_stash_reset(void* stash, void* stream);      /* as gens_sdf_grad_stash_reset */
/* number of 1 KB pieces in the weight stream of gens_sdf_grad_f16, padding included (0 = unsupported level count) */
int gens_sdf_grad_f16_pieces(int n_levels);

/* ------------------------------------------------------------------------------------------------------------
 * K21  depth-wise 2-D convolutions of the MnasNet trunk (k21_depthwise.hip): nn.Conv2d(c, c, k, padding = k / 2, stride, groups = c,
 *      bias = False) with k in {3, 5}, stride in {1, 2} -- torchvision's MNASNet layers used by feature_network_mnasnet.py:53-103 -- for which
 *      MIOpen falls back to its naive kernels on gfx950.  NCHW float32, contiguous.
 *   in (n, c, h, w), weight (c, 1, k, k), out / grad_out (n, c, oh, ow) with oh = (h + 2 (k / 2) - k) / stride + 1 (ow likewise).
 *   wgrad: partial (gens_depthwise_conv2d_wgrad_parts(...), c, k, k) receives per-slice sums; the weight gradient is their sum over axis 0
 *   in slice order (deterministic).  Other kernel sizes / strides: GENS_ELIMIT (the caller keeps its own convolution for those).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_depthwise_conv2d_fwd(const float* in, const float* weight, int n, int c, int h, int w, int k, int stride, float* out, void* stream);
int gens_depthwise_conv2d_dgrad(const float* grad_out, const float* weight, int n, int c, int h, int w, int k, int stride, float* grad_in,
                                void* stream);
int gens_depthwise_conv2d_wgrad_parts(int n, int c, int h, int w, int k, int stride);
int gens_depthwise_conv2d_wgrad(const float* in, const float* grad_out, int n, int c, int h, int w, int k, int stride, float* partial,
                                void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K22  BatchNorm2d in training mode (batch statistics) [+ the ReLU that follows it] of the MnasNet trunk (k22_batchnorm.hip): the
 *      nn.BatchNorm2d(c, momentum = 1 - 0.9997) [+ nn.ReLU] pairs of torchvision's MNASNet used by feature_network_mnasnet.py:53-103.
 *      NCHW float32 contiguous, x / y / grad_* (n, c, hw).  gamma / beta (c) or NULL (no affine).
 *   fwd: y = [max](((x - mean_c) * rstd_c) * gamma_c + beta_c[, 0]) with the batch's mean and BIASED variance; mean_rstd (c, 2) receives
 *        (mean, rstd) for the backward pass; running_mean / running_var (c) or NULL are moved by `momentum` (running_var with the unbiased
 *        variance), num_batches_tracked (DEVICE int64 or NULL) is incremented -- nn.BatchNorm2d's bookkeeping.
 *   bwd: grad_x (or NULL), grad_gamma, grad_beta (c, or NULL) from grad_y; relu: the decision y > 0 is recomputed from x.
 *   scratch: DEVICE, gens_batchnorm2d_scratch_doubles(n, c, hw) doubles, contents irrelevant before and after.
 * ---------------------------------------------------------------------------------------------------------- */
int64_t gens_batchnorm2d_scratch_doubles(int n, int c, int hw);
int gens_batchnorm2d_train_fwd(const float* x, const float* gamma, const float* beta, int n, int c, int hw, float eps, float momentum, int relu,
                               float* y, float* mean_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               double* scratch, void* stream);
int gens_batchnorm2d_train_bwd(const float* x, const float* grad_y, const float* mean_rstd, const float* gamma, const float* beta, int n, int c,
                               int hw, int relu, float* grad_x, float* grad_gamma, float* grad_beta, double* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K20  the reference's sampler boundary in full generality (k20_grid_sample.hip): what cuda_gridsample.py:7-14 exports and K2 does not
 *      cover -- grid_sample_2d, padding_mode 'border', align_corners=False, batches, channel counts that are not multiples of four.
 *     forward            F.grid_sample(bilinear) = aten::grid_sampler_2d / _3d        (cuda_gridsample.py:28, 79)
 *     backward           aten::grid_sampler_2d_backward / _3d_backward                 (cuda_gridsample.py:38-50, 94-108)
 *     backward-backward  gridsample_grad2.grad2_2d / grad2_3d                          (gridsample_cuda.cpp:26-56, gridsample_cuda.cu:27-533)
 *   ndim      2 or 3;  in_size HOST int[ndim]: the input's spatial shape, slowest axis first ((H, W) or (D, H, W))
 *   input     (n, c, *in_size) contiguous float32 (the reference's NCHW / NCDHW);  grid (n, n_out, ndim), last axis (x, y[, z]) -> (W, H[, D])
 *   out / grad_out / gg_out  (n, c, n_out);  grad_grid / gg_grid (n, n_out, ndim)
 *   padding_mode  0 = 'zeros', 1 = 'border' (the two the reference asserts, cuda_gridsample.py:8,13; it passes the index, :32,83)
 *   grad_input    like input, ACCUMULATES (float atomics): zero it first, as the reference does (gridsample_cuda.cu:553-555, 620-622); NULL = not
 *                 wanted (bwd) / the reference's unused output (bwd2).  gg_input: like input or NULL (`grad2_grad_input is None`, :113-114).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_grid_sample_fwd(const float* input, const float* grid, int ndim, int n, int c, const int* in_size, int64_t n_out,
                         int padding_mode, int align_corners, float* out, void* stream);
int gens_grid_sample_bwd(const float* grad_out, const float* input, const float* grid, int ndim, int n, int c, const int* in_size,
                         int64_t n_out, int padding_mode, int align_corners, float* grad_input, float* grad_grid, void* stream);
int gens_grid_sample_bwd2(const float* gg_input, const float* gg_grid, const float* grad_out, const float* input, const float* grid,
                          int ndim, int n, int c, const int* in_size, int64_t n_out, int padding_mode, int align_corners,
                          float* gg_out, float* grad_input, float* grad_grid, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K17  the SDF network of a training / fine-tune step: value, gradient, `smooth` vector and the loss backward
 *      (sdf_network.py:98-154: SDFNetwork.sdf, SDFNetwork.gradient with create_graph twice; their autograd backward under
 *      loss.backward(); call sites implicit_surface.py:179-191,257,305,490; sampler Function pair cuda_gridsample.py:71-123,
 *      whose second-backward outputs are constants in the loss backward: third order through the sampler is dropped)
 *   Architecture and n_levels as gens_sdf_mlp, scale = 1 (confs/gens.conf:78).  All buffers are caller-allocated device memory.
 *   gens_sdf_train_pack: w[0..5] / b[0..5]: HOST arrays of device pointers to the EFFECTIVE (weight norm applied) row-major
 *     matrices (128 x 27, 128 x K, 101 x K, 128 x K x 3; K = 128 + 20 n_levels) and biases of lin0..lin5; writes the forward /
 *     transposed B streams wf[l] ([4][ceil((K_l + 1) / 8)][64] float4, bias in reduction row K_l) and wb[l] ([ceil(K_l / 32)][16][64]
 *     float4) that the other entry points (and gens_sdf_mlp) read.
 *   index / n / n_device (all three entry points) as in gens_sdf_mlp: point i of the launch is pts[index[i]] and its results (cotangents)
 *     live at row index[i] of the dense arrays; only min(n, *n_device) points exist -- the masked evaluation of
 *     implicit_surface.py:174-191 with the count left on the device (gens_compact_points).  NULL index = identity, NULL n_device = n.
 *   gens_sdf_train_fwd: pts -> y (N), g (N, 3) = dy/dx, s (N, 3) = d(sum_k g_k)/dx, written at index[i].  w_last: row 0 of lin6 (K
 *     floats), b_last: DEVICE pointer to its bias.  stash: scratch of gens_sdf_train_stash_bytes(n, 0) bytes.
 *   gens_sdf_train_bwd: cotangents y_bar (N), g_bar (N, 3), s_bar (N, 3) (NULL = zero; read at index[i]) -> operand rows of the
 *     weight-gradient products, POINT-major: npad = 32 ceil(n / 32) points x 4 sweeps, the rows of the live points being the
 *     contiguous range [0, 4 * 32 ceil(n_live / 32)) (rows of padding points inside it contribute zero, rows beyond it are NOT written:
 *     pass the count to gens_gemm_tn_batch_live):
 *       lop (npad, 4, 6, 128), rh (5, npad, 4, 128), re (npad, 4, KP - 128), r0 (npad, 4, 32), KP = 8 ceil((K + 1) / 8):
 *       dL/dW_l[:, :128]  = lop[:, :, l, :]^T rh[l - 1]                  (l = 1..5; layer 3's columns are [h_2 | pe] / sqrt 2)
 *       dL/dW_l[:, 128:K] , dL/db_l = lop[:, :, l, :]^T re               (column K - 128 of re is the bias input)
 *       dL/dW_0, dL/db_0  = lop[:, :, 0, :]^T r0                         (column 27 = bias input)
 *       dL/dw_last, dL/db_last = column sums of w6_part (npad / 16, KP): columns [0, K) and column K (zero rows for dead workgroups)
 *     (gens_gemm_tn_batch runs the products in one launch) and f_hat, mu_f, lam_f (npad, 4 n_levels) for
 *     gens_sdf_train_scatter.  stash: gens_sdf_train_stash_bytes(n, 1) bytes.
 *   gens_sdf_train_scatter: adds dL/dvolume into g_vols[l] (planar (4, X, Y, Z), pre-zeroed or accumulating):
 *       w f_hat + (grad w . s_bar) mu_f + (grad w . g_bar) lam_f per corner  (float atomics).
 * ---------------------------------------------------------------------------------------------------------- */
/*   gens_sdf_train_pack_wn: gens_sdf_train_pack from the RAW weight-normed parameters (nn.utils.weight_norm, sdf_network.py:90-91:
 *     W = g v / |v| per output row): v / g / b = HOST arrays of the 7 device pointers weight_v (rows_l, cols_l), weight_g (rows_l), bias
 *     (rows_l) of lin0..lin6; scale: 7 caller-allocated arrays of rows_l floats (g / |v|, kept for the backward); also writes the effective
 *     output row w_last (K floats) and its bias b_last (1).  Two launches.
 *   gens_sdf_train_wgrad: the parameter gradients of a step in one launch -- cc = gens_gemm_tn_batch's result for the eleven products in
 *     the order (l = 1..5: [lop_l^T rh_l (128 x 128) | lop_l^T re (128 x KP - 128)], then lop_0^T r0 (128 x 32)), w6_sum (KP) = the column
 *     sums of w6_part -> dv[l] (rows_l, cols_l), dg[l] (rows_l), db[l] (rows_l) for lin0..lin6, weight norm's backward included. */
int gens_sdf_train_pack_wn(const float* const* v, const float* const* g, const float* const* b, int n_levels, float* const* scale,
                           float* const* wf, float* const* wb, float* w_last, float* b_last, void* stream);
int gens_sdf_train_wgrad(const float* const* v, const float* const* g, int n_levels, const float* cc, const float* w6_sum,
                         float* const* dv, float* const* dg, float* const* db, void* stream);
int64_t gens_sdf_train_stash_bytes(int64_t n, int backward);
int gens_sdf_train_pack(const float* const* w, const float* const* b, int n_levels, float* const* wf, float* const* wb, void* stream);
int gens_sdf_train_fwd(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                       const float* const* wb, const float* w_last, const float* b_last, const float* pts, const int64_t* index,
                       int64_t n, const int32_t* n_device, void* stash, float* y_out, float* g_out, float* s_out, void* stream);
int gens_sdf_train_bwd(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                       const float* const* wb, const float* w_last, const float* pts, const int64_t* index, int64_t n,
                       const int32_t* n_device, const float* y_bar, const float* g_bar, const float* s_bar, void* stash, float* lop,
                       float* rh, float* re, float* r0, float* f_hat, float* mu_f, float* lam_f, float* w6_part, void* stream);
int gens_sdf_train_scatter(const int* dims, int n_levels, const float* pts, const float* g_bar, const float* s_bar,
                           const float* f_hat, const float* mu_f, const float* lam_f, const int64_t* index, int64_t n,
                           const int32_t* n_device, float* const* g_vols, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K7  lookup_feature + BlendingNetwork.forward fused, inference only
 *     (projector.py:278-349 followed by blending_network.py:69-118, as called from implicit_surface.py:196-199)
 *   feats / hw / imgs / w2c / intr / c2w / nv as in gens_lookup_feature_fwd (n_levels <= 5).
 *   weights: HOST array of 21 device pointers in the order
 *     ray_dir_fc.0 W,b  ray_dir_fc.2 W,b  base_fc.0 W,b  base_fc.2 W,b  vis_fc.0 W,b  vis_fc.2 W[0:32],b[0:32]  vis_fc.2 W[32]
 *     vis_fc2.0 W,b  vis_fc2.2 W  rgb_fc.0 W,b  rgb_fc.2 W,b  rgb_fc.4 W
 *   (matrices as grouped MFMA B streams -- the layout documented at gens_sdf_mlp, K padded to a multiple of 8 --
 *   biases zero-padded to a multiple of 32; gens_amd.ops.BlendPlan builds them);
 *   scalars: HOST float[4] = { vis_fc.2 bias[32], vis_fc2.2 bias, rgb_fc.4 bias, |s| }.
 *   index as in gens_sdf_mlp.  rgb_out (N_total, 3); vis_out (N_total, S) uint8 or NULL, written at index[i].
 * ---------------------------------------------------------------------------------------------------------- */
int gens_blend_views(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c,
                     const float* intr, const float* c2w, int nv, const float* const* weights, const float* scalars,
                     const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out,
                     void* stream);

/* gens_blend_views for TWO, THREE or FOUR source views (nv = 3, 4, 5: the view counts of the shipped configurations -- num_src_view = 2
 * in the test protocol and the fine-tune configs, confs/gens.conf:24, confs/gens_finetune.conf:15; 4 in training, confs/gens.conf:9) in
 * the transposed dataflow of gens_sdf_value (k7t_blend.hip): one wavefront owns 64 (point, view) rows (three views: 48 + 16 dead rows),
 * the weights are the A operand of v_mfma_f32_16x16x4_f32, the activations of the eleven layers stay in registers in "quad layout", the
 * mean / variance columns of base_fc.0 are multiplied once per point.
 * Same inputs and outputs as gens_blend_views except the weights (the same stream for every view count):
 *   wstream: DEVICE, 16-byte aligned, (gens_blend_views_t_groups(n_levels) + 2) x 1 KB: A fragments in consumption order
 *   (gens_amd.ops._pack_blend_t), the two trailing groups zero;  tab: DEVICE (10, 4, 8) float32: accumulator-layout biases and the
 *   three single-output rows per lane group;  scalars: HOST float[4] as gens_blend_views.
 * gens_blend_views4 / gens_blend_views4_groups: the ABI-4 names (nv = 5 only). */
int gens_blend_views_t(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                       const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars, const float* pts,
                       const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream);
int gens_blend_views_t_groups(int n_levels);
/* gens_blend_pack_t: wstream / tab / the four scalars of gens_blend_views_t from the 23 RAW nn.Linear parameters of a BlendingNetwork
 * (HOST array of device pointers in the order of gens_blend_train_fwd), in one launch; scalars_dev: DEVICE float[4].
 * gens_blend_views_t_dev: gens_blend_views_t with those scalars read from device memory -- the forward pass of a TRAINING step, whose
 * colour network changes every step (gens_blend_train_bwd recomputes what it needs). */
int gens_blend_pack_t(const float* const* weights, int n_levels, float* wstream, float* tab, float* scalars_dev, void* stream);
int gens_blend_views_t_dev(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                           const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars_dev, const float* pts,
                           const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream);
int gens_blend_views4(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                      const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars, const float* pts,
                      const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream);
int gens_blend_views4_groups(int n_levels);

/* ------------------------------------------------------------------------------------------------------------
 * K18  lookup_feature + BlendingNetwork.forward of a training / fine-tune step, and their backward
 *      (projector.py:278-349, blending_network.py:69-118 as called from implicit_surface.py:196-199; first order only)
 *   feats / hw / imgs / w2c / intr / c2w / nv as in gens_lookup_feature_fwd (n_levels <= 5).
 *   weights: HOST array of 23 device pointers to the RAW nn.Linear parameters (row major (out, in)) in the order
 *     ray_dir_fc.0 W,b  ray_dir_fc.2 W,b  base_fc.0 W,b  base_fc.2 W,b  vis_fc.0 W,b  vis_fc.2 W,b  vis_fc2.0 W,b  vis_fc2.2 W,b
 *     rgb_fc.0 W,b  rgb_fc.2 W,b  rgb_fc.4 W,b  s
 *   index / n / n_device as in gens_sdf_mlp: point i of the launch is pts[index[i]], its colour / flags / cotangent live at row index[i]
 *     of the dense arrays, only min(n, *n_device) points exist (NULL index = identity, NULL n_device = n).
 *   fwd: pts -> rgb_out (N, 3), vis_out (N, S) uint8 (NULL to skip), written at index[i].
 *   bwd: g_rgb (N, 3) cotangent of rgb_out -> for each of the 11 layers the operand rows of its weight-gradient product over
 *     rows = gens_blend_train_rows(n, nv) = 32 ceil(n / floor(32 / S)) (under a device-side count only the rows of the first
 *     ceil(n_live / floor(32 / S)) workgroups are written: gens_gemm_tn_batch_live):  r_ops[l] (rows, even(in_l + 1)) = [layer input | 1 | 0],
 *     l_ops[l] (rows, even(out_l)) = cotangent of the pre-activation (zero padded; even(x) = x rounded up to a multiple of 2):
 *     [dW_l | db_l] = the leading out_l x (in_l + 1) block of l_ops[l]^T r_ops[l]   (gens_gemm_tn_batch);
 *     g_feat (n, S, 3 + 4 n_levels), COMPACT (row i = point i of the launch): cotangent of the looked-up rows for
 *     gens_lookup_feature_bwd[_idx] (NULL to skip);  s_part (rows / 32): partial sums of d loss / d |s| (zero for dead workgroups).
 * ---------------------------------------------------------------------------------------------------------- */
int64_t gens_blend_train_rows(int64_t n, int nv);
int gens_blend_train_fwd(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                         const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                         const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream);
int gens_blend_train_bwd(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                         const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                         const int32_t* n_device, const float* g_rgb, float* const* r_ops, float* const* l_ops, float* g_feat,
                         float* s_part, void* stream);
/* The same backward with the weight-gradient sums INSIDE the launch (round 4): persistent workgroups -- gens_blend_train_acc_parts(n, nv) of them --
 * multiply L^T [R | 1] of their row tiles out of LDS on the matrix cores, the sums in registers, and leave one block of
 * gens_blend_train_acc_floats(n_levels) floats each in `parts` (parts x floats); `cc` (floats) = their sum in part order = the eleven blocks
 * even(out_l) x even(in_l + 1), concatenated, that gens_gemm_tn_batch returns for l_ops / r_ops above (the input of gens_blend_train_wgrad).  No
 * operand rows: 619 MB per launch neither written nor read again. */
int gens_blend_train_acc_parts(int64_t n, int nv);
int gens_blend_train_acc_floats(int n_levels);
int gens_blend_train_bwd_acc(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                             const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                             const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc, void* stream);
/* The same backward TRANSPOSED (round 6; two to four source views, nv = 3 .. 5 -- other counts: gens_blend_train_bwd_acc): a wave owns 16 (point,
 * view) rows and all of their forward, reverse and weight-gradient work -- the weights of the eleven layers in LDS as the A operand of
 * v_mfma_f32_16x16x4_f32, the activations in registers from layer to layer and, in [channel][row] order, in a wave-private LDS store the weight-
 * gradient products read both operands from; four independent waves per workgroup, one persistent workgroup per compute unit, no barrier behind the
 * weight load.  gens_blend_train_t_parts(n, nv) = the number of workgroups = blocks of gens_blend_train_acc_floats(n_levels) floats in `parts` AND
 * entries of s_part (one partial of d loss / d |s| per workgroup; 0 = nothing to launch or a view count this kernel is not built for); cc, g_feat as in
 * gens_blend_train_bwd_acc.  Results equal gens_blend_train_bwd_acc's up to float32 summation order.
 * gens_blend_train_bwd_t_dump additionally leaves the operand rows r_ops / l_ops of gens_blend_train_bwd (same widths) for
 * rows = 16 ceil(n / (16 / G)) rows, G = 2 (nv = 3) or 4 lanes per point: row 16 tile + G point + view (three source views: every fourth row is
 * empty) -- the two kernels compared layer by layer (tests). */
int gens_blend_train_t_parts(int64_t n, int nv);
int gens_blend_train_bwd_t(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                           const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                           const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc, void* stream);
int gens_blend_train_bwd_t_dump(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc,
                                float* const* r_ops, float* const* l_ops, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K9  the two F.grid_sample(align_corners=True) reads of surface_patch_warp   (projector.py:406-416)
 *   image (H, W, C_pad) texels of one view; xy (P, 2) PIXEL coordinates (the normalise/un-normalise pair of
 *   :404-405 and align_corners=True cancel); out (P, C).  bwd: g_out (P, C) -> g_xy (P, 2), overwritten.
 *   gens_upsample2d_into restates F.interpolate(mode="bilinear") (implicit_surface.py:316-325) writing channels
 *   [c_off, c_off+C) of a (n, H, W, C_pad_dst) texel tensor from src (n, C, hs, ws) NCHW.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_patch_sample_fwd(const float* image, int h, int w, int c, const float* xy, int64_t p, float* out,
                          void* stream);
int gens_patch_sample_bwd(const float* image, int h, int w, int c, const float* xy, const float* g_out, int64_t p,
                          float* g_xy, void* stream);
int gens_upsample2d_into(const float* src, int n, int c, int hs, int ws, float* dst, int h, int w, int c_pad_dst,
                         int c_off, void* stream);
/* ... and the whole cat([f0, up(f1), up(f2), ...], 1) of implicit_surface.py:313-326 in ONE launch (ABI 12): srcs[i] is map i, (n, C_i, hs_i, ws_i)
 * NCHW with chw[3 i .. 3 i + 2] = (C_i, hs_i, ws_i); every texel of dst (n, H, W, C_pad_dst) is written whole, its pad channels with zeros (no
 * fill beforehand); the values are gens_upsample2d_into's bit for bit. */
int gens_upsample2d_cat(const float* const* srcs, const int* chw, int n_maps, int n, float* dst, int h, int w, int c_pad_dst,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K13  compute_LNCC, the photometric patch statistic of the loss     (models/losses/ncc.py:7-55, loss.py:36)
 *   ref (1, B, P, C), src (S, B, P, C) contiguous float32 (render_core's ref_gray_val / sampled_gray_val; P = 121, C = 12);
 *   S * C <= 64.  ncc (B): mean over channels of clamp(1 - cc, 0, 2), mean of the two smallest source views;
 *   sel (B, 2) int32: the two selected source indices (consumed by the backward).
 *   bwd: g_ncc (B) -> g_ref like ref, g_src like src, both overwritten.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_lncc_fwd(const float* ref, const float* src, int64_t n_rays, int n_src, int n_patch, int n_ch, float* ncc, int32_t* sel,
                  void* stream);
int gens_lncc_bwd(const float* ref, const float* src, const float* g_ncc, const int32_t* sel, int64_t n_rays, int n_src, int n_patch,
                  int n_ch, float* g_ref, float* g_src, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K12  iso-surface extraction of the SDF lattice: replaces mcubes.marching_cubes(u, threshold)
 *      (implicit_surface.py:423; PyMCubes==0.1.4, requirements.txt:11 -- third-party, classic marching cubes)
 *   u (X, Y, Z) float32, z fastest (the lattice of implicit_surface.py:407-421); iso = threshold.
 *   gens_mc_classify: per lattice point p  vmask[p] bit a = the edge p -> p + e_a straddles iso (a = x, y, z),
 *     vcount[p] = popcount, cases[p] = Bourke case index of the cell with origin p (bit n: u[corner n] < iso),
 *     tcount[p] = tri_count[cases[p]] (0 where p is not a cell origin).  tri_count: DEVICE (256) uint8.
 *   The caller turns vcount / tcount into exclusive scans voff / toff (int32) and allocates
 *     vertices (V, 3) float64 in INDEX coordinates and triangles (T, 3) int32.
 *   gens_mc_emit: vertex of edge (p, a) = p + e_a (iso - u_p) / (u_q - u_p) in float64, stored at voff[p] + rank of a;
 *     triangles of cell p from tri_table (DEVICE (256, table_stride) int8 edge ids, -1 padded; gens_amd/mc_tables.py)
 *     at toff[p].  Order: vertices by (p, a), triangles by (p, table order).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_mc_classify(const float* u, int x, int y, int z, float iso, const uint8_t* tri_count, uint8_t* vmask,
                     uint8_t* vcount, uint8_t* cases, uint8_t* tcount, void* stream);
int gens_mc_emit(const float* u, int x, int y, int z, float iso, const int8_t* tri_table, int table_stride,
                 const uint8_t* vmask, const int32_t* voff, const uint8_t* cases, const uint8_t* tcount, const int32_t* toff,
                 double* vertices, int32_t* triangles, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K10  tv_regularization, one level per call                      (implicit_surface.py:135-150)
 *   vol (4, X, Y, Z), mask (X, Y, Z); partial (n_blocks, 4) per-block sums [tx_num, ty_num, tz_num, mx_count]
 *   (n_blocks = gens_tv_blocks(x*y*z)); the host finishes sqrt((tx+ty+tz)/(count+1e-8)) * 0.5^level (Q13).
 * bwd: g_vol (4,X,Y,Z) = coef * d(tx_num+ty_num+tz_num)/d vol, overwritten; coef = g * 0.5^level / (2*tv*(count+1e-8)).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_tv_blocks(int64_t n_voxels);
int gens_tv_fwd(const float* vol, const float* mask, int x, int y, int z, float* partial, void* stream);
int gens_tv_bwd(const float* vol, const float* mask, int x, int y, int z, float coef, float* g_vol, void* stream);
/* the same with the coefficient = coef * coef_dev[0], coef_dev a device scalar (the upstream gradient stays on the device: no host sync) */
int gens_tv_bwd_scaled(const float* vol, const float* mask, int x, int y, int z, float coef, const float* coef_dev, float* g_vol,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K11  the lattice of extract_geometry                            (implicit_surface.py:407-418)
 *   pts (count, 3): lattice points first .. first+count-1 of a res^3 grid in x-major order, coordinates equal to
 *   torch.linspace(bmin, bmax, res).
 * ---------------------------------------------------------------------------------------------------------- */
int gens_lattice_points(const float* bmin3_host, const float* bmax3_host, int res, int64_t first, int64_t count,
                        float* pts, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K14  C (m x n) = A^T B for tall row-major operands A (k x m), B (k x n), k >> m, n: the weight-gradient product of the training
 *      step (dW = dY^T X, what aten::mm computes inside torch.nn.functional.linear's backward; the reference reaches it through
 *      autograd from sdf_network.py:98-123 and blending_network.py:69-118).  Exact float32 (fp32 MFMA), K split into slabs whose
 *      partial results are added in a fixed order (deterministic).  workspace: gens_gemm_tn_slabs(k, m, n) * m * n floats.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_gemm_tn_slabs(int64_t k, int m, int n);
int gens_gemm_tn(const float* a, const float* b, int64_t k, int m, int n, float* workspace, float* c, void* stream);
/* Up to 12 products C_p (m_p x n_p) = A_p^T B_p over the same k rows in one launch; A_p / B_p are column blocks of row-major buffers
 * with leading dimensions lda[p] / ldb[p] (HOST arrays of device pointers / ints).  c: the C_p concatenated (row major each).
 * workspace: gens_gemm_tn_batch_workspace(count, m, n, k) floats.  Partial sums are added in a fixed order (deterministic). */
int64_t gens_gemm_tn_batch_workspace(int count, const int* m, const int* n, int64_t k);
int gens_gemm_tn_batch(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                       const int* n, int64_t k, float* workspace, float* c, void* stream);
/* The same with the number of rows that EXIST left on the device: only the first min(k, k_rows ceil(*k_live / k_div)) rows of the
 * operands are read -- *k_live work items of the producer, k_div of which share a workgroup that wrote k_rows rows (gens_sdf_train_bwd:
 * 32 points -> 128 rows; gens_blend_train_bwd: floor(32 / S) points -> 32 rows) under a device-side point count. */
int gens_gemm_tn_batch_live(int count, const float* const* a, const int* lda, const float* const* b, const int* ldb, const int* m,
                            const int* n, int64_t k, const int32_t* k_live, int k_div, int k_rows, float* workspace, float* c,
                            void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K15  the 3 x 3 x 3 convolutions of the cost-volume U-Net (reg_network.py:7-50: nn.Conv3d(k=3, padding=1, stride 1 | 2) and
 *      nn.ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1), i.e. aten::convolution / convolution_backward as MIOpen runs
 *      them), batch 1, float32.  A coarse tensor P (cp, X, Y, Z) and a fine tensor Q (cq, sX, sY, sZ) are tied by W[cp][cq][27]:
 *        gather    P[a][o] = bias[a] + sum W[a][b][t] Q[b][s o + t - 1]      Conv3d forward (a = out, b = in), ConvTranspose3d dgrad
 *        scatter2  Q[b][i] = sum W[a][b][t] P[a][(i + 1 - t) / 2]            ConvTranspose3d forward (a = in, b = out), Conv3d(s=2) dgrad
 *        wgrad     dW[a][b][t] = sum_o P[a][o] Q[b][s o + t - 1]             both weight gradients, in W's own layout
 *      dims_p = {X, Y, Z} (host).  Weight layouts (the host permutes the small tensor): gather w (cq, 27, cpp), cpp = cp rounded up
 *      to 8 (cp > 4) or 4, zero-filled; scatter2 w (cp, 27, cqp), cqp likewise from cq.  bias may be NULL.  A stride-1 scatter is a
 *      gather with the 27 taps reversed and the channel roles swapped.  wgrad writes partial sums (parts, cpp4, cqp8, 27) with
 *      parts = gens_conv3d_wgrad_parts(...), cpp4 = cp rounded up to 4, cqp8 = cq rounded up to 8; the caller adds over `parts`
 *      (fixed order: deterministic).  Every tensor must be smaller than 2 GiB.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_conv3d_gather(const float* q, const float* w, const float* bias, int cp, int cq, const int* dims_p, int stride,
                       float* p, void* stream);
int gens_conv3d_scatter2(const float* p, const float* w, int cp, int cq, const int* dims_p, float* q, void* stream);
int gens_conv3d_wgrad_parts(int cp, int cq, const int* dims_p);
/* the number of parts gens_conv3d_wgrad writes for THIS stride (the stride-1 matrix-core kernel cuts the volume its own way); the workspace
 * of a call must hold gens_conv3d_wgrad_parts_strided(cp, cq, dims_p, stride) parts */
int gens_conv3d_wgrad_parts_strided(int cp, int cq, const int* dims_p, int stride);
int gens_conv3d_wgrad(const float* p, const float* q, int cp, int cq, const int* dims_p, int stride, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K16  InstanceNorm3d (affine=False, biased variance) + ReLU after every convolution of the U-Net (reg_network.py:16-17,39-40:
 *      aten::instance_norm + aten::relu_ and their backward), batch 1, float32 planes x (c, n).
 *        gens_instnorm_stats           partials (c, blocks, 2) float64: sums of x and x^2 per workgroup, blocks = gens_instnorm_blocks(c, n)
 *        gens_instnorm_relu_fwd        y = max((x - mean) * rstd, 0), mean_rstd (c, 2) float32
 *        gens_instnorm_relu_bwd_stats  partials (c, blocks, 2) float64: sums of g and g xhat, g = gy [xhat > 0], xhat = (x - mean) * rstd
 *        gens_instnorm_relu_bwd        gx = rstd (g - m1 - xhat m2), g_means (c, 2) float32 = {m1 = mean(g), m2 = mean(g xhat)}
 *        gens_instnorm_finish          partials -> (c, 2) float32 in one launch, float64 inside: mode 0 = (mean, 1 / sqrt(max(E[x^2] - mean^2, 0) + eps))
 *                                      from gens_instnorm_stats' partials, mode 1 = (m1, m2) from gens_instnorm_relu_bwd_stats' (eps ignored)
 *      A batch (nn.InstanceNorm2d of the feature decoder, feature_network_mnasnet.py:29-50) is n * c planes: pass c = n * c.
 * ---------------------------------------------------------------------------------------------------------- */
int gens_instnorm_blocks(int c, int64_t n);
int gens_instnorm_stats(const float* x, int c, int64_t n, double* partials, void* stream);
int gens_instnorm_finish(const double* partials, int c, int64_t n, double eps, int mode, float* out, void* stream);
int gens_instnorm_relu_fwd(const float* x, const float* mean_rstd, int c, int64_t n, float* y, void* stream);
/* the same + skip (c, n): the decoder blocks add the encoder's tensor right after norm + ReLU (reg_network.py:158) */
int gens_instnorm_relu_add_fwd(const float* x, const float* mean_rstd, const float* skip, int c, int64_t n, float* y, void* stream);
int gens_instnorm_relu_bwd_stats(const float* x, const float* gy, const float* mean_rstd, int c, int64_t n, double* partials, void* stream);
int gens_instnorm_relu_bwd(const float* x, const float* gy, const float* mean_rstd, const float* g_means, int c, int64_t n, float* gx,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K19  step-boundary kernels of a training step: what the reference does with dozens of scalar-sized PyTorch launches around the
 *      hot-path kernels, each in ONE launch on the device and without host synchronisation (k19_step.hip).
 *
 * gens_scene_setup: the camera constants of a scene --
 *     w2c = inverse(c2ws)                                  (volume.py:34, projector.py:322)
 *     ks[l] = intrs with rows 0-1 times 0.5^l, l < 8       (volume.py:24-25, projector.py:317-318; Q2)
 *     rot_inv = inverse(c2ws[0, :3, :3])                   (implicit_surface.py:242,245)
 *     kinv_ref = inverse(intrs)[0, :3, :3]                 (projector.py:364)
 *   written to `cams`, gens_scene_cams_floats(nv) floats laid out as
 *     [w2c nv x 16][ks GENS_MAX_LEVELS x nv x 16][rot_inv 9 + 3 pad][kinv_ref 9 + 3 pad][status: int32, bit 0 = a singular matrix][3 pad].
 *   Inverses: float64 Gauss-Jordan with partial pivoting, rounded once to float32; a singular matrix gives NaNs and sets the status bit
 *   (torch.inverse raises: the host checks the bit where it synchronises anyway).
 * gens_pack_maps / gens_unpack_maps: gens_pack_nchw / gens_unpack_nhwc for up to 8 maps in one launch; src / dst: HOST arrays of device
 *   pointers, nchw: HOST int[4 * n_maps] = (n, C, H, W) per map.
 * gens_compact_points: the index list of a step's masked SDF evaluation (implicit_surface.py:121-124,174-177,256-257,484-497): rows
 *   [0, n_ray_pts) are ray samples with flags from gens_ray_points (none set: the first min(10, n) of them, Q7), the next n_always rows
 *   are always selected (the random points), the rest are the pseudo points with their own flags.  idx (n) int64: selected rows in
 *   increasing order; counts (3) int32 = {selected, selected ray samples, selected pseudo points}.  n < 2^24; two launches (counts per
 *   workgroup, ordered write); scratch: gens_compact_points_scratch(n) 4-byte words of device memory.
 *   Optional outputs of the same launch (NULL to skip): the values the reference's dense tensors hold for UNSELECTED rows (Q8) --
 *   y_fill (n): 100 for ray samples, 0 for pseudo points (implicit_surface.py:125,497); g_fill / s_fill (n, 3): 0; rgb_fill
 *   (n_ray_pts, 3): 0; vis_fill (n_ray_pts, n_src): 0 -- and scalars (4) = {max of z (nz floats; :301), inv_s = clip(exp(10 variance),
 *   1e-6, 1e6) (variance_network.py:11, :206), 1 / inv_s, 1 if inv_s is inside the clip range else 0}.
 * gens_tv_levels_fwd / _bwd: tv_regularization (implicit_surface.py:135-150; Q13) of all levels.  vols[l] (4, X, Y, Z), masks[l]
 *   (X, Y, Z): HOST arrays of device pointers; dims HOST int[3 * n_levels]; Z % 4 == 0, 16-byte aligned planes.
 *   fwd: partial: scratch of gens_tv_levels_blocks(dims, n_levels) float4; out (1 + n_levels): out[0] = tv_reg, out[1 + l] = the
 *   level's backward coefficient 0.5^l / (2 tv_l den_l).  bwd: g (1) = d loss / d tv_reg (DEVICE) -> g_vols[l] (overwritten).
 * ---------------------------------------------------------------------------------------------------------- */
/* gens_patch_warp_fwd / _bwd: surface_patch_warp (projector.py:353-437 as called from implicit_surface.py:301-328) in one launch each.
 *   rays_o, rays_d (B, 3); z (B) = the clamped crossing depth z_vals_sdf0 (:300-303), the only differentiable input; g0 (B, 3) = the SDF
 *   gradient at the surface point, un-normalised (normalised and rotated into the reference camera in-kernel, :308-310; used detached);
 *   c2ws, intrs (nv, 4, 4), kinv_ref (3, 3) = inverse(intrs)[0, :3, :3] (gens_scene_setup); tex (nv, H, W, C_pad) texels of the warp
 *   features (gens_upsample2d_into), C <= 16 channels; patch: odd patch size (11).
 *   fwd -> ref (1, B, P, C), sampled (nv - 1, B, P, C), P = patch^2 in row-major (y, x) order.  bwd: g_sampled -> g_z (B), overwritten. */
int gens_patch_warp_fwd(const float* rays_o, const float* rays_d, const float* z, const float* g0, int64_t n_rays, const float* c2ws,
                        const float* intrs, const float* kinv_ref, int nv, const float* tex, int h, int w, int c, int patch, float* ref,
                        float* sampled, void* stream);
int gens_patch_warp_bwd(const float* rays_o, const float* rays_d, const float* z, const float* g0, int64_t n_rays, const float* c2ws,
                        const float* intrs, const float* kinv_ref, int nv, const float* tex, int h, int w, int c, int patch,
                        const float* g_sampled, float* g_z, void* stream);
/* gens_loss_fwd / _bwd: Loss.forward (models/losses/loss.py:24-93) in one launch, its backward in one launch.  The argument block: */
typedef struct {
    const float *color, *target;            /* color_fine, targets["color"]: (B, 3) */
    const uint8_t* valid;                   /* valid_mask (B) */
    int64_t b;
    const float* sparse;                    /* sparse_sdf (n_sparse) */
    int64_t n_sparse;
    float sparse_scale;
    const float* pseudo;                    /* pseudo_sdf (n_pseudo) or NULL */
    int64_t n_pseudo;
    const float *ncc, *mid_in;              /* compute_LNCC's result (B), mid_inside_sphere (B) */
    const float *depth, *pseudo_depth_t, *depth_t;      /* render_depth (B); targets["pseudo_depth"], targets["depth"] (B) or NULL */
    const float *ge, *se, *tv;              /* gradient_error, smooth_error, tv_reg: DEVICE scalars */
    float w_color, w_igr, w_sparse, w_mfc, w_smooth, w_tv, w_pseudo_sdf, w_pseudo_depth;
    float* out;                             /* (16): loss, color, eikonal, sparse, mfc, smooth, tv, depth, pseudo_sdf, pseudo_depth, 4 denominators, 2 spare */
    const float* g;                         /* backward: d L / d loss, DEVICE scalar */
    float *g_color, *g_sparse, *g_pseudo, *g_ncc, *g_depth, *g_scalars;      /* backward outputs (NULL to skip); g_scalars (3): ge, se, tv */
} gens_loss_args;
int gens_loss_fwd(const gens_loss_args* args, void* stream);
int gens_loss_bwd(const gens_loss_args* args, void* stream);
/* gens_coarse_z: the coarse depths of a render (implicit_surface.py:356-363) in one launch: z (B, n) = near + (far - near) * steps[j]
 *   (+ (t_rand[ray] - 0.5) * 2 / n when t_rand != NULL); near / far: one float each (per_ray = 0) or (B) (per_ray = 1); steps (n) =
 *   torch.linspace(0, 1, n).
 * gens_blend_train_wgrad: the 23 parameter gradients of a BlendingNetwork in one launch from cc = gens_gemm_tn_batch's result for
 *   gens_blend_train_bwd's eleven products and s_part (n_part); n_feat = 3 + 4 n_levels; grads: HOST array of 23 device pointers (row-major
 *   weights and biases in the order of gens_blend_train_fwd, then s). */
int gens_coarse_z(const float* near, const float* far, int per_ray, const float* steps, const float* t_rand, int64_t n_rays, int n, float* z,
                  void* stream);
int gens_blend_train_wgrad(const float* cc, const float* s_part, int n_part, const float* s, int n_feat, float* const* grads, void* stream);
int64_t gens_scene_cams_floats(int nv);
int gens_scene_setup(const float* c2ws, const float* intrs, int nv, float* cams, void* stream);
int gens_pack_maps(const float* const* src, float* const* dst, const int* nchw, int n_maps, void* stream);
int gens_unpack_maps(const float* const* src, float* const* dst, const int* nchw, int n_maps, void* stream);
/* Views taken out of up to 16 per-scene maps in one launch (ABI 12): dst[k][j] = src[k][index[j]], j < n_sel, where a view of map k is
 * floats_per_view[k] contiguous floats (a multiple of 4; both sides 16-byte aligned) and src[k] holds views_in_src[k] of them; index: n_sel int64 on the
 * DEVICE (negative values count from the end, as torch's).  What fine-tuning's `self.features[i][view_ids]` (gens.py:151-153) does to the frozen
 * pyramid -- here for the maps and their texel / warp layouts together. */
int gens_select_views(const float* const* src, float* const* dst, const int* floats_per_view, const int* views_in_src, int n_maps,
                      const int64_t* index, int n_sel, void* stream);
int64_t gens_compact_points_scratch(int64_t n);   /* 4-byte words of scratch for n rows */
int gens_compact_points(const uint8_t* valid, int64_t n_ray_pts, int64_t n_always, int64_t n, int64_t* idx, int32_t* counts,
                        float* y_fill, float* g_fill, float* s_fill, float* rgb_fill, uint8_t* vis_fill, int n_src, const float* z,
                        int64_t nz, const float* variance, float* scalars, int32_t* scratch, void* stream);
int gens_tv_levels_blocks(const int* dims, int n_levels);
int gens_tv_levels_fwd(const float* const* vols, const float* const* masks, const int* dims, int n_levels, float* partial, float* out,
                       void* stream);
int gens_tv_levels_bwd(const float* const* vols, const float* const* masks, const int* dims, int n_levels, const float* out,
                       const float* g, float* const* g_vols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENS_HIP_H */
