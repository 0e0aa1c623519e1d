/*
 * epnet_ops.h -- C ABI of the MI355X (gfx950) point-cloud geometry library, libepnet_hip.so.
 *
 * Every entry point below replaces one launcher of the reference's three CUDA extensions
 * (pointnet2_cuda, iou3d_cuda, roipool3d_cuda); the reference interface it stands in for is
 * cited as path:line relative to the reference checkout. Conventions:
 *
 *   - plain pointers and sizes only; no torch / pybind types;
 *   - every device pointer is a HIP device address of a contiguous fp32 / int32 / int64 array
 *     laid out exactly as the reference lays it out (shapes in the comments);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all device entry
 *     points are asynchronous on that stream, allocate nothing, keep no state between calls and
 *     are re-entrant (the reference's ops are called concurrently from several host threads
 *     under nn.DataParallel, tools/train_rcnn.py:221-223);
 *   - return value: EPNET_OK (0) or a negative EPNET_E* code; the library never calls exit()
 *     (the reference does: e.g. pointnet2_lib/pointnet2/src/ball_query_gpu.cu:62-65);
 *   - scratch memory is supplied by the caller (`*_workspace_bytes` + `workspace`), replacing
 *     the per-call cudaMalloc/cudaFree of lib/utils/iou3d/src/iou3d.cpp:87,98 and
 *     lib/utils/roipool3d/src/roipool3d_kernel.cu:214,222,231-232.
 *
 * Arithmetic contract (DESIGN.md "Parity definition"): IEEE fp32, source order, no fused
 * contraction; integer outputs bit-exact versus oracle/ (the CPU restatement of the reference
 * kernels), float copies exact, float sums to 1e-5.
 */
#ifndef EPNET_OPS_H
#define EPNET_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPNET_ABI_VERSION 1

#define EPNET_OK 0
#define EPNET_EINVAL (-1)   /* bad size / NULL pointer */
#define EPNET_ELAUNCH (-2)  /* hipLaunch / hipGetLastError reported a failure */
#define EPNET_ENOMEM (-3)   /* workspace too small */
#define EPNET_ELIMIT (-4)   /* problem size outside what the kernels support */

typedef void *epnet_stream_t;

int epnet_abi_version(void);
const char *epnet_strerror(int code);
/* last HIP error string recorded by this thread's most recent failing call ("" if none) */
const char *epnet_last_hip_error(void);

/* ----------------------------------------------------------------------------------------
 * pointnet2 (pointnet2_lib/pointnet2/src/pointnet2_api.cpp:10-24)
 * -------------------------------------------------------------------------------------- */

/* furthest_point_sampling_kernel_launcher, sampling_gpu.cu:211-253 (wrapper sampling.cpp:36-46).
 * xyz (B,N,3) f32; temp (B,N) f32 in/out running min squared distance, caller-filled with 1e10
 * (pointnet2_utils.py:26); idx (B,M) i32 out. idx[:,0] = 0. Tie-breaks reproduce the reference
 * block reduction for block size opt_n_threads(N) (cuda_utils.h:10-14). temp may be NULL: the
 * kernel then starts from 1e10 and does not write the distances back. */
int epnet_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                  epnet_stream_t stream);

/* gather_points_kernel_launcher_fast, sampling_gpu.cu:26-43. points (B,C,N), idx (B,M) -> out (B,C,M) */
int epnet_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                        float *out, epnet_stream_t stream);

/* gather_points_grad_kernel_launcher_fast, sampling_gpu.cu:65-83.
 * grad_out (B,C,M), idx (B,M) -> grad_points (B,C,N) accumulated into (caller-zeroed, pointnet2_utils.py:67) */
int epnet_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                             float *grad_points, epnet_stream_t stream);

/* ball_query_kernel_launcher_fast, ball_query_gpu.cu:48-66 (note: new_xyz before xyz, ball_query.cpp:14).
 * new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32. Every slot is written (zeros when the
 * ball is empty; the reference leaves the caller's zero fill, pointnet2_utils.py:218). */
int epnet_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx, epnet_stream_t stream);

/* Same result as epnet_ball_query, with caller-supplied device scratch (16-byte aligned) that lets the
 * library index the scene spatially (Morton-sorted copy + per-bucket boxes) instead of scanning all N
 * points per centre; bit-identical output. epnet_ball_query_workspace_bytes() == 0 means the direct scan
 * is used anyway (small or very large scenes) and workspace may be NULL. */
size_t epnet_ball_query_workspace_bytes(int b, int n, int m);
int epnet_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                        int *idx, void *workspace, size_t workspace_bytes, epnet_stream_t stream);

/* group_points_kernel_launcher_fast, group_points_gpu.cu:69-86.
 * points (B,C,N), idx (B,M,ns) -> out (B,C,M,ns) */
int epnet_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, epnet_stream_t stream);

/* group_points_grad_kernel_launcher_fast, group_points_gpu.cu:27-44.
 * grad_out (B,C,M,ns), idx (B,M,ns) -> grad_points (B,C,N) accumulated into (caller-zeroed) */
int epnet_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points, epnet_stream_t stream);

/* Fused tail of QueryAndGroup.forward (pointnet2_utils.py:250-257): one call writes
 *   out (B, 3+C, M, ns) = [ xyz[b, idx[b,i,s], :] - new_xyz[b,i,:]   (3 channels, if use_xyz)
 *                           features[b, :, idx[b,i,s]]               (C channels) ]
 * reading xyz in its (B,N,3) layout -- no transposed copy of xyz, no separate centre-subtraction pass over the
 * grouped tensor and no torch.cat copy of it. features may be NULL when c == 0. Values are bit-identical to the
 * reference composition (a gather and one fp32 subtraction). */
int epnet_group_concat(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                       const float *features, const int *idx, float *out, int use_xyz, epnet_stream_t stream);
/* the same with caller scratch: where the feature rows are too long for on-chip staging (n > 16384 points: BASELINE config
 * 5) the features are turned point-major once in the scratch and gathered as contiguous rows. workspace_bytes = 0 (or
 * workspace NULL): plain epnet_group_concat. */
size_t epnet_group_concat_workspace_bytes(int b, int c, int n, int npoints, int nsample);
int epnet_group_concat_ws(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                          const float *features, const int *idx, float *out, int use_xyz, void *workspace,
                          size_t workspace_bytes, epnet_stream_t stream);

/* gradient of the above w.r.t. features: grad_out (B, 3+C | C, M, ns) -> grad_features (B,C,N) accumulated into */
int epnet_group_concat_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                            float *grad_features, int use_xyz, epnet_stream_t stream);

/* Gradients of the two grouping forms with caller-supplied device scratch: the scatter-add is inverted (the
 * positions are grouped by target point, then every target sums its own list out of LDS) so that no atomic
 * is needed. Same result up to the summation order, which the reference's atomicAdd leaves unspecified too.
 * A workspace size of 0 means the scratch-free kernels are used anyway. */
size_t epnet_group_points_grad_workspace_bytes(int b, int n, int npoints, int nsample);
int epnet_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                               float *grad_points, void *workspace, size_t workspace_bytes, epnet_stream_t stream);
int epnet_group_concat_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                               float *grad_features, int use_xyz, void *workspace, size_t workspace_bytes,
                               epnet_stream_t stream);

/* three_nn_kernel_launcher_fast, interpolate_gpu.cu:55-74.
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 squared distances, idx (B,n,3) i32 */
int epnet_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx, epnet_stream_t stream);

/* Same result as epnet_three_nn with caller-supplied device scratch (16-byte aligned): the known points are
 * indexed spatially and each unknown point only visits the buckets that can hold one of its three
 * nearest; bit-identical output. A workspace size of 0 means the direct scan is used (workspace may be NULL). */
size_t epnet_three_nn_workspace_bytes(int b, int n, int m);
int epnet_three_nn_ws(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                      void *workspace, size_t workspace_bytes, epnet_stream_t stream);

/* three_interpolate_kernel_launcher_fast, interpolate_gpu.cu:99-117 (argument order b,c,m,n).
 * points (B,C,m), idx (B,n,3), weight (B,n,3) -> out (B,C,n) */
int epnet_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                            const float *weight, float *out, epnet_stream_t stream);

/* three_interpolate_grad_kernel_launcher_fast, interpolate_gpu.cu:144-161 (argument order b,c,n,m).
 * grad_out (B,C,n), idx, weight (B,n,3) -> grad_points (B,C,m) accumulated into (caller-zeroed) */
int epnet_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                 const float *weight, float *grad_points, epnet_stream_t stream);

/* the groupings of all scales of an MSG level in one call (two scales: the feature rows are staged in LDS once for both).
 * nsamples / idx / out are HOST arrays of nscales entries; out[k] is (b, 3+c | c, npoints, nsamples[k]). Same results as
 * nscales calls of epnet_group_concat. */
int epnet_group_concat_multi(int b, int c, int n, int npoints, int nscales, const int *nsamples, const float *xyz,
                             const float *new_xyz, const float *features, const int *const *idx, float *const *out,
                             int use_xyz, epnet_stream_t stream);

/* atomic-free form of the above with caller scratch (inverse index over the known points); 0 bytes = not used */
size_t epnet_three_interpolate_grad_workspace_bytes(int b, int n, int m);
int epnet_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                    const float *weight, float *grad_points, void *workspace, size_t workspace_bytes,
                                    epnet_stream_t stream);

/* The first shared-MLP layer of an SA level folded into its grouping. The reference materialises
 * [xyz[idx] - centre ; features[idx]] (pointnet2_utils.py:250-257) and runs a 1x1 convolution W over it
 * (pointnet2_modules.py:61); that convolution is linear and pointwise, so W . [dxyz ; F[:, idx]] =
 * W_xyz . dxyz + (W_f . F)[:, idx]. With z = W_f . F (b, c, n) computed by the caller (a dense GEMM over n columns instead of
 * npoints * nsample), this writes the layer's pre-activations out (b, c, npoints, nsample):
 *   out[b,co,m,s] = z[b,co,idx[b,m,s]] + (w_xyz[co][0]*dx + w_xyz[co][1]*dy + w_xyz[co][2]*dz) (+ bias[co]),
 *   (dx,dy,dz) = xyz[b,idx[b,m,s]] - new_xyz[b,m]; w_xyz (c,3), bias (c) or NULL. */
int epnet_group_linear(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz, const float *z,
                       const int *idx, const float *w_xyz, const float *bias, float *out, epnet_stream_t stream);

/* gradient of epnet_group_linear w.r.t. w_xyz: grad_w (c,3) += sum over (b,m,s) of grad_out[b,co,m,s] * (xyz[b,idx[b,m,s]] -
 * new_xyz[b,m])[k]; the caller zero-fills grad_w (accumulated with float atomics: summation order unspecified). The gradient
 * w.r.t. z is epnet_group_points_grad of grad_out, the one w.r.t. bias its sum over (b,m,s). */
int epnet_group_linear_grad_w(int b, int c, int n, int npoints, int nsample, const float *grad_out, const float *xyz,
                              const float *new_xyz, const int *idx, float *grad_w, epnet_stream_t stream);

/* the neighbourhood max-pool of an SA level, F.max_pool2d(kernel_size=[1, nsample]) of pointnet2_modules.py:61-68: x
 * (rows, nsample) contiguous (rows = B * C * npoint) -> out (rows) = the row maximum (ties: lowest position; NaN
 * propagates), arg (rows) i32 or NULL = its position for the backward. epnet_pool_max_grad: grad_x (rows, nsample) =
 * grad_out[row] at arg[row], zero elsewhere (every element is written). */
int epnet_pool_max(long long rows, int nsample, const float *x, float *out, int *arg, epnet_stream_t stream);
int epnet_pool_max_grad(long long rows, int nsample, const float *grad_out, const int *arg, float *grad_x,
                        epnet_stream_t stream);

/* LI-Fusion's point-to-pixel sampler: Feature_Gather of lib/net/pointnet2_msg.py:107-120 =
 * torch grid_sample(feature_map (b,c,h,w), xy (b,1,n,2) in [-1,1]) -> out (b,c,n), bilinear, zero padding, with the
 * torch.gather of xy over the FPS indices (:214-217) folded in: idx (b,n) i32 or NULL picks the rows of xy (b,n_src,2)
 * (NULL: n_src == n, row q of xy), xy_out (b,n,2) or NULL receives the picked coordinates for the next level.
 * align_corners as in torch (the reference was written for torch <= 1.2, where grid_sample behaved as align_corners=True).
 * epnet_feature_gather_grad: grad_feature_map (b,c,h,w), zero-filled by the caller, += the bilinear scatter of grad_out
 * (b,c,n) at xy (b,n,2) (float atomics). */
int epnet_feature_gather(int b, int c, int h, int w, int n_src, int n, int align_corners, const float *feature_map,
                         const float *xy, const int *idx, float *out, float *xy_out, epnet_stream_t stream);
int epnet_feature_gather_grad(int b, int c, int h, int w, int n, int align_corners, const float *grad_out, const float *xy,
                              float *grad_feature_map, epnet_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * scene index: one spatial sort of a level's points (1024 <= n <= 65536), built once in caller scratch and
 * shared by the sampling and both ball queries of that level (the reference has no counterpart: every one of
 * its kernels scans all n points). Results are identical to the plain entry points.
 * epnet_scene_index_bytes returns 0 where no index applies; the *_indexed entry points then (or with
 * index == NULL) run the plain path. For n > 16384 the last part of the buffer is scratch of the sampling kernel
 * (its running distances in sorted order): two samplings over ONE index buffer must not run concurrently there;
 * the ball queries and three_nn never touch that part. The three sampling entry points therefore take the index as a NON-const
 * pointer (they write that tail; the sorted points and boxes in front of it are only read), everybody else as const.
 * -------------------------------------------------------------------------------------- */
size_t epnet_scene_index_bytes(int b, int n);
int epnet_scene_index_build(int b, int n, const float *xyz, void *index, size_t index_bytes, epnet_stream_t stream);
/* The index of the n points xyz_src[idx[0 .. n)] of every scene (rows of a (b, n_src, 3) cloud: the centres a sampling has just
 * picked, pointnet2_modules.py:39-45), which are also written to gathered (b, n, 3) in that order: the centre gather of an SA level
 * and the index build of the next level in one launch. 1024 <= n <= 16384; idx values in [0, n_src). */
int epnet_scene_index_build_gathered(int b, int n_src, int n, const float *xyz_src, const int *idx, float *gathered, void *index,
                                     size_t index_bytes, epnet_stream_t stream);
/* same contract as epnet_furthest_point_sampling (sampling_gpu.cu:211-253) */
int epnet_furthest_point_sampling_indexed(int b, int n, int m, const float *xyz, void *index,
                                          size_t index_bytes, float *temp, int *idx, epnet_stream_t stream);
/* the head of an SA module in one call (pointnet2_modules.py:39-45): furthest point sampling from a fresh state
 * (all running distances 1e10, pointnet2_utils.py:26) and new_xyz (b,m,3) = the selected rows of xyz. temp = scratch
 * (b,n) or NULL (allowed for 64 <= n <= 16384); index = scene index of xyz or NULL */
int epnet_sample_centres(int b, int n, int m, const float *xyz, void *index, size_t index_bytes, float *temp,
                         int *idx, float *new_xyz, epnet_stream_t stream);
/* epnet_sample_centres for the levels of a sampling pyramid (SA level l+1 samples the centres of level l). Furthest point
 * sampling is nested: while every round's maximum is unique, the first m samples of a furthest-point sequence ARE the
 * furthest-point samples of that sequence -- idx = 0 .. m-1, bit for bit what sampling_gpu.cu:94-209 computes on the centres
 * (its tie-break matters among equal maxima only). prefix_in (b ints or NULL): leading rounds of the sampling that produced xyz
 * in which the maximum was unique up to exact twins -- points with identical coordinates, as the reference's loader creates when
 * it pads a short scene (kitti_rcnn_dataset.py:338-342): the unpicked twin is at distance 0 from then on and cannot be sampled
 * while the maximum is positive, so the sequence of sampled COORDINATES does not depend on the tie-break (= the prefix_out of
 * that call); scenes with prefix_in[b] >= m take the identity, the others
 * run the rounds. prefix_out (b ints or NULL): the same knowledge about this call's output (at least that many rounds), 0 where
 * the kernel cannot tell; ties are looked for during the first prefix_cap rounds only (<= 0: all) -- pass the next level's m.
 * new_xyz may be NULL here (indices only: the centres then come out of epnet_scene_index_build_gathered of the next level). */
int epnet_sample_centres_chain(int b, int n, int m, const float *xyz, void *index, size_t index_bytes, float *temp,
                               int *idx, float *new_xyz, const int *prefix_in, int *prefix_out, int prefix_cap,
                               epnet_stream_t stream);
/* same contract as epnet_three_nn (interpolate_gpu.cu:55-74); known_index = scene index of `known` (NULL: plain
 * path), unknown_index = scene index of `unknown` or NULL */
int epnet_three_nn_indexed(int b, int n, int m, const float *unknown, const float *known, const void *unknown_index,
                           size_t unknown_index_bytes, const void *known_index, size_t known_index_bytes, float *dist2,
                           int *idx, epnet_stream_t stream);
/* same contract as epnet_ball_query (ball_query_gpu.cu:48-66) */
int epnet_ball_query_indexed(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                             const void *index, size_t index_bytes, int *idx, epnet_stream_t stream);
/* the nscales ball queries of an MSG level (same centres, same points, nested balls) in ONE launch: every distance is
 * computed once. radii / nsamples / idx are HOST arrays of nscales entries, idx[k] a device (b, m, nsamples[k])
 * buffer. Same results as nscales calls of epnet_ball_query_indexed (which is what happens unless nscales == 2). */
int epnet_ball_query_indexed_multi(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                   const float *new_xyz, const float *xyz, const void *index, size_t index_bytes,
                                   int *const *idx, epnet_stream_t stream);
/* The same queries with the centres served in THEIR spatial order (ball_query_gpu.cu:9-45 gives thread i centre i; the centres of an
 * SA level come in furthest-point order, which scatters consecutive centres over the whole scene). centre_index = the scene index
 * (epnet_scene_index_build / _build_gathered) of the cloud new_xyz (b, m, 3), in that order -- the next SA level samples the
 * centres, so the stack has it anyway: wave w serves the w-th centre of that order, its neighbours walk the same point buckets,
 * which are then cache hits. nscales 1 or 2 (otherwise, or with centre_index == NULL, or for m < 1024:
 * epnet_ball_query_indexed_multi). Results identical to epnet_ball_query per scale. */
int epnet_ball_query_ordered(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz,
                             const float *xyz, const void *index, size_t index_bytes, const void *centre_index,
                             size_t centre_index_bytes, int *const *idx, epnet_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * iou3d (lib/utils/iou3d/src/iou3d.cpp:174-179); boxes are (N,5) [x1,y1,x2,y2,ry] f32
 * -------------------------------------------------------------------------------------- */

/* boxesoverlapLauncher, iou3d_kernel.cu:354-363: ans (num_a,num_b) rotated-rectangle overlap area */
int epnet_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                            float *ans_overlap, epnet_stream_t stream);

/* boxesioubevLauncher, iou3d_kernel.cu:365-372: ans (num_a,num_b) rotated BEV IoU */
int epnet_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                        float *ans_iou, epnet_stream_t stream);

/* boxes_iou3d_gpu of lib/utils/iou3d/iou3d_utils.py:21-53 in one launch: boxes (.,7) [x,y,z,h,w,l,ry] ->
 * ans (num_a,num_b) 3-D IoU = BEV overlap x height overlap / union volume (clamped at 1e-7). The reference
 * composes boxes3d_to_bev_torch + boxes_overlap_bev_gpu + ~10 elementwise torch kernels; same fp32 operations
 * in the same order here. */
int epnet_boxes_iou3d(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_iou3d,
                      epnet_stream_t stream);

/* the same for k (a_i, b_i) PAIRS -> ans (k): one launch for what lib/rpn/proposal_target_layer.py:220-247
 * does with up to 640 single-pair calls per scene */
int epnet_boxes_iou3d_pairs(int k, const float *boxes_a, const float *boxes_b, float *ans_iou3d, epnet_stream_t stream);

/* aug_roi_by_noise_torch, lib/rpn/proposal_target_layer.py:220-247, for all k sampled ROIs of
 * a scene in one launch. The reference's host loop per ROI -- `while temp_iou < pos_thresh and cnt < aug_times`:
 * coin (keep the ROI, p = 0.2) or random_aug_box3d (:250-275), single-pair boxes_iou3d_gpu, read the IoU back --
 * takes its random draws from the caller's tables instead of np.random / torch.rand, indexed [roi][try]:
 * keep_draw (k, aug_times) u8 (1 = keep the original box), noise (k, aug_times, 7) f32 = pos_shift[3], hwl_scale[3],
 * angle_rot (the noisy box is [xyz + shift, hwl * scale, ry + rot]). roi_boxes3d (k,7) is updated in place with the
 * box of the last try (:242), iou_of_rois (k) receives iou3d_src[k] when that try kept the ROI (or no try ran) and
 * the last IoU otherwise (:243-246). gt_boxes3d (k,7) = the ground-truth box assigned to each ROI. tries (k) i32 or
 * NULL: per-ROI try limit min(tries[i], aug_times) -- foreground ROIs get ROI_FG_AUG_TIMES tries and background ROIs
 * one (:164-176), so the ROIs of a whole batch go through one launch. */
int epnet_aug_roi_by_noise(int k, int aug_times, float pos_thresh, float *roi_boxes3d, const float *gt_boxes3d,
                           const float *iou3d_src, const int *tries, const unsigned char *keep_draw,
                           const float *noise, float *iou_of_rois, epnet_stream_t stream);

/* ProposalLayer.forward after the box decoding, lib/rpn/proposal_layer.py:34-55 with distance_based_proposal :58-119
 * (distance_based != 0: two bins 0 < z <= 40 and 40 < z <= 80 with 70 % / 30 % of the pre- and post-NMS budgets, the far bin
 * falling back to the near bin's next boxes when it is empty) or score_based_proposal :121-142 (one bin), for all b
 * scenes with no host synchronisation. proposals (b,n,7) decoded boxes, scores (b,n), order (b,n)
 * i64 = positions sorted by descending score (torch.sort, :35). rotated: RPN.NMS_TYPE 'rotate' (nms_gpu) or 'normal'
 * (nms_normal_gpu) (:104-107; score_based_proposal always uses the rotated one, :137). Results: ret_bbox3d
 * (b, post_nms_top_n, 7) and ret_scores (b, post_nms_top_n), zero-padded behind the kept boxes (:38-39, :52-54);
 * ret_count (b) i32 or NULL = kept boxes per scene. */
size_t epnet_rpn_proposals_workspace_bytes(int b, int distance_based, int pre_nms_top_n, int post_nms_top_n);
int epnet_rpn_proposals(int b, int n, const float *proposals, const float *scores, const int64_t *order,
                        int distance_based, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated,
                        void *workspace, size_t workspace_bytes, float *ret_bbox3d, float *ret_scores, int *ret_count,
                        epnet_stream_t stream);

/* bytes of device scratch epnet_nms / epnet_nms_normal need for `boxes_num` boxes */
size_t epnet_nms_workspace_bytes(int boxes_num);

/* nms_gpu, iou3d.cpp:73-120 (nmsLauncher iou3d_kernel.cu:374-379 + the host sweep :100-116).
 * boxes (N,5) sorted by descending score. The suppression bit-mask and the greedy sweep both
 * run on the device: keep (N) i64 device array receives the kept positions in increasing order,
 * *num_keep (device i32) their count. The caller copies those back if it wants the reference's
 * host-side `keep` (the Python shim does). */
int epnet_nms(const float *boxes, int boxes_num, float nms_overlap_thresh, void *workspace,
              size_t workspace_bytes, int64_t *keep, int *num_keep, epnet_stream_t stream);

/* nms_normal_gpu, iou3d.cpp:123-170 (axis-aligned IoU of the [x1,y1,x2,y2] part, iou3d_kernel.cu:295-303) */
int epnet_nms_normal(const float *boxes, int boxes_num, float nms_overlap_thresh, void *workspace,
                     size_t workspace_bytes, int64_t *keep, int *num_keep, epnet_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * roipool3d (lib/utils/roipool3d/src/roipool3d.cpp:198-203); boxes3d are (.,7) [x,y,z,h,w,l,ry]
 * -------------------------------------------------------------------------------------- */

size_t epnet_roipool3d_workspace_bytes(int batch_size, int boxes_num, int sampled_pts_num);

/* roipool3dLauncher, roipool3d_kernel.cu:209-237 (also serves forward_slow, :197-206: same result).
 * xyz (B,N,3), boxes3d (B,M,7), pts_feature (B,N,C) -> pooled_features (B,M,S,3+C) f32,
 * pooled_empty_flag (B,M) i32. Rows of empty boxes are left untouched and the flag of non-empty
 * boxes is left untouched, as in the reference (caller zero-fills both, roipool3d_utils.py:21-23). */
int epnet_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                    int sampled_pts_num, const float *xyz, const float *boxes3d,
                    const float *pts_feature, float *pooled_features, int *pooled_empty_flag,
                    void *workspace, size_t workspace_bytes, epnet_stream_t stream);

/* Host-memory ops: these are CPU ops in the reference itself (called from DataLoader worker
 * processes, lib/datasets/kitti_rcnn_dataset.py:672,767,811,1029,1157), not a fallback.
 * pts_in_boxes3d_cpu, roipool3d.cpp:97-125: pts (N,3), boxes3d (M,7) -> pts_flag (M,N) i64 */
int epnet_pts_in_boxes3d_host(int64_t *pts_flag, const float *pts, const float *boxes3d,
                              int64_t boxes_num, int64_t pts_num);

/* roipool3d_cpu, roipool3d.cpp:127-195: -> pooled_pts (M,S,3), pooled_features (M,S,C), flag (M) i64 */
int epnet_roipool3d_host(const float *pts, const float *boxes3d, const float *pts_feature,
                         float *pooled_pts, float *pooled_features, int64_t *pooled_empty_flag,
                         int64_t boxes_num, int64_t pts_num, int64_t feature_len,
                         int64_t sampled_pts_num);

#ifdef __cplusplus
}
#endif
#endif /* EPNET_OPS_H */
