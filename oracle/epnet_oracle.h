/* epnet_oracle.h -- TEST INFRASTRUCTURE. Prototypes of the CPU restatement in epnet_oracle.c
 * (see that file's header for the citation and parity rules). */
#ifndef EPNET_ORACLE_H
#define EPNET_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
int oracle_opt_n_threads(int work_size);
void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs);
void oracle_gather_points(int b, int c, int n, int m, const float *points, const int *idx, float *out);
void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points);
void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx);
void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out);
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx, float *grad_points);
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx);
void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out);
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points);
float oracle_box_overlap(const float *box_a, const float *box_b);
float oracle_iou_bev(const float *box_a, const float *box_b);
float oracle_iou_normal(const float *a, const float *b);
void oracle_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans);
void oracle_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans);
void oracle_nms_mask(int boxes_num, float thresh, const float *boxes, unsigned long long *mask, int rotated);
int oracle_nms_sweep(int boxes_num, const unsigned long long *mask, long long *keep);
int oracle_nms(int boxes_num, float thresh, const float *boxes, long long *keep, int rotated);
int oracle_pt_in_box3d(float x, float y, float z, float cx, float bottom_y, float cz, float h, float w, float l, float angle);
void oracle_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                      const float *xyz, const float *boxes3d, const float *pts_feature, float *pooled_features,
                      int *pooled_empty_flag);
void oracle_pts_in_boxes3d(long long *pts_flag, const float *pts, const float *boxes3d, long boxes_num, long pts_num);
void oracle_roipool3d_cpu(const float *pts, const float *boxes3d, const float *pts_feature, float *pooled_pts,
                          float *pooled_features, long long *pooled_empty_flag, long boxes_num, long pts_num,
                          long feature_len, long sampled_pts_num);
#ifdef __cplusplus
}
#endif
#endif
