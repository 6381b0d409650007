/* hvpr_cpu.h — C-ABI of libhvpr_cpu.so: the CPU natives of the reference's GT-sampling augmentation ("next" row f4 of
 * SURVEY.md §8f).  Host code for data-loader workers, no GPU.  Replaces the absent
 *   pcdet/ops/roiaware_pool3d  points_in_boxes_cpu   (call sites pcdet/utils/box_utils.py:85, pcdet/datasets/kitti/kitti_dataset.py:217)
 *   pcdet/ops/iou3d_nms        boxes_bev_iou_cpu     (call sites pcdet/datasets/augmentor/database_sampler.py:184-185)
 * Boxes are (x, y, z, dx, dy, dz, heading) float32 rows, (x, y, z) the centre.  Return 0, or -1 on invalid arguments. */
#ifndef HVPR_CPU_H
#define HVPR_CPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* out [n_boxes, n_points] int32: 1 when |z - cz| <= dz/2 and, in the box frame, |x| < dx/2 and |y| < dy/2 */
int hvpr_points_in_boxes_cpu(const float *points, int n_points, int point_stride, const float *boxes, int n_boxes, int32_t *out);
/* out [n, m] float32: rotated BEV IoU */
int hvpr_boxes_bev_iou_cpu(const float *boxes_a, int n, const float *boxes_b, int m, float *out);
#ifdef __cplusplus
}
#endif
#endif
