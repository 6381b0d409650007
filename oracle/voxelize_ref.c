/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (hvpr_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / reported CPU baseline.
 *
 * CPU restatement of the point -> voxel generator the reference calls on its
 * dataloader workers:
 *   call site   pcdet/datasets/processor/data_processor.py:43-75
 *   algorithm   third-party spconv v1.x `VoxelGeneratorV2.generate` /
 *               `points_to_voxel` (traveller59/spconv; un-vendored, un-pinned:
 *               setup.py:41, README.md:26-27) — absent from /root/reference.
 *   in-tree corroboration: the structurally identical numba loop in
 *               tools/vis.py:9-60 (fp32 floor((p-lo)/vs) :37, bounds test
 *               :38-40, reversed zyx coordinate :41, first-touch map :44-50,
 *               `break` at max_voxels :47-48).
 * PINNED (voxel index part) by fixture G13, tests/golden/g13_voxel_index.npz =
 * the reference's own tools/vis.py:9-60 executed as plain Python by
 * tests/golden/make_golden.py (numba.jit -> identity): cell -> voxel-id order,
 * the range-border tests, the V1 stop at max_voxels and the per-voxel point
 * counts are checked in tests/test_oracle_golden.py (this file, mode 1) and on
 * the GPU in tests/test_gpu_stage1.py.  NOT pinned by any reference-held
 * vector (spconv is absent): the V2 `continue` at the
 * cap and the order of the <= max_points points stored inside a voxel — both
 * restated from the published spconv algorithm.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* mode 0: V2 semantics — a point that would open voxel number max_voxels is
 *         skipped, later points of already-open voxels are still accepted.
 * mode 1: V1 / tools/vis.py semantics — the loop stops at that point.
 * map:   caller-provided int32[nz*ny*nx], every entry -1 on entry; restored to
 *        -1 on exit (as the spconv generator does for the cells it touched).
 * Returns the number of voxels produced.                                       */
int hvpr_oracle_voxelize(const float *points, int n_points, int n_feat,
                         const float *range_lo /*[3] x,y,z*/, const float *voxel_size /*[3]*/,
                         const int *grid /*[3] nx,ny,nz*/, int max_points, int max_voxels,
                         int mode, float *voxels /*[max_voxels,max_points,n_feat] zeroed*/,
                         int *coords /*[max_voxels,3] z,y,x*/, int *num_points /*[max_voxels] zeroed*/,
                         int *map)
{
    int voxel_num = 0;
    const int nx = grid[0], ny = grid[1];
    for (int i = 0; i < n_points; ++i) {
        const float *p = points + (size_t)i * n_feat;
        int c3[3];
        int dropped = 0;
        for (int j = 0; j < 3; ++j) {
            /* one IEEE fp32 subtract, one IEEE fp32 divide, floor — in the points' dtype */
            volatile float d = p[j] - range_lo[j];
            volatile float q = d / voxel_size[j];
            float c = floorf(q);
            if (c < 0.0f || c >= (float)grid[j]) { dropped = 1; break; }
            c3[j] = (int)c;
        }
        if (dropped) continue;
        const int cell = (c3[2] * ny + c3[1]) * nx + c3[0];
        int vid = map[cell];
        if (vid == -1) {
            if (voxel_num >= max_voxels) {
                if (mode == 1) break;
                continue;
            }
            vid = voxel_num++;
            map[cell] = vid;
            coords[vid * 3 + 0] = c3[2];
            coords[vid * 3 + 1] = c3[1];
            coords[vid * 3 + 2] = c3[0];
        }
        const int k = num_points[vid];
        if (k < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + k) * n_feat, p, sizeof(float) * n_feat);
            num_points[vid] = k + 1;
        }
    }
    for (int v = 0; v < voxel_num; ++v) {
        const int cell = (coords[v * 3] * ny + coords[v * 3 + 1]) * nx + coords[v * 3 + 2];
        map[cell] = -1;
    }
    return voxel_num;
}
