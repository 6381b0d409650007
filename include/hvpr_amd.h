/*
 * hvpr_amd — C-ABI of the MI355X (gfx950) hot path of HVPR's voxel-point encoding + BEV detection.
 *
 * Every entry point
 *   - takes plain device pointers + sizes and a stream (hipStream_t passed as void*); no torch types;
 *   - enqueues on that stream and returns without synchronising;
 *   - never allocates, frees or retains memory: inputs, outputs and workspaces are the caller's
 *     (sizes from the matching *_workspace_bytes function);
 *   - returns 0 on success, a negative hvpr_status otherwise (see hvpr_status_string); nothing throws.
 *   - data-dependent sizes (voxel counts, NMS keep counts) are written to device int32 words.
 *
 * Each function cites the reference interface it replaces (paths relative to the reference repo
 * cvlab-yonsei/HVPR).  The reference's own native boundary for these was three pybind11 torch
 * extensions built by setup.py:52-110 (iou3d_nms_cuda, pointnet2_batch_cuda, pointnet2_stack_cuda)
 * plus the external spconv voxel generator; their sources are not part of the reference snapshot.
 */
#ifndef HVPR_AMD_H
#define HVPR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *hvpr_stream_t; /* hipStream_t */

enum hvpr_status {
    HVPR_OK = 0,
    HVPR_ERR_INVALID_ARG = -1,   /* null pointer, negative size, inconsistent dims            */
    HVPR_ERR_UNSUPPORTED = -2,   /* a shape the kernels are not built for (documented per op) */
    HVPR_ERR_WORKSPACE = -3,     /* workspace too small                                        */
    HVPR_ERR_LAUNCH = -4,        /* hipGetLastError() != hipSuccess after the launch           */
    HVPR_ERR_TIMEOUT = -5        /* hvpr_voxelize_workspace_status: a one-launch index kernel gave up a wait (workspace needs a reset) */
};

int hvpr_abi_version(void);     /* 5 (history: csrc/abi.hip); size every workspace / packed buffer with the *_bytes / *_floats functions */
const char *hvpr_status_string(int status);

/* SyncBatchNorm across ranks (reference: tools/train.py:119-120, --sync_bn -> torch.nn.SyncBatchNorm).  The training entry points
 * that hold a whole BatchNorm inside ONE call — hvpr_pillar_vfe_train_fwd_f32 / hvpr_pillar_vfe_bwd_f32 (two BatchNorm1d) and
 * hvpr_spatial_gate_train_fwd_f32 / _bwd_f32 (one BatchNorm2d) — hand their per-rank sums to this hook between two of their
 * launches: fn must replace buf[0..n) (device doubles; buf[0] is the element count, then sum x, sum x^2 forward or sum dy,
 * sum dy xhat backward) by the sum over all ranks, enqueued so that work submitted to `stream` afterwards sees the result; it
 * returns 0 on success (anything else makes the entry point return HVPR_ERR_LAUNCH).  With the hook set the statistics, and the
 * input gradient, are those of the global batch; parameter gradients stay per-rank sums (the data-parallel wrapper reduces them),
 * as in torch.nn.SyncBatchNorm.  fn == NULL (the default) restores per-rank statistics.  This one pointer pair is the only state
 * the library keeps between calls; set it before the first training call, from one thread.
 * (The BatchNorms of hvpr_bn_* take the other route: their sums / apply halves are separate entry points.) */
typedef int (*hvpr_allreduce_fn)(double *buf, int n, hvpr_stream_t stream, void *ctx);
void hvpr_set_batchnorm_allreduce(hvpr_allreduce_fn fn, void *ctx);

/* ---------------------------------------------------------------------------------------------
 * a1  Voxelizer.  Replaces spconv.utils.VoxelGenerator[V2].generate as called from
 *     pcdet/datasets/processor/data_processor.py:43-75 (CPU dataloader in the reference).
 *     Bit-exact first-touch semantics: voxel id = order of first appearance in the point array,
 *     first `max_points` points of a voxel kept in point order, coords = floor((p-lo)/vs) in fp32.
 *
 *     points        [n_points, point_stride] f32; xyz at columns xyz_col..xyz_col+2; the n_feat
 *                   columns starting at xyz_col are copied into the voxels (n_feat >= 3).
 *     frame_offsets [batch+1] i32 device; frame b owns points [frame_offsets[b], frame_offsets[b+1]).
 *     cap_mode      0 = VoxelGeneratorV2 (`continue` once max_voxels is reached),
 *                   1 = VoxelGenerator v1 / tools/vis.py:47-48 (`break`).
 *     voxels        [capacity, max_points, n_feat] f32 out, zero padded.
 *     coords        [capacity, 4] i32 out  [b, z, y, x]   (dataset.py:161-166 prepends b)
 *     num_points    [capacity] i32 out
 *     voxel_offsets [batch+1] i32 out; frame b produced rows [voxel_offsets[b], voxel_offsets[b+1]).
 *     capacity      rows available in the three outputs; >= sum_b min(N_b, max_voxels) is always enough.
 *     max_points <= 63.
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_voxelize_workspace_bytes(int max_batch, int max_points, int nx, int ny, int nz);
/* one-time (and after any failed call): puts the workspace in its idle state.  A workspace sized and reset for
 * (max_batch, max_points) serves every call with batch <= max_batch and n_points <= max_points on the same grid;
 * each call returns it to idle. */
int hvpr_voxelize_workspace_reset(void *workspace, size_t workspace_bytes, int max_batch, int max_points, int nx, int ny,
                                  int nz, hvpr_stream_t stream);
/* SYNCHRONISES `stream` and reads the workspace's error word (dimensions as it was sized with): HVPR_OK, or HVPR_ERR_TIMEOUT when a
 * one-launch index kernel (hvpr_encode_fwd_f32, index_mode 1) gave up a wait since the last reset — that call and every later
 * mode-1 call on this workspace reported zero pillars; reset the workspace before using it again.  Call it where the caller
 * synchronises anyway (the detector does when it reads its results back). */
int hvpr_voxelize_workspace_status(const void *workspace, size_t workspace_bytes, int max_batch, int max_points, int nx, int ny,
                                   int nz, hvpr_stream_t stream);
int hvpr_voxelize_f32(const float *points, int n_points, int point_stride, int xyz_col, int n_feat,
                      const int32_t *frame_offsets, int batch, float lo_x, float lo_y, float lo_z, float vs_x,
                      float vs_y, float vs_z, int nx, int ny, int nz, int max_points, int max_voxels, int cap_mode,
                      float *voxels, int32_t *coords, int32_t *num_points, int32_t *voxel_offsets, int capacity,
                      void *workspace, size_t workspace_bytes, int ws_max_batch, int ws_max_points,
                      hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a2  Pillar VFE, eval mode (BatchNorm folded by the caller).  Replaces
 *     PillarVFE_Scale.forward, pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221 (+ PFNLayer :29-49)
 *     for USE_ABSLOTE_XYZ=True, WITH_DISTANCE=False, NUM_FILTERS=[32,64], NUM_SCALE_FEATURES=[16,32],
 *     4 raw point features.
 *
 *     voxels [M, P, 4] f32 zero padded (P <= 32), num_points [M] i32, coords [M,4] i32 [b,z,y,x].
 *     w0 [16,10] b0 [16]: folded Linear(10->16)+BN;  w1 [64,32] b1 [64]: folded Linear(32->64)+BN;
 *     ws0 [16,5] bs0 [16], ws1 [32,16] bs1 [32]: the folded scale-stream layers.
 *     pillar_features [M,64], pillar_scale_features [M,32], pillar_mask [M,P] (may be NULL) f32 out.
 *     m_device: optional device i32 holding the live row count (<= M); NULL means M rows.
 * ------------------------------------------------------------------------------------------- */
int hvpr_pillar_vfe_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P,
                            const int32_t *m_device, float vs_x, float vs_y, float vs_z, float off_x, float off_y,
                            float off_z, const float *w0, const float *b0, const float *w1, const float *b1,
                            const float *ws0, const float *bs0, const float *ws1, const float *bs1,
                            float *pillar_features, float *pillar_scale_features, float *pillar_mask,
                            hvpr_stream_t stream);

/* a2 (training)  The two PFN layers of PillarVFE_Scale with BATCH-statistics BatchNorm (pillar_vfe.py:184-221, PFNLayer :29-49:
 *     linear -> BatchNorm1d over all M * P slots, padded ones included -> ReLU -> max over the slots -> concat), forward and
 *     backward, for the same fixed shapes as hvpr_pillar_vfe_fwd_f32 (P == 32).  No (M, P, C) tensor is materialised: every pass
 *     recomputes from the voxels (three forward passes, two backward passes + two for the statistics).
 *     w0 [16,10], gamma0 / beta0 [16], w1 [64,32], gamma1 / beta1 [64]: the Linear weights and BatchNorm affine parameters;
 *     eps: BatchNorm's; vs_* / off_*: voxel size and (voxel_size / 2 + range minimum) per axis as in the eval entry point.
 *     fwd: pillar_features [M,64]; mean0 / var0 [16], mean1 / var1 [64] = batch mean and BIASED variance of both layers (for the
 *          caller's running statistics).
 *     bwd: d_pillar_features [M,64] -> dw0 [16,10], dgamma0, dbeta0 [16], dw1 [64,32], dgamma1, dbeta1 [64] (overwritten; the
 *          batch statistics are recomputed, the workspace carries no state between calls).  The voxels get no gradient.
 *     Deterministic (fixed-order sums, finished in double).  workspace: hvpr_pillar_vfe_train_workspace_bytes(). */
size_t hvpr_pillar_vfe_train_workspace_bytes(void);
int hvpr_pillar_vfe_train_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, const float *w0,
                                  const float *gamma0, const float *beta0, const float *w1, const float *gamma1, const float *beta1,
                                  float eps, float vs_x, float vs_y, float vs_z, float off_x, float off_y, float off_z,
                                  float *pillar_features, float *mean0, float *var0, float *mean1, float *var1, void *workspace,
                                  size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_pillar_vfe_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, const float *w0,
                            const float *gamma0, const float *beta0, const float *w1, const float *gamma1, const float *beta1, float eps,
                            float vs_x, float vs_y, float vs_z, float off_x, float off_y, float off_z, const float *d_pillar_features,
                            float *dw0, float *dgamma0, float *dbeta0, float *dw1, float *dgamma1, float *dbeta1, void *workspace,
                            size_t workspace_bytes, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a3  Memory read-out, eval branch.  Replaces MemoryUnit_Agg.forward (eval),
 *     pcdet/models/backbones_2d/map_to_bev/memory_module.py:60-77: logits = f.W^T, top-k items,
 *     softmax over the k selected logits, weighted sum of the k items.
 *     f [M,64], bank [n_items,64] -> out [M,64]; topk_idx [M,k] i32 (may be NULL; the k selected ids, order unspecified).
 *     k <= 32, channels == 64.
 *     bank_packed (REQUIRED): the output of hvpr_memory_bank_pack_f32 for the same bank (only that function may produce it) — IEEE fp16 tiles in the matrix-core
 *     operand layout + the 64 channel maxima max_j |bank[j][c]|; pack once per weight update.  The kernel pre-filters on the
 *     fp16 matrix cores (v_mfma_f32_16x16x32_f16) with a rigorous rounding-error bound and re-checks the ~30 surviving candidates per row in exact
 *     fp32 from `bank` (logit = butterfly-tree sum of the 64 fp32 products), so the selected ids are the exact fp32 top-k
 *     (value descending, id ascending on ties); the k selected rows are read from `bank` too.
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_memory_bank_packed_floats(int n_items);   /* size of the packed copy in floats: ceil(n_items/16) * 512 + 128 */
int hvpr_memory_bank_pack_f32(const float *bank, int n_items, float *packed, hvpr_stream_t stream);
int hvpr_memory_readout_fwd_f32(const float *f, int M, const int32_t *m_device, const float *bank, const float *bank_packed,
                                int n_items, int k, float *out, int32_t *topk_idx, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a4  Scatter to the dense BEV canvases.  Replaces the eval branch of
 *     PointPillarScatter_Agg_Memory_1_scale.forward, map_to_bev/pointpillar_scatter.py:169-222
 *     (and plain PointPillarScatter :5-37 when memory/scale inputs are NULL).
 *     The canvases are written channels-last: spatial [B, ny, nx, c_pillar + c_mem],
 *     spatial_scale [B, ny, nx, c_scale] — logically (B, C, ny, nx) tensors in torch's
 *     channels_last memory format.  Every element is written exactly once (zeros included).
 *     coords [M,4] i32 [b,z,y,x]; cell_map: workspace of B*ny*nx i32.
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_scatter_workspace_bytes(int batch, int nx, int ny);
int hvpr_scatter_bev_fwd_f32(const float *pillar_features, int c_pillar, const float *memory_features, int c_mem,
                             const float *scale_features, int c_scale, const int32_t *coords, int M,
                             const int32_t *m_device, int batch, int nx, int ny, float *spatial,
                             float *spatial_scale, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);

/* a3+a4 fused, the form PointPillarScatter_Agg_Memory_1_scale.forward (eval, pointpillar_scatter.py:169-222) uses: the
 *     memory read-out also fills the scatter cell map (one launch less than the two calls above, same results).
 *     Fixed channel counts 64 (pillar) + 64 (memory) + 32 (scale).  memory_features [M,64] is an output. */
int hvpr_memory_scatter_fwd_f32(const float *pillar_features, const float *scale_features, const int32_t *coords, int M,
                                const int32_t *m_device, const float *bank, const float *bank_packed, int n_items, int k, int batch,
                                int nx, int ny,
                                float *memory_features, float *spatial, float *spatial_scale, void *workspace,
                                size_t workspace_bytes, hvpr_stream_t stream);

/* a1..a4 fused — points to BEV canvases in five launches (three with index_mode 1), the form the detector's eval forward uses when it is handed raw
 *     points.  Same results, bit for bit, as hvpr_voxelize_f32 -> hvpr_pillar_vfe_fwd_f32 -> hvpr_memory_scatter_fwd_f32
 *     (data_processor.py:43-75, pillar_vfe.py:184-221, memory_module.py:60-77, pointpillar_scatter.py:169-222) with
 *       - the voxel gather fused into the VFE (the padded `voxels` tensor becomes an optional OUTPUT, may be NULL),
 *       - the pillar / scale / memory cells of the canvases written by the VFE and the read-out themselves, and every other
 *         cell cleared by extra workgroups of the (latency-bound) VFE launch: no scatter pass, no cell map.
 *     Arguments as in the three separate calls; n_feat must be 4, nz 1, max_points <= 32, channels 64 + 64 + 32.
 *     voxel_offsets[batch] is the live pillar count M (device word); rows >= M of the per-pillar outputs are unspecified.
 *     capacity below M truncates: the first `capacity` rows are written, the canvas cells of the dropped pillars stay zero.
 *     index_mode selects the launch form of the index phase (cell keys -> voxel ranks -> arena), same results either way:
 *       0  three launches (K1 keys, K2 rank scan, K3 arena fill).  No workgroup waits for another one beyond the scan's look-back:
 *          safe for any number of calls in flight on any number of streams, and for several processes sharing a GPU.  The default
 *          every C caller should use unless it has read the next paragraph.
 *       1  ONE launch (up to 32 768 points; beyond that mode 0 is taken) whose at most 32 owner workgroups meet at two grid
 *          barriers on the compute units of one XCD — 2.6 us less per hvpr_car frame.  Owners that are resident wait for the ones
 *          that are not, so two such launches in flight at once could each hold compute units the other one needs.  The library
 *          rules that out inside a process: it takes the one-launch form only if the previous one-launch kernel on this device
 *          went to the same stream or has completed (one event query), and falls back to the three launches otherwise.  It
 *          cannot see other processes, and it cannot decide anything inside a stream capture (a captured call with index_mode 1
 *          always holds the one-launch form: replay such graphs one at a time per device — the detector's frame pipeline has one
 *          encode lane).  Every wait inside the kernel is bounded (2 s): a launch that could not complete raises a sticky error word
 *          in the workspace, reports zero pillars (voxel_offsets all 0; every other output of that call is unspecified), leaves the
 *          barrier words idle, and so does every later mode-1 call on that workspace until hvpr_voxelize_workspace_reset;
 *          hvpr_voxelize_workspace_status reads the word — check it wherever results are read back.
 *     pillar_mask may be NULL.  workspace: hvpr_voxelize_workspace_bytes / _reset, as for hvpr_voxelize_f32.
 *     Weight / bias pointers 16-byte aligned.
 *     canvas_state (may be NULL): [batch * ny * nx] bytes that travel with ONE pair of canvases the caller keeps between calls.
 *     NULL: every element of both canvases is written (zeros included), the canvases may hold anything on entry.
 *     Non-NULL: 1 = the cell holds a pillar of the previous call, and the caller guarantees the canvases are zero wherever it
 *     says 0 (start: canvases and state all zero).  Only the stale cells are then cleared (~2.4 MB instead of 47 MB per
 *     hvpr_car frame) and the state is updated; results are the same dense canvases. */
int hvpr_encode_fwd_f32(const float *points, int n_points, int point_stride, int xyz_col, int n_feat,
                        const int32_t *frame_offsets, int batch, float lo_x, float lo_y, float lo_z, float vs_x, float vs_y,
                        float vs_z, int nx, int ny, int nz, int max_points, int max_voxels, int cap_mode, float off_x,
                        float off_y, float off_z, const float *w0, const float *b0, const float *w1, const float *b1,
                        const float *ws0, const float *bs0, const float *ws1, const float *bs1, const float *bank,
                        const float *bank_packed, int n_items, int k, float *voxels, int32_t *coords, int32_t *num_points, int32_t *voxel_offsets,
                        int capacity, float *pillar_features, float *pillar_scale_features, float *pillar_mask,
                        float *memory_features, float *spatial, float *spatial_scale, uint8_t *canvas_state, void *workspace,
                        size_t workspace_bytes, int ws_max_batch, int ws_max_points, int index_mode, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5/a6  BEV backbone + head convolutions: implicit GEMM on the fp32 matrix cores, NHWC.
 *     Replaces the cuDNN convolutions behind BaseBEVBackbone_Scale.forward (eval),
 *     pcdet/models/backbones_2d/base_bev_backbone.py:280-315 — ZeroPad2d(1)+Conv3x3(s)+BN+ReLU (:154-169),
 *     the shared SFM step x_att = gate*ReLU(BN(conv(x_att))) + x_att (:291-295), ConvTranspose2d(k=s)+BN+ReLU
 *     (:177-188) written into its slice of the concat (:303-304) — and the three 1x1 head convolutions of
 *     AnchorHeadSingle.forward, pcdet/models/dense_heads/anchor_head_single.py:112-121.
 *
 *     in        [N, H, W, Cin] f32 (Cin % 8 == 0)
 *     w_packed  [taps, Cin/8, 2, cout_pad, 4] f32 (input channel = 8*chunk + 4*half + i), BatchNorm scale folded in;
 *               taps = 9 (3x3, pad 1) or 1 (1x1).
 *               For up > 1 (ConvTranspose2d with kernel == stride == up) taps = 1 and the gemm column is
 *               (ky*up + kx)*cout + co.
 *     bias      [cout_pad] f32 by gemm column (folded BatchNorm shift or the conv bias)
 *     gate/resid  both NULL, or gate [N, OH, OW] and resid [N, OH, OW, resid_cstride]: y = gate*y + resid
 *     out       [N, OH*up, OW*up, out_cstride] f32, channels written at out_coff .. out_coff+cout
 *     tile_cfg  0: 128 px x 128 ch   1: 64 px x 64 ch   2: 128 px x 64 ch   (cout_pad % tile channels == 0)
 * ------------------------------------------------------------------------------------------- */
int hvpr_conv2d_nhwc_f32(const float *in, int N, int H, int W, int Cin, const float *w_packed, const float *bias,
                         int taps, int stride, int cout, int cout_pad, int up, int relu, const float *gate,
                         const float *resid, int resid_cstride, float *out, int out_cstride, int out_coff,
                         int tile_cfg, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5  SpatialAttention gate.  Replaces ChannelPool + ConvLayer(2->1, 3x3, bias) + BatchNorm + sigmoid of
 *     pcdet/models/backbones_2d/spatial_attention.py:47-62.  y [N,H,W,C] NHWC -> gate [N,H,W].
 *     w18: conv weight (1,2,3,3) flattened [max-channel 9 taps | mean-channel 9 taps]; bn_scale/bn_shift: folded BN.
 * ------------------------------------------------------------------------------------------- */
int hvpr_spatial_gate_f32(const float *y, int N, int H, int W, int C, const float *w18, float conv_bias, float bn_scale,
                          float bn_shift, float *gate, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a7  Anchors + box decode + direction fix + class sigmoid/max.  Replaces
 *     AnchorHeadTemplate.generate_predicted_boxes (dense_heads/anchor_head_template.py:293-340),
 *     ResidualCoder.decode_torch (utils/box_coder_utils.py:45-77), limit_period (utils/common_utils.py:20-23) and the
 *     sigmoid + max over classes of Detector3DTemplate.post_processing (detectors/detector3d_template.py:206-207,241-246).
 *     head [N,H,W, n_anchor*(n_class + 7 + n_dir_bins)] = [cls | box | dir] NHWC (output of the three 1x1 convs);
 *     x_shifts [W], y_shifts [H]: anchor centres exactly as AnchorGenerator's arange (anchor_generator.py:34-39);
 *     anchor_table [n_anchor,5] = z centre, dx, dy, dz, rotation.  Anchor id = (y*W + x)*n_anchor + a.
 *     Outputs: batch_cls_preds [N,A,n_class] (may be NULL), batch_box_preds [N,A,7], scores [N,A] (may be NULL),
 *     labels [N,A] i32, 1-based (may be NULL).
 * ------------------------------------------------------------------------------------------- */
int hvpr_head_decode_f32(const float *head, int N, int H, int W, int head_channels, int n_anchor, int n_class,
                         int n_dir_bins, const float *x_shifts, const float *y_shifts, const float *anchor_table,
                         float dir_offset, float dir_limit_offset, float period, float *batch_cls_preds,
                         float *batch_box_preds, float *scores, int32_t *labels, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a8  Score filter + top-k.  Replaces `scores >= thresh`, torch.topk and mask.nonzero() of
 *     class_agnostic_nms, pcdet/models/model_utils/model_nms_utils.py:8-16,22-24.
 *     scores [batch, n_scores]; order [batch, pre_max] i32 = ids sorted by (score desc, id asc);
 *     sorted_scores [batch, pre_max] (may be NULL); counts [batch] i32 = min(#passing, pre_max).
 *     use_thresh = 0 keeps every non-NaN score.  pre_max <= 8192.
 *     workspace: zero-filled by the caller ONCE; every call returns it zero-filled (no per-call memset).
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_score_topk_workspace_bytes(int batch, int n_scores);
int hvpr_score_topk_f32(const float *scores, int batch, int n_scores, float score_thresh, int use_thresh, int pre_max,
                        int32_t *order, float *sorted_scores, int32_t *counts, void *workspace, size_t workspace_bytes,
                        hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a8  Rotated-BEV NMS and pairwise overlaps.  Replaces the absent pcdet/ops/iou3d_nms natives (setup.py:53-62):
 *     nms_gpu (call site model_nms_utils.py:17-19), boxes_overlap_bev_gpu / boxes_iou_bev / boxes_iou3d_gpu
 *     (detectors/detector3d_template.py:298,303).  Boxes [x,y,z,dx,dy,dz,heading], rows `box_stride` floats apart.
 *     hvpr_nms_bev_f32: candidate i is boxes[order ? order[i] : i], candidates already in descending score;
 *       n_device (may be NULL) holds the live count <= n_max (<= 16384); IoU > thresh suppresses (strict);
 *       keep[0..keep_count) = kept candidate positions, or order[position] when map_through_order != 0;
 *       at most max_keep are produced (NMS_POST_MAXSIZE).
 *     hvpr_boxes_pairwise_f32: mode 0 BEV overlap area, 1 BEV IoU, 2 3D IoU; out [n, m]; rows 7 floats apart.
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_nms_workspace_bytes(int n_max);
int hvpr_nms_bev_f32(const float *boxes, int box_stride, const int32_t *order, const int32_t *n_device, int n_max,
                     float thresh, int max_keep, int map_through_order, int32_t *keep, int32_t *keep_count,
                     void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
/* tail of post_processing (detector3d_template.py:255-259): out_*[r] = *[keep[r]] for r < max_keep (boxes 7 columns; labels and
 * the selected ids widened to int64 as the reference returns them); keep[] rows past the live count must hold valid ids */
int hvpr_gather_predictions_f32(const float *boxes, int box_stride, const float *scores, const int32_t *labels,
                                const int32_t *keep, int max_keep, float *out_boxes, float *out_scores, int64_t *out_labels,
                                int64_t *out_selected, hvpr_stream_t stream);
int hvpr_boxes_pairwise_f32(const float *boxes_a, int n, const float *boxes_b, int m, int mode, float *out,
                            hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9 (training)  PointNet++ index ops.  Replace the absent natives of pcdet/ops/pointnet2/pointnet2_batch (setup.py:94-109)
 *     behind PointnetSAModuleMSG / PointnetFPModule (pcdet/models/backbones_3d/pointnet2_backbone.py:27-34,43-47,82,86-89).
 *     Indices only (no gradient); tie rule: lowest index.  Distances are fp32 (dx*dx + dy*dy) + dz*dz.
 *     hvpr_furthest_point_sample_f32: xyz [B,N,3] -> idx [B,npoint] i32, first pick = index 0, N <= 32768.
 *     hvpr_ball_query_f32: first nsample points (index order) with d2 < radius^2; the first hit pre-fills all slots; zeros
 *                          when there is none.  xyz [B,N,3], new_xyz [B,M,3] -> idx [B,M,nsample] i32.
 *     hvpr_three_nn_f32:   unknown [B,n,3], known [B,m,3] -> dist [B,n,3] (sqrt of d2, ascending), idx [B,n,3] i32.
 * ------------------------------------------------------------------------------------------- */
int hvpr_furthest_point_sample_f32(const float *xyz, int B, int N, int npoint, int32_t *idx, hvpr_stream_t stream);
int hvpr_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int M, float radius, int nsample, int32_t *idx,
                        hvpr_stream_t stream);
int hvpr_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist, int32_t *idx,
                      hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9 (training)  Differentiable gathers of the point stream, in the reference's channel-major layout.  Replace
 *     grouping_operation / gather_operation / three_interpolate (+ their backward) of the absent
 *     pcdet/ops/pointnet2/pointnet2_batch natives (setup.py:94-109; used by PointnetSAModuleMSG / PointnetFPModule,
 *     pcdet/models/backbones_3d/pointnet2_backbone.py:27-34,43-47,82,86-89).
 *     hvpr_group_points_f32:        features [B,C,N], idx [B,npoint,nsample] i32 -> out [B,C,npoint,nsample]
 *                                   (nsample == 1: gather_operation).
 *     hvpr_group_points_grad_f32:   grad_out [B,C,npoint,nsample] -> grad_features [B,C,N], overwritten (zeroed, then
 *                                   scatter-added: fp32 atomics, summation order unspecified).
 *     hvpr_three_interpolate_f32:   features [B,C,m], idx / weight [B,n,3] -> out [B,C,n] = (f0 w0 + f1 w1) + f2 w2.
 *     hvpr_three_interpolate_grad_f32: grad_out [B,C,n] -> grad_features [B,C,m], overwritten as above.
 * ------------------------------------------------------------------------------------------- */
int hvpr_group_points_f32(const float *features, const int32_t *idx, int B, int C, int N, int npoint, int nsample, float *out,
                          hvpr_stream_t stream);
int hvpr_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint, int nsample,
                               float *grad_features, hvpr_stream_t stream);
int hvpr_three_interpolate_f32(const float *features, const int32_t *idx, const float *weight, int B, int C, int m, int n,
                               float *out, hvpr_stream_t stream);
int hvpr_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx, const float *weight, int B, int C, int m, int n,
                                    float *grad_features, hvpr_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * a9 (training)  Row-layout data movement around the shared MLPs of the point stream (PointnetSAModuleMSG,
 *     pcdet/models/backbones_3d/pointnet2_backbone.py:27-34; PointnetFPModule :40-47,86-89).  The MLPs themselves (1x1 convolution
 *     + train-mode BatchNorm + ReLU per layer) run on hvpr_conv2d_nhwc_f32 / hvpr_conv2d_wgrad_nhwc_f32 / hvpr_bn_* over rows
 *     [samples, cpad]: channels contiguous, zero columns up to cpad (a multiple of 8).
 *     hvpr_group_rows_f32:       QueryAndGroup(use_xyz): xyz [B,N,3], features [B,N,C] (null when C == 0), new_xyz [B,npoint,3],
 *                                idx [B,npoint,nsample] i32 -> out [B*npoint*nsample, cpad], row = [xyz[idx] - new_xyz | features[idx] | 0].
 *     hvpr_group_rows_grad_f32:  grad_out [rows, cpad] -> grad_features [B,N,C], overwritten (zeroed + fp32 atomics).
 *     hvpr_max_samples_f32:      y [G, nsample, C] -> out [G, C] = max over the samples of a group, argmax [G, C] u8 (lowest sample
 *                                on ties); nsample <= 255.        hvpr_max_samples_grad_f32: grad_out [G,C] -> grad_y [G,nsample,C].
 *     hvpr_fp_rows_f32:          PointnetFPModule's input: known [B,m,C1], idx / weight [B,n,3], skip [B,n,C2] (null when C2 == 0)
 *                                -> out [B*n, cpad], row = [(k0 w0 + k1 w1) + k2 w2 | skip | 0].
 *     hvpr_fp_rows_grad_f32:     grad_out [B*n, cpad] -> grad_known [B,m,C1] (zeroed + fp32 atomics), grad_skip [B,n,C2] (may be null).
 * ------------------------------------------------------------------------------------------- */
int hvpr_group_rows_f32(const float *xyz, const float *features, const float *new_xyz, const int32_t *idx, int B, int N, int C,
                        int npoint, int nsample, int cpad, float *out, hvpr_stream_t stream);
int hvpr_group_rows_grad_f32(const float *grad_out, const int32_t *idx, int B, int N, int C, int npoint, int nsample, int cpad,
                             float *grad_features, hvpr_stream_t stream);
int hvpr_max_samples_f32(const float *y, long long G, int nsample, int C, float *out, uint8_t *argmax, hvpr_stream_t stream);
int hvpr_max_samples_grad_f32(const float *grad_out, const uint8_t *argmax, long long G, int nsample, int C, float *grad_y,
                              hvpr_stream_t stream);
int hvpr_fp_rows_f32(const float *known, const int32_t *idx, const float *weight, const float *skip, int B, int m, int n, int C1,
                     int C2, int cpad, float *out, hvpr_stream_t stream);
int hvpr_fp_rows_grad_f32(const float *grad_out, const int32_t *idx, const float *weight, int B, int m, int n, int C1, int C2,
                          int cpad, float *grad_known, float *grad_skip, hvpr_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * a11 (training)  SpatialAttention with BATCH statistics (pcdet/models/backbones_2d/spatial_attention.py:47-63): ChannelPool ->
 *     conv3x3 2 -> 1 (+ bias) -> BatchNorm2d(1) -> sigmoid, forward and backward on NHWC y [N,H,W,C] (C % 4 == 0).  Parameters are
 *     DEVICE pointers (w18 = conv weight [1,2,3,3] flattened, conv_bias / gamma / beta one float each).
 *     fwd: pooled [N,H,W,2] (max, mean), argmax [N,H,W] i32, a [N,H,W] (conv output), stats [3] = batch mean, biased variance,
 *          1/sqrt(var + eps) of a, gate [N,H,W].     bwd: dgate [N,H,W] -> dy [N,H,W,C] (overwritten), dw18 [18], dbias, dgamma,
 *          dbeta [1] each; the batch statistics are differentiated through.  Sums: per-workgroup fp32 partials, final sums in
 *          double in a fixed order (deterministic).
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_spatial_gate_train_workspace_bytes(int N, int H, int W);
int hvpr_spatial_gate_train_fwd_f32(const float *y, int N, int H, int W, int C, const float *w18, const float *conv_bias,
                                    const float *gamma, const float *beta, float eps, float *pooled, int32_t *argmax, float *a,
                                    float *stats, float *gate, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_spatial_gate_train_bwd_f32(const float *dgate, const float *gate, const float *a, const float *stats, const float *pooled,
                                    const int32_t *argmax, const float *w18, const float *gamma, int N, int H, int W, int C,
                                    float *dy, float *dw18, float *dbias, float *dgamma, float *dbeta, void *workspace,
                                    size_t workspace_bytes, hvpr_stream_t stream);
/* a10 (training)  the index half of get_score (pointpillar_scatter.py:70-75): for every pillar row of `pillars` [M,64] the k
 * rows of `points` [N,64] with the largest dot product, idx [M,k] i32 in DESCENDING order (ties: lower index first).  N is
 * unlimited (items are walked in blocks of 2048); points_packed = hvpr_memory_bank_pack_f32(points, N) (fp16 operand tiles +
 * channel maxima).  Exact fp32 top-k through the same pre-filter + re-check scheme as hvpr_memory_readout_fwd_f32. */
int hvpr_point_pillar_topk_f32(const float *pillars, int M, const float *points, const float *points_packed, int N, int k,
                               int32_t *idx, hvpr_stream_t stream);
/* a10 (training)  backward of the row gather `points[idx]` of get_score (pointpillar_scatter.py:76): dst [n_dst, row_floats] is
 * overwritten with the scatter-add of src [m, row_floats] at rows idx [m] (fp32 atomics; out-of-range ids are ignored).  The
 * forward is hvpr_gather_rows_f32. */
int hvpr_scatter_add_rows_f32(const float *src, const int32_t *idx, long long m, int row_floats, int n_dst, float *dst,
                              hvpr_stream_t stream);

/* a9 / a10 (training)  The scattering gradients WITHOUT atomics (a point belongs to many groups — backward of QueryAndGroup,
 * pointnet2_backbone.py:27-34; a known point feeds many unknown ones — backward of three_interpolate, :40-47; backward of the row
 * gather `points[idx]`, pointpillar_scatter.py:76): dst[d][c] = sum, over the edges e = rowptr[d] .. rowptr[d+1]-1 of destination d
 * and IN THAT ORDER, of edge_w[e] * src[edge_row[e] * src_stride + src_off + c], c < C (edge_w may be null: weights 1; edge_row
 * may be null: edge e reads row e).  One
 * sequential fp32 sum per output element: two runs give the same bits.  The caller sorts the edges by destination (stable) once;
 * dst [n_dst, dst_stride] is overwritten. */
int hvpr_segment_sum_rows_f32(const float *src, long long src_stride, int src_off, int C, const int32_t *edge_row,
                              const float *edge_w, const int32_t *rowptr, long long n_dst, float *dst, long long dst_stride,
                              hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a10 (training)  MemAE memory addressing with hard shrinkage, MemoryUnit_Agg.forward training branch,
 *     map_to_bev/memory_module.py:31-48 (+ hard_shrink_relu :85-87), without materialising the (R, n_items) attention:
 *         a = softmax(x W^T), s = relu(a - l) a / (|a - l| + 1e-12), t = s / max(||s||_1, 1e-12), y = t W
 *     x [R,64] (the k positive point features of every pillar, flattened), bank [n_items <= 2048, 64], shrink_thres l > 0.
 *     fwd: y [R,64]; row_stats [R,4] (softmax max, partition sum, ||s||_1) is what the backward needs.
 *     bwd: dx [R,64] and dbank [n_items,64] (overwritten) from dy [R,64]; row_scratch [2 R] floats; no atomics — every sum
 *          is formed in a fixed order (deterministic).
 *     workspace: hvpr_memory_train_workspace_bytes(n_items), no state between calls.
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_memory_train_workspace_bytes(int n_items);
int hvpr_memory_train_fwd_f32(const float *x, long long R, const float *bank, int n_items, float shrink_thres, float *y, float *row_stats,
                              void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_memory_train_bwd_f32(const float *x, const float *dy, long long R, const float *bank, int n_items, float shrink_thres,
                              const float *row_stats, float *dx, float *dbank, float *row_scratch, void *workspace,
                              size_t workspace_bytes, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a11 (training)  Two-stream BEV backbone, pcdet/models/backbones_2d/base_bev_backbone.py:228-279.
 *     Forward convolution and data gradient: hvpr_conv2d_nhwc_f32 (the data gradient of a stride-1 3x3 conv is the same conv
 *     on the flipped, transposed weights; stride 2 on the zero-upsampled gradient; ConvTranspose k = s as the 1x1 GEMM).
 *     hvpr_conv2d_wgrad_nhwc_f32: weight gradient on the fp32 matrix cores.  x [N,H,W,Cin] and dz [N,OH,OW,Cout] NHWC
 *         (OH = (H + 2 - 3) / stride + 1 for taps == 9 (3x3, pad 1), = H for taps == 1) -> dw [Cout, Cin, k, k] (torch layout),
 *         overwritten.  Split-K over pixel tiles into `workspace` partials, summed in a fixed order: deterministic.
 *     hvpr_bn_stats_nhwc_f32: per-channel batch mean, biased variance and 1/sqrt(var + eps) of z [P, C] (train-mode BatchNorm,
 *         double-precision final sums).
 *     hvpr_bn_train_affine_f32: what train-mode nn.BatchNorm2d does with the batch moments besides normalising, one launch:
 *         scale = gamma * invstd, shift = beta - mean * scale, and (running_mean / running_var both non-NULL)
 *         running = (1 - momentum) * running + momentum * mean | momentum_unbiased * var (momentum_unbiased = momentum * n / (n - 1)),
 *         *num_batches_tracked += 1 (int64, may be NULL).
 *     hvpr_bn_relu_fwd_nhwc_f32: y = max(0, z * scale + shift) (relu == 0: no max); scale = gamma * invstd, shift = beta - mean * scale.
 *     hvpr_bn_relu_bwd_nhwc_f32: dz, dgamma, dbeta of y = relu(gamma * (z - mean) * invstd + beta) with BATCH statistics
 *         (the mean / variance terms are differentiated through).
 *     gate [P] + resid [P,C] (both or neither; may be NULL): the SFM step fused in, y = gate[p] * relu(...) + resid
 *         (x_att = attention(sfm(x_att), y) + x_att, base_bev_backbone.py:250-255); the backward then also returns dgate [P]
 *         (= sum_c relu(...) * dy, overwritten) and propagates gate * dy; d resid = dy.
 *     C % 4 == 0, C <= 1024 for the reductions.
 * ------------------------------------------------------------------------------------------- */
/* hvpr_conv2d_wino_wgrad_nhwc_f32: the same weight gradient for 3x3 / stride 1 / pad 1 in the Winograd F(2x2,3x3) domain
 *     (dU = sum_blocks (A dY At) . (Bt d B), dw = Gt dU G): 16 instead of 36 fp32 multiplies per 2x2 block and (co, ci) pair;
 *     x [N,H,W,Cin], dz [N,H,W,Cout] -> dw [Cout,Cin,3,3], deterministic split-K like the direct form. */
size_t hvpr_conv2d_wino_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int hvpr_conv2d_wino_wgrad_nhwc_f32(const float *x, int N, int H, int W, int Cin, const float *dz, int Cout, float *dw,
                                    void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
size_t hvpr_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int taps, int stride);
int hvpr_conv2d_wgrad_nhwc_f32(const float *x, int N, int H, int W, int Cin, const float *dz, int Cout, int taps, int stride, float *dw,
                               void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
size_t hvpr_bn_workspace_bytes(long long P, int C);
int hvpr_bn_stats_nhwc_f32(const float *z, long long P, int C, float eps, float *mean, float *var, float *invstd, void *workspace,
                           size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_bn_train_affine_f32(const float *mean, const float *var, const float *invstd, int C, const float *gamma, const float *beta,
                             float momentum, float momentum_unbiased, float *running_mean, float *running_var,
                             long long *num_batches_tracked, float *scale, float *shift, hvpr_stream_t stream);
int hvpr_bn_relu_fwd_nhwc_f32(const float *z, long long P, int C, const float *scale, const float *shift, int relu, const float *gate,
                              const float *resid, float *y, hvpr_stream_t stream);
/* ... into / out of a channel slice [coff, coff + C) of a wider NHWC tensor with cstride channels per pixel (no gate): the backbone's
 * deconvolution branches write straight into the 384-channel concatenation and take their gradient out of its gradient. */
/* Data gradient of a 3x3 stride-2 (pad 1) convolution, gathered per output-pixel parity class (1 / 2 / 2 / 4 taps) instead of a
 * stride-1 convolution over a zero-upsampled gradient: dz [N, H, W, Cin] (the layer's output gradient; Cin = its output channels) ->
 * dx [N, OH, OW, out_cstride] channels [out_coff, out_coff + cout) (its input's gradient; H == (OH + 2 - 3) / 2 + 1, W likewise).
 * w_packed = hvpr_conv2d_nhwc_f32's weight image of the layer's filter with the channel axes swapped ((cout, Cin, 3, 3), taps NOT
 * flipped), bias [cout_pad] (zeros for a plain gradient), cout_pad a multiple of 64. */
int hvpr_conv2d_s2_dgrad_nhwc_f32(const float *dz, int N, int H, int W, int Cin, const float *w_packed, const float *bias, int cout,
                                  int cout_pad, int OH, int OW, float *dx, int out_cstride, int out_coff, hvpr_stream_t stream);
int hvpr_bn_relu_fwd_slice_nhwc_f32(const float *z, long long P, int C, const float *scale, const float *shift, int relu, float *y,
                                    int y_cstride, int y_coff, hvpr_stream_t stream);
int hvpr_bn_relu_bwd_slice_nhwc_f32(const float *dy, int dy_cstride, int dy_coff, const float *z, long long P, int C, const float *scale,
                                    const float *shift, const float *mean, const float *invstd, int relu, float *dz, float *dgamma,
                                    float *dbeta, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_bn_relu_bwd_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                              const float *mean, const float *invstd, int relu, const float *gate, float *dgate, float *dz, float *dgamma,
                              float *dbeta, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
/* The same backward in its two halves, for SyncBatchNorm (tools/train.py:119-120 converts every BatchNorm when --sync_bn is given):
 * _sums leaves the LOCAL d gamma = sum dy_m * xhat and d beta = sum dy_m (dy_m = dy through the ReLU / gate); the caller all-reduces
 * them over the ranks; _apply computes dz (and dgate) from the sums and 1 / count of the GLOBAL batch.  hvpr_bn_relu_bwd_nhwc_f32 is
 * _sums followed by _apply with the local sums and 1 / P. */
int hvpr_bn_relu_bwd_sums_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                                   const float *mean, const float *invstd, int relu, const float *gate, float *dgamma, float *dbeta,
                                   void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_bn_relu_bwd_apply_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                                    const float *mean, const float *invstd, int relu, const float *gate, float *dgate, float *dz,
                                    const float *dgamma_total, const float *dbeta_total, double inv_count, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a12 (training)  Anchor target assignment for ONE anchor set (one class of ANCHOR_GENERATOR_CONFIG) and all frames of the batch —
 *     AxisAlignedTargetAssigner.assign_targets_single (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:
 *     113-213; POS_FRACTION < 0: no sampling, NORM_BY_NUM_EXAMPLES false, MATCH_HEIGHT false, as hvpr.yaml:114-124) with
 *     boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:252-323) and ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:
 *     13-43).  Two launches, no host round trip (the reference: two `.cpu().numpy()` arg-maxes per frame, :148,:153).
 *     anchors        [n_anchors, 7] f32 of this set, order (z, y, x, size, rotation); per_loc of them per BEV location
 *     gt_boxes       [batch, n_gt, 8] f32 [x, y, z, dx, dy, dz, heading, class]; trailing all-zero rows are padding (:53-57);
 *                    a row takes part when class_names[class - 1] is this set's class (class_index, 0-based; python's negative
 *                    index for class 0, as the reference has it), n_gt <= 256
 *     outputs in the HEAD's anchor order: entry ((a / per_loc) * loc_stride + loc_offset + a % per_loc) of frame b, rows of
 *     anchors_total entries — loc_stride = anchors per location over all sets, loc_offset = this set's first one:
 *     labels [batch, anchors_total] i32 (-1 don't care, 0 background, class id), reg_targets [batch, anchors_total, 7],
 *     reg_weights [batch, anchors_total] (1 on positives), pos_count [batch] i32 += positives of this set (zero it before the first set).
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_assign_targets_workspace_bytes(int batch, int n_gt);
int hvpr_assign_targets_f32(const float *anchors, int n_anchors, const float *gt_boxes, int batch, int n_gt, int class_index,
                            int n_classes, float matched_thr, float unmatched_thr, int per_loc, int loc_stride, int loc_offset,
                            long long anchors_total, int32_t *labels, float *reg_targets, float *reg_weights, int32_t *pos_count,
                            void *workspace, size_t workspace_bytes, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a13 (training)  The three losses of ONE prediction stream of the anchor head AND their gradients w.r.t. the predictions, one launch
 *     + a fixed-order sum — get_cls_layer_loss / get_box_reg_layer_loss (pcdet/models/dense_heads/anchor_head_template.py:101-260):
 *     sigmoid focal loss (alpha, gamma = 2; pcdet/utils/loss_utils.py:9-72) with weights 1 / max(#positives of the frame, 1) on
 *     positives and negatives, smooth-L1 (beta) on the sin-difference-encoded residuals (:153-160; loss_utils.py:75-136, NaN targets
 *     ignored), cross entropy on the direction bin of the ground-truth heading (:162-176; loss_utils.py:181-206); each summed over
 *     the batch, / batch, x its LOSS_WEIGHTS entry.
 *     cls_preds [batch, n_anchors, num_class], box_preds [batch, n_anchors, 7], dir_preds [batch, n_anchors, num_dir_bins] or NULL
 *     (NHWC head outputs viewed per anchor); labels / reg_targets / pos_count from hvpr_assign_targets_f32; anchor_rot [n_anchors]
 *     the anchors' headings in the same order; code_weights: 7 floats in HOST memory.
 *     losses [3] (device): cls, loc, dir.  grad_* : d(loss_i)/d(pred), same shapes as the predictions (loss weights and 1 / batch
 *     included: a caller scales them by the upstream gradient of the respective scalar).  num_class <= 3, num_dir_bins <= 8.
 *
 *     hvpr_mse_loss_f32: get_mem_loss (:262-275) — mean((x - target)^2) / rows * weight over [rows, cols] matrices, and its gradient
 *     w.r.t. x (the target is a constant there: `target.detach()`, :268).
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_rpn_losses_workspace_bytes(int batch, long long n_anchors);
int hvpr_rpn_losses_f32(const float *cls_preds, const float *box_preds, const float *dir_preds, const int32_t *labels,
                        const float *reg_targets, const float *anchor_rot, const int32_t *pos_count, int batch, long long n_anchors,
                        int num_class, int num_dir_bins, float alpha, float gamma, float beta, const float *code_weights,
                        float cls_weight, float loc_weight, float dir_weight, float dir_offset, float *losses, float *grad_cls,
                        float *grad_box, float *grad_dir, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
size_t hvpr_mse_loss_workspace_bytes(void);
int hvpr_mse_loss_f32(const float *x, const float *target, long long rows, int cols, float weight, float *loss, float *grad_x,
                      void *workspace, size_t workspace_bytes, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a14 (training)  Optimiser step over ONE flat fp32 parameter buffer (and matching flat gradient / moment buffers, all
 *     16-byte aligned): decoupled weight decay p *= 1 - weight_decay * lr, then Adam with bias correction at `step` (1-based)
 *     — OptimWrapper.step, tools/train_utils/optimization/fastai_optim.py:132-149 (true_wd, the optimiser's own weight_decay
 *     forced to 0) around torch.optim.Adam(betas = (beta1, beta2)).  grad_scale_device (may be NULL): a device float every
 *     gradient is multiplied by first — the clip coefficient min(1, clip / (||g|| + 1e-6)) of
 *     clip_grad_norm_ (tools/train_utils/train_utils.py:41), so that clipping costs no extra pass and no host sync.
 * ------------------------------------------------------------------------------------------- */
int hvpr_fused_adam_truewd_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step,
                               const float *grad_scale_device, hvpr_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * a5 / a11  Stride-1 3x3 convolution (pad 1) by Winograd F(2x2, 3x3) on the fp32 matrix cores: the same operation as
 *     hvpr_conv2d_nhwc_f32 with taps = 9, stride = 1, up = 1 — BaseBEVBackbone_Scale's Conv3x3 + BN + ReLU and the SFM step
 *     (pcdet/models/backbones_2d/base_bev_backbone.py:228-315) — with 16 instead of 36 fp32 multiplies per 2x2 output block and
 *     (cin, cout) pair.  All arithmetic is fp32; the result differs from the direct kernel by summation order only
 *     (observed <= 2e-6 relative to the output scale per layer, tests/test_gpu_conv_wino.py).
 *
 *     hvpr_conv2d_wino_pack_f32: weight [cout, cin, 3, 3] f32 (torch OIHW), optional per-output-channel scale [cout] (folded
 *         BatchNorm), -> packed [hvpr_conv2d_wino_packed_floats(cin, cout)] f32 = U = G g Gt in the kernel's stage image
 *         [cout_pad/64][cin/8][16][2][64][4] (cout_pad = cout rounded up to 64; transform evaluated in double, rounded once).
 *         adjoint != 0: `weight` is the [cin, cout, 3, 3] filter of the layer whose DATA GRADIENT is wanted: packs
 *         w'[o][i][u][v] = weight[i][o][2-u][2-v].
 *     hvpr_conv2d_wino_nhwc_f32: in [N,H,W,Cin] (Cin % 8 == 0), bias [cout_pad], gate / resid / out / out_cstride / out_coff as
 *         in hvpr_conv2d_nhwc_f32 (cout % 4 == 0).  px_groups: 1 = 8 x 16 output pixels x 64 channels per workgroup (4 waves),
 *         2 = 16 x 16 pixels x 64 channels (8 waves sharing the filter stage), 4 = 8 x 16 pixels x 32 channels (4 waves; twice
 *         the tiles for launches that do not fill the chip).
 * ------------------------------------------------------------------------------------------- */
size_t hvpr_conv2d_wino_packed_floats(int cin, int cout);
int hvpr_conv2d_wino_pack_f32(const float *weight, const float *scale, int cout, int cin, int adjoint, float *packed,
                              hvpr_stream_t stream);
int hvpr_conv2d_wino_nhwc_f32(const float *in, int N, int H, int W, int Cin, const float *w_packed, const float *bias, int cout,
                              int relu, const float *gate, const float *resid, int resid_cstride, float *out, int out_cstride,
                              int out_coff, int px_groups, float *bn_partials, hvpr_stream_t stream);
/* bn_partials (optional, training): [hvpr_conv2d_wino_stats_rows(N, H, W)][2][cout] f32, overwritten with the per-pixel-tile sum
 * and sum of squares of the raw output (px_groups == 1, relu == 0, no gate) — the batch statistics of the layer's BatchNorm without
 * a pass over the written tensor; hvpr_bn_finalize_partials_f32 turns them into mean / biased variance / 1/sqrt(var + eps)
 * (count = N * H * W; sums finished in double, fixed order: deterministic). */
int hvpr_conv2d_wino_stats_rows(int N, int H, int W);
int hvpr_bn_finalize_partials_f32(const float *partials, int rows, int C, long long count, float eps, float *mean, float *var,
                                  float *invstd, hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5 optional precision modes: 3x3 convolutions on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation)
 *     with split operands.  n_planes = 2 ("bf16x3"): x = hi + lo, x*w ~= hi*hi + hi*lo + lo*hi, error ~2^-16 relative per
 *     product — SURVEY.md §8d allows a reduced-precision path evidenced within the 1e-3 tolerance.  n_planes = 3 ("bf16x6"):
 *     x = hi + mid + lo exactly, six products, dropped terms <= 2^-23 relative = the accuracy of the fp32 matrix-core kernel
 *     (fp32 emulation).  hvpr_conv2d_nhwc_f32 stays the parity reference.
 *     Split-bf16 NHWC: each group of 8 channels of a pixel is n_planes x [8 x bf16] (16 bytes per plane).
 *     hvpr_split_bf16_f32 / hvpr_unsplit_bf16_f32: fp32 NHWC (n_floats % 8 == 0) <-> split-bf16 NHWC.
 *     hvpr_conv2d_nhwc_bf16x3: in_split [N,H,W,Cin] split, w_split [9, Cin/8, n_planes, cout_pad] x 16 B, bias [cout_pad] f32;
 *     out fp32 NHWC (out_split = 0) or split NHWC (1) with out_cstride channels at channel offset out_coff; optional fused
 *     y = gate*y + resid with resid in split form.  Cin % 16 == 0, cout % 4 == 0, strides/offsets % 8 == 0.
 *     tile_cfg 0 = 128 px x 64 ch, 1 = 64 px x 64 ch, 2 = 256 px x 64 ch (stride 1, two planes); three planes: stride 1 only.
 * ------------------------------------------------------------------------------------------- */
int hvpr_split_bf16_f32(const float *src, long long n_floats, int n_planes, void *dst, hvpr_stream_t stream);
int hvpr_unsplit_bf16_f32(const void *src, long long n_floats, int n_planes, float *dst, hvpr_stream_t stream);
int hvpr_conv2d_nhwc_bf16x3(const void *in_split, int N, int H, int W, int Cin, const void *w_split, const float *bias,
                            int stride, int cout, int cout_pad, int relu, const float *gate, const void *resid_split,
                            int resid_cstride, void *out, int out_split, int out_cstride, int out_coff, int tile_cfg,
                            int n_planes, hvpr_stream_t stream);
/* ConvTranspose2d(kernel = stride = up)+BN+ReLU of the same modes (base_bev_backbone.py:177-188): split input, w_split
 * [1, Cin/8, n_planes, cout_pad] x 16 B with gemm column (ky*up + kx)*cout + co, fp32 output written into channels
 * [out_coff, out_coff + cout) of an NHWC tensor of out_cstride channels at up x the resolution.  Cin % 64 == 0. */
int hvpr_deconv_nhwc_bf16x3(const void *in_split, int N, int H, int W, int Cin, const void *w_split, const float *bias, int cout,
                            int cout_pad, int up, int relu, float *out, int out_cstride, int out_coff, int n_planes,
                            hvpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f2 ("next" row)  KITTI point pre-processing in front of the voxelizer, on the device.
 *     hvpr_point_flags_f32  mode 0: mask_points_by_range (pcdet/utils/common_utils.py:59-62: x,y in [lo,hi] inclusive;
 *                           range_xy = {x0,y0,x1,y1}) AND, when the two fov matrices are given, KittiDataset.get_fov_flag
 *                           (pcdet/datasets/kitti/kitti_dataset.py:100-116; fov_lidar_to_rect = (V2C^T R0^T) [4][3],
 *                           fov_rect_to_img = P2^T [4][3], both row-major float32 as calibration_kitti.py:65-84 forms them);
 *                           mode 1: the near flag of DataProcessor.sample_points (data_processor.py:89-90): |xyz| < near_thresh.
 *                           points [n, stride] f32 with x,y,z in columns 0..2; flags [n] u8.
 *     hvpr_compact_rows_f32 dst = src[flags != 0] (stable), *count = number of rows (clamped to capacity).
 *     hvpr_gather_rows_f32  dst[r] = src[idx[r]] (points[choice] of sample_points; idx i32, out-of-range rows are zeros).
 * ------------------------------------------------------------------------------------------- */
int hvpr_point_flags_f32(const float *points, int n, int stride, int mode, const float *range_xy, float near_thresh,
                         const float *fov_lidar_to_rect, const float *fov_rect_to_img, int img_h, int img_w, uint8_t *flags,
                         hvpr_stream_t stream);
size_t hvpr_compact_workspace_bytes(int n);
int hvpr_compact_rows_f32(const float *src, int n, int row_floats, const uint8_t *flags, float *dst, int capacity,
                          int32_t *count, void *workspace, size_t workspace_bytes, hvpr_stream_t stream);
int hvpr_gather_rows_f32(const float *src, int n_src, int row_floats, const int32_t *idx, int m, float *dst,
                         hvpr_stream_t stream);
/* frame_offsets [batch+1] i32 of a collated point array (dataset.py:161-166: column 0 = batch index, frames contiguous and
 * ascending): offsets[b] = first row of frame b (empty frames included), offsets[batch] = n_points. */
int hvpr_frame_offsets_f32(const float *points, int n_points, int point_stride, int batch, int32_t *frame_offsets,
                           hvpr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HVPR_AMD_H */
