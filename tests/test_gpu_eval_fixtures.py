"""GPU: the evaluation-path fixtures the reference's own modules generated (tests/golden/make_golden.py), DIRECTLY on the HIP path —
one hop, kernel -> fixture, where tests/test_gpu_e2e.py goes kernel -> oracle -> fixture:
  G4  BaseBEVBackbone_Scale eval (base_bev_backbone.py:280-315)                 -> product BaseBEVBackbone_Scale (conv_wino / conv_igemm / gate kernels)
  G5  the full-size hvpr_car anchor grid (anchor_generator.py:17-60)            -> product AnchorGenerator tables + hvpr_head_decode_f32
  G7  limit_period table (common_utils.py:20-23)                                -> hvpr_head_decode_f32's direction fix
  G6  ResidualCoder.encode_torch (box_coder_utils.py:13-43)                     -> hvpr_assign_targets_f32's encoder
(G1, G2, G3, G5-head: tests/test_gpu_stage1.py / test_gpu_post.py; G8-G10, G16, G17, G4-train: test_gpu_train_fixtures.py.)"""
import os

import numpy as np
import pytest
import torch

from detparams import det_state
from hvpr_amd import anchor_head, bev_backbone, kernels
from hvpr_amd.config import AttrDict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, rtol=1e-3):
    ref = np.asarray(ref)
    rms = float(np.sqrt(np.mean(np.square(ref, dtype=np.float64))))
    np.testing.assert_allclose(np.asarray(got), ref, rtol=rtol, atol=rtol * max(rms, 1e-30))


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x6"])
def test_g4_backbone_eval_on_the_hip_path(golden_dir, tag, precision, observed):
    z = np.load(os.path.join(golden_dir, f"g4_backbone_{tag}.npz"))
    shapes = {str(n): tuple(eval(str(s))) for n, s in zip(z["param_names"], z["param_shapes"])}
    filt = [shapes[f"blocks.{i}.1.weight"][0] for i in range(3)]
    sfilt = [shapes[f"scale_layers.{i}.1.weight"][0] for i in range(3)]
    if any(c % 8 for c in filt + sfilt):
        pytest.skip("the convolution kernels take channel counts that are multiples of 8 (hvpr.yaml: 32 ... 512); this fixture has a 12")
    if precision != "fp32" and any(c % 16 for c in filt + sfilt):
        pytest.skip("split-bf16 kernels: channel multiples of 16")
    cfg = AttrDict(LAYER_NUMS=[int(v) for v in z["layer_nums"]], SFM_LAYER_NUMS=[int(v) for v in z["sfm_layer_nums"]],
                   LAYER_STRIDES=[int(v) for v in z["layer_strides"]], NUM_FILTERS=filt, NUM_SCALE_FILTERS=sfilt,
                   UPSAMPLE_STRIDES=[int(v) for v in z["upsample_strides"]], NUM_UPSAMPLE_FILTERS=[filt[0]] * 3)
    m = bev_backbone.BaseBEVBackbone_Scale(model_cfg=cfg, input_channels=z["spatial_features"].shape[1])
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in det_state(shapes, int(z["param_seed"])).items()}, strict=False)
    assert not missing.unexpected_keys and all("num_batches_tracked" in k for k in missing.missing_keys), missing
    m = m.to(DEV).eval()
    m.set_conv_precision(precision)
    with torch.no_grad():
        d = m({"spatial_features": torch.from_numpy(z["spatial_features"]).to(DEV).contiguous(memory_format=torch.channels_last),
               "spatial_scale_features": torch.from_numpy(z["spatial_scale_features"]).to(DEV).contiguous(memory_format=torch.channels_last),
               "batch_size": 2})
    got, ref = d["spatial_features_2d"].cpu().numpy(), z["spatial_features_2d"]
    assert got.shape == ref.shape
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    observed(f"G4-eval[{tag}, {precision}] on the HIP path: max-norm error {err:.2e} (element-wise bar 1e-3)")
    _close(got, ref)


def test_g5_full_size_anchor_grid_through_the_decode_kernel(golden_dir):
    """Zero box residuals decode to the anchors themselves (x, y, z, dx, dy, dz bit for bit): the product's anchor tables at hvpr_car
    size through hvpr_head_decode_f32 against the reference's AnchorGenerator (fixture: every 97th anchor, the x row, the y column,
    float64 column sums)."""
    z = np.load(os.path.join(golden_dir, "g5_anchors_full.npz"))
    ag = anchor_head.AnchorGenerator(np.array([0, -19.84, -2.5, 47.36, 19.84, 0.5], np.float32),
                                     [dict(anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-1.78], align_center=False)])
    xs, ys = ag.shifts(np.array([296, 248]), False)
    np.testing.assert_array_equal(xs.numpy(), z["x_row"])
    np.testing.assert_array_equal(ys.numpy(), z["y_col"])
    zc = float(torch.tensor(-1.78, dtype=torch.float32) + torch.tensor(1.56, dtype=torch.float32) / 2)
    table = torch.tensor([[zc, 3.9, 1.6, 1.56, 0.0], [zc, 3.9, 1.6, 1.56, 1.57]], dtype=torch.float32)
    head = torch.zeros((1, 248, 296, 2 * (1 + 7 + 2)), dtype=torch.float32, device=DEV)
    _, box, _, _ = kernels.head_decode(head, 2, 1, 2, xs.to(DEV), ys.to(DEV), table.to(DEV), 0.78539, 0.0, np.pi)
    a = box[0].cpu().numpy()
    assert a.shape[0] == int(np.prod(z["shape"][:-1]))
    np.testing.assert_array_equal(a[z["sample_idx"]][:, :6], z["sample"][:, :6])
    np.testing.assert_allclose(a[:, :6].astype(np.float64).sum(0), z["sum64"][:6], rtol=1e-12)
    # heading: the direction fix maps the anchor's 0 / 1.57 into [offset, offset + pi) (anchor_head_template.py:327-333)
    rot = z["sample"][:, 6]
    want = (rot - 0.78539) - np.floor((rot - 0.78539) / np.pi) * np.pi + 0.78539
    np.testing.assert_allclose(a[z["sample_idx"]][:, 6], want, rtol=0, atol=1e-6)


def test_g7_limit_period_table_through_the_decode_kernel(golden_dir):
    """limit_period(val, 0, pi) of the reference (common_utils.py:20-23) = the direction fix of hvpr_head_decode_f32 with DIR_OFFSET 0,
    anchors of heading 0 and the table's values as heading residuals (equal direction logits: bin 0)."""
    z = np.load(os.path.join(golden_dir, "g6_g7_coder.npz"))
    val = z["lp_val"]
    n = len(val)
    head = torch.zeros((1, 1, n, 1 + 7 + 2), dtype=torch.float32)
    head[0, 0, :, 1 + 6] = torch.from_numpy(val)
    xs = torch.zeros(n)
    table = torch.tensor([[0.0, 1.0, 1.0, 1.0, 0.0]])
    _, box, _, _ = kernels.head_decode(head.to(DEV), 1, 1, 2, xs.to(DEV), torch.zeros(1, device=DEV), table.to(DEV), 0.0, 0.0, np.pi)
    np.testing.assert_array_equal(box[0, :, 6].cpu().numpy(), z["lp_0_pi"])


def test_g6_residual_encoding_through_the_assigner_kernel(golden_dir):
    """ResidualCoder.encode_torch of the reference on 64 random (box, anchor) pairs = the regression targets hvpr_assign_targets_f32
    writes: one anchor set per pair (a single anchor; its only ground truth overlaps it, so the match is forced whatever the IoU)."""
    z = np.load(os.path.join(golden_dir, "g6_g7_coder.npz"))
    L = kernels.lib()
    checked = 0
    for i in range(len(z["boxes"])):
        an = torch.from_numpy(z["anchors"][i:i + 1].copy()).to(DEV)
        gt = torch.from_numpy(np.concatenate([z["boxes"][i], [1.0]]).astype(np.float32)[None, None]).to(DEV)
        lab = torch.empty((1, 1), dtype=torch.int32, device=DEV)
        tgt = torch.empty((1, 1, 7), dtype=torch.float32, device=DEV)
        w = torch.empty((1, 1), dtype=torch.float32, device=DEV)
        pos = torch.zeros((1,), dtype=torch.int32, device=DEV)
        ws = torch.empty(256, dtype=torch.uint8, device=DEV)
        kernels.check(L.hvpr_assign_targets_f32(an.data_ptr(), 1, gt.data_ptr(), 1, 1, 0, 1, 0.6, 0.45, 1, 1, 0, 1, lab.data_ptr(), tgt.data_ptr(),
                                                w.data_ptr(), pos.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_assign_targets_f32")
        if int(lab.item()) != 1:          # the pair does not overlap in the BEV after axis snapping: nothing is encoded
            assert int(pos.item()) == 0 and float(tgt.abs().sum()) == 0.0
            continue
        np.testing.assert_allclose(tgt[0, 0].cpu().numpy(), z["enc"][i], rtol=1e-6, atol=1e-6)
        checked += 1
    assert checked >= 32          # (the other pairs do not overlap in the BEV after axis snapping)
