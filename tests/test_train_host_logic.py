"""CPU: the torch-level training logic of the product (target assigner, losses, Adam-onecycle, train branches of the
memory / point-pillar attention) against fixtures produced by the reference's own code (tests/golden/make_golden.py G8-G10).
The same cases run on cuda:0 in tests/test_gpu_train_fixtures.py."""
import train_fixture_cases as C


def test_g8_target_assignment_and_losses(golden_dir):
    C.run_g8(golden_dir)


def test_g8_no_ground_truth_is_all_background(golden_dir):
    C.run_g8_no_gt(golden_dir)


def test_g8_batched_assigner_passes_equal_small_batches(golden_dir):
    C.run_g8_batch_passes(golden_dir)


def test_g9_onecycle_schedule_and_true_weight_decay(golden_dir):
    C.run_g9(golden_dir)


def test_g10_point_pillar_attention_and_memory_train_branch(golden_dir):
    C.run_g10(golden_dir)


def test_g16_torch_forms_reproduce_the_reference_gradients_float64(golden_dir):
    """The comparator of the GPU parity tests (tests/torch_forms.py: scatter training branch, get_score, memory addressing, head)
    + the product's target assigner and losses.py against the reference's own modules in float64: values, and the gradient of each
    of the three losses w.r.t. pillar / point features and the memory bank to 1e-6 norm-wise, exact zeros where the reference detaches."""
    import torch
    rep = C.run_g16(golden_dir, "cpu", torch.float64, value_rtol=1e-5, grad_tol=1e-6)
    assert len(rep) == 13 and max(e for e, _ in rep.values()) < 1e-6


def test_g16_torch_forms_fp32(golden_dir):
    import torch
    C.run_g16(golden_dir, "cpu", torch.float32, value_rtol=1e-5, grad_tol=1e-5)


def test_g17_driver_loop_makes_the_reference_loops_calls(golden_dir):
    """The loop the GPU test drives the product's plugin objects with (train_fixture_cases.drive_like_g17) is call for call the
    reference's train_one_epoch (fixture G17, recorded from tools/train_utils/train_utils.py:9-61 itself)."""
    C.run_g17_protocol(golden_dir)


def test_model_func_keeps_tensors_out_of_the_progress_bar():
    """train_utils.py:45-51 hands disp_dict to tqdm's set_postfix: model_fn_decorator must return the memory items as the FOURTH
    value (train_utils.py:38, printed per epoch :100-101) and keep them out of disp_dict."""
    import torch
    from hvpr_amd import optim

    class M(torch.nn.Module):
        global_step = 0

        def update_global_step(self):
            self.global_step += 1

        def forward(self, batch_dict):
            return {"loss": torch.ones(2)}, {"rpn_loss": torch.tensor(1.0)}, {"items": torch.zeros(2000, 64)}
    m = M()
    loss, tb, disp, items = optim.model_fn_decorator()(m, {})
    assert loss.item() == 1.0 and items.shape == (2000, 64) and disp == {} and m.global_step == 1


def test_training_modules_refuse_cpu_tensors():
    """The product path has no CPU fallback: get_score / the memory training branch / the backbone, head and VFE training forwards
    raise on CPU tensors; the torch forms the fixtures above run through live in tests/torch_forms.py."""
    import numpy as np
    import pytest
    import torch
    from hvpr_amd import map_to_bev
    from hvpr_amd.config import AttrDict
    cfg = AttrDict(NUM_BEV_FEATURES=128, NUM_COORD_POINTS=3, NUM_PT_FEATURES=64, NUM_SCALE_FEATURES=32, NUM_K=20, NUM_M=2000, SHRINK_TH=0.0025)
    m = map_to_bev.PointPillarScatter_Agg_Memory_1_scale(cfg, np.array([12, 10, 1])).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.get_score(torch.randn(100, 64), torch.randn(5, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.memory(torch.randn(5, 64), 20, torch.randn(5, 20, 64))
    # the dense training modules as well: no silent torch path on CPU tensors
    from hvpr_amd import detector
    from hvpr_amd.config import hvpr_car_cfg
    import copy
    cfg = copy.deepcopy(hvpr_car_cfg())
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [0, -2.56, -3, 5.12, 2.56, 1]
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True)).train()
    canv = {"spatial_features": torch.randn(1, 128, 32, 32), "spatial_features_point": torch.randn(1, 128, 32, 32),
            "spatial_scale_features": torch.randn(1, 32, 32, 32)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.backbone_2d(dict(canv))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.dense_head({"spatial_features_2d": torch.randn(1, 384, 32, 32), "spatial_features_point_2d": torch.randn(1, 384, 32, 32)})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.vfe({"voxels": torch.rand(3, 32, 4), "voxel_num_points": torch.tensor([1, 2, 3]), "voxel_coords": torch.zeros(3, 4)})


def test_frame_ranges_slices_for_grouped_rows_and_none_otherwise():
    """map_to_bev._frame_ranges: rows grouped by frame in ascending order (what the voxelizer, the point stream and the reference's
    collate produce) come back as per-frame (lo, hi) slices from ONE host read — empty frames included; anything else (unsorted,
    an index outside [0, B)) comes back as None, and the training branch falls back to the reference's boolean masks
    (pointpillar_scatter.py:101-104)."""
    import torch
    from hvpr_amd.map_to_bev import _frame_ranges
    a = torch.tensor([0, 0, 0, 2, 2, 3], dtype=torch.int32)            # frame 1 empty
    b = torch.tensor([0, 1, 1, 1, 2, 3, 3], dtype=torch.int64)
    ra, rb = _frame_ranges([a, b], 4)
    assert ra == [(0, 3), (3, 3), (3, 5), (5, 6)]
    assert rb == [(0, 1), (1, 4), (4, 5), (5, 7)]
    for r, col in ((ra, a), (rb, b)):
        for f, (lo, hi) in enumerate(r):
            assert torch.equal(torch.arange(lo, hi), torch.nonzero(col == f).flatten())
    unsorted = torch.tensor([0, 2, 1, 1], dtype=torch.int32)
    outside = torch.tensor([0, 1, 4], dtype=torch.int32)
    negative = torch.tensor([-1, 0, 1], dtype=torch.int32)
    empty = torch.zeros(0, dtype=torch.int32)
    ru, ro, rn, re = _frame_ranges([unsorted, outside, negative, empty], 4)
    assert ru is None and ro is None and rn is None
    assert re == [(0, 0)] * 4
