"""CPU: the torch-level training logic of the product (target assigner, losses, Adam-onecycle, train branches of the
memory / point-pillar attention) against fixtures produced by the reference's own code (tests/golden/make_golden.py G8-G10).
The same cases run on cuda:0 in tests/test_gpu_train_fixtures.py."""
import train_fixture_cases as C


def test_g8_target_assignment_and_losses(golden_dir):
    C.run_g8(golden_dir)


def test_g8_no_ground_truth_is_all_background(golden_dir):
    C.run_g8_no_gt(golden_dir)


def test_g9_onecycle_schedule_and_true_weight_decay(golden_dir):
    C.run_g9(golden_dir)


def test_g10_point_pillar_attention_and_memory_train_branch(golden_dir):
    C.run_g10(golden_dir)


def test_training_ops_refuse_cpu_tensors_unless_the_reference_form_is_switched_on():
    """The product path has no CPU fallback: get_score / the memory training branch raise on CPU tensors; the torch reference
    forms the fixtures above run through are opt-in."""
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
