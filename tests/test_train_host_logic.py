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
