"""GPU: the training rows a10 / a12 / a13 / a14 on cuda:0 against the SAME reference-generated fixtures the CPU suite uses
(G8 target assigner + losses, G9 one-cycle + 3 optimiser steps, G10 train-branch memory + get_score): the device code paths
(target assigner on device, top-k through the read-out kernel, scatter autograd) differ from the CPU ones."""
import pytest

import train_fixture_cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_g8_target_assignment_and_losses_on_gpu(golden_dir):
    C.run_g8(golden_dir, DEV, rtol=1e-4)


def test_g8_no_ground_truth_on_gpu(golden_dir):
    C.run_g8_no_gt(golden_dir, DEV)


def test_g9_onecycle_three_steps_on_gpu(golden_dir):
    C.run_g9(golden_dir, DEV, rtol=1e-4)


def test_g10_get_score_and_memory_train_branch_on_gpu(golden_dir):
    C.run_g10(golden_dir, DEV, rtol=1e-4)
