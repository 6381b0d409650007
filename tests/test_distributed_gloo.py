"""CPU, world_size 2, gloo: the multi-GPU harness of the inference path (frame sharding, barriers, max-over-ranks time,
result merge) — the same code bench.py runs over RCCL."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from hvpr_amd import distributed
    r, lr, w = distributed.init("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    mine = distributed.shard_frames(11, rank, world)
    distributed.barrier(dev)
    slow = distributed.max_over_ranks(1.0 + rank, dev)            # rank 1 is the slow one
    merged = distributed.gather_results({f: f * f for f in mine}, rank, world)
    distributed.barrier(dev)
    distributed.finalize()
    q.put((rank, mine, slow, merged))


def test_two_ranks_shard_and_merge():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, s0, g0), (r1, m1, s1, g1) = got
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)     # every frame exactly once
    assert s0 == s1 == 2.0                                                   # max over ranks
    assert g0 == g1 == {f: f * f for f in range(11)}


def test_single_process_is_a_noop():
    import torch
    from hvpr_amd import distributed
    assert distributed.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    assert distributed.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    assert distributed.gather_results({1: "a"}, 0, 1) == {1: "a"}


# ------------------------------------------------------------------------------------------------ a15: DDP training
class _DenseTrainStack:
    """The dense (torch-autograd) half of the training graph — BEV backbone (two streams, shared weights) + anchor head +
    target assigner + losses — on a 32x32 canvas so that it runs on CPU.  Built lazily inside each spawned worker."""

    @staticmethod
    def build():
        import copy
        import torch
        from hvpr_amd import detector, synthetic_weights
        from hvpr_amd.config import hvpr_car_cfg
        cfg = copy.deepcopy(hvpr_car_cfg())
        cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [0, -2.56, -3, 5.12, 2.56, 1]
        model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
        synthetic_weights.load_synthetic(model, seed=4, cls_bias=-4.595)
        import torch_forms
        torch_forms.patch(model)          # CPU: the torch forms of the dense modules (the product's forwards are HIP-only)

        class Stack(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.backbone_2d, self.dense_head = model.backbone_2d, model.dense_head

            def forward(self, d):
                d = self.dense_head(self.backbone_2d(dict(d)))
                rpn, rpn_point, mem, _, _ = self.dense_head.get_loss()
                return rpn + rpn_point + mem
        return Stack().train()

    @staticmethod
    def batch(seed):
        import numpy as np
        import torch
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        gt = np.zeros((1, 2, 8), np.float32)
        gt[0, 0] = [2.0 + 0.3 * seed, 0.4, -1.0, 3.9, 1.6, 1.56, 0.3, 1]
        return {"spatial_features": r(1, 128, 32, 32), "spatial_features_point": r(1, 128, 32, 32), "spatial_scale_features": r(1, 32, 32, 32),
                "point_positive_features": r(5, 64), "memory_positive_features": r(5, 64), "memory_items": r(10, 64),
                "gt_boxes": torch.from_numpy(gt), "batch_size": 1}


def _local_grads(seed):
    import torch
    torch.manual_seed(0)
    m = _DenseTrainStack.build()
    m(_DenseTrainStack.batch(seed)).backward()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.set_num_threads(2)
    from hvpr_amd import distributed
    distributed.init("gloo")
    torch.manual_seed(0)
    ddp = distributed.wrap_ddp(_DenseTrainStack.build(), torch.device("cpu"))
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    ddp(_DenseTrainStack.batch(rank)).backward()                       # every rank trains on its own shard
    got = {k: p.grad.clone() for k, p in ddp.module.named_parameters() if p.grad is not None}
    want = [_local_grads(r) for r in range(world)]                      # what each rank would have had alone
    worst = 0.0
    for k, g in got.items():
        mean = sum(w[k] for w in want) / world
        worst = max(worst, float((g - mean).abs().max() / (mean.abs().max() + 1e-12)))
    distributed.finalize()
    q.put((rank, len(got), worst))


def test_ddp_gradients_are_the_rank_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in procs)
    for rank, n, worst in got:
        assert n > 20 and worst < 1e-4, (rank, n, worst)


# ------------------------------------------------------------------------------------------------ a14 + a15: flat optimiser under DDP
def _flat_adam_on_cpu(opt):
    """TEST stand-in for the ONE kernel launch of FusedAdamOneCycle._launch (hvpr_fused_adam_truewd_f32, csrc/train_ops.hip): the
    same arithmetic with torch ops on the flat buffers.  Everything around it — the flat parameter / gradient views, zero_grad,
    the gradient collection under DDP, the device-side clip scale — is the product's code and is what this test exercises."""
    import math
    import torch
    gs = 1.0 if opt._scale is None else float(opt._scale)
    g = opt.flat_g * gs
    bc1, bc2 = 1.0 - opt._mom ** opt.steps, 1.0 - opt.beta2 ** opt.steps
    opt.flat_p.mul_(1.0 - opt.wd * opt._lr)
    opt.exp_avg.add_((g - opt.exp_avg) * (1.0 - opt._mom))
    opt.exp_avg_sq.mul_(opt.beta2).add_((1.0 - opt.beta2) * g * g)
    opt.flat_p.sub_((opt._lr / bc1) * opt.exp_avg / (opt.exp_avg_sq.sqrt() / math.sqrt(bc2) + opt.eps))


def _flat_opt_worker(rank, world, port, q, bucket_view):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy
    import types
    import torch
    torch.set_num_threads(2)
    from hvpr_amd import distributed, optim
    distributed.init("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Flatten(),
                              torch.nn.Linear(8 * 6 * 6, 4))
    ref_nets = [copy.deepcopy(net) for _ in range(world)]                # what each rank would compute alone
    if bucket_view:
        ddp = torch.nn.parallel.DistributedDataParallel(net, gradient_as_bucket_view=True)
    else:
        ddp = distributed.wrap_ddp(net, torch.device("cpu"))
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = optim.FusedAdamOneCycle(ddp, wd=0.01)
    opt._launch = types.MethodType(_flat_adam_on_cpu, opt)
    ref = copy.deepcopy(net)                                             # same weights (views into flat_p are deep-copied as tensors)
    ref_opt = optim.AdamOneCycle(ref, wd=0.01)
    xs = [[torch.randn(5, 3, 8, 8, generator=torch.Generator().manual_seed(100 * it + r)) for r in range(world)] for it in range(4)]
    ptrs, worst = [], 0.0
    for it in range(4):
        opt.zero_grad()
        ddp(xs[it][rank]).pow(2).mean().backward()
        ptrs.append([p.grad.data_ptr() for p in opt.params])
        opt.clip_grad_norm(0.05)
        opt.step()
        # reference: the rank-mean gradient through the per-tensor optimiser
        ref_opt.zero_grad()
        grads = []
        for r in range(world):
            m = copy.deepcopy(ref).train()
            m(xs[it][r]).pow(2).mean().backward()
            grads.append([p.grad for p in m.parameters()])
        ref.train()
        ref(xs[it][rank])                                               # running statistics of this rank's own batch, as DDP keeps them
        for i, p in enumerate(ref.parameters()):
            p.grad = sum(g[i] for g in grads) / world
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        ref_opt.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            worst = max(worst, float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12)))
    own = [g.data_ptr() for g in opt._grad_views]
    # DDP itself rebuilds its buckets once, after the first backward (gradient-ready order): with bucket views the addresses may
    # move between step 0 and step 1, never afterwards
    first = 1 if bucket_view else 0
    stable = all(p == ptrs[first] for p in ptrs[first:])
    in_flat = ptrs[0] == own
    distributed.finalize()
    q.put((rank, stable, in_flat, worst))


def _run_flat_opt(bucket_view):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_opt_worker, args=(r, 2, port, q, bucket_view)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return sorted(q.get(timeout=10) for _ in procs)


def test_flat_optimizer_under_ddp_keeps_grad_addresses_and_matches_adam_onecycle():
    """FusedAdamOneCycle inside DistributedDataParallel, two gloo ranks: p.grad keeps ONE address over the steps (no re-pointing
    between the optimiser and DDP), with wrap_ddp's setting the gradients live in the optimiser's own flat buffer, and four
    clipped steps equal AdamOneCycle on the rank-mean gradients."""
    for rank, stable, in_flat, worst in _run_flat_opt(bucket_view=False):
        assert stable and in_flat, (rank, stable, in_flat)
        assert worst < 2e-5, (rank, worst)


def test_flat_optimizer_under_ddp_bucket_views():
    """The same with gradient_as_bucket_view=True (DDP owns p.grad): addresses stay stable as well — zero_grad zeroes the bucket
    views in place instead of re-pointing — and the results are the same."""
    for rank, stable, in_flat, worst in _run_flat_opt(bucket_view=True):
        assert stable, rank
        assert worst < 2e-5, (rank, worst)


# ------------------------------------------------------------------------------------------------ the --gpus N launcher
def test_launch_local_two_ranks(tmp_path):
    """distributed.launch_local — what `bench.py --gpus N` / `tools/bench_train.py --gpus N` call when WORLD_SIZE is unset:
    N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a collective sees all of them, exit code 0."""
    import json
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_worker.py")
    rc = distributed.launch_local(2, [worker, str(tmp_path), "ok"], timeout=240)
    assert rc == 0
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, g in enumerate(got):
        assert (g["rank"], g["local_rank"], g["world"], g["seen"], g["master"]) == (r, r, 2, 2, "127.0.0.1")
        assert g["times"] == [10.0, 11.0] and g["slowest"] == 11.0


def test_launch_local_propagates_a_failing_rank(tmp_path):
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_worker.py")
    assert distributed.launch_local(2, [worker, str(tmp_path), "fail"], timeout=240) != 0


def test_launch_local_ends_the_others_when_a_rank_dies_before_the_rendezvous(tmp_path):
    """A rank that exits before init_process_group leaves the others waiting in the rendezvous (not in a collective): the
    launcher must end them too, well inside the rendezvous' own timeout."""
    import time
    from hvpr_amd import distributed
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_worker.py")
    t0 = time.time()
    rc = distributed.launch_local(2, [worker, str(tmp_path), "fail_early"], timeout=240)
    assert rc != 0 and rc != 124 and time.time() - t0 < 120
    assert not os.path.exists(os.path.join(str(tmp_path), "rank0.json"))


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` must fail loudly when fewer than N GPUs are visible (this box has none), not run one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for script in ("bench.py", os.path.join("tools", "bench_train.py")):
        p = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout), (script, p.stderr[-500:])


# ------------------------------------------------------------------------------------------------ SyncBatchNorm (tools/train.py:119-120)
def _syncbn_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from hvpr_amd import conv_train, distributed
    distributed.init("gloo")
    g = torch.Generator().manual_seed(3)
    x_full = torch.randn(2, 6, 5, 8, generator=g) * 2.0 + 0.7            # (N, H, W, C): rank r owns sample r
    wout = torch.randn(2, 6, 5, 8, generator=g)
    gamma, beta, eps = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g), 1e-3
    x = x_full[rank:rank + 1].reshape(-1, 8)
    # what conv_train.bn_statistics / _bn_backward do around the kernels, with the kernels' sums written in torch
    mean, var, invstd, n = conv_train.sync_moments(x.sum(0), (x * x).sum(0), x.shape[0], eps)
    xhat = (x - mean) * invstd
    y = torch.relu(xhat * gamma + beta)
    dy = wout[rank:rank + 1].reshape(-1, 8) * (y > 0)
    dgamma_l, dbeta_l = (dy * xhat).sum(0), dy.sum(0)
    dg_t, db_t = conv_train.sync_backward_sums(dgamma_l, dbeta_l, n)
    dz = gamma * invstd * (dy - db_t - xhat * dg_t)
    # a condition one rank alone sees (its batch has no pillar) must be acted on by all of them: the statistics' all-reduce follows
    conv_train.set_sync_batchnorm(True)
    agree = (conv_train.any_rank_true(rank == 1, torch.device("cpu")), conv_train.any_rank_true(False, torch.device("cpu")))
    conv_train.set_sync_batchnorm(None)
    assert conv_train.any_rank_true(rank == 1, torch.device("cpu")) == (rank == 1)       # no group: the rank's own flag
    distributed.finalize()
    q.put((rank, y.numpy(), dz.numpy(), dgamma_l.numpy(), dbeta_l.numpy(), mean.numpy(), var.numpy(), float(n), agree))


def test_sync_batchnorm_two_ranks_batch1_equal_one_process_batch2():
    """SyncBatchNorm over the own BatchNorm kernels: the collective half (conv_train.sync_moments / sync_backward_sums — float64
    all-reduces of (sum x, sum x^2, count) and (sum dy xhat, sum dy)) on two gloo ranks with one sample each, against train-mode
    BatchNorm + ReLU over both samples in one process (float64 autograd): output, input gradient, and the SUM of the ranks' weight /
    bias gradients.  (The kernels on either side of the all-reduces are covered on the GPU: tests/test_gpu_conv_train.py and the
    one-rank RCCL test in tests/test_gpu_distributed.py.)"""
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    x_full = (torch.randn(2, 6, 5, 8, generator=g) * 2.0 + 0.7).double().requires_grad_(True)
    wout = torch.randn(2, 6, 5, 8, generator=g).double()
    gamma = (torch.rand(8, generator=g) + 0.5).double().requires_grad_(True)
    beta = torch.randn(8, generator=g).double().requires_grad_(True)
    y = torch.relu(torch.nn.functional.batch_norm(x_full.permute(0, 3, 1, 2), None, None, gamma, beta, True, 0.0, 1e-3)).permute(0, 2, 3, 1)
    (y * wout).sum().backward()
    for r in range(2):
        _, yr, dzr, _, _, mean, var, n, agree = [torch.from_numpy(t) if hasattr(t, "dtype") else t for t in got[r]]
        assert n == 60.0 and agree == (True, False)
        torch.testing.assert_close(yr.double().view(1, 6, 5, 8), y[r:r + 1].detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dzr.double().view(1, 6, 5, 8), x_full.grad[r:r + 1], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(mean.double(), x_full.detach().reshape(-1, 8).mean(0), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(var.double(), x_full.detach().reshape(-1, 8).var(0, unbiased=False), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(torch.from_numpy(got[0][3] + got[1][3]).double(), gamma.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(torch.from_numpy(got[0][4] + got[1][4]).double(), beta.grad, rtol=1e-4, atol=1e-5)
