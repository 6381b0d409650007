"""CPU: checkpoint save / resume in the reference's file layout (tools/train_utils/train_utils.py:124-150,
pcdet/models/detectors/detector3d_template.py:320-375): {'epoch','it','model_state','optimizer_state','version'}."""
import copy

import torch

from hvpr_amd import detector, optim, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg


def _small_cfg():
    cfg = copy.deepcopy(hvpr_car_cfg())
    cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [0, -2.56, -3, 5.12, 2.56, 1]
    return cfg


def _fake_step(model, opt, sched, it, seed):
    g = torch.Generator().manual_seed(seed)
    sched.step(it)
    opt.zero_grad()
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
    opt.step()


def test_checkpoint_round_trip_resumes_bit_identically(tmp_path):
    cfg = _small_cfg()
    a = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(a, seed=7)
    oa = optim.build_optimizer(a, cfg.OPTIMIZATION)
    sa, _ = optim.build_scheduler(oa, 10, 1, -1, cfg.OPTIMIZATION)
    for it in range(2):
        _fake_step(a, oa, sa, it, 100 + it)
    a.update_global_step()
    class Wrapped(torch.nn.Module):            # what DistributedDataParallel looks like to checkpoint_state: `.module`
        def __init__(self, m):
            super().__init__()
            self.module = m
    path = optim.save_checkpoint(optim.checkpoint_state(Wrapped(a), oa, epoch=3, it=2), str(tmp_path / "checkpoint_epoch_3"))
    assert path.endswith("checkpoint_epoch_3.pth")
    ck = torch.load(path)
    assert set(ck) == {"epoch", "it", "model_state", "optimizer_state", "version"} and ck["epoch"] == 3 and ck["it"] == 2
    assert all(v.device.type == "cpu" for v in ck["model_state"].values())
    assert "vfe.pfn_layers.0.linear.weight" in ck["model_state"] and "global_step" in ck["model_state"]     # reference key names

    b = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    ob = optim.build_optimizer(b, cfg.OPTIMIZATION)
    sb, _ = optim.build_scheduler(ob, 10, 1, -1, cfg.OPTIMIZATION)
    it, epoch = b.load_params_with_optimizer(path, to_cpu=True, optimizer=ob)
    assert (it, epoch) == (2, 3) and int(b.global_step) == 1
    assert (ob.lr, ob.mom) == (oa.lr, oa.mom)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    _fake_step(a, oa, sa, 2, 555)
    _fake_step(b, ob, sb, 2, 555)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k                                   # Adam moments and step counts came back too


def test_reference_format_checkpoint_loads_by_key_and_shape(tmp_path):
    """A file as the reference writes it (model_state only is enough for tools/test.py --ckpt): matching keys load, foreign
    keys and wrong shapes are skipped (detector3d_template.py:332-340)."""
    cfg = _small_cfg()
    a = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    synthetic_weights.load_synthetic(a, seed=8)
    st = {k: v.clone() for k, v in a.state_dict().items()}
    st["roi_head.not_ours.weight"] = torch.zeros(3)
    st["dense_head.conv_cls.bias"] = torch.zeros(7)                  # wrong shape: skipped
    path = str(tmp_path / "ref.pth")
    torch.save({"epoch": 80, "it": 1.0, "model_state": st, "optimizer_state": None, "version": "pcdet+0.3.0"}, path)
    b = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    n_loaded, n_total = b.load_params_from_file(path, to_cpu=True)
    assert n_loaded == n_total - 1
    assert torch.equal(b.state_dict()["backbone_2d.blocks.0.1.weight"], a.state_dict()["backbone_2d.blocks.0.1.weight"])
