"""Optimiser step of the training path: Adam with one-cycle lr / momentum and TRUE weight decay.

Restates build_optimizer 'adam_onecycle' (tools/train_utils/optimization/__init__.py:19-32), OptimWrapper.step
(fastai_optim.py:132-149: p *= 1 - wd*lr for every group, BatchNorm included, the optimiser's own weight_decay forced to 0),
OneCycle (learning_schedules_fastai.py:60-77) with cosine annealing (:53-57), and the clipped step of train_one_epoch
(tools/train_utils/train_utils.py:35-42).  Parameter updates use torch's foreach/fused kernels instead of the reference's
per-tensor python loops."""
import math

import torch
import torch.nn as nn


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr: low -> max over the first pct_start of the steps, then max -> low/1e4; momentum: moms[0] -> moms[1] -> moms[0]."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, total_step
        low = lr_max / div_factor
        bounds = [0, int(pct_start * total_step), total_step]
        self.lr_phases = [(bounds[0], bounds[1], low, lr_max), (bounds[1], bounds[2], lr_max, low / 1e4)]
        self.mom_phases = [(bounds[0], bounds[1], moms[0], moms[1]), (bounds[1], bounds[2], moms[1], moms[0])]
        optimizer.lr, optimizer.mom = low, moms[0]

    def step(self, step):
        for start, end, a, b in self.lr_phases:
            if step >= start:
                self.optimizer.lr = annealing_cos(a, b, (step - start) / (end - start))
        for start, end, a, b in self.mom_phases:
            if step >= start:
                self.optimizer.mom = annealing_cos(a, b, (step - start) / (end - start))


class AdamOneCycle:
    """Adam(betas=(mom, 0.99)) over two parameter groups (non-BatchNorm / BatchNorm), decoupled weight decay on both."""

    def __init__(self, model, wd, lr=3e-3, beta2=0.99):
        bn_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
        bn, rest = [], []
        for m in model.modules():
            if next(m.children(), None) is not None:
                continue            # leaves only (the reference flattens the model into leaf modules, __init__.py:26-27)
            for p in m.parameters(recurse=False):
                if p.requires_grad:
                    (bn if isinstance(m, bn_types) else rest).append(p)
        owned = {id(p) for p in bn + rest}
        rest += [p for p in model.parameters() if p.requires_grad and id(p) not in owned]   # parameters held by non-leaf modules
        self.opt = torch.optim.Adam([{"params": rest}, {"params": bn}], lr=lr, betas=(0.9, beta2), weight_decay=0.0)
        self.wd, self._lr, self._mom, self.beta2 = wd, lr, 0.9, beta2

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr = float(v)
        for g in self.opt.param_groups:
            g["lr"] = self._lr

    @property
    def mom(self):
        return self._mom

    @mom.setter
    def mom(self, v):
        self._mom = float(v)
        for g in self.opt.param_groups:
            g["betas"] = (self._mom, self.beta2)

    @property
    def param_groups(self):
        return self.opt.param_groups

    def zero_grad(self):
        self.opt.zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self):
        for g in self.opt.param_groups:          # true weight decay first, on every trainable tensor, BN included
            ps = [p for p in g["params"] if p.requires_grad]
            if ps:
                torch._foreach_mul_(ps, 1.0 - self.wd * self._lr)
        self.opt.step()

    def state_dict(self):
        """Inner Adam state (step counters + both moments per parameter) plus the wrapper's own scalars, so that a resumed
        run continues with the same lr / momentum until the scheduler's next step."""
        sd = self.opt.state_dict()
        sd["hvpr_onecycle"] = {"lr": self._lr, "mom": self._mom, "wd": self.wd, "beta2": self.beta2}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        extra = sd.pop("hvpr_onecycle", None)
        self.opt.load_state_dict(sd)
        if extra is not None:
            self.wd, self.beta2 = float(extra["wd"]), float(extra["beta2"])
            self.lr, self.mom = extra["lr"], extra["mom"]


def build_optimizer(model, optim_cfg):
    assert optim_cfg.OPTIMIZER == "adam_onecycle", "hvpr path: adam_onecycle (hvpr.yaml:158)"
    return AdamOneCycle(model, wd=optim_cfg.WEIGHT_DECAY)


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total = total_iters_each_epoch * total_epochs
    return OneCycle(optimizer, total, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR, optim_cfg.PCT_START), None


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    """The dict tools/train_utils/train_utils.py:124-140 writes: {'epoch', 'it', 'model_state', 'optimizer_state', 'version'}.
    A DDP wrapper is unwrapped (its `.module` holds the reference's key names) and every tensor goes to the CPU, so a file
    written by one rank loads anywhere."""
    model_state = None
    if model is not None:
        inner = model.module if hasattr(model, "module") else model
        model_state = type(inner.state_dict())((k, v.detach().cpu()) for k, v in inner.state_dict().items())

    def to_cpu(o):
        if torch.is_tensor(o):
            return o.detach().cpu()
        if isinstance(o, dict):
            return {k: to_cpu(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(to_cpu(v) for v in o)
        return o
    return {"epoch": epoch, "it": it, "model_state": model_state,
            "optimizer_state": None if optimizer is None else to_cpu(optimizer.state_dict()), "version": "hvpr_amd"}


def save_checkpoint(state, filename="checkpoint"):
    """train_utils.py:143-150: one file `<filename>.pth`."""
    path = "{}.pth".format(filename)
    torch.save(state, path)
    return path


def model_fn_decorator():
    """pcdet/models/__init__.py (absent upstream file) with the 4-tuple HVPR's driver expects (train_utils.py:38)."""
    from .detector import load_data_to_gpu

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret, tb_dict, disp_dict = model(batch_dict)
        loss = ret["loss"].mean()
        (model.module if hasattr(model, "module") else model).update_global_step()
        return loss, tb_dict, disp_dict, disp_dict.get("items")

    return model_func


def train_step(model, optimizer, scheduler, batch_dict, it, grad_norm_clip):
    """One iteration of train_one_epoch (train_utils.py:25-42): schedule, zero_grad, forward, backward, clip, step."""
    scheduler.step(it)
    model.train()
    optimizer.zero_grad()
    ret, tb_dict, disp_dict = model(batch_dict)
    loss = ret["loss"].mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm_clip)
    optimizer.step()
    (model.module if hasattr(model, "module") else model).update_global_step()
    return loss.detach(), tb_dict
