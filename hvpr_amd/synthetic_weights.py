"""Deterministic synthetic weights for tests and bench.py (no checkpoints can be downloaded here): every parameter and
BatchNorm statistic is drawn from a numpy Generator keyed by (seed, crc32(parameter name)), so that the GPU model and
the CPU oracle can be given bit-identical weights without storing 15 M floats.  BatchNorm statistics are non-trivial
(mean ~ N(0, .5), var ~ U(.5, 2)) so that folding is exercised.  The head follows the reference's init_weights():
conv_box.weight ~ N(0, 0.001) and (by default in bench.py) conv_cls.bias = -log(99)."""
import zlib

import numpy as np
import torch


def det_tensor(name, shape, seed):
    rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    if name.endswith("running_var"):
        return rng.uniform(0.5, 2.0, shape).astype(np.float32)
    if name.endswith("running_mean"):
        return rng.normal(0.0, 0.5, shape).astype(np.float32)
    if name.endswith("memory.weight"):
        return rng.uniform(-0.125, 0.125, shape).astype(np.float32)
    if name.endswith("conv_box.weight"):
        # the reference's own head initialisation: nn.init.normal_(conv_box.weight, 0, 0.001)
        # (pcdet/models/dense_heads/anchor_head_single.py:35-38) — decoded boxes stay anchor-sized
        return rng.normal(0.0, 0.001, shape).astype(np.float32)
    if len(shape) == 1:
        if name.endswith("bias"):
            return rng.normal(0.0, 0.3, shape).astype(np.float32)
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    fan_in = shape[0] if ("deblocks" in name and len(shape) == 4) else int(np.prod(shape[1:]))
    return rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)


def synthetic_state(model, seed=0, cls_bias=None):
    """-> {name: ndarray} for every float entry of model.state_dict()."""
    out = {}
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        out[k] = det_tensor(k, v.shape, seed)
    if cls_bias is not None:
        for k in out:
            if k.endswith("conv_cls.bias"):
                out[k] = np.full_like(out[k], cls_bias)
    return out


def load_synthetic(model, seed=0, cls_bias=None):
    st = synthetic_state(model, seed, cls_bias)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    return st
