"""Deterministic, name-keyed synthetic parameters shared by make_golden.py (which loads them into the
reference modules) and the tests (which rebuild them instead of storing 15 M floats per fixture)."""
import zlib

import numpy as np


def det_tensor(name, shape, seed):
    rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    if name.endswith("running_var"):
        return rng.uniform(0.5, 2.0, shape).astype(np.float32)
    if name.endswith("running_mean"):
        return rng.normal(0.0, 0.5, shape).astype(np.float32)
    if name.endswith("memory.weight"):
        return rng.uniform(-0.125, 0.125, shape).astype(np.float32)
    if len(shape) == 1:
        if name.endswith("bias"):
            return rng.normal(0.0, 0.3, shape).astype(np.float32)
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    fan_in = shape[0] if ("deblocks" in name and len(shape) == 4) else int(np.prod(shape[1:]))
    return rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)


def det_state(shapes, seed):
    """shapes: {name: shape} -> {name: ndarray}; num_batches_tracked entries are skipped."""
    return {k: det_tensor(k, s, seed) for k, s in shapes.items() if "num_batches_tracked" not in k}
