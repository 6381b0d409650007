"""Multi-GPU harness of the inference path: whole frames are sharded over ranks, replicas only — no data-path collective
(the reference's eval sampler does the same, pcdet/datasets/__init__.py:18-38).  The process group is used
for the start/stop barriers, the max-over-ranks time and the host-side merge of results (reference: pickle files + two
barriers, pcdet/utils/common_utils.py:174-195).  Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def launch_local(n_ranks, argv, env=None, timeout=None):
    """Start `n_ranks` fresh child processes `python argv...`, one per GPU of this node, each with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set (reference launcher: tools/train.py:61-70,
    pcdet/utils/common_utils.py:141-154 — one process per GPU, tcp://127.0.0.1 rendezvous).  The caller must not have
    touched the GPU: children are started with subprocess (fork + exec of a new interpreter), never by re-exec'ing a
    process that initialised HIP.  Returns the largest exit code (0 = every rank succeeded); when one rank fails the others
    are terminated."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    procs = []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    rc = 0
    try:
        pending = list(procs)
        import time
        t_end = None if timeout is None else time.time() + timeout
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is not None:
                    pending.remove(p)
                    rc = max(rc, abs(c))
                    if c != 0:                    # one rank died: the others would wait in a collective forever
                        for q in pending:
                            q.terminate()
            if t_end is not None and time.time() > t_end:
                for q in pending:
                    q.terminate()
                rc = max(rc, 124)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def ranks_seen(device):
    """Number of ranks that take part in the process group's collectives: an all-reduce (sum) of ones — RCCL on the GPU."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def gather_floats(value, device):
    """The python float of every rank, in rank order, on every rank."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def shard_frames(n_frames, rank, world):
    """Frame ids of this rank: rank, rank+world, ... (every frame exactly once over the ranks)."""
    return list(range(rank, n_frames, world))


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device):
    """Max of a python float over the ranks (the timed region of the slowest rank)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local, rank, world):
    """Merge per-frame results {frame_id: obj} of all ranks on every rank (host objects)."""
    if not dist.is_initialized():
        return dict(local)
    out = [None] * world
    dist.all_gather_object(out, local)
    merged = {}
    for d in out:
        merged.update(d)
    return merged


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()


def convert_sync_batchnorm(model, process_group=None):
    """The counterpart of `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` (reference tools/train.py:119-120, --sync_bn) for
    this path: its BatchNorms run the library's own kernels, so there is no module to swap — from now on the batch statistics of
    EVERY train-mode BatchNorm of the model are those of the global batch: conv_train.bn_relu / sfm_step (backbone, head,
    point-stream MLPs, VFE scale stream) all-reduce their per-rank sums over `process_group` (default group when None); the two
    BatchNorm1d inside the fused PFN kernels and SpatialAttention's BatchNorm reach the same all-reduce through the library's hook
    (hvpr_set_batchnorm_allreduce).  Returns the model unchanged; call before wrap_ddp."""
    from . import conv_train
    conv_train.set_sync_batchnorm(process_group if process_group is not None else True)
    return model


def wrap_ddp(model, device):
    """Data-parallel training wrapper (reference: tools/train.py:143-145): one process per GPU, gradient all-reduce over
    RCCL ("nccl" backend) overlapped with backward in ~25 MB buckets; the ~62 MB of fp32 gradients take ~0.7 ms on the
    8-GPU xGMI ring, far below the backward time.  find_unused_parameters stays False: every parameter of
    the training graph gets a gradient.  gradient_as_bucket_view stays False (torch's default, what the reference's train.py
    gets): DDP then writes the reduced buckets back INTO the existing p.grad tensors, which with FusedAdamOneCycle are views of
    its flat gradient buffer — p.grad keeps its address and the optimiser steps straight from the flat buffer."""
    if not dist.is_initialized():
        return model
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=True, gradient_as_bucket_view=False)
