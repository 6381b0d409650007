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
