"""N>1 path on CPU: two gloo ranks shard streams, run independent 'replicas', and meet only for the barrier,
the max-over-ranks timing and the optional fixed-shape result gather (thewhisper_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_streams, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from thewhisper_amd.dist import Replicas, shard_streams

    rep = Replicas(backend="gloo", device=torch.device("cpu"))
    mine = shard_streams(n_streams, rank, world)
    # stand-in for the per-rank hot path: tokens depend only on the stream id, so any mis-routing is visible
    toks = np.stack([np.arange(5, dtype=np.int32) + 100 * s for s in mine]) if mine else np.zeros((0, 5), np.int32)
    rep.barrier()
    t = rep.max_float(0.25 * (rank + 1))
    n = rep.sum_int(len(mine) * 5)
    got = rep.gather_tokens(mine, toks, n_streams)
    q.put((rank, mine, t, n, None if got is None else {k: v.tolist() for k, v in got.items()}))
    rep.close()


def test_shard_streams_partition():
    from thewhisper_amd.dist import shard_streams

    for world in (1, 2, 4, 8):
        parts = [shard_streams(128, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(128))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert shard_streams(5, 1, 2) == [1, 3]
    with pytest.raises(ValueError):
        shard_streams(4, 2, 2)


@pytest.mark.timeout(120)
def test_two_rank_gloo_replicas():
    world, n_streams = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert res[0][2] == res[1][2] == 0.5           # max over ranks
    assert res[0][3] == res[1][3] == 35            # sum over ranks
    gathered = res[0][4]
    assert res[1][4] is None and sorted(gathered) == list(range(7))
    for s in range(7):
        assert gathered[s] == [100 * s + i for i in range(5)]
