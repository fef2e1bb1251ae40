"""N > 1 path on CPU: world_size-2 gloo run of the counter reduction / max-timing / done logic
that bench.py and the Monte-Carlo loop use over RCCL on the GPU box."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvbs2_amd.parallel import all_done, reduce_counters, reduce_max, shard_frames
    lo, hi = shard_frames(101, rank, world)
    # every rank "decodes" its shard: FRA = shard size, BE/FE synthetic but rank-dependent
    local = [hi - lo, 10 * (rank + 1), rank + 1]
    tot = reduce_counters(local)
    tmax = reduce_max(0.5 + rank)
    done_lo = all_done(rank + 1, max_fe=100)
    done_hi = all_done(60 * (rank + 1), max_fe=100)
    q.put((rank, tot, tmax, done_lo, done_hi, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_counter_reduction_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tot, tmax, done_lo, done_hi, _ in res:
        assert tot == [101, 30, 3]          # same reduced {FRA, BE, FE} on every rank
        assert tmax == 1.5
        assert done_lo is False and done_hi is True
    assert res[0][5] == (0, 51) and res[1][5] == (51, 101)


def test_reductions_are_identity_without_process_group():
    sys.path.insert(0, ROOT)
    from dvbs2_amd.parallel import reduce_counters, reduce_max
    assert reduce_counters([3, 2, 1]) == [3, 2, 1] and reduce_max(2.5) == 2.5
