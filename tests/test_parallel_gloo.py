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


def _rdv_worker(rank, world, path, delay_s, q):
    import ctypes as C
    import time
    sys.path.insert(0, ROOT)
    from dvbs2_amd import lib_binding as B
    L = C.CDLL(B.lib_path())
    L.dvbs2hip_rendezvous.restype = C.c_int
    L.dvbs2hip_rendezvous.argtypes = [C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int32]
    buf = (C.c_ubyte * 128)()
    time.sleep(delay_s)
    if rank == 0:
        for i in range(128):
            buf[i] = (7 * i + 3) & 255
    rc = L.dvbs2hip_rendezvous(rank, world, path.encode(), buf, 128, 20000)
    q.put((rank, rc, bytes(buf)))


def test_rccl_bootstrap_rendezvous_through_a_file(tmp_path):
    """dvbs2hip_monitor_reduce_init's out-of-band step (the communicator id from rank 0 to the other ranks through files) between three CPU
    processes, in a directory that still holds what a crashed earlier run left there -- an id file of the old format, a complete ack and a
    hello with another nonce: nobody may take the stale payload.  Readers that start before rank 0 and one that starts after it all end up
    with rank 0's 128 bytes, nothing is left behind, and a side whose partner never shows up times out with an error instead of hanging."""
    import ctypes as C
    from dvbs2_amd import lib_binding as B
    path = str(tmp_path / "rdv")
    stale = bytes(range(128))
    open(path, "wb").write(stale)                                            # what round 2's protocol would have read and trusted
    open(path + ".ack.1", "wb").write(b"N" * 16 + stale)                     # a complete answer to somebody else's nonce
    open(path + ".hello.2", "wb").write(b"O" * 16)                           # a reader that died: rank 0 answers it, the live rank 2 replaces it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdv_worker, args=(r, 3, path, d, q)) for r, d in ((1, 0.0), (0, 0.5), (2, 1.2))]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    want = bytes((7 * i + 3) & 255 for i in range(128))
    assert [g[0] for g in got] == [0, 1, 2] and all(g[1] == 0 and g[2] == want for g in got)
    assert sorted(os.listdir(tmp_path)) == ["rdv"]                           # the exchange cleaned up after itself (the old-format file is not its own)
    L = C.CDLL(B.lib_path())
    L.dvbs2hip_rendezvous.argtypes = [C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int32]
    buf = (C.c_ubyte * 128)()
    assert L.dvbs2hip_rendezvous(1, 2, str(tmp_path / "nobody").encode(), buf, 128, 100) == -3      # DVBS2HIP_EHIP: timed out (no rank 0)
    assert L.dvbs2hip_rendezvous(0, 2, str(tmp_path / "nobody").encode(), buf, 128, 100) == -3      # ... and no rank 1
    assert L.dvbs2hip_rendezvous(1, 2, str(tmp_path / "no_such_dir" / "x").encode(), buf, 128, 100) == -1
    assert L.dvbs2hip_rendezvous(0, 1, str(tmp_path / "alone").encode(), buf, 128, 100) == 0        # one rank: nothing to exchange
    assert L.dvbs2hip_rendezvous(2, 2, path.encode(), buf, 128, 100) == -1                          # rank outside the world
    assert sorted(os.listdir(tmp_path)) == ["rdv"]
