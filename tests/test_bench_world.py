"""bench.py's N > 1 control path before hardware meets it: the file the driver launches with `torch.distributed.run --nproc-per-node N`, here as N
processes on CPU -- gloo instead of RCCL, tests/bench_stub.py in place of the receiver (a test-side stand-in, not a fallback of the product: bench.py's default runtime
refuses to start without a GPU) -- world sizes 2 and 8: process group, barriers around the timed region, max over the ranks' times, the counters' all-reduce, the per-rank
all-gather, the extras on every rank, ONE JSON line from rank 0 and none from the others, and what the N > 1 line says about the figures that belong to the N = 1 line.
The reference's counterpart is the cross-clone monitor reduction of /root/reference src/mains/TX_RX_BB/main.cpp:118-125,155-161."""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, outdir, extras):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    import bench_stub
    sys.stdout = open(os.path.join(outdir, "out.%d" % rank), "w")
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--frames", str(FRAMES), "--quad-launches", "1", "--self-check-steps", "2", "--self-check-seconds", "0", "--no-cpu-baseline"]
    bench.main(argv + ([] if extras else ["--no-extras"]), runtime=bench_stub.StubRuntime)
    sys.stdout.flush()
    with open(os.path.join(outdir, "calls.%d" % rank), "w") as fh:
        json.dump(bench_stub.StubRx.calls, fh)


def _run(world, tmp_path, extras=True):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, str(tmp_path), extras)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, "a rank failed or hung"
    outs = [open(os.path.join(tmp_path, "out.%d" % r)).read() for r in range(world)]
    calls = [json.load(open(os.path.join(tmp_path, "calls.%d" % r))) for r in range(world)]
    return outs, calls


@pytest.mark.parametrize("world", [2, 8])
def test_bench_control_path_with_n_ranks(world, tmp_path):
    outs, calls = _run(world, tmp_path)
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][-2000:]
    assert all(not [l for l in o.splitlines() if l.startswith("{")] for o in outs[1:])          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["ber"]["FRA"] == world * FRAMES                                                       # the counters were summed over the ranks
    assert len(d["per_rank"]["fec_frames_per_s"]) == world and d["per_rank"]["min"] <= d["per_rank"]["max"]
    # value = all ranks' frames over the SLOWEST rank's time: never more than the sum of the ranks' own rates
    assert d["fec_frames_per_s"] <= sum(d["per_rank"]["fec_frames_per_s"]) * 1.0001
    assert abs(d["value"] - d["fec_frames_per_s"] * 57472) < 1e-6 * d["value"]
    # what belongs to the N = 1 line is named, not silently dropped
    assert d["cpu_baseline"] == "N=1 line only" and d["roofline"]["live_pmc"] == "N=1 line only"
    ex = d["extra"]
    assert "N=1 line only" in ex["host_socket_form"] and "N=1 line only" in ex["configs"]["4"] and set(ex["skipped_at_this_n"]) == {"configs.4", "host_socket_form", "ref_config", "sync_located"} and ex["spa"]["QPSK-N_8/9"]["frames"] == FRAMES
    assert ex["configs"]["2"]["bit_errors"] == 0 and ex["configs"]["3"]["n_ite"] == 20 and ex["natural_order"]["frames"] == FRAMES
    assert set(ex["four_way"]["variants"]) == {"4.0dB_fixed", "3.0dB_fixed", "4.0dB_stop", "3.0dB_stop"}
    # every rank went through the same extras in the same order (lock step: none of them holds a collective, so a rank that skipped one would not hang the others -- it would
    # leave rank 0's figures taken next to idle GPUs)
    assert all(c == calls[0] for c in calls[1:])
    assert ["set_ldpc_schedule", 1] in calls[0] and ["create", "16APSK-N_8/9"] in calls[0]


def test_bench_control_path_one_rank_under_a_process_group(tmp_path):
    """world = 1 with RANK set (what `torch.distributed.run --nproc-per-node 1` gives): the process group exists, the reductions are identities, nothing is skipped for N > 1."""
    outs, _ = _run(1, tmp_path, extras=False)
    d = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["ber"]["FRA"] == FRAMES and "cpu_baseline" not in d and d["extra"]["configs"] is None


def test_default_runtime_refuses_to_run_without_a_gpu():
    """the stand-in is opt-in: bench.py's own runtime has no fallback"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sys.path.insert(0, ROOT)
    import bench
    with pytest.raises(SystemExit) as e:
        bench.Runtime(0)
    assert "no CPU fallback" in str(e.value)
