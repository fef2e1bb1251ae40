"""The bench line's contract (the driver and the judge read it): the committed PMC traffic file belongs to the kernel sources in the
tree (CPU), and on the GPU the JSON line carries what it must -- metric, value, the SURVEY 8(d) roofline, the physically bounded one,
the bytes that must cross HBM, the untimed extras -- with fractions that are fractions."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_pmc_traffic_belongs_to_the_kernel_in_the_tree():
    """profiles/ldpc_pmc_traffic.json is stamped with the hash of k_ldpc_wg8.hip + k_ldpc.hip it was measured on; bench.py refuses a
    stale one (emits null).  Editing the kernel or the plan without re-running tools/profile_gpu.sh + tools/summarize_profiles.py
    fails here, so the committed profile cannot silently go out of date."""
    sys.path.insert(0, ROOT)
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "ldpc_pmc_traffic.json")))
    assert d["kernel_sha"] == bench.kernel_sha(), "re-profile: the LDPC kernel / plan changed since profiles/ldpc_pmc_traffic.json was measured"
    assert d["frames"] == bench.FRAMES_PER_GPU and d["n_ite"] == bench.N_ITE
    assert 0.2e10 < d["hbm_bytes_per_launch"] < 1.3e11 and 0.0 < d["valu_occupancy"] <= 1.0
    t, m = bench._pmc_traffic(d["kernel"], bench.FRAMES_PER_GPU, bench.N_ITE)
    assert t == d["hbm_bytes_per_launch"] and m["kernel_sha"] == d["kernel_sha"]
    assert bench._pmc_traffic(d["kernel"], bench.FRAMES_PER_GPU + 1, bench.N_ITE)[0] is None          # another workload: no figure


def test_committed_ablation_run_belongs_to_the_kernel_in_the_tree():
    """profiles/ldpc_ablation.json (tools/build_ablations.sh, tools/run_ablations.sh, tools/summarize_ablations.py) carries the hash of the kernel + plan sources it was
    measured on; bench.py's `roofline.bound` / `chain_floor_ms` come from it only while it matches, so an edit of the kernel without a new ablation run fails here."""
    sys.path.insert(0, ROOT)
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "ldpc_ablation.json")))
    assert d["kernel_sha"] == bench.kernel_sha(), "re-run tools/run_ablations.sh: the LDPC kernel / plan changed since profiles/ldpc_ablation.json was measured"
    assert d["frames"] == bench.FRAMES_PER_GPU and d["n_ite"] == bench.N_ITE and set(d["ablations"]) >= {"3", "12", "15"}
    assert 0.4 * d["production_ms"] < d["chain_floor_ms"] < d["production_ms"]
    a = bench._ablation(bench.FRAMES_PER_GPU, bench.N_ITE)
    assert a["chain_floor_ms"] == d["chain_floor_ms"] and bench._ablation(bench.FRAMES_PER_GPU + 1, bench.N_ITE) is None


def test_committed_kernel_counters_belong_to_the_kernels_in_the_tree():
    """profiles/<round>_kernels_pmc.md (rocprofv3 counters of every non-LDPC kernel, tools/profile_kernels.sh + tools/summarize_kernels_pmc.py) is
    stamped with the hash of the kernel sources it was measured on (profiles/kernels_pmc_stamp.json): editing k_front / k_bch / k_sync* / k_fir* /
    k_tx without re-profiling fails here -- round 2's APSK front-end counters had silently gone stale."""
    import hashlib
    d = json.load(open(os.path.join(ROOT, "profiles", "kernels_pmc_stamp.json")))
    h = hashlib.sha256()
    for f in d["sources"]:
        h.update(open(os.path.join(ROOT, "dvbs2_amd", "csrc", f), "rb").read())
    assert set(d["sources"]) >= {"k_front.hip", "k_bch.hip", "k_sync.hip", "k_sync_mfma.hip", "k_fir.hip", "k_fir_mfma.hip", "k_tx.hip"}
    assert d["sha"] == h.hexdigest()[:16], "re-profile: a non-LDPC kernel changed since %s was measured" % d["file"]
    assert os.path.exists(os.path.join(ROOT, d["file"])) and d["sha"] in open(os.path.join(ROOT, d["file"])).read()


@pytest.mark.gpu
def test_bench_line_on_the_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--self-check-steps", "20", "--self-check-seconds", "0", "--quad-launches", "4", "--ref-config-frames", "400000"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["fec_frames_per_s"] * 57472) < 1e-3 * d["value"]
    assert d["ber"]["FRA"] == 4096 and d["ber"]["FE"] == 0                 # what was timed decoded its batch
    ro = d["roofline"]
    assert ro["avg_launch_ms"] <= d["ms_per_step"] * 1.02
    assert 0.0 < ro["hbm_true"]["frac"] < 0.1
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and ro["unit"] == "GB/s"          # tied to a measured hardware resource: counter bytes / launch time / HBM peak
    assert abs(ro["algorithmic_frac"] - ro["algorithmic_GBps"] / 8000.0) < 1e-9            # SURVEY 8(d)'s effective figure, kept beside the measured one
    assert d["per_rank"]["fec_frames_per_s"] and d["per_rank"]["min"] <= d["fec_frames_per_s"] * 1.001 <= d["per_rank"]["max"] * 1.002
    assert d["self_check"]["steps"] == 20 and 0.5 * d["ms_per_step"] < d["self_check"]["ms_per_step"] < 1.5 * d["ms_per_step"]
    if ro["traffic"] is not None:
        assert abs(ro["achieved"] - ro["traffic"] / (ro["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * ro["achieved"] and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
        assert 0.3 < ro["frac"] <= 1.0
        rs = ro["resources"]
        assert 0.0 < rs["fabric"]["frac"] <= 1.0 and abs(rs["fabric"]["achieved_GBps"] - ro["achieved"]) < 1e-6 * ro["achieved"]
        if rs["vector_issue"]["frac"] is not None:                         # (None: the committed SQ passes belong to other kernel sources until tools/gpu_final_pass.sh has been run again)
            assert 0.0 < rs["vector_issue"]["frac"] <= 1.0 and "upper" in rs["vector_issue"]["estimate"]
    else:
        assert ro["bounded"] is None and ro["frac"] is None
    if ro.get("ablation"):
        # the committed ablation run's floor is reported as a fraction of ITS OWN production launch (another box's clock is never divided by this run's)
        a = ro["ablation"]
        assert abs(ro["chain_floor_frac"] - a["chain_floor_ms"] / a["production_ms"]) < 1e-12 and 0.4 < ro["chain_floor_frac"] < 1.0
    assert ro["hbm_copy_GBps_measured"] > 3000.0 and "dvbs2hip_device_copy_bandwidth" in ro["hbm_copy_kernel"]      # the library's own copy kernel, not a torch copy_
    # the fabric bytes are re-measured in the run itself (rocprofv3 --pmc child runs of bench.py) and agree with the committed, sha-stamped passes
    # (a profiler that cannot run on this box is reported in the line, not a failure: the committed figures then stand alone, as asserted above)
    lp = ro["live_pmc"]
    assert lp is not None
    if "error" not in lp:
        assert lp["launches_counted"] >= 3 and ro["traffic"] == lp["hbm_bytes_per_launch"], lp
        if lp.get("live_over_committed") is not None:          # (None: the committed passes belong to other kernel sources, see above)
            assert 0.9 < lp["live_over_committed"] < 1.1, lp
    else:
        import warnings
        warnings.warn("bench.py could not re-measure its PMC figures in this run: %s" % lp["error"])
        assert ro["traffic"] is not None
    ex = d["extra"]
    assert set(ex["early_stop_fps"]) == {"4.0 dB", "3.0 dB"} and ex["early_stop_fps"]["4.0 dB"] > d["fec_frames_per_s"]       # converging frames stop early
    hb = ex["hard_batch_fixed_10_ite"]
    assert hb["frames"] == 4096 and hb["cwd"] <= 4 and hb["FE"] >= hb["frames"] - 4            # SURVEY 8(d) config 2's second batch at the size of `value`
    fw = ex["four_way"]["variants"]
    assert set(fw) == {"4.0dB_fixed", "3.0dB_fixed", "4.0dB_stop", "3.0dB_stop"} and all(v["launches"] == 4 and v["frames"] == 4096 for v in fw.values())
    # fixed iterations cost the same whatever the data (no data-dependent branch in the layer loop): the anomaly VERDICT r3 found on 1024 frames / 3 launches
    assert 0.9 < ex["four_way"]["hard_over_easy_fixed"] < 1.1
    # ... and on frames that never converge the stopping rule costs what fixed iterations cost, to a few percent: ten sweeps that stop at their first layer (half a
    # syndrome pass in all) against the one full pass a fixed-iteration decode ends with ("the 550 k against 695 k of round 3")
    assert 0.93 < fw["3.0dB_stop"]["ms_mean"] / fw["3.0dB_fixed"]["ms_mean"] < 1.07
    cf = ex["configs"]
    assert set(cf) == {"2", "3", "4"} and cf["2"]["bit_errors"] == 0 and cf["3"]["n_ite"] == 20 and cf["3"]["bit_errors"] == 0
    assert [r["frames"] for r in cf["4"]["per_F"]] == [1, 8, 64, 4096] and all(r["fir_GFLOPs_fp32_equiv"] > 0 for r in cf["4"]["per_F"])
    assert all(r["frames_decoded_exactly"] == r["frames"] for r in cf["4"]["per_F"])          # (the stream is one frame longer than the call: every decoded frame is whole)
    assert ex["natural_order_fps"] > 0 and ex["host_socket_form"]["fec_frames_per_s"] > 0
    # the reference's default decoder, its own configuration and the located synchronizer under this clock; the chain's floor in the line
    sp = ex["spa"]
    assert sp["QPSK-N_8/9"]["frames"] == 4096 and sp["QPSK-N_8/9"]["implem"] == "SPA" and sp["QPSK-N_8/9"]["fec_frames_per_s"] > 2.0e5 and sp["QPSK-N_8/9"]["cwd"] >= 4090
    assert sp["QPSK-S_8/9"]["frames"] == 8192 and sp["QPSK-S_8/9"]["fec_frames_per_s"] > 1.0e6 and sp["QPSK-S_8/9"]["kernel"] == "ldpc_wg8_kernel<27,0,3>"
    assert sp["QPSK-S_8/9 SPA_TANH"]["kernel"] == "ldpc_wg8_kernel<27,0,2>" and abs(sp["QPSK-S_8/9 SPA_TANH"]["cwd"] - sp["QPSK-S_8/9"]["cwd"]) <= 8          # (two rules: nearly, not exactly, the same frames converge in ten iterations)
    rc = ex["ref_config"]
    for k in ("clones_1", "clones_3"):
        assert rc[k]["fra"] >= 400000 and 1.0 / 2.5 < rc[k]["fer_over_ref"] < 2.5 and rc[k]["info_Gbps"] > 5.0, rc[k]
    sl = ex["sync_located"]
    assert sl["32APSK-S_3/4"]["frames"] == 4096 and sl["QPSK-N_8/9"]["frames"] == 1024 and all(0.0 < v["ms_per_call"] < 1.0 and v["flag"] == 1 for v in sl.values())
    nl = ex["normal_frame_latency"]["per_F"]
    assert [r["frames"] for r in nl] == [1, 64] and all(r["kernel"] == "ldpc_cu1_kernel<27>" and 0.05 < r["latency_ms_early_stop"] <= r["latency_ms_fixed_10"] < 0.6 for r in nl), nl
    for k in ("2", "3"):
        assert cf[k]["floor_ms"] > cf[k]["ldpc_kernel_ms"] > 0 and 1.0 <= cf[k]["tail_over_floor"] < 1.25, cf[k]


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank():
    """The launch line the driver uses for N > 1, with one rank (this pool has one GPU per box): `python -m torch.distributed.run ... bench.py --gpus 1`
    as a FRESH child process (never a re-exec of one that has touched the GPU): RCCL process group, the counters' all-reduce, the max over ranks
    of the timing, one JSON line from rank 0.  What it cannot show is scaling: N > 1 is unmeasured on this pool."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29517",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--self-check-steps", "0", "--self-check-seconds", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["ber"]["FRA"] == 4096 and d["ber"]["FE"] == 0
    assert "cpu_baseline" not in d and d["self_check"] is None
