"""The LDPC plan of the normal-frame codes on CPU (tools/plan_probe.cpp links the library's host code; no GPU): which image mode the planner
picks, that the parked-row schedule (k_ldpc.hip plan_parked: on-chip set with the same number of LDS slots in every layer, pairs of rows sharing
an LDS position and a register slot, verified there by simulating a full cycle) exists for the DVB-S2 N = 64800 code, and the table's contract
with the kernel (the first NL slots of every layer, and only those, are LDS accesses)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    from dvbs2_amd import build
    build.build_lib()
    exe = str(tmp_path_factory.mktemp("plan") / "plan_probe")
    lib = os.path.join(ROOT, "dvbs2_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-I", os.path.join(ROOT, "dvbs2_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "plan_probe.cpp"), "-L", lib, "-ldvbs2hip", "-Wl,-rpath," + lib, "-o", exe], stderr=subprocess.DEVNULL)

    def run(modcod, spa=False, mode=None):
        env = dict(os.environ)
        env.pop("DVBS2HIP_LDPC_FAST_MODE", None)
        if mode:
            env["DVBS2HIP_LDPC_FAST_MODE"] = mode
        return subprocess.run([exe, modcod] + (["spa"] if spa else []), env=env, capture_output=True, text=True, check=True).stdout
    return run


@pytest.mark.parametrize("spa,mode,want,lds_slots,parked,glob_rows", [
    (False, None, 5, 15, 39, 180 - 55 - 39), (True, None, 4, 14, 32, 180 - 55 - 32), (False, "park4", 4, 14, 32, 93), (False, "static", 3, 9, 0, 125)])
def test_normal_frame_plan(probe, spa, mode, want, lds_slots, parked, glob_rows):
    out = probe("QPSK-N_8/9", spa, mode)
    head = out.splitlines()[0]
    assert "plan: ''" in head and "mode %d wg8 1 dups_in_lds 1" % want in head, head
    assert "LDS rows 55 (info 55) global rows %d " % glob_rows in head, head
    hyb = out.splitlines()[1]
    assert hyb.startswith("hybrid: %d LDS slots per layer ok | parked rows %d," % (lds_slots, parked)), hyb
    if parked:                      # a pair swaps twice per iteration and a swap moves two rows
        m = re.search(r"row moves per iteration (\d+), swaps in the table (\d+)", hyb)
        assert int(m.group(1)) == 4 * parked and int(m.group(2)) == 2 * parked, hyb


def test_short_frame_plan_is_all_lds(probe):
    assert "mode 0 wg8 1" in probe("QPSK-S_8/9").splitlines()[0]
