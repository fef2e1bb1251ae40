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
    (False, None, 5, 15, 39, 180 - 55 - 39), (True, "park4", 4, 14, 32, 180 - 55 - 32), (False, "park4", 4, 14, 32, 93), (False, "static", 3, 9, 0, 125)])
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


def test_mode6_one_frame_per_cu_plan(probe):
    """Mode 6 (k_ldpc_cu1.hip, DVBS2HIP_LDPC_FAST_MODE=cu1): every bit-group row of the N = 64800 8/9 code on chip -- 108 LDS positions + 72 pairs of rows that share a position and
    a register slot of the row-keeping waves (a MAXIMUM matching: greedy finds 71 of the 72 pairs among the information rows, so parity rows pair too).  tools/plan_probe.cpp
    replays one cycle of the tables the way the kernel reads them: every slot of every layer is an LDS access whose base is the position that holds its row in THAT layer, the
    duplicate edges are the first slots (conflict entry i = slot i), p_c / p_{c-1} the last two, no swap touches a row the current layer uses, the cycle closes, and the
    start-of-frame placement of the parity groups (position or register slot) is consistent."""
    assert probe("QPSK-N_8/9", True, None).splitlines()[0] == probe("QPSK-N_8/9", True, "cu1").splitlines()[0]      # (round 5) the sum-product decoder's default on normal frames
    assert "gwork words %d" % (27 * 7200) in probe("QPSK-N_8/9", True, None).splitlines()[0]                          # its messages: one word per edge, inside the Infinity Cache
    out = probe("QPSK-N_8/9", False, "cu1")
    head = out.splitlines()[0]
    assert "plan: ''" in head and "mode 6 wg8 1 dups_in_lds 1" in head and "LDS rows 108 (info 108) global rows 0" in head, head
    cu1 = [l for l in out.splitlines() if l.startswith("cu1:")][0]
    assert cu1 == "cu1: positions 108 pairs 72 swaps per iteration 144 max duplicate edges per layer 3 tables ok", cu1


def test_natural_order_hazard_planes_are_empty_for_the_dvbs2_codes(probe):
    """k_ldpc_nat.hip requests a check's posteriors NAT_AHEAD (24) checks before it is computed, which is legal only if no check shares a bit with one of the 25 before
    it (cyclically, the forwarded parity bit aside): the host's second hazard plane.  It is empty for every code of the library (a code for which it is not gets the
    one-check-ahead kernel behind a drain)."""
    for modcod in ("QPSK-N_8/9", "QPSK-S_8/9", "QPSK-S_3/5", "32APSK-S_3/4"):
        line = [l for l in probe(modcod).splitlines() if l.startswith("natural order:")][0]
        assert " 0 of " in line and ", 0 with one of the 25 before them" in line, line
