"""TEST-SIDE stand-in for the receiver, used only by tests/test_bench_world.py to run bench.py's N > 1 CONTROL PATH (process group, barriers, reductions, per-rank gather,
rank-0 printing, extras in lock step) on CPU with gloo.  It is NOT a CPU implementation of the product: it decodes nothing (hard decisions of the channel LLRs stand in for
the decoder's output, the "chain" hands back what the "TX mirror" sent) and lives under tests/ -- nothing in dvbs2_amd/ or bench.py's default path imports it."""
import ctypes as C
import types

import numpy as np
import torch

from dvbs2_amd import params as P

B = types.SimpleNamespace(K_LDPC=0, K_BCH=1, K_FRONT=4, SCHED_QC=0, SCHED_NATURAL=1)


def _view(ptr, n, ctype, dtype):
    return np.frombuffer((ctype * n).from_address(int(ptr)), dtype=dtype)


class StubRx:
    calls = []          # (rank-local) log of the entry points bench.py drove, in order

    def __init__(self, modcod, max_frames=1, n_ite=10, alpha=1.0, early_stop=False, device=0, **_):
        mc = P.get_modcod(modcod)
        self.mc, self.max_frames = mc, max_frames
        self.N_ldpc, self.K_ldpc, self.K_bch = mc.N_ldpc, mc.K_ldpc, mc.K_bch
        self.pl_frame = 90 + mc.N_ldpc // mc.bps + 36 * ((mc.N_ldpc // mc.bps // 90 - 1) // 16)
        rp, ad = P.load_ldpc_table(mc.ldpc_table)
        self.ldpc_edges = 360 * len(ad) + 2 * (mc.N_ldpc - mc.K_ldpc) - 1
        self._n, self._timing, self._sent, self._sched = 0, False, {}, B.SCHED_QC
        StubRx.calls.append(("create", modcod))

    def decode_siho_dev(self, llr, cwd, bits, F):
        x = _view(llr, F * self.N_ldpc, C.c_float, np.float32).reshape(F, self.N_ldpc)
        _view(bits, F * self.K_ldpc, C.c_int32, np.int32).reshape(F, self.K_ldpc)[:] = x[:, :self.K_ldpc] < 0
        if cwd:
            _view(cwd, F, C.c_int8, np.int8)[:] = 0
        self._n += 1

    def tx_bb_dev(self, info_in, seed, sigma, info_out, pl, F):
        sent = np.random.default_rng(seed).integers(0, 2, (F, self.K_bch)).astype(np.int32)
        _view(info_out, F * self.K_bch, C.c_int32, np.int32)[:] = sent.reshape(-1)
        self._sent[int(pl)] = sent
        StubRx.calls.append(("tx_bb_dev", F))

    def rx_bb_dev(self, pl, sigma, info, cwd_l, cwd_b, F):
        _view(info, F * self.K_bch, C.c_int32, np.int32)[:] = self._sent[int(pl)].reshape(-1)
        self._n += 1

    def synchronize(self):
        pass

    def timing_enable(self, on=True):
        self._timing = on

    def timing_reset(self):
        self._n = 0

    def timing_get(self, k):
        return 1.0 * self._n, self._n

    def set_ldpc_params(self, n_ite, alpha=1.0, early_stop=False):
        pass

    def set_ldpc_schedule(self, s):
        self._sched = s
        StubRx.calls.append(("set_ldpc_schedule", s))

    def ldpc_kernel_name(self):
        return "stub (tests/bench_stub.py)" if self._sched == B.SCHED_QC else "stub natural order"

    def device_copy_GBps(self, nbytes=1 << 30, reps=5):
        return 1.0

    def close(self):
        StubRx.calls.append(("close",))


class StubRuntime:
    def __init__(self, local_rank):
        self.torch, self.B, self.Dvbs2Hip = torch, B, StubRx
        self.dev = torch.device("cpu")
        self.backend, self.is_stub = "gloo", True

    def sync(self):
        pass

    def init_process_group(self, dist):
        dist.init_process_group(self.backend)

    def empty_cache(self):
        pass
