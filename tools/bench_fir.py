"""FIR (a5) vector kernel vs matrix-core kernel: device time, roofline fractions and error against a float64 convolution.
usage: python tools/bench_fir.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(mode):
    import numpy as np, torch
    from dvbs2_amd.receiver import Dvbs2Hip
    from dvbs2_amd import lib_binding as B, params as P
    dev = torch.device("cuda", 0)
    taps = P.rrc_taps(0.2, 2, 20).astype(np.float64)
    out = []
    for n_cplx, F in ((6804, 1), (6804, 64), (6804, 4096), (66564, 1024)):
        rx = Dvbs2Hip("32APSK-S_3/4", max_frames=max(F, 1))
        rx.set_filter_kernel(B.FIR_VALU if mode == "valu" else B.FIR_MFMA)
        g = torch.Generator(device=dev); g.manual_seed(1)
        x = torch.randn((F, 2 * n_cplx), dtype=torch.float32, device=dev, generator=g); y = torch.empty_like(x)
        rx.filter_reset(); rx.filter_dev(x.data_ptr(), y.data_ptr(), n_cplx, F); rx.synchronize()
        # error on the first 40000 complex samples against float64 (zero history)
        m = min(40000, n_cplx * F)
        xs = x.view(-1)[: 2 * m].cpu().numpy().astype(np.float64).reshape(-1, 2)
        ref = np.stack([np.convolve(xs[:, 0], taps)[:m], np.convolve(xs[:, 1], taps)[:m]], axis=1)
        err = float(np.max(np.abs(y.view(-1)[: 2 * m].cpu().numpy().reshape(-1, 2) - ref)))
        rx.timing_enable(True); rx.timing_reset()
        for _ in range(20): rx.filter_dev(x.data_ptr(), y.data_ptr(), n_cplx, F)
        ms, cnt = rx.timing_get(B.K_FIR); rx.timing_enable(False); ms /= cnt
        n = n_cplx * F
        out.append({"n_cplx": n_cplx, "frames": F, "kernel_ms": ms, "GBps": 16 * n / ms / 1e6, "frac_hbm_8TBps": 16 * n / ms / 1e6 / 8000,
                    "fp32_equiv_TFLOPs": 324 * n / ms / 1e9, "max_abs_err_vs_f64": err})
        rx.close()
    return out

if __name__ == "__main__":
    res = {mode: run(mode) for mode in ("valu", "mfma")}
    for mode, rows in res.items():
        for r in rows: print(mode, r["n_cplx"], r["frames"], "%.4f ms  %.0f GB/s (%.2f of 8 TB/s)  %.1f fp32-equivalent TFLOP/s  max err vs f64 %.2e" % (r["kernel_ms"], r["GBps"], r["frac_hbm_8TBps"], r["fp32_equiv_TFLOPs"], r["max_abs_err_vs_f64"]))
    if len(sys.argv) > 1: json.dump(res, open(sys.argv[1], "w"), indent=1)
