"""Wall time per call of the three N4 synchronizers (sync_case of tools/bench_kernels.py) for one MODCOD: one line (tools/archive/r04_sync_ab.sh).
usage: python tools/sync_time.py [modcod] [frames]"""
import os, re, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
src = open(os.path.join(HERE, "bench_kernels.py")).read()
src = "\n".join(l for l in src.splitlines() if not re.match(r"^(res\[|out = |json\.dump|print\()", l))      # the definitions without the cases
ns = {"__name__": "bench_kernels_defs", "__file__": os.path.join(HERE, "bench_kernels.py")}
exec(compile(src, "bench_kernels.py", "exec"), ns)
modcod = sys.argv[1] if len(sys.argv) > 1 else "32APSK-S_3/4"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = ns["sync_case"](modcod, F, reps=20)
print("%s F=%d frame_sync %.4f ms  L&R %.4f ms  pilot %.4f ms  locked %s delay %d" % (modcod, F, r["frame_sync(corr+metric+delay)"]["call_ms"], r["luise_reggiannini"]["call_ms"],
                                                                                      r["pilot_freq_phase"]["call_ms"], r["locked"], r["delay_of_the_last_frame"]))
