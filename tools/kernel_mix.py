"""Instruction mix of a kernel's LAYER LOOP, read from the built code object (no GPU): how many of its vector instructions carry the 64-bit VOP3 / VOP3P encoding.
Why it matters (tools/probe_issue2.hip, profiles/r04_probe_issue.txt): a SIMD issues a VOP3-encoded instruction every ~4.1 cycles and a VOP1 / VOP2 / VOPC one every
~2.05, whatever the number of waves, so the vector-issue time of a kernel is N_vop2 x 2.05 + N_vop3 x 4.1 cycles per SIMD -- not N x 2 (round 3) and not N x 4 (round 2).
usage: python tools/kernel_mix.py [translation unit] [substring of the demangled kernel name]      e.g.  k_ldpc_wg8 "ldpc_wg8_kernel<27, 5, false>" """
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CYC_VOP2, CYC_VOP3 = 2.05, 4.1


def disassemble(tu):
    o = os.path.join(ROOT, "dvbs2_amd", "lib", tu + ".hip.o")
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb"), os.path.join(td, "co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, o])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co, "--unbundle"])
        return subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def functions(txt):
    out, cur = {}, None
    for l in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            cur = m.group(1); out[cur] = []; continue
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-F]+):\s*([0-9A-F]{8})", l)
        if cur and m:
            tgt = re.search(r"<\S+\+0x([0-9a-f]+)>\s*$", l)
            out[cur].append((int(m.group(2), 16), m.group(1), int(m.group(3), 16), int(tgt.group(1), 16) if tgt else None))
    return out


def classify(op, w0):
    if not op.startswith("v_"):
        return "other"
    return "vop3" if (w0 >> 26) == 0b110100 else "vop2"


def layer_loop(ins, marker="v_med3_f32"):
    """the smallest loop (backward branch) that holds most of the kernel's marker instructions"""
    base = ins[0][0]
    total = sum(1 for a, op, w, t in ins if op == marker)
    best = None
    for a, op, w, t in ins:
        if op.startswith(("s_branch", "s_cbranch")) and t is not None and base + t < a:
            lo, hi = base + t, a
            n = sum(1 for a2, op2, _, _ in ins if lo <= a2 <= hi and op2 == marker)
            if n >= 0.6 * total and (best is None or hi - lo < best[1] - best[0]):
                best = (lo, hi)
    return best


def mix(tu, pat, marker="v_med3_f32"):
    txt = disassemble(tu)
    res = []
    for name, ins in functions(txt).items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat not in dem or not ins:
            continue
        loop = layer_loop(ins, marker)
        sel = [i for i in ins if loop and loop[0] <= i[0] <= loop[1]] if loop else ins
        c = {"vop2": 0, "vop3": 0, "other": 0}
        for a, op, w, t in sel:
            c[classify(op, w)] += 1
        nv = c["vop2"] + c["vop3"]
        res.append({"kernel": re.sub(r"\(.*\)$", "", dem).replace("void dvbs2::", ""), "loop_bytes": (loop[1] - loop[0]) if loop else None, "valu": nv, "vop3": c["vop3"], "vop3_share": c["vop3"] / nv if nv else 0.0,
                    "other": c["other"], "cycles_per_valu": (c["vop2"] * CYC_VOP2 + c["vop3"] * CYC_VOP3) / nv if nv else None})
    return res


if __name__ == "__main__":
    tu = sys.argv[1] if len(sys.argv) > 1 else "k_ldpc_wg8"
    pat = sys.argv[2] if len(sys.argv) > 2 else "ldpc_wg8_kernel<27, 5, false>"
    for r in mix(tu, pat):
        print("%-50s layer loop %s bytes: %d vector instructions, %d (%.0f %%) VOP3-encoded -> %.2f SIMD cycles per vector instruction; %d scalar / memory / other"
              % (r["kernel"], r["loop_bytes"], r["valu"], r["vop3"], 100 * r["vop3_share"], r["cycles_per_valu"], r["other"]))
