"""Instruction mix of a kernel's LAYER LOOP, read from the built code object (no GPU): how many of its vector instructions carry the 64-bit VOP3 / VOP3P encoding.
Why it matters (tools/probe_issue2.hip, profiles/r04_probe_issue.txt): a SIMD issues a VOP3-encoded instruction every ~4.1 cycles and a VOP1 / VOP2 / VOPC one every
~2.07, whatever the number of waves -- and so does a VOP2 instruction that reads an SGPR; the transcendentals take 8 -- so the vector-issue time of a kernel is the sum of
its instructions' prices per SIMD, not N x 2 (round 3) and not N x 4 (round 2).
usage: python tools/kernel_mix.py [translation unit] [substring of the demangled kernel name]      e.g.  k_ldpc_wg8 "ldpc_wg8_kernel<27, 5, 0>" """
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# SIMD cycles per instruction with four waves on the SIMD (profiles/r04_probe_issue.txt, probe_issue2 / 3 / 4): only the SIMPLE two-operand class -- v_mov_b32, v_and / v_or / v_xor_b32,
# v_add / v_sub / v_mul_f32, v_add_u32 with register or inline-constant operands -- issues every 2.07 cycles (two waves at full speed side by side); the same with a 32-bit literal 2.6;
# EVERYTHING ELSE 4.2-4.25, the waves of the SIMD one after the other: the VOP3 / VOP3P encoding (v_fma_f32, v_med3_f32, v_pk_add_f32, a VOP2 opcode with |abs| ...), SDWA and DPP,
# an SGPR operand, VOPC (v_cmp_*_e32 writes vcc) and the vcc readers (v_cndmask_b32_e32, v_addc_co_u32), and -- probe_issue4, measured after the first pricing of this round -- plain
# VOP1 / VOP2 opcodes outside the simple class: v_min / v_max (f32 and u32), v_lshlrev_b32, v_cvt_*, v_mul_u32_u24, v_fmac_f32.  v_exp / v_log / v_rcp ... 8.06.
CYC = {"plain": 2.07, "slow": 4.25, "vop3": 4.2, "sgpr": 4.25, "literal": 2.6, "trans": 8.06}
CYC_VOP2, CYC_VOP3 = CYC["plain"], CYC["vop3"]
FAST = ("v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_exp_legacy_f32", "v_log_legacy_f32")


def disassemble(tu):
    o = os.path.join(ROOT, "dvbs2_amd", "lib", tu + ".hip.o")
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb"), os.path.join(td, "co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, o], stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co, "--unbundle"])
        return subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def functions(txt):
    out, cur = {}, None
    for l in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            cur = m.group(1); out[cur] = []; continue
        m = re.match(r"^\s+(\S+)\s(.*?)//\s*([0-9A-F]+):\s*([0-9A-F]{8})( [0-9A-F]{8})?", l)
        if cur and m:
            tgt = re.search(r"<\S+\+0x([0-9a-f]+)>\s*$", l)
            out[cur].append((int(m.group(3), 16), m.group(1), int(m.group(4), 16), int(tgt.group(1), 16) if tgt else None, m.group(2), m.group(5) is not None))
    return out


def classify(op, w0, operands="", two_words=False):
    """other | trans | vop3 | sgpr | slow | literal | plain"""
    if not op.startswith("v_"):
        return "other"
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in TRANS:
        return "trans"
    if (w0 >> 26) == 0b110100:
        return "vop3"
    srcs = operands.split(",")[1:] if not base.startswith("v_cmp") else operands.split(",")      # (VOPC e32 writes vcc implicitly: every listed operand is a source)
    if any(re.search(r"(^|[^a-z])s\d+|s\[\d+:\d+\]|ttmp|m0", x) for x in srcs):
        return "sgpr"
    if base not in FAST or op.endswith(("_sdwa", "_dpp")):
        return "slow"
    if two_words and (w0 & 0x1FF) == 0xFF:
        return "literal"
    return "plain"


def layer_loop(ins, marker="v_med3_f32"):
    """the smallest loop (backward branch) that holds most of the kernel's marker instructions"""
    base = ins[0][0]
    total = sum(1 for i in ins if i[1] == marker)
    if total == 0:
        return None
    best = None
    for a, op, w, t, _o, _w2 in ins:
        if op.startswith(("s_branch", "s_cbranch")) and t is not None and base + t < a:
            lo, hi = base + t, a
            n = sum(1 for i2 in ins if lo <= i2[0] <= hi and i2[1] == marker)
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
        c = {k: 0 for k in list(CYC) + ["other"]}
        for a, op, w, t, opnds, w2 in sel:
            c[classify(op, w, opnds, w2)] += 1
        nv = sum(c[k] for k in CYC)
        res.append({"kernel": re.sub(r"\(.*\)$", "", dem).replace("void dvbs2::", ""), "loop_bytes": (loop[1] - loop[0]) if loop else None, "valu": nv, "vop3": c["vop3"], "vop3_share": c["vop3"] / nv if nv else 0.0,
                    "mix": {k: c[k] for k in CYC}, "other": c["other"], "cycles_per_valu": sum(c[k] * CYC[k] for k in CYC) / nv if nv else None})
    return res


_CACHE = {}


def price_kernel(kernel_name):
    """SIMD cycles per NON-transcendental vector instruction of the kernel whose demangled name contains `kernel_name` (rocprofv3's kernel name, arguments and
    `void` stripped), from the static mix of its layer loop (the loop that holds its v_med3_f32, if it has one) or of the whole function.  None if not found."""
    import glob
    base = re.sub(r"^void\s+", "", kernel_name).split("(")[0].replace("dvbs2::", "").strip()
    base_n = re.sub(r"\s+", "", base)
    for o in sorted(glob.glob(os.path.join(ROOT, "dvbs2_amd", "lib", "*.hip.o"))):
        tu = os.path.basename(o)[:-6]
        if tu not in _CACHE:
            try:
                fs = functions(disassemble(tu))
                _CACHE[tu] = {n: (i, re.sub(r"\s+", "", re.sub(r"\(.*\)$", "", subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()).replace("void dvbs2::", "").replace("dvbs2::", ""))) for n, i in fs.items() if i}
            except Exception:
                _CACHE[tu] = {}
        for n, (ins, dem) in _CACHE[tu].items():
            if dem == base_n or dem.replace(",false>", ">") == base_n or dem.replace(",0>", ">") == base_n:
                loop = layer_loop(ins)
                sel = [i for i in ins if loop and loop[0] <= i[0] <= loop[1]] if loop else ins
                c = {k: 0 for k in list(CYC) + ["other"]}
                for a, op, w, t, opnds, w2 in sel:
                    c[classify(op, w, opnds, w2)] += 1
                nt = sum(c[k] for k in CYC if k != "trans")
                if nt:
                    return sum(c[k] * CYC[k] for k in CYC if k != "trans") / nt, {"tu": tu, "scope": "layer loop" if loop else "whole kernel", "mix": {k: c[k] for k in CYC}}
    return None, None


if __name__ == "__main__":
    tu = sys.argv[1] if len(sys.argv) > 1 else "k_ldpc_wg8"
    pat = sys.argv[2] if len(sys.argv) > 2 else "ldpc_wg8_kernel<27, 5, 0>"
    for r in mix(tu, pat):
        print("%-50s layer loop %s bytes: %d vector instructions %s -> %.2f SIMD cycles per vector instruction; %d scalar / memory / other"
              % (r["kernel"], r["loop_bytes"], r["valu"], r["mix"], r["cycles_per_valu"], r["other"]))
