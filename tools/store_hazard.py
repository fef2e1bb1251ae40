"""The store-data hazard of docs/negative_results.md, looked for in the code objects: a MUBUF / MTBUF store of more than 8 bytes whose soffset field is an SGPR, followed
with no instruction in between by a vector-ALU instruction that writes one of the store's data registers.  LLVM's hazard recognizer exempts exactly that form
(GCNHazardRecognizer: the wait state is inserted only when soffset is NOT a register); on gfx950 the stored data were seen to change (tools/det_check.py, round 3).
CPU-only: python tools/store_hazard.py [-v]   ->   one line per translation unit, the offending pairs listed; exit code 1 if there is any."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
WIDE = re.compile(r"^\s*(buffer_store_dwordx[34]|tbuffer_store_format_xyzw?|buffer_store_format_xyzw?)\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
INS = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//")


def vregs(op):
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", op)
    return {int(m.group(1))} if m else set()


def disassemble(obj):
    with tempfile.NamedTemporaryFile(suffix=".co") as tf, tempfile.NamedTemporaryFile(suffix=".fb") as fb:
        if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb.name, obj], capture_output=True).returncode:
            return None
        if subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb.name, "--output=" + tf.name, "--unbundle"], capture_output=True).returncode:
            return None
        return subprocess.run([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", tf.name], capture_output=True, text=True).stdout


def scan(txt):
    """-> (wide stores, of them with an SGPR soffset, [(kernel, store line, next line)] hazards)"""
    wide = sgpr = 0
    bad = []
    kern = "?"
    lines = txt.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            kern = m.group(1)
            continue
        mi = INS.match(l)
        if not mi:
            continue
        w = WIDE.match(mi.group(1) + " " + mi.group(2))
        if not w:
            continue
        wide += 1
        soff = w.group(5).rstrip(",")
        if not re.match(r"(s\d+|m0|ttmp\d+|vcc_lo|vcc_hi)$", soff):
            continue
        sgpr += 1
        data = set(range(int(w.group(2)), int(w.group(3)) + 1))
        nxt = next((INS.match(x) for x in lines[i + 1:i + 4] if INS.match(x)), None)
        if nxt and nxt.group(1).startswith("v_") and not nxt.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
            dst = nxt.group(2).split(",")[0].strip()
            if vregs(dst) & data:
                bad.append((kern, l.split("//")[0].strip(), lines[i + 1].split("//")[0].strip()))
    return wide, sgpr, bad


def report():
    out = {}
    for o in sorted(glob.glob(os.path.join(ROOT, "dvbs2_amd", "lib", "*.hip.o"))):
        txt = disassemble(o)
        if txt is not None:
            out[os.path.basename(o)[:-6]] = scan(txt)
    return out


if __name__ == "__main__":
    rc = 0
    for tu, (wide, sgpr, bad) in report().items():
        print("%-16s %4d buffer stores of 12 / 16 bytes, %3d with an SGPR soffset, %d followed by a write of their data registers" % (tu, wide, sgpr, len(bad)))
        for k, a, b in bad:
            rc = 1
            print("    %s:\n        %s\n        %s" % (subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90], a, b))
    sys.exit(rc)
