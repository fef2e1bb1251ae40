"""Register / spill / LDS figures of every kernel in libdvbs2hip.so's objects, read from the code objects' metadata
(llvm-readelf --notes).  CPU-only: python tools/kernel_regs.py [pattern]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels():
    out = []
    for o in sorted(glob.glob(os.path.join(ROOT, "dvbs2_amd", "lib", "*.hip.o"))):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf, tempfile.NamedTemporaryFile(suffix=".fb") as fb:
            r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb.name, o], capture_output=True)
            if r.returncode:
                continue
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb.name, "--output=" + tf.name, "--unbundle"], capture_output=True)
            if r.returncode:
                continue
            txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", tf.name], capture_output=True, text=True).stdout
        for b in txt.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
            name = g("name")
            if name == "?":
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*\)$", "", dem).replace("void dvbs2::", "")
            out.append(dict(file=os.path.basename(o)[:-6], name=dem, vgpr=int(g("vgpr_count")), vgpr_spill=int(g("vgpr_spill_count")), sgpr=int(g("sgpr_count")),
                            sgpr_spill=int(g("sgpr_spill_count")), lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%-16s %-60s %5s %6s %5s %6s %7s %7s" % ("file", "kernel", "vgpr", "vspill", "sgpr", "sspill", "lds", "scratch"))
    for k in kernels():
        if pat in k["name"]:
            print("%-16s %-60s %5d %6d %5d %6d %7d %7d" % (k["file"], k["name"][:60], k["vgpr"], k["vgpr_spill"], k["sgpr"], k["sgpr_spill"], k["lds"], k["scratch"]))
