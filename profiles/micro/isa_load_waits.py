"""Lists, per kernel, the vector loads that are waited for on the spot (an `s_waitcnt vmcnt(0)` within a few instructions of the
load): the signature of a load the compiler sank into a branch, or of conditionally issued loads it cannot count (DESIGN.md
section 8). Compiles every .hip of epnet_amd/csrc to gfx950 assembly with the Makefile's flags (no GPU needed).

    python profiles/micro/isa_load_waits.py [min_tight_loads]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "epnet_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only"]


def main():
    least = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            out = os.path.join(tmp, name + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, os.path.join(CSRC, name)], check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
            for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
                lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
                loads = tight = 0
                for i, l in enumerate(lines):
                    if l.startswith(("global_load", "buffer_load")):
                        loads += 1
                        if any("s_waitcnt" in x and "vmcnt(0)" in x for x in lines[i + 1:i + 5]):
                            tight += 1
                if tight >= least:
                    demangled = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                    print("%-16s %4d loads, %3d waited for on the spot   %s" % (name, loads, tight, demangled[:110]))


if __name__ == "__main__":
    main()
