#!/usr/bin/env python
"""Per-kernel resources read from the gfx950 code objects of the built library: VGPRs, SGPRs, spilled registers, private
(scratch) and LDS bytes -- the evidence for "no spills" claims (profiles/rNN_kernel_meta.txt).

    python scripts/kernel_meta.py [object files ...]      default: sugar_amd/build/*.o
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        if subprocess.call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], stderr=subprocess.DEVNULL) != 0:
            return []  # no device code in this object
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in re.split(r"- \.agpr_count:", notes)[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        out.append((name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                    g("group_segment_fixed_size")))
    return out


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "sugar_amd", "build", "*.o")))
    print(f"{'kernel':78s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for o in objs:
        print(f"# {os.path.relpath(o, ROOT)}")
        for k in sorted(kernels(o)):
            print(f"{k[0][:78]:78s} {k[1]:>5s} {k[2]:>5s} {k[3]:>6s} {k[4]:>6s} {k[5]:>7s} {k[6]:>6s}")


if __name__ == "__main__":
    main()
