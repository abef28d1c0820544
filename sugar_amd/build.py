"""Build recipe for the gfx950 rasterizer library (sugar_amd/libsugar_raster.so).

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
The library is built IN-TREE (git-ignored, but shipped to the GPU box by gpurun).

    python -m sugar_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("SGR_BUILD_OUT") or os.path.join(HERE, "libsugar_raster.so")  # (SGR_BUILD_OUT + SGR_EXTRA_DEFS: variant builds for A/B runs)
ARCH = "gfx950"

# translation unit -> extra flags.  The per-Gaussian and binning kernels carry the pixel-exact contract
# (individually rounded IEEE ops, see csrc/preprocess.hip); the blend kernels may contract to FMA.
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"],
    "binning.hip": ["-ffp-contract=off"],
    "binning2.hip": ["-ffp-contract=off"],
    "blend.hip": ["-fno-slp-vectorize"] + os.environ.get("SGR_BLEND_DEFS", "").split(),  # the auto-formed v_pk_* pairs cost more v_mov shuffles than they save
    "knn.hip": ["-ffp-contract=off"],
    "mesh_raster.hip": ["-ffp-contract=off"],  # bit-exact with oracle/mesh_rasterizer.c
    "loss.hip": [],
    "adam.hip": [],
    "activations.hip": [],
    "field.hip": [],
    "pick.hip": [],
    "capi.hip": [],
    "train.hip": [],
}
COMMON = (["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
          + os.environ.get("SGR_EXTRA_DEFS", "").split())


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the gfx950 rasterizer cannot be built")


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/sugar_raster.h"]:
        p = os.path.join(CSRC, name)
        if os.path.isfile(p):
            h.update(name.encode())
            h.update(open(p, "rb").read())
    h.update(repr((SOURCES, COMMON)).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    stamp = OUT + ".stamp"
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build") if OUT.endswith("libsugar_raster.so") else OUT + ".build"
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    # (variant builds for same-box A/B runs: SGR_SRC_OVERRIDE="preprocess.hip=/tmp/old_preprocess.hip,..." compiles another version of
    # a translation unit against the current headers)
    override = dict(kv.split("=", 1) for kv in os.environ.get("SGR_SRC_OVERRIDE", "").split(",") if "=" in kv)
    for src, extra in SOURCES.items():
        path = override.get(src) or os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *extra, f"-I{CSRC}", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT, *objs]
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return OUT


# ---- the PyTorch C++ extension `_C` (csrc/torch_ext.cpp): host code only, compiled with g++ against the torch headers and linked to
# libsugar_raster.so by name (rpath $ORIGIN); built in-tree next to it
EXT_OUT = os.path.join(HERE, "_C_ext.so")


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "torch_ext.cpp")
    h = hashlib.sha256()
    for p in (src, os.path.join(HERE, "..", "include", "sugar_raster.h")):
        h.update(open(p, "rb").read())
    h.update(torch.__version__.encode())
    dig = h.hexdigest()
    stamp = EXT_OUT + ".stamp"
    if not force and os.path.exists(EXT_OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return EXT_OUT
    if not os.path.exists(os.path.join(HERE, "libsugar_raster.so")):
        build()
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    try:
        inc = ce.include_paths(device_type="cuda")
    except TypeError:
        inc = ce.include_paths(True)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C_ext",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           *["-I" + i for i in inc], "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include", src, "-o", EXT_OUT,
           "-L" + libdir, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
           "-L" + HERE, "-l:libsugar_raster.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,$ORIGIN", "-Wl,--no-as-needed"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("building the torch C++ extension failed:\n" + r.stdout.decode())
    with open(stamp, "w") as f:
        f.write(dig)
    return EXT_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ext(force="--force" in sys.argv, verbose=True))
