"""Reduce the PMC passes of scripts/r04_on_box.sh to profiles/pmc_blend_fwd.json (what bench.py's roofline.traffic and roofline_valu
read): per launch of k_blend_fwd_w, HBM bytes from FETCH_SIZE / WRITE_SIZE with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64: doubled; WRITE_SIZE as reported; KiB units) and
the wave-instruction count from SQ_INSTS_VALU.
    python scripts/pmc_reduce.py gpurun_out/r05 [workload] [file prefix] > profiles/pmc_blend_fwd[_<workload>].json
(workload: the bench's --workload the passes ran with, default "metric"; prefix: "pmc_" or "pmc_<workload>_" in front of the counter
name in the pass files)"""
import json
import os
import sys


def avg(path, counter, kernel):
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[4] == counter and kernel in line:
            return float(f[1]), float(f[2]), float(f[3])
    raise SystemExit(f"{counter} of {kernel} not found in {path}")


d = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "metric"
prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc_"
tag = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(os.path.normpath(d))   # the rNN the pass files are committed under
fetch = avg(os.path.join(d, prefix + "FETCH_SIZE.txt"), "FETCH_SIZE", "k_blend_fwd_w")
write = avg(os.path.join(d, prefix + "WRITE_SIZE.txt"), "WRITE_SIZE", "k_blend_fwd_w")
try:
    valu = avg(os.path.join(d, prefix + "SQ_INSTS_VALU.txt"), "SQ_INSTS_VALU", "k_blend_fwd_w")
except (SystemExit, OSError):
    valu = (None, None, None)   # (the instruction-count pass is taken for the metric workload only)
kname = "k_blend_fwd_wx" if "k_blend_fwd_wx" in open(os.path.join(d, prefix + "FETCH_SIZE.txt")).read() else "k_blend_fwd_w"  # (wx: exact alpha, the default since round 6)
out = {
    "kernel": kname, "workload": workload, "workload_key": workload,
    "source": [f"profiles/{tag}_{prefix}FETCH_SIZE.txt", f"profiles/{tag}_{prefix}WRITE_SIZE.txt"],
    "FETCH_SIZE_KiB_per_launch": fetch[0], "FETCH_SIZE_KiB_min_max": fetch[1:], "WRITE_SIZE_KiB_per_launch": write[0],
    "correction": "gfx950 rocprofv3 FETCH_SIZE counts 128-B requests as 64 B: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "hbm_bytes_per_launch": (2.0 * fetch[0] + write[0]) * 1024.0,
    "valu_wave_insts_per_launch": valu[0], "simd_issue_interval_ns": 1.25,
    "valu_source": [f"profiles/{tag}_{prefix}SQ_INSTS_VALU.txt", "profiles/r02_valu_issue_rate.txt"],
    "note": "launch order: tiles by depth class of the camera's previous visit (default); the first, raster-ordered visit of a camera fetches "
            "less (the min of FETCH_SIZE_KiB_min_max)",
}
try:   # the backward blend's traffic from the same passes (bench.py: roofline_other_kernels)
    bf = avg(os.path.join(d, prefix + "FETCH_SIZE.txt"), "FETCH_SIZE", "k_blend_bwd_w")
    bw = avg(os.path.join(d, prefix + "WRITE_SIZE.txt"), "WRITE_SIZE", "k_blend_bwd_w")
    out["blend_bwd_hbm_bytes_per_launch"] = (2.0 * bf[0] + bw[0]) * 1024.0
    out["blend_bwd_valu_wave_insts_per_launch"] = avg(os.path.join(d, prefix + "SQ_INSTS_VALU.txt"), "SQ_INSTS_VALU", "k_blend_bwd_w")[0]
except (SystemExit, OSError):
    pass
print(json.dumps(out, indent=1))
