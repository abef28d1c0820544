"""debug: lane utilisation of the blend kernels (needs SGR_BLEND_DEFS="-DSGR_COUNT -DSGR_BWD_REF_A")"""
import numpy as np
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import build, synthetic as syn
build.build()
from tests import parity_utils as pu
lib = C.CDLL(os.path.join(os.path.dirname(build.__file__), "libsugar_raster.so"))
for name in ("metric", "config2"):
    scene, cams, bg = syn.make_config(name)
    buf = (C.c_ulonglong * 8)()
    lib.sgr_debug_counts(buf)
    cam = cams[0]
    g = np.random.default_rng(0).standard_normal((3, cam.image_height, cam.image_width)).astype(np.float32)
    pu.run_hip(scene, cam, bg, grad_out=g)
    torch.cuda.synchronize()
    lib.sgr_debug_counts(buf)
    print(name, "backward: (entry, block) pairs", buf[4], "with a contributing pixel", buf[5], f"({buf[5] / max(buf[4], 1):.3f})",
          "contributing lanes per pair", buf[6] / max(buf[4], 1), flush=True)
