"""debug: lane utilisation of the forward blend (needs SGR_BLEND_DEFS=-DSGR_COUNT)"""
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
    pu.run_hip(scene, cams[0], bg)
    torch.cuda.synchronize()
    lib.sgr_debug_counts(buf)
    it, ok, rows, blocks = buf[0], buf[1], buf[2], buf[3]
    print(name, "wave-iterations", it, "ok lanes/iter", ok / it, "8x2 row pairs/iter", rows / it, "4x4 blocks/iter", blocks / it, flush=True)
