"""Developer diagnostic (GPU box): time of the list-write pass (stage 2) without a walk hint, with the real hint and with
a hint of one entry per tile (which lets the pass skip nearly every chunk)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, _lib
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink

scene, cams, bg = syn.make_config(sys.argv[1] if len(sys.argv) > 1 else "metric")
cam = cams[0]
W, H = cam.image_width, cam.image_height
T = ((W + 15) // 16) * ((H + 15) // 16)
dev = torch.device("cuda:0")
lib = _lib.load()
st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3,
                                   cam.campos.to(dev), False, False)
a = {k: getattr(scene, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros_like(a["means3D"])
hint = torch.zeros(T, dtype=torch.int32, device=dev)


def run(name, **sink):
    ms = (C.c_double * 7)(); cnt = (C.c_int64 * 7)()
    for it in range(6):
        if it == 1:
            lib.sgr_profile_enable(127)
        with torch.no_grad(), grad_sink(**sink):
            GaussianRasterizer(st)(a["means3D"], m2, a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    torch.cuda.synchronize()
    lib.sgr_profile_enable(0)
    lib.sgr_profile_read(ms, cnt, 7)
    print(f"{name:12s} count+scan {1e3 * ms[1] / cnt[1]:7.1f} us   write {1e3 * ms[2] / cnt[2]:7.1f} us   blend fwd {1e3 * ms[4] / cnt[4]:7.1f} us")


run("no hint", tile_need_out=hint)
run("real hint", tile_need=hint)
run("hint = 1", tile_need=torch.ones_like(hint))
