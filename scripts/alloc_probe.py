"""Why is the reference-shaped forward slow at config 5 when it keeps the host round trip?  Per call: wall time, device allocations
/ frees made by the caching allocator, reserved bytes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sugar_amd import synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "config5"
scene, cams, bg = syn.make_config(name)
t = {k: getattr(scene, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros_like(t["means3D"])
bg = bg.to(dev)
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
rows = []
for it in range(48):
    cam = cams[it % len(cams)]
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.viewmatrix, cam.projmatrix, 3, cam.campos, False, False)
    ms0 = torch.cuda.memory_stats(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        GaussianRasterizer(st)(t["means3D"], m2, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ms1 = torch.cuda.memory_stats(dev)
    rows.append(dict(it=it, call_ms=round(1e3 * (t1 - t0), 3), total_ms=round(1e3 * (t2 - t0), 3), R=_C.last_forward["num_rendered"],
                     dev_alloc=ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0), dev_free=ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
                     reserved_GB=round(ms1.get("reserved_bytes.all.current", 0) / 2**30, 2), sizes={k: v.numel() for k, v in _C.last_forward.items() if torch.is_tensor(v) and v.dtype == torch.uint8}))
for r in rows:
    print(json.dumps(r))
