"""Round-4 re-measurement (verdict item 6b): what do the forward's latency-bound stages cost WHILE the SH half of Adam streams
through HBM on a second stream?  Stage events (sgr_profile_*) of a sync-free forward on stream B, alone and with
k_sh_adam_from_views running back to back on stream A (optionally with a higher stream priority for B)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sugar_amd import _lib, synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink
from sugar_amd.train_step import GaussianParams, FlatAdam

STAGES = ["preprocess", "bin_count_scan", "bin_scatter", "depth_sort", "blend_fwd", "blend_bwd", "preprocess_bwd", "hint_repair"]
lib = _lib.load()
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("metric")
P = scene.means3D.shape[0]
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
t = {k: getattr(scene, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros_like(t["means3D"])
bg = bg.to(dev)
params = GaussianParams(scene, dev)
opt = FlatAdam(params)
dcol = torch.randn(1, P + 1, 3, device=dev) * 1e-3
campos = cams[0].campos.reshape(1, 3).contiguous()
hdr, ev = torch.zeros(16, dtype=torch.int32).pin_memory(), torch.cuda.Event()


def forward(cam, cap):
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.viewmatrix, cam.projmatrix, 3, cam.campos, False, False)
    with torch.no_grad(), grad_sink(binning_capacity=cap, header_out=hdr, header_event=ev):
        GaussianRasterizer(st)(t["means3D"], m2, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])


def stage_ms():
    ms = (C.c_double * len(STAGES))(); cnt = (C.c_int64 * len(STAGES))()
    lib.sgr_profile_read(ms, cnt, len(STAGES))
    return {n: (ms[i] / cnt[i] if cnt[i] else 0.0) for i, n in enumerate(STAGES)}


def run(contend, prio):
    sB = torch.cuda.Stream(dev, priority=-1 if prio else 0)
    sA = torch.cuda.Stream(dev)
    cap = 24 * P
    with torch.cuda.stream(sB):
        for i in range(8): forward(cams[i % 8], cap)
    torch.cuda.synchronize()
    lib.sgr_profile_enable((1 << 5) - 1)
    stop = torch.zeros(1)
    t0 = time.perf_counter()
    for i in range(16):
        if contend:
            with torch.cuda.stream(sA):
                for _ in range(3):  # keep stream A busy for the whole forward (3 x ~190 us)
                    opt.begin_step(); opt.step_sh((params.params["xyz"], campos, dcol, 3))
        with torch.cuda.stream(sB):
            forward(cams[i % 8], cap)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / 16
    lib.sgr_profile_enable(0)
    r = stage_ms()
    r["wall_ms_per_iteration"] = wall
    return r


out = {"alone": run(False, False), "with_sh_adam_on_another_stream": run(True, False), "same_with_high_priority_forward_stream": run(True, True)}
# the SH-Adam kernel alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    opt.begin_step(); opt.step_sh((params.params["xyz"], campos, dcol, 3))
torch.cuda.synchronize()
out["sh_adam_alone_ms"] = 1e3 * (time.perf_counter() - t0) / 20
print(json.dumps(out, indent=1))
