"""Developer diagnostic (GPU box): how much of the tile lists does the metric workload walk, per tile and per 8x8-tile
super-tile (the unit whose 512-entry chunks the hinted write pass can skip)?  python scripts/hint_stats.py [config]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, _lib
from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
from tests import parity_utils as pu

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
scene, cams, bg = syn.make_config(name)
cam = cams[0]
W, H = cam.image_width, cam.image_height
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
lib = _lib.load()
dev = torch.device("cuda:0")
hint = torch.zeros(T, dtype=torch.int32, device=dev)
with grad_sink(tile_need_out=hint):
    hp = pu.run_hip(scene, cam, bg)
off = lib.sgr_img_tile_walked_offset(W, H)
walked = _C.last_forward["img"][off: off + 4 * T].view(torch.int32).cpu().numpy().astype(np.int64)
cnt = np.diff(hp["tile_start"].astype(np.int64))
need = np.minimum(hint.cpu().numpy().astype(np.int64), cnt)
print(f"{name}: R = {cnt.sum()}, walked = {walked.sum()} ({walked.sum() / cnt.sum():.3f}), hinted = {need.sum()} ({need.sum() / cnt.sum():.3f})")
frac = np.where(cnt > 0, need / np.maximum(cnt, 1), 0.0)
print("per-tile need/count quantiles (tiles with entries):", np.quantile(frac[cnt > 0], [0.1, 0.5, 0.9, 0.99]).round(3))
sgx, sgy = (gx + 7) // 8, (gy + 7) // 8
f2 = np.zeros((sgy * 8, sgx * 8)); f2[:gy, :gx] = frac.reshape(gy, gx)
c2 = np.zeros((sgy * 8, sgx * 8)); c2[:gy, :gx] = cnt.reshape(gy, gx)
fs = f2.reshape(sgy, 8, sgx, 8).max(axis=(1, 3))
cs = c2.reshape(sgy, 8, sgx, 8).sum(axis=(1, 3))
print("per-super-tile max need/count:", np.quantile(fs[cs > 0], [0.1, 0.5, 0.9]).round(3), " instance-weighted mean:", (fs * cs).sum() / cs.sum())
full = (f2.reshape(sgy, 8, sgx, 8) >= 0.999).sum(axis=(1, 3))
print("tiles walking their whole list: %d of %d; super-tiles containing one: %d of %d" % ((frac >= 0.999).sum(), (cnt > 0).sum(), (full > 0).sum(), (cs > 0).sum()))
nc = hp["n_contrib"].reshape(H, W)
ft = hp["final_T"].reshape(H, W)
print("pixels with final_T > 1e-4 (never saturated): %.3f" % (ft > 1e-4).mean())
