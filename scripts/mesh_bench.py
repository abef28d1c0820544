"""Timing of the HIP mesh z-buffer at BASELINE size (splat mesh of P Gaussians at 1920x1080, faces_per_pixel 10 and 1)."""
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import mesh_scenes as ms  # noqa: E402
from sugar_amd.mesh_raster import rasterize_face_verts  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H, W = 1080, 1920
t = torch.as_tensor(ms.splat_like(P, 21, W, H), device="cuda:0")
out = {"P": P, "faces": 2 * P}
for K, attrs in ((10, True), (10, False), (1, False)):
    for _ in range(3):
        r = rasterize_face_verts(t, [0], [2 * P], (H, W), 0.0, K, True, want_bary=attrs, want_dists=attrs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        r = rasterize_face_verts(t, [0], [2 * P], (H, W), 0.0, K, True, want_bary=attrs, want_dists=attrs)
    torch.cuda.synchronize()
    out[f"ms_K{K}_attrs{int(attrs)}"] = (time.perf_counter() - t0) / n * 1e3
    out["covered"] = float((r[0][0, ..., 0] >= 0).float().mean())
print(json.dumps(out))
