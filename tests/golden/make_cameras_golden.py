#!/usr/bin/env python
"""Golden camera matrices from the reference's own loader: a cameras.json (written here with the reference's
camera_to_JSON, gaussian_splatting/utils/camera_utils.py:62-82) is read by sugar_scene/cameras.py:load_gs_cameras, and the
world_view_transform / full_proj_transform / camera_center of the resulting GSCamera objects are stored next to the JSON
text.  Run in the build container (needs /root/reference); `Tensor.cuda()` is patched to the identity (GSCamera hard-codes
it, sugar_scene/cameras.py:209-210) and the stand-in pytorch3d package satisfies the module's imports.

    python tests/golden/make_cameras_golden.py   -> tests/golden/cameras_golden.npz
"""
import json
import math
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    for p in (os.path.join(REF, "gaussian_splatting"), REF, ROOT):
        sys.path.insert(0, p)
    from sugar_amd import shims
    shims.install()
    for name in ("open3d", "plyfile"):  # imported by modules on the way, never used here
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.PlyData = m.PlyElement = object
                sys.modules[name] = m
    import sugar_scene.cameras as rc
    import scene  # noqa: F401  (the reference's gaussian_splatting/scene: resolves its circular import with utils first)
    from utils.camera_utils import camera_to_JSON  # the reference's gaussian_splatting/utils
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        rng = np.random.default_rng(5)
        entries = []
        for i in range(6):
            q = rng.standard_normal(4); q /= np.linalg.norm(q)
            r, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                          [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                          [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
            T = rng.standard_normal(3) * 2
            W, H = (1600, 1063) if i % 2 else (3000, 2000)   # the second size is capped at max_img_size=1920
            cam = types.SimpleNamespace(R=R, T=T, image_name=f"img_{(7 * i) % 6:03d}", width=W, height=H,
                                        FovY=math.radians(38 + i), FovX=math.radians(55 + 2 * i))
            entries.append(camera_to_JSON(i, cam))
        text = json.dumps(entries)
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "src", "images"))
            open(os.path.join(d, "src", "images", "img_000.png"), "wb").close()
            with open(os.path.join(d, "cameras.json"), "w") as f:
                f.write(text)
            cams = rc.load_gs_cameras(os.path.join(d, "src"), d + os.sep, load_gt_images=False)
        out = {"json": np.frombuffer(text.encode(), dtype=np.uint8)}
        out["names"] = np.array([c.image_name for c in cams])
        out["sizes"] = np.array([[c.image_height, c.image_width] for c in cams], dtype=np.int64)
        out["fov"] = np.array([[c.FoVx, c.FoVy] for c in cams], dtype=np.float64)
        out["world_view"] = np.stack([c.world_view_transform.numpy() for c in cams])
        out["full_proj"] = np.stack([c.full_proj_transform.numpy() for c in cams])
        out["center"] = np.stack([c.camera_center.numpy() for c in cams])
    finally:
        torch.Tensor.cuda = real_cuda
    path = os.path.join(HERE, "cameras_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
