"""On-disk formats either side of the hot path, read and written without `plyfile`:

  * the 3DGS point-cloud PLY (gaussian_splatting/scene/gaussian_model.py:177-256, save_ply / load_ply): binary little
    endian, one `vertex` element with float32 properties
        x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*(M-1)-1)  opacity  scale_0..2  rot_0..3
    raw (pre-activation) values; f_rest is stored CHANNEL-major (`features_rest.transpose(1, 2).flatten(1)`: all R
    coefficients, then G, then B), the rasterizer wants [P, M, 3].
  * `cameras.json` (gaussian_splatting/utils/camera_utils.py:62-82 writes it; sugar_scene/cameras.py:15-139 reads it):
    per camera id, img_name, width, height, position (camera centre), rotation (camera-to-world), fx, fy.

`cameras_from_json` builds the rasterizer's matrices (world_view_transform, full_proj_transform, camera_center of
sugar_scene/cameras.py:203-212) once, as float32 tensors on the requested device -- the reference re-derives them per
render call through `.cpu().numpy()` and `np.linalg.inv` (sugar_scene/sugar_model.py:2131-2150).
"""
from __future__ import annotations

import json
import math
import os

import numpy as np
import torch

from .synthetic import Camera

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def gaussian_ply_attributes(M: int):
    """construct_list_of_attributes, gaussian_model.py:177-189"""
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * (M - 1))]
    return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_gaussian_ply(path, xyz, features, opacity, scaling, rotation):
    """save_ply, gaussian_model.py:191-208.  `features` is [P, M, 3] (DC coefficient first); raw parameter values."""
    to_np = lambda t: t.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(t) else np.asarray(t, np.float32)
    xyz, features, opacity, scaling, rotation = map(to_np, (xyz, features, opacity, scaling, rotation))
    P, M = features.shape[0], features.shape[1]
    f_dc = features[:, 0, :]                                         # [P,3]: f_dc_c
    f_rest = features[:, 1:, :].transpose(0, 2, 1).reshape(P, -1)    # channel-major, as transpose(1,2).flatten(1)
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(P, 1), scaling, rotation], axis=1)
    names = gaussian_ply_attributes(M)
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())


def _read_ply_vertices(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: an element precedes 'vertex'; not a 3DGS point cloud")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: only binary_little_endian PLY is supported (got {fmt})")
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        dt = np.dtype(props)
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
    return data


def load_gaussian_ply(path, device="cpu"):
    """load_ply, gaussian_model.py:215-256 -> dict(xyz[P,3], features[P,M,3], opacity[P,1], scaling[P,3], rotation[P,4]),
    raw parameter values as float32 tensors on `device`; properties are looked up by name, any order."""
    v = _read_ply_vertices(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], dtype=np.float32)
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    rest = sorted([n for n in names if n.startswith("f_rest_")], key=lambda n: int(n.split("_")[-1]))
    if len(rest) % 3:
        raise ValueError(f"{path}: {len(rest)} f_rest_* properties (not a multiple of 3)")
    M = len(rest) // 3 + 1
    if M not in (1, 4, 9, 16):
        raise ValueError(f"{path}: {M} SH coefficients per channel")
    feats = np.zeros((xyz.shape[0], M, 3), dtype=np.float32)
    for c in range(3):
        feats[:, 0, c] = col(f"f_dc_{c}")
        for k in range(M - 1):
            feats[:, 1 + k, c] = col(rest[c * (M - 1) + k])   # reshape((P, 3, M-1)) of gaussian_model.py:236
    scales = np.stack([col(n) for n in sorted([n for n in names if n.startswith("scale_")], key=lambda n: int(n.split("_")[-1]))], axis=1)
    rots = np.stack([col(n) for n in sorted([n for n in names if n.startswith("rot")], key=lambda n: int(n.split("_")[-1]))], axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return dict(xyz=t(xyz), features=t(feats), opacity=t(col("opacity")[:, None]), scaling=t(scales), rotation=t(rots))


# ---------------------------------------------------------------- cameras
def focal2fov(focal, pixels):
    """sugar_utils/graphics_utils.py:90-91"""
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    """sugar_utils/graphics_utils.py:87-88"""
    return pixels / (2 * math.tan(fov / 2))


def projection_matrix(znear, zfar, fovX, fovY):
    """getProjectionMatrix, sugar_utils/graphics_utils.py:65-85"""
    tanHalfFovY, tanHalfFovX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanHalfFovY * znear, tanHalfFovX * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_RT(R, T, fov_x, fov_y, width, height, znear=0.01, zfar=100.0, device="cpu") -> Camera:
    """GSCamera's matrices (sugar_scene/cameras.py:203-212) from the 3DGS convention: R = camera-to-world rotation (stored
    transposed, 'glm'), T = world-to-camera translation."""
    Rt = np.zeros((4, 4), dtype=np.float64)             # getWorld2View2, sugar_utils/graphics_utils.py:51-63
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(T, dtype=np.float64)
    Rt[3, 3] = 1.0
    world_view = torch.tensor(np.float32(Rt)).transpose(0, 1)
    proj = projection_matrix(znear, zfar, fov_x, fov_y).transpose(0, 1)
    full = (world_view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = world_view.inverse()[3, :3]
    return Camera(image_height=int(height), image_width=int(width), tanfovx=math.tan(fov_x * 0.5), tanfovy=math.tan(fov_y * 0.5),
                  viewmatrix=world_view.contiguous().to(device), projmatrix=full.contiguous().to(device),
                  campos=center.contiguous().to(device))


def cameras_from_json(path, device="cpu", image_resolution=1, max_img_size=1920, remove_indices=()):
    """load_gs_cameras(load_gt_images=False), sugar_scene/cameras.py:15-139: duplicates of an image name keep the last
    entry, cameras are sorted by image name, sizes are scaled by `image_resolution` and capped at `max_img_size`.
    Returns (cameras, names)."""
    with open(path) as f:
        entries = json.load(f)
    entries = [e for i, e in enumerate(entries) if i not in set(remove_indices)]
    by_name = {}
    for e in entries:
        by_name[e["img_name"]] = e
    if len(by_name) != len(entries):
        entries = list(by_name.values())
    entries = sorted(entries, key=lambda e: e["img_name"])
    cams, names = [], []
    for e in entries:
        W2C = np.zeros((4, 4))
        W2C[:3, :3] = np.array(e["rotation"]); W2C[:3, 3] = np.array(e["position"]); W2C[3, 3] = 1
        Rt = np.linalg.inv(W2C)
        T, R = Rt[:3, 3], Rt[:3, :3].transpose()
        width, height = e["width"], e["height"]
        fov_y, fov_x = focal2fov(e["fy"], height), focal2fov(e["fx"], width)
        downscale = image_resolution if image_resolution in (1, 2, 4, 8) else 1
        if max(height, width) > max_img_size:
            downscale = (max(height, width) / max_img_size) * downscale
        h, w = round(height / downscale), round(width / downscale)
        cams.append(camera_from_RT(R, T, fov_x, fov_y, w, h, device=device))
        names.append(e["img_name"])
    return cams, names


def camera_to_json(cam_id, name, R, T, fov_x, fov_y, width, height):
    """camera_to_JSON, gaussian_splatting/utils/camera_utils.py:62-82"""
    Rt = np.zeros((4, 4)); Rt[:3, :3] = np.asarray(R).transpose(); Rt[:3, 3] = np.asarray(T); Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    return {"id": cam_id, "img_name": name, "width": width, "height": height, "position": W2C[:3, 3].tolist(),
            "rotation": [x.tolist() for x in W2C[:3, :3]], "fy": fov2focal(fov_y, height), "fx": fov2focal(fov_x, width)}
