"""Seeded synthetic scenes and orbit cameras for tests and bench.py.

No dataset or checkpoint exists in this environment, so every BASELINE.json config is a synthetic
stand-in (SURVEY.md section 8d / BASELINE.md section 3).  The camera algebra restates the reference:
  * getProjectionMatrix  -- sugar_utils/graphics_utils.py:65-85
  * world_view_transform = W2C^T, full_proj_transform = W2C^T @ P^T, camera_center = inverse(W2C^T)[3,:3]
    -- sugar_scene/cameras.py:209-212
All tensors are float32 on CPU; callers move them to the device.
"""
from __future__ import annotations

import math
from typing import NamedTuple

import torch

SH_C0 = 0.28209479177387814


class Scene(NamedTuple):
    means3D: torch.Tensor    # [P,3]
    scales: torch.Tensor     # [P,3]  already activated (exp)
    rotations: torch.Tensor  # [P,4]  unit quaternions, real part first
    opacities: torch.Tensor  # [P,1]  already activated (sigmoid)
    shs: torch.Tensor        # [P,16,3] SH degree 3, coefficient-major, RGB innermost


class Camera(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] = W2C^T (row-vector convention)
    projmatrix: torch.Tensor  # [4,4] = W2C^T @ P^T
    campos: torch.Tensor      # [3]


# name -> (P, W, H, seed, s_lo, s_hi, max_norm, bg)   
CONFIGS = {
    "config1": (10_000, 256, 256, 0, 0.01, 0.08, None, 0.0),
    "config2": (300_000, 800, 800, 1, 0.003, 0.02, 0.8, 1.0),
    "config3": (2_000_000, 1920, 1080, 2, 0.002, 0.03, None, 0.0),
    "config5": (6_000_000, 3840, 2160, 5, 0.001, 0.01, None, 0.0),
    "metric": (1_000_000, 1920, 1080, 7, 0.002, 0.03, None, 0.0),
}


def make_scene(P: int, seed: int, s_lo: float, s_hi: float, max_norm: float | None = None,
               sh_degree: int = 3) -> Scene:
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(P, 3, generator=g) * 2.0 - 1.0
    if max_norm is not None:
        # confine to a ball of radius max_norm by rescaling points that fall outside
        n = means.norm(dim=1, keepdim=True).clamp_min(1e-12)
        means = torch.where(n > max_norm, means * (max_norm * torch.rand(P, 1, generator=g) / n), means)
    log_s = torch.rand(P, 3, generator=g) * (math.log(s_hi) - math.log(s_lo)) + math.log(s_lo)
    scales = torch.exp(log_s)
    q = torch.randn(P, 4, generator=g)
    rotations = q / q.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.0)
    M = (sh_degree + 1) ** 2
    dc = (torch.rand(P, 1, 3, generator=g) - 0.5) / SH_C0
    rest = torch.randn(P, M - 1, 3, generator=g) * 0.1
    shs = torch.cat([dc, rest], dim=1).contiguous()
    return Scene(means.contiguous(), scales.contiguous(), rotations.contiguous(), opacities.contiguous(), shs)


def get_projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """sugar_utils/graphics_utils.py:65-85"""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
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


def look_at_camera(eye, target, W: int, H: int, fovx_deg: float = 60.0, znear: float = 0.01,
                   zfar: float = 100.0) -> Camera:
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    fwd = target - eye
    fwd = fwd / fwd.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)  # camera y points down (COLMAP convention, +z forward)
    R_w2c = torch.stack([right, down, fwd], dim=0)  # rows = camera axes in world coords
    t = -R_w2c @ eye
    W2C = torch.eye(4, dtype=torch.float64)
    W2C[:3, :3] = R_w2c
    W2C[:3, 3] = t
    W2C = W2C.float()
    fovx = math.radians(fovx_deg)
    tanfovx = math.tan(fovx / 2)
    tanfovy = tanfovx * H / W
    fovy = 2 * math.atan(tanfovy)
    viewmatrix = W2C.transpose(0, 1).contiguous()
    proj = get_projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    projmatrix = (viewmatrix.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = viewmatrix.inverse()[3, :3].contiguous()
    return Camera(H, W, tanfovx, tanfovy, viewmatrix, projmatrix, campos)


def orbit_cameras(W: int, H: int, n: int = 8, radius: float = 3.0, elev_deg: float = 15.0) -> list[Camera]:
    cams = []
    el = math.radians(elev_deg)
    for k in range(n):
        az = math.radians(k * 360.0 / n)
        eye = (radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el))
        cams.append(look_at_camera(eye, (0.0, 0.0, 0.0), W, H))
    return cams


def scattered_cameras(W: int, H: int, n: int = 200, seed: int = 0) -> list[Camera]:
    """n look-at cameras on a shell around the scene (azimuth uniform, elevation -10..50 degrees, radius 2.5..3.5): the camera count
    of a real capture (SuGaR's scenes have 100-300 training views), for bench.py's hint-robustness leg"""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, generator=g)
    cams = []
    for k in range(n):
        az = 2 * math.pi * float(u[k, 0]); el = math.radians(-10.0 + 60.0 * float(u[k, 1])); r = 2.5 + float(u[k, 2])
        cams.append(look_at_camera((r * math.cos(el) * math.cos(az), r * math.cos(el) * math.sin(az), r * math.sin(el)), (0.0, 0.0, 0.0), W, H))
    return cams


def make_config(name: str, P: int | None = None):
    """Returns (scene, cameras, bg[3]) for a BASELINE config name; P may override the Gaussian count."""
    P0, W, H, seed, s_lo, s_hi, max_norm, bg = CONFIGS[name]
    scene = make_scene(P0 if P is None else P, seed, s_lo, s_hi, max_norm)
    return scene, orbit_cameras(W, H), torch.full((3,), bg)
