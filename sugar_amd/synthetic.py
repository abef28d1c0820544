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
    # config 4 (refine mode): s_lo / s_hi are unused -- the in-plane scales come from the triangles, the third one is the
    # mesh thickness (make_bound_scene)
    "config4": (1_000_000, 1920, 1080, 4, None, None, None, 0.0),
    # the same mesh-bound scene as a FRESHLY BOUND refined model has it (round 6): opacities 0.9999 (sugar_model.py:281-283: a model
    # bound to a surface mesh starts from inverse_sigmoid(0.9999)), in-plane scales = the reference's initial value exactly
    # (:323-326: the triangles are covered), no in-plane rotation (:337-339) -- an opaque surface, where config 4 above is a
    # half-transparent, mid-refinement state whose depth render floats off the surface
    "config4_opaque": (1_000_000, 1920, 1080, 4, None, None, None, 0.0),
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


class BoundScene(NamedTuple):
    """BASELINE config 4: Gaussians bound to the triangles of a surface mesh, as SuGaR's refine-mode model derives them
    (sugar_scene/sugar_model.py:149-228 construction, :384-397 positions, :399-442 scales, :444-475 rotations)."""
    scene: Scene               # what crosses the rasterizer boundary: flat Gaussians, scales[:, 0] = thickness
    verts: torch.Tensor        # [V,3]   the model's `_points` (mesh vertices)
    faces: torch.Tensor        # [F,3]   int64
    plane_scales: torch.Tensor # [P,2]   activated in-plane scales (`scale_activation(_scales)`)
    complex_rot: torch.Tensor  # [P,2]   `_quaternions`: the learned in-plane rotation, one complex number per Gaussian
    thickness: float           # `surface_mesh_thickness` = spatial extent of the cameras / 1e6 (:165-166)
    n_per_triangle: int


# barycentric coordinates and circle radius per Gaussians-per-triangle (sugar_model.py:171-213)
_BARY = {
    1: ([[1 / 3, 1 / 3, 1 / 3]], 1.0 / 2.0 / math.sqrt(3.0)),
    3: ([[1 / 2, 1 / 4, 1 / 4], [1 / 4, 1 / 2, 1 / 4], [1 / 4, 1 / 4, 1 / 2]], 1.0 / 2.0 / (math.sqrt(3.0) + 1.0)),
    4: ([[1 / 3, 1 / 3, 1 / 3], [2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]], 1.0 / (4.0 * math.sqrt(3.0))),
    6: ([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12],
         [5 / 12, 5 / 12, 1 / 6]], 1.0 / (4.0 + 2.0 * math.sqrt(3.0))),
}


def grid_surface_mesh(n_u: int, n_v: int, seed: int, radius: float = 0.8):
    """A consistently oriented triangle mesh of 2*n_u*n_v faces: a latitude-longitude band (the polar caps are left open)
    on a sphere whose radius is modulated by a few lobes, vertices jittered inside their cells so that no two triangles are
    congruent.  Returns (verts[V,3] float32, faces[F,3] int64), V = (n_v+1)*n_u."""
    g = torch.Generator().manual_seed(seed)
    i = torch.arange(n_v + 1, dtype=torch.float64).reshape(-1, 1)
    j = torch.arange(n_u, dtype=torch.float64).reshape(1, -1)
    jit = torch.rand(2, n_v + 1, n_u, generator=g, dtype=torch.float64) - 0.5
    th = math.pi * (0.04 + 0.92 * (i + 0.5 * jit[0]) / n_v)
    ph = 2.0 * math.pi * (j + 0.5 * jit[1]) / n_u
    r = radius * (1.0 + 0.12 * torch.sin(3.0 * ph) * torch.sin(4.0 * th) + 0.06 * torch.cos(5.0 * th + ph))
    verts = torch.stack([r * torch.sin(th) * torch.cos(ph), r * torch.sin(th) * torch.sin(ph), r * torch.cos(th)], dim=-1)
    verts = verts.reshape(-1, 3).float().contiguous()
    ii = torch.arange(n_v).reshape(-1, 1)
    jj = torch.arange(n_u).reshape(1, -1)
    a = ii * n_u + jj
    b = ii * n_u + (jj + 1) % n_u
    c = (ii + 1) * n_u + jj
    d = (ii + 1) * n_u + (jj + 1) % n_u
    faces = torch.stack([torch.stack([a, c, d], dim=-1), torch.stack([a, d, b], dim=-1)], dim=2).reshape(-1, 3)
    return verts, faces.long().contiguous()


def _matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """[N,3,3] rotation matrices -> unit quaternions, real part first (the branch with the largest component, as
    pytorch3d.transforms.matrix_to_quaternion chooses it; q and -q are the same rotation)"""
    m = R.reshape(-1, 9).unbind(-1)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m
    q_abs = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
                        dim=-1).clamp_min(0.0).sqrt()
    cand = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(dim=-1)
    q = cand[torch.arange(cand.shape[0]), best]
    return q / q.norm(dim=-1, keepdim=True)


def bound_gaussians(verts, faces, plane_scales, complex_rot, thickness: float, n_per_triangle: int):
    """The Gaussians SuGaR's surface-bound model hands to the rasterizer, from the mesh and the model's in-plane parameters:
      positions  barycentric combinations of the triangle's vertices                     sugar_model.py:384-397
      scales     [thickness, plane_scales]  (the FIRST axis is the thin one)              sugar_model.py:438-442
      rotations  columns (face normal, R_1, R_2): R_1, R_2 = the triangle's first edge and its in-plane normal, turned by
                 the unit complex number                                                   sugar_model.py:449-475
    Returns (means3D[P,3], scales[P,3], quaternions[P,4])."""
    bary = torch.tensor(_BARY[n_per_triangle][0], dtype=torch.float32)           # [n,3]
    fv = verts[faces]                                                             # [F,3,3]
    means = (fv[:, None] * bary[None, :, :, None]).sum(dim=-2).reshape(-1, 3)
    n = n_per_triangle
    nrm = torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)
    R0 = torch.nn.functional.normalize(nrm / nrm.norm(dim=-1, keepdim=True).clamp_min(1e-6), dim=-1)
    b1 = torch.nn.functional.normalize(fv[:, 0] - fv[:, 1], dim=-1)
    b2 = torch.nn.functional.normalize(torch.linalg.cross(R0, b1, dim=-1), dim=-1)
    z = torch.nn.functional.normalize(complex_rot, dim=-1).view(len(faces), n, 2)
    R1 = z[..., 0:1] * b1[:, None] + z[..., 1:2] * b2[:, None]
    R2 = -z[..., 1:2] * b1[:, None] + z[..., 0:1] * b2[:, None]
    R = torch.cat([R0[:, None, :, None].expand(-1, n, -1, -1), R1[..., None], R2[..., None]], dim=-1).reshape(-1, 3, 3)
    quats = _matrix_to_quaternion(R)
    scales = torch.cat([torch.full((means.shape[0], 1), float(thickness)), plane_scales], dim=-1)
    return means.contiguous(), scales.contiguous(), quats.contiguous()


def make_bound_scene(P: int, seed: int, n_per_triangle: int = 1, extent: float = 3.3, sh_degree: int = 3, opaque: bool = False) -> BoundScene:
    """~P flat Gaussians on a grid surface mesh, in a mid-refinement state: in-plane scales = the reference's initial value
    (shortest edge x the inscribed-circle factor, sugar_model.py:323-326) times a log-uniform factor per axis in [0.5, 2],
    a uniformly random in-plane rotation, opacities and SH as in the other configs.  The Gaussian count is 2*n_u*n_v*n."""
    n_faces = max(P // n_per_triangle, 8)
    n_v = max(2, int(round(math.sqrt(n_faces / 4.0))))
    n_u = max(3, n_faces // (2 * n_v))
    verts, faces = grid_surface_mesh(n_u, n_v, seed)
    F_, n = faces.shape[0], n_per_triangle
    Pn = F_ * n
    g = torch.Generator().manual_seed(seed + 1000)
    fv = verts[faces]
    s0 = (fv - fv[:, [1, 2, 0]]).norm(dim=-1).min(dim=-1)[0] * _BARY[n][1]
    s0 = s0.clamp_min(1e-7).reshape(F_, 1, 1).expand(-1, n, 2).reshape(-1, 2)
    plane = s0 * torch.exp((torch.rand(Pn, 2, generator=g) * 2.0 - 1.0) * math.log(2.0))
    z = torch.randn(Pn, 2, generator=g)
    z = z / z.norm(dim=-1, keepdim=True)
    thickness = extent / 1_000_000.0
    opacities = torch.sigmoid(torch.randn(Pn, 1, generator=g) * 2.0)
    if opaque:   # a freshly bound model (sugar_model.py:281-283, 323-339); the draws above keep the generator's state aligned
        plane = s0.clone()
        z = torch.zeros(Pn, 2); z[:, 0] = 1.0
        opacities = torch.full((Pn, 1), 0.9999)
    means, scales, quats = bound_gaussians(verts, faces, plane, z, thickness, n)
    M = (sh_degree + 1) ** 2
    dc = (torch.rand(Pn, 1, 3, generator=g) - 0.5) / SH_C0
    rest = torch.randn(Pn, M - 1, 3, generator=g) * 0.1
    shs = torch.cat([dc, rest], dim=1).contiguous()
    return BoundScene(Scene(means, scales, quats, opacities.contiguous(), shs), verts, faces, plane.contiguous(),
                      z.contiguous(), thickness, n)


# ---- BASELINE config 3's inputs at the rasterizer boundary (the coarse-SDF step makes TWO rasterizer calls per iteration)
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def sh_to_rgb(shs: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor) -> torch.Tensor:
    """`SuGaR.get_points_rgb` (sugar_model.py:839-883) in plain torch, SH degree 3: eval_sh on the normalised directions
    camera -> point (sugar_utils/spherical_harmonics.py:117-172), + 0.5, clamped at 0.  This is what the coarse trainers hand to
    the rasterizer as `colors_precomp` (coarse_sdf.py:51, sugar_model.py:2187-2200).  Works on any device."""
    d = torch.nn.functional.normalize(means3D - campos.reshape(1, 3), dim=-1)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = shs
    res = SH_C0 * sh[:, 0]
    res = res - _SH_C1 * y * sh[:, 1] + _SH_C1 * z * sh[:, 2] - _SH_C1 * x * sh[:, 3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    res = (res + _SH_C2[0] * xy * sh[:, 4] + _SH_C2[1] * yz * sh[:, 5] + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
           + _SH_C2[3] * xz * sh[:, 7] + _SH_C2[4] * (xx - yy) * sh[:, 8])
    res = (res + _SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + _SH_C3[1] * xy * z * sh[:, 10]
           + _SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
           + _SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + _SH_C3[5] * z * (xx - yy) * sh[:, 14]
           + _SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return (res + 0.5).clamp_min(0.0).contiguous()


def depth_as_colour(means3D: torch.Tensor, viewmatrix: torch.Tensor):
    """coarse_sdf.py:575-590: the view-space depth of every Gaussian centre, expanded to three channels, as the colour of a
    depth render, and its maximum as the background.  `viewmatrix` is the row-vector matrix the rasterizer takes (W2C^T).
    Returns (point_depth[P,3], bg[3])."""
    z = means3D @ viewmatrix[:3, 2:3] + viewmatrix[3, 2]
    point_depth = z.expand(-1, 3).contiguous()
    return point_depth, (z.max() + torch.zeros(3, dtype=means3D.dtype, device=means3D.device)).contiguous()


def make_config(name: str, P: int | None = None):
    """Returns (scene, cameras, bg[3]) for a BASELINE config name; P may override the Gaussian count."""
    P0, W, H, seed, s_lo, s_hi, max_norm, bg = CONFIGS[name]
    if name in ("config4", "config4_opaque"):
        scene = make_bound_scene(P0 if P is None else P, seed, opaque=name == "config4_opaque").scene
    else:
        scene = make_scene(P0 if P is None else P, seed, s_lo, s_hi, max_norm)
    return scene, orbit_cameras(W, H), torch.full((3,), bg)
