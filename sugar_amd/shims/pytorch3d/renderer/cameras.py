"""Camera algebra of `pytorch3d.renderer.cameras` that SuGaR uses: `FoVPerspectiveCameras` with
`R`, `T`, `K`, `znear`, `zfar`, indexing, `get_world_to_view_transform`, `get_projection_transform`,
`get_full_projection_transform`, `get_camera_center`, `transform_points`, `unproject_points`, and
`_get_sfm_calibration_matrix` (call sites: sugar_scene/cameras.py:8-9,226-326,540; sugar_scene/sugar_model.py:272-273,
699-715,829,1902,1959,1971,2129,2153-2164; sugar_trainers/coarse_sdf.py:575-616,646-649).

No kernels here: 4x4 matrices in pytorch3d's row-vector convention (`X_view = X_world @ R + T`, `p' = [p 1] @ M`).
Restated from the public pytorch3d 0.7.4 API (the version the reference pins, environment.yml:161) -- pytorch3d is not
installed in this environment, so this module is PARITY-UNPINNED against pytorch3d itself; what the tests pin is its
consistency with the reference's own Gaussian-splatting camera (sugar_scene/cameras.py:convert_camera_from_gs_to_pytorch3d
projects to the same pixels as the rasterizer's viewmatrix/projmatrix) and project/unproject round trips.
"""
from __future__ import annotations

import math

import torch


class Transform3d:
    """A batch of 4x4 matrices applied to row vectors: points_h @ matrix (pytorch3d.transforms.Transform3d, reduced to
    what the camera code needs)."""

    def __init__(self, matrix: torch.Tensor):
        if matrix.dim() == 2:
            matrix = matrix[None]
        self._matrix = matrix

    def get_matrix(self) -> torch.Tensor:
        return self._matrix

    def inverse(self) -> "Transform3d":
        # a handful of 4x4 matrices: inverted on the host in float64 (torch.inverse on a ROCm device goes through a solver
        # library whose launch costs tens of milliseconds per call -- 240 ms per view in the level-set sampler)
        m = self._matrix
        if m.is_cuda and not m.requires_grad:
            return Transform3d(torch.inverse(m.detach().double().cpu()).to(device=m.device, dtype=m.dtype))
        return Transform3d(torch.inverse(m))

    def compose(self, *others: "Transform3d") -> "Transform3d":
        m = self._matrix
        for o in others:
            m = m @ o._matrix  # self first, then the others
        return Transform3d(m)

    def transform_points(self, points: torch.Tensor, eps=None) -> torch.Tensor:
        pts = points if points.dim() == 3 else points[None]
        if pts.dim() != 3 or pts.shape[-1] != 3:
            raise ValueError(f"Expected points to have dim = 2 or dim = 3: got shape {tuple(points.shape)}")
        ones = torch.ones(*pts.shape[:2], 1, dtype=pts.dtype, device=pts.device)
        out = torch.cat([pts, ones], dim=2) @ self._matrix  # [N,P,4] (N = 1 broadcasts)
        denom = out[..., 3:]
        if eps is not None:
            sign = denom.sign() + (denom == 0.0).type_as(denom)
            denom = sign * torch.clamp(denom.abs(), eps)
        out = out[..., :3] / denom
        if out.shape[0] == 1 and points.dim() == 2:
            out = out.reshape(points.shape)
        return out


def _get_sfm_calibration_matrix(N, device, focal_length, principal_point, orthographic: bool = False) -> torch.Tensor:
    """K[N,4,4] = [[fx 0 px 0] [0 fy py 0] [0 0 0 1] [0 0 1 0]] (perspective) -- pytorch3d.renderer.cameras"""
    focal_length = torch.as_tensor(focal_length, dtype=torch.float32, device=device)
    principal_point = torch.as_tensor(principal_point, dtype=torch.float32, device=device)
    if focal_length.dim() in (0, 1) or focal_length.shape[-1] == 1:
        fx = fy = focal_length.reshape(-1)
    else:
        fx, fy = focal_length.unbind(1)
    px, py = principal_point.reshape(-1, 2).unbind(1)
    K = fx.new_zeros(N, 4, 4)
    K[:, 0, 0] = fx
    K[:, 1, 1] = fy
    if orthographic:
        K[:, 0, 3] = px
        K[:, 1, 3] = py
        K[:, 2, 2] = 1.0
        K[:, 3, 3] = 1.0
    else:
        K[:, 0, 2] = px
        K[:, 1, 2] = py
        K[:, 3, 2] = 1.0
        K[:, 2, 3] = 1.0
    return K


def _batched(v, N, device, shape=()):
    t = torch.as_tensor(v, dtype=torch.float32, device=device)
    if t.dim() == len(shape):
        t = t[None]
    return t.expand(N, *t.shape[1:]).clone() if t.shape[0] == 1 and N > 1 else t


class FoVPerspectiveCameras:
    def __init__(self, znear=1.0, zfar=100.0, aspect_ratio=1.0, fov=60.0, degrees: bool = True, R=None, T=None, K=None,
                 device="cpu"):
        device = torch.device(device)
        R = torch.eye(3)[None] if R is None else R
        T = torch.zeros(1, 3) if T is None else T
        R = torch.as_tensor(R, dtype=torch.float32).to(device)
        T = torch.as_tensor(T, dtype=torch.float32).to(device)
        if R.dim() == 2:
            R = R[None]
        if T.dim() == 1:
            T = T[None]
        N = max(R.shape[0], T.shape[0], 1 if K is None else (K.shape[0] if K.dim() == 3 else 1))
        self.device = device
        self.R = _batched(R, N, device, (3, 3))
        self.T = _batched(T, N, device, (3,))
        self.K = None if K is None else _batched(K.to(device), N, device, (4, 4))
        self.znear = _batched(znear, N, device)
        self.zfar = _batched(zfar, N, device)
        self.aspect_ratio = _batched(aspect_ratio, N, device)
        self.fov = _batched(fov, N, device)
        self.degrees = degrees

    # ---- container behaviour
    def __len__(self):
        return self.R.shape[0]

    def __getitem__(self, index):
        if isinstance(index, int):
            index = [index]
        elif isinstance(index, torch.Tensor):
            index = index.reshape(-1).tolist() if index.dtype != torch.bool else index
        sel = lambda t: None if t is None else t[index]
        out = FoVPerspectiveCameras.__new__(FoVPerspectiveCameras)
        out.device, out.degrees = self.device, self.degrees
        for k in ("R", "T", "K", "znear", "zfar", "aspect_ratio", "fov"):
            setattr(out, k, sel(getattr(self, k)))
        return out

    def to(self, device):
        device = torch.device(device)
        out = FoVPerspectiveCameras.__new__(FoVPerspectiveCameras)
        out.device, out.degrees = device, self.degrees
        for k in ("R", "T", "K", "znear", "zfar", "aspect_ratio", "fov"):
            v = getattr(self, k)
            setattr(out, k, None if v is None else v.to(device))
        return out

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def clone(self):
        return self.to(self.device)

    # ---- what MeshRasterizer asks a camera (pytorch3d CamerasBase)
    def is_perspective(self):
        return True

    def in_ndc(self):
        return True

    def get_znear(self):
        return self.znear

    # ---- transforms
    def get_world_to_view_transform(self, **kwargs) -> Transform3d:
        R, T = kwargs.get("R", self.R), kwargs.get("T", self.T)
        N = R.shape[0]
        M = torch.zeros(N, 4, 4, dtype=R.dtype, device=R.device)
        M[:, :3, :3] = R
        M[:, 3, :3] = T
        M[:, 3, 3] = 1.0
        return Transform3d(M)

    def get_camera_center(self, **kwargs) -> torch.Tensor:
        return self.get_world_to_view_transform(**kwargs).inverse().get_matrix()[:, 3, :3]

    def compute_projection_matrix(self, znear, zfar, fov, aspect_ratio, degrees) -> torch.Tensor:
        N = len(self)
        K = torch.zeros(N, 4, 4, dtype=torch.float32, device=self.device)
        if degrees:
            fov = (math.pi / 180) * fov
        tan_half = torch.tan(fov / 2)
        max_y = tan_half * znear
        min_y = -max_y
        max_x = max_y * aspect_ratio
        min_x = -max_x
        z_sign = 1.0
        K[:, 0, 0] = 2.0 * znear / (max_x - min_x)
        K[:, 1, 1] = 2.0 * znear / (max_y - min_y)
        K[:, 0, 2] = (max_x + min_x) / (max_x - min_x)
        K[:, 1, 2] = (max_y + min_y) / (max_y - min_y)
        K[:, 3, 2] = z_sign
        K[:, 2, 2] = z_sign * zfar / (zfar - znear)
        K[:, 2, 3] = -(zfar * znear) / (zfar - znear)
        return K

    def get_projection_transform(self, **kwargs) -> Transform3d:
        K = kwargs.get("K", self.K)
        if K is None:
            K = self.compute_projection_matrix(self.znear, self.zfar, self.fov, self.aspect_ratio, self.degrees)
        return Transform3d(K.transpose(1, 2).contiguous())

    def get_full_projection_transform(self, **kwargs) -> Transform3d:
        return self.get_world_to_view_transform(**kwargs).compose(self.get_projection_transform(**kwargs))

    def transform_points(self, points, eps=None, **kwargs) -> torch.Tensor:
        return self.get_full_projection_transform(**kwargs).transform_points(points, eps=eps)

    def unproject_points(self, xy_depth: torch.Tensor, world_coordinates: bool = True, scaled_depth_input: bool = False,
                         **kwargs) -> torch.Tensor:
        to_ndc = self.get_full_projection_transform(**kwargs) if world_coordinates else self.get_projection_transform(**kwargs)
        if scaled_depth_input:
            xy_sdepth = xy_depth
        else:
            Km = self.get_projection_transform(**kwargs).get_matrix()
            shape = [1] * xy_depth.dim()
            shape[0] = Km.shape[0]
            f1 = Km[:, 2, 2].reshape(shape)
            f2 = Km[:, 3, 2].reshape(shape)
            sdepth = (f1 * xy_depth[..., 2:3] + f2) / xy_depth[..., 2:3]
            xy_sdepth = torch.cat((xy_depth[..., 0:2], sdepth), dim=-1)
        return to_ndc.inverse().transform_points(xy_sdepth)


PerspectiveCameras = None  # (not used by SuGaR's train / extraction path)
