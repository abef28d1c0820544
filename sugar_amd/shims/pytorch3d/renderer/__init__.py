"""`pytorch3d.renderer` as far as SuGaR's train / level-set path needs it: the camera algebra (cameras.py) is functional;
`TexturesVertex` / `TexturesUV` are
containers; mesh rasterization is outside this package's scope -- `RasterizationSettings` and `MeshRasterizer` can be CONSTRUCTED
(SuGaR builds them unconditionally, sugar_scene/sugar_model.py:1880-1893, before it knows whether the Gaussian-depth path
is taken) and raise when a mesh is actually rasterized."""
import torch

from .._placeholder import out_of_scope
from . import cameras  # noqa: F401
from .cameras import FoVPerspectiveCameras  # noqa: F401



class TexturesVertex:
    """per-vertex colours carried by a `Meshes` (sugar_scene/sugar_model.py:557): a container, nothing samples it here"""
    def __init__(self, verts_features):
        if torch.is_tensor(verts_features) and verts_features.dim() == 3:
            verts_features = list(verts_features.unbind(0))
        self._verts_features = list(verts_features)

    def verts_features_list(self):
        return self._verts_features

    def verts_features_packed(self):
        return torch.cat(self._verts_features, dim=0)

    def verts_features_padded(self):
        return torch.nn.utils.rnn.pad_sequence(self._verts_features, batch_first=True)

    def __getitem__(self, i):
        return TexturesVertex([self._verts_features[k] for k in ([i] if isinstance(i, int) else i)])

    def to(self, device):
        return TexturesVertex([v.to(device) for v in self._verts_features])


class TexturesUV:
    """a UV atlas carried by a `Meshes` (sugar_scene/sugar_model.py:538-546, 682-690, 717-725, 2636; the reference also
    assigns `_maps_padded` directly, :1468): a container, sampling it needs the mesh rasterizer and is out of scope"""
    def __init__(self, maps, faces_uvs, verts_uvs, padding_mode="border", align_corners=True, sampling_mode="bilinear"):
        self._maps_padded = torch.stack(list(maps)) if isinstance(maps, (list, tuple)) else maps
        self._faces_uvs = list(faces_uvs.unbind(0)) if torch.is_tensor(faces_uvs) else list(faces_uvs)
        self._verts_uvs = list(verts_uvs.unbind(0)) if torch.is_tensor(verts_uvs) else list(verts_uvs)
        self.padding_mode, self.align_corners, self.sampling_mode = padding_mode, align_corners, sampling_mode

    def maps_padded(self):
        return self._maps_padded

    def faces_uvs_list(self):
        return self._faces_uvs

    def verts_uvs_list(self):
        return self._verts_uvs

    def to(self, device):
        return TexturesUV(self._maps_padded.to(device), [f.to(device) for f in self._faces_uvs],
                          [v.to(device) for v in self._verts_uvs], self.padding_mode, self.align_corners, self.sampling_mode)

    def sample_textures(self, *a, **k):
        raise NotImplementedError("pytorch3d.renderer.TexturesUV.sample_textures needs the mesh rasterizer's fragments, which are "
                                  "outside the scope of the sugar_amd stand-in package; install pytorch3d")


PerspectiveCameras = out_of_scope("renderer.PerspectiveCameras")


class RasterizationSettings:
    def __init__(self, image_size=256, blur_radius=0.0, faces_per_pixel=1, bin_size=None, max_faces_per_bin=None,
                 perspective_correct=None, clip_barycentric_coords=None, cull_backfaces=False, z_clip_value=None,
                 cull_to_frustum=False, **kw):
        self.__dict__.update(image_size=image_size, blur_radius=blur_radius, faces_per_pixel=faces_per_pixel, bin_size=bin_size,
                             max_faces_per_bin=max_faces_per_bin, perspective_correct=perspective_correct,
                             clip_barycentric_coords=clip_barycentric_coords, cull_backfaces=cull_backfaces,
                             z_clip_value=z_clip_value, cull_to_frustum=cull_to_frustum, **kw)


class MeshRasterizer:
    def __init__(self, cameras=None, raster_settings=None):
        self.cameras, self.raster_settings = cameras, raster_settings

    def __call__(self, *a, **k):
        raise NotImplementedError("pytorch3d.renderer.MeshRasterizer: mesh rasterization is outside the scope of the sugar_amd "
                                  "stand-in package; use the Gaussian-depth path (use_gaussian_depth=True) or install pytorch3d")

    forward = __call__
