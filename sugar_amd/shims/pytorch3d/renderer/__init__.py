"""`pytorch3d.renderer` as far as SuGaR's train / level-set path needs it: the camera algebra (cameras.py) is functional;
`TexturesVertex` / `TexturesUV` are containers; `RasterizationSettings` / `MeshRasterizer` / `Fragments` (mesh/) give the hard
(blur_radius = 0) z-buffer the level-set sampler reads (sugar_scene/sugar_model.py:1880-1893,1927-1928,1966) on the HIP kernel
of this package (sgr_rasterize_meshes).  Shading / texture sampling stay out of scope."""
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


from .mesh import Fragments, MeshRasterizer, RasterizationSettings, rasterize_meshes  # noqa: E402,F401
