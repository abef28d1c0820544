"""`pytorch3d.renderer` as far as SuGaR's train / level-set path needs it: the camera algebra (cameras.py) is functional;
mesh rasterization is outside this package's scope -- `RasterizationSettings` and `MeshRasterizer` can be CONSTRUCTED
(SuGaR builds them unconditionally, sugar_scene/sugar_model.py:1880-1893, before it knows whether the Gaussian-depth path
is taken) and raise when a mesh is actually rasterized."""
from .._placeholder import out_of_scope
from . import cameras  # noqa: F401
from .cameras import FoVPerspectiveCameras  # noqa: F401

TexturesUV = out_of_scope("renderer.TexturesUV")
TexturesVertex = out_of_scope("renderer.TexturesVertex")
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
