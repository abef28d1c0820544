from .._placeholder import out_of_scope
from . import cameras  # noqa: F401

TexturesUV = out_of_scope("renderer.TexturesUV")
TexturesVertex = out_of_scope("renderer.TexturesVertex")
RasterizationSettings = out_of_scope("renderer.RasterizationSettings")
MeshRasterizer = out_of_scope("renderer.MeshRasterizer")
FoVPerspectiveCameras = out_of_scope("renderer.FoVPerspectiveCameras")
PerspectiveCameras = out_of_scope("renderer.PerspectiveCameras")
