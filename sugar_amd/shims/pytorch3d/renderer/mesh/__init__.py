"""`pytorch3d.renderer.mesh` as far as SuGaR's level-set sampler needs it: the hard (blur_radius = 0) mesh rasterizer on the
HIP z-buffer kernel of this package, pytorch3d's Python-side near-plane clipping restated, and the `Fragments` tuple."""
from .clip import ClipFrustum, ClippedFaces, clip_faces, convert_clipped_rasterization_to_original_faces  # noqa: F401
from .rasterize_meshes import rasterize_meshes  # noqa: F401
from .rasterizer import Fragments, MeshRasterizer, RasterizationSettings  # noqa: F401
