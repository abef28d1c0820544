"""`simple_knn._C` of the reference (simple-knn/ext.cpp:15-17, spatial.cu:15-26), backed by sgr_dist2 / sgr_knn of
include/sugar_raster.h.  GPU tensors only; the HIP library must be built."""
from sugar_amd.knn import distCUDA2, knn_points  # noqa: F401
