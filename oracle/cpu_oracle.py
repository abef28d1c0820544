"""ctypes/numpy driver for oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT.

Composes the C restatement's stages in the order of CudaRasterizer::Rasterizer::forward / backward
(DGR/cuda_rasterizer/rasterizer_impl.cu:198-336, :340-434) and returns every intermediate array so
the parity tests can compare the HIP path stage by stage.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("cpu_rasterizer.c", "mesh_rasterizer.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_scan.restype = C.c_int64
        _LIB.orc_getHigherMsb.restype = C.c_uint32
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().orc_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
            sh_degree=3, scale_modifier=1.0):
    """Returns a dict with the outputs (color [3,H,W], radii [P]) and all GeometryState / BinningState /
    ImageState arrays of the reference (rasterizer_impl.h:33-66)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(-1)
    shs = _f32(shs); colors_precomp = _f32(colors_precomp)
    scales = _f32(scales); rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp)
    viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos); bg = _f32(bg)
    M = 0 if shs is None else shs.shape[1]
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    L.orc_preprocess(C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales), C.c_float(scale_modifier),
                     _p(rotations), _p(opacities), _p(shs), _p(st["clamped"]), _p(cov3D_precomp), _p(colors_precomp),
                     _p(viewmatrix), _p(projmatrix), _p(campos), C.c_int(W), C.c_int(H), C.c_float(tanfovx),
                     C.c_float(tanfovy), _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]),
                     _p(st["rgb"]), _p(st["conic_opacity"]), _p(st["tiles_touched"]))
    st["point_offsets"] = np.zeros(P, np.uint32)
    R = int(L.orc_scan(C.c_int(P), _p(st["tiles_touched"]), _p(st["point_offsets"])))
    st["num_rendered"] = R
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st["grid"] = (gx, gy)
    st["point_list_keys"] = np.zeros(R, np.uint64)
    st["point_list"] = np.zeros(R, np.uint32)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    L.orc_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]),
              _p(st["point_offsets"]), C.c_int64(R), _p(st["point_list_keys"]), _p(st["point_list"]), _p(st["ranges"]))
    feat = colors_precomp if colors_precomp is not None else st["rgb"]
    st["final_T"] = np.zeros(W * H, np.float32)
    st["n_contrib"] = np.zeros(W * H, np.uint32)
    st["color"] = np.zeros((3, H, W), np.float32)
    st["walked"] = np.zeros(gx * gy, np.uint32)
    L.orc_render_forward(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["means2D"]), _p(feat),
                         _p(st["conic_opacity"]), _p(st["final_T"]), _p(st["n_contrib"]), _p(bg), _p(st["color"]),
                         _p(st["walked"]))
    st["_inputs"] = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                         cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos,
                         bg=bg, tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier)
    return st


def backward(st, dL_dout_color):
    """Mirrors Rasterizer::backward (rasterizer_impl.cu:340-434) with the zero-initialised outputs of
    RasterizeGaussiansBackwardCUDA (DGR/rasterize_points.cu:151-159)."""
    L = lib()
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    i = st["_inputs"]
    dL = _f32(dL_dout_color).reshape(3, H, W)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32))
    colors = i["colors_precomp"] if i["colors_precomp"] is not None else st["rgb"]
    L.orc_render_backward(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(i["bg"]), _p(st["means2D"]),
                          _p(st["conic_opacity"]), _p(colors), _p(st["final_T"]), _p(st["n_contrib"]), _p(dL),
                          _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    cov3D_ptr = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st["cov3D"]
    L.orc_preprocess_backward(C.c_int(P), C.c_int(D), C.c_int(M), _p(i["means3D"]), _p(st["radii"]), _p(i["shs"]),
                              _p(st["clamped"]), _p(i["scales"]), _p(i["rotations"]), C.c_float(i["scale_modifier"]),
                              _p(cov3D_ptr), _p(i["viewmatrix"]), _p(i["projmatrix"]), C.c_int(W), C.c_int(H),
                              C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _p(i["campos"]), _p(g["dL_dmeans2D"]),
                              _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]), _p(g["dL_dcov3D"]),
                              _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def dist2(points):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, np.float32)
    lib().orc_dist2(C.c_int(P), _p(points), _p(out))
    return out


def set_threads(n: int):
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(n))
    except OSError:
        pass
