"""Driver for oracle/_ref/libref_rasterizer.so -- TEST INFRASTRUCTURE, NOT PRODUCT.

The library is the REFERENCE rasterizer itself (DGR/cuda_rasterizer/*.cu, compiled unmodified by hipcc for gfx950 by
oracle/ref_build/build_ref.sh).  It exists only on a GPU; the -m gpu tests use it to pin parity against "the reference
run here", and tests/ab_reference.py uses it as the A/B timing baseline.  It cannot be a cpu_baseline.

The reference's private scratch layout (GeometryState / ImageState / BinningState, rasterizer_impl.h:21-66 and
rasterizer_impl.cu:155-194) is decoded here so that the depth-sorted tile lists and ranges can be compared exactly:
every array starts at the next 128-byte boundary after the previous one (`obtain`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_rasterizer.so")  # compiler-default FMA contraction (A/B baseline)
LIB_PATH_NOCONTRACT = os.path.join(_HERE, "_ref", "libref_rasterizer_nocontract.so")  # -ffp-contract=off (bit-exact pin)
LIB_PATH_KNN = os.path.join(_HERE, "_ref", "libref_simple_knn.so")  # the reference's simple-knn (distCUDA2), -ffp-contract=off
_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_libs = {}
_variant = "default"


def available() -> bool:
    return os.path.exists(LIB_PATH) and os.path.exists(LIB_PATH_NOCONTRACT) and os.path.exists(LIB_PATH_KNN)


def dist2(points: torch.Tensor) -> torch.Tensor:
    """The reference's `distCUDA2` (simple_knn.cu:185-221 through spatial.cu:15-26): mean squared distance to the three
    nearest other points.  points: float32 [P,3] on the GPU."""
    if "knn" not in _libs:
        L = C.CDLL(LIB_PATH_KNN)
        L.ref_dist2.restype = None
        L.ref_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _libs["knn"] = L
    pts = points.contiguous().float()
    out = torch.zeros(pts.shape[0], dtype=torch.float32, device=pts.device)  # spatial.cu:19: torch::full({P}, 0.0)
    torch.cuda.synchronize(pts.device)
    _libs["knn"].ref_dist2(pts.shape[0], C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()))
    torch.cuda.synchronize(pts.device)
    return out


def use(variant: str):
    """Select "default" (FMA contraction as the compiler does it) or "nocontract" (individually rounded ops)."""
    global _variant
    assert variant in ("default", "nocontract")
    _variant = variant


def lib():
    if _variant not in _libs:
        L = C.CDLL(LIB_PATH if _variant == "default" else LIB_PATH_NOCONTRACT)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.ref_forward.restype = i
        L.ref_forward.argtypes = [_ALLOC, vp, _ALLOC, vp, _ALLOC, vp, i, i, i, vp, i, i, vp, vp, vp, vp, vp, f, vp, vp, vp,
                                  vp, vp, f, f, i, vp, vp, i]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, i]
        _libs[_variant] = L
    return _libs[_variant]


def _p(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


class _Scratch:
    def __init__(self, dev):
        self.dev, self.t, self.cbs = dev, {}, {}

    def cb(self, name):
        def alloc(_u, n):
            t = torch.empty(int(n), dtype=torch.uint8, device=self.dev)
            self.t[name] = t
            return t.data_ptr()
        fn = _ALLOC(alloc)
        self.cbs[name] = fn
        return fn


def forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=3, scale_modifier=1.0, debug=False):
    """All tensors float32 on the GPU.  Returns a state dict (device tensors + scratch)."""
    dev = means3D.device
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    out = torch.zeros(3, H, W, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    sc = _Scratch(dev)
    torch.cuda.synchronize(dev)
    R = lib().ref_forward(sc.cb("geom"), None, sc.cb("binning"), None, sc.cb("img"), None, P, sh_degree, M, _p(bg), W, H,
                          _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), float(scale_modifier),
                          _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), float(tanfovx),
                          float(tanfovy), 0, _p(out), _p(radii), int(debug))
    torch.cuda.synchronize(dev)
    sc.cbs.clear()  # (break the scratch -> callback -> closure -> scratch cycle: the buffers are freed by reference count)
    return dict(P=P, W=W, H=H, M=M, D=sh_degree, R=int(R), color=out, radii=radii, geom=sc.t["geom"],
                binning=sc.t["binning"], img=sc.t["img"],
                inputs=dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                            cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos, bg=bg,
                            tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier))


def backward(st, dL_dout):
    dev = st["color"].device
    P, W, H, M = st["P"], st["W"], st["H"], st["M"]
    i = st["inputs"]
    z = lambda *s: torch.zeros(*s, device=dev)  # rasterize_points.cu:151-159
    g = dict(dL_dmeans3D=z(P, 3), dL_dmeans2D=z(P, 3), dL_dcolors=z(P, 3), dL_dconic=z(P, 2, 2), dL_dopacity=z(P, 1),
             dL_dcov3D=z(P, 6), dL_dsh=z(P, M, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    dL = dL_dout.contiguous()
    torch.cuda.synchronize(dev)
    lib().ref_backward(P, st["D"], M, st["R"], _p(i["bg"]), W, H, _p(i["means3D"]), _p(i["shs"]), _p(i["colors_precomp"]),
                       _p(i["scales"]), float(i["scale_modifier"]), _p(i["rotations"]), _p(i["cov3D_precomp"]),
                       _p(i["viewmatrix"]), _p(i["projmatrix"]), _p(i["campos"]), float(i["tanfovx"]), float(i["tanfovy"]),
                       _p(st["radii"]), _p(st["geom"]), _p(st["binning"]), _p(st["img"]), _p(dL), _p(g["dL_dmeans2D"]),
                       _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]),
                       _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]), 0)
    torch.cuda.synchronize(dev)
    return g


def _carve(base_ptr: int, spec):
    """Replays `obtain` (rasterizer_impl.h:21-27): returns {name: byte offset} for [(name, count, itemsize)]."""
    offs, cur = {}, base_ptr
    for name, count, size in spec:
        cur = (cur + 127) & ~127
        offs[name] = cur - base_ptr
        cur += count * size
    return offs


def decode(st):
    """numpy views of the reference's internal state: sorted tile lists, ranges, projected geometry."""
    P, W, H, R = st["P"], st["W"], st["H"], st["R"]
    N = W * H
    out = {}
    img = st["img"].cpu().numpy()
    o = _carve(st["img"].data_ptr(), [("accum_alpha", N, 4), ("n_contrib", N, 4), ("ranges", N, 8)])
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out["final_T"] = img[o["accum_alpha"]: o["accum_alpha"] + 4 * N].view(np.float32)
    out["n_contrib"] = img[o["n_contrib"]: o["n_contrib"] + 4 * N].view(np.uint32)
    out["ranges"] = img[o["ranges"]: o["ranges"] + 8 * T].view(np.uint32).reshape(T, 2)
    binb = st["binning"].cpu().numpy()
    o = _carve(st["binning"].data_ptr(), [("point_list", R, 4), ("point_list_unsorted", R, 4), ("keys", R, 8)])
    out["point_list"] = binb[o["point_list"]: o["point_list"] + 4 * R].view(np.uint32)
    out["point_list_keys"] = binb[o["keys"]: o["keys"] + 8 * R].view(np.uint64)
    geom = st["geom"].cpu().numpy()
    o = _carve(st["geom"].data_ptr(), [("depths", P, 4), ("clamped", 3 * P, 1), ("radii", P, 4), ("means2D", P, 8),
                                       ("cov3D", 6 * P, 4), ("conic_opacity", P, 16), ("rgb", 3 * P, 4),
                                       ("tiles_touched", P, 4)])
    out["depths"] = geom[o["depths"]: o["depths"] + 4 * P].view(np.float32)
    out["clamped"] = geom[o["clamped"]: o["clamped"] + 3 * P].reshape(P, 3)
    out["means2D"] = geom[o["means2D"]: o["means2D"] + 8 * P].view(np.float32).reshape(P, 2)
    out["cov3D"] = geom[o["cov3D"]: o["cov3D"] + 24 * P].view(np.float32).reshape(P, 6)
    out["conic_opacity"] = geom[o["conic_opacity"]: o["conic_opacity"] + 16 * P].view(np.float32).reshape(P, 4)
    out["rgb"] = geom[o["rgb"]: o["rgb"] + 12 * P].view(np.float32).reshape(P, 3)
    out["tiles_touched"] = geom[o["tiles_touched"]: o["tiles_touched"] + 4 * P].view(np.uint32)
    out["color"] = st["color"].cpu().numpy()
    out["radii"] = st["radii"].cpu().numpy()
    out["num_rendered"] = R
    return out
