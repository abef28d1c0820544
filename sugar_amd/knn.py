"""Host side of the k-NN helpers (C ABI: sgr_dist2, sgr_knn in include/sugar_raster.h).

  distCUDA2(points[P,3]) -> float[P]         simple-knn/spatial.cu:15-26 (mean squared distance to the 3 nearest others)
  knn_points(p1[1,N,3], p2[1,M,3], K)        the call shape SuGaR uses from pytorch3d.ops
                                             (sugar_scene/sugar_model.py:49,235,1028,1342): returns (dists[1,N,K] squared,
                                             ascending; idx[1,N,K] int64; None)
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib


class _KNN(NamedTuple):
    dists: torch.Tensor
    idx: torch.Tensor
    knn: Optional[torch.Tensor]


def _check(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP k-NN needs tensors on a ROCm device; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


SUPPORTED_K = (1, 2, 3, 4, 8, 16, 32)  # kernel instantiations; other K <= 32 take the next one and a prefix of its result
GRID_THRESHOLD = 4096  # below this the exhaustive LDS-tiled kernel is as fast; both give identical results


def distCUDA2(points: torch.Tensor, method: str = "auto") -> torch.Tensor:
    lib = _lib.load()
    points = _check(points, "points")
    P = points.shape[0]
    out = torch.zeros(P, dtype=torch.float32, device=points.device)  # spatial.cu:19 (full 0.0)
    if P:
        with torch.cuda.device(points.device):
            stream = C.c_void_p(torch.cuda.current_stream(points.device).cuda_stream)
            if method == "grid" or (method == "auto" and P >= GRID_THRESHOLD):
                scratch = torch.empty(lib.sgr_knn_grid_scratch_bytes(P), dtype=torch.uint8, device=points.device)
                rc = lib.sgr_dist2_grid(P, C.c_void_p(points.data_ptr()), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(scratch.data_ptr()), stream)
            else:
                rc = lib.sgr_dist2(P, C.c_void_p(points.data_ptr()), C.c_void_p(out.data_ptr()), stream)
        if rc < 0:
            raise RuntimeError(f"sgr_dist2 failed ({rc})")
    return out


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1=None, lengths2=None, norm: int = 2, K: int = 1, version: int = -1,
               return_nn: bool = False, return_sorted: bool = True, method: str = "auto") -> _KNN:
    """pytorch3d.ops.knn_points for the call shape SuGaR uses (one cloud per call, 3-D, squared L2, sorted).  The
    remaining pytorch3d arguments are accepted for signature compatibility; unsupported values raise."""
    if lengths1 is not None or lengths2 is not None or norm != 2:
        raise NotImplementedError("knn_points: lengths1/lengths2/norm=1 are not supported by the HIP k-NN")
    lib = _lib.load()
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[0] != 1 or p2.shape[0] != 1 or p1.shape[2] != 3 or p2.shape[2] != 3:
        raise RuntimeError("knn_points: expected p1[1,N,3], p2[1,M,3] (the shapes SuGaR uses)")
    q, r = _check(p1[0], "p1"), _check(p2[0], "p2")
    N, M = q.shape[0], r.shape[0]
    K_req = int(K)
    if K_req < 1 or K_req > SUPPORTED_K[-1]:
        raise RuntimeError(f"knn_points: K must be in 1..{SUPPORTED_K[-1]}")
    K = next(k for k in SUPPORTED_K if k >= K_req)  # the kernels are instantiated for these; a prefix of a longer list is exact
    d = torch.empty(N, K, dtype=torch.float32, device=q.device)
    i = torch.empty(N, K, dtype=torch.int64, device=q.device)
    with torch.cuda.device(q.device):
        stream = C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        if method == "grid" or (method == "auto" and M >= GRID_THRESHOLD):
            scratch = torch.empty(lib.sgr_knn_grid_scratch_bytes(M), dtype=torch.uint8, device=q.device)
            rc = lib.sgr_knn_grid(N, C.c_void_p(q.data_ptr()), M, C.c_void_p(r.data_ptr()), int(K), C.c_void_p(d.data_ptr()),
                                  C.c_void_p(i.data_ptr()), C.c_void_p(scratch.data_ptr()), stream)
        else:
            rc = lib.sgr_knn(N, C.c_void_p(q.data_ptr()), M, C.c_void_p(r.data_ptr()), int(K), C.c_void_p(d.data_ptr()),
                             C.c_void_p(i.data_ptr()), stream)
    if rc < 0:
        raise RuntimeError(f"sgr_knn failed ({rc})")
    if K != K_req:
        d, i = d[:, :K_req].contiguous(), i[:, :K_req].contiguous()
    return _KNN(d[None], i[None], r[i][None] if return_nn else None)


def knn_points_pytorch3d(original):
    """Wrap an installed pytorch3d.ops.knn_points: SuGaR's call shape on a ROCm device goes to the HIP k-NN, anything else
    (batched clouds, lengths, other dimensions, L1) to the original."""
    def knn_points_dispatch(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
        ok = (p1.is_cuda and p1.dim() == 3 and p2.dim() == 3 and p1.shape[0] == 1 and p2.shape[0] == 1 and p1.shape[2] == 3 and
              p2.shape[2] == 3 and lengths1 is None and lengths2 is None and norm == 2 and 1 <= K <= SUPPORTED_K[-1] and
              p1.dtype == torch.float32 and p2.dtype == torch.float32 and not (p1.requires_grad or p2.requires_grad))
        if ok:
            return knn_points(p1, p2, K=K, return_nn=return_nn)
        return original(p1, p2, lengths1=lengths1, lengths2=lengths2, norm=norm, K=K, version=version, return_nn=return_nn,
                        return_sorted=return_sorted)
    return knn_points_dispatch
