"""pytorch3d.ops entry points SuGaR uses (sugar_scene/sugar_model.py:7): `knn_points` on the HIP k-NN of this package and
`estimate_pointcloud_normals` built on it."""
import torch

from sugar_amd.knn import knn_points  # noqa: F401  (exact, squared distances ascending, int64 indices)


def estimate_pointcloud_normals(pointclouds, neighborhood_size: int = 50, disambiguate_directions: bool = True, *,
                                use_symeig_workaround: bool = True):
    """Restatement of the published pytorch3d 0.7.4 algorithm (ops/points_normals.py): per point the covariance of its
    `neighborhood_size` nearest neighbours (the point itself included) about their mean, normal = eigenvector of the
    smallest eigenvalue, flipped so that at least half of the neighbours lie on its positive side.  pointclouds[1,N,3] on a
    ROCm device.  PARITY UNPINNED (third-party code absent from the reference tree; call site sugar_model.py:957)."""
    if not torch.is_tensor(pointclouds) or pointclouds.dim() != 3 or pointclouds.shape[0] != 1:
        raise ValueError("estimate_pointcloud_normals: expected a tensor [1, N, 3]")
    pts = pointclouds[0].float()
    N = pts.shape[0]
    if N <= neighborhood_size:
        raise ValueError("The neighborhood_size argument has to be strictly smaller than the number of points in the cloud.")
    K = int(neighborhood_size)
    if K > 32:
        raise ValueError("neighborhood_size up to 32 is supported")
    idx = knn_points(pts[None].contiguous(), pts[None].contiguous(), K=K).idx[0]
    nbrs = pts[idx]                                   # [N,K,3]
    central = nbrs - nbrs.mean(dim=1, keepdim=True)
    cov = (central.unsqueeze(3) * central.unsqueeze(2)).mean(dim=1)  # [N,3,3]
    _, vecs = torch.linalg.eigh(cov.double() if use_symeig_workaround else cov)  # ascending eigenvalues
    normals = vecs[:, :, 0].to(pts.dtype)
    if disambiguate_directions:
        df = nbrs - pts[:, None]
        proj = (normals[:, None] * df).sum(2)
        n_pos = (proj > 0).to(pts.dtype).sum(1, keepdim=True)
        flip = (n_pos < 0.5 * K).to(pts.dtype)
        normals = (1.0 - 2.0 * flip) * normals
    return normals[None]
