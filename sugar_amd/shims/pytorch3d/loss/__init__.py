"""`pytorch3d.loss` as far as SuGaR's trainers use it: the two mesh regularisers applied to the surface mesh the refine-mode
Gaussians are bound to (sugar_trainers/refine.py:778-783, coarse_density.py:741-746, coarse_sdf.py:727-732; the Laplacian
is always called with method="uniform", :163-166).  Differentiable torch over the stand-in `Meshes`; restated from the public
pytorch3d 0.7.4 definitions, PARITY-UNPINNED against pytorch3d itself (not installed here) -- tests/test_shims.py checks them
against closed-form values on small meshes.

  mesh_normal_consistency : for every pair of faces sharing an edge (v0,v1) with opposite vertices a and b,
                            1 - cos(n0, n1) with n0 = (v1-v0) x (a-v0), n1 = -(v1-v0) x (b-v0); each mesh's pairs are averaged,
                            then the meshes are.
  mesh_laplacian_smoothing: uniform Laplacian L = D^-1 A - I; the loss is the mean over vertices of |L v|, averaged over meshes.
"""
from __future__ import annotations

import torch


def _face_pairs_by_edge(edge_of_incidence: torch.Tensor):
    """incidences (face-edge slots) sorted by edge id -> index pairs (i, j) into the sorted order, one per pair of faces
    sharing an edge (k faces on an edge give k(k-1)/2 pairs)"""
    E = int(edge_of_incidence.max()) + 1 if len(edge_of_incidence) else 0
    count = torch.bincount(edge_of_incidence, minlength=E)
    start = torch.cumsum(count, 0) - count
    two = (count == 2).nonzero().reshape(-1)
    pairs = [torch.stack([start[two], start[two] + 1], dim=1)]
    for e in (count > 2).nonzero().reshape(-1).tolist():  # non-manifold edges: rare, every pair
        s, k = int(start[e]), int(count[e])
        ij = torch.combinations(torch.arange(s, s + k, device=count.device), r=2)
        pairs.append(ij)
    return torch.cat(pairs, dim=0)


def mesh_normal_consistency(meshes):
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
    N = len(meshes)
    verts = meshes.verts_packed()
    faces = meshes.faces_packed()
    edges = meshes.edges_packed()
    F = faces.shape[0]
    with torch.no_grad():
        edge_idx = meshes.faces_packed_to_edges_packed().reshape(F * 3)          # slot (f, k) -> edge id
        face_verts = faces[:, None, :].expand(F, 3, 3).reshape(F * 3, 3)          # the slot's face
        edge_idx, order = edge_idx.sort()
        face_verts = face_verts[order]
        pair = _face_pairs_by_edge(edge_idx)
        if pair.shape[0] == 0:
            return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
        v0_idx = edges[edge_idx, 0]
        v1_idx = edges[edge_idx, 1]
        opposite = (face_verts != v0_idx[:, None]) & (face_verts != v1_idx[:, None])
        # (a degenerate face repeating a vertex would give 0 or 2 candidates: take the first)
        other = torch.where(opposite, face_verts, torch.full_like(face_verts, -1)).max(dim=1).values.clamp_min(0)
        mesh_of_pair = meshes.verts_packed_to_mesh_idx()[v0_idx[pair[:, 0]]]
        weights = 1.0 / torch.bincount(mesh_of_pair, minlength=N)[mesh_of_pair].float()
    v0 = verts[v0_idx[pair[:, 0]]]
    v1 = verts[v1_idx[pair[:, 0]]]
    a = verts[other[pair[:, 0]]]
    b = verts[other[pair[:, 1]]]
    n0 = torch.cross(v1 - v0, a - v0, dim=1)
    n1 = -torch.cross(v1 - v0, b - v0, dim=1)
    loss = 1.0 - torch.cosine_similarity(n0, n1, dim=1)
    return (loss * weights).sum() / N


def mesh_laplacian_smoothing(meshes, method: str = "uniform"):
    if method != "uniform":
        raise NotImplementedError(f"pytorch3d.loss.mesh_laplacian_smoothing(method={method!r}): the sugar_amd stand-in has the "
                                  "'uniform' Laplacian only (the one SuGaR's trainers use); install pytorch3d for 'cot'/'cotcurv'")
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
    N = len(meshes)
    verts = meshes.verts_packed()
    V = verts.shape[0]
    with torch.no_grad():
        e = meshes.edges_packed()
        deg = torch.zeros(V, dtype=verts.dtype, device=verts.device)
        deg.index_add_(0, e.reshape(-1), torch.ones(e.numel(), dtype=verts.dtype, device=verts.device))
        inv_deg = torch.where(deg > 0, 1.0 / deg.clamp_min(1.0), torch.zeros_like(deg))
        mesh_idx = meshes.verts_packed_to_mesh_idx()
        weights = 1.0 / meshes.num_verts_per_mesh()[mesh_idx].to(verts.dtype)
    nb = torch.zeros_like(verts)
    nb = nb.index_add(0, e[:, 0], verts[e[:, 1]]).index_add(0, e[:, 1], verts[e[:, 0]])
    lap = nb * inv_deg[:, None] - verts                      # L v, L = D^-1 A - I
    return (lap.norm(dim=1) * weights).sum() / N
