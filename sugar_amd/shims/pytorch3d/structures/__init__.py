"""`pytorch3d.structures` as far as SuGaR's surface-bound (refine) model needs it: a `Meshes` container built from lists of
vertex / face tensors with the accessors the reference calls -- `verts_list`, `faces_list`, `faces_normals_list`,
`verts_packed`, `faces_packed`, `edges_packed`, `textures` (call sites: sugar_scene/sugar_model.py:350,451,552-560,581,964;
sugar_trainers/refine.py:778-783 hands the same object to the two mesh regularisers in `pytorch3d.loss`).

Plain differentiable torch, no kernels: the bound model's Gaussian parameters are derived from the mesh (barycentric
positions, in-plane scales, rotations from the face frame) and THEN cross the rasterizer boundary this package accelerates.
Restated from the public pytorch3d 0.7.4 API (environment.yml:161); pytorch3d is not installed here, so this module is
PARITY-UNPINNED against pytorch3d itself.  Face normals follow pytorch3d's `mesh_face_areas_normals`: the cross product
(v1 - v0) x (v2 - v0) divided by max(its length, 1e-6); every reference call site normalises them again."""
from __future__ import annotations

import torch

from .._placeholder import out_of_scope

Pointclouds = out_of_scope("structures.Pointclouds")


class Meshes:
    def __init__(self, verts=None, faces=None, textures=None, *, verts_normals=None):
        if torch.is_tensor(verts):
            verts = list(verts.unbind(0)) if verts.dim() == 3 else [verts]
        if torch.is_tensor(faces):
            faces = list(faces.unbind(0)) if faces.dim() == 3 else [faces]
        if verts is None or faces is None or len(verts) != len(faces):
            raise ValueError("Meshes: verts and faces must be lists of the same length")
        for v, f in zip(verts, faces):
            if v.dim() != 2 or v.shape[1] != 3 or f.dim() != 2 or f.shape[1] != 3:
                raise ValueError("Meshes: expected verts [V,3] and faces [F,3]")
        self._verts = list(verts)
        self._faces = [f.long() for f in faces]
        self.textures = textures
        self.device = self._verts[0].device if self._verts else torch.device("cpu")

    def __len__(self):
        return len(self._verts)

    def __getitem__(self, i):
        if isinstance(i, int):
            i = [i]
        tex = self.textures[i] if self.textures is not None and hasattr(self.textures, "__getitem__") else self.textures
        return Meshes([self._verts[k] for k in i], [self._faces[k] for k in i], textures=tex)

    def to(self, device):
        tex = self.textures.to(device) if self.textures is not None and hasattr(self.textures, "to") else self.textures
        return Meshes([v.to(device) for v in self._verts], [f.to(device) for f in self._faces], textures=tex)

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def isempty(self):
        return len(self) == 0 or all(len(v) == 0 for v in self._verts)

    # ---- lists
    def verts_list(self):
        return self._verts

    def faces_list(self):
        return self._faces

    def faces_normals_list(self):
        out = []
        for v, f in zip(self._verts, self._faces):
            tri = v[f]
            n = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=-1)
            out.append(n / n.norm(dim=-1, keepdim=True).clamp_min(1e-6))
        return out

    def num_verts_per_mesh(self):
        return torch.tensor([len(v) for v in self._verts], device=self.device)

    def num_faces_per_mesh(self):
        return torch.tensor([len(f) for f in self._faces], device=self.device)

    # ---- packed (all meshes concatenated; face indices offset into the packed vertex array)
    def verts_packed(self):
        return torch.cat(self._verts, dim=0) if len(self._verts) > 1 else self._verts[0]

    def _offsets(self):
        off, acc = [], 0
        for v in self._verts:
            off.append(acc)
            acc += len(v)
        return off

    def faces_packed(self):
        return torch.cat([f + o for f, o in zip(self._faces, self._offsets())], dim=0)

    def faces_normals_packed(self):
        return torch.cat(self.faces_normals_list(), dim=0)

    def verts_packed_to_mesh_idx(self):
        return torch.cat([torch.full((len(v),), k, dtype=torch.int64, device=self.device) for k, v in enumerate(self._verts)])

    def faces_packed_to_mesh_idx(self):
        return torch.cat([torch.full((len(f),), k, dtype=torch.int64, device=self.device) for k, f in enumerate(self._faces)])

    def edges_packed(self):
        """unique undirected edges [E,2] with v0 < v1, sorted by (v0, v1) -- pytorch3d's order"""
        return self._edges()[0]

    def faces_packed_to_edges_packed(self):
        """[F,3]: for face (a,b,c) the edge ids of (b,c), (c,a), (a,b) -- pytorch3d's column order"""
        return self._edges()[1]

    def _edges(self):
        f = self.faces_packed()
        V = max(int(sum(len(v) for v in self._verts)), 1)
        e = torch.cat([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], dim=0)
        e = torch.sort(e, dim=1).values
        key = e[:, 0] * V + e[:, 1]
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        edges = torch.stack([uniq // V, uniq % V], dim=1)
        return edges, inv.reshape(3, -1).t().contiguous()

    def edges_packed_to_mesh_idx(self):
        e = self.edges_packed()
        return self.verts_packed_to_mesh_idx()[e[:, 0]]
