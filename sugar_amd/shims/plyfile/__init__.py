"""Minimal stand-in for the `plyfile` package (absent from the ROCm image), activated by `sugar_amd.shims.install()` only when
the real one cannot be imported.  It covers the calls the reference makes on a 3DGS point cloud and nothing more:

  gaussian_splatting/scene/gaussian_model.py:207-208   PlyElement.describe(structured_array, 'vertex'); PlyData([el]).write(path)
  gaussian_splatting/scene/gaussian_model.py:216-247   PlyData.read(path).elements[0][name] / .properties[i].name
  gaussian_splatting/scene/dataset_readers.py:108-129  PlyData.read(path)['vertex'][name]

Only the vertex element of a binary_little_endian file is read (the format both writers above produce); reading goes through
sugar_amd.io, the same parser `load_gaussian_ply` uses."""
from __future__ import annotations

import numpy as np

_NAMES = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint"}


class PlyProperty:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, np.dtype(dtype)

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.dtype.str!r})"


class PlyElement:
    def __init__(self, name, data):
        if data.dtype.names is None:
            raise ValueError("PlyElement needs a structured array (one field per property)")
        self.name, self.data = name, data
        self.properties = tuple(PlyProperty(n, data.dtype[n]) for n in data.dtype.names)

    @staticmethod
    def describe(data, name, **kwargs):
        if kwargs:
            raise NotImplementedError(f"plyfile stand-in: PlyElement.describe({', '.join(kwargs)}=...) is not covered")
        return PlyElement(name, np.asarray(data))

    @property
    def count(self):
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="<", comments=(), obj_info=()):
        if text:
            raise NotImplementedError("plyfile stand-in: only binary_little_endian files are written")
        self.elements = list(elements)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)

    @staticmethod
    def read(path):
        from sugar_amd.io import _read_ply_vertices
        return PlyData([PlyElement("vertex", _read_ply_vertices(str(path)))])

    def write(self, path):
        header = "ply\nformat binary_little_endian 1.0\n"
        for e in self.elements:
            header += f"element {e.name} {len(e.data)}\n"
            for p in e.properties:
                code = p.dtype.str.lstrip("<>=|")
                if code not in _NAMES:
                    raise NotImplementedError(f"plyfile stand-in: property {p.name!r} of type {p.dtype}")
                header += f"property {_NAMES[code]} {p.name}\n"
        header += "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            for e in self.elements:
                f.write(np.ascontiguousarray(e.data.astype(e.data.dtype.newbyteorder("<"))).tobytes())
