from .._placeholder import out_of_scope

Meshes = out_of_scope("structures.Meshes")
Pointclouds = out_of_scope("structures.Pointclouds")
