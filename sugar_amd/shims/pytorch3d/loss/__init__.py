from .._placeholder import out_of_scope_fn

mesh_laplacian_smoothing = out_of_scope_fn("loss.mesh_laplacian_smoothing")
mesh_normal_consistency = out_of_scope_fn("loss.mesh_normal_consistency")
