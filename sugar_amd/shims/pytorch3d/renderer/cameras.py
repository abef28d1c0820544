from .._placeholder import out_of_scope_fn

_get_sfm_calibration_matrix = out_of_scope_fn("renderer.cameras._get_sfm_calibration_matrix")
