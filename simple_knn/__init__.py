"""Drop-in for the reference's `simple_knn` package (gaussian_splatting/submodules/simple-knn): `from simple_knn._C import
distCUDA2` resolves to the MI355X HIP implementation."""
