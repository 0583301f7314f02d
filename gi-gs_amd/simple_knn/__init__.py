"""Import-compatible stand-in for the reference's `simple_knn` extension (submodules/simple-knn): `simple_knn._C.distCUDA2`."""
