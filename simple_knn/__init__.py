"""Drop-in for the `simple_knn` package DreamScene imports (gs_renderer.py:9): `from simple_knn._C import distCUDA2`."""
