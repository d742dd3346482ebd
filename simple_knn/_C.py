"""`simple_knn._C` stand-in: re-exports the MI355X-native kernel (dreamscene_amd/csrc/knn.hip)."""
from dreamscene_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
