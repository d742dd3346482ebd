"""distCUDA2 replacement. CPU: the oracle's two paths agree; GPU: HIP vs oracle."""
import numpy as np
import pytest
import torch


def _clouds():
    rng = np.random.default_rng(0)
    from dreamscene_amd import synth
    yield "uniform-5k", rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    yield "object-60k", synth.g_object(60000, seed=2)["means3D"]
    yield "indoor-sheets", synth.g_indoor(seed=1, per_wall=8000)["means3D"]
    dup = rng.normal(size=(3000, 3)).astype(np.float32)
    dup[1000:1500] = dup[:500]                       # exact duplicates: distance 0 to a different point
    yield "duplicates", dup
    yield "line", np.stack([np.linspace(0, 1, 777), np.zeros(777), np.zeros(777)], 1).astype(np.float32)
    yield "tiny-5", rng.normal(size=(5, 3)).astype(np.float32)
    yield "one-cell", (rng.normal(size=(300, 3)) * 1e-3 + 5.0).astype(np.float32)


def test_oracle_paths_agree():
    from oracle import knn_oracle as K
    rng = np.random.default_rng(3)
    p = rng.normal(size=(1500, 3)).astype(np.float32)
    a = K.mean_dist2_brute(p)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=4)
    b = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(a, b, rtol=2e-5)


def test_dropin_package_exports_symbol():
    import simple_knn._C as C
    assert callable(C.distCUDA2)
    with pytest.raises(Exception):
        C.distCUDA2(torch.zeros(10, 3))          # CPU tensor: refuses loudly


@pytest.mark.gpu
@pytest.mark.parametrize("name,pts", list(_clouds()))
def test_knn_hip_vs_oracle(built_lib, name, pts):
    from oracle import knn_oracle as K
    from simple_knn._C import distCUDA2
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    ref = K.mean_dist2(pts)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-12, err_msg=name)


@pytest.mark.gpu
def test_knn_degenerate_sizes(built_lib):
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.zeros(0, 3, device="cuda:0")).shape == (0,)
    one = distCUDA2(torch.zeros(1, 3, device="cuda:0"))
    assert one.shape == (1,) and float(one[0]) > 1e30          # no neighbours: FLT_MAX stand-ins
    same = distCUDA2(torch.ones(100, 3, device="cuda:0"))
    assert float(same.abs().max()) == 0.0
