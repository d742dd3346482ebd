"""-m gpu: the pair count N of a view reaches the host EARLY (include/gsrast.h, GSR_N_PENDING; radix_sort.h, kOsEarlyN): the first
workgroup of the depth sort's first pass stores the sum of the tile rectangles of the visible Gaussians into the caller's
page-locked word, and nothing stores it a second time. Checked: the early word equals the exact count of the two-phase
forward (= the oracle's N, tests/test_gpu_parity.py) for single views, batches and culled-everything inputs; a call re-arms
the word at once without a stale store landing in it; grids beyond 256 x 256 tiles (no early store) still deliver."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.util import settings_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(P, res, seed=11, cams=4):
    from dreamscene_amd import synth
    g = synth.g_object(P, seed=seed, K=16)
    cs = synth.object_cameras(8, res, res)[:cams]
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    return t, cs


def _exact_N(t, s):
    from dreamscene_amd import rasterizer as R
    out, _ = R.rasterize_forward_raw(s, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None, mode="sync")
    return int(out["N"])


def test_early_word_equals_the_exact_count(built_lib):
    from dreamscene_amd import rasterizer as R
    t, cams = _scene(60_000, 384)
    sets = [settings_for(c, np.ones(3, np.float32), 3, DEV) for c in cams]
    exact = [_exact_N(t, s) for s in sets]
    assert min(exact) > 0
    ws = R._workspace(torch.device(DEV), torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
    for rep in range(3):               # auto mode from the second call on: capacity mode, the host polls the early word
        for s, n in zip(sets, exact):
            out, _ = R.rasterize_forward_raw(s, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
            assert int(out["N"]) == n
            # nothing stores the word a second time: after the stream has drained it still holds this call's count
            torch.cuda.synchronize()
            if rep >= 1:               # (the very first call of a size has no capacity yet and takes the exact two-phase forward)
                assert int(ws.n_pinned_np[0]) == n
    # every Gaussian culled (behind the camera): N = 0 arrives as a count, not as "pending"
    far = dict(t, means3D=t["means3D"] + torch.tensor([0.0, 0.0, 1e4], device=DEV))
    s0 = sets[0]
    for _ in range(2):
        out, _ = R.rasterize_forward_raw(s0, far["means3D"], far["opacities"], far["shs"], None, far["scales"], far["rotations"], None)
    n_far = int(out["N"])
    assert n_far == _exact_N(far, s0)


def test_back_to_back_calls_never_read_a_stale_count(built_lib):
    """Calls on one stream share the page-locked word: call j + 1 re-arms it while call j's kernels are still running. Alternating
    two cameras with different counts must return each camera's own count every time."""
    from dreamscene_amd import rasterizer as R
    t, cams = _scene(120_000, 512)
    sets = [settings_for(c, np.ones(3, np.float32), 3, DEV) for c in cams[:2]]
    # make the two counts clearly different: the second camera sees the object at half the size
    sets[1] = sets[1]._replace(tanfovx=sets[1].tanfovx * 2.0, tanfovy=sets[1].tanfovy * 2.0)
    exact = [_exact_N(t, s) for s in sets]
    assert exact[0] != exact[1]
    for i in range(40):
        k = i & 1
        out, _ = R.rasterize_forward_raw(sets[k], t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None,
                                         want_aux=False)
        assert int(out["N"]) == exact[k], f"call {i}: count {int(out['N'])} of camera {k}, expected {exact[k]} (other: {exact[1 - k]})"
    torch.cuda.synchronize()


def test_batched_views_and_wide_grids(built_lib):
    from dreamscene_amd import rasterizer as R, synth
    from dreamscene_amd.views import rasterize_views_forward_raw
    t, cams = _scene(80_000, 320)
    sets = [settings_for(c, np.ones(3, np.float32), 3, DEV) for c in cams]
    exact = [_exact_N(t, s) for s in sets]
    for rep in range(3):
        res = rasterize_views_forward_raw(sets, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
        assert [int(o["N"]) for o, _ in res] == exact
    # a grid of 257 x 1 tiles: not the column path, no early store -- the count still arrives (copy behind the projection)
    g2, _ = _scene(5_000, 64, cams=1)
    c0 = synth.object_cameras(1, 64, 64)[0]
    wide = settings_for(c0, np.ones(3, np.float32), 3, DEV)._replace(image_height=16, image_width=4112)
    n_sync = _exact_N(g2, wide)
    for _ in range(3):
        out, _ = R.rasterize_forward_raw(wide, g2["means3D"], g2["opacities"], g2["shs"], None, g2["scales"], g2["rotations"], None)
        assert int(out["N"]) == n_sync
