"""CPU, world_size 2, gloo: the view-sharded data-parallel path (dreamscene_amd/multiview.py). Each rank renders its
share of the views forward+backward and the per-view parameter gradients are summed by ONE in-place all-reduce of
the packed arena; the result must equal the sequential accumulation over all views that the reference performs
(training/object_trainer.py:302-382). The renderer is the CPU oracle here (test infrastructure): the HIP rasterizer
cannot run without a GPU, and the data-parallel logic is renderer-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import small_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_view_fn():
    from dreamscene_amd import synth
    from oracle import torch_oracle as TO

    def rasterize_view(params, cam, grad_out, upstream):
        dt = torch.float64
        t = {k: v.detach().to(dt).requires_grad_(True) for k, v in params.items()}
        m2d = torch.zeros(t["means3D"].shape[0], 3, dtype=dt, requires_grad=True)
        s = TO.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, torch.ones(3, dtype=dt), 1.0,
                        torch.tensor(cam.world_view_transform, dtype=dt), torch.tensor(cam.full_proj_transform, dtype=dt),
                        3, torch.tensor(cam.camera_center, dtype=dt), False, False)
        img, radii, da = TO.rasterize(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                      rotations=t["rotations"], settings=s)
        gi, gda = upstream
        ((img * torch.tensor(gi, dtype=dt)).sum() + (da * torch.tensor(gda, dtype=dt)).sum()).backward()
        for k in grad_out:
            grad_out[k].copy_(t[k].grad.to(grad_out[k].dtype))
        return dict(means2D_grad=m2d.grad.to(torch.float32), radii=radii)
    return rasterize_view


def _scene(n_views):
    from dreamscene_amd import synth
    g, _ = small_scene(P=200, H=48, W=48, K=16, seed=5)
    cams = synth.object_cameras(n_views, 48, 48, radius=3.0)
    ups = [synth.upstream_grads(48, 48, i) for i in range(n_views)]
    return g, cams, ups


def _worker(rank, world, port, n_views, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dreamscene_amd import multiview
    g, cams, ups = _scene(n_views)
    params = {k: torch.tensor(v) for k, v in g.items()}
    arena = multiview.GradArena(200, 16, "cpu")
    outs = multiview.render_views_data_parallel(_oracle_view_fn(), params, cams, ups, arena)
    # per-view densification statistics, reduced so every replica decides identically
    acc = None
    for o in outs:
        st = multiview.reduce_view_stats(o["means2D_grad"], o["radii"])
        acc = st if acc is None else (acc[0] + st[0], acc[1] + st[1], torch.maximum(acc[2], st[2]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=arena.flat.numpy(), norm=acc[0].numpy(), vis=acc[1].numpy(),
             maxr=acc[2].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [2, 4])
def test_two_rank_allreduce_equals_sequential_accumulation(tmp_path, n_views):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_views, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]), "ranks disagree after the all-reduce"
    # sequential single-process accumulation over the same views
    from dreamscene_amd import multiview
    g, cams, ups = _scene(n_views)
    params = {k: torch.tensor(v) for k, v in g.items()}
    fn = _oracle_view_fn()
    seq = multiview.GradArena(200, 16, "cpu")
    tmp = multiview.GradArena(200, 16, "cpu")
    norm = torch.zeros(200)
    vis = torch.zeros(200)
    maxr = torch.zeros(200, dtype=torch.int32)
    for cam, up in zip(cams, ups):
        o = fn(params, cam, tmp.views, up)
        seq.flat += tmp.flat
        norm += torch.norm(o["means2D_grad"][:, :2], dim=-1)
        vis += (o["radii"] > 0).float()
        maxr = torch.maximum(maxr, o["radii"].to(torch.int32))
    ref = seq.flat.numpy()
    np.testing.assert_allclose(r0["flat"], ref, rtol=0, atol=1e-6 * max(1.0, float(np.abs(ref).max())))
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(r0["norm"], norm.numpy(), atol=1e-6)
    assert np.array_equal(r0["vis"], vis.numpy()) and np.array_equal(r0["maxr"], maxr.numpy())
    assert np.array_equal(r0["norm"], r1["norm"])
