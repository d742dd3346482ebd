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

from tests.util import rel_scale, small_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_view_fn():
    from dreamscene_amd import synth
    from oracle import torch_oracle as TO

    def rasterize_view(params, cam, grad_out, upstream, accumulate=False):
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
        for k in grad_out:          # (K8's accumulate mode: add this view to what the arena holds)
            gk = t[k].grad.to(grad_out[k].dtype).reshape(grad_out[k].shape)
            grad_out[k].add_(gk) if accumulate else grad_out[k].copy_(gk)
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
        o = fn(params, cam, tmp.views, up, False)
        seq.flat += tmp.flat
        norm += torch.norm(o["means2D_grad"][:, :2], dim=-1)
        vis += (o["radii"] > 0).float()
        maxr = torch.maximum(maxr, o["radii"].to(torch.int32))
    ref = seq.flat.numpy()
    np.testing.assert_allclose(r0["flat"], ref, rtol=0, atol=1e-6 * rel_scale(ref))
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(r0["norm"], norm.numpy(), atol=1e-6)
    assert np.array_equal(r0["vis"], vis.numpy()) and np.array_equal(r0["maxr"], maxr.numpy())
    assert np.array_equal(r0["norm"], r1["norm"])


# ------------------------------------------------------------------------------------------------ GradExchange
def _fake_rank_grads(rank, P, K, D, row_frac, seed=0):
    """What a rank's arena looks like after its views' backward: zeros in the SH columns beyond the active degree and in
    the rows of Gaussians nothing composited; everything else arbitrary."""
    from dreamscene_amd import multiview
    rng = np.random.default_rng(100 * seed + rank)
    a = multiview.GradArena(P, K, "cpu")
    nb = (D + 1) ** 2
    rows = rng.random(P) < row_frac
    for name, t in a.views.items():
        x = rng.normal(size=tuple(t.shape)).astype(np.float32)
        if name == "shs":
            x[:, nb:, :] = 0.0
        x[~rows] = 0.0
        t.copy_(torch.tensor(x))
    return a


def _exchange_worker(rank, world, port, P, K, D, row_frac, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreamscene_amd import multiview
    a = _fake_rank_grads(rank, P, K, D, row_frac)
    ex = multiview.GradExchange(a, sh_degree=D, mode=mode)
    ex.reduce()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=a.flat.numpy(), fmt=ex.last["format"],
             nbytes=ex.last.get("bytes_per_rank", 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("D,row_frac,mode,expect", [(3, 1.0, "auto", "dense"), (0, 1.0, "auto", "dense"),
                                                    (1, 0.9, "dense", "dense"), (3, 0.05, "auto", "rows"),
                                                    (0, 0.05, "rows", "rows"), (2, 0.0, "auto", "rows"),
                                                    (3, 1.0, "direct", "direct"), (1, 0.5, "direct", "direct"),
                                                    (3, 0.1, "sparse_rs", "sparse_rs"), (0, 0.6, "sparse_rs", "sparse_rs"),
                                                    (2, 0.0, "sparse_rs", "sparse_rs")])
def test_grad_exchange_formats_equal_plain_sum(tmp_path, D, row_frac, mode, expect):
    """Active-degree columns only / non-zero rows only on the wire: the arena ends up with exactly the sum a plain dense
    all-reduce of everything would give, identical on both ranks."""
    world, P, K = 2, 257, 16
    port = _free_port()
    mp.spawn(_exchange_worker, args=(world, port, P, K, D, row_frac, mode, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert str(r0["fmt"]) == expect == str(r1["fmt"])
    assert np.array_equal(r0["flat"], r1["flat"]), "replicas must hold bit-identical sums"
    ref = _fake_rank_grads(0, P, K, D, row_frac).flat.numpy() + _fake_rank_grads(1, P, K, D, row_frac).flat.numpy()
    assert np.array_equal(r0["flat"], ref), np.abs(r0["flat"] - ref).max()      # two addends: the sum is exact either way
    F = 11 + 3 * (D + 1) ** 2
    dense_full = 2 * (world - 1) / world * 4 * (11 + 3 * K) * P
    if expect in ("dense", "direct"):
        assert int(r0["nbytes"]) <= 2 * (world - 1) / world * 4 * (F * P + 16)       # active columns only (+ alignment pad)
    else:
        assert int(r0["nbytes"]) < 0.25 * dense_full or row_frac == 0.0


def _adam_ref(p, g, m, v, step, lr, betas, eps):
    """torch.optim.Adam's single-tensor update on flat shards (test double of gsr_adam_step)."""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(eps), value=-1.0) if isinstance(lr, float) and lr == 1.0 else \
        p.sub_(lr / bc1 * m / ((v.sqrt() / (bc2 ** 0.5)) + eps))


def _rs_worker(rank, world, port, P, K, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreamscene_amd import multiview
    a = _fake_rank_grads(rank, P, K, 3, 1.0)
    n = a.flat.numel()
    params = torch.tensor(np.random.default_rng(7).normal(size=n).astype(np.float32))
    m, v = torch.zeros(n), torch.zeros(n)
    lr = torch.tensor(np.random.default_rng(8).uniform(1e-3, 1e-2, size=n).astype(np.float32))
    ex = multiview.GradExchange(a, sh_degree=3)
    ex.reduce_scatter_adam(params, m, v, 1, lambda lo, hi: lr[lo:hi], adam=_adam_ref)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=params.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_sharded_adam_all_gather(tmp_path):
    """Every rank updates its 1/W of the flat parameter arena from the reduce-scattered gradient sum, then the updated
    parameters are all-gathered: same parameters everywhere, equal to one replicated Adam step on the summed gradient."""
    world, P, K = 2, 256, 16
    port = _free_port()
    mp.spawn(_rs_worker, args=(world, port, P, K, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["params"], r1["params"])
    g = _fake_rank_grads(0, P, K, 3, 1.0).flat + _fake_rank_grads(1, P, K, 3, 1.0).flat
    n = g.numel()
    params = torch.tensor(np.random.default_rng(7).normal(size=n).astype(np.float32))
    lr = torch.tensor(np.random.default_rng(8).uniform(1e-3, 1e-2, size=n).astype(np.float32))
    _adam_ref(params, g, torch.zeros(n), torch.zeros(n), 1, lr, (0.9, 0.999), 1e-15)
    assert np.array_equal(r0["params"], params.numpy())


@pytest.mark.parametrize("mode,D,row_frac", [("dense", 3, 1.0), ("direct", 3, 0.4), ("direct", 1, 1.0), ("rows", 2, 0.1),
                                             ("sparse_rs", 3, 0.16), ("sparse_rs", 0, 0.5)])
def test_grad_exchange_formats_at_world_size_8(tmp_path, mode, D, row_frac):
    """The node the driver scales to: 8 ranks. P = 257 is no multiple of 8 (`direct` pads its slices, the last owner of
    `sparse_rs` holds a short range). Replicas bit-identical; the formats that add in RANK ORDER (direct, rows, sparse_rs)
    give exactly ((g0 + g1) + g2) + ... in fp32; the ring of `dense` associates as gloo pleases (1e-6 of the float64 sum)."""
    world, P, K = 8, 257, 16
    port = _free_port()
    mp.spawn(_exchange_worker, args=(world, port, P, K, D, row_frac, mode, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert all(str(r["fmt"]) == mode for r in rs)
    for r in rs[1:]:
        assert np.array_equal(rs[0]["flat"], r["flat"]), "replicas must hold bit-identical sums"
    parts = [_fake_rank_grads(r, P, K, D, row_frac).flat.numpy() for r in range(world)]
    exact = np.sum([p.astype(np.float64) for p in parts], axis=0)
    np.testing.assert_allclose(rs[0]["flat"], exact, rtol=0, atol=2e-6 * rel_scale(exact))
    if mode != "dense":
        ordered = parts[0].copy()
        for p in parts[1:]:
            ordered += p
        assert np.array_equal(rs[0]["flat"], ordered), np.abs(rs[0]["flat"] - ordered).max()
