"""-m gpu: the device side of the sparse gradient-exchange formats (csrc/exchange.hip: gsr_rows_pack / gsr_rows_unpack) against
the torch index arithmetic it replaces (multiview.GradExchange with the device helper switched off) -- bit for bit: packing
copies, unpacking adds one value per element. The wire formats themselves (every format, replicas identical, sums equal to
the single-process sum over the views: training/object_trainer.py:302-382) are tests/test_multiview_gloo.py (CPU) and
tests/test_multirank_gpu.py (two ranks, device tensors, these kernels underneath)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _arena(P, K, seed, frac):
    from dreamscene_amd import multiview
    a = multiview.GradArena(P, K, torch.device(DEV))
    g = torch.Generator(device="cpu").manual_seed(seed)
    rows = torch.rand(P, generator=g) < frac
    a.flat.copy_(torch.randn(a.flat.shape, generator=g).to(DEV))
    for v in a.views.values():                      # unreached rows are exactly zero, as K8 leaves them
        v[~rows.to(DEV)] = 0
    bits = torch.zeros(((P + 63) // 64) * 64, dtype=torch.int64)
    bits[:P] = rows.to(torch.int64)
    words = (bits.view(-1, 64) << torch.arange(64, dtype=torch.int64)).sum(1)      # (two's complement: bit 63 wraps correctly)
    a.reached.copy_(words.to(DEV))
    a.reached_valid = True
    return a, rows


@pytest.mark.parametrize("P,K,D,frac", [(1000, 16, 3, 0.2), (64 * 37, 16, 1, 0.5), (4099, 4, 1, 0.03), (513, 16, 0, 1.0),
                                        (300, 9, 2, 0.0), (200_000, 16, 3, 0.16)])
def test_pack_and_unpack_equal_the_torch_path(built_lib, P, K, D, frac):
    from dreamscene_amd import multiview
    a, rows = _arena(P, K, 7, frac)
    ex = multiview.GradExchange(a, sh_degree=D, mode="rows")
    assert ex._dev_rows is not None
    idx, msg = ex._message()                        # HIP: bitmap -> ascending indices + gathered rows
    dev_helper, ex._dev_rows = ex._dev_rows, None
    ref_idx = ex.nonzero_rows()                     # torch: the same bitmap expanded
    ref_msg = ex._rows_of(ref_idx) if ref_idx.numel() else torch.zeros((0, ex.row_floats), device=DEV)
    assert idx.dtype == torch.int32 and torch.equal(idx.to(torch.int64), ref_idx)
    assert int(idx.numel()) == int(rows.sum())
    assert torch.equal(msg, ref_msg)
    # add a message (rank-order accumulation) and store one (disjoint owners): torch reference first, then the kernels
    g = torch.Generator(device="cpu").manual_seed(11)
    other = torch.randn(msg.shape, generator=g).to(DEV)
    base = a.flat.clone()
    ex._add_rows(ref_idx, other)
    want_add = a.flat.clone()
    a.flat.copy_(base)
    ex._set_rows(ref_idx, other)
    want_set = a.flat.clone()
    ex._dev_rows = dev_helper
    a.flat.copy_(base)
    ex._add_rows(idx, other)
    assert torch.equal(a.flat, want_add)
    a.flat.copy_(base)
    ex._set_rows(idx, other)
    assert torch.equal(a.flat, want_set)
    # SH columns beyond the active degree are never touched
    nb = (D + 1) ** 2
    if nb < K:
        a.flat.copy_(base)
        ex._set_rows(idx, torch.full_like(other, 5.0))
        assert torch.equal(a.views["shs"][:, nb:, :], base[ex._offset_of_shs():].view(P, K, 3)[:, nb:, :])


def test_pack_retries_when_the_message_outgrows_its_buffers(built_lib):
    from dreamscene_amd import multiview
    a, rows = _arena(50_000, 16, 3, 0.02)
    ex = multiview.GradExchange(a, sh_degree=3, mode="rows")
    idx0, _ = ex._message()
    a2, rows2 = _arena(50_000, 16, 5, 0.9)          # far more rows than the buffers of the first call hold
    ex.arena = a2
    idx, msg = ex._message()
    assert int(idx.numel()) == int(rows2.sum()) > 10 * int(idx0.numel())
    ex._dev_rows = None
    assert torch.equal(msg, ex._rows_of(ex.nonzero_rows()))


def test_owner_side_of_sparse_rs_on_a_row_major_slice(built_lib):
    """unpack into a row-major buffer with a row base and a touched bitmap, then pack the touched rows again: the owner
    side of the sparse reduce-scatter."""
    from dreamscene_amd import multiview
    F, per, lo = 23, 1000, 5000
    dr = multiview._DeviceRows(torch.device(DEV))
    mine = torch.zeros((per, F), device=DEV)
    rs = dr.rowset([(mine, F, F)], per)
    touched = torch.zeros((per + 63) // 64, dtype=torch.int64, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    ref = torch.zeros((per, F))
    seen = torch.zeros(per, dtype=torch.bool)
    for src in range(4):
        li = torch.randperm(per, generator=g)[:150].sort().values
        rows = torch.randn((150, F), generator=g)
        dr.unpack(rs, (li + lo).to(torch.int32).to(DEV), rows.to(DEV), mode=0, row_base=lo, touched=touched)
        ref.index_add_(0, li, rows)
        seen[li] = True
    idx, got = multiview._DeviceRows(torch.device(DEV)).pack(rs, touched, F, 0)
    assert torch.equal(idx.cpu().to(torch.int64), torch.nonzero(seen).reshape(-1))
    assert torch.equal(got.cpu(), ref[seen])


@pytest.mark.parametrize("W,per", [(8, 3_687_500), (2, 1000), (5, 1003), (1, 77), (3, 4)])
def test_slice_sum_of_the_direct_format_equals_the_rank_order_adds(built_lib, W, per):
    """multiview._sum_slices on a GPU (gsr_sum_slices) against the W - 1 in-place torch adds it replaces: same association
    (((s0 + s1) + s2) + ...), hence the same bits -- every rank applies it to the slice it owns."""
    from dreamscene_amd import multiview
    g = torch.Generator(device="cpu").manual_seed(W * 1000 + per % 997)
    recv = (torch.randn(W * per, generator=g) * torch.exp(4.0 * torch.randn(W * per, generator=g))).to(DEV)
    want = recv.view(W, per)[0].clone()
    for r in range(1, W):
        want.add_(recv.view(W, per)[r])
    got = multiview._sum_slices(recv, W, per)
    assert got.data_ptr() != recv.data_ptr() and torch.equal(got, want)
    # misaligned slices take the scalar form
    off = recv[1:1 + (W * per - 1) // W * W]
    per2 = off.numel() // W
    if per2:
        off = off.contiguous() if not off.is_contiguous() else off
        want2 = off.view(W, per2)[0].clone()
        for r in range(1, W):
            want2.add_(off.view(W, per2)[r])
        assert torch.equal(multiview._sum_slices(off, W, per2), want2)
