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


@pytest.mark.parametrize("P,K,D,W,frac", [(1000, 16, 3, 3, 0.2), (64 * 37 + 5, 16, 1, 8, 0.4), (4099, 4, 1, 2, 0.03),
                                          (513, 16, 0, 5, 1.0), (300, 9, 2, 4, 0.0), (200_000, 16, 3, 8, 0.16)])
def test_row_messages_give_the_rank_ordered_sum(built_lib, P, K, D, W, frac):
    """gsr_rowmsg_pack / gsr_rowmsg_apply (the host-read-free `rows` exchange): W ranks' arenas packed into self-describing
    messages, applied in ONE launch -> every row of the union holds ((g_0 + g_1) + ...) over the ranks that sent it, the same
    bits the rank-by-rank torch adds give (what the sequential accumulation of training/object_trainer.py:302-382 becomes across
    ranks); rows nobody sent stay zero; a message that does not fit leaves the arena untouched and says so."""
    from dreamscene_amd import multiview
    arenas = [_arena(P, K, 100 + r, frac) for r in range(W)]
    ex = [multiview.GradExchange(a, sh_degree=D, mode="rows") for a, _ in arenas]
    nb = (D + 1) ** 2
    # reference: zero, then the ranks' rows added in rank order (torch index arithmetic, what the `rows` format always did)
    ref = multiview.GradArena(P, K, torch.device(DEV))
    ex_ref = multiview.GradExchange(ref, sh_degree=D, mode="rows")
    ex_ref._dev_rows = None
    for r in range(W):
        ex[r]._dev_rows, held = None, ex[r]._dev_rows
        idx = ex[r].nonzero_rows()
        if idx.numel():
            ex_ref._add_rows(idx, ex[r]._rows_of(idx))
        ex[r]._dev_rows = held
    counts = [int(rows.sum()) for _, rows in arenas]
    cap = (max(counts) + 1023) // 1024 * 1024 + 1024
    msgs = multiview._RowMessages(torch.device(DEV))
    F = ex[0].row_floats
    msg, allm, nbytes = msgs.buffers(P, F, W, cap)
    for r in range(W):
        msgs.pack(ex[r]._arena_rowset(), arenas[r][0].reached, cap)
        allm[r * nbytes:(r + 1) * nbytes].copy_(msg)
        hdr = msg[:16].view(torch.int32).cpu().tolist()
        assert hdr == [counts[r], cap, P, F], (hdr, counts[r])
    for r in (0, W - 1):                       # every rank ends with the same bits
        a = arenas[r][0]
        own = a.flat.clone()
        union_bits = torch.full_like(a.reached, -1)           # (every word must be stored, the empty ones too)
        msgs.apply(ex[r]._arena_rowset(), W, cap, touched=union_bits)
        ok, worst = msgs.result()
        assert ok and worst == max(counts)
        want = torch.zeros_like(a.reached)
        for q in range(W):
            want |= arenas[q][0].reached
        assert torch.equal(union_bits, want), f"rank {r}: the union bitmap apply leaves for the next backward (zero_outside)"
        for name in ("means3D", "scales", "rotations", "opacities"):
            assert torch.equal(a.views[name], ref.views[name]), f"rank {r}: {name}"
        assert torch.equal(a.views["shs"][:, :nb, :], ref.views["shs"][:, :nb, :]), f"rank {r}: active SH columns"
        if nb < K:                             # SH columns beyond the active degree are never touched
            assert torch.equal(a.views["shs"][:, nb:, :], own[ex[r]._offset_of_shs():].view(P, K, 3)[:, nb:, :])
        a.flat.copy_(own)
    # capacity too small: the header says so, nothing is applied
    if max(counts) > 1024:
        small = 1024
        msg, allm, nbytes = msgs.buffers(P, F, W, small)
        for r in range(W):
            msgs.pack(ex[r]._arena_rowset(), arenas[r][0].reached, small)
            allm[r * nbytes:(r + 1) * nbytes].copy_(msg)
        a = arenas[0][0]
        own = a.flat.clone()
        union_bits = torch.full_like(a.reached, -1)
        msgs.apply(ex[0]._arena_rowset(), W, small, touched=union_bits)
        ok, worst = msgs.result()
        assert not ok and worst == max(counts)
        torch.cuda.synchronize()
        assert torch.equal(a.flat, own) and bool((union_bits == -1).all())         # nothing applied: nothing stored


@pytest.mark.parametrize("P,K,D,W,frac", [(1000, 16, 3, 3, 0.3), (64 * 37 + 5, 16, 1, 8, 0.4), (4099, 4, 1, 2, 0.05),
                                          (513, 16, 0, 5, 1.0), (300, 9, 2, 4, 0.0), (200_000, 16, 3, 8, 0.16)])
def test_slice_messages_give_the_sparse_reduce_scatter(built_lib, P, K, D, W, frac):
    """The device form of `sparse_rs`, all W ranks emulated on one GPU: gsr_rowmsg_pack_slices (rank -> owner messages), the
    all-to-all as copies, gsr_rowmsg_reduce at every owner, the all-gather as copies, gsr_rowmsg_apply_slices on every rank -> the
    same bits as the rank-ordered torch adds of the `rows` format (training/object_trainer.py:302-382 across ranks); capacities
    too small in either phase: the status says so (which phase, how many rows) and no arena is touched."""
    from dreamscene_amd import multiview
    arenas = [_arena(P, K, 300 + r, frac) for r in range(W)]
    ex = [multiview.GradExchange(a, sh_degree=D, mode="sparse_rs") for a, _ in arenas]
    nb, F = (D + 1) ** 2, ex[0].row_floats
    ref = multiview.GradArena(P, K, torch.device(DEV))
    ex_ref = multiview.GradExchange(ref, sh_degree=D, mode="rows")
    ex_ref._dev_rows = None
    for r in range(W):
        ex[r]._dev_rows, held = None, ex[r]._dev_rows
        idx = ex[r].nonzero_rows()
        if idx.numel():
            ex_ref._add_rows(idx, ex[r]._rows_of(idx))
        ex[r]._dev_rows = held
    per = ex[0]._slice_rows(W)
    masks = [rows for _, rows in arenas]
    c1 = max([int(m[o * per:(o + 1) * per].sum()) for m in masks for o in range(W)] + [0])
    union = torch.stack(masks).any(0)
    c2 = max([int(union[o * per:(o + 1) * per].sum()) for o in range(W)] + [0])

    def run(cap1, cap2):
        ms = [multiview._RowMessages(torch.device(DEV)) for _ in range(W)]
        for r in range(W):
            ms[r].slice_buffers(P, F, W, per, cap1, cap2)
            ms[r].pack_slices(ex[r]._arena_rowset(), arenas[r][0].reached, W, per, cap1)
        n1, n2 = ms[0].n1, ms[0].n2
        for o in range(W):                      # all-to-all: owner o receives slice o of every rank, in rank order
            for r in range(W):
                ms[o].recv1[r * n1:(r + 1) * n1].copy_(ms[r].send1[o * n1:(o + 1) * n1])
        for o in range(W):
            ms[o].reduce_owned(max(0, min(per, P - o * per)), per, F, W, cap1, cap2)
        for r in range(W):                      # all-gather
            for o in range(W):
                ms[r].all2[o * n2:(o + 1) * n2].copy_(ms[o].own2)
        res = []
        want = torch.zeros_like(arenas[0][0].reached)
        for q in range(W):
            want |= arenas[q][0].reached
        for r in range(W):
            union_bits = torch.full_like(want, -1)
            ms[r].apply_slices(ex[r]._arena_rowset(), W, per, cap2, touched=union_bits)
            ok, worst = ms[r].result()
            res.append((ok, worst, ms[r].worst_in))
            torch.cuda.synchronize()
            # the owners' bitmaps side by side = the union over the ranks (stored only if the messages were applied)
            assert torch.equal(union_bits, want) if ok else bool((union_bits == -1).all()), f"rank {r}: union bitmap"
        return res

    own = [a.flat.clone() for a, _ in arenas]
    big1, big2 = (c1 + 511) // 512 * 512 + 512, (c2 + 511) // 512 * 512 + 512
    res = run(big1, big2)
    assert all(ok for ok, _, _ in res) and {w for _, w, _ in res} == {c2} and {wi for _, _, wi in res} == {c1}, (res, c1, c2)
    for r in range(W):
        a = arenas[r][0]
        for name in ("means3D", "scales", "rotations", "opacities"):
            assert torch.equal(a.views[name], ref.views[name]), f"rank {r}: {name}"
        assert torch.equal(a.views["shs"][:, :nb, :], ref.views["shs"][:, :nb, :]), f"rank {r}: active SH columns"
        if nb < K:
            assert torch.equal(a.views["shs"][:, nb:, :], own[r][ex[r]._offset_of_shs():].view(P, K, 3)[:, nb:, :])
        a.flat.copy_(own[r])
    if c1 > 512:                                 # first phase too small: every owner poisons its message, nobody applies
        res = run(512, big2)
        assert not any(ok for ok, _, _ in res) and all(wi == c1 for _, _, wi in res), res
        assert all(torch.equal(arenas[r][0].flat, own[r]) for r in range(W))
    if c2 > 512:                                 # second phase too small
        res = run(big1, 512)
        assert not any(ok for ok, _, _ in res) and all(w == c2 and wi == c1 for _, w, wi in res), res
        assert all(torch.equal(arenas[r][0].flat, own[r]) for r in range(W))


@pytest.mark.parametrize("F", [1, 3, 4, 5, 61, 255, 300, 1024])
def test_row_messages_of_any_width(built_lib, F):
    """The message kernels on a plain row-major set: one float per lane for rows of fewer than 4 floats, one dwordx4 per lane from 4
    on (the last chunk of a row overlapping its neighbour when F is no multiple of 4), rows of more than 256 floats walked in groups of
    64 chunks -- against the rank-ordered torch adds (tools/fuzz_rowmsg.py is the wide version of this test)."""
    from dreamscene_amd import multiview
    dev = torch.device(DEV)
    P, W, frac = 64 * 3 + 17, 3, 0.4
    g = torch.Generator().manual_seed(F)
    dr = multiview._DeviceRows(dev)
    ranks, ref, first = [], torch.zeros((P, F)), torch.ones(P, dtype=torch.bool)
    for r in range(W):
        mask = torch.rand(P, generator=g) < frac
        dense = torch.randn((P, F), generator=g)
        dense[~mask] = 0
        a = dense.to(dev)
        bits = torch.zeros(((P + 63) // 64) * 64, dtype=torch.int64)
        bits[:P] = mask.to(torch.int64)
        words = (bits.view(-1, 64) << torch.arange(64, dtype=torch.int64)).sum(1).to(dev)
        ranks.append((dr.rowset([(a, F, F)], P), words, a))
        new, old = mask & first, mask & ~first
        ref[new] = dense[new]
        ref[old] = ref[old] + dense[old]
        first &= ~mask
    cap = 1024
    ms = multiview._RowMessages(dev)
    msg, allm, nbytes = ms.buffers(P, F, W, cap)
    for r, (rs, words, _) in enumerate(ranks):
        ms.pack(rs, words, cap)
        allm[r * nbytes:(r + 1) * nbytes].copy_(msg)
    rs, _, a = ranks[0]
    ms.apply(rs, W, cap)
    ok, _ = ms.result()
    torch.cuda.synchronize()
    assert ok and torch.equal(a.cpu(), ref)
