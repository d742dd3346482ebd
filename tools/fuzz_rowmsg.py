"""One-off widening of tests/test_exchange_rows.py on the GPU box: the row-message kernels (gsr_rowmsg_pack / _pack_slices / _reduce /
_apply / _apply_slices) on random ROW-MAJOR and split row sets -- row widths 1 ... 1024 (the F <= 64 several-rows-per-instruction map
and the F > 64 chunked map), 1 ... 16 messages (both template instances), row counts around the 64-row word boundaries, densities 0 ...
1 -- against the rank-ordered torch adds. usage: python tools/fuzz_rowmsg.py [n_seeds] [first_seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamscene_amd import _lib, multiview  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()


def one(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 1000, 4099, 20000, 70001]))
    # (4 ... 256: one dwordx4 chunk per lane, 64 / chunks rows per instruction; 257 ... 1024: more than 64 chunks per row, walked in
    #  groups of 64 lanes; 1 ... 3: the one-float-per-lane instances)
    F = int(rng.choice([1, 2, 3, 4, 5, 7, 14, 23, 38, 59, 63, 64, 65, 100, 128, 200, 255, 256, 257, 300, 513, 1024]))
    if F > 256:
        P = min(P, 4099)
    W = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16]))
    frac = float(rng.choice([0.0, 0.01, 0.1, 0.3, 0.9, 1.0]))
    split = bool(rng.integers(0, 2)) and F >= 3          # the row set as two regions with padding between rows
    g = torch.Generator().manual_seed(seed)
    dr = multiview._DeviceRows(dev)
    ranks = []
    for r in range(W):
        mask = torch.rand(P, generator=g) < frac
        dense = torch.randn((P, F), generator=g) * torch.exp(3 * torch.randn((P, 1), generator=g))
        dense[~mask] = 0
        if split:
            f1 = F // 3 + 1
            a = torch.zeros((P, f1 + 2)); b = torch.zeros((P, F - f1))
            a[:, :f1] = dense[:, :f1]; b[:] = dense[:, f1:]
            a, b = a.to(dev), b.to(dev)
            rs = dr.rowset([(a, f1, f1 + 2), (b, F - f1, F - f1)], P)
            read = (lambda a=a, b=b, f1=f1: torch.cat([a[:, :f1], b], dim=1))
        else:
            a = dense.to(dev)
            rs = dr.rowset([(a, F, F)], P)
            read = (lambda a=a: a.clone())
        bits = torch.zeros(((P + 63) // 64) * 64, dtype=torch.int64)
        bits[:P] = mask.to(torch.int64)
        words = (bits.view(-1, 64) << torch.arange(64, dtype=torch.int64)).sum(1).to(dev)
        ranks.append((rs, words, mask, dense, read))
    # the ranks' rows added in rank order: the first holder's row as it is, every later holder's added to the running sum
    ref = torch.zeros((P, F))
    first = torch.ones(P, dtype=torch.bool)
    for _, _, mask, dense, _ in ranks:
        new = mask & first
        ref[new] = dense[new]
        old = mask & ~first
        ref[old] = ref[old] + dense[old]
        first &= ~mask
    counts = [int(m.sum()) for _, _, m, _, _ in ranks]
    union = torch.stack([m for _, _, m, _, _ in ranks]).any(0)
    # ---- rows format
    cap = (max(counts) + 1023) // 1024 * 1024 + 1024
    ms = multiview._RowMessages(dev)
    msg, allm, nbytes = ms.buffers(P, F, W, cap)
    for r, (rs, words, _, _, _) in enumerate(ranks):
        ms.pack(rs, words, cap)
        allm[r * nbytes:(r + 1) * nbytes].copy_(msg)
    fails = []
    for r in (0, W - 1):
        rs, words, mask, dense, read = ranks[r]
        before = read()
        ms.apply(rs, W, cap)
        ok, worst = ms.result()
        got = read().cpu()
        want = torch.where(union.unsqueeze(1), ref, before.cpu())
        if not (ok and worst == max(counts) and torch.equal(got, want)):
            fails.append(("rows", r, ok, worst, max(counts), float((got - want).abs().max())))
    # ---- sparse_rs format on fresh copies of the ranks' rows (apply above overwrote ranks 0 and W-1)
    for r in (0, W - 1):
        rs, words, mask, dense, read = ranks[r]
        for k in range(rs.n_regions):
            t = rs._keep[k]
            f0 = sum(int(rs.regions[j].width) for j in range(k))
            t[:, :int(rs.regions[k].width)].copy_(dense[:, f0:f0 + int(rs.regions[k].width)].to(dev))
    per = ((P + W - 1) // W + 63) // 64 * 64
    c1 = max([int(m[o * per:(o + 1) * per].sum()) for _, _, m, _, _ in ranks for o in range(W)] + [0])
    c2 = max([int(union[o * per:(o + 1) * per].sum()) for o in range(W)] + [0])
    cap1, cap2 = (c1 + 511) // 512 * 512 + 512, (c2 + 511) // 512 * 512 + 512
    mss = [multiview._RowMessages(dev) for _ in range(W)]
    for r, (rs, words, _, _, _) in enumerate(ranks):
        mss[r].slice_buffers(P, F, W, per, cap1, cap2)
        mss[r].pack_slices(rs, words, W, per, cap1)
    n1, n2 = mss[0].n1, mss[0].n2
    for o in range(W):
        for r in range(W):
            mss[o].recv1[r * n1:(r + 1) * n1].copy_(mss[r].send1[o * n1:(o + 1) * n1])
        mss[o].reduce_owned(max(0, min(per, P - o * per)), per, F, W, cap1, cap2)
    for r in (0, W - 1):
        for o in range(W):
            mss[r].all2[o * n2:(o + 1) * n2].copy_(mss[o].own2)
        rs, words, mask, dense, read = ranks[r]
        before = read()
        mss[r].apply_slices(rs, W, per, cap2)
        ok, worst = mss[r].result()
        got = read().cpu()
        want = torch.where(union.unsqueeze(1), ref, before.cpu())
        if not (ok and worst == c2 and mss[r].worst_in == c1 and torch.equal(got, want)):
            fails.append(("sparse_rs", r, ok, worst, c2, mss[r].worst_in, c1, float((got - want).abs().max())))
    return dict(seed=seed, P=P, F=F, W=W, frac=frac, split=split), fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n):
        cfg, fails = one(seed)
        if fails:
            bad += 1
            print("FAIL", cfg, fails, flush=True)
    print(f"fuzz_rowmsg: {n - bad} of {n} configurations equal to the rank-ordered torch adds")
