"""How many 8x8 blocks of a 16x16 tile does a (splat, tile) pair of K7's traversal really hit? -- the number that decides a
"whole tile per wave" K7 (one wave walks the four blocks of a tile, keeps the ten per-splat sums in registers across them and
reduces + commits ONCE per (splat, tile) instead of once per (splat, block)). Computed on the CPU from the
scalar C oracle's lists (oracle/gsr_oracle.c: point_list, ranges, n_contrib, the projected splat table) with the kernels' own
gates at the pixel centres: block (tile, b) is HIT by list entry e when some pixel of the block has e < n_contrib (still alive),
power <= 0 and alpha >= 1/255. Instruction model from render.hip's ISA (DESIGN.md "Issue costs"): per hit block ~65 VALU for the
per-pixel work + 27 for reduce10 / convert / address / atomic today; whole-tile: 68 per hit block (3 more: accumulate instead of
assign), 10 to clear the sums + 27 once per (splat, tile).   usage: python tools/k7_block_stats.py [P] [res] [--init-opacity] [--indoor]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import synth  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(args[0]) if args else 500_000
res = int(args[1]) if len(args) > 1 else 1024
init = "--init-opacity" in sys.argv
indoor = "--indoor" in sys.argv
K, D = (4, 1) if indoor else (16, 3)
g = synth.g_indoor(seed=0, per_wall=P // 5, K=K) if indoor else synth.g_object(P, seed=0, K=K, init_opacity=init)
cam = (synth.indoor_cameras if indoor else synth.object_cameras)(8, res, res)[1 if indoor else 0]
CO.build()
Pn = g["means3D"].shape[0]
v = CO.make_view(Pn, K, D, res, res, cam.tanfovx, cam.tanfovy, [1, 1, 1], cam.world_view_transform, cam.full_proj_transform,
                 cam.camera_center)
f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
gx = (res + 15) // 16
hist = np.zeros(5, np.int64)           # (splat, tile) pairs by number of hit blocks (index 0: evaluated, nothing hit)
hit_blocks = 0
xy, co, ncon = f["xy"], f["conic_opacity"], f["n_contrib"].astype(np.int64)
yy, xx = np.mgrid[0:16, 0:16]
blk = ((yy >> 3) * 2 + (xx >> 3)).reshape(-1)            # block index of each pixel of a tile
for t in range(f["ranges"].shape[0]):
    r0, r1 = int(f["ranges"][t, 0]), int(f["ranges"][t, 1])
    if r1 <= r0:
        continue
    ty, tx = divmod(t, gx)
    nc = ncon[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16].reshape(-1)
    depth = int(nc.max())
    if depth == 0:
        continue
    ids = f["point_list"][r0:r0 + depth].astype(np.int64)
    px = (tx * 16 + xx).reshape(-1).astype(np.float32)
    py = (ty * 16 + yy).reshape(-1).astype(np.float32)
    dx = xy[ids, 0][:, None] - px[None, :]
    dy = xy[ids, 1][:, None] - py[None, :]
    A, B, C_, op = co[ids, 0][:, None], co[ids, 1][:, None], co[ids, 2][:, None], co[ids, 3][:, None]
    power = -0.5 * (A * dx * dx + C_ * dy * dy) - B * dx * dy
    alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 0.0)))
    live = np.arange(depth)[:, None] < nc[None, :]
    hit = live & (power <= 0) & (alpha >= 1.0 / 255.0)                    # [entries, 256 pixels]
    hb = np.stack([hit[:, blk == b].any(axis=1) for b in range(4)], axis=1)   # [entries, 4 blocks]
    nb = hb.sum(axis=1)
    hist += np.bincount(nb, minlength=5)[:5]
    hit_blocks += int(nb.sum())
pairs = int(hist[1:].sum())
now = 92 * hit_blocks
tile = sum(int(hist[b]) * (37 + 68 * b) for b in range(1, 5))
print(f"{'indoor' if indoor else 'object'} P={Pn} res={res} init={init}: (splat, tile) pairs with a hit {pairs}, hit blocks {hit_blocks} "
      f"(mean {hit_blocks / max(pairs, 1):.2f} per pair); by blocks hit 1/2/3/4: {hist[1:].tolist()} "
      f"({[round(100 * int(x) / max(pairs, 1), 1) for x in hist[1:]]} %)")
print(f"VALU model: per-block K7 {now / 1e6:.1f} M, whole-tile K7 {tile / 1e6:.1f} M -> x{tile / max(now, 1):.3f}")
