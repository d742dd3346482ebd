"""View-level data parallelism for the rasterizer: one view per GPU, Gaussian gradients summed over ranks.

The reference renders C_batch_size views one after another against the same parameters and sums their
gradients in a single backward (training/object_trainer.py:302-382, training/scene_trainer.py:801-881); it has
no distributed code. Views are independent, so the path shards with no data-path collective in
forward/backward; the only exchange is the sum of per-view parameter gradients, one in-place all-reduce per step
over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

GradArena: every parameter gradient of one view is written by the HIP backward straight into ONE flat fp32
buffer (planar layout [means3D P*3 | scales P*3 | rotations P*4 | opacities P | shs P*K*3]); the all-reduce runs
in place on that buffer and the tensors handed to autograd are views of it -- no pack / unpack copies.
Per-view densification statistics (norm of means2D.grad, radii>0, max radii; gs_renderer.py:1034-1065) are not
linear in the view, so they are reduced separately (reduce_view_stats).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

FIELDS = (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacities", 1), ("shs", None))


class GradArena:
    def __init__(self, P: int, K: int, device, dtype=torch.float32):
        self.P, self.K = int(P), int(K)
        sizes = [(n, P * (3 * K if w is None else w)) for n, w in FIELDS]
        # keep every region 16-byte aligned (the C ABI requires it for shs / rotations)
        offs, total = {}, 0
        for n, sz in sizes:
            total = (total + 3) & ~3
            offs[n] = (total, sz)
            total += sz
        self.flat = torch.zeros(total, dtype=dtype, device=device)
        shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
        self.views: Dict[str, torch.Tensor] = {n: self.flat[o:o + sz].view(shapes[n]) for n, (o, sz) in offs.items()}
        # one bit per Gaussian: "may hold a non-zero row" -- written by K8 itself (GsrGrads.reached_mask: it classifies the
        # Gaussians K7 reached anyway), so that an exchange need not re-scan the 236 B / Gaussian arena for its non-zero
        # rows. Valid only while the HIP backward is what fills the arena: anything else that writes `flat` calls touch().
        self.reached = torch.zeros((P + 63) // 64, dtype=torch.int64, device=device)
        self.reached_valid = False
        # "every row outside `reached` is zero" (GsrGrads.zero_outside): true of a fresh arena (all zero, empty bitmap) and kept
        # by every HIP backward that overwrites the arena; K8 then clears only the rows the bitmap names instead of everything
        # nothing reached (84 % of the rows at C3). A torch op that writes `flat` (or a view of it) in place shows in the
        # version counter; anything that writes it through raw pointers calls touch().
        self.zero_outside_reached = True
        self._maintained = frozenset(n for n, _ in FIELDS)     # the regions the last overwriting backward wrote (all: a fresh arena)
        self._k8_version = self.flat._version
        self._mask_owner = None       # whose per-view rows the bitmap describes as well (rasterize_backward_views_raw, `persistent`)

    def touch(self) -> None:
        """The arena was written by something other than the HIP backward: the reached-row bitmap no longer describes it."""
        self.reached_valid = False
        self.zero_outside_reached = False

    def zero_outside_ok(self, regions=None) -> bool:
        """Are the rows outside the reached bitmap known to be zero -- in `regions` (names of FIELDS; default all)? True when nothing
        but the HIP backward wrote the arena since the bitmap was left AND that backward wrote those regions: a call with per-view
        scales leaves the arena's `scales` region as it was, under a NEW bitmap (its scale gradients go to a [V,P,3] tensor of their
        own), so the next call that writes `scales` must clear all of it."""
        need = frozenset(n for n, _ in FIELDS) if regions is None else frozenset(regions)
        return self.zero_outside_reached and self.flat._version == self._k8_version and need <= self._maintained

    def verify_zero_outside(self, regions=None) -> bool:
        """The invariant behind GsrGrads.zero_outside, checked the slow way (a scan of the arena; debugging / tests): is every row
        outside the reached bitmap zero in `regions` (default: the regions the last overwriting backward wrote)? A writer that
        goes through raw pointers and forgets touch() shows here."""
        sh = torch.arange(64, device=self.reached.device, dtype=torch.int64)
        inside = (((self.reached.unsqueeze(1) >> sh) & 1).reshape(-1)[:self.P]).to(torch.bool)
        names = self._maintained if regions is None else regions
        return all(bool((self.views[n].reshape(self.P, -1)[~inside] == 0).all()) for n in names)

    def reached_rows(self) -> Optional[torch.Tensor]:
        """Ascending indices of the Gaussians whose row MAY be non-zero (a superset of the non-zero rows), from the bitmap
        K8 wrote; None when the bitmap is not valid."""
        if not self.reached_valid:
            return None
        sh = torch.arange(64, device=self.reached.device, dtype=torch.int64)
        bits = ((self.reached.unsqueeze(1) >> sh) & 1).reshape(-1)[:self.P]
        return torch.nonzero(bits).reshape(-1)

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


def allreduce_grads(arena: GradArena, group=None, async_op: bool = False):
    """Sum the packed per-view gradients over all ranks, in place: the plain dense exchange (236 B / Gaussian at K = 16).
    No-op without an initialised process group. GradExchange below sends less."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    arena.touch()          # the arena now holds the sum over ALL ranks: this rank's reached-row bitmap no longer describes it
    if _host_staged(group, arena.flat):
        host = arena.flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        arena.flat.copy_(host)
        return None
    return dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _host_staged(group, *tensors) -> bool:
    """gloo is the debugging / one-GPU-box backend here: its collectives take device tensors through the host."""
    return dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    if _host_staged(group, out, inp):
        W = dist.get_world_size(group)
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(W)]
        dist.all_gather(parts, inp.cpu(), group=group)
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).reshape(out.shape))
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _all_to_all_single(out: torch.Tensor, inp: torch.Tensor, group, output_split_sizes=None, input_split_sizes=None) -> None:
    if _host_staged(group, out, inp):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=output_split_sizes,
                               input_split_sizes=input_split_sizes, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes,
                               group=group)


def _all_gather_list(outs, inp: torch.Tensor, group) -> None:
    if _host_staged(group, inp):
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in outs]
        dist.all_gather(parts, inp.cpu(), group=group)
        for o, p in zip(outs, parts):
            o.copy_(p)
    else:
        dist.all_gather(outs, inp, group=group)


def _sum_slices(recv: torch.Tensor, W: int, per: int) -> torch.Tensor:
    """((s_0 + s_1) + s_2) + ... over the W slices of `per` floats in recv: the local sum of the `direct` format, in rank
    order. On a GPU one pass of the library (csrc/exchange.hip gsr_sum_slices: W reads + 1 write per element instead of the
    3 (W - 1) of W - 1 in-place adds; same association, same bits); the torch loop elsewhere."""
    if recv.is_cuda and recv.dtype == torch.float32 and recv.is_contiguous():
        from . import _lib as L
        mine = torch.empty(per, dtype=recv.dtype, device=recv.device)
        with torch.cuda.device(recv.device):
            L.check(L.load().gsr_sum_slices(recv.data_ptr(), W, per, per, mine.data_ptr(),
                                            torch.cuda.current_stream(recv.device).cuda_stream), "gsr_sum_slices")
        return mine
    mine = recv.view(W, per)[0].clone()
    for r in range(1, W):
        mine.add_(recv.view(W, per)[r])
    return mine


class _DeviceRows:
    """(index, row) messages of a row set packed / applied by the HIP library (csrc/exchange.hip: gsr_rows_pack /
    gsr_rows_unpack) instead of torch index arithmetic over five tensors -- measured at C3 (15.7 % non-zero rows): bitmap ->
    indices 81 us + gather 78 us + 41 us per applied message in torch (profiles/r05_exchange_device_c3.json). Only for cuda
    tensors; the CPU (gloo) tests keep the torch path, which is also the reference the kernels are tested against."""

    def __init__(self, device):
        from . import _lib as L
        self.L, self.lib, self.dev = L, L.load(), device
        self.cap = 0
        self.idx = self.rows = self.scratch = None
        self.count = torch.zeros(1, dtype=torch.int32, device=device)

    def rowset(self, regions, n_rows):
        """regions: [(tensor whose row i starts at element i * stride, width, stride)]"""
        rs = self.L.GsrRowSet()
        rs.rows, rs.n_regions = int(n_rows), len(regions)
        for k, (t, width, stride) in enumerate(regions):
            rs.regions[k].ptr, rs.regions[k].width, rs.regions[k].stride = t.data_ptr(), int(width), int(stride)
        rs._keep = [t for t, _, _ in regions]
        return rs

    def pack(self, rs, mask: torch.Tensor, F: int, guess: int):
        """-> (idx int32 [n], rows [n, F]): the rows of `rs` whose bit is set in `mask`, ascending. One host read (n).
        LIFETIME: the two results are VIEWS of this object's message buffers -- the next pack() on the same _DeviceRows
        overwrites (or reallocates) them. A caller that keeps a message across another pack uses a second _DeviceRows
        (GradExchange._own_rows does for the owner-side pack of sparse_rs) or clones."""
        n_rows = int(rs.rows)
        sb = int(self.lib.gsr_rows_scratch_bytes(n_rows))
        if self.scratch is None or self.scratch.numel() < sb:
            self.scratch = torch.empty(sb + 256, dtype=torch.uint8, device=self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        cap = max(self.cap, int(guess), 1024)
        while True:
            if self.idx is None or self.idx.numel() < cap or self.rows.shape[1] != F or self.rows.shape[0] < cap:
                self.idx = torch.empty(cap, dtype=torch.int32, device=self.dev)
                self.rows = torch.empty((cap, F), dtype=torch.float32, device=self.dev)
            with torch.cuda.device(self.dev):
                self.L.check(self.lib.gsr_rows_pack(rs, mask.data_ptr(), self.idx.data_ptr(), self.rows.data_ptr(), cap,
                                                    self.count.data_ptr(), self.scratch.data_ptr(), self.scratch.numel(),
                                                    stream), "gsr_rows_pack")
            n = int(self.count.item())              # (the host read every row format needs: it sizes the wire buffers)
            if n <= cap:
                self.cap = max(self.cap, int(n * 1.25) + 1024)
                return self.idx[:n], self.rows[:n]
            cap = int(n * 1.25) + 1024

    def unpack(self, rs, idx: torch.Tensor, rows: torch.Tensor, mode: int, row_base: int = 0, touched=None):
        n = int(idx.numel())
        if n == 0:
            return
        if idx.dtype != torch.int32:
            idx = idx.to(torch.int32)
        rows = rows.contiguous()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rows_unpack(rs, idx.data_ptr(), rows.data_ptr(), n, int(row_base), int(mode),
                                                  touched.data_ptr() if touched is not None else None, stream),
                         "gsr_rows_unpack")


class _RowMessages:
    """Self-describing row messages (csrc/exchange.hip, gsr_rowmsg_pack / gsr_rowmsg_apply): the `rows` exchange of a cuda
    arena without a count on the host. ONE pack launch, ONE fixed-size all-gather, ONE apply launch; the capacity is speculated
    and the apply kernel reports -- into a page-locked word, when it STARTS -- whether every message fitted and the largest
    count, which is all the capacity policy looks at (the same number on every rank: the capacities stay equal)."""

    PENDING = 0

    def __init__(self, device):
        from . import _lib as L
        self.L, self.lib, self.dev = L, L.load(), device
        self.key = None
        self.msg = self.all = None
        self.status = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.status_np = self.status.numpy()
        self.armed = False
        self.worst_in = 0

    def buffers(self, P: int, F: int, W: int, cap: int):
        key = (P, F, W, cap)
        if key != self.key:
            nbytes = int(self.lib.gsr_rowmsg_bytes(P, F, cap))
            self.msg = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            self.all = torch.empty(W * nbytes, dtype=torch.uint8, device=self.dev)
            self.key, self.nbytes = key, nbytes
        return self.msg, self.all, self.nbytes

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    # ---- the sparse reduce-scatter: slices of slice_rows rows, one owner each
    def slice_buffers(self, P: int, F: int, W: int, slice_rows: int, cap1: int, cap2: int):
        key = ("slices", P, F, W, slice_rows, cap1, cap2)
        if key != self.key:
            n1 = int(self.lib.gsr_rowmsg_bytes(slice_rows, F, cap1))
            n2 = int(self.lib.gsr_rowmsg_bytes(slice_rows, F, cap2))
            mk = lambda n: torch.empty(n, dtype=torch.uint8, device=self.dev)
            self.send1, self.recv1, self.own2, self.all2 = mk(W * n1), mk(W * n1), mk(n2), mk(W * n2)
            self.key, self.n1, self.n2 = key, n1, n2
        return self.send1, self.recv1, self.own2, self.all2

    def pack_slices(self, rs, mask: torch.Tensor, W: int, slice_rows: int, cap1: int) -> None:
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rowmsg_pack_slices(rs, mask.data_ptr(), self.send1.data_ptr(), self.n1, int(W), int(slice_rows),
                                                         int(cap1), self._stream()), "gsr_rowmsg_pack_slices")

    def reduce_owned(self, rows_here: int, slice_rows: int, F: int, W: int, cap1: int, cap2: int) -> None:
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rowmsg_reduce(int(rows_here), int(slice_rows), int(F), self.recv1.data_ptr(), self.n1, int(W),
                                                    int(cap1), self.own2.data_ptr(), int(cap2), self._stream()), "gsr_rowmsg_reduce")

    def apply_slices(self, rs, W: int, slice_rows: int, cap2: int, touched: Optional[torch.Tensor] = None) -> None:
        self.status_np[0] = self.PENDING
        self.armed = True
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rowmsg_apply_slices(rs, self.all2.data_ptr(), self.n2, int(W), int(slice_rows), int(cap2),
                                                          self.status.data_ptr(),
                                                          touched.data_ptr() if touched is not None else None, self._stream()),
                         "gsr_rowmsg_apply_slices")

    def pack(self, rs, mask: torch.Tensor, cap: int) -> None:
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rowmsg_pack(rs, mask.data_ptr(), self.msg.data_ptr(), int(cap),
                                                  torch.cuda.current_stream(self.dev).cuda_stream), "gsr_rowmsg_pack")

    def apply(self, rs, W: int, cap: int, touched: Optional[torch.Tensor] = None) -> None:
        self.status_np[0] = self.PENDING
        self.armed = True
        with torch.cuda.device(self.dev):
            self.L.check(self.lib.gsr_rowmsg_apply(rs, self.all.data_ptr(), self.nbytes, int(W), int(cap), self.status.data_ptr(),
                                                   touched.data_ptr() if touched is not None else None,
                                                   torch.cuda.current_stream(self.dev).cuda_stream), "gsr_rowmsg_apply")

    def result(self):
        """(applied, largest count) of the last apply(); waits for the apply kernel to START (not for it to finish)."""
        if not self.armed:
            return True, 0
        spins = 0
        while int(self.status_np[0]) == self.PENDING:
            spins += 1
            if spins > 4096:            # (not fine-grained page-locked memory: the word shows at kernel end)
                torch.cuda.current_stream(self.dev).synchronize()
                if int(self.status_np[0]) == self.PENDING:      # (another stream's apply: wait for the device)
                    torch.cuda.synchronize(self.dev)
                if int(self.status_np[0]) == self.PENDING:
                    self.armed = False
                    raise RuntimeError("the apply kernel of a row-message exchange never stored its status (launch failed?)")
        self.armed = False
        v = int(self.status_np[0])
        self.worst_in = (v >> 33) & 0x7FFFFFFF           # (owners' messages: the largest count any owner received)
        return (v & 3) == 1, (v >> 2) & 0x7FFFFFFF


class ExchangeHandle:
    """reduce(async_op=True): the exchange runs on the exchange's own stream; wait() makes the caller's current stream wait for it
    (no host synchronisation)."""

    def __init__(self, event, dev):
        self._event, self._dev = event, dev

    def wait(self) -> None:
        if self._event is not None:
            torch.cuda.current_stream(self._dev).wait_event(self._event)
            self._event = None


class GradExchange:
    """Sum of the arena over the ranks of a step, moving only what can be non-zero.

    What a rank's arena holds after its views' backward (K8 writes zeros everywhere else):
      * SH columns beyond the ACTIVE degree D are identically zero (K8 writes dL/dSH for (D+1)^2 coefficients only;
        D starts at 0 and grows by one every 500 steps, gs_renderer.py:185, 578-580): 11 + 3 (D+1)^2 floats per Gaussian
        can differ from zero, not 11 + 3 K  (14 of 59 at D = 0, 23 at D = 1, 38 at D = 2);
      * rows of Gaussians no pixel composited are identically zero: culled ones (indoor scenes: 94 % of 2 M per view) and
        everything behind the opaque front layers of an object (early termination T < 1e-4).
    Wire formats (`mode`; "auto" picks per step: on a cuda arena with K8's reached bitmap between the message form of `sparse_rs` and
    `dense` from the capacities the message form has learnt -- no host read --, elsewhere from the non-zero row fraction, one small
    host read):
      * "dense": all-reduce of [geometry 11 P | active SH columns 3 (D+1)^2 P] -- the arena itself when D is the stored
        degree (zero-copy), else a packed staging buffer (one strided copy each way);
      * "direct": the dense exchange spelled as the two one-hop steps a fully connected xGMI node allows: ONE all-to-all
        (rank r receives slice r of every rank's buffer: W - 1 messages of |buffer| / W per rank, all seven links busy at
        once), a local sum in rank order, ONE all-gather of the reduced slices. Same bytes as the ring, but every byte
        crosses one link once instead of travelling W - 1 hops in lock step: (W-1)/W |buffer| / (7 x 153 GB/s) per phase
        (2 x 96 us for 118 MB at W = 8) against the ring's 2 (W-1)/W |buffer| / busbw. The local sum is one pass of the library
        on a GPU (gsr_sum_slices, 26 us at C3). `bench.py --exchange measure` times it beside `dense` (no xGMI node has run it yet);
      * "rows":  every rank contributes only its non-zero rows: all-gather of the row counts, then of (row index, row
        values) padded to the largest count; each rank adds the contributions in RANK ORDER, so all replicas end up
        with bit-identical sums (as they do with the ring all-reduce) and keep taking identical optimizer steps.
        Received bytes per rank: sum_r nnz_r (4 + 4 F) with F = 11 + 3 (D+1)^2, against 2 (W-1)/W 4 F P for the dense
        ring -- fewer bytes below ~ 2 (W-1) / W^2 of the rows per rank (22 % at W = 8); "auto" takes it below half of that
        (the format also pays a gather, a scatter-add and two host reads).
      * "sparse_rs": the sparse form of reduce-scatter + all-gather. Rank o owns the rows [o P/W, (o+1) P/W). Every rank
        sends each owner only ITS non-zero rows of that range (one all-to-all with uneven splits: (W-1)/W nnz_r (4 + 4F)
        bytes out), the owner adds them in rank order, and the owners' reduced rows (the union over the ranks' views)
        are all-gathered. Per rank at W = 8, C3 (15.7 % of the rows per rank, union ~30 %): 16.5 + 31.5 = 48 MB against
        the dense ring's 206 MB. Three small host reads per step (row counts).
    `reduce_scatter_adam` is the sharded-optimizer form of the dense exchange (flat parameter arena required).

    "rows" on a cuda arena whose reached bitmap K8 left is valid takes the DEVICE form (`_RowMessages`): one pack launch, one
    fixed-size all-gather of self-describing messages, one apply launch that stores the rank-ordered sums -- no count on the
    host, no zero-fill of the arena, no index list on the wire. The capacity is speculated (it starts at P / 4 rows and follows
    1.25 x the largest count any rank ever sent); `strict` (default True) polls the apply kernel's status word -- written when
    the kernel starts, so the device is not drained -- and repeats the exchange with more room if a message did not fit (the
    arena is untouched then). strict=False never reads: the status of step k is looked at when step k + 1 calls reduce() (or
    finish()); a step that overflowed leaves its arena UN-reduced and is counted in `overflowed_steps` -- for loops whose
    consumer can tolerate or detect that (bench.py checks the counter after its timed region).
    reduce(async_op=True): the whole exchange on the exchange's own stream, ordered behind the caller's current stream;
    ExchangeHandle.wait() joins -- the per-view statistics all-reduce and the optimizer's prologue run beside it."""

    GEOM = ("means3D", "scales", "rotations", "opacities")

    def __init__(self, arena: GradArena, sh_degree: Optional[int] = None, group=None, mode: str = "auto",
                 rows_below: Optional[float] = None, strict: bool = True):
        if mode not in ("auto", "dense", "rows", "direct", "sparse_rs"):
            raise ValueError("mode is 'auto', 'dense', 'rows', 'sparse_rs' or 'direct'")
        self.arena, self.group, self.mode = arena, group, mode
        self.rows_below = rows_below
        self.strict = bool(strict)
        self.overflowed_steps = 0
        self._msgs = None             # _RowMessages (device form of "rows")
        self._rows_cap = 0            # speculated capacity of a row message (rows); the same on every rank
        self._rs_caps = [0, 0]        # sparse_rs: capacity of a (rank -> owner) message, of an owner's reduced message
        self._settle = None           # how the pending status word is to be read: "rows" / "sparse_rs"
        self._side = None             # the exchange's own stream (async_op)
        self.sh_degree = None
        self.set_sh_degree(sh_degree)
        self.last = {}          # what the last reduce() did (format, bytes): for logs / bench lines
        self._dev_rows = _DeviceRows(arena.flat.device) if arena.flat.is_cuda else None

    # ---- layout
    def set_sh_degree(self, sh_degree: Optional[int]) -> None:
        """Active SH degree of the coming steps (None = all stored coefficients)."""
        K = self.arena.K
        nb = K if sh_degree is None else min(K, (int(sh_degree) + 1) ** 2)
        self.sh_degree, self.nb = sh_degree, nb
        P = self.arena.P
        self.row_floats = 11 + 3 * nb
        self._staging = None
        if nb < K:      # [geometry regions as laid out in the arena (incl. their alignment padding) | packed active SH]
            self._staging = torch.zeros(self._offset_of_shs() + P * nb * 3, dtype=self.arena.flat.dtype,
                                        device=self.arena.flat.device)

    def _offset_of_shs(self) -> int:
        es = self.arena.flat.element_size()
        return (self.arena.views["shs"].data_ptr() - self.arena.flat.data_ptr()) // es

    def wire_buffer(self) -> torch.Tensor:
        """[geometry | active SH columns] as one contiguous tensor: the arena itself at full degree, else the staging
        buffer filled from the arena (pack)."""
        if self._staging is None:
            return self.arena.flat
        P, nb, o = self.arena.P, self.nb, self._offset_of_shs()
        self._staging[:o].copy_(self.arena.flat[:o])
        self._staging[o:o + P * nb * 3].view(P, nb, 3).copy_(self.arena.views["shs"][:, :nb, :])
        return self._staging[:o + P * nb * 3]

    def _unpack(self, wire: torch.Tensor) -> None:
        if self._staging is None:
            return
        P, nb, o = self.arena.P, self.nb, self._offset_of_shs()
        self.arena.flat[:o].copy_(wire[:o])
        self.arena.views["shs"][:, :nb, :].copy_(wire[o:o + P * nb * 3].view(P, nb, 3))

    # ---- non-zero rows
    def nonzero_rows(self) -> torch.Tensor:
        """Indices (ascending) of the Gaussians whose gradient row has any non-zero entry on this rank -- or, when K8 left
        its reached-row bitmap (GradArena.reached_rows), the Gaussians K7 reached: a superset (a reached Gaussian's ten
        sums may still be zero), 62 KB to look at instead of a scan of the whole arena."""
        idx = self.arena.reached_rows() if hasattr(self.arena, "reached_rows") else None
        if idx is not None:
            return idx
        v = self.arena.views
        P, nb = self.arena.P, self.nb
        m = (v["means3D"] != 0).any(1) | (v["scales"] != 0).any(1) | (v["rotations"] != 0).any(1) | \
            (v["opacities"].reshape(P) != 0) | (v["shs"][:, :nb, :].reshape(P, nb * 3) != 0).any(1)
        return torch.nonzero(m).reshape(-1)

    def _arena_rowset(self):
        v, P, K, nb = self.arena.views, self.arena.P, self.arena.K, self.nb
        return self._dev_rows.rowset([(v["means3D"], 3, 3), (v["scales"], 3, 3), (v["rotations"], 4, 4), (v["opacities"], 1, 1),
                                      (v["shs"], 3 * nb, 3 * K)], P)

    def _message(self):
        """(idx, rows): this rank's non-zero rows as a message, indices ascending. On a cuda arena whose reached bitmap K8 left
        is valid: ONE pack through the HIP library (idx int32); otherwise the torch path (idx int64)."""
        a = self.arena
        if self._dev_rows is not None and getattr(a, "reached_valid", False):
            guess = getattr(self, "_last_n", 0)
            idx, rows = self._dev_rows.pack(self._arena_rowset(), a.reached, self.row_floats, int(guess * 1.25))
            self._last_n = int(idx.numel())
            return idx, rows
        idx = self.nonzero_rows()
        rows = self._rows_of(idx) if idx.numel() else torch.zeros((0, self.row_floats), dtype=a.flat.dtype, device=a.flat.device)
        return idx, rows

    def _rows_of(self, idx: torch.Tensor) -> torch.Tensor:
        v = self.arena.views
        n, nb = idx.numel(), self.nb
        return torch.cat([v["means3D"][idx], v["scales"][idx], v["rotations"][idx], v["opacities"][idx].reshape(n, 1),
                          v["shs"][idx, :nb, :].reshape(n, nb * 3)], dim=1)

    def _add_rows(self, idx: torch.Tensor, rows: torch.Tensor) -> None:
        if self._dev_rows is not None:
            self._dev_rows.unpack(self._arena_rowset(), idx, rows, mode=0)
            return
        idx = idx.to(torch.int64)
        v = self.arena.views
        n, nb = idx.numel(), self.nb
        v["means3D"].index_add_(0, idx, rows[:, 0:3])
        v["scales"].index_add_(0, idx, rows[:, 3:6])
        v["rotations"].index_add_(0, idx, rows[:, 6:10])
        v["opacities"].index_add_(0, idx, rows[:, 10:11].reshape((n,) + tuple(v["opacities"].shape[1:])))
        v["shs"][:, :nb, :].index_add_(0, idx, rows[:, 11:].reshape(n, nb, 3))

    # ---- the exchange
    def reduce(self, async_op: bool = False):
        """Leaves the sum over all ranks in the arena, on every rank, bit-identical across ranks. Afterwards the arena's
        reached-row bitmap (this rank's views only) no longer describes its contents: GradArena.touch().
        async_op (cuda arenas): the exchange is enqueued on the exchange's own stream behind everything the caller's current
        stream holds, and an ExchangeHandle is returned; the arena must not be touched before handle.wait()."""
        if async_op and self.arena.flat.is_cuda and dist.is_available() and dist.is_initialized() \
                and dist.get_world_size(self.group) > 1:
            dev = self.arena.flat.device
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            self._side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._side):
                self.reduce(False)
                ev = torch.cuda.Event()
                ev.record(self._side)
            return ExchangeHandle(ev, dev)
        kept = self.arena.flat._version == getattr(self.arena, "_k8_version", None) and \
            getattr(self.arena, "zero_outside_reached", False)
        try:
            self._reduce()
        finally:
            if self.last.get("format") != "none":
                self.arena.touch()
                if kept and self.last.get("device") and self.last.get("union_bitmap"):
                    # The message forms store the union of the ranks' rows and leave its bitmap in arena.reached (or nothing at
                    # all, when a message overflowed): rows outside the bitmap are as zero as they were before -- the next
                    # backward may keep clearing only what the bitmap names (GsrGrads.zero_outside). The bitmap no longer says
                    # what THIS rank has to send (reached_valid stays False until the next backward).
                    self.arena.zero_outside_reached = True
        return ExchangeHandle(None, None) if async_op else None

    def finish(self) -> bool:
        """strict=False: looks at the status of the last device-form exchange (waits for its apply kernel to start). False = a
        message had not fitted: that step's arena was left un-reduced (counted in overflowed_steps, capacity raised)."""
        return self._settle_rows()

    def _settle_rows(self) -> bool:
        if self._msgs is None or not self._msgs.armed:
            return True
        ok, worst = self._msgs.result()
        if self._settle == "sparse_rs":
            self._grow_rs_caps(self._msgs.worst_in, worst)
        else:
            self._grow_cap(worst)
        if not ok:
            self.overflowed_steps += 1
        return ok

    def _grow_rs_caps(self, worst_in: int, worst_out: int) -> None:
        per = self._slice_rows(dist.get_world_size(self.group))
        lim = (per + 1023) // 1024 * 1024
        for i, wv in enumerate((worst_in, worst_out)):
            if wv >= 0x7FFFFFFF:          # (a poisoned owner message: its own count is unknown -- the first phase overflowed)
                continue
            want = (int(wv * 1.25) + 1535) // 512 * 512
            if want > self._rs_caps[i]:
                self._rs_caps[i] = min(want, lim)

    def _slice_rows(self, W: int) -> int:
        return ((self.arena.P + W - 1) // W + 63) // 64 * 64

    def _reduce_sparse_rs_device(self, W: int) -> None:
        """`sparse_rs` on the device, no host read: every rank packs W slice messages (ONE launch), ONE all-to-all with equal
        splits hands owner o its slice from every rank, the owner reduces the W messages into the message of their union (ONE
        launch), ONE all-gather moves the owners' messages and ONE launch stores them -- the arena is written by that last
        launch only, and only if every message of both phases fitted."""
        P, F = self.arena.P, self.row_floats
        if self._msgs is None:
            self._msgs = _RowMessages(self.arena.flat.device)
        self._settle_rows()
        per = self._slice_rows(W)
        lim = (per + 1023) // 1024 * 1024
        if self._rs_caps[0] == 0:
            self._rs_caps = [min((max(per // 4, 512) + 511) // 512 * 512, lim), min((max(per // 2, 512) + 511) // 512 * 512, lim)]
        rs = self._arena_rowset()
        r = dist.get_rank(self.group)
        rows_here = max(0, min(per, P - r * per))
        tries = 0
        while True:
            cap1, cap2 = self._rs_caps
            send1, recv1, own2, all2 = self._msgs.slice_buffers(P, F, W, per, cap1, cap2)
            self._msgs.pack_slices(rs, self.arena.reached, W, per, cap1)
            _all_to_all_single(recv1, send1, self.group)
            self._msgs.reduce_owned(rows_here, per, F, W, cap1, cap2)
            _all_gather_into(all2, own2, self.group)
            # (touched: the owners' bitmaps side by side = the union over the ranks, left in the arena's own bitmap -- _kept_sparse)
            self._msgs.apply_slices(rs, W, per, cap2, touched=self.arena.reached)
            self._settle = "sparse_rs"
            self.last = dict(format="sparse_rs", device=True, row_floats=F, cap_rows=[cap1, cap2], host_reads=0,
                             bytes_per_rank=int((W - 1) * (self._msgs.n1 + self._msgs.n2)), union_bitmap=True)
            if not self.strict:
                return
            ok, worst = self._msgs.result()
            self.last.update(rows_max=[self._msgs.worst_in, worst], host_reads=1)
            self._grow_rs_caps(self._msgs.worst_in, worst)
            if ok:
                return
            tries += 1                              # nothing was applied: the arena still holds this rank's own gradients
            if tries > 3:
                raise RuntimeError(f"slice messages do not fit their capacities {self._rs_caps} after {tries} attempts")

    def _grow_cap(self, worst: int) -> None:
        want = (int(worst * 1.25) + 2047) // 1024 * 1024
        if want > self._rows_cap:
            self._rows_cap = min(want, (self.arena.P + 1023) // 1024 * 1024)

    def _reduce_rows_device(self, W: int) -> None:
        """The `rows` format on the device (see the class docstring)."""
        P, F = self.arena.P, self.row_floats
        if self._msgs is None:
            self._msgs = _RowMessages(self.arena.flat.device)
        self._settle_rows()                        # (strict=False: the previous step's status, long since written)
        if self._rows_cap == 0:
            self._rows_cap = min((max(P // 4, 1024) + 1023) // 1024 * 1024, (P + 1023) // 1024 * 1024)
        rs = self._arena_rowset()
        tries = 0
        while True:
            cap = self._rows_cap
            msg, allm, nbytes = self._msgs.buffers(P, F, W, cap)
            self._msgs.pack(rs, self.arena.reached, cap)
            _all_gather_into(allm, msg, self.group)
            self._msgs.apply(rs, W, cap, touched=self.arena.reached)       # (the union bitmap: _kept_sparse)
            self._settle = "rows"
            self.last = dict(format="rows", device=True, row_floats=F, cap_rows=cap, bytes_per_rank=int((W - 1) * nbytes),
                             host_reads=0, union_bitmap=True)
            if not self.strict:
                return
            ok, worst = self._msgs.result()
            self.last.update(rows_max=worst, host_reads=1)
            self._grow_cap(worst)
            if ok:
                return
            tries += 1                              # a message did not fit: nothing was applied, the arena is intact
            if tries > 2:
                raise RuntimeError(f"row messages of {worst} rows do not fit a capacity of {cap} after {tries} attempts")

    def _reduce(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            self.last = dict(format="none", bytes_sent=0)
            return
        W = dist.get_world_size(self.group)
        P, F = self.arena.P, self.row_floats
        mode = self.mode
        idx = counts = None
        msg_rows = None
        on_device = self._dev_rows is not None and getattr(self.arena, "reached_valid", False)
        if mode == "auto" and on_device:
            # no count on the host: the sparse reduce-scatter in its message form unless the capacities it has learnt say that it
            # would move more than half of what the dense ring moves (the same decision on every rank: the capacities are)
            cap1, cap2 = self._rs_caps
            sparse_bytes = (W - 1) * (cap1 + cap2) * 4 * F
            dense_bytes = 2 * (W - 1) / W * 4 * F * P
            if self.rows_below is not None and cap1:
                mode = "sparse_rs" if cap1 * W <= 1.25 * self.rows_below * P + 2048 * W else "dense"
            else:
                mode = "sparse_rs" if (cap1 == 0 or sparse_bytes < 0.5 * dense_bytes) else "dense"
            if mode == "dense" and cap1:
                self._rs_probe = getattr(self, "_rs_probe", 0) + 1      # (look again every 64 steps: the row sets change slowly)
                if self._rs_probe % 64 == 0:
                    mode = "sparse_rs"
        if mode == "rows" and on_device:
            self._reduce_rows_device(W)
            return
        if mode in ("auto", "rows"):      # ("sparse_rs" counts per owner itself)
            idx, msg_rows = self._message()
            cnt = torch.empty(W, dtype=torch.int64, device=idx.device)
            _all_gather_into(cnt, torch.tensor([idx.numel()], dtype=torch.int64, device=idx.device), self.group)
            counts = [int(c) for c in cnt.tolist()]           # (the one host read of the exchange)
            if mode == "auto":
                # received bytes: rows format sum_r n_r (4 + 4F) vs dense ring 2 (W-1)/W 4 F P (the same decision on every
                # rank). The rows format also pays a gather, a scatter-add and two host reads, so it must win clearly:
                # by default it is taken below HALF the dense bytes.
                rows_bytes = sum(counts) * (4 + 4 * F)
                dense_bytes = 2 * (W - 1) / W * 4 * F * P
                use_rows = (sum(counts) / W <= self.rows_below * P) if self.rows_below is not None else \
                    (rows_bytes < 0.5 * dense_bytes)
                mode = "rows" if use_rows else "dense"
        if mode == "dense":
            wire = self.wire_buffer()
            if _host_staged(self.group, wire):
                host = wire.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                wire.copy_(host)
            else:
                dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
            self._unpack(wire)
            self.last = dict(format="dense", row_floats=F, bytes_per_rank=int(2 * (W - 1) / W * 4 * wire.numel()))
            return
        if mode == "direct":
            wire = self.wire_buffer()
            n = wire.numel()
            per = (n + W - 1) // W
            send = wire if per * W == n else torch.cat([wire, wire.new_zeros(per * W - n)])
            recv = torch.empty_like(send)
            _all_to_all_single(recv, send, self.group)                        # slice r of every rank -> rank r
            mine = _sum_slices(recv, W, per)                                  # rank order: identical sums everywhere
            _all_gather_into(send, mine, self.group)
            if send is not wire:
                wire.copy_(send[:n])
            self._unpack(wire)
            self.last = dict(format="direct", row_floats=F, bytes_per_rank=int(2 * (W - 1) / W * 4 * n))
            return
        if mode == "sparse_rs":
            if self._dev_rows is not None and getattr(self.arena, "reached_valid", False):
                self._reduce_sparse_rs_device(W)
            else:
                self._reduce_sparse_rs(W)
            return
        nmax = max(counts)
        dev, dt = self.arena.flat.device, self.arena.flat.dtype
        my_idx = torch.zeros(max(nmax, 1), dtype=torch.int32, device=dev)       # (4-byte indices on the wire: P < 2^31)
        my_rows = torch.zeros((max(nmax, 1), F), dtype=dt, device=dev)
        n = idx.numel()
        my_idx[:n] = idx
        if n:
            my_rows[:n] = msg_rows
        all_idx = [torch.empty_like(my_idx) for _ in range(W)]
        all_rows = [torch.empty_like(my_rows) for _ in range(W)]
        _all_gather_list(all_idx, my_idx, self.group)
        _all_gather_list(all_rows, my_rows, self.group)
        self.arena.flat.zero_()
        for r in range(W):                                    # rank order: the same association on every rank
            if counts[r]:
                self._add_rows(all_idx[r][:counts[r]], all_rows[r][:counts[r]])
        self.last = dict(format="rows", row_floats=F, rows=counts, bytes_per_rank=int(sum(counts) * (4 + 4 * F)))

    def _set_rows(self, idx: torch.Tensor, rows: torch.Tensor) -> None:
        if self._dev_rows is not None:
            self._dev_rows.unpack(self._arena_rowset(), idx, rows, mode=1)
            return
        idx = idx.to(torch.int64)
        v = self.arena.views
        n, nb = idx.numel(), self.nb
        v["means3D"][idx] = rows[:, 0:3]
        v["scales"][idx] = rows[:, 3:6]
        v["rotations"][idx] = rows[:, 6:10]
        v["opacities"][idx] = rows[:, 10:11].reshape((n,) + tuple(v["opacities"].shape[1:]))
        v["shs"][idx, :nb, :] = rows[:, 11:].reshape(n, nb, 3)

    def _reduce_sparse_rs(self, W: int) -> None:
        P, F = self.arena.P, self.row_floats
        dev, dt = self.arena.flat.device, self.arena.flat.dtype
        r = dist.get_rank(self.group)
        per = (P + W - 1) // W
        idx, rows = self._message()                                       # ascending: contiguous per owner
        bounds = torch.searchsorted(idx, (torch.arange(0, W + 1, device=dev, dtype=torch.int64) * per).to(idx.dtype))
        send_counts = (bounds[1:] - bounds[:-1]).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        _all_to_all_single(recv_counts, send_counts, self.group)
        sc, rc = [int(x) for x in send_counts.tolist()], [int(x) for x in recv_counts.tolist()]      # host read 1
        rows = rows.contiguous()
        idx32 = idx.to(torch.int32)
        got_idx = torch.empty(sum(rc), dtype=torch.int32, device=dev)
        got_rows = torch.empty((sum(rc), F), dtype=dt, device=dev)
        _all_to_all_single(got_idx, idx32, self.group, output_split_sizes=rc, input_split_sizes=sc)
        _all_to_all_single(got_rows, rows, self.group, output_split_sizes=rc, input_split_sizes=sc)
        # owner: add the contributions in rank order (got_* are ordered by source rank) into the owned slice
        lo = r * per
        n_own = max(0, min(per, P - lo))
        mine = torch.zeros((max(n_own, 1), F), dtype=dt, device=dev)
        if self._dev_rows is not None:
            # the owned slice as a row-major row set: messages applied in rank order by the library, the rows they touched
            # collected as a bitmap and packed again (csrc/exchange.hip)
            dr = self._dev_rows
            own_set = dr.rowset([(mine, F, F)], max(n_own, 1))
            touched_bits = torch.zeros((max(n_own, 1) + 63) // 64, dtype=torch.int64, device=dev)
            off = 0
            for src in range(W):
                if rc[src]:
                    dr.unpack(own_set, got_idx[off:off + rc[src]], got_rows[off:off + rc[src]], mode=0, row_base=lo,
                              touched=touched_bits)
                off += rc[src]
            if not hasattr(self, "_own_rows"):
                self._own_rows = _DeviceRows(dev)
            own_local, own_rows = self._own_rows.pack(own_set, touched_bits, F, 0)
            own_idx = own_local.to(torch.int64)
        else:
            touched = torch.zeros(max(n_own, 1), dtype=torch.bool, device=dev)
            off = 0
            for src in range(W):
                if rc[src]:
                    li = got_idx[off:off + rc[src]].to(torch.int64) - lo
                    mine.index_add_(0, li, got_rows[off:off + rc[src]])
                    touched[li] = True
                off += rc[src]
            own_idx = torch.nonzero(touched[:n_own]).reshape(-1) if n_own else torch.zeros(0, dtype=torch.int64, device=dev)
            own_rows = mine[own_idx] if own_idx.numel() else None
        cnt = torch.empty(W, dtype=torch.int64, device=dev)
        _all_gather_into(cnt, torch.tensor([own_idx.numel()], dtype=torch.int64, device=dev), self.group)
        counts = [int(c) for c in cnt.tolist()]                           # host read 2 (own_idx.numel() was the third)
        nmax = max(max(counts), 1)
        my_idx = torch.zeros(nmax, dtype=torch.int32, device=dev)
        my_rows = torch.zeros((nmax, F), dtype=dt, device=dev)
        my_idx[:own_idx.numel()] = (own_idx + lo).to(torch.int32)
        if own_idx.numel():
            my_rows[:own_idx.numel()] = own_rows
        all_idx = torch.empty(W * nmax, dtype=torch.int32, device=dev)
        all_rows = torch.empty((W * nmax, F), dtype=dt, device=dev)
        _all_gather_into(all_idx, my_idx, self.group)
        _all_gather_into(all_rows, my_rows, self.group)
        self.arena.flat.zero_()
        for o in range(W):                                                # disjoint row ranges: plain stores
            if counts[o]:
                self._set_rows(all_idx[o * nmax:o * nmax + counts[o]], all_rows[o * nmax:o * nmax + counts[o]])
        self.last = dict(format="sparse_rs", row_floats=F, rows_sent=sum(sc) - sc[r], rows_reduced=counts,
                         bytes_per_rank=int((sum(sc) - sc[r]) * (4 + 4 * F) + (sum(counts) - counts[r]) * (4 + 4 * F)))

    def reduce_scatter_adam(self, param_flat: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
                            lr_of, betas=(0.9, 0.999), eps: float = 1e-15, adam=None):
        """The dense exchange in its sharded-optimizer form (full stored degree only): reduce-scatter of the arena, every
        rank updates ITS 1/W of the flat parameter arena (`param_flat`, laid out like the gradient arena: the trainer's
        leaves are views of it) with `adam(param_shard, grad_shard, m_shard, v_shard, step, lr tensor-or-callable)`, then
        all-gather of the updated parameters. Same wire bytes as the all-reduce, 1/W of the optimizer's HBM traffic per
        rank, and the second half of the exchange carries PARAMETERS, which the next forward needs anyway.
        lr_of(lo, hi) -> per-element learning rates of flat elements [lo, hi) (the reference uses one lr per group)."""
        if self._staging is not None:
            raise ValueError("reduce_scatter_adam needs the full stored SH degree (the arena is the wire buffer)")
        W = dist.get_world_size(self.group) if dist.is_initialized() else 1
        r = dist.get_rank(self.group) if dist.is_initialized() else 0
        n = self.arena.flat.numel()
        per = (n + W - 1) // W
        if per * W != n:
            raise ValueError(f"arena size {n} must be a multiple of the world size {W} (pad P)")
        lo, hi = r * per, (r + 1) * per
        gshard = torch.empty(per, dtype=self.arena.flat.dtype, device=self.arena.flat.device)
        if W > 1 and dist.get_backend(self.group) != "gloo":
            dist.reduce_scatter_tensor(gshard, self.arena.flat, op=dist.ReduceOp.SUM, group=self.group)
        elif W > 1:       # gloo (the CPU tests) has no reduce-scatter: same result through an all-reduce
            dist.all_reduce(self.arena.flat, op=dist.ReduceOp.SUM, group=self.group)
            gshard.copy_(self.arena.flat[lo:hi])
        else:
            gshard.copy_(self.arena.flat)
        adam(param_flat[lo:hi], gshard, exp_avg[lo:hi], exp_avg_sq[lo:hi], step, lr_of(lo, hi), betas, eps)
        if W > 1:
            dist.all_gather_into_tensor(param_flat, param_flat[lo:hi].clone(), group=self.group)
        self.last = dict(format="reduce_scatter+adam+all_gather", bytes_per_rank=int(2 * (W - 1) / W * 4 * n))


def reduce_view_stats(means2D_grad: torch.Tensor, radii: torch.Tensor, group=None):
    """Per-view densification statistics reduced so that every replica takes identical densify decisions:
    sum over views of ||means2D.grad[:, :2]||, count of views in which the Gaussian was visible, max radius."""
    norm = torch.norm(means2D_grad[:, :2], dim=-1)
    vis = (radii > 0).to(norm.dtype)
    maxr = radii.to(torch.int32).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        packed = torch.stack([norm, vis], dim=0)
        if _host_staged(group, packed):          # (gloo on device tensors: the one-GPU-box tests)
            hp, hm = packed.cpu(), maxr.cpu()
            dist.all_reduce(hp, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(hm, op=dist.ReduceOp.MAX, group=group)
            packed.copy_(hp); maxr.copy_(hm)
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(maxr, op=dist.ReduceOp.MAX, group=group)
        norm, vis = packed[0], packed[1]
    return norm, vis, maxr


def reduce_step(exchange: "GradExchange", means2D_grad: torch.Tensor, radii: torch.Tensor, stats_group=None):
    """The two exchanges of a data-parallel step side by side: the gradient exchange starts on the exchange's own stream (behind
    everything the caller's stream holds -- K8 included), the per-view densification statistics (reduce_view_stats: two small
    all-reduces, gs_renderer.py:1034-1065) run on the caller's stream beside it, then the caller's stream joins. stats_group: a
    process group of its own for the statistics (dist.new_group()) lets RCCL run the two exchanges concurrently -- collectives of
    ONE communicator execute one after the other whatever stream they were issued from; without it only the exchange's own kernels
    (pack / reduce / apply) overlap the statistics. Returns reduce_view_stats' (norm sum, visible count, max radius)."""
    handle = exchange.reduce(async_op=True)
    stats = reduce_view_stats(means2D_grad, radii, stats_group if stats_group is not None else exchange.group)
    if handle is not None:
        handle.wait()
    return stats


def shard_views(n_views: int, rank: int, world: int):
    """Static round-robin view -> rank assignment (view i runs on rank i % world)."""
    return [i for i in range(n_views) if i % world == rank]


def _accumulate_style(fn) -> str:
    """How the per-view callback takes the `accumulate` flag: "keyword" (a parameter of that name, or **kwargs),
    "positional" (a fifth positional parameter under any name, or *args) or "none" (the four-argument contract of round 1:
    the callback always overwrites grad_out)."""
    import inspect
    try:
        ps = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return "keyword"
    if any(p.kind is p.VAR_KEYWORD or p.name == "accumulate" for p in ps):
        return "keyword"
    n_pos = sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in ps)
    if n_pos >= 5 or any(p.kind is p.VAR_POSITIONAL for p in ps):
        return "positional"
    return "none"


def _accepts_accumulate(fn) -> bool:
    return _accumulate_style(fn) != "none"


def render_views_data_parallel(rasterize_view, params: Dict[str, torch.Tensor], cameras, upstream, arena: GradArena,
                               group=None, exchange: Optional[GradExchange] = None):
    """Render this rank's share of `cameras` (fwd+bwd) and leave the SUM over all views of every parameter
    gradient in `arena` on every rank.

    rasterize_view(params, camera, grad_out, upstream, accumulate=...) runs one view forward+backward and writes that
    view's parameter gradients into the tensors of grad_out (the arena's views): overwriting them when accumulate is
    False, ADDING to them when it is True -- K8's accumulate mode (RasterContext.accumulate): the sum over a rank's views is
    formed on the device by the kernel that produces the gradients, no extra pass over the arena. The flag is passed by
    keyword to a callback with a parameter named `accumulate` (or **kwargs), as the fifth positional argument to one with
    five or more positional parameters (or *args); a four-parameter callback is taken to overwrite always and the sum over
    the rank's views is then formed here (clone + add; the arena's reached-row bitmap is invalidated, since the callback's
    K8 overwrote it with the last view's rows).
    Equivalent, to fp32 summation order, to the sequential accumulation the reference performs."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = shard_views(len(cameras), rank, world)
    outs = []
    # A callback written against the earlier contract takes four arguments and always OVERWRITES grad_out: for those the
    # sum over this rank's views is formed here (clone + add per extra view), as it was before K8 learnt to accumulate.
    style = _accumulate_style(rasterize_view)
    for j, vi in enumerate(mine):
        if style == "keyword":
            outs.append(rasterize_view(params, cameras[vi], arena.views, upstream[vi], accumulate=j > 0))
        elif style == "positional":
            outs.append(rasterize_view(params, cameras[vi], arena.views, upstream[vi], j > 0))
        elif j == 0:
            outs.append(rasterize_view(params, cameras[vi], arena.views, upstream[vi]))
        else:
            held = arena.flat.clone()
            outs.append(rasterize_view(params, cameras[vi], arena.views, upstream[vi]))
            arena.flat.add_(held)
            arena.touch()      # K8 (accumulate = 0) left the LAST view's rows in the bitmap; the arena holds the sum of all
    if not mine:
        arena.flat.zero_()
        arena.touch()
    if exchange is not None:
        exchange.reduce()
    else:
        allreduce_grads(arena, group)
    return outs
