"""Vocab-row sharding across the GPUs of one node (one process per GPU).

Target-vocab rows are independent in every configuration the kept API supports, so the path shards by rows with
replicated weights and source embeddings, and the only data-path collective is the all-gather that reassembles the
full [N, E] matrices on every GPU (RCCL over xGMI; backend "nccl" is RCCL on ROCm).  The reference shards the row
batch over local devices the same way (scripts/transfer.py:90-91, zett/utils.py:26) and pads the last shard
(scripts/transfer.py:63-67).  Per-row results do not depend on the shard a row lands in, so the gathered matrix is
bit-identical for every P (tests/test_full_size_gpu.py).

Overlap.  A plain "compute the shard, then gather" leaves the GPUs idle for the whole exchange: at 8 GPUs the headline
vocabulary is ~10 ms of compute per rank against ~0.94 GB received per rank.  A caller has ONE vocabulary, so there is
no "next step" to hide the exchange under.  Instead the vocabulary is cut into `chunks` row blocks (multiples of the
world size), each block is sharded over the ranks, and the all-gather of block k runs on RCCL's stream while the
forward of block k + 1 computes; block k is gathered straight into its place of the final matrix (rank shards of one
block are adjacent rows), so no re-assembly copy follows.  Only the last block's exchange is exposed — and of that, only
pred_out's: pred_in and bias leave as soon as the first output head has written them, under the second head's GEMMs
(RowGather, "early start").  Two interchangeable transports: RCCL's all-gather, and a direct fan-out in which every rank
sends its shard to its seven peers at once, one xGMI link each (RowGather, mode "fanout").
"""
from __future__ import annotations

from typing import Callable, List, NamedTuple, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    per = -(-int(n_rows) // int(world))
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


class Block(NamedTuple):
    """One row block of the vocabulary as this rank sees it."""
    start: int      # first global row of the block
    rows: int       # global rows in the block
    per: int        # nominal shard height: every rank contributes `per` rows to the gather (the last ranks pad)
    lo: int         # this rank's global rows [lo, hi) of the block (hi - lo <= per, 0 for surplus ranks)
    hi: int


def plan_blocks(n_rows: int, world: int, rank: int, chunks: int = 2, min_rows_per_shard: int = 4096) -> List[Block]:
    """Cut [0, n_rows) into at most `chunks` blocks of equal size (a multiple of `world`, so that block k starts at row
    k * world * per and its gathered shards land in place); fewer blocks when a shard would drop under
    `min_rows_per_shard` rows: small forwards waste the GPU (256-row GEMM tiles on ~2.4 packed positions per row, the
    per-distinct-id table that shrinks more slowly than the rows: a 4 096-row forward of the headline workload runs at
    76 % of the 32 768-row rate per row, a 2 048-row one lower still), which would cost more than the exchange it hides.
    At 8 GPUs the headline vocabulary (4 096 rows per rank) therefore stays one block per rank; 2 and 4 GPUs, and the
    50 k / 262 k-row vocabularies, get two."""
    n_rows, world, chunks = int(n_rows), int(world), max(1, int(chunks))
    if n_rows <= 0:
        return []
    chunks = max(1, min(chunks, n_rows // max(1, world * min_rows_per_shard)))
    per = -(-n_rows // (world * chunks))              # shard height of a full block
    step = per * world
    out = []
    start = 0
    while start < n_rows:
        rows = min(step, n_rows - start)
        p = per if rows == step else -(-rows // world)
        lo = min(start + rank * p, start + rows)
        hi = min(lo + p, start + rows)
        out.append(Block(start, rows, p, lo, hi))
        start += rows
    return out


def padded_rows(blocks: Sequence[Block], world: int) -> int:
    """Rows of the buffers the blocks are gathered into (>= n_rows: the last block may carry padding rows at its end)."""
    return sum(b.per * world for b in blocks)


def rank_row_counts(n_rows: int, world: int, chunks: int = 2, min_rows_per_shard: int = 4096) -> List[int]:
    """Rows plan_blocks hands to each rank over all its blocks (sums to n_rows)."""
    return [sum(b.hi - b.lo for b in plan_blocks(n_rows, world, r, chunks, min_rows_per_shard)) for r in range(int(world))]


def affinity_order(surface_forms: torch.Tensor, world: int, pad_token_id: int, n_ids: int, chunks: int = 2,
                   min_rows_per_shard: int = 4096) -> torch.Tensor:
    """Row order for the vocabulary-sharded path in which rows that share source ids land on the same rank (int64 [n_rows] on the
    device: `surface_forms.index_select(0, order)` is the matrix predict_sharded should then shard CONTIGUOUSLY, and
    `out.index_copy_(0, order, gathered)` undoes it).  zett_partition_rows (csrc/partition.hip.h) groups the rows by rank —
    capacities = the row counts plan_blocks gives each rank — and the groups are laid out block by block, rank by rank, exactly
    where plan_blocks(…, rank) will look for them.  Deterministic and communication-free: every rank computes the same order
    from the same matrix.  Why: a rank's forward runs the hoisted input projection once per DISTINCT source id of its shard and
    layer 0's Q/K/V once per distinct (id, position) pair; the reference hands its devices the rows in random order
    (scripts/transfer.py:54-67, 90-91), contiguous shards are no better, and at 8 ranks the headline vocabulary then carries
    8 340 ids per rank where this order leaves ~5 970 and gives the pair lever its repeats back."""
    import ctypes as C

    from . import _lib
    if not surface_forms.is_cuda:
        raise RuntimeError("zett_amd computes on MI355X only: affinity_order takes the surface-form matrix on a cuda (ROCm) device")
    world = int(world)
    sfm = surface_forms.to(torch.int32).contiguous()
    n, seq = sfm.shape
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=sfm.device)
    lib = _lib.load()
    caps = rank_row_counts(n, world, chunks, min_rows_per_shard)
    ws_bytes = C.c_int64(0)
    _lib.check(lib.zett_partition_workspace_bytes(n, int(n_ids), C.byref(ws_bytes)), "zett_partition_workspace_bytes")
    ws = torch.empty((ws_bytes.value,), dtype=torch.uint8, device=sfm.device)
    perm = torch.empty((n,), dtype=torch.int32, device=sfm.device)
    index = sfm.device.index if sfm.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(sfm.device):
        stream = torch.cuda.current_stream(sfm.device).cuda_stream
        _lib.check(lib.zett_partition_rows(C.c_void_p(sfm.data_ptr()), n, seq, int(pad_token_id), int(n_ids), world, (C.c_int32 * world)(*caps),
                                           C.c_void_p(perm.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes.value, index, C.c_void_p(stream)),
                   "zett_partition_rows")
    # where plan_blocks looks for rank r's rows: per block [lo, hi).  src[i] = position in `perm` (grouped by rank) of the row that
    # goes to place i of the sharded order — pure arithmetic on the block plan, cached per (n, world, chunks)
    key = (n, world, int(chunks), int(min_rows_per_shard), str(sfm.device))
    src = _ORDER_CACHE.get(key)
    if src is None:
        import numpy as np
        src_np = np.empty(n, dtype=np.int64)
        base = 0
        for r in range(world):
            taken = 0
            for b in plan_blocks(n, world, r, chunks, min_rows_per_shard):
                src_np[b.lo:b.hi] = np.arange(base + taken, base + taken + (b.hi - b.lo))
                taken += b.hi - b.lo
            base += caps[r]
        src = torch.from_numpy(src_np).to(sfm.device)
        if len(_ORDER_CACHE) > 16:
            _ORDER_CACHE.clear()
        _ORDER_CACHE[key] = src
    order = perm.long().index_select(0, src)
    order.zett_plan = (n, world, int(chunks), int(min_rows_per_shard))          # the block plan the order was laid out for (predict_sharded checks it)
    return order


_ORDER_CACHE = {}
_WARNED = set()


def _warn_hw_queues(world: int, on_gpu: bool) -> None:
    """One process per GPU without GPU_MAX_HW_QUEUES: the exchange and the plan-ahead then share the forward's hardware queue and
    nothing overlaps (NOTEBOOK R4.9).  Launchers call zett_amd.configure_hw_queues() before the first CUDA call; a library user who
    did not is told once."""
    import os
    import warnings
    if world > 1 and on_gpu and "GPU_MAX_HW_QUEUES" not in os.environ and "hwq" not in _WARNED:
        _WARNED.add("hwq")
        warnings.warn("zett_amd: more than one rank and GPU_MAX_HW_QUEUES is unset: the row exchange will not overlap the forward "
                      "(the HIP runtime's default 4 hardware queues are shared with RCCL). Call zett_amd.configure_hw_queues() before the "
                      "first torch.cuda call of the process, or export GPU_MAX_HW_QUEUES=8.")

GATHER_MODES = ("allgather", "fanout")


def resolve_gather_mode(mode: str, world: int) -> str:
    """"auto" -> the transport the budget picks: on xGMI a ring all-gather is bound by ONE of a GPU's seven
    links (world - 1 steps x shard / 153 GB/s), the direct fan-out uses all of them at once (shard / 153 GB/s).  From four
    ranks on the ring's exposed tail (~3 ms per output at 8 GPUs on the headline vocabulary, against 0.45 ms) is a third of a
    shard's compute, so the fan-out is the default there; at two ranks both move one shard over one link and RCCL's
    all-gather is one call instead of a grouped send/recv pair."""
    if mode == "auto":
        return "fanout" if int(world) >= 4 else "allgather"
    if mode not in GATHER_MODES:
        raise ValueError(f"gather mode must be one of {GATHER_MODES + ('auto',)}")
    return mode


class RowGather:
    """Asynchronous reassembly of row blocks into full matrices on every rank.

    `add(block, tensors, ready=None)` pads this rank's shard of the block to its nominal height and starts, per tensor, the
    exchange that puts the block's rows [block.start, block.start + world * block.per) of the full buffer on every rank;
    `finish(n_rows)` waits for everything and returns the full tensors cut to n_rows.  Blocks must be added in order; every
    block but the last has rows == world * per, so the shards of a block are adjacent rows of the final matrix and nothing
    has to be re-assembled.

    mode "allgather": one `all_gather_into_tensor` per tensor (RCCL picks the algorithm; on xGMI a ring is bound by ONE of
        the seven links: 7 steps x shard / 153 GB/s).
    mode "fanout": every rank sends its shard straight into its rows of every peer's buffer (batched isend / irecv: RCCL
        runs the seven transfers of a rank concurrently, one per xGMI link: shard / 153 GB/s, a seventh of the ring's time);
        the rank's own rows are a local copy.  Same bytes in the same places: the two modes are interchangeable, the tests
        demand identical results.

    Early start.  `tensors` = (pred_in, pred_out | None, bias) of ONE forward.  pred_in and bias are final long before the
    forward ends (the second output head still runs three GEMMs).  With `ready(name, stream)` — a callable that makes
    `stream` wait until output `name` ("in" / "bias") of that forward is complete (HipEngine.stream_wait_output) — their
    exchange is issued on a side stream behind that point instead of behind the whole forward; pred_out follows in stream
    order.  This is what hides half of the exchange when a rank has ONE block (8 GPUs on the headline vocabulary)."""

    def __init__(self, blocks: Sequence[Block], group=None, mode: str = "auto", order: Optional[torch.Tensor] = None):
        self.blocks = list(blocks)
        # `order` (affinity_order: int64 [n_rows], place in the sharded order -> vocabulary row): the blocks are gathered in the
        # sharded order and every block is scattered into vocabulary order as soon as ITS exchange has completed — on a side
        # stream, under whatever the compute stream does next (the next block's forward; for pred_in / bias of an early start,
        # the second output head): only the last block's pred_out scatter is exposed.
        self.order = order
        self.out: Optional[List[Optional[torch.Tensor]]] = None
        self._late: Optional[torch.cuda.Stream] = None
        self._pending = []           # CPU tensors (gloo tests): (tensor index, lo, hi) scattered in finish()
        self.group = group
        self.world = dist.get_world_size(group)
        self.mode = resolve_gather_mode(mode, self.world)
        self.rank = dist.get_rank(group)
        self.total = padded_rows(self.blocks, self.world)
        self.full: Optional[List[Optional[torch.Tensor]]] = None
        self._works = []
        self._keep = []
        self._side: Optional[torch.cuda.Stream] = None
        self.exposed_ms: Optional[float] = None        # set by finish(timed=True): how long the compute stream waited for the exchange
        _warn_hw_queues(self.world, torch.cuda.is_available() and dist.get_backend(group) == "nccl")

    def _exchange(self, block: Block, t: torch.Tensor, full: torch.Tensor) -> list:
        first_work = len(self._works)
        t = t[: block.hi - block.lo]
        if t.shape[0] != block.per:                  # short shard (end of the vocabulary, or a surplus rank): pad
            pad = torch.zeros((block.per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], dim=0)
        t = t.contiguous()
        dst = full[block.start: block.start + self.world * block.per]
        if self.mode == "allgather":
            self._works.append(dist.all_gather_into_tensor(dst, t, group=self.group, async_op=True))
        else:
            ranks = list(range(self.world))
            to_global = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
            ops = []
            for peer in ranks:
                if peer == self.rank:
                    continue
                ops.append(dist.P2POp(dist.isend, t, to_global(peer), self.group))
                ops.append(dist.P2POp(dist.irecv, dst[peer * block.per: (peer + 1) * block.per], to_global(peer), self.group))
            dst[self.rank * block.per: (self.rank + 1) * block.per].copy_(t)
            if ops:
                self._works.extend(dist.batch_isend_irecv(ops))
        self._keep.append(t)
        return self._works[first_work:]

    def _scatter(self, i: int, block: Block, works: list, early: bool) -> None:
        """Rows [block.start, block.start + world * per) of the gathered buffer i -> their vocabulary rows of self.out[i], behind
        the block's exchange, on a side stream (zett_scatter_rows)."""
        n = int(self.order.shape[0])
        lo, hi = block.start, min(block.start + self.world * block.per, n)
        if hi <= lo:
            return
        full, out = self.full[i], self.out[i]
        if not full.is_cuda:
            self._pending.append((i, lo, hi))
            return
        import ctypes as C

        from . import _lib
        cur = torch.cuda.current_stream(full.device)
        if early:
            side = self._side                       # (the exchange was issued from it)
        else:
            if self._late is None:
                self._late = torch.cuda.Stream(device=full.device)
            side = self._late
            side.wait_stream(cur)                   # behind the local copy / the point the exchange was issued from
        row_bytes = full[0].numel() * full.element_size() if full.dim() > 1 else full.element_size()
        with torch.cuda.stream(side):
            for w in works:
                w.wait()                            # the side stream waits for the collective; nobody else does yet
            index = full.device.index if full.device.index is not None else torch.cuda.current_device()
            if row_bytes != 4 and row_bytes % 16:   # rows zett_scatter_rows does not take (a 16-bit bias, E * element size not a multiple of 16)
                out.index_copy_(0, self.order[lo:hi], full[lo:hi])
            else:
                _lib.check(_lib.load().zett_scatter_rows(C.c_void_p(full.data_ptr() + lo * row_bytes), C.c_void_p(out.data_ptr()),
                                                         C.c_void_p(self.order.data_ptr() + lo * 8), hi - lo, row_bytes, index,
                                                         C.c_void_p(side.cuda_stream)), "zett_scatter_rows")
        # finish() must not wait for these again: the compute stream gets them through wait_stream(side), and a SECOND wait() on a
        # gloo send / recv work blocks for ever (its completion has been consumed: the two-ranks-on-one-device bench hung there)
        done = {id(w) for w in works}
        self._works = [w for w in self._works if id(w) not in done]
        full.record_stream(side)
        out.record_stream(side)

    def add(self, block: Block, tensors: Sequence[Optional[torch.Tensor]], ready: Optional[Callable] = None) -> None:
        if self.full is None:
            self.full = [None if t is None else torch.empty((self.total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in tensors]
            if self.order is not None:
                if self.order.dtype != torch.int64 or not self.order.is_contiguous():
                    raise ValueError("order must be a contiguous int64 tensor")
                n = int(self.order.shape[0])
                self.out = [None if t is None else torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in tensors]
        early = ()
        first = next(t for t in tensors if t is not None)
        if ready is not None and first.is_cuda and len(tensors) == 3 and tensors[1] is not None:
            # pred_in and bias leave behind their own completion point, on a side stream; pred_out in stream order
            if self._side is None:
                self._side = torch.cuda.Stream(device=first.device)
            ready("bias", self._side)
            ready("in", self._side)
            with torch.cuda.stream(self._side):
                for i in (0, 2):
                    works = self._exchange(block, tensors[i], self.full[i])
                    tensors[i].record_stream(self._side)
                    if self.order is not None:
                        self._scatter(i, block, works, early=True)
            early = (0, 2)
        for i, (t, full) in enumerate(zip(tensors, self.full)):
            if t is None or i in early:
                continue
            works = self._exchange(block, t, full)
            if self.order is not None:
                self._scatter(i, block, works, early=False)

    def finish(self, n_rows: int, timed: bool = False):
        ev0 = ev1 = None
        if timed and self.full is not None and any(f is not None and f.is_cuda for f in self.full):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for w in self._works:
            w.wait()           # the compute stream waits for the collective; the host does not block (nccl)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if self._late is not None:
            torch.cuda.current_stream().wait_stream(self._late)
        for i, lo, hi in self._pending:              # (CPU tensors: the exchange above has completed)
            self.out[i].index_copy_(0, self.order[lo:hi], self.full[i][lo:hi])
        self._pending.clear()
        if ev1 is not None:
            ev1.record()
            ev1.synchronize()
            self.exposed_ms = float(ev0.elapsed_time(ev1))
        self._works.clear()
        self._keep.clear()
        assert self.full is not None
        if self.order is not None:
            return tuple(self.out)
        return tuple(None if f is None else f[:n_rows] for f in self.full)


def all_gather_rows(local: torch.Tensor, n_rows: int, per: int, group=None) -> torch.Tensor:
    """All-gather row shards of equal nominal height `per`; returns the first n_rows rows."""
    world = dist.get_world_size(group)
    if local.shape[0] != per:                      # last shard(s): pad to the nominal height
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:n_rows]


class SharedTable:
    """The hoisted input-projection table computed ONCE across the ranks (SURVEY.md 8e's optional second exchange; zett_table_* of
    ABI 8).  `input_projection(in_scaler(source_embeddings[id]))` depends on the source id only; a rank of a row-sharded prediction
    otherwise computes it for the distinct ids of ITS rows — 8 340 of the headline vocabulary's 29 187 on each of 8 ranks, 2.3 x the
    work in all.  Here every rank passes the WHOLE vocabulary's surface-form matrix (as the id-affinity order needs it too), the
    ranks agree on the distinct ids without talking (the same plan on the same matrix), rank r computes rows [r * per, (r + 1) * per)
    of the folded 16-bit table, and one all-gather per buffer (2 bytes per element + 8 per row: 239 MB at the headline) completes
    it everywhere.  A table row's bits do not depend on the rank or tile that computed it: `predict` gives zett_forward's rows bit
    for bit.  f16 arithmetic (the default) only — the folded table does not exist in bf16 / f32 mode (ValueError).

        shared = SharedTable(engine, sfm_all, source_embeddings)            # every rank, same arguments
        full = predict_sharded(shared.predict(lang), sfm_all, ready=engine.stream_wait_output)

    Without a process group it is one rank's whole table (the single-GPU proxy and the tests)."""

    def __init__(self, engine, surface_forms_all: torch.Tensor, source_embeddings: torch.Tensor, group=None, only_rank: Optional[int] = None, world: Optional[int] = None,
                 buffers=None, pieces: Optional[int] = None):
        """only_rank / world: build what THAT rank of `world` ranks computes, without an exchange (the single-GPU proxy of a P-GPU step);
        buffers = (table, stats) of an earlier complete table of the same matrix: the slice is written into them instead of new ones.
        pieces: a rank's share as that many row ranges, each all-gathered (asynchronously, on the backend's stream) as soon as it is computed, so that
        the exchange of piece k runs under the computation of piece k + 1.  Table row (k * world + r) * cs + i is row i of rank r's piece k: the rows
        of one piece of all ranks are contiguous, which is what all_gather_into_tensor takes.  Default: one piece per 4 096 rows of a rank's share, at
        most four — a piece is a GEMM of that many rows, and under 4 096 rows (one round of 256 x 256 tiles at H = 4096) it would leave CUs idle; so
        the headline's 3 648 rows per rank go in one piece (nothing to overlap with), Llama-3's 16 k in four."""
        live = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size(group) if live else 1)
        self.rank = only_rank if only_rank is not None else (dist.get_rank(group) if live else 0)
        self.engine = engine
        self.id_slot, self.id_list, self.n_ids = engine.table_plan(surface_forms_all)
        share = -(-max(self.n_ids, 1) // self.world)
        exchange = live and self.world > 1 and only_rank is None
        self.pieces = int(pieces) if pieces else max(1, min(4, share // 4096))
        self.piece_rows = -(-share // self.pieces)                       # cs: rows of one piece of one rank
        self.per = self.piece_rows * self.pieces                        # rows of a rank's share (the last ranks' may hold padding)
        self.table, self.stats = buffers if buffers is not None else engine.table_buffers(self.per * self.world)
        if self.table.shape[0] < self.per * self.world or self.stats.shape[0] < self.per * self.world:
            raise ValueError("buffers are smaller than the table")
        self.ranges = []
        works = []
        cs = self.piece_rows
        for k in range(self.pieces):
            lo = min((k * self.world + self.rank) * cs, self.n_ids)
            hi = min(lo + cs, self.n_ids)
            self.ranges.append((lo, hi))
            engine.table_rows(self.id_list, lo, hi - lo, source_embeddings, self.table, self.stats)
            if exchange:
                # (the padding rows behind n_ids travel too and are never read: id_slot only names rows < n_ids.  Bytes: a dtype every backend carries)
                for buf in (self.table.view(torch.uint8), self.stats):
                    region = buf[k * self.world * cs:(k + 1) * self.world * cs]
                    mine = region[self.rank * cs:(self.rank + 1) * cs].clone()
                    works.append(dist.all_gather_into_tensor(region, mine, group=group, async_op=True))
        for w in works:
            w.wait()          # (device tensors: the current stream waits; the host does not)
        self.rows = self.ranges[0] if self.pieces == 1 else None

    def bytes_received(self) -> int:
        """what a rank receives in the table exchange"""
        return int((self.world - 1) * self.per * (self.table.shape[1] * self.table.element_size() + 8))

    def predict(self, lang_index: int) -> Callable:
        return lambda rows: self.engine.forward_table(rows, self.table, self.stats, self.id_slot, lang_index)


def predict_sharded(predict: Callable, target_surface_forms: torch.Tensor, group=None, chunks: int = 2,
                    ready: Optional[Callable] = None, mode: str = "auto", prepare: Optional[Callable] = None,
                    order: Optional[torch.Tensor] = None, min_rows_per_shard: int = 4096):
    """Run `predict(rows) -> (pred_in, pred_out | None, bias)` on this rank's rows and return the full result on every
    rank.  The vocabulary is processed in `chunks` row blocks whose exchange overlaps the next block's forward (module
    docstring); chunks = 1 is the plain shard-then-gather.  `ready` (RowGather: early start of pred_in / bias) and `mode`
    ("auto" | "allgather" | "fanout": resolve_gather_mode) are handed to the RowGather.

    `predict` is typically ``lambda rows: engine.forward(rows, source_embeddings, lang)`` with
    ``ready=engine.stream_wait_output``.  Without an initialised process group this is just ``predict(target_surface_forms)``.

    `order` (``affinity_order(...)``: int64 [n] on the device) — shard the rows in THAT order instead of vocabulary order: rank r
    computes rows order[lo:hi] of every block, and every gathered block is scattered to its vocabulary rows behind ITS exchange, on a
    side stream (RowGather, zett_scatter_rows).  Same rows, same bits.  An order from affinity_order carries the block plan it was laid out
    for (`chunks`, `min_rows_per_shard`): sharding it for another plan raises.

    `prepare(rows, stream)` (``engine.prepare``: zett_forward_prepare) — with more than one block per rank, the plan of block
    k + 1 is enqueued as soon as block k's forward is, on the engine's own stream behind a side stream that holds nothing but
    the surface forms' readiness: block k + 1's zett_forward then does not wait on the host for block k's kernels.  Needs the
    surface forms as the int32, contiguous tensor on the engine's device that `predict` hands to the engine unchanged.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return predict(target_surface_forms)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(target_surface_forms.shape[0])
    blocks = plan_blocks(n, world, rank, chunks, min_rows_per_shard)
    if not blocks:
        return predict(target_surface_forms)
    if order is not None:
        if order.shape[0] != n:
            raise ValueError("order must hold one entry per row")
        plan = getattr(order, "zett_plan", None)
        if plan is not None and plan != (n, world, int(chunks), int(min_rows_per_shard)):
            # still a permutation, so the result would be right — but the groups would straddle the shards and the affinity be lost silently
            raise ValueError(f"order was built by affinity_order for (rows, world, chunks, min_rows_per_shard) = {plan}, "
                             f"predict_sharded is sharding for {(n, world, int(chunks), int(min_rows_per_shard))}")
        order = order.to(torch.int64).contiguous()
        vocabulary_order = target_surface_forms
        target_surface_forms = vocabulary_order.index_select(0, order)
    gather = RowGather(blocks, group, mode, order=order)

    def rows_of(b):
        # (more ranks than rows in the block: compute one dummy row, contribute none)
        return target_surface_forms[b.lo:b.hi] if b.hi > b.lo else target_surface_forms[:1]

    ahead = None
    if prepare is not None and len(blocks) > 1 and target_surface_forms.is_cuda:
        ahead = torch.cuda.Stream(device=target_surface_forms.device)
        ahead.wait_stream(torch.cuda.current_stream(target_surface_forms.device))      # the surface forms are complete behind this point
    for k, b in enumerate(blocks):
        rows = rows_of(b)
        outs = predict(rows)
        if ahead is not None and k + 1 < len(blocks):
            prepare(rows_of(blocks[k + 1]), ahead)
        if b.hi - b.lo == 0:
            outs = tuple(None if t is None else t[:0] for t in outs)
        gather.add(b, outs, ready)
    return gather.finish(n)
