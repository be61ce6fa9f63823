"""Data parallelism for the train step: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl").

The reference is single-device (main.py:96,120,143) and has no collective; the north star adds DP only.  Rows of a
minibatch are independent through forward/backward (LayerNorm is per row; the losses are means over rows,
ADER.py:93,120-121,136-137), so each rank takes a contiguous slice of the train rows AND of the exemplar rows, scales
its local loss terms by the GLOBAL sub-batch sizes (1/B_train,global and lambda/B_ex,global) and the dense gradients are
SUM-reduced; every rank then applies the identical Adam update.  Dropout counters are keyed by the global row index
(Engine.row0) so the masks do not depend on the number of ranks.

Exchange schemes (all give the same update as a single process on the global batch):
  * catalog-sharded table (Engine.dp_mode = "catalog", Engine._train_step_catalog; bench.py --gpus N): every rank OWNS 1/W of the
    table rows; only the rows the inputs touch, the bf16 representations, per-row softmax partials and the per-position
    gradient rows travel.  Nothing proportional to the table size is exchanged.
  * sharded table update (default with bf16 logits, Engine._fused_table_adam_sharded): the 600 MB dense table gradient is
    never exchanged.  Ranks all-gather the INPUTS of the table-gradient product (~16 MB each), each updates its row shard
    of the table for the global batch inside the fused gradient+Adam kernel, and the updated rows are all-gathered
    (half the bytes of an all-reduce, and Adam's table traffic drops by the number of ranks).
  * dense all-reduce (f32 logits, distilled steps): below.

Exchange: the gradient lives in one flat buffer (ader_amd.engine.param_layout).  It is reduced in a few large buckets
(xGMI rings are per-link bound: few, large collectives) -- the table rows [0, max_item] first, then the small block
parameters.  `backend="gloo"` runs the same logic on CPU tensors for the world_size-2 tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="nccl"):
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local


def pad_rows(x, n, fill=0):
    """x (numpy array or torch tensor, first dimension = rows) padded with `fill` rows to n rows (unchanged if it has them).
    Padding rows are id-0 sessions with label 0 / teacher row -1: weight 0 in every loss kernel (csrc/logits.hip:k_build_rowinfo)."""
    pad = n - len(x)
    if pad <= 0:
        return x
    if isinstance(x, torch.Tensor):
        return torch.cat([x, torch.full((pad,) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=x.device)])
    x = np.asarray(x)
    return np.concatenate([x, np.full((pad,) + x.shape[1:], fill, dtype=x.dtype)])


class CollectiveGuard:
    """First-contact insurance for a multi-rank run (the 8-GPU node is the driver's: nothing in the build container can execute
    RCCL with more than one rank).  While `on`, every collective of the engine / the exchange hooks first announces itself --
    (call site, kind, shape, dtype[, split sizes]) -- through an all_gather_object and every rank compares: a rank that reaches
    another call site, or the same one with another shape or with split sizes that do not mirror its peers', raises RuntimeError
    naming the rank and the call site BEFORE the data collective is issued (where a mismatch would hang or corrupt).  The log of
    one guarded step is what bench.py prints per rank ahead of its timed region.  Off (the default), a check costs one attribute
    test."""

    def __init__(self):
        self.on, self.log, self.group = False, [], None

    def start(self, group=None):
        self.on, self.log, self.group = True, [], group

    def stop(self):
        self.on = False
        return self.log

    def check(self, site, kind, shape, dtype, splits=None):
        if not self.on:
            return
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        rec = (site, kind, tuple(int(d) for d in shape), str(dtype).replace("torch.", ""),
               None if splits is None else (tuple(int(x) for x in splits[0]), tuple(int(x) for x in splits[1])))
        k = len(self.log)
        self.log.append(rec)
        peers = [None] * world
        dist.all_gather_object(peers, rec, group=self.group)
        for j, o in enumerate(peers):
            if o[:2] != rec[:2]:
                raise RuntimeError("collective #%d of the step: rank %d is at %s (%s), rank %d at %s (%s) -- the ranks diverged"
                                   % (k, rank, rec[0], rec[1], j, o[0], o[1]))
            if splits is None and o[2:4] != rec[2:4]:
                raise RuntimeError("collective #%d (%s at %s): rank %d passes %s %s, rank %d %s %s" % (k, kind, site, rank, rec[3],
                                                                                                          rec[2], j, o[3], o[2]))
            if splits is not None and (o[4][0][rank] != rec[4][1][j] or rec[4][0][j] != o[4][1][rank]):
                raise RuntimeError("collective #%d (%s at %s): rank %d sends %d rows to rank %d, which expects %d; it expects %d from "
                                   "rank %d, which sends %d" % (k, kind, site, rank, rec[4][0][j], j, o[4][1][rank], rec[4][1][j], j,
                                                                 o[4][0][rank]))

    def describe(self, rank):
        return ["[rank %d] collective %2d: %-22s %-28s %s %s%s" % (rank, i, kind, site, dt, list(shape),
                                                                    "" if sp is None else "  send %s recv %s" % (list(sp[0]), list(sp[1])))
                for i, (site, kind, shape, dt, sp) in enumerate(self.log)]


guard = CollectiveGuard()


def shard_bounds(n, world, rank):
    """Contiguous, near-equal split of n rows: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(x, world, rank, fill=0):
    """Rank `rank`'s slice of the rows of x (numpy array or torch tensor, first dimension = rows), padded with `fill` to
    ceil(n / world) rows: every rank of a step holds the SAME number of rows (>= 1 when n >= 1), so gathered tensors have equal
    shapes on every rank and no rank skips a collective.  Padding rows are id-0 sessions with label 0 (ex_trow -1): the loss
    kernels give them weight 0 (csrc/logits.hip:k_build_rowinfo)."""
    n = len(x)
    lo, hi = shard_bounds(n, world, rank)
    per = -(-n // world)
    part = x[lo:hi]
    pad = per - (hi - lo)
    if pad <= 0:
        return part
    if isinstance(x, torch.Tensor):
        filler = torch.full((pad,) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=x.device)
        return torch.cat([part, filler])
    import numpy as np
    x = np.asarray(x)
    return np.concatenate([x[lo:hi], np.full((pad,) + x.shape[1:], fill, dtype=x.dtype)])


def global_ids_host(seq, pos, world):
    """ids of every rank's positions for one step of the catalog-sharded scheme, on the HOST: [world, n_all] int32 with
    n_all = rows * T + rows per rank (its input positions, then its labels), rows = ceil(n / world) after shard_rows' padding.
    seq [n, T] / pos [n]: the GLOBAL batch (numpy) -- every rank builds the same batches, so every rank can form this without any
    communication; Engine.train_step(ids_host=...) then needs no device-to-host synchronisation for the packed row exchange."""
    seq, pos = np.asarray(seq), np.asarray(pos)
    return np.stack([np.concatenate([shard_rows(seq, world, r).reshape(-1), shard_rows(pos, world, r)]).astype(np.int32)
                     for r in range(world)])


def bucket_ranges(total, table_elems, bucket_elems):
    """[(lo, hi)] covering [0, total): the used table rows in large buckets, then everything after the table."""
    out, lo = [], 0
    while lo < table_elems:
        hi = min(table_elems, lo + bucket_elems)
        out.append((lo, hi))
        lo = hi
    return out


def allreduce_flat(grad, used_table_elems, table_span, bucket_elems=64 << 20, group=None):
    """SUM-reduce grad[0:used_table_elems] (table rows that can be non-zero) and grad[table_span:] (all other
    parameters) in place.  Rows above max_item are zero on every rank and are skipped."""
    for lo, hi in bucket_ranges(grad.numel(), used_table_elems, bucket_elems):
        guard.check("dist.allreduce_flat:table", "all_reduce", (hi - lo,), grad.dtype)
        dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, group=group)
    if table_span < grad.numel():
        guard.check("dist.allreduce_flat:small", "all_reduce", (grad.numel() - table_span,), grad.dtype)
        dist.all_reduce(grad[table_span:], op=dist.ReduceOp.SUM, group=group)


def gather_lists(local, world, group=None):
    """[list of rank 0, list of rank 1, ...] of picklable host objects (ranks, selected indices): tiny payloads."""
    if world == 1:
        return [local]
    out = [None] * world
    dist.all_gather_object(out, local, group=group)
    return out


def split_groups(sizes, world):
    """Contiguous split of label groups into `world` chunks of near-equal ROW counts: [(g_lo, g_hi)] per rank
    (herding groups are independent units)."""
    n = len(sizes)
    total = int(sum(sizes))
    bounds, acc, g = [0], 0, 0
    for r in range(1, world):
        target = total * r / world
        while g < n and acc + sizes[g] / 2.0 <= target:
            acc += sizes[g]
            g += 1
        bounds.append(g)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


class DataParallel:
    """Installs the gradient exchange into an Engine.  Usage per step:
         dp.set_step(n_train_local_offset)  (row offset of this rank's first row in the global batch)
         engine.train_step(seq_local, pos_local, max_item, lr, n_train_global=..., n_ex_global=..., ...)"""

    def __init__(self, engine, rank=None, world=None, group=None):
        r, w, _ = env_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.engine = engine
        self.group = group
        self.max_item = engine.item_num
        if self.world > 1:
            assert engine.dp_world == self.world and engine.dp_rank == self.rank, \
                "construct the Engine with dp_rank/dp_world (the table layout is sharded at allocation time)"
            engine.dp_group = group
            engine.grad_hook = self._exchange          # dense path (f32 logits / distilled steps / dp_sharded = False)
            engine.grad_early_hook = self._early       # ... its table part starts right after the logits backward
            # identical replicas: broadcast rank 0's state once
            for t in (engine.theta, engine.adam_m, engine.adam_v):
                dist.broadcast(t, src=0, group=group)
            engine.refresh_shadow()          # bf16 copies derived from theta

    def set_rows(self, global_row0, max_item, ex_row0=None):
        """global_row0: index of this rank's first train row in the global batch; ex_row0: global index of its first exemplar row
        (= n_train_global + offset of the rank's exemplar slice; the global batch is [train rows | exemplar rows], main.py:229).
        The dropout counters of every site are keyed by these global rows, so W ranks draw exactly the masks of one process."""
        self.engine.row0 = int(global_row0)
        self.engine.row0_ex = int(ex_row0) if ex_row0 is not None else 0
        self.engine._ex_row0_set = ex_row0 is not None
        self.max_item = int(max_item)

    def _early(self, eng, max_item):
        """Dense path, overlapped with backward (north star: "RCCL all-reduce of dense gradients over xGMI overlapped with
        backward"): the logits backward has just written the dense term of the table gradient -- 99.8 % of the gradient bytes,
        produced FIRST in backward -- so its bucketed all-reduce is started now, asynchronously, and runs on RCCL's stream while
        the transformer blocks' backward computes.  The sparse input-embedding rows do not exist yet: every rank keeps them as
        per-position rows and _exchange adds the rows of ALL ranks after the reduction (they are 15 MB per rank against 600 MB)."""
        H = eng.H
        self.max_item = int(max_item)
        if not self.early_pays(eng, max_item):
            return None
        works = []
        for lo, hi in bucket_ranges(eng.grad.numel(), (self.max_item + 1) * H, self.bucket_elems):
            guard.check("DataParallel._early:table", "all_reduce(async)", (hi - lo,), eng.grad.dtype)
            works.append(dist.all_reduce(eng.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    early = "auto"      # "auto" | "always" | "never": see early_pays

    def early_pays(self, eng, max_item):
        """Dense path: is the early (overlapped) all-reduce of the table gradient worth what it costs?  It forces the sparse
        input-embedding rows to travel as per-position rows -- every rank receives (W - 1) x positions x (H + 1) x 4 bytes on top of
        the all-reduce's ~2 x table bytes.  At the headline catalog (1.2 GB of all-reduce traffic against ~0.1 GB of rows) the
        overlap wins; at the shipped datasets' catalogs (YOOCHOOSE: 31 MB against 8 x 18 MB of rows at 8 ranks) the rows cost more
        than the whole table, so every rank scatters its rows into its gradient first and ONE dense all-reduce follows the backward
        (_exchange's plain branch).  DESIGN.md section 5 has the byte model."""
        if self.early != "auto":
            return self.early == "always"
        positions = int(eng._act["B"]) * eng.T if getattr(eng, "_act", None) else 0
        rows_bytes = (self.world - 1) * positions * (eng.H + 1) * 4
        table_bytes = 2 * (int(max_item) + 1) * eng.H * 4
        return rows_bytes < table_bytes

    bucket_elems = 64 << 20          # 256 MB buckets: xGMI rings are per-link bound -- few, large collectives

    def _exchange(self, eng):
        from ._lib import call, ptr
        guard.check("DataParallel._exchange:loss", "all_reduce", eng.loss.shape, eng.loss.dtype)
        dist.all_reduce(eng.loss, group=self.group)
        H = eng.H
        table_span = eng.layout["pos"][0]
        works, eng._early = eng._early, None
        if works is None:
            allreduce_flat(eng.grad, (self.max_item + 1) * H, table_span, group=self.group)
            return
        seq, dx = eng._dp_rows
        eng._dp_rows = None
        W = self.world
        ids_g = torch.empty(W * seq.numel(), dtype=seq.dtype, device=seq.device)
        rows_g = torch.empty(W * dx.numel(), dtype=dx.dtype, device=dx.device)
        guard.check("DataParallel._exchange:ids", "all_gather", (seq.numel(),), seq.dtype)
        dist.all_gather_into_tensor(ids_g, seq.contiguous().view(-1), group=self.group)      # small bucket: ids + per-position rows ...
        guard.check("DataParallel._exchange:rows", "all_gather", (dx.numel(),), dx.dtype)
        dist.all_gather_into_tensor(rows_g, dx.contiguous().view(-1), group=self.group)
        if table_span < eng.grad.numel():
            guard.check("DataParallel._exchange:small", "all_reduce", (eng.grad.numel() - table_span,), eng.grad.dtype)
            dist.all_reduce(eng.grad[table_span:], op=dist.ReduceOp.SUM, group=self.group)   # ... and every non-table parameter
        for w in works:
            w.wait()                                                                  # table buckets (started before the blocks backward)
        # every rank adds the SAME gathered rows in the SAME order (bucketed by id, position order inside a table row): the reduced
        # gradient stays bit-identical across the replicas -- float atomics here let theta / m / v drift apart between ranks
        lab0 = eng.buf("dp_lab0", (1,), torch.int32, zero=True)
        ids_s, order, sp_start, _, _, _, _ = eng._sparse_lists(ids_g, lab0, self.max_item)
        call("ader_scatter_rows_ordered", ptr(ids_s), ptr(order), ptr(sp_start), sp_start.numel() - 1, ptr(rows_g), H, eng.V,
             float(np.sqrt(np.float32(H))), ptr(eng.gradient("emb")), eng._stream())
