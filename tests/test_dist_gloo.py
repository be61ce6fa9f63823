"""Data-parallel path on CPU: world_size-2 gloo processes.  Each rank takes a contiguous slice of the train rows and of
the exemplar rows, scales its loss terms by the GLOBAL sub-batch sizes, and the flat gradient buffer is SUM-reduced with
ader_amd.dist.allreduce_flat; the result must equal the single-process full-batch gradient (ADER.py:120-121,136-137 are
means over rows).  The oracle provides the local gradients here (no GPU in this container)."""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ader_amd import dist as adist
from ader_amd.engine import param_layout
from oracle import ader_ref_cpu as R

ITEMS, T, H, L, HEADS, N, NP = 120, 12, 16, 2, 2, 100, 80
N_TRAIN, N_EX, LAM = 11, 5, 0.7


def _problem():
    rs = np.random.RandomState(0)
    params = R.init_params(ITEMS, T, H, L, seed=1, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    for k in params:
        if k.endswith("_b"):
            params[k] = torch.randn(params[k].shape, generator=g, dtype=torch.float64) * 0.1
    seq = np.zeros((N_TRAIN + N_EX, T), dtype=np.int64)
    for b in range(len(seq)):
        ln = rs.randint(1, T + 1)
        seq[b, T - ln:] = rs.randint(1, N + 1, size=ln)
    pos = rs.randint(1, N + 1, size=N_TRAIN)
    teacher = torch.from_numpy(rs.standard_normal((N_EX, NP)))
    return params, seq, pos, teacher


def _flat(grads, layout, total):
    buf = torch.zeros(total, dtype=torch.float64)
    for k, (off, shp) in layout.items():
        buf[off:off + grads[k].numel()] = grads[k].reshape(-1)
    return buf


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, seq, pos, teacher = _problem()
    layout, total = param_layout(ITEMS, T, H, L)
    lo, hi = adist.shard_bounds(N_TRAIN, world, rank)
    elo, ehi = adist.shard_bounds(N_EX, world, rank)
    local_seq = np.concatenate([seq[lo:hi], seq[N_TRAIN + elo:N_TRAIN + ehi]])
    _, grads = R.loss_and_grads(params, local_seq, pos[lo:hi], N, L, HEADS, ex_logits=teacher[elo:ehi], lambda_=LAM,
                                training=True, rate=0.0, n_train_global=N_TRAIN, n_ex_global=N_EX)
    buf = _flat(grads, layout, total)
    adist.allreduce_flat(buf, (N + 1) * H, layout["pos"][0], bucket_elems=257)    # several ragged buckets
    if rank == 0:
        torch.save(buf, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gradient_equals_full_batch():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "g.pt")
        mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        got = torch.load(out)
    params, seq, pos, teacher = _problem()
    layout, total = param_layout(ITEMS, T, H, L)
    _, grads = R.loss_and_grads(params, seq, pos, N, L, HEADS, ex_logits=teacher, lambda_=LAM, training=True, rate=0.0)
    ref = _flat(grads, layout, total)
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-12)
    # rows above max_item are skipped by the exchange and must be zero anyway
    emb_off = layout["emb"][0]
    assert torch.all(got[emb_off + (N + 1) * H: layout["pos"][0]] == 0)


def test_shard_bounds_and_buckets():
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [adist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    r = adist.bucket_ranges(10_000, 1000, 300)
    assert r == [(0, 300), (300, 600), (600, 900), (900, 1000)]


def test_dropout_counter_is_keyed_by_global_row():
    """A shard that starts at global row r0 draws the masks of rows r0.. of the full batch (Engine.row0 / oracle row0)."""
    params, seq, pos, teacher = _problem()
    p32 = {k: v.float() for k, v in params.items()}
    full = R.forward_rep(p32, seq, L, HEADS, training=True, rate=0.4, seed=5, step=3)
    part = R.forward_rep(p32, seq[6:11], L, HEADS, training=True, rate=0.4, seed=5, step=3, row0=6)
    assert torch.equal(full[6:11], part)


def test_parameter_layout_is_aligned_and_disjoint():
    layout, total = param_layout(1000, 50, 150, 2)
    spans = sorted((off, off + int(np.prod(shp))) for off, shp in layout.values())
    assert all(off % 64 == 0 for off, _ in spans)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    assert spans[-1][1] <= total and total % 64 == 0


# ----------------------------------------------------------------------------------------------- sharded eval / herding
class _FakeModel:
    """rank_targets / herding stand-ins that are pure functions of their inputs (the sharding logic is what is tested)."""

    class _Eng:
        H = 4

        def herding_select(self, seq_rows, offs, quota, max_item):
            n = len(seq_rows)
            sel = np.zeros(n, dtype=np.int64)
            cnt = np.zeros(len(quota), dtype=np.int32)
            for g in range(len(quota)):
                rows = seq_rows[offs[g]:offs[g + 1]]
                order = np.argsort(-(rows.sum(axis=1) % 7), kind="stable")[:quota[g]]
                sel[offs[g]:offs[g] + len(order)] = order
                cnt[g] = len(order)
            return sel, cnt

        def teacher_logits(self, rows, max_item):
            return torch.zeros(len(rows), 3)

    engine = _Eng()

    def rank_targets(self, seq, pos, max_item):
        return [int((int(np.asarray(s).sum()) * 31 + int(p)) % 50) for s, p in zip(seq, pos)]


def _host_problem():
    rs = np.random.RandomState(3)
    sessions = [rs.randint(1, 40, size=rs.randint(2, 9)).tolist() for _ in range(173)]
    return sessions


def _host_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    from ader_amd.data import Evaluator
    from ader_amd.exemplar import ExemplarGenerator
    random.seed(0)
    np.random.seed(0)
    ev = Evaluator(_host_problem(), False, 10, 16, 40, "test", _FakeModel(), None, shard=(rank, world))
    ev.evaluate(1)
    gen = ExemplarGenerator(_host_problem(), 60, False, 16, 10, 0.0, 40, shard=(rank, world))
    saved = gen.herding_selection(None, _FakeModel())
    torch.save({"ranks": ev.ranks, "res": ev.results(), "saved": saved, "rows": np.asarray(gen.store.rows)}, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluator_and_herding_equal_single_process():
    """evaluation batches and herding label groups are independent units; with W = 2 every rank must end with
    the single-process rank list (same order => same float64 metric sums) and the single-process exemplar rows."""
    import random
    from ader_amd.data import Evaluator
    from ader_amd.exemplar import ExemplarGenerator
    random.seed(0)
    np.random.seed(0)
    ev = Evaluator(_host_problem(), False, 10, 16, 40, "test", _FakeModel(), None)
    ev.evaluate(1)
    gen = ExemplarGenerator(_host_problem(), 60, False, 16, 10, 0.0, 40)
    saved = gen.herding_selection(None, _FakeModel())
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r%d.pt")
        mp.spawn(_host_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        for r in range(2):
            got = torch.load(out % r, weights_only=False)
            assert got["ranks"] == ev.ranks and got["res"] == ev.results()
            assert got["saved"] == saved and np.array_equal(got["rows"], np.asarray(gen.store.rows))


def test_split_groups_is_contiguous_and_balanced():
    sizes = [5, 1, 1, 30, 2, 2, 2, 9, 1]
    for w in (1, 2, 3, 4, 8):
        ch = adist.split_groups(sizes, w)
        assert ch[0][0] == 0 and ch[-1][1] == len(sizes)
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:])) and all(lo <= hi for lo, hi in ch)


def test_pack_counts_and_global_ids_on_the_host():
    """Host side of the packed catalog exchange (no GPU): dist.global_ids_host lays out every rank's (input positions, labels) of a
    global batch exactly as the ranks feed them (shard_rows padding included), and engine.pack_counts_host counts, per (owner,
    destination), the rows that travel -- checked against a brute-force loop."""
    import numpy as np
    from ader_amd import dist as adist
    from ader_amd.engine import pack_counts_host           # (pure numpy; importing the package does not load the HIP library)
    ns = {"pack_counts_host": pack_counts_host}
    rs = np.random.RandomState(0)
    n, T, W, S = 37, 5, 4, 16
    seq = rs.randint(0, 70, size=(n, T)).astype(np.int32)
    pos = rs.randint(1, 70, size=n).astype(np.int32)
    ids = adist.global_ids_host(seq, pos, W)
    per = -(-n // W)
    assert ids.shape == (W, per * T + per)
    for r in range(W):
        lo, hi = adist.shard_bounds(n, W, r)
        assert ids[r, :(hi - lo) * T].tolist() == seq[lo:hi].reshape(-1).tolist()
        assert ids[r, per * T:per * T + (hi - lo)].tolist() == pos[lo:hi].tolist()
        assert not ids[r, (hi - lo) * T:per * T].any() and not ids[r, per * T + (hi - lo):].any()      # padding rows: id 0, label 0
    C_all, C_pos = ns["pack_counts_host"](ids, per * T, S)
    for o in range(W):
        for d in range(W):
            own = [(min((i - 1) // S, W - 1) if i > 0 else -1) for i in ids[d].tolist()]
            assert C_all[o][d] == sum(1 for x in own if x == o)
            assert C_pos[o][d] == sum(1 for x in own[:per * T] if x == o)


# ---------------------------------------------------------------------------------------------- first-contact guard
def _guard_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = adist.guard
    res = {}
    g.start()
    g.check("site_a", "all_gather", (4, 3), torch.float32)                        # identical on both ranks: passes
    g.check("site_b", "all_to_all(uneven)", (5,), torch.float32, ([2, 3], [2, 1]) if rank == 0 else ([1, 4], [3, 4]))   # mirrored splits
    res["ok_log"] = len(g.stop())
    for name, args0, args1 in (
            ("shape", ("site_c", "all_gather", (4, 3), torch.float32), ("site_c", "all_gather", (4, 5), torch.float32)),
            ("site", ("site_d", "all_reduce", (7,), torch.float32), ("site_e", "all_gather", (7,), torch.float32)),
            ("splits", ("site_f", "all_to_all(uneven)", (5,), torch.float32, ([2, 3], [2, 1])),
             ("site_f", "all_to_all(uneven)", (5,), torch.float32, ([1, 4], [9, 4])))):
        g.start()
        try:
            g.check(*(args0 if rank == 0 else args1))
            res[name] = None
        except RuntimeError as e:
            res[name] = str(e)
        g.stop()
    g.check("off", "all_gather", (1,), torch.float32)                             # off: a no-op (no collective is issued)
    torch.save(res, out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_collective_guard_names_the_rank_and_the_call_site_on_a_mismatch():
    """dist.CollectiveGuard (what bench.py --gpus N runs around its first step): matching collectives pass, a rank that arrives
    with another shape, at another call site, or with split sizes that do not mirror its peer's raises on EVERY rank, before the
    data collective would be issued, with the ranks and the call site in the message."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "g.pt")
        mp.spawn(_guard_worker, args=(2, port, out), nprocs=2, join=True)
        res = [torch.load(out + ".%d" % r) for r in range(2)]
    for r in range(2):
        assert res[r]["ok_log"] == 2
        assert "site_c" in res[r]["shape"] and "rank %d" % r in res[r]["shape"] and "[4, 5]" in res[r]["shape"].replace("(", "[").replace(")", "]")
        assert "diverged" in res[r]["site"] and "site_d" in res[r]["site"] and "site_e" in res[r]["site"]
        assert "site_f" in res[r]["splits"] and "expects" in res[r]["splits"]


def test_dense_path_picks_plain_or_overlapped_allreduce_by_bytes():
    """DataParallel.early_pays (no GPU, no process group): the overlapped table all-reduce forces the sparse rows to travel as
    per-position rows -- (W - 1) x positions x (H + 1) x 4 bytes per rank -- so it is taken only where the table outweighs them:
    at the headline catalog, not at the shipped datasets' (DESIGN.md section 5's byte model)."""
    from ader_amd import dist as adist

    class _Eng:
        T, H = 50, 150

        def __init__(self, rows):
            self._act = {"B": rows}

    dp = adist.DataParallel.__new__(adist.DataParallel)
    dp.world, dp.early = 8, "auto"
    assert dp.early_pays(_Eng(512), 1_000_000)                  # cfg-S: 1.2 GB of all-reduce traffic against 0.1 GB of rows
    assert not dp.early_pays(_Eng(614), 25_750)                 # YOOCHOOSE (configs[3]): 31 MB against 8 x 18 MB
    assert not dp.early_pays(_Eng(399), 43_105)                 # DIGINETICA
    dp.world = 2
    assert dp.early_pays(_Eng(614), 25_750)                     # two ranks: one peer's rows (18 MB) against 31 MB
    dp.early = "never"
    assert not dp.early_pays(_Eng(512), 1_000_000)
    dp.early = "always"
    assert dp.early_pays(_Eng(614), 25_750)


def test_shape_roofline_accounting():
    """bench.shape_roofline: the floor is the sum of its parts, executed work exceeds credited work by the row padding and the three
    passes, and the YOOCHOOSE shape lands where the round-5 review priced it (~75 us)."""
    import bench
    r = bench.shape_roofline(25750, 512, 102, int(0.9 * 25750), 3078, 0.384)
    assert abs(sum(r["floor_parts_ms"].values()) - r["floor_ms"]) < 1e-4 and 0.065 < r["floor_ms"] < 0.080
    assert abs(r["frac"] - r["floor_ms"] / 0.384) < 1e-3
    assert r["logit_flops_executed"] > 3 * r["logit_flops_credited"] and r["session_flops_executed"] == 3 * r["session_flops_credited"]
    big = bench.shape_roofline(1_000_000, 512, 128, 900_000, 640 * 50, 3.0)
    assert big["floor_ms"] > 2.0 and big["floor_parts_ms"]["logit_mfma"] > big["floor_parts_ms"]["hbm"] > big["floor_parts_ms"]["session_mfma"]
