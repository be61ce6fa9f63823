"""Host feeders (ader_amd.data) against golden vectors produced by the reference's own util.py
(tests/golden/make_golden.py).  Bit-exact: integer/index work."""
import json
import os
import random
import zlib

import numpy as np
import pytest

from ader_amd import data as D


def unragged(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def test_sampler_prefix_label_batches(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    sess = unragged(g["in_flat"], g["in_off"])
    random.seed(0)
    np.random.seed(0)
    sm = D.Sampler(sess, 50, 16, is_subseq=False)
    assert sm.prepared_data == unragged(g["prepared_flat"], g["prepared_off"])
    assert sm.data_indices == g["indices0"].tolist()
    assert sm.batch_num() == int(g["batch_num"])
    seqs, poss = [], []
    for s in sm.prepared_data:
        if len(s) >= 2:
            a, b = sm.label_generator(s)
            seqs.append(a)
            poss.append(int(b))
    assert np.array_equal(np.stack(seqs), g["lg_seq"])
    assert np.array_equal(np.array(poss), g["lg_pos"])
    nb = sm.batch_num()
    bseq, bpos, bsz = [], [], []
    for _ in range(2 * nb + nb // 2):
        seq, pos = sm.sampler()
        bsz.append(len(seq))
        bseq.extend(seq)
        bpos.extend(pos)
    assert bsz == g["batches_size"].tolist()
    assert np.array_equal(np.stack(bseq), g["batches_seq"])
    assert np.array_equal(np.array(bpos), g["batches_pos"])


def test_sampler_subseq_and_exemplar_paths(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    sess = unragged(g["in_flat"], g["in_off"])
    random.seed(3)
    sm2 = D.Sampler(sess, 50, 7, is_subseq=True)
    assert sm2.data_indices == g["sub_indices0"].tolist()
    assert len(sm2.prepared_data) == int(g["sub_prepared_n"])
    random.seed(5)
    ex_s = unragged(g["ex_in_flat"], g["ex_in_off"])
    ex = [[s, l.tolist()] for s, l in zip(ex_s, g["ex_in_logits"])]
    sm3 = D.Sampler([], 50, 2)
    sm3.add_exemplar(ex)
    eseq, epos, elog, esz = [], [], [], []
    for _ in range(2 * sm3.batch_num() + 1):
        s, p, l = sm3.exemplar_sampler()
        esz.append(len(s))
        eseq.extend(s)
        epos.extend(p)
        elog.extend(l)
    assert esz == g["ex_sizes"].tolist()
    assert np.array_equal(np.stack(eseq), g["ex_seq"])
    assert np.array_equal(np.array(epos), g["ex_pos"])
    assert np.array_equal(np.array(elog, dtype=np.float64), g["ex_logits"])


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 1000, 65537, 131072, 300001])
def test_native_shuffle_is_random_shuffle(n):
    """ader_host_shuffle (csrc/host_feed.hip) against random.shuffle itself: same permutation and the same generator state after,
    from several points of the stream (incl. across a regeneration of the Mersenne-Twister block)."""
    for seed, burn in ((0, 0), (7, 623), (123456789, 1000)):
        random.seed(seed)
        for _ in range(burn):
            random.random()
        st0 = random.getstate()
        want = list(range(n))
        random.shuffle(want)
        st_want = random.getstate()
        random.setstate(st0)
        a = np.arange(n, dtype=np.int64)
        D.shuffle_like_python(a)
        assert a.tolist() == want
        assert random.getstate() == st_want


def test_epoch_planned_feeder_yields_the_reference_batches():
    """Sampler.to_device plans a whole epoch at once (shuffled order, invalid rows dropped per batch, one upload) and serves batches
    as slices: same batches, same index arrays and the same consumption of the `random` stream as the per-batch host path that the
    golden vectors above pin -- across three reshuffles, with sessions too short to train on in the data.  (The tensor device is
    the CPU here; tests/test_gpu_shim.py runs it on the GPU.)"""
    import random
    import torch
    rs = np.random.RandomState(4)
    sessions = [rs.randint(1, 90, size=rs.randint(1, 30)).tolist() for _ in range(157)]
    out = []
    for dev in (False, True):
        random.seed(11)
        smp = D.Sampler(sessions, 50, 16)
        if dev:
            smp.to_device(torch.device("cpu"))
        got = []
        for _ in range(3 * smp.batch_num() + 2):
            seq, pos, idx = smp.next_exemplar_batch()
            if dev:
                assert seq.dtype == torch.int32 and seq.is_contiguous() and pos.is_contiguous()
                assert smp.last_idx_dev.dtype == torch.int32 and np.array_equal(smp.last_idx_dev.numpy(), idx)
                seq, pos = seq.numpy(), pos.numpy()
            got.append((np.array(seq), np.array(pos), np.array(idx)))
        out.append((got, random.random()))
    assert out[0][1] == out[1][1]
    for (a, b, c), (d, e, f) in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, d) and np.array_equal(b, e) and np.array_equal(c, f)


def test_evaluators_over_an_unchanged_list_share_the_packed_rows_not_the_shuffle():
    """One Evaluator is built per epoch over the same validation list (main.py:264-266): the second one reuses the first one's
    packed rows, but its Sampler still shuffles a fresh index list -- same batches and same `random` stream as without the cache."""
    rs = np.random.RandomState(8)
    data = [rs.randint(1, 50, size=rs.randint(1, 12)).tolist() for _ in range(300)]
    got = []
    for cached in (True, False):
        random.seed(3)
        D.Evaluator._packed.clear()
        seqs = []
        for epoch in range(3):
            if not cached:
                D.Evaluator._packed.clear()
            ev = D.Evaluator(data, True, 20, 64, 49, "valid", None, None)
            smp = ev.evaluate_sampler
            seqs.append([smp.next_batch() for _ in range(smp.batch_num())])
        got.append((seqs, random.random()))
        if cached:
            assert len(D.Evaluator._packed) == 1
    assert got[0][1] == got[1][1]
    for ea, eb in zip(got[0][0], got[1][0]):
        for (sa, pa), (sb, pb) in zip(ea, eb):
            assert np.array_equal(sa, sb) and np.array_equal(pa, pb)
    data.append([1, 2, 3])                                   # a changed list is packed again
    ev = D.Evaluator(data, True, 20, 64, 49, "valid", None, None)
    assert len(ev.evaluate_sampler.prepared_data) == 301


@pytest.mark.parametrize("n,maxlen", [(600, 50), (600, 4), (5000, 50)])
def test_rows_only_sampler_equals_the_prefix_expansion(n, maxlen):
    """An evaluator's Sampler cuts its rows straight from the flat item array (ader_host_prefix_rows) instead of building every prefix
    list first: same rows, same validity flags, same length, same shuffle -- empty, 1-, 2- and 3-item sessions included."""
    rs = np.random.RandomState(n + maxlen)
    data = [rs.randint(1, 30000, size=rs.randint(0, 70)).tolist() for _ in range(n)] + [[], [5], [1, 2], [1, 2, 3]]
    random.seed(1)
    a = D.Sampler(data, maxlen, 64, is_subseq=False)
    ra = random.random()
    random.seed(1)
    b = D.Sampler(data, maxlen, 64, is_subseq=False, rows_only=True)
    rb = random.random()
    assert len(a.prepared_data) == len(b.prepared_data) and a.batch_num() == b.batch_num()
    assert np.array_equal(a._valid, b._valid) and np.array_equal(a._rows, b._rows)
    assert ra == rb and a.data_indices == b.data_indices


def test_split_data(golden_dir):
    g = np.load(os.path.join(golden_dir, "split.npz"))
    s = np.load(os.path.join(golden_dir, "sampler.npz"))
    sess = unragged(s["in_flat"], s["in_off"])
    random.seed(0)
    np.random.seed(0)
    sm = D.Sampler(sess, 50, 16)
    valid, train = sm.split_data(0.1, return_train=True)
    assert valid == unragged(g["valid_flat"], g["valid_off"])
    assert train == unragged(g["train_flat"], g["train_off"])
    assert sm.data_indices == g["indices_after"].tolist()
    assert sm.batch_num() == int(g["batch_num"])
    seq, pos = sm.sampler()
    assert np.array_equal(np.stack(seq), g["first_batch_seq"])
    assert np.array_equal(np.array(pos), g["first_batch_pos"])


def test_evaluator_results(golden_dir):
    g = np.load(os.path.join(golden_dir, "evaluator.npz"))
    for k in ("edges", "all_hit", "none", "mixed"):
        got = D.recall_mrr(g[k + "_ranks"].tolist())
        assert np.array_equal(np.array(got, dtype=np.float64), g[k + "_results"]), k
    ev = D.Evaluator.__new__(D.Evaluator)
    ev.ranks = g["edges_ranks"].tolist()
    ev.mode = "valid"
    assert ev.display(3) == str(g["edges_display"])


def test_group_by_label_and_quota(golden_dir):
    from ader_amd.exemplar import draw_quotas
    g = np.load(os.path.join(golden_dir, "exemplar_init.npz"))
    data = unragged(g["data_flat"], g["data_off"])
    for tag, disable_m in (("prop", False), ("equal", True)):
        random.seed(0)
        np.random.seed(0)
        groups = D.group_by_label(data, 32, 50)
        assert list(groups.keys()) == g[tag + "_group_order"].tolist()
        assert [len(v) for v in groups.values()] == g[tag + "_group_sizes"].tolist()
        assert np.array_equal(np.concatenate(list(groups.values())), g[tag + "_rows"])
        quota = draw_quotas(groups, 100, disable_m, 40)
        assert np.array_equal(quota, g[tag + "_quota"])


def _crc(list_of_lists):
    flat = np.array([x for s in list_of_lists for x in s], dtype=np.int64)
    off = np.zeros(len(list_of_lists) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in list_of_lists])
    return int(zlib.crc32(flat.tobytes()) ^ zlib.crc32(off.tobytes()))


@pytest.mark.parametrize("dataset", ["DIGINETICA", "YOOCHOOSE"])
def test_dataloader_periods(golden_dir, dataset):
    root = os.path.join(os.path.dirname(os.path.dirname(golden_dir)), "data")
    if not os.path.isfile(os.path.join(root, dataset + ".npz")):
        pytest.skip("packed dataset not present")
    ref = json.load(open(os.path.join(golden_dir, "dataloader.json")))[dataset]
    dl = D.DataLoader(dataset, root=root)
    assert dl.num_periods() == len(ref) + 1
    for row in ref:
        tr, _ = dl.train_loader(row["period"] - 1)
        te, info = dl.evaluate_loader(row["period"])
        assert len(tr) == row["train_sessions"]
        assert sum(len(s) for s in tr) == row["train_actions"]
        assert _crc(tr) == row["train_crc"]
        assert len(te) == row["test_sessions"]
        assert _crc(te) == row["test_crc"]
        assert info == row["test_info"]
        assert dl.max_item() == row["max_item"]


def test_exemplar_store_round_trip(tmp_path):
    """Exemplars written next to a checkpoint come back identical, and the {item: [[session, logits]]} view (reference
    util.py:433: non-zero inputs followed by the label) is rebuilt from the rows."""
    import torch
    from ader_amd.exemplar import ExemplarStore
    rows = np.array([[0, 0, 5, 9, 3], [0, 7, 7, 2, 9], [1, 2, 3, 4, 3]], dtype=np.int32)      # [inputs(4) | label]
    logits = torch.arange(3 * 6, dtype=torch.float32).view(3, 6)
    st = ExemplarStore(rows, logits, 6)
    back = ExemplarStore.load(st.save(str(tmp_path / "exemplars.pt")))
    assert np.array_equal(back.rows, rows) and torch.equal(back.logits, logits) and back.max_item == 6 and len(back) == 3
    view = back.by_label()
    assert sorted(view) == [3, 9] and [e[0] for e in view[3]] == [[5, 9, 3], [1, 2, 3, 4, 3]] and view[9][0][0] == [7, 7, 2, 9]
    assert torch.equal(view[9][0][1], logits[1])


def test_published_curves_fixture():
    """tests/golden/results_svg_curves.json (the reference's results.svg, recovered by tests/golden/make_results_curves.py): 2 datasets
    x 5 methods x 2 metrics x 16 periods, and the 16-period averages the reference's paper / poster quote for ADER on DIGINETICA."""
    import json
    cur = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_svg_curves.json")))["curves"]
    assert sorted(cur) == ["DIGINETICA", "YOOCHOOSE"]
    for ds in cur:
        assert sorted(cur[ds]) == ["ADER", "Dropout", "EWC", "Finetune", "Joint"]
        for m in cur[ds].values():
            assert len(m["recall20"]) == 16 and len(m["mrr20"]) == 16
    avg = lambda v: sum(v) / 16      # noqa: E731
    assert abs(avg(cur["DIGINETICA"]["ADER"]["recall20"]) - 50.21) < 0.02 and abs(avg(cur["DIGINETICA"]["ADER"]["mrr20"]) - 17.32) < 0.02
    assert abs(avg(cur["YOOCHOOSE"]["ADER"]["recall20"]) - 72.38) < 0.02 and abs(avg(cur["YOOCHOOSE"]["ADER"]["mrr20"]) - 36.71) < 0.02
    # ADER is the best continual method on both datasets, fine-tuning the worst (the figure's message)
    for ds in cur:
        r = {k: avg(v["recall20"]) for k, v in cur[ds].items()}
        assert r["ADER"] > r["Dropout"] > r["EWC"] > r["Finetune"]


# ---------------------------------------------------------------------------------------- array data plane (PackedSessions)
def _rand_sessions(rs, n, hi=30000, maxlen=70):
    return [rs.randint(1, hi, size=rs.randint(0, maxlen)).tolist() for _ in range(n)] + [[], [5], [1, 2], [1, 2, 3]]


def test_packed_sessions_is_a_list_of_lists():
    """PackedSessions (flat item array + (start, length) per session) behaves as the list of lists the reference passes between its
    stages (util.py:138-143, 188-216, 382-393): len, indexing, slicing, iteration, ==, extend, and the prefix expansion."""
    rs = np.random.RandomState(0)
    data = _rand_sessions(rs, 300)
    ps = D.PackedSessions.from_lists(data)
    assert len(ps) == len(data) and ps == data and list(ps) == data and ps.tolist() == data
    assert ps[7] == data[7] and ps[-1] == data[-1] and ps[10:20] == data[10:20]
    idx = rs.permutation(len(data))[:50]
    assert ps.take(idx) == [data[i] for i in idx] and ps[idx] == [data[i] for i in idx]
    expanded = []
    for s in data:                                             # util.py:138-143
        expanded.append(s)
        for cut in range(1, len(s) - 1):
            expanded.append(s[:len(s) - cut])
    pre = ps.prefixes()
    assert pre == expanded and pre.flat is ps.flat             # a prefix is a shorter length over the same items
    a, b = ps.take(idx[:20]), ps.take(idx[20:])
    a.extend(b)                                                # same flat array: nothing copied
    assert a == [data[i] for i in idx] and a.flat is ps.flat
    other = _rand_sessions(rs, 40)
    a.extend(D.PackedSessions.from_lists(other))               # another flat array
    a.extend(other[:3])                                        # plain lists
    assert a == [data[i] for i in idx] + other + other[:3] and ps == data
    rows, _ = D.pack_rows(data, 50)
    want = [r[r != 0].tolist() for r in rows]
    assert D.PackedSessions.from_rows(rows) == want
    import random as _r
    _r.seed(3)
    x = _r.sample(ps, 17)
    _r.seed(3)
    assert x == _r.sample(data, 17)                            # (EWC: random.sample over the exemplar sessions, main.py:229)


@pytest.mark.parametrize("is_subseq", [False, True])
def test_sampler_over_packed_sessions_equals_the_list_sampler(is_subseq):
    """Same rows, validity flags, shuffled order, batches, split and RNG streams whether the Sampler is given lists or a PackedSessions."""
    rs = np.random.RandomState(5)
    data = _rand_sessions(rs, 700)
    out = []
    for packed in (False, True):
        random.seed(2)
        np.random.seed(2)
        sm = D.Sampler(D.PackedSessions.from_lists(data) if packed else data, 50, 64, is_subseq=is_subseq)
        first = (sm._rows.copy(), sm._valid.copy(), sm.data_indices, sm.prepared_data if packed else list(sm.prepared_data))
        valid, train = sm.split_data(0.1, return_train=True)
        batches = [sm.next_batch() for _ in range(sm.batch_num() + 3)]
        out.append((first, valid, train, sm._rows.copy(), sm.data_indices, batches, random.random(), np.random.rand()))
    (fa, va, ta, ra, ia, ba, xa, ya), (fb, vb, tb, rb, ib, bb, xb, yb) = out
    assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1]) and fa[2] == fb[2] and fb[3] == fa[3]
    assert isinstance(vb, D.PackedSessions) and vb == va and tb == ta
    assert np.array_equal(ra, rb) and ia == ib and (xa, ya) == (xb, yb)
    for (sa, pa), (sb, pb) in zip(ba, bb):
        assert np.array_equal(sa, sb) and np.array_equal(pa, pb)


def test_exemplar_store_feeds_the_sampler_like_the_flattened_list():
    """Sampler.add_exemplar(store) (rows taken as they are) == add_exemplar([[session, logits], ...]) (util.py:173-186): same rows,
    same shuffle, same batches; and group_by_label over PackedSessions == over lists."""
    import torch
    from ader_amd.exemplar import ExemplarStore
    rs = np.random.RandomState(9)
    sess = [s for s in _rand_sessions(rs, 400, hi=60) if len(s) >= 2]
    rows, valid = D.pack_rows(sess, 50)
    assert valid.all()
    logits = torch.arange(len(sess) * 3, dtype=torch.float32).view(len(sess), 3)
    store = ExemplarStore(rows, logits, 3)
    assert store.sessions() == [r[r != 0].tolist() for r in rows]
    got = []
    for how in ("list", "store"):
        random.seed(4)
        sm = D.Sampler([], 50, 16)
        sm.add_exemplar([[s, logits[i]] for i, s in enumerate(store.sessions())] if how == "list" else store)
        b = [sm.next_exemplar_batch() for _ in range(sm.batch_num() + 2)]
        got.append((sm._rows.copy(), sm._valid.copy(), sm.data_indices, b, random.random(), sm.exemplar_sampler()[2]))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]) and got[0][2] == got[1][2] and got[0][4] == got[1][4]
    for (sa, pa, ia), (sb, pb, ib) in zip(got[0][3], got[1][3]):
        assert np.array_equal(sa, sb) and np.array_equal(pa, pb) and np.array_equal(ia, ib)
    assert all(torch.equal(x, y) for x, y in zip(got[0][5], got[1][5]))
    groups = []
    for packed in (False, True):
        random.seed(1)
        groups.append(D.group_by_label(D.PackedSessions.from_lists(sess) if packed else sess, 32, 50))
    assert list(groups[0]) == list(groups[1]) and all(np.array_equal(groups[0][k], groups[1][k]) for k in groups[0])
