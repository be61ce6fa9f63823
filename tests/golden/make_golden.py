#!/usr/bin/env python
"""Generate golden vectors from the *reference's own* numpy/Python host code.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's ``util.py`` (DataLoader / Sampler / Evaluator / ExemplarGenerator,
reference util.py:17-522) with a stub ``tensorflow`` module in ``sys.modules``
(TensorFlow is not installed; those classes are numpy-only) and records
inputs + outputs as small fixtures next to this script.  Nothing of the
reference's source travels: the fixtures are data (inputs, expected outputs).

    python tests/golden/make_golden.py

Fixtures written (all consumed by tests/test_golden_*.py):
  sampler.npz        Sampler prefix expansion, label_generator, batch order
  split.npz          Sampler.split_data under fixed seeds
  dataloader.json    per-period DataLoader statistics + checksums, both datasets
  evaluator.npz      Evaluator.results() on crafted rank lists
  exemplar_init.npz  ExemplarGenerator.__init__ grouping + multinomial quotas
  herding.json       ExemplarGenerator.herding() selected-index lists (seeded inputs)
"""
import json
import os
import random
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference_util():
    class _Dummy:
        def __getattr__(self, k):
            return _Dummy()

        def __call__(self, *a, **k):
            return _Dummy()

    class _Mod(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Dummy()

    for n in ("tensorflow", "tensorflow.compat", "tensorflow.compat.v1"):
        sys.modules[n] = _Mod(n)
    sys.modules["tensorflow"].compat = sys.modules["tensorflow.compat"]
    sys.modules["tensorflow.compat"].v1 = sys.modules["tensorflow.compat.v1"]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import util  # noqa: E402  (the reference's util.py)

    return util


def ragged(list_of_lists):
    flat = np.array([x for s in list_of_lists for x in s], dtype=np.int64)
    off = np.zeros(len(list_of_lists) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in list_of_lists])
    return flat, off


def crafted_sessions():
    rs = np.random.RandomState(1234)
    sess = [
        [5, 9],                                   # len 2: no prefixes
        [3, 4, 7],                                # len 3: one prefix
        [11],                                     # len 1: kept in prepared_data, skipped by sampler()
        list(range(100, 151)),                    # len 51: inputs exactly fill maxlen
        list(rs.randint(1, 1000, size=200)),      # len 200: truncated to last 50 inputs
        [8, 8, 8, 2],                             # repeated items
    ]
    for _ in range(20):
        L = int(rs.randint(2, 12))
        sess.append([int(x) for x in rs.randint(1, 400, size=L)])
    return [[int(x) for x in s] for s in sess]


def gen_sampler(util):
    sess = crafted_sessions()
    random.seed(0)
    np.random.seed(0)
    sm = util.Sampler(sess, 50, 16, is_subseq=False)
    out = {}
    out["in_flat"], out["in_off"] = ragged(sess)
    out["prepared_flat"], out["prepared_off"] = ragged(sm.prepared_data)
    out["indices0"] = np.array(sm.data_indices, dtype=np.int64)
    out["batch_num"] = np.array(sm.batch_num())
    # label_generator on every prepared row with >= 2 items
    seqs, poss = [], []
    for s in sm.prepared_data:
        if len(s) >= 2:
            a, b = sm.label_generator(s)
            seqs.append(a)
            poss.append(int(b))
    out["lg_seq"] = np.stack(seqs).astype(np.int32)
    out["lg_pos"] = np.array(poss, dtype=np.int32)
    # 2.5 epochs of batches (covers the wrap-around reshuffle, ragged last batch, len-1 skip)
    nb = sm.batch_num()
    bseq, bpos, bsz = [], [], []
    for _ in range(2 * nb + nb // 2):
        seq, pos = sm.sampler()
        bsz.append(len(seq))
        bseq.extend(seq)
        bpos.extend(int(p) for p in pos)
    out["batches_seq"] = np.stack(bseq).astype(np.int32)
    out["batches_pos"] = np.array(bpos, dtype=np.int32)
    out["batches_size"] = np.array(bsz, dtype=np.int64)
    # is_subseq=True path (no expansion)
    random.seed(3)
    sm2 = util.Sampler(sess, 50, 7, is_subseq=True)
    out["sub_indices0"] = np.array(sm2.data_indices, dtype=np.int64)
    out["sub_prepared_n"] = np.array(len(sm2.prepared_data))
    # exemplar sampler path: add_exemplar + exemplar_sampler
    random.seed(5)
    ex = [[[4, 5, 6], [0.1, 0.2, 0.3]], [[9, 1], [1.0, 2.0, 3.0]], [[7, 7, 3, 2], [-1.0, 0.0, 1.0]],
          [[2, 3], [0.5, 0.5, 0.5]], [[6, 5, 4, 3, 2], [3.0, 2.0, 1.0]]]
    sm3 = util.Sampler([], 50, 2)
    sm3.add_exemplar(ex)
    eseq, epos, elog, esz = [], [], [], []
    for _ in range(2 * sm3.batch_num() + 1):
        s, p, l = sm3.exemplar_sampler()
        esz.append(len(s))
        eseq.extend(s)
        epos.extend(int(x) for x in p)
        elog.extend(l)
    out["ex_seq"] = np.stack(eseq).astype(np.int32)
    out["ex_pos"] = np.array(epos, dtype=np.int32)
    out["ex_logits"] = np.array(elog, dtype=np.float64)
    out["ex_sizes"] = np.array(esz, dtype=np.int64)
    out["ex_in_flat"], out["ex_in_off"] = ragged([e[0] for e in ex])
    out["ex_in_logits"] = np.array([e[1] for e in ex], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


def gen_split(util):
    sess = crafted_sessions()
    random.seed(0)
    np.random.seed(0)
    sm = util.Sampler(sess, 50, 16)
    valid, train = sm.split_data(0.1, return_train=True)
    out = {}
    out["valid_flat"], out["valid_off"] = ragged(valid)
    out["train_flat"], out["train_off"] = ragged(train)
    out["indices_after"] = np.array(sm.data_indices, dtype=np.int64)
    out["batch_num"] = np.array(sm.batch_num())
    seq, pos = sm.sampler()
    out["first_batch_seq"] = np.stack(seq).astype(np.int32)
    out["first_batch_pos"] = np.array([int(p) for p in pos], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "split.npz"), **out)


def _crc(list_of_lists):
    flat, off = ragged(list_of_lists)
    return int(zlib.crc32(flat.astype(np.int64).tobytes()) ^ zlib.crc32(off.tobytes()))


def gen_dataloader(util):
    res = {}
    for ds in ("DIGINETICA", "YOOCHOOSE"):
        dl = util.DataLoader(ds)
        dl.path = os.path.join(REF, "data", ds)
        n_periods = len([f for f in os.listdir(dl.path) if f.endswith(".txt")])
        rows = []
        for period in range(1, n_periods):
            tr, _ = dl.train_loader(period - 1)
            te, info = dl.evaluate_loader(period)
            rows.append({
                "period": period,
                "train_sessions": len(tr),
                "train_actions": int(sum(len(s) for s in tr)),
                "train_crc": _crc(tr),
                "test_sessions": len(te),
                "test_actions": int(sum(len(s) for s in te)),
                "test_crc": _crc(te),
                "test_info": info,
                "max_item": int(dl.max_item()),
            })
        res[ds] = rows
    with open(os.path.join(HERE, "dataloader.json"), "w") as f:
        json.dump(res, f, indent=0)


def gen_evaluator(util):
    ev = util.Evaluator.__new__(util.Evaluator)
    cases = {
        "edges": [0, 9, 10, 19, 20, 21, 1000],
        "all_hit": [0, 0, 1, 2],
        "none": [20, 50, 99],
        "mixed": [int(x) for x in np.random.RandomState(7).randint(0, 60, size=257)],
    }
    out = {}
    for k, ranks in cases.items():
        ev.ranks = list(ranks)
        out[k + "_ranks"] = np.array(ranks, dtype=np.int64)
        out[k + "_results"] = np.array(ev.results(), dtype=np.float64)
    ev.ranks = cases["edges"]
    ev.mode = "valid"
    out["edges_display"] = np.array(ev.display(3))
    np.savez_compressed(os.path.join(HERE, "evaluator.npz"), **out)


def gen_exemplar_init(util):
    rs = np.random.RandomState(99)
    max_item = 40
    data = []
    for _ in range(300):
        L = int(rs.randint(2, 9))
        # skewed labels so quotas are uneven
        body = [int(x) for x in rs.randint(1, max_item + 1, size=L - 1)]
        label = int(min(max_item, 1 + rs.geometric(0.15)))
        data.append(body + [label])
    out = {}
    out["data_flat"], out["data_off"] = ragged(data)
    for tag, disable_m in (("prop", False), ("equal", True)):
        random.seed(0)
        np.random.seed(0)
        eg = util.ExemplarGenerator(data, 100, disable_m, 32, 50, 0.3, max_item)
        out[tag + "_quota"] = np.array(eg.item_count, dtype=np.int64)
        out[tag + "_group_order"] = np.array(list(eg.sess_by_item.keys()), dtype=np.int64)
        rows = np.concatenate([np.stack(v) for v in eg.sess_by_item.values()]).astype(np.int32)
        out[tag + "_rows"] = rows
        out[tag + "_group_sizes"] = np.array([len(v) for v in eg.sess_by_item.values()], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "exemplar_init.npz"), **out)


def herding_inputs(seed, n, H, dup):
    """Seeded candidate representations (float32).  ``dup``>0 copies that many rows."""
    rs = np.random.RandomState(seed)
    rep = rs.standard_normal((n, H)).astype(np.float32)
    # make it look like an LN output (shared direction + noise) so norms/means are non-trivial
    rep = (rep * np.float32(0.7) + rs.standard_normal((1, H)).astype(np.float32)).astype(np.float32)
    for k in range(dup):
        src = int(rs.randint(0, n))
        dst = int(rs.randint(0, n))
        rep[dst] = rep[src]
    return rep


def gen_herding(util):
    from collections import defaultdict

    H = 150
    cases = []
    spec = []
    for n in (1, 2, 3, 5, 7, 64, 200, 1710):
        for m in sorted(set([0, 1, max(1, n // 2), n, n + 3])):
            spec.append((n, m, 0))
    for n in (7, 64, 200):
        for m in (max(1, n // 3), n):
            spec.append((n, m, 3))          # duplicate-containing characterisation set
    for i, (n, m, dup) in enumerate(spec):
        seed = 1000 + i
        rep = herding_inputs(seed, n, H, dup)
        logits = np.zeros((n, 2), dtype=np.float32)
        seq = np.zeros((n, 51), dtype=np.int32)
        seq[:, -2] = np.arange(1, n + 1)     # identity tag (recovered from the stored session)
        seq[:, -1] = 77
        eg = util.ExemplarGenerator.__new__(util.ExemplarGenerator)
        eg.exemplars = defaultdict(list)
        # reference call site passes min(m, len(seq)) (util.py:457)
        counter = eg.herding(rep, logits, seq, 77, min(m, n))
        sel = [e[0][0] - 1 for e in eg.exemplars[77]]
        assert all(e[0][-1] == 77 for e in eg.exemplars[77])
        # n == 2 with m < n is an exact mathematical tie (t[0] == t[1] == (1 + d0.d1)/2): the
        # reference's pick is BLAS rounding noise -> characterisation set, like the duplicate cases.
        klass = "characterise" if (dup > 0 or (n == 2 and 0 < m < n)) else "exact"
        cases.append({"seed": seed, "n": n, "m": m, "H": H, "dup": dup, "class": klass,
                      "selected": [int(x) for x in sel], "counter": int(counter)})
    with open(os.path.join(HERE, "herding.json"), "w") as f:
        json.dump({"numpy": np.__version__, "cases": cases}, f)


def main():
    util = _import_reference_util()
    gen_sampler(util)
    gen_split(util)
    gen_evaluator(util)
    gen_exemplar_init(util)
    gen_herding(util)
    gen_dataloader(util)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
