"""Host-side feeders of the hot path: the MI355X build's counterparts of the reference's
``DataLoader`` (util.py:17-107), ``Sampler`` (util.py:110-273) and ``Evaluator`` (util.py:276-350).

Same class names, constructor arguments, method names and RNG consumption order as the reference
(``random.shuffle`` of the index list at construction / after add_exemplar / after split_data / at
epoch wrap; ``np.random.shuffle`` in split_data), so a seeded run draws the same batches; pinned by
tests/golden/{sampler,split,dataloader,evaluator}.npz which were produced by the reference itself.

What is different is the representation: sub-sequences are packed once into an int32 matrix
``rows[n, maxlen+1]`` (inputs right-aligned in zeros, label last) so a batch is one fancy-index
gather that can be handed to the device as a single contiguous buffer, instead of per-row Python
loops on every step (reference util.py:218-239).
"""
import itertools
import math
import os
import random
from collections.abc import Mapping, Sequence

import numpy as np

_DATA_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


# --------------------------------------------------------------------------------------- PackedSessions
class PackedSessions(Sequence):
    """A list of sessions (lists of item ids) held as arrays: one flat int32 item array and, per session, (start, length) into it.

    The reference hands Python lists of lists from stage to stage -- DataLoader -> Sampler (prefix expansion, util.py:138-143) ->
    split_data (util.py:188-216) -> exemplar candidates -> ExemplarGenerator (regrouped by label, util.py:382-393) -- and every stage
    walks them element by element.  This is the same sequence surface (len, indexing, iteration, extend, ==; an element is a plain
    list when somebody asks for one) over arrays, so the stages above run as array operations: a prefix of a session is the same
    start with a shorter length (no copy), a split is a fancy index of (start, length), and the packed rows of the feeders are cut
    from the flat array in native code (ader_host_pack_rows_at)."""
    __slots__ = ("flat", "starts", "lens")

    def __init__(self, flat, starts, lens):
        self.flat = np.ascontiguousarray(flat, dtype=np.int32)
        self.starts = np.ascontiguousarray(starts, dtype=np.int64)
        self.lens = np.ascontiguousarray(lens, dtype=np.int64)

    # -- construction
    @classmethod
    def from_counts(cls, flat, counts):
        """Sessions stored back to back in `flat`, `counts[i]` items each."""
        counts = np.asarray(counts, dtype=np.int64)
        starts = np.zeros(len(counts), dtype=np.int64)
        if len(counts) > 1:
            np.cumsum(counts[:-1], out=starts[1:])
        return cls(flat, starts, counts)

    @classmethod
    def from_lists(cls, sessions):
        if isinstance(sessions, PackedSessions):
            return cls(sessions.flat, sessions.starts, sessions.lens)
        sessions = sessions if isinstance(sessions, (list, tuple)) else list(sessions)
        n = len(sessions)
        lens = np.fromiter(map(len, sessions), dtype=np.int64, count=n)
        flat = np.fromiter(itertools.chain.from_iterable(sessions), dtype=np.int32, count=int(lens.sum()))
        return cls.from_counts(flat, lens)

    @classmethod
    def from_rows(cls, rows):
        """Sessions of packed rows [n, maxlen+1] in the reference's stored form (util.py:433): the non-zero inputs, then the label."""
        rows = np.asarray(rows)
        nz = rows != 0
        return cls.from_counts(rows[nz], nz.sum(axis=1))

    # -- the sequence surface
    def __len__(self):
        return len(self.starts)

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            s = int(self.starts[i])
            return self.flat[s:s + int(self.lens[i])].tolist()
        if isinstance(i, slice):
            return PackedSessions(self.flat, self.starts[i], self.lens[i])
        return self.take(i)

    def __iter__(self):
        flat = self.flat
        for s, n in zip(self.starts.tolist(), self.lens.tolist()):
            yield flat[s:s + n].tolist()

    def __eq__(self, other):
        if isinstance(other, PackedSessions):
            other = other.tolist()
        return isinstance(other, (list, tuple)) and self.tolist() == list(other)

    __hash__ = None

    def __repr__(self):
        return "PackedSessions(%d sessions, %d items)" % (len(self), int(self.lens.sum()))

    def tolist(self):
        return list(iter(self))

    def take(self, idx):
        """The sessions idx[0], idx[1], ... (no item is copied)."""
        idx = np.asarray(idx)
        return PackedSessions(self.flat, self.starts[idx], self.lens[idx])

    def extend(self, other):
        """list.extend: in place.  Sessions over the same flat array are appended without copying an item."""
        if not isinstance(other, PackedSessions):
            other = PackedSessions.from_lists(other)
        if len(other) == 0:
            return
        if other.flat is self.flat:
            shift = 0
        else:
            shift = len(self.flat)
            self.flat = np.concatenate([self.flat, other.flat])
        self.starts = np.concatenate([self.starts, other.starts + shift])
        self.lens = np.concatenate([self.lens, other.lens])

    def prefixes(self):
        """Every session followed by its prefixes down to length 2 (the Sampler's expansion, util.py:138-143): a session of length l
        gives lengths l, l-1, ..., 2 (itself alone when l <= 2)."""
        counts = np.maximum(self.lens - 1, 1)
        rep = np.repeat(np.arange(len(self), dtype=np.int64), counts)
        first = np.zeros(len(self), dtype=np.int64)
        if len(self) > 1:
            np.cumsum(counts[:-1], out=first[1:])
        k = np.arange(int(counts.sum()), dtype=np.int64) - first[rep]
        return PackedSessions(self.flat, self.starts[rep], self.lens[rep] - k)


# --------------------------------------------------------------------------------------- DataLoader
class DataLoader:
    """Loads ``period_N`` interaction logs (reference util.py:17-107).

    Sources, tried in order: ``<root>/<dataset>.npz`` (packed int32 arrays ``sess_<p>``/``item_<p>``
    in file order, written by tools/pack_dataset.py) or ``<root>/<dataset>/period_<p>.txt``
    ("sessId itemId" per line, util.py:46)."""

    def __init__(self, dataset, root=None):
        self.item_set = set()
        self.is_remove_item = True
        self.dataset = dataset
        self.root = root or _DATA_ROOT
        self._packed = None
        npz = os.path.join(self.root, dataset + ".npz")
        if os.path.isfile(npz):
            self._packed = np.load(npz)

    def num_periods(self):
        if self._packed is not None:
            return len([k for k in self._packed.files if k.startswith("item_")])
        d = os.path.join(self.root, self.dataset)
        return len([f for f in os.listdir(d) if f.endswith(".txt")])

    def _read(self, period):
        if self._packed is not None:
            return (self._packed["sess_%d" % period].astype(np.int64),
                    self._packed["item_%d" % period].astype(np.int64))
        path = os.path.join(self.root, self.dataset, "period_%d.txt" % period)
        arr = np.loadtxt(path, dtype=np.int64, ndmin=2)
        return arr[:, 0], arr[:, 1]

    @staticmethod
    def _group(sess, item):
        """Sessions in first-appearance order, items in file order (dict insertion order, util.py:42-52), as a PackedSessions
        (a sequence of lists held as arrays)."""
        if len(sess) == 0:
            return PackedSessions.from_counts(np.zeros(0, np.int32), np.zeros(0, np.int64))
        uniq, first, inv = np.unique(sess, return_index=True, return_inverse=True)
        order_of_group = np.argsort(first, kind="stable")          # groups by first appearance
        rank = np.empty_like(order_of_group)
        rank[order_of_group] = np.arange(len(uniq))
        g = rank[inv]
        idx = np.argsort(g, kind="stable")                          # stable: keeps file order in a group
        counts = np.bincount(g, minlength=len(uniq))
        return PackedSessions.from_counts(item[idx], counts)

    def train_loader(self, period):
        sess, item = self._read(period)
        self.item_set.update(np.unique(item).tolist())
        sessions = self._group(sess, item)
        info = 'Train set information: total number of action: %d.' % len(item)
        print(info)
        return sessions, info

    def evaluate_loader(self, period):
        sess, item = self._read(period)
        total_num = len(item)
        removed_num = 0
        if self.is_remove_item:
            known = np.fromiter(self.item_set, dtype=np.int64, count=len(self.item_set))
            keep = np.isin(item, known)
            removed_num += int((~keep).sum())
            sess, item = sess[keep], item[keep]
        # (items that survive are already members of item_set: util.py:84-85 adds nothing new)
        sessions = self._group(sess, item)
        if self.is_remove_item:
            single = sessions.lens == 1
            removed_num += int(single.sum())
            sessions = sessions.take(np.flatnonzero(~single))
        info = ('Test set information: original total number of action: %d, removed number of action: %d.'
                % (total_num, removed_num))
        return sessions, info

    def max_item(self):
        return max(self.item_set)


# --------------------------------------------------------------------------------------- Sampler
def shuffle_like_python(a):
    """random.shuffle on an int64 numpy array: same permutation, same consumption of the `random` module's stream (the
    reference shuffles its index list at every epoch and repack, util.py:149,186,211,237,261, and every later draw of the run --
    validation split, exemplar selection -- continues that stream).  Long arrays go through ader_host_shuffle (csrc/host_feed.hip:
    CPython's algorithm on the generator's state; 2-11 ms of pure-Python loop per epoch on the shipped datasets otherwise)."""
    n = len(a)
    if n >= 256:
        from . import _lib
        st = random.getstate()
        mt = np.array(st[1], dtype=np.uint32)
        _lib.call("ader_host_shuffle", mt.ctypes.data, a.ctypes.data, n)
        random.setstate((st[0], tuple(mt.tolist()), st[2]))
        return
    lst = a.tolist()
    random.shuffle(lst)
    a[:] = lst


def pack_rows(sessions, maxlen):
    """[n, maxlen+1] int32: up to the last `maxlen` inputs right-aligned in zeros, then the label
    (= last item).  Rows of sessions shorter than 2 are all-zero and flagged invalid
    (the reference skips them when batching, util.py:226-227).  Long lists: one pass over the items into a flat array, the rows
    cut from it in native code (ader_host_pack_rows) -- the per-row Python loop was a tenth of an end-to-end run."""
    n = len(sessions)
    rows = np.zeros((n, maxlen + 1), dtype=np.int32)
    valid = np.zeros(n, dtype=bool)
    if isinstance(sessions, PackedSessions):
        if n:
            from . import _lib
            _lib.call("ader_host_pack_rows_at", sessions.flat.ctypes.data, len(sessions.flat), sessions.starts.ctypes.data, sessions.lens.ctypes.data, n,
                      int(maxlen), rows.ctypes.data, valid.ctypes.data)
        return rows, valid
    if n >= 512:
        from . import _lib
        lens = np.fromiter(map(len, sessions), dtype=np.int64, count=n)
        flat = np.fromiter(itertools.chain.from_iterable(sessions), dtype=np.int32, count=int(lens.sum()))
        _lib.call("ader_host_pack_rows", flat.ctypes.data, lens.ctypes.data, n, int(maxlen), rows.ctypes.data, valid.ctypes.data)
        return rows, valid
    for i, s in enumerate(sessions):
        L = len(s)
        if L <= 1:
            continue
        valid[i] = True
        k = min(L - 1, maxlen)
        rows[i, maxlen - k:maxlen] = s[L - 1 - k:L - 1]
        rows[i, maxlen] = s[L - 1]
    return rows, valid


class Sampler:
    def __init__(self, data, maxlen, batch_size, is_subseq=False, packed=None, rows_only=False):
        """packed: (prepared_data, rows, valid) of an earlier Sampler over the SAME data (Evaluator keeps them: the reference builds a
        new evaluator -- hence a new Sampler -- every epoch over an unchanged validation set, util.py:276-290; the shuffle of the new
        index list is the part of that which consumes the `random` stream, and it is kept)."""
        self.maxlen = maxlen
        self.batch_size = batch_size
        self.batch_counter = 0
        self.logits = []
        self.prepared_data = []
        if packed is not None:
            self.prepared_data = packed[0]
            self._repack(packed[1], packed[2])
            return
        if isinstance(data, PackedSessions):
            # array data plane: the prefix expansion is (start, shorter length) pairs over the same flat item array, the rows are
            # cut from it in native code; no list is built (the same rows, order and shuffle as the list path below)
            self.prepared_data = data[:] if is_subseq else data.prefixes()
            self._repack()
            return
        if rows_only and not is_subseq and len(data) >= 512:
            # an evaluator reads only the packed rows: they are cut for every session and its prefixes straight from the flat item
            # array (ader_host_prefix_rows, the order of the loop below) and the ~6 prefix lists per session are never built;
            # prepared_data keeps its length (batch_num / data_size)
            from . import _lib
            n = len(data)
            lens = np.fromiter(map(len, data), dtype=np.int64, count=n)
            flat = np.fromiter(itertools.chain.from_iterable(data), dtype=np.int32, count=int(lens.sum()))
            total = int(np.where(lens > 2, lens - 1, 1).sum())
            rows = np.zeros((total, maxlen + 1), dtype=np.int32)
            valid = np.zeros(total, dtype=bool)
            _lib.call("ader_host_prefix_rows", flat.ctypes.data, lens.ctypes.data, n, int(maxlen), rows.ctypes.data, valid.ctypes.data)
            self.prepared_data = range(total)
            self._repack(rows, valid)
            return
        if not is_subseq:
            # a session of length l yields itself and its prefixes down to length 2 (util.py:138-143)
            for session in data:
                self.prepared_data.append(session)
                for cut in range(1, len(session) - 1):
                    self.prepared_data.append(session[:len(session) - cut])
        else:
            self.prepared_data.extend(data)
        self._repack()

    # -- internal
    def _repack(self, rows=None, valid=None):
        self._rows, self._valid = pack_rows(self.prepared_data, self.maxlen) if rows is None else (np.ascontiguousarray(rows), valid)
        self._rows_dev = self._seq_dev = self._lab_dev = self._epoch = None
        self.last_idx_dev = None
        self._logit_mat = None
        self._perm = np.arange(len(self.prepared_data), dtype=np.int64)
        shuffle_like_python(self._perm)

    @property
    def data_indices(self):
        """The shuffled index list of the reference's Sampler (util.py:148-149), kept as an int64 array."""
        return self._perm.tolist()

    def _advance(self):
        self.batch_counter += 1
        if self.batch_counter == self.batch_num():
            self.batch_counter = 0
            shuffle_like_python(self._perm)
            self._epoch = None

    def _next_indices(self):
        lo = self.batch_counter * self.batch_size
        idx = self._perm[lo:lo + self.batch_size].copy()
        idx = idx[self._valid[idx]] if len(idx) else idx
        self._advance()
        return idx

    def _plan_epoch(self):
        """Device-resident feeder: the batches of the whole epoch at once -- the shuffled order (the reference's `random` stream, as
        _next_indices reads it) with the invalid rows dropped per batch, uploaded in ONE copy per epoch; a batch is then a slice of
        that device array (per step this replaces a list -> array conversion, a filter and a pageable host-to-device copy)."""
        import torch
        perm = self._perm
        keep = self._valid[perm] if len(perm) else np.zeros(0, dtype=bool)
        bounds = np.minimum(np.arange(self.batch_num() + 1, dtype=np.int64) * self.batch_size, len(perm))
        offs = np.concatenate([[0], np.cumsum(keep)])[bounds]
        flat = perm[keep]
        dev = torch.from_numpy(flat).to(self._seq_dev.device)
        self._epoch = (flat, offs.tolist(), dev, dev.to(torch.int32))

    # -- reference surface
    def label_generator(self, session):
        rows, _ = pack_rows([session], self.maxlen)
        return rows[0, :self.maxlen].copy(), np.array(session[-1], dtype=np.int32)

    def add_exemplar(self, exemplar):
        """exemplar: iterable of [session, teacher_logits] (util.py:173-186), or an ExemplarStore -- its packed rows ARE the rows of
        its sessions (a stored session is the row's non-zero inputs and label), so nothing is unpacked and packed again; the teacher
        logits stay the store's [E, N] tensor."""
        if hasattr(exemplar, "rows") and hasattr(exemplar, "logits"):
            rows = np.ascontiguousarray(np.asarray(exemplar.rows), dtype=np.int32)
            assert rows.ndim == 2 and rows.shape[1] == self.maxlen + 1
            if not isinstance(self.prepared_data, PackedSessions):
                self.prepared_data = PackedSessions.from_lists(self.prepared_data)
            n0 = len(self.prepared_data)
            self.prepared_data.extend(PackedSessions.from_rows(rows))
            self.logits = exemplar.logits
            if n0:
                old_rows, old_valid = pack_rows(self.prepared_data[:n0], self.maxlen)
                rows = np.concatenate([old_rows, rows])
            valid = (rows[:, self.maxlen] != 0) & (rows[:, self.maxlen - 1] != 0)      # a label and at least one input
            self._repack(rows, valid)
            return
        self.logits = []
        for session, logits in exemplar:
            self.prepared_data.append(session)
            self.logits.append(logits)
        self._repack()

    def split_data(self, valid_portion, return_train=False):
        n = len(self.prepared_data)
        sidx = np.arange(n, dtype='int32')
        np.random.shuffle(sidx)
        n_train = int(np.round(n * (1. - valid_portion)))
        if isinstance(self.prepared_data, PackedSessions):
            valid_data, train_data = self.prepared_data.take(sidx[n_train:]), self.prepared_data.take(sidx[:n_train])
        else:
            valid_data = [self.prepared_data[s] for s in sidx[n_train:]]
            train_data = [self.prepared_data[s] for s in sidx[:n_train]]
        self.prepared_data = train_data
        self._repack(self._rows[sidx[:n_train]], self._valid[sidx[:n_train]])       # (a row depends on its own session only)
        return (valid_data, train_data) if return_train else valid_data

    def to_device(self, device):
        """GPU-resident feeder: the packed [n, maxlen+1] rows live on the device; a batch is then an index_select
        by the host-shuffled indices (same `random` stream as the reference) -- 2 KB of indices cross PCIe instead of the rows.
        Call again after split_data / add_exemplar (they repack)."""
        import torch
        self._rows_dev = torch.from_numpy(self._rows).to(device)
        self._seq_dev = self._rows_dev[:, :self.maxlen].contiguous()        # gathered by index_select: contiguous outputs, no slicing copies
        self._lab_dev = self._rows_dev[:, self.maxlen].contiguous()
        self._epoch = None
        # fraction of real positions (sessions are left-padded to maxlen, util.py:161-169): what the engine's packed session
        # kernels go by when the batches no longer pass through the host (Engine.pack_density)
        self.density = float(np.count_nonzero(self._rows[:, :self.maxlen])) / max(self._rows[:, :self.maxlen].size, 1)
        return self

    def _rows_of(self, idx):
        if getattr(self, "_rows_dev", None) is None:
            rows = self._rows[idx]
            return np.ascontiguousarray(rows[:, :self.maxlen]), np.ascontiguousarray(rows[:, self.maxlen])
        import torch
        rows = self._rows_dev.index_select(0, torch.from_numpy(idx).to(self._rows_dev.device, non_blocking=True))
        return rows[:, :self.maxlen].contiguous(), rows[:, self.maxlen].contiguous()

    def _next_device_batch(self):
        if self._epoch is None:
            self._plan_epoch()
        flat, offs, dev, dev32 = self._epoch
        o0, o1 = int(offs[self.batch_counter]), int(offs[self.batch_counter + 1])
        self._advance()
        idx_dev = dev[o0:o1]
        self.last_idx_dev = dev32[o0:o1]            # the same row indices as an int32 device tensor (e.g. teacher rows of the batch)
        return self._seq_dev.index_select(0, idx_dev), self._lab_dev.index_select(0, idx_dev), flat[o0:o1]

    def next_index_slice(self):
        """Device-resident feeder without the gather: (idx, offset, count) -- the next batch is rows idx[offset : offset + count] of
        the packed rows `rows_dev()` (idx: the epoch's int64 index array on the device, the shuffled order with the invalid rows
        dropped per batch).  Same batch sequence and `random` stream as next_batch(); the gather itself belongs to the consumer
        (Engine.train_step_fed cuts the step's inputs from it in one launch)."""
        if self._epoch is None:
            self._plan_epoch()
        _, offs, dev, _ = self._epoch
        o0, o1 = offs[self.batch_counter], offs[self.batch_counter + 1]
        self._advance()
        return dev, o0, o1 - o0

    def rows_dev(self):
        """The packed rows [n, maxlen+1] int32 on the device (after to_device())."""
        return self._rows_dev

    def next_batch(self):
        """Fast path: (seq [b, maxlen] int32, pos [b] int32), contiguous; numpy arrays, or device tensors after to_device()."""
        if getattr(self, "_seq_dev", None) is not None:
            return self._next_device_batch()[:2]
        return self._rows_of(self._next_indices())

    def next_exemplar_batch(self):
        """Fast path: (seq, pos, row indices into the exemplar list) -- teacher logits stay wherever
        the caller keeps them (e.g. one [E, Np] device tensor) and are gathered by index (Sampler.last_idx_dev: the indices as an
        int32 device tensor, after to_device())."""
        if getattr(self, "_seq_dev", None) is not None:
            return self._next_device_batch()
        idx = self._next_indices()
        seq, pos = self._rows_of(idx)
        return seq, pos, idx

    def sampler(self):
        seq, pos = self.next_batch()
        return tuple(seq), tuple(pos)

    def exemplar_sampler(self):
        seq, pos, idx = self.next_exemplar_batch()
        return tuple(seq), tuple(pos), tuple(self.logits[i] for i in idx)

    def data_size(self):
        return len(self.prepared_data)

    def batch_num(self):
        return math.ceil(len(self.prepared_data) * 1.0 / self.batch_size)


# --------------------------------------------------------------------------------------- Evaluator
def recall_mrr(ranks):
    """(MRR@20, RECALL@20, MRR@10, RECALL@10) from 0-based ranks (reference util.py:329-339)."""
    r = np.asarray(ranks, dtype=np.int64)
    n = len(r)
    out = []
    for k in (20, 10):
        hit = r[r < k]
        # sequential float64 sum in list order, as Python's sum() over the filtered list does
        mrr = 0.0
        for x in hit.tolist():
            mrr += 1.0 / (x + 1)
        out.extend([mrr / n, len(hit) / n])
    return tuple(out)


class Evaluator:
    """Evaluates the rank of the ground-truth next item among all `max_item` items
    (reference util.py:276-350).  `model.predict(sess, seq, item_idx)` must return the full rank
    matrix as in the reference; when the model offers `rank_targets(seq, pos, max_item)` (the HIP
    count-greater kernel) that is used instead -- only pred[label-1] is ever read (util.py:325)."""

    _packed = {}        # (id(data), len, maxlen, is_subseq) -> (data, (prepared_data, rows, valid)): see __init__

    def __init__(self, data, is_subseq, maxlen, batch_size, max_item, mode, model, sess, shard=(0, 1)):
        self.shard = shard          # (rank, world): evaluation batches are independent units
        self.max_item = max_item
        self.model = model
        self.sess = sess
        self.ranks = []
        self.mode = mode
        self.desc = 'Validating epoch ' if mode == 'valid' else 'Testing epoch '
        # (the packed rows of an unchanged session list are kept between evaluators -- one is built per epoch, main.py:264-266 -- and
        #  only the new Sampler's shuffle, which the `random` stream sees, is repeated)
        key = (id(data), len(data), maxlen, bool(is_subseq))
        hit = Evaluator._packed.get(key)
        if hit is not None and hit[0] is data:
            self.evaluate_sampler = Sampler(data, maxlen, batch_size, is_subseq=is_subseq, packed=hit[1])
        else:
            self.evaluate_sampler = Sampler(data, maxlen, batch_size, is_subseq=is_subseq, rows_only=True)
            smp = self.evaluate_sampler
            if len(Evaluator._packed) >= 4:
                Evaluator._packed.pop(next(iter(Evaluator._packed)))
            Evaluator._packed[key] = (data, (smp.prepared_data, smp._rows, smp._valid))

    @classmethod
    def clear_cache(cls):
        """Drop the packed rows kept between evaluators (main.py: at the start of every period -- the previous period's validation
        and test lists are not held alive).  A cached list must not be mutated in place between two evaluators built from it."""
        cls._packed.clear()

    def evaluate(self, epoch):
        rank, world = self.shard
        fast = getattr(self.model, "rank_targets", None)
        mine = []                    # [(batch index, ranks)] of the batches this rank evaluates
        todo = []                    # fast path: (batch index, seq, pos) of this rank's batches, ranked by ONE call below
        for b in range(self.evaluate_sampler.batch_num()):
            seq, pos = self.evaluate_sampler.next_batch()       # every rank walks the same batch sequence
            if len(pos) == 0 or b % world != rank:
                continue
            if fast is not None:
                todo.append((b, seq, pos))
            else:
                pred = self.model.predict(self.sess, seq, list(range(1, self.max_item + 1)))
                mine.append((b, [int(p[i - 1]) for p, i in zip(pred, pos)]))
        if todo:
            # the rank of a row's target depends on that row alone: all batches of the evaluation go through ONE rank_targets call
            # (chunked on the device, one copy of the ranks back) instead of a host round trip per batch
            r = np.asarray(self.model.rank_targets(np.concatenate([s_ for _, s_, _ in todo]), np.concatenate([p_ for _, _, p_ in todo]),
                                                   self.max_item)).tolist()
            o = 0
            for b, _, pos in todo:
                mine.append((b, r[o:o + len(pos)]))
                o += len(pos)
        if world > 1:                # ranks back in batch order on every rank: the metric sums run in the reference's order
            from . import dist as _dist
            mine = sorted(x for part in _dist.gather_lists(mine, world) for x in part)
        self.ranks = [x for _, r in mine for x in r]
        return self.display(epoch)

    def results(self):
        return recall_mrr(self.ranks)

    def display(self, epoch):
        r = self.results()
        info = 'epoch:%d, %s (MRR@20: %.4f, RECALL@20: %.4f, MRR@10: %.4f, RECALL@10: %.4f)' \
               % (epoch, self.mode, r[0], r[1], r[2], r[3])
        if getattr(self, "shard", (0, 1))[0] == 0:
            print(info)
        return info


def load_exemplars(exemplar_pre):
    """Flatten {item: [[session, logits], ...]} to a list (reference main.py:54-65)."""
    out = []
    for item in exemplar_pre.values():
        if isinstance(item, list):
            out.extend(i for i in item if i)
    return out


class LabelGroups(Mapping):
    """{label: [n_label, maxlen+1] int32 rows} in the order the labels are first visited, held as ONE row matrix (groups back to
    back) + offsets: what ExemplarGenerator keeps per period (util.py:382-393 builds a dict of lists of sessions).  A mapping like the
    dict it replaces; `rows`, `labels`, `offs` serve the selectors without re-concatenating ~10^4 slices."""

    def __init__(self, labels, offs, rows):
        self.labels, self.offs, self.rows = labels, offs, rows
        self._at = None

    def __len__(self):
        return len(self.labels)

    def __iter__(self):
        return iter(self.labels.tolist())

    def __getitem__(self, label):
        if self._at is None:
            self._at = {k: i for i, k in enumerate(self.labels.tolist())}
        i = self._at[int(label)]
        return self.rows[self.offs[i]:self.offs[i + 1]]

    @property
    def sizes(self):
        return np.diff(self.offs)


def group_by_label(data, batch_size, maxlen):
    """Bucket candidate sub-sequences by label in the reference's visiting order
    (ExemplarGenerator.__init__, util.py:382-393): batches drawn from a shuffled is_subseq Sampler.
    Returns a LabelGroups: {label: [n_label, maxlen+1] int32 rows}, labels in the order of their first appearance, the rows of a label
    in visiting order."""
    smp = Sampler(data, maxlen, batch_size, is_subseq=True)
    # One pass over the batches of the shuffled Sampler == its shuffled order with the invalid rows dropped; the pass ends with the
    # Sampler's reshuffle (Sampler._advance), which the `random` stream sees and is therefore made here too.  Grouping: a stable sort
    # by label, the groups then ordered by the position of their first row; ONE gather for all rows.
    n_batches = smp.batch_num()
    order = smp._perm[smp._valid[smp._perm]] if len(smp._perm) else smp._perm
    if n_batches > 0:
        shuffle_like_python(smp._perm)
    if len(order) == 0:
        return LabelGroups(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros((0, maxlen + 1), np.int32))
    labels = smp._rows[order, maxlen]
    o = np.argsort(labels, kind="stable")
    sl = labels[o]
    starts = np.flatnonzero(np.concatenate([[True], sl[1:] != sl[:-1]]))
    ends = np.concatenate([starts[1:], [len(sl)]])
    gs = np.argsort(o[starts], kind="stable")                 # groups by the position in `order` of their first row
    sizes = (ends - starts)[gs]
    offs = np.zeros(len(gs) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offs[1:])
    pick = np.repeat(starts[gs] - offs[:-1], sizes) + np.arange(len(sl), dtype=np.int64)      # sorted positions, group after group
    return LabelGroups(sl[starts[gs]].astype(np.int64), offs, smp._rows[order[o[pick]]])
