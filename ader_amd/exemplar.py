"""Exemplar selection: the MI355X build's counterpart of the reference's ``ExemplarGenerator``
(util.py:353-461: ``__init__`` grouping + multinomial quotas, ``herding``, ``herding_selection``).

The reference issues one ``sess.run`` per distinct label (~17k per DIGINETICA period, util.py:447-455)
and runs the greedy herding loop in numpy on the host.  Here all candidates go through ONE batched
eval-mode encode on the GPU, the herding loop runs as one segmented HIP kernel launch over all label
groups (csrc/herding.hip), and the teacher logits of the selected rows stay on the device as one
[E, N] tensor instead of Python float lists (util.py:433).
"""
from collections import defaultdict

import numpy as np

from . import data as _data


def draw_quotas(groups, exemplar_size, disable_m, max_item):
    """Per-item exemplar quotas (util.py:393-399): multinomial(m, freq/sum(freq)) on numpy's legacy
    global RNG; `disable_m` (--equal_exemplar) uses uniform probabilities over all max_item items."""
    item_count = np.zeros(max_item)
    if isinstance(groups, _data.LabelGroups):
        item_count[groups.labels - 1] += groups.sizes
    else:
        for label, rows in groups.items():
            item_count[label - 1] += len(rows)
    if disable_m:
        item_count = np.ones_like(item_count)
    item_prob = item_count / item_count.sum()
    return np.int32(np.random.multinomial(n=exemplar_size, pvals=item_prob, size=1)[0])


def herding_max_steps(m):
    """Loop bound of `while ... and step_t < 1.1 * m` (util.py:425), evaluated in float64 as Python does."""
    lim = 1.1 * float(m)
    k = int(lim)
    while k < lim:
        k += 1
    return k


class ExemplarStore:
    """Selected exemplars of one period: sessions as [E, maxlen+1] int32 rows (inputs ‖ label) and
    their teacher logits [E, N] (float32, device tensor when produced by the GPU path)."""

    def __init__(self, rows, logits, max_item):
        self.rows = rows
        self.logits = logits
        self.max_item = max_item

    def __len__(self):
        return int(self.rows.shape[0])

    def sessions(self):
        """Sessions in the reference's stored form: non-zero inputs followed by the label (util.py:433) -- as a PackedSessions (a
        sequence of lists held as arrays: what main.py hands on as the next period's exemplar candidates)."""
        return _data.PackedSessions.from_rows(self.rows)

    # The reference keeps the exemplars in memory only (main.py:312): a crash loses them.  Here they can be written next to
    # the period's checkpoint and read back (rows + teacher logits + the catalog size they were computed for).
    def save(self, path):
        import torch
        lg = self.logits.detach().cpu().contiguous() if hasattr(self.logits, "detach") else torch.as_tensor(np.asarray(self.logits))
        rows = torch.as_tensor(np.ascontiguousarray(np.asarray(self.rows), dtype=np.int32))
        torch.save({"rows": rows, "logits": lg, "max_item": int(self.max_item)}, path)
        return path

    @classmethod
    def load(cls, path, device=None):
        import torch
        d = torch.load(path, weights_only=True)        # tensors and plain numbers only: nothing is unpickled
        lg = d["logits"]
        if device is not None:          # on the device the rows keep the 16-byte aligned stride Engine.teacher_logits gives them
            pad = torch.empty((lg.shape[0], (lg.shape[1] + 3) // 4 * 4), dtype=lg.dtype, device=device)[:, :lg.shape[1]]
            pad.copy_(lg)
            lg = pad
        return cls(d["rows"].numpy(), lg, int(d["max_item"]))

    def by_label(self):
        """{item: [[session, logits_row], ...]} -- the reference's `fast_exemplar` view (util.py:433), rows in store order."""
        out = defaultdict(list)
        for i, r in enumerate(np.asarray(self.rows)):
            out[int(r[-1])].append([r[r != 0].tolist(), self.logits[i]])
        return out


class ExemplarGenerator:
    """Same constructor arguments as the reference (util.py:366-374)."""

    def __init__(self, data, exemplar_size, disable_m, batch_size, maxlen, dropout_rate, max_item, shard=(0, 1)):
        self.shard = shard          # (rank, world): label groups are independent herding units
        self._exemplars, self._view = defaultdict(list), None
        self.m = exemplar_size
        self.max_item = max_item
        self.maxlen = maxlen
        self.dropout_rate = dropout_rate
        self.sess_by_item = _data.group_by_label(data, batch_size, maxlen)
        self.item_count = draw_quotas(self.sess_by_item, exemplar_size, disable_m, max_item)
        self.store = None

    @property
    def exemplars(self):
        """{item: [[session, logits_row], ...]} -- the reference's `fast_exemplar` (util.py:433), built from the store when somebody
        asks for it (main.py hands the store itself to the next period: ~30k small lists per period are not built to be re-packed)."""
        if self._view is not None:
            labels, counts, sel_rows, logits = self._view
            self._view = None
            p = 0
            for label, c in zip(labels, counts):
                if c or label in self._keep_empty:
                    self._exemplars[label] = [[r[r != 0].tolist(), logits[p + i]] for i, r in enumerate(sel_rows[p:p + c])]
                p += c
        return self._exemplars

    def _set_view(self, labels, counts, sel_rows, logits, keep_empty=False):
        self._keep_empty = set(labels) if keep_empty else set()
        self._view = (list(labels), [int(c) for c in counts], sel_rows, logits)

    def _segments(self):
        g = self.sess_by_item
        labels, offs, rows = g.labels.tolist(), g.offs, g.rows
        quota = np.minimum(self.item_count[g.labels - 1].astype(np.int64), g.sizes).astype(np.int32)
        return labels, offs, quota, rows

    def herding_selection(self, sess, model):
        """Select exemplars by herding (util.py:436-461).  `model` is an ader_amd.model.Ader; `sess` is
        accepted for call-surface compatibility.  Returns the number of exemplars saved."""
        labels, offs, quota, rows = self._segments()
        rank, world = self.shard
        if world == 1:
            sel_idx, sel_cnt = model.engine.herding_select(rows[:, :self.maxlen], offs, quota, self.max_item)
        else:
            # each rank selects inside its contiguous chunk of label groups; the per-group index lists are exchanged
            # (quotas were drawn from the same RNG stream on every rank, util.py:398)
            from . import dist as _dist
            chunks = _dist.split_groups(np.diff(offs).tolist(), world)
            g0, g1 = chunks[rank]
            r0, r1 = int(offs[g0]), int(offs[g1])
            if g1 > g0:
                li, lc = model.engine.herding_select(rows[r0:r1, :self.maxlen], offs[g0:g1 + 1] - r0, quota[g0:g1],
                                                     self.max_item)
            else:
                li, lc = np.zeros(0, np.int64), np.zeros(0, np.int32)
            parts = _dist.gather_lists((np.asarray(li), np.asarray(lc)), world)
            sel_idx = np.concatenate([np.asarray(p[0]) for p in parts])
            sel_cnt = np.concatenate([np.asarray(p[1]) for p in parts])
        keep = []
        for g, label in enumerate(labels):
            c = int(sel_cnt[g])
            ids = sel_idx[offs[g]:offs[g] + c] + offs[g]
            keep.append(ids)
        keep = np.concatenate(keep) if keep else np.zeros(0, np.int64)
        sel_rows = rows[keep]
        logits = model.engine.teacher_logits(sel_rows[:, :self.maxlen], self.max_item)
        self.store = ExemplarStore(sel_rows, logits, self.max_item)
        # reference-shaped view {item: [[session, logits_row], ...]} (logits rows are views of the store): built on demand
        self._set_view(labels, sel_cnt[:len(labels)], sel_rows, logits, keep_empty=True)
        return int(len(keep))

    def loss_selection(self, sess, model, first_only=False):
        """Exemplars with the SMALLEST loss per label (util.py:463-495), ranking by the per-row cross entropy.

        first_only (driver flag `--selection loss_ref`): what the reference's code EXECUTES rather than what it documents -- the first
        candidate of every label with a quota >= 1 (see below): the exemplar set of the poster's `loss` columns, if the published
        code produced them.

        The reference fetches `model.loss` -- the batch MEAN, a scalar (ADER.py:93) -- so its `loss.argsort()[:m]` sees a 0-d array,
        yields [0] and always keeps just the FIRST candidate of every label.  This implementation does
        what the method documents ("selects exemplars by ranking loss"): one batched eval-mode pass gives every candidate's own
        -log softmax(logits)[label]; per label the min(m, n) rows of smallest loss are kept, ties by candidate order (stable
        argsort, as numpy's on equal keys).  Labels with quota 0 are skipped (`if m < 0.5: continue`, util.py:481)."""
        labels, offs, quota, rows = self._segments()
        loss = model.engine.row_losses(rows[:, :self.maxlen], rows[:, self.maxlen], self.max_item).cpu().numpy()
        keep, counts = [], []
        for g, label in enumerate(labels):
            n = int(offs[g + 1] - offs[g])
            m = int(self.item_count[label - 1])
            c = 0
            if m >= 1:
                ids = (np.zeros(1, np.int64) if first_only else
                       np.argsort(loss[offs[g]:offs[g + 1]], kind="stable")[:min(m, n)])
                keep.append(ids + offs[g])
                c = len(ids)
            counts.append(c)
        keep = np.concatenate(keep) if keep else np.zeros(0, np.int64)
        sel_rows = rows[keep]
        logits = model.engine.teacher_logits(sel_rows[:, :self.maxlen], self.max_item)
        self.store = ExemplarStore(sel_rows, logits, self.max_item)
        self._set_view(labels, counts, sel_rows, logits)
        return int(len(keep))

    def randomly_selection(self, sess, model):
        """Random exemplars per label (util.py:497-522): np.random.choice without replacement, in group order."""
        labels, offs, quota, rows = self._segments()
        keep, counts = [], []
        for g, label in enumerate(labels):
            n = int(offs[g + 1] - offs[g])
            m = int(self.item_count[label - 1])
            c = 0
            if m > 0:
                ids = np.random.choice(n, min(m, n), replace=False)
                keep.append(ids + offs[g])
                c = len(ids)
            counts.append(c)
        keep = np.concatenate(keep) if keep else np.zeros(0, np.int64)
        sel_rows = rows[keep]
        logits = model.engine.teacher_logits(sel_rows[:, :self.maxlen], self.max_item)
        self.store = ExemplarStore(sel_rows, logits, self.max_item)
        self._set_view(labels, counts, sel_rows, logits)
        return int(len(keep))
