"""Shared pieces of the engine modules: epilogue / dropout-site ids, the host half of the dropout counter spec, the HIP-event section
timer, the flat parameter layout and the side-stream probe."""
import ctypes
import time as _time

import numpy as np
import torch

from .. import _lib

EPI_BIAS, EPI_BIAS_RELU_DROP, EPI_BIAS_DROP_RES_MASK, EPI_RELUDROPGRAD, EPI_ADD = range(5)
SITE_EMB = 0


def site_attn(l):
    return 1 + 3 * l


def site_ffn1(l):
    return 2 + 3 * l


def site_ffn2(l):
    return 3 + 3 * l


# ------------------------------------------------------------------ dropout counter keys (host side of the spec)
def _lowbias32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_key(seed, step, site):
    a = _lowbias32((seed & 0xFFFFFFFF) ^ 0x9E3779B9)
    b = (a + (step & 0xFFFFFFFF) * 0x85EBCA6B + site * 0xC2B2AE35) & 0xFFFFFFFF
    return _lowbias32(b)


class _Drop:
    """Descriptor of one dropout site for one step (include/ader_hip.h: AderDrop): key, threshold, scale and the counter
    offsets of the two local row segments -- rows [0, split_rows) continue at global row `row0`, the rows after them at global
    row `row0_2` (a data-parallel rank holds a slice of the train rows followed by a slice of the exemplar rows)."""

    __slots__ = ("c", "_ref")

    def __init__(self, seed, step, site, rate, training, per_row, row0=0, split_rows=None, row0_2=0):
        c = _lib.AderDrop()
        if training and rate > 0.0:
            c.key = dropout_key(seed, step, site)
            c.thr = int(round(float(rate) * 16777216.0))
            c.scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))
        else:
            c.key, c.thr, c.scale = 0, 0, 1.0
        c.base = (row0 * per_row) & 0xFFFFFFFF
        if split_rows is None:
            c.split, c.base2 = 0xFFFFFFFF, 0
        else:
            c.split = (split_rows * per_row) & 0xFFFFFFFF
            c.base2 = ((row0_2 - split_rows) * per_row) & 0xFFFFFFFF     # local index + base2 = global index of a segment-2 element
        self.c = c
        self._ref = ctypes.byref(c)

    def args(self):
        return (self._ref,)


# input slots of a native launch plan (plan.py; include/ader_hip.h: ader_step_enqueue `inputs`)
IN_SEQ, IN_POS, IN_EXPOS, IN_EXTROW, IN_TEACHER, IN_LR, IN_IDX_T, IN_IDX_E, N_INPUTS = range(9)


class StepF(float):
    """A float launcher argument that changes from step to step (Adam's lr_t): the plan recorder patches it from input `slot`."""
    __slots__ = ("slot",)

    def __new__(cls, v, slot):
        o = float.__new__(cls, v)
        o.slot = slot
        return o


class SectionTimer:
    """HIP-event timing of named launch groups on the stream the kernels are launched on (bench.py roofline leg).
    Events are recorded around each section; elapsed times are read back after a sync with collect()."""

    def __init__(self, only=None, every=1):
        self.pending = []
        self.totals = {}
        self.counts = {}
        self.only = only          # restrict the event pairs to these sections (each pair costs stream time)
        self.every = max(1, int(every))   # ... and to every n-th occurrence of a section
        self.seen = {}

    class _Ctx:
        def __init__(self, owner, name):
            self.o, self.name = owner, name

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

        def __exit__(self, *exc):
            self.b.record(torch.cuda.current_stream())
            self.o.pending.append((self.name, self.a, self.b))

    def section(self, name):
        if self.only is not None and name not in self.only:
            return _NULL
        k = self.seen.get(name, 0)
        self.seen[name] = k + 1
        if k % self.every:
            return _NULL
        return SectionTimer._Ctx(self, name)

    def collect(self):
        torch.cuda.synchronize()
        for name, a, b in self.pending:
            self.totals[name] = self.totals.get(name, 0.0) + a.elapsed_time(b)
            self.counts[name] = self.counts.get(name, 0) + 1
        self.pending = []
        return {k: self.totals[k] / self.counts[k] for k in self.totals}


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def param_layout(item_num, T, H, L, align=64, table_rows_alloc=None):
    """name -> (offset, shape) in the flat buffer; every tensor starts on a 256-byte boundary.  `table_rows_alloc`
    reserves extra (zero, never used) rows after the item table so that it splits into equal row shards."""
    names = [("emb", (item_num + 1, H)), ("pos", (T, H))]
    for l in range(L):
        p = "b%d." % l
        names += [(p + "ln1_g", (H,)), (p + "ln1_b", (H,)),
                  (p + "wq", (H, H)), (p + "bq", (H,)), (p + "wk", (H, H)), (p + "bk", (H,)),
                  (p + "wv", (H, H)), (p + "bv", (H,)),
                  (p + "ln2_g", (H,)), (p + "ln2_b", (H,)),
                  (p + "w1", (H, H)), (p + "b1", (H,)), (p + "w2", (H, H)), (p + "b2", (H,))]
    names += [("lnf_g", (H,)), ("lnf_b", (H,))]
    layout, off = {}, 0
    for n, shp in names:
        layout[n] = (off, shp)
        off += int(np.prod(shp))
        if n == "emb" and table_rows_alloc is not None:
            off = max(off, int(table_rows_alloc) * H)
        off = (off + align - 1) // align * align
    return layout, off


_SIDE_STREAMS = {}
SIDE_PROBE = {}       # (device index, main stream) -> [(overlaps, dependency latency in s, stream)] of every probed candidate


def side_stream(device, main):
    """The stream the engine's second lane runs on (sparse lists under the block kernels, small launches under the table update),
    shared by every engine of this process on (device, main).

    Not simply torch.cuda.Stream(priority=-1): HIP multiplexes its streams over four hardware queues per priority, and on the
    MI355X boxes ONE of the four high-priority queues answers a cross-stream dependency in ~180 us instead of ~33 us -- an engine
    whose side stream landed on it stepped in 1.24 ms instead of 0.39 ms at the real-data shapes (every 4th stream of torch's
    pool, stable within a process: profiles/r5_packed/side_stream_queues.txt; that is what the "not reproducible" 2x end-to-end
    outliers of tools/e2e_breakdown.py were -- the 4th engine of a process).  A normal-priority stream that shares the MAIN
    stream's hardware queue overlaps nothing (0.54 ms).  So: four consecutive high-priority pool streams (one per hardware queue)
    are probed once -- a main -> side -> main ping-pong of 24 tiny launches for the dependency latency, and one small side launch
    beside ~0.2 ms of main-stream work for the overlap -- and the best one that overlaps is kept."""
    key = (torch.device(device).index or 0, main.cuda_stream)
    if key in _SIDE_STREAMS:
        return _SIDE_STREAMS[key]
    dev = torch.device(device)
    with torch.cuda.device(dev), torch.cuda.stream(main):
        x = torch.zeros(1 << 12, device=dev)
        y = torch.zeros(1 << 12, device=dev)
        big = torch.zeros(1 << 24, device=dev)
        ev_m, ev_s = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for pr in (-1, 0):
            for _ in range(4):
                s = torch.cuda.Stream(device=dev, priority=pr)
                lats = []
                for rep in range(4):                              # (first pass: the runtime creates the hardware queue; then the
                    torch.cuda.synchronize(dev)                   #  median of three host-timed samples -- one alone is noisy)
                    t0 = _time.perf_counter()
                    for i in range(24 if rep else 4):
                        x.add_(1.0)
                        s.wait_stream(main)
                        with torch.cuda.stream(s):
                            y.add_(1.0)
                        main.wait_stream(s)
                    torch.cuda.synchronize(dev)
                    if rep:
                        lats.append((_time.perf_counter() - t0) / 24)
                lat = sorted(lats)[1]
                for i in range(6):
                    big.add_(1.0)
                ev_m.record(main)
                with torch.cuda.stream(s):
                    y.add_(1.0)
                    ev_s.record(s)
                torch.cuda.synchronize(dev)
                overlaps = ev_s.elapsed_time(ev_m) > 0.02          # the side launch finished well before the main-stream work did
                cand = (not overlaps, lat, s)
                SIDE_PROBE.setdefault(key, []).append((overlaps, lat, s))
                if best is None or cand[:2] < best[:2]:
                    best = cand
            if best is not None and not best[0]:
                break                                              # a high-priority stream that overlaps: done
        del big
    _SIDE_STREAMS[key] = best[2]
    return best[2]


def _check(cond, msg):
    """Shape / dtype / range violations of the operator surface raise RuntimeError (what TF's InvalidArgumentError
    becomes; never an AssertionError, which `python -O` would drop)."""
    if not cond:
        raise RuntimeError(msg)


def pack_counts_host(ids_host, n_pos, shard_items):
    """[owner][destination] row counts of the packed catalog exchange from the GLOBAL batch on the host: ids_host [W, n_all] int32
    (rank d's input positions, then its labels).  Returns (C_all, C_pos) as lists of lists -- what csrc/pack_plan.hip computes on the
    device, without a device-to-host synchronisation."""
    ids = np.asarray(ids_host)
    W = ids.shape[0]
    own = np.where(ids > 0, np.minimum((ids - 1) // shard_items, W - 1), -1)
    dst = np.broadcast_to(np.arange(W)[:, None], ids.shape)

    def counts(o, d):
        m = o >= 0
        return np.bincount(o[m] * W + d[m], minlength=W * W).reshape(W, W).tolist()

    return counts(own, dst), counts(own[:, :n_pos], dst[:, :n_pos])
