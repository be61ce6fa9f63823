"""Native step driver (host side): a train step is recorded ONCE per (shape, mode) into a launch plan and replayed by ONE C call.

Reference: main.py:220-256 -- one `sess.run(train_op)` per step; TF's executor walks the ops, Python only feeds.  Here a step is ~20
launcher calls on two streams; issued one ctypes call at a time they cost 0.22-0.27 ms of host time per step against 0.38 ms of GPU
time at the shipped datasets' shapes.  So: the FIRST step of a (shape, mode) runs through the ordinary Python path with a recorder
attached to `_lib.call` and to the stream edges (`Engine._edge`); its launch sequence becomes an `AderStepPlan` (csrc/step_plan.hip),
which copies every host descriptor and knows the three things that differ between two steps of one shape: the input pointers /
scalars (patches), the dropout keys (key = f(seed, step, site)) and the stream handles.  Every later step of that shape is
`ader_step_enqueue(plan, inputs, step, main, side)`.

A plan holds raw device pointers of workspace tensors: it is retired with the descriptor cache whenever the workspace allocates or
evicts (`Engine._ws_gen`).  Everything that shapes the recorded arguments is part of the plan's key (`_Native._plan_key`); with
`Engine.plan_verify` every replayed step is instead run through Python again, recorded, and compared slot by slot and descriptor
byte by descriptor byte with what the plan would have issued (tests)."""
import ctypes
import struct

import numpy as np
import torch

from .. import _lib
from .._lib import AderDrop, AderStepBlob, AderStepKey, AderStepOp, AderStepPatch, call, ptr
from .common import IN_EXPOS, IN_EXTROW, IN_IDX_E, IN_IDX_T, IN_LR, IN_POS, IN_SEQ, IN_TEACHER, N_INPUTS, StepF, _check, dropout_key

_M64 = 0xFFFFFFFFFFFFFFFF


class Recorder:
    """What `_lib.call` and `Engine._edge` report while a step is being recorded: (launcher name, argument tuple) and
    (None, (waiting stream, awaited stream)) in issue order.  Holding the tuples keeps every ctypes descriptor alive until the plan
    has copied it."""

    def __init__(self):
        self.items = []

    def launch(self, name, args):
        self.items.append((name, args))

    def wait(self, waiter, other):
        self.items.append((None, (waiter, other)))


def _walk(ctype, off=0):
    """(offset, kind) of every pointer field ('ptr') and every AderDrop ('drop') inside a ctypes type."""
    if ctype is AderDrop:
        yield off, "drop"
    elif isinstance(ctype, type) and issubclass(ctype, ctypes.Structure):
        for name, ft in ctype._fields_:
            yield from _walk(ft, off + getattr(ctype, name).offset)
    elif isinstance(ctype, type) and issubclass(ctype, ctypes.Array):
        for i in range(ctype._length_):
            yield from _walk(ctype._type_, off + i * ctypes.sizeof(ctype._type_))
    elif ctype is ctypes.c_void_p:
        yield off, "ptr"


_WALKS = {}


def _walk_cached(ctype):
    w = _WALKS.get(ctype)
    if w is None:
        w = _WALKS[ctype] = tuple(_walk(ctype))
    return w


def _fbits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


class PlanError(RuntimeError):
    pass


def lower(items, lanes, ranges, seed, step, n_sites):
    """Recorded items -> (ops, blobs, patches, keys) in the C ABI's terms.
    lanes: {stream handle: lane}; ranges: [(input slot, base address, bytes)] of the step's input tensors; a pointer that falls into
    one of them becomes a patch (input + delta).  Dropout descriptors are found by type, their site by matching the recorded key."""
    site_of = {dropout_key(seed, step, s): s for s in range(n_sites)}
    ops = (AderStepOp * len(items))()
    blobs, patches, keys = [], [], []          # blobs: (op, arg, host object, ctype)

    def in_range(v):
        for slot, base, nb in ranges:
            if nb > 0 and base <= v < base + nb:
                return slot, v - base
        return None

    for i, (name, args) in enumerate(items):
        o = ops[i]
        if name is None:
            w, a = lanes.get(args[0]), lanes.get(args[1])
            if w is None or a is None:
                raise PlanError("stream edge between streams that are not the engine's two lanes")
            o.kind, o.stream, o.other = 1, w, a
            continue
        fn = _lib.step_fn_index(name)
        if fn < 0:
            raise PlanError("%s is not a step launcher" % name)
        argt = _lib._SIGS[name]
        if len(args) != len(argt) or len(args) > _lib.STEP_MAX_ARGS:
            raise PlanError("%s: %d arguments recorded, %d declared" % (name, len(args), len(argt)))
        lane = lanes.get(args[-1] or 0)
        if lane is None:
            raise PlanError("%s was launched on a stream that is not one of the engine's two lanes" % name)
        o.kind, o.fn, o.stream, o.n_args = 0, fn, lane, len(args)
        for j, (a, t) in enumerate(zip(args[:-1], argt[:-1])):
            if t is _lib.P:
                if a is None:
                    v = 0
                elif isinstance(a, int):
                    v = a
                    hit = in_range(v)
                    if hit is not None:
                        patches.append((-1, i, j, hit[0], 0, hit[1]))
                else:
                    obj = a._obj if hasattr(a, "_obj") else a          # byref(struct) or a ctypes array / struct instance
                    if not isinstance(obj, (ctypes.Structure, ctypes.Array)):
                        raise PlanError("%s argument %d: unsupported host object %r" % (name, j, type(a)))
                    b = len(blobs)
                    blobs.append((i, j, obj, type(obj)))
                    addr = ctypes.addressof(obj)
                    for off, kind in _walk_cached(type(obj)):
                        if kind == "ptr":
                            pv = ctypes.c_uint64.from_address(addr + off).value
                            hit = in_range(pv)
                            if hit is not None:
                                patches.append((b, 0, 0, hit[0], off, hit[1]))
                        else:
                            d = AderDrop.from_address(addr + off)
                            if d.thr:
                                site = site_of.get(d.key)
                                if site is None:
                                    raise PlanError("%s: a dropout key that is not this step's key of any site" % name)
                                keys.append((b, site, off + AderDrop.key.offset))
                    v = 0          # (filled by the plan with the address of its copy)
            elif t is _lib.F:
                v = _fbits(a)
                if isinstance(a, StepF):
                    patches.append((-1, i, j, a.slot, 0, 0))
            else:
                v = int(a) & _M64
            o.args[j] = v
    return ops, blobs, patches, keys


class StepPlan:
    """An AderStepPlan and what the host needs to drive it."""

    def __init__(self, items, lanes, ranges, seed, step, n_sites):
        ops, blobs, patches, keys = lower(items, lanes, ranges, seed, step, n_sites)
        self.n_ops = len(ops)
        self.names = [n for n, _ in items]
        self.blob_sizes = [ctypes.sizeof(b[3]) for b in blobs]
        cb = (AderStepBlob * max(len(blobs), 1))()
        for k, (i, j, obj, _) in enumerate(blobs):
            cb[k].op, cb[k].arg, cb[k].src, cb[k].bytes = i, j, ctypes.addressof(obj), ctypes.sizeof(obj)
        cp = (AderStepPatch * max(len(patches), 1))()
        for k, (b, i, j, slot, off, delta) in enumerate(patches):
            cp[k].blob, cp[k].op, cp[k].arg, cp[k].input, cp[k].offset, cp[k].delta = b, i, j, slot, off, delta
        ck = (AderStepKey * max(len(keys), 1))()
        for k, (b, site, off) in enumerate(keys):
            ck[k].blob, ck[k].site, ck[k].offset = b, site, off
        h = ctypes.c_void_p()
        call("ader_step_plan_create", ops, len(ops), cb, len(blobs), cp, len(patches), ck, len(keys), seed & 0xFFFFFFFF, ctypes.byref(h))
        self.handle = h
        self.inputs = (ctypes.c_uint64 * N_INPUTS)()
        self._enqueue = _lib.load().ader_step_enqueue
        self.n_patches, self.n_keys = len(patches), len(keys)

    def enqueue(self, step, main, side):
        rc = self._enqueue(self.handle, self.inputs, N_INPUTS, step & 0xFFFFFFFF, main, side)
        if rc != 0:
            op = call("ader_step_plan_failed_op", self.handle)
            raise _lib.AderHipError("native step: op %d (%s) failed with code %d" % (op, self.names[op] if 0 <= op < self.n_ops else "?", rc))

    def peek(self, step):
        """(ops, [descriptor bytes]) the plan would issue for self.inputs at `step` (nothing is launched)."""
        ops = (AderStepOp * self.n_ops)()
        bufs = [ctypes.create_string_buffer(max(n, 1)) for n in self.blob_sizes]
        outp = (ctypes.c_void_p * max(len(bufs), 1))(*[ctypes.addressof(b) for b in bufs])
        call("ader_step_plan_peek", self.handle, self.inputs, N_INPUTS, step & 0xFFFFFFFF, ops, outp)
        return ops, [b.raw[:n] for b, n in zip(bufs, self.blob_sizes)]

    def destroy(self):
        if self.handle is not None:
            _lib.load().ader_step_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


class _Slice:
    """The tail of a device index array from address `p` on, with the three methods the recorder asks of an input tensor."""

    def __init__(self, t, p):
        self.t, self.p = t, p

    def data_ptr(self):
        return self.p

    def numel(self):
        return max(0, self.t.numel() - (self.p - self.t.data_ptr()) // self.t.element_size())

    def element_size(self):
        return self.t.element_size()


class _Native:
    # ---------------------------------------------------------------------------------------- native step driver
    def _plan_state(self):
        if getattr(self, "_plans", None) is None or self._plans_gen != self._ws_gen:
            for p in (getattr(self, "_plans", None) or {}).values():
                if p is not None:
                    p.destroy()
            self._plans, self._plans_gen = {}, self._ws_gen
        return self._plans

    def _native_ok(self, kw):
        """The step forms whose launch sequence is pure launcher calls + stream edges (no torch kernel, no collective, no host read)."""
        return (self.native_step and self.dp_world == 1 and self.lx3 and self.lfast and self.seq_fused and self.fuse_adam
                and self.x3_update == "tab16" and self.kd_fast and self.grad_hook is None and self.grad_early_hook is None
                and self.timer is None and (self.ewc is None or self.ewc["lam"] == 0.0) and self._pending_loss is None
                and not (kw.get("teacher") is not None and kw.get("ex_pos") is not None))

    def _plan_key(self, tag, B_all, n_train, N, lr, rate, lambda_, teacher, has_expos, n_tg, n_eg, pack, extra=()):
        tk = None if teacher is None else (teacher.data_ptr(), tuple(teacher.shape), teacher.stride(0))
        return (tag, B_all, n_train, N, float(rate), float(lambda_), tk, has_expos, n_tg, n_eg, pack, self.row0, self.row0_ex,
                getattr(self, "_ex_row0_set", False), self.seed, self.prune_last, self.fuse_final_ln, self.late_side_stream,
                self.lists_side_stream, self.atb_batch, self.cache_descriptors, self.pack_window, self.pack_density, self._grad_hi <= N,
                self.beta1, self.beta2, self.eps, extra)

    def _native_step(self, seq, pos, max_item, lr, kw):
        """train_step through a launch plan; returns None when this step form is not plannable (the caller takes the Python path)."""
        if not self._native_ok(kw):
            return None
        teacher, ex_pos, ex_trow = kw.get("teacher"), kw.get("ex_pos"), kw.get("ex_trow")
        seq = self._seq_in(seq)
        pos = self._dev_i32(pos)
        B_all, n_train = seq.shape[0], pos.shape[0]
        n_ex = B_all - n_train
        N = int(max_item)
        if not (seq.dim() == 2 and seq.shape[1] == self.T and 0 < n_train <= B_all and 1 <= N <= self.item_num and N >= self._grad_hi):
            return None
        if n_ex > 0:
            if teacher is not None:
                ex_trow = self._dev_i32(ex_trow if ex_trow is not None else np.arange(n_ex))
                Np = teacher.shape[1]
                if not (teacher.dtype == torch.float32 and teacher.stride(1) == 1 and Np <= N and ex_trow.shape[0] == n_ex
                        and ((n_train + 127) // 128 + (n_ex + 127) // 128) * 128 <= self.MAX_ROWS_FAST):
                    return None
                self._teacher_lse(teacher, Np)           # (once per teacher tensor: outside the plan)
                ex_pos = None
            elif ex_pos is not None:
                ex_pos = self._dev_i32(ex_pos)
                if ex_pos.shape[0] != n_ex:
                    return None
            else:
                return None
        else:
            teacher = ex_pos = ex_trow = None
        pack = bool(self._use_pack(seq))
        key = self._plan_key("step", B_all, n_train, N, lr, kw.get("rate", 0.0), kw.get("lambda_", 0.0), teacher, ex_pos is not None,
                             kw.get("n_train_global"), kw.get("n_ex_global"), pack)
        kw2 = dict(kw)
        kw2.update(teacher=teacher, ex_pos=ex_pos, ex_trow=ex_trow)
        kw2.pop("ids_host", None)
        kw2.pop("pack_counts", None)
        ins = ((IN_SEQ, seq), (IN_POS, pos), (IN_EXPOS, ex_pos), (IN_EXTROW, ex_trow), (IN_TEACHER, teacher))
        return self._plan_run(key, ins, (), lr, lambda: self._train_step(seq, pos, N, lr, **kw2))

    def _plan_run(self, key, tensors, scalars, lr, python_step):
        """Replay the plan of `key`, or run `python_step` under the recorder and keep its plan.  tensors: ((slot, tensor | None), ...)
        the step's input tensors (patched by address range); scalars: ((slot, raw 64-bit value), ...)."""
        plans = self._plan_state()
        plan = plans.get(key, False)
        step = self.global_step
        if plan and not self.plan_verify:
            inp = plan.inputs
            for slot, t in tensors:
                if t is not None:
                    inp[slot] = t.data_ptr()
            for slot, v in scalars:
                inp[slot] = v
            inp[IN_LR] = _fbits(self._lr_t(lr))
            self._held = tensors                          # inputs stay alive until the next step has been enqueued behind this one
            plan.enqueue(step, self._main.cuda_stream, self._side.cuda_stream if self._side is not None else 0)
            self.b1p = np.float32(self.b1p * np.float32(self.beta1))
            self.b2p = np.float32(self.b2p * np.float32(self.beta2))
            self.global_step += 1
            self.plan_hits += 1
            return self.loss
        if plan is None and not self.plan_verify:
            return python_step()                          # a form that could not be planned: not tried again until the workspace changes
        rec = _lib.recorder = Recorder()
        gen0 = self._ws_gen
        self._keep_density = True
        try:
            out = python_step()
        finally:
            _lib.recorder = None
            self._keep_density = False
        self.plan_misses += 1
        if self._ws_gen != gen0 or self._plans_gen != gen0:
            return out                                    # the step allocated workspace: pointers of this recording may be stale
        lanes = {self._main.cuda_stream: 0}
        if self._side is not None:
            lanes[self._side.cuda_stream] = 1
        ranges = [(slot, t.data_ptr(), t.numel() * t.element_size()) for slot, t in tensors if t is not None]
        if plan:                                          # verify mode: this step ran through Python; compare with the plan's version
            self._plan_compare(plan, rec.items, lanes, ranges, step, tensors, scalars)
            return out
        try:
            plans[key] = StepPlan(rec.items, lanes, ranges, self.seed, step, 1 + 3 * self.L)
        except PlanError as e:
            plans[key] = None
            self.plan_errors.append(str(e))
        return out

    def _plan_compare(self, plan, items, lanes, ranges, step, tensors, scalars):
        """plan_verify: the Python-driven step just recorded vs. the plan's patched ops for the same inputs."""
        inp = plan.inputs
        for slot, t in tensors:
            if t is not None:
                inp[slot] = t.data_ptr()
        for slot, v in scalars:
            inp[slot] = v
        lr_bits = [_fbits(a) for n, args in items if n is not None for a in args if isinstance(a, StepF)]
        inp[IN_LR] = lr_bits[0] if lr_bits else 0
        ops, blobs = plan.peek(step)
        ref_ops, ref_blobs, _, _ = lower(items, lanes, ranges, self.seed, step, 1 + 3 * self.L)
        _check(len(ref_ops) == plan.n_ops, "plan_verify: %d ops recorded, the plan holds %d" % (len(ref_ops), plan.n_ops))
        blob_at = {(b[0], b[1]): k for k, b in enumerate(ref_blobs)}
        for i in range(plan.n_ops):
            a, b = ops[i], ref_ops[i]
            _check((a.kind, a.stream, a.n_args) == (b.kind, b.stream, b.n_args) and (a.kind == 1 or a.fn == b.fn) and
                   (a.kind == 0 or a.other == b.other), "plan_verify: op %d (%s) differs in kind / launcher / lane" % (i, plan.names[i]))
            if a.kind == 1:
                continue
            for j in range(a.n_args - 1):
                if (i, j) in blob_at:
                    k = blob_at[(i, j)]
                    obj = ref_blobs[k][2]
                    want = ctypes.string_at(ctypes.addressof(obj), ctypes.sizeof(obj))
                    _check(blobs[k] == want, "plan_verify: op %d (%s) descriptor argument %d differs from the Python-driven step"
                           % (i, plan.names[i], j))
                else:
                    _check(a.args[j] == b.args[j], "plan_verify: op %d (%s) argument %d: plan 0x%x, Python-driven step 0x%x"
                           % (i, plan.names[i], j, a.args[j], b.args[j]))
        self.plan_verified += 1

    # ---------------------------------------------------------------------------------------- device-fed steps
    def train_step_fed(self, feed, max_item, lr, rate=0.0, *, teacher=None, lambda_=0.0, onehot=False):
        """One train step whose batch is cut on the device (csrc/feed.hip) from the GPU-resident packed rows of the Samplers:
        feed = (rows_t, idx_t, o_t, n_t, Bt, rows_e, idx_e, o_e, n_e, Be) -- rows_* [*, T+1] int32 packed rows; idx_* int64 device
        index arrays of the epoch plan, the batch being idx[o : o + n]; Bt / Be the nominal row counts the batch is padded to with
        weight-0 rows (main.py --fixed_batches).  Exemplar rows are distilled against `teacher` [*, Np] (teacher row = exemplar index,
        main.py:220-221) or, onehot, replayed with their labels (ADER.py:126-131).  Same step as
        train_step(cat(seq_t, seq_e), pos_t, ..., n_train_global=n_t, n_ex_global=n_e) with the dropout counters of the unpadded batch."""
        self._refresh_stream()
        rows_t, idx_t, o_t, n_t, Bt, rows_e, idx_e, o_e, n_e, Be = feed
        if n_e <= 0 or Be <= 0 or (teacher is None and not onehot):
            n_e, Be, rows_e, idx_e, o_e = 0, 0, None, None, 0
        T = self.T
        _check(0 < n_t <= Bt and 0 <= n_e <= Be and rows_t.dtype == torch.int32 and rows_t.shape[1] == T + 1 and idx_t.dtype == torch.int64,
               "train_step_fed: packed rows [*, maxlen+1] int32, int64 index arrays, 0 < n_t <= Bt, 0 <= n_e <= Be")
        kd = teacher is not None and n_e > 0
        self._in_step = True
        try:
            seq = self.buf("fd_seq", (Bt + Be, T), torch.int32)
            pos = self.buf("fd_pos", (Bt,), torch.int32)
            ex_pos = self.buf("fd_expos", (Be,), torch.int32) if (n_e > 0 and not kd) else None
            ex_trow = self.buf("fd_trow", (Be,), torch.int32) if kd else None
            pt = idx_t.data_ptr() + 8 * o_t
            pe = (idx_e.data_ptr() + 8 * o_e) if n_e > 0 else 0
            # dropout counters / loss weights of the UNPADDED batch [n_t train rows | n_e exemplar rows] (main.py:229)
            self.row0, self.row0_ex, self._ex_row0_set = 0, n_t, Be > 0
            kw = dict(rate=rate, n_train_global=n_t)
            if n_e > 0:
                kw.update(n_ex_global=n_e, lambda_=lambda_)
                if kd:
                    kw.update(teacher=teacher, ex_trow=ex_trow)
                else:
                    kw.update(ex_pos=ex_pos)
            self._density_now = None

            def python_step():
                call("ader_feed_step", ptr(rows_t), pt, n_t, Bt, ptr(rows_e), pe, n_e, Be, T, ptr(seq), ptr(pos), ptr(ex_pos), ptr(ex_trow),
                     self._stream())
                return self._train_step(seq, pos, max_item, lr, **kw)

            N = int(max_item)
            if kd:
                self._teacher_lse(teacher, teacher.shape[1])
            ok = (self._native_ok(kw) and 1 <= N <= self.item_num and N >= self._grad_hi and
                  (not kd or (teacher.dtype == torch.float32 and teacher.stride(1) == 1 and teacher.shape[1] <= N and
                              ((Bt + 127) // 128 + (Be + 127) // 128) * 128 <= self.MAX_ROWS_FAST)))
            if not ok:
                return python_step()
            pack = bool(self._use_pack(seq))
            key = self._plan_key("fed", Bt + Be, Bt, N, lr, rate, lambda_ if n_e > 0 else 0.0, teacher if kd else None, ex_pos is not None,
                                 n_t, n_e, pack, (rows_t.data_ptr(), rows_e.data_ptr() if rows_e is not None else 0, Be))
            # the index slices are patched by VALUE range of the whole epoch arrays (pt / pe point into them)
            tens = ((IN_IDX_T, idx_t), (IN_IDX_E, idx_e if n_e > 0 else None))
            return self._plan_run_fed(key, tens, pt, pe, lr, python_step)
        finally:
            self._in_step = False

    def _plan_run_fed(self, key, tens, pt, pe, lr, python_step):
        """_plan_run for a device-fed step: the two index pointers are inputs given by address (base of the epoch array + offset)."""
        plans = self._plan_state()
        plan = plans.get(key, False)
        if plan and not self.plan_verify:
            inp = plan.inputs
            inp[IN_IDX_T], inp[IN_IDX_E] = pt, pe
            inp[IN_LR] = _fbits(self._lr_t(lr))
            self._held = tens
            plan.enqueue(self.global_step, self._main.cuda_stream, self._side.cuda_stream if self._side is not None else 0)
            self.b1p = np.float32(self.b1p * np.float32(self.beta1))
            self.b2p = np.float32(self.b2p * np.float32(self.beta2))
            self.global_step += 1
            self.plan_hits += 1
            return self.loss
        # record / verify through the generic path: the index arrays as address ranges, the slices' addresses as their inputs
        tensors = tuple((slot, None if t is None else _Slice(t, p)) for (slot, t), p in zip(tens, (pt, pe)))
        return self._plan_run(key, tensors, (), lr, python_step)
