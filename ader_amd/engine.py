"""Device state and step orchestration of the ADER hot path on one MI355X.

Holds the trainable state of the reference graph (item table [V,H], positional table [T,H], per block
LN1, dense Q/K/V, LN2, conv1d x2, final LN) in ONE flat float32 buffer (plus Adam m/v and the gradient in the
same layout, so dense Adam is a single flat kernel and the data-parallel gradient exchange is a single buffer),
the saved activations, and issues the HIP launchers of include/ader_hip.h on torch's current stream.

Nothing here computes on the CPU: without libader_hip.so / a GPU the constructor raises.
"""
import collections
import ctypes
import math
import os
import time as _time

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

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
                for rep in range(2):                              # (first pass: the runtime creates the hardware queue)
                    torch.cuda.synchronize(dev)
                    t0 = _time.perf_counter()
                    for i in range(24 if rep else 4):
                        x.add_(1.0)
                        s.wait_stream(main)
                        with torch.cuda.stream(s):
                            y.add_(1.0)
                        main.wait_stream(s)
                    torch.cuda.synchronize(dev)
                    lat = (_time.perf_counter() - t0) / 24
                for i in range(6):
                    big.add_(1.0)
                ev_m.record(main)
                with torch.cuda.stream(s):
                    y.add_(1.0)
                    ev_s.record(s)
                torch.cuda.synchronize(dev)
                overlaps = ev_s.elapsed_time(ev_m) > 0.02          # the side launch finished well before the main-stream work did
                cand = (not overlaps, lat, s)
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


class Engine:
    MAX_ROWS = 1024        # padded batch rows per launch of the exact-f32 logit kernels (per-row state in LDS) and of the eval paths
    MAX_ROWS_FAST = 4096   # ... of a train step whose logits run on the flash kernels (logits_dtype bf16 / x3): 128-row chunks

    def __init__(self, item_num, maxlen=50, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device="cuda:0",
                 logits_dtype="x3", gemm="x3", dp_rank=0, dp_world=1):
        if not torch.cuda.is_available():
            raise _lib.AderHipError("ader_amd.Engine needs an MI355X (no CPU fallback)")
        _lib.load()
        _check(hidden_units <= 159 and maxlen <= 64 and hidden_units % num_heads == 0,
               "ader_amd.Engine: hidden_units <= 159, maxlen <= 64, hidden_units %% num_heads == 0 (got %d, %d, %d)"
               % (hidden_units, maxlen, num_heads))
        self.item_num, self.T, self.H, self.L, self.heads = item_num, maxlen, hidden_units, num_blocks, num_heads
        self.V = item_num + 1
        self.seed = seed
        _check(logits_dtype in ("f32", "bf16", "x3"), "logits_dtype must be 'f32', 'bf16' or 'x3' (got %r)" % (logits_dtype,))
        # "bf16": logit GEMMs on v_mfma_f32_32x32x16_bf16 with bf16-rounded operands (fp32 master table, fp32 accumulate and
        # softmax); "x3": the same kernels at float32 grade -- every product as three bf16 MFMAs on hi/lo operand splits
        # (~2^-16 relative, the reference's fp32 arithmetic of ADER.py:91-93 on the bf16 matrix cores), no bf16 shadow;
        # "f32": the exact f32-MFMA kernels of logits.hip.  Distilled rows take the float32 kernels in every mode.
        self.logits_dtype = logits_dtype
        self.lfast = logits_dtype in ("bf16", "x3") and hidden_units % 2 == 0      # flash forward + fused table update
        self.lx3 = logits_dtype == "x3"
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        # data parallelism: the item table is split into `dp_world` equal shards of whole 128-item tiles (row 0 excluded)
        self.dp_rank, self.dp_world, self.dp_group = int(dp_rank), int(dp_world), None
        self.shard_items = -(-item_num // (128 * self.dp_world)) * 128
        self.V_alloc = 1 + self.dp_world * self.shard_items
        self.layout, self.P = param_layout(item_num, maxlen, hidden_units, num_blocks, table_rows_alloc=self.V_alloc)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.theta = torch.zeros(self.P, **f32)
        self.adam_m = torch.zeros(self.P, **f32)
        self.adam_v = torch.zeros(self.P, **f32)
        self.grad = torch.zeros(self.P, **f32)
        # bf16 shadow of the item table streamed by the bf16 logit GEMMs ([V][168], 336-B rows); Adam keeps it in sync
        self.shadow = (torch.zeros(self.V_alloc * 168, dtype=torch.bfloat16, device=self.device)
                       if logits_dtype == "bf16" and hidden_units % 2 == 0 else None)
        # block GEMMs: "x3" = bf16 hi/lo split on the bf16 matrix cores (float32-grade accuracy), "f32" = exact f32 MFMA
        _check(gemm in ("x3", "f32"), "gemm must be 'x3' or 'f32' (got %r)" % (gemm,))
        self.gemm_x3 = gemm == "x3" and hidden_units % 2 == 0 and hidden_units <= 150
        # whole forward stack in one launch (seq_fwd.hip); the per-op kernels remain for the shapes it does not cover
        self.seq_fused = (self.gemm_x3 and num_heads == 1 and maxlen <= 64 and num_blocks <= _lib.SEQ_MAXL)
        # packed session tiles (csrc/seqp_*.hip): the session kernels run on the REAL positions only, several short sessions per
        # 64-row tile.  True / False, or "auto": packed when the batch is sparse enough to pay for it -- decided from the host copy of
        # the batch when the caller passes one (numpy input_seq), else from Engine.pack_density, the fraction of real positions the
        # feeder announces (Sampler.to_device / main.py set it from the dataset; None = unknown = not packed)
        self.pack_sessions = "auto"
        self.pack_density = None
        self.pack_window = (17, 49, 224)   # stream window of the short class: at least / at most / tile count aimed at (seqp_plan.hip)
        self._pack_now = False
        self.lists_side_stream = True      # build the sparse lists under the block kernels
        self.late_side_stream = True       # small-parameter gradients / Adam run beside the (HBM-bound) fused table update
        self._late, self._late_on, self._late_force = [], False, False
        self._st_ptr, self._main, self._in_step = None, None, False
        self._pin = {}
        self.atb_batch = True          # x3 mode: all weight-gradient products of a backward pass in one launch
        self._atb_q = []
        self.attn_x3 = gemm == "x3" and (hidden_units // num_heads) % 2 == 0      # bf16x3 attention core (attn_x3.hip)
        self._wnames = ["b%d.%s" % (l, w) for l in range(num_blocks) for w in ("wq", "wk", "wv", "w1", "w2")]
        self._widx = {k: i for i, k in enumerate(self._wnames)}
        self.wbf = None
        if self.gemm_x3:
            self._woffs = torch.tensor([self.layout[k][0] for k in self._wnames], dtype=torch.int64, device=self.device)
            self.wbf = torch.zeros(call("ader_wprep_elems", len(self._wnames)), dtype=torch.bfloat16, device=self.device)
            self._wplane = 4 * 160 * 168 * 2   # bytes per weight
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.loss = torch.zeros(1, **f32)
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.b1p, self.b2p = np.float32(self.beta1), np.float32(self.beta2)   # TF keeps beta powers in float32 variables
        self.global_step = 0
        self.row0 = 0            # global index of local row 0 (data-parallel shard offset of the dropout counters)
        self.split_rows, self.row0_ex = None, 0     # ... and of the first local exemplar row (set per step under data parallelism)
        self._grad_hi = 0
        self._ws = {}
        self._ws_store = {}
        self._ws_gen, self._dcache, self._dcache_gen = 0, {}, -1
        self.cache_descriptors = True      # packed session kernels: descriptors of a step shape are built once (host time, see _dc)
        self.grad_hook = None    # called between backward and Adam (data-parallel gradient exchange)
        # data-parallel dense path: called right after the logits backward has written the table gradient's dense term (99.8 % of
        # the gradient bytes) so that its all-reduce runs UNDER the blocks backward; returns the pending collectives
        self.grad_early_hook = None
        self._early, self._dp_rows = None, None
        self.ewc = None          # EWC baseline (EWC.py): dict(F=, prev=, lam=) -> quadratic penalty added between backward and Adam
        self.timer = None        # optional SectionTimer
        self.prune_last = True   # final block: query/FFN path only for position T-1 (exact; see forward())
        # float32-grade flash forward: the backward of the final LayerNorm runs inside the merge launch of the logit forward (the
        # workgroup that forms a row of dRep also forms LN_f'(dRep)): k_ln_bwd and its kernel boundary leave the critical path
        self.fuse_final_ln = True
        self._lnf_done = None
        # single-GPU bf16-logits steps: apply Adam to the item table inside the table-gradient GEMM (the table gradient
        # is never written to memory); needs the complete gradient locally, so it is off whenever a grad_hook is set
        self.fuse_adam = True
        self.dp_sharded = True   # dp_world > 1 with bf16 logits: row-sharded table update instead of a dense all-reduce
        # dp_world > 1, vanilla bf16 steps: "catalog" = every rank OWNS 1/W of the table rows (parameters, Adam state, shadow),
        # streams only those in the logit kernels and never receives the other ranks' rows except the few its inputs need
        # (_train_step_catalog); "replicated" = every rank holds the whole table (the two schemes of dist.py)
        self.dp_mode = "replicated"
        # catalog mode: dp_pack = only owned rows travel (uneven all-to-all; costs one host sync per step -- the host cannot run
        # ahead of the GPU any more -- and ~20 small bookkeeping launches) instead of a dense 15 MB block per peer.  Default:
        # from 8 ranks on, where the dense blocks add up to ~210 MB received per rank and step against ~26 MB packed; below
        # that the dense exchange keeps the step free of host synchronisation (ADER_DP_PACK=0/1 overrides).
        _p = os.environ.get("ADER_DP_PACK")
        self.dp_pack = (self.dp_world >= 8) if _p is None else (_p == "1")
        self.comm_syncs = None   # catalog packed exchange: host synchronisations of the last step (0 with ids_host, else 1)
        # ... with host-side split sizes (ids_host / pack_counts): compare them with the device plan every n-th step (0 = never; a
        # host synchronisation each time -- tests and the first steps of a new data pipeline)
        self.check_pack_counts = int(os.environ.get("ADER_CHECK_PACK_COUNTS", "0"))
        self.kd_split = True     # distilled steps: train rows on the bf16 / fused path, exemplar rows on the exact-f32 kernels
        self.kd_fast = True      # ... exemplar rows on the flash path too (teacher readout + fused KD update)
        # bf16 mode, fused table update: "sh" = k_tab16 (operand from the shadow rows, three workgroups per CU: the faster form),
        # "resident" = k_tab_upd (theta tile read once and kept in LDS, no shadow read: 8 % fewer bytes, 10 % slower)
        self.bf16_update = "sh"
        # x3 mode, fused table update: "tab16" = k_tab16x3 (16x16x32 tiles, three workgroups per CU, rep chunks by LDS-DMA as
        # conflict-free LDS images: the faster form), "tab32" = the round-2 kernel k_tab_upd<X3> (kept for kernel-vs-kernel tests)
        self.x3_update = "tab16"     # (an attribute, not an environment variable: a stray setting must not switch kernels)
        self._table_stale = False
        self._pending_loss, self._img_ready = None, False     # late loss sum / operand images of a deferred fused update
        self._mv_sharded = False   # dp: Adam m/v of the table are current only for the rank's own rows (see _gather_if_sharded)
        # raw device addresses of every parameter / gradient tensor (the flat buffers never move)
        self._pp = {k: self.theta.data_ptr() + 4 * off for k, (off, _) in self.layout.items()}
        self._gp = {k: self.grad.data_ptr() + 4 * off for k, (off, _) in self.layout.items()}
        self.init_params(seed)

    # ---------------------------------------------------------------------------------------- parameters
    def view(self, buf, name):
        off, shp = self.layout[name]
        return buf[off:off + int(np.prod(shp))].view(*shp)

    def param(self, name):
        return self.view(self.theta, name)

    def gradient(self, name):
        return self.view(self.grad, name)

    def init_params(self, seed):
        """TF defaults at the reference call sites: Glorot-uniform tables/kernels, zero biases, LN gamma=1 beta=0."""
        self._refresh_stream()
        g = torch.Generator().manual_seed(seed)
        for name, (off, shp) in self.layout.items():
            base = name.split(".")[-1]
            if base in ("emb", "pos", "wq", "wk", "wv", "w1", "w2"):
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                t = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()
            elif base.endswith("_g"):
                t = torch.ones(shp)
            else:
                t = torch.zeros(shp)
            self.param(name).copy_(t)
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.b1p, self.b2p = np.float32(self.beta1), np.float32(self.beta2)
        self.global_step = 0
        self.refresh_shadow()

    def refresh_shadow(self):
        """Rebuild the bf16 copies derived from the fp32 master parameters (after init / load / any direct write):
        the shadow item table of the bf16 logit GEMMs and the hi/lo weight planes of the bf16x3 block GEMMs."""
        self._refresh_stream()
        if self.shadow is not None:
            call("ader_lbf_shadow_refresh", self._pp["emb"], ptr(self.shadow), self.V, self.H, self._stream())
        self.refresh_weights()

    def refresh_weights(self):
        self._refresh_stream()
        if self.wbf is not None:
            call("ader_wprep", ptr(self.theta), ptr(self._woffs), len(self._wnames), self.H, ptr(self.wbf), self._stream())

    def load_params(self, params):
        for k, v in params.items():
            self.param(k).copy_(torch.as_tensor(v, dtype=torch.float32))
        self.refresh_shadow()

    def export_params(self):
        return {k: self.param(k).detach().cpu().clone() for k in self.layout}

    def _gather_if_sharded(self):
        """Data-parallel sharded / catalog steps update Adam m/v (and, in catalog mode, theta) of the rank's OWN table rows only.
        Before anything that needs the whole optimiser state on every rank -- a checkpoint, or a dense Adam step (distilled
        periods) -- the shards are all-gathered.  A collective: every rank reaches these points together (state_dict / adam)."""
        if self.dp_world > 1 and self._mv_sharded:
            self.gather_table_state()
            self._mv_sharded = False

    def state_dict(self, to_cpu=False):
        """Per-variable tensors of theta / Adam m / Adam v (the reference's Saver stores per-variable tensors too, main.py:209):
        the item table is trimmed to its item_num+1 real rows, so a state written with W ranks loads with any other W.
        Tensors stay on the device unless `to_cpu` (Saver.save)."""
        self.sync_table()
        self._gather_if_sharded()

        def pack(buf):
            out = {}
            for k in self.layout:
                t = self.view(buf, k).detach().clone()
                out[k] = t.cpu() if to_cpu else t
            return out

        return {"format": 2, "theta": pack(self.theta), "m": pack(self.adam_m), "v": pack(self.adam_v),
                "b1p": float(self.b1p), "b2p": float(self.b2p), "global_step": self.global_step}

    def load_state_dict(self, sd):
        self._refresh_stream()
        if isinstance(sd["theta"], dict):
            for buf, key in ((self.theta, "theta"), (self.adam_m, "m"), (self.adam_v, "v")):
                missing = set(self.layout) - set(sd[key])
                if missing:
                    raise KeyError("state_dict[%r] lacks %s" % (key, sorted(missing)))
                for k in self.layout:
                    dst, src = self.view(buf, k), sd[key][k]
                    if tuple(src.shape) != tuple(dst.shape):
                        raise ValueError("state_dict[%r][%r]: shape %s, expected %s" % (key, k, tuple(src.shape), tuple(dst.shape)))
                    dst.copy_(src)
        else:       # flat buffers of an earlier build (valid only for the same number of ranks)
            self.theta.copy_(sd["theta"])
            self.adam_m.copy_(sd["m"])
            self.adam_v.copy_(sd["v"])
        self.b1p, self.b2p = np.float32(sd["b1p"]), np.float32(sd["b2p"])
        self.global_step = int(sd["global_step"])
        self._mv_sharded = False
        self._table_stale = False
        self.refresh_shadow()

    # ---------------------------------------------------------------------------------------- workspaces
    def buf(self, name, shape, dtype=torch.float32, zero=False):
        """Persistent workspace tensor.  zero=True clears it when it is (re)allocated: the session kernels skip a session's leading
        padding rows, whose activation / gradient rows then keep whatever the buffer held -- stale values are harmless (every use
        multiplies them by an exact zero), the NaN bit patterns of fresh memory are not.  (The clear is a torch fill on torch's
        current stream: only for buffers first touched on the main stream -- the launchers' side streams are not torch's.)"""
        t = self._ws.get(name)
        if t is not None and t.dtype == dtype and t.shape == shape:      # (allocation-free: ~70 calls per step)
            return t
        key = (tuple(int(d) for d in shape), dtype)
        # A name keeps one tensor PER SHAPE (the few most recent): batches whose row count alternates -- the exemplar sampler's
        # ragged batches -- must not reallocate, let alone refill, ~40 buffers per step; and two shapes of one name that are live in
        # the same step (local and global batch of the data-parallel schemes) must not share memory.
        per = self._ws_store.setdefault(name, collections.OrderedDict())
        t = per.get(key)
        if t is None:
            if zero and self._main is not None and self._st_ptr is not None and self._st_ptr != self._main.cuda_stream:
                raise RuntimeError("Engine.buf(%r, zero=True) first requested inside a side-stream section: its fill would run on "
                                   "the main stream, unordered with the side-stream kernels" % name)
            t = (torch.zeros if zero else torch.empty)(key[0], dtype=dtype, device=self.device)
            per[key] = t
            self._ws_gen += 1          # (cached launch descriptors hold raw pointers of workspace tensors: any allocation or eviction retires them)
            while len(per) > 4:
                per.popitem(last=False)
        else:
            per.move_to_end(key)
        self._ws[name] = t
        return t

    def _stream(self):
        """Raw handle of the stream the launchers enqueue on: torch's current stream, looked up once per public entry point
        (_refresh_stream) and switched explicitly around the side-stream blocks -- torch.cuda.current_stream() costs ~10 us
        and is needed ~30 times per step."""
        return self._st_ptr if self._st_ptr is not None else torch.cuda.current_stream().cuda_stream

    def _refresh_stream(self):
        if self._in_step:            # nested entry points of one train step: the stream was looked up at its start
            return
        self._main = torch.cuda.current_stream()
        self._st_ptr = self._main.cuda_stream

    class _OnStream:
        """Launch on `stream` inside the block.  The launchers take the stream handle explicitly, so only Engine._stream() has to
        change; torch's own current stream is switched as well only while a SectionTimer is recording (its events go to torch's
        current stream) -- the torch.cuda.stream() context costs ~20 us of host time per use, three times per step."""

        def __init__(self, eng, stream):
            self.eng, self.stream = eng, stream
            self.ctx = torch.cuda.stream(stream) if eng.timer is not None else None

        def __enter__(self):
            if self.ctx is not None:
                self.ctx.__enter__()
            self.prev, self.eng._st_ptr = self.eng._st_ptr, self.stream.cuda_stream

        def __exit__(self, *exc):
            self.eng._st_ptr = self.prev
            return self.ctx.__exit__(*exc) if self.ctx is not None else False

    def _sec(self, name):
        return self.timer.section(name) if self.timer is not None else _NULL

    def _dev_i32(self, x):
        """int32 device tensor of a host array / tensor.  Host arrays go through a small ring of pinned staging buffers and an
        asynchronous copy: a pageable .to(device) would block the host until the GPU has drained (one sync per step)."""
        if isinstance(x, torch.Tensor):
            if x.device == self.device and x.dtype == torch.int32 and x.is_contiguous():
                return x
            return x.to(device=self.device, dtype=torch.int32).contiguous()
        a = np.ascontiguousarray(x, dtype=np.int32)
        ring = self._pin.setdefault(a.shape, {"bufs": [], "evs": [], "i": 0})
        if len(ring["bufs"]) < 8:
            ring["bufs"].append(torch.empty(a.shape, dtype=torch.int32).pin_memory())
            ring["evs"].append(None)
            k = len(ring["bufs"]) - 1
        else:
            k = ring["i"] = (ring["i"] + 1) % 8
            if ring["evs"][k] is not None:
                ring["evs"][k].synchronize()          # the copy that last used this buffer (8 transfers ago) is long done
        ring["bufs"][k].numpy()[...] = a
        out = ring["bufs"][k].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["evs"][k] = ev
        return out

    # ---------------------------------------------------------------------------------------- forward
    def _drop(self, step, site, rate, training, per_row):
        """Dropout descriptor of a site for this step: counters keyed by the GLOBAL row, so W ranks draw the masks of
        one process.  Engine.row0 = global index of local row 0; with exemplar rows in the batch (Engine.split_rows local train
        rows first) the rows after them continue at global row Engine.row0_ex."""
        return _Drop(self.seed, step, site, rate, training, per_row, self.row0, self.split_rows, self.row0_ex)

    _ND = (None,)

    def _dc(self, key):
        """Descriptor cache of the packed session path: the ctypes structs, argument tuples and saved-activation dicts of a step
        shape are built once and reused -- per step only the batch pointer and the dropout keys change (the host spent ~90 us
        per step rebuilding them: ~90 workspace lookups, ~170 pointer conversions).  Every entry is retired as soon as the
        workspace allocates or evicts anything (Engine.buf bumps _ws_gen), so a raw pointer never outlives its tensor."""
        if self._dcache_gen != self._ws_gen:
            self._dcache, self._dcache_gen = {}, self._ws_gen
        return self._dcache.get(key)

    def _dc_put(self, key, val):
        if self._dcache_gen == self._ws_gen:      # (building the entry may itself have allocated: then it is not kept)
            self._dcache[key] = val
        return val

    def _rekey(self, drops, step):
        """New step, same shape: the persistent dropout descriptors of a cached entry take the step's keys in place."""
        for dr, site in drops:
            if dr.c.thr:
                dr.c.key = dropout_key(self.seed, step, site)

    def _gemm(self, A, wname, bname, C, aux, seq, M, epi, trans=0, drop=None, rmap=(1, 0)):
        d = drop.args() if drop is not None else self._ND
        bias = self._pp[bname] if bname is not None else None
        if self.gemm_x3:
            call("ader_gemm_x3", ptr(A), self.wbf.data_ptr() + self._widx[wname] * self._wplane, bias, ptr(C), ptr(aux), ptr(seq),
                 M, self.H, epi, trans, rmap[0], rmap[1], *d, self._stream())
        else:
            call("ader_gemm_rows", ptr(A), self._pp[wname], bias, ptr(C), ptr(aux), ptr(seq), M, self.H, epi, trans, rmap[0],
                 rmap[1], *d, self._stream())

    def _atb(self, A, G, wname, bname, slab, M, pack=None):
        """dW = A^T.G, db = colsum(G).  In x3 mode the products are queued (A and G stay untouched until the end of the
        backward pass) and issued as one batched launch by _atb_flush.  pack: the operands are in tile order (packed session
        kernels): the plan tells the product which rows exist."""
        if self.gemm_x3 and self.atb_batch:
            self._atb_q.append((A, G, self._gp[wname], self._gp[bname], M, pack))
            if len(self._atb_q) == 16:
                self._atb_flush()
            return
        fn = "ader_gemm_atb_x3" if self.gemm_x3 else "ader_gemm_atb"
        call(fn, ptr(A), ptr(G), ptr(slab), self._gp[wname], self._gp[bname], M, self.H, self._stream())

    def _late_call(self, name, *args):
        """A launch whose result only feeds the small-parameter update: issued now, or queued for the side stream that runs
        beside the fused table update (_fused_table_adam)."""
        if self._late_on:
            self._late.append((name, args))
        else:
            call(name, *args, self._stream())

    def _flush_late(self):
        """Issue the queued small launches; the LayerNorm partial reductions of all blocks go out as ONE batched launch."""
        late, self._late = self._late, []
        red = [a for n, a in late if n == "ader_reduce_slabs"]
        if len(red) > 1:
            for i0 in range(0, len(red), 8):        # (a launch takes up to 8 jobs: 2 per block + the final LayerNorm)
                rr = red[i0:i0 + 8]
                n = len(rr)
                VP, LA, IA = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
                call("ader_reduce_slabs_batch", VP(*[a[0] for a in rr]), LA(*[a[1] for a in rr]), IA(*[a[2] for a in rr]),
                     IA(*[a[3] for a in rr]), IA(*[a[4] for a in rr]), IA(*[a[5] for a in rr]), VP(*[a[6] for a in rr]),
                     VP(*[a[7] for a in rr]), n, self._stream())
            late = [(nm, a) for nm, a in late if nm != "ader_reduce_slabs"]
        for name, args in late:
            call(name, *args, self._stream())

    def _atb_flush(self):
        q, self._atb_q = self._atb_q, []
        if not q:
            return
        n = len(q)
        VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
        Ms = IA(*[it[4] for it in q])
        if any(it[5] is not None for it in q):
            # tile-ordered operands: bound the rows on the device, mask the unwritten rows of every tile, and share the workgroups
            # out by the rows expected to exist (the compact tensors of a pruned block are plain: every row exists)
            Mp = IA(*[(min(it[4], it[5]["est"]) if it[5] is not None else it[4]) for it in q])
            Md = VP(*[(it[5]["hdr"].data_ptr() + 4 if it[5] is not None else None) for it in q])
            Tr = VP(*[(it[5]["trows"].data_ptr() if it[5] is not None else None) for it in q])
            slabs = call("ader_gemm_atb_batch_slabs", Mp, n)
            slab = self.buf("atb_slab", (slabs * 160 * 160,))
            call("ader_gemm_atb_x3_batch_pk", VP(*[it[0].data_ptr() for it in q]), VP(*[it[1].data_ptr() for it in q]),
                 VP(*[it[2] for it in q]), VP(*[it[3] for it in q]), Ms, Mp, Md, Tr, n, ptr(slab), self.H, self._stream())
            return
        slabs = call("ader_gemm_atb_batch_slabs", Ms, n)
        slab = self.buf("atb_slab", (slabs * 160 * 160,))
        call("ader_gemm_atb_x3_batch", VP(*[it[0].data_ptr() for it in q]), VP(*[it[1].data_ptr() for it in q]),
             VP(*[it[2] for it in q]), VP(*[it[3] for it in q]), Ms, n, ptr(slab), self.H, self._stream())

    def forward(self, seq, training=False, rate=0.0, step=0, save=False):
        """seq int32 [B,T] (device).  Returns rep [B,H]; with save=True keeps activations for backward.
        The final block computes only position T-1 of its query / FFN path (Engine.prune_last): the representation is
        x[:, -1, :] (ADER.py:85) and rows interact only through K/V, so the other T-1 rows of that block are dead work."""
        self._refresh_stream()
        if save:
            self._lnf_done = None          # (a fused final-LayerNorm backward belongs to the forward it followed)
        if self.seq_fused:
            if self._use_pack(seq):
                return self._forward_packed(seq, training, rate, step, save)
            return self._forward_fused(seq, training, rate, step, save)
        B, T, H, L = seq.shape[0], self.T, self.H, self.L
        rows = B * T
        st = self._stream()
        tag = "t" if save else "e"
        A = {"B": B, "seq": seq, "rate": rate, "training": training, "step": step}
        per_row = T * H
        pp = self._pp
        d0 = self._drop(step, SITE_EMB, rate, training, per_row)
        x = self.buf(tag + "x0", (rows, H))
        call("ader_embed_fwd", ptr(seq), pp["emb"], pp["pos"], ptr(x), rows, T, H, self.V, *d0.args(), ptr(self.status), st)
        A["d_emb"] = d0
        last_map = (T, T - 1)
        for l in range(L):
            p = "b%d." % l
            n = lambda s: "%s%d%s" % (tag, l, s)   # noqa: E731
            pruned = self.prune_last and l == L - 1
            da = self._drop(step, site_attn(l), rate, training, self.heads * T * T)
            d1 = self._drop(step, site_ffn1(l), rate, training, per_row)
            d2 = self._drop(step, site_ffn2(l), rate, training, per_row)
            q_in = self.buf(n("qin"), (rows, H))
            mean1, std1 = self.buf(n("m1"), (rows,)), self.buf(n("s1"), (rows,))
            kmask, qmask = self.buf(n("km"), (rows,)), self.buf(n("qm"), (rows,))
            call("ader_ln_fwd", ptr(x), H, ptr(q_in), H, pp[p + "ln1_g"], pp[p + "ln1_b"], ptr(mean1), ptr(std1), ptr(kmask),
                 ptr(qmask), rows, H, st)
            K, Vv = self.buf(n("K"), (rows, H)), self.buf(n("V"), (rows, H))
            self._gemm(x, p + "wk", p + "bk", K, None, None, rows, EPI_BIAS)
            self._gemm(x, p + "wv", p + "bv", Vv, None, None, rows, EPI_BIAS)
            if not pruned:
                Q = self.buf(n("Q"), (rows, H))
                self._gemm(q_in, p + "wq", p + "bq", Q, None, None, rows, EPI_BIAS)
                x1 = self.buf(n("x1"), (rows, H))
                Pm = self.buf(n("P"), (B * self.heads * T * T,))
                call("ader_attn_x3_fwd" if self.attn_x3 else "ader_attn_fwd", ptr(Q), ptr(K), ptr(Vv), ptr(q_in), ptr(kmask),
                     ptr(qmask), ptr(x1), ptr(Pm), B, T, H, self.heads, *da.args(), st)
                y = self.buf(n("y"), (rows, H))
                mean2, std2 = self.buf(n("m2"), (rows,)), self.buf(n("s2"), (rows,))
                call("ader_ln_fwd", ptr(x1), H, ptr(y), H, pp[p + "ln2_g"], pp[p + "ln2_b"], ptr(mean2), ptr(std2), None, None,
                     rows, H, st)
                h1d = self.buf(n("h1"), (rows, H))
                self._gemm(y, p + "w1", p + "b1", h1d, None, None, rows, EPI_BIAS_RELU_DROP, drop=d1)
                x2 = self.buf(n("x2"), (rows, H))
                self._gemm(h1d, p + "w2", p + "b2", x2, y, seq, rows, EPI_BIAS_DROP_RES_MASK, drop=d2)
                A[l] = dict(pruned=False, x=x, q_in=q_in, mean1=mean1, std1=std1, kmask=kmask, qmask=qmask, Q=Q, K=K, V=Vv,
                            P=Pm, x1=x1, y=y, mean2=mean2, std2=std2, h1d=h1d, da=da, d1=d1, d2=d2)
                x = x2
            else:
                # compact [B,H] tensors of row T-1
                x_last = x.view(B, T, H)[:, T - 1, :]
                qin_l = self.buf(n("qinL"), (B, H))
                m1l, s1l, qml = self.buf(n("m1L"), (B,)), self.buf(n("s1L"), (B,)), self.buf(n("qmL"), (B,))
                call("ader_ln_fwd", ptr(x_last), T * H, ptr(qin_l), H, pp[p + "ln1_g"], pp[p + "ln1_b"], ptr(m1l), ptr(s1l),
                     None, ptr(qml), B, H, st)
                Ql = self.buf(n("QL"), (B, H))
                self._gemm(qin_l, p + "wq", p + "bq", Ql, None, None, B, EPI_BIAS)
                x1l = self.buf(n("x1L"), (B, H))
                Pl = self.buf(n("PL"), (B * self.heads * T,))
                call("ader_attn_last_fwd", ptr(Ql), ptr(K), ptr(Vv), ptr(qin_l), ptr(kmask), ptr(qml), ptr(x1l), ptr(Pl), B, T, H,
                     self.heads, *da.args(), st)
                yl = self.buf(n("yL"), (B, H))
                m2l, s2l = self.buf(n("m2L"), (B,)), self.buf(n("s2L"), (B,))
                call("ader_ln_fwd", ptr(x1l), H, ptr(yl), H, pp[p + "ln2_g"], pp[p + "ln2_b"], ptr(m2l), ptr(s2l), None, None,
                     B, H, st)
                h1l = self.buf(n("h1L"), (B, H))
                self._gemm(yl, p + "w1", p + "b1", h1l, None, None, B, EPI_BIAS_RELU_DROP, drop=d1, rmap=last_map)
                x2l = self.buf(n("x2L"), (B, H))
                self._gemm(h1l, p + "w2", p + "b2", x2l, yl, seq, B, EPI_BIAS_DROP_RES_MASK, drop=d2, rmap=last_map)
                A[l] = dict(pruned=True, x=x, q_in=qin_l, mean1=m1l, std1=s1l, kmask=kmask, qmask=qml, Q=Ql, K=K, V=Vv, P=Pl,
                            x1=x1l, y=yl, mean2=m2l, std2=s2l, h1d=h1l, da=da, d1=d1, d2=d2)
                x = x2l
        rep = self.buf(tag + "rep", (B, H))
        meanf, stdf = self.buf(tag + "mf", (B,)), self.buf(tag + "sf", (B,))
        if self.prune_last:
            call("ader_ln_fwd", ptr(x), H, ptr(rep), H, pp["lnf_g"], pp["lnf_b"], ptr(meanf), ptr(stdf), None, None, B, H, st)
        else:
            x_last = x.view(B, T, H)[:, T - 1, :]
            call("ader_ln_fwd", ptr(x_last), T * H, ptr(rep), H, pp["lnf_g"], pp["lnf_b"], ptr(meanf), ptr(stdf), None, None,
                 B, H, st)
        A.update(xL=x, rep=rep, meanf=meanf, stdf=stdf)
        if save:
            self._act = A
        return rep

    def _forward_fused(self, seq, training, rate, step, save):
        """forward() as one launch of ader_seq_fwd (seq_fwd.hip): same buffers, layouts and saved-activation dict as the
        per-op path above, so the backward pass does not care which one ran."""
        B, T, H, L = seq.shape[0], self.T, self.H, self.L
        rows = B * T
        tag = "t" if save else "e"
        A = {"B": B, "seq": seq, "rate": rate, "training": training, "step": step}
        per_row = T * H
        pp = self._pp
        d = _lib.AderSeqFwd()
        d0 = self._drop(step, SITE_EMB, rate, training, per_row)
        A["d_emb"] = d0
        x = self.buf(tag + "x0", (rows, H), zero=True)
        rep = self.buf(tag + "rep", (B, H))
        meanf, stdf = self.buf(tag + "mf", (B,)), self.buf(tag + "sf", (B,))
        d.seq, d.emb, d.pos, d.x0, d.status = ptr(seq), pp["emb"], pp["pos"], ptr(x), ptr(self.status)
        d.lnf_g, d.lnf_b, d.rep, d.meanf, d.stdf = pp["lnf_g"], pp["lnf_b"], ptr(rep), ptr(meanf), ptr(stdf)
        d.B, d.T, d.H, d.V, d.L = B, T, H, self.V, L
        d.sqrtH = float(np.sqrt(np.float32(H)))
        d.sqrt_dh = float(np.sqrt(np.float32(H // self.heads)))
        d.d_emb = d0.c
        for l in range(L):
            p = "b%d." % l
            n = lambda s: "%s%d%s" % (tag, l, s)   # noqa: E731
            pruned = self.prune_last and l == L - 1
            da = self._drop(step, site_attn(l), rate, training, self.heads * T * T)
            d1 = self._drop(step, site_ffn1(l), rate, training, per_row)
            d2 = self._drop(step, site_ffn2(l), rate, training, per_row)
            M, sfx = (B, "L") if pruned else (rows, "")
            kmask = self.buf(n("km"), (rows,), zero=True)
            K, Vv = self.buf(n("K"), (rows, H), zero=True), self.buf(n("V"), (rows, H), zero=True)
            q_in = self.buf(n("qin" + sfx), (M, H), zero=True)
            mean1, std1, qmask = self.buf(n("m1" + sfx), (M,), zero=True), self.buf(n("s1" + sfx), (M,), zero=True), self.buf(n("qm" + sfx), (M,), zero=True)
            Q, x1, y = self.buf(n("Q" + sfx), (M, H), zero=True), self.buf(n("x1" + sfx), (M, H), zero=True), self.buf(n("y" + sfx), (M, H), zero=True)
            Pm = self.buf(n("P" + sfx), (B * self.heads * T * (1 if pruned else T),), zero=True)
            mean2, std2 = self.buf(n("m2" + sfx), (M,), zero=True), self.buf(n("s2" + sfx), (M,), zero=True)
            h1d, x2 = self.buf(n("h1" + sfx), (M, H), zero=True), self.buf(n("x2" + sfx), (M, H), zero=True)
            k = d.blk[l]
            for i, w in enumerate(("wq", "wk", "wv", "w1", "w2")):
                k.w[i] = self.wbf.data_ptr() + self._widx[p + w] * self._wplane
            for i, bn in enumerate(("bq", "bk", "bv", "b1", "b2")):
                k.bias[i] = pp[p + bn]
            k.ln1_g, k.ln1_b, k.ln2_g, k.ln2_b = pp[p + "ln1_g"], pp[p + "ln1_b"], pp[p + "ln2_g"], pp[p + "ln2_b"]
            k.q_in, k.mean1, k.std1, k.kmask, k.qmask = ptr(q_in), ptr(mean1), ptr(std1), ptr(kmask), ptr(qmask)
            k.Q, k.K, k.V, k.P, k.x1, k.y = ptr(Q), ptr(K), ptr(Vv), ptr(Pm), ptr(x1), ptr(y)
            k.mean2, k.std2, k.h1d, k.x2 = ptr(mean2), ptr(std2), ptr(h1d), ptr(x2)
            k.d_attn, k.d_ffn1, k.d_ffn2 = da.c, d1.c, d2.c
            k.pruned = 1 if pruned else 0
            A[l] = dict(pruned=pruned, x=x, q_in=q_in, mean1=mean1, std1=std1, kmask=kmask, qmask=qmask, Q=Q, K=K, V=Vv, P=Pm,
                        x1=x1, y=y, mean2=mean2, std2=std2, h1d=h1d, da=da, d1=d1, d2=d2)
            x = x2
        call("ader_seq_fwd", ctypes.byref(d), self._stream())
        A.update(xL=x, rep=rep, meanf=meanf, stdf=stdf)
        if save:
            self._act = A
        return rep

    # ---------------------------------------------------------------------------------------- packed session tiles
    PACK_DENSITY_MAX = 0.45      # "auto": pack when at most this fraction of the [B,T] positions is real

    def _seq_in(self, seq):
        """input_seq of a public entry point -> int32 device tensor.  A batch that arrives from the host (the reference-style feed
        dict) shows "auto" packing its density: the fraction of real positions."""
        if isinstance(seq, torch.Tensor):
            self._density_now = None
        else:
            a = np.asarray(seq)
            self._density_now = float(np.count_nonzero(a)) / max(a.size, 1)
        return self._dev_i32(seq)

    def _use_pack(self, seq):
        if not (self.seq_fused and self.prune_last and self.H <= 150 and self.H % 2 == 0 and seq.shape[0] <= 4096):
            return False
        ps = self.pack_sessions
        if ps == "auto":
            d = getattr(self, "_density_now", None)
            if d is None:
                d = self.pack_density
            return d is not None and d <= self.PACK_DENSITY_MAX
        return bool(ps)

    def _pack_plan(self, seq, tag):
        """ader_seq_pack_plan for this batch: the tile layout of its real positions (device arrays; nothing comes back to the host --
        the launches that follow are sized by the bound max_tiles = B and read the true counts on the device)."""
        B, T = seq.shape[0], self.T
        i32 = torch.int32
        n = B * 64
        hdr = self.buf(tag + "pq_hdr", (8,), i32, zero=True)
        trows = self.buf(tag + "pq_trows", (B,), i32, zero=True)
        ids, lpos = self.buf(tag + "pq_ids", (n,), i32, zero=True), self.buf(tag + "pq_lpos", (n,), i32, zero=True)
        gpos, info = self.buf(tag + "pq_gpos", (n,), i32, zero=True), self.buf(tag + "pq_info", (n,), i32, zero=True)
        srow0, slen = self.buf(tag + "pq_srow0", (B,), i32, zero=True), self.buf(tag + "pq_slen", (B,), i32, zero=True)
        c = _lib.AderSeqPack()
        c.hdr, c.tile_rows, c.ids, c.lpos, c.gpos, c.info, c.srow0, c.slen = (ptr(hdr), ptr(trows), ptr(ids), ptr(lpos), ptr(gpos),
                                                                              ptr(info), ptr(srow0), ptr(slen))
        split = -1 if self.split_rows is None else int(self.split_rows)
        w1_min, w1_max, target = self.pack_window
        ref = ctypes.byref(c)
        plan_args = (B, T, int(self.row0), split, int(self.row0_ex), w1_min, w1_max, target, ref)
        call("ader_seq_pack_plan", ptr(seq), *plan_args, self._stream())
        d = self.pack_density if self.pack_density is not None else 0.15
        est = int(min(n, max(64, 1.25 * d * B * T + 64)))          # rows expected to exist: how the weight-gradient workgroups are shared out
        return dict(c=c, ref=ref, hdr=hdr, trows=trows, ids=ids, lpos=lpos, gpos=gpos, info=info, srow0=srow0, slen=slen,
                    B=B, rows=n, max_tiles=B, est=est, plan_args=plan_args)

    def unpack_rows(self, t, pack=None, pruned=False):
        """Tile-ordered activation [B*64, ...] of the last packed forward -> the session-indexed [B*T, ...] layout of the unpacked
        kernels, zeros at the padding positions (tests and diagnostics; a host synchronisation)."""
        pk = pack if pack is not None else self._act["pack"]
        if pruned:
            return t
        B, T = pk["B"], self.T
        nt = int(pk["hdr"][0].item())
        tr = pk["trows"][:nt].long()
        r = torch.arange(64, device=self.device)
        ok = (r[None, :] < tr[:, None]).reshape(-1)
        rows = torch.nonzero(ok).reshape(-1)
        lp = pk["lpos"][:nt * 64][ok].long()
        out = torch.zeros((B * T,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
        out[lp] = t[rows]
        return out

    def _forward_packed(self, seq, training, rate, step, save):
        """forward() on packed tiles (ader_seq_pack_plan + ader_seqp_fwd): the saved-activation dict has the keys of the unpacked
        path, the tensors of the K / V side and of unpruned blocks in tile order ([B*64, ..], see include/ader_hip.h)."""
        B, T, H, L = seq.shape[0], self.T, self.H, self.L
        tag = "pt" if save else "pe"
        ck = ("fwdp", tag, B, bool(training), float(rate), self.seed, self.row0, self.split_rows, self.row0_ex, self.pack_window,
              self.pack_density, self.prune_last)
        ent = self._dc(ck) if self.cache_descriptors else None
        if ent is not None:
            # same step shape as before: only the batch pointer and the dropout keys are new
            d, A, pk, drops, plan_args = ent
            self._rekey(drops, step)
            sp = ptr(seq)
            d.seq = sp
            d.d_emb = A["d_emb"].c
            for l in range(L):
                k, S = d.blk[l], A[l]
                k.d_attn, k.d_ffn1, k.d_ffn2 = S["da"].c, S["d1"].c, S["d2"].c
            A["seq"], A["step"] = seq, step
            st = self._stream()
            call("ader_seq_pack_plan", sp, *plan_args, st)
            call("ader_seqp_fwd", ctypes.byref(d), pk["ref"], pk["max_tiles"], st)
            if save:
                self._act = A
            return A["rep"]
        pk = self._pack_plan(seq, tag)
        rows = pk["rows"]
        A = {"B": B, "seq": seq, "rate": rate, "training": training, "step": step, "pack": pk}
        per_row = T * H
        pp = self._pp
        d = _lib.AderSeqFwd()
        d0 = self._drop(step, SITE_EMB, rate, training, per_row)
        A["d_emb"] = d0
        drops = [(d0, SITE_EMB)]
        x = self.buf(tag + "x0", (rows, H), zero=True)
        rep = self.buf(tag + "rep", (B, H))
        meanf, stdf = self.buf(tag + "mf", (B,)), self.buf(tag + "sf", (B,))
        d.seq, d.emb, d.pos, d.x0, d.status = ptr(seq), pp["emb"], pp["pos"], ptr(x), ptr(self.status)
        d.lnf_g, d.lnf_b, d.rep, d.meanf, d.stdf = pp["lnf_g"], pp["lnf_b"], ptr(rep), ptr(meanf), ptr(stdf)
        d.B, d.T, d.H, d.V, d.L = B, T, H, self.V, L
        d.sqrtH = float(np.sqrt(np.float32(H)))
        d.sqrt_dh = float(np.sqrt(np.float32(H // self.heads)))
        d.d_emb = d0.c
        for l in range(L):
            p = "b%d." % l
            n = lambda s: "%s%d%s" % (tag, l, s)   # noqa: E731
            pruned = self.prune_last and l == L - 1
            da = self._drop(step, site_attn(l), rate, training, self.heads * T * T)
            d1 = self._drop(step, site_ffn1(l), rate, training, per_row)
            d2 = self._drop(step, site_ffn2(l), rate, training, per_row)
            drops += [(da, site_attn(l)), (d1, site_ffn1(l)), (d2, site_ffn2(l))]
            M, sfx = (B, "L") if pruned else (rows, "")
            kmask = self.buf(n("km"), (rows,), zero=True)
            K, Vv = self.buf(n("K"), (rows, H), zero=True), self.buf(n("V"), (rows, H), zero=True)
            q_in = self.buf(n("qin" + sfx), (M, H), zero=True)
            mean1, std1, qmask = self.buf(n("m1" + sfx), (M,), zero=True), self.buf(n("s1" + sfx), (M,), zero=True), self.buf(n("qm" + sfx), (M,), zero=True)
            Q, x1, y = self.buf(n("Q" + sfx), (M, H), zero=True), self.buf(n("x1" + sfx), (M, H), zero=True), self.buf(n("y" + sfx), (M, H), zero=True)
            Pm = self.buf(n("P" + sfx), (B * T if pruned else rows * 64,), zero=True)
            mean2, std2 = self.buf(n("m2" + sfx), (M,), zero=True), self.buf(n("s2" + sfx), (M,), zero=True)
            h1d, x2 = self.buf(n("h1" + sfx), (M, H), zero=True), self.buf(n("x2" + sfx), (M, H), zero=True)
            k = d.blk[l]
            for i, w in enumerate(("wq", "wk", "wv", "w1", "w2")):
                k.w[i] = self.wbf.data_ptr() + self._widx[p + w] * self._wplane
            for i, bn in enumerate(("bq", "bk", "bv", "b1", "b2")):
                k.bias[i] = pp[p + bn]
            k.ln1_g, k.ln1_b, k.ln2_g, k.ln2_b = pp[p + "ln1_g"], pp[p + "ln1_b"], pp[p + "ln2_g"], pp[p + "ln2_b"]
            k.q_in, k.mean1, k.std1, k.kmask, k.qmask = ptr(q_in), ptr(mean1), ptr(std1), ptr(kmask), ptr(qmask)
            k.Q, k.K, k.V, k.P, k.x1, k.y = ptr(Q), ptr(K), ptr(Vv), ptr(Pm), ptr(x1), ptr(y)
            k.mean2, k.std2, k.h1d, k.x2 = ptr(mean2), ptr(std2), ptr(h1d), ptr(x2)
            k.d_attn, k.d_ffn1, k.d_ffn2 = da.c, d1.c, d2.c
            k.pruned = 1 if pruned else 0
            A[l] = dict(pruned=pruned, x=x, q_in=q_in, mean1=mean1, std1=std1, kmask=kmask, qmask=qmask, Q=Q, K=K, V=Vv, P=Pm,
                        x1=x1, y=y, mean2=mean2, std2=std2, h1d=h1d, da=da, d1=d1, d2=d2)
            x = x2
        call("ader_seqp_fwd", ctypes.byref(d), pk["ref"], pk["max_tiles"], self._stream())
        A.update(xL=x, rep=rep, meanf=meanf, stdf=stdf)
        if save:
            self._act = A
        if self.cache_descriptors:
            self._dc_put(ck, (d, A, pk, drops, pk["plan_args"]))
        return rep

    def _lnf_desc(self, B):
        """AderLnfBwd of the forward just saved (prune_last: xL / meanf / stdf are compact [B, ..]), or None when not fused."""
        A = self._act
        if not (self.fuse_final_ln and self.prune_last and self.lx3):
            self._lnf_done = None
            return None
        H = self.H
        dxl = self.buf("dx_L", (B, H))
        fslab = self.buf("lnf_slab_rows", (B * 2 * H,))
        c = _lib.AderLnfBwd()
        c.x, c.mean, c.std, c.gamma, c.dx, c.slab = ptr(A["xL"]), ptr(A["meanf"]), ptr(A["stdf"]), self._pp["lnf_g"], ptr(dxl), ptr(fslab)
        self._lnf_done = (dxl, fslab, B, c)
        return ctypes.byref(c)

    # ---------------------------------------------------------------------------------------- loss rows
    def _rowinfo(self, B, pos, n_train, ex_pos, ex_trow, N, Np, w_train, w_ex, teacher, tag="ri_"):
        Bp = (B + 63) // 64 * 64
        st = self._stream()
        lab = self.buf(tag + "lab", (Bp,), torch.int32)
        ncol = self.buf(tag + "ncol", (Bp,), torch.int32)
        wrow = self.buf(tag + "w", (Bp,))
        trow = self.buf(tag + "trow", (Bp,), torch.int32)
        tlse = self.buf(tag + "tlse", (Bp,))
        n_ex = B - n_train
        call("ader_build_rowinfo", ptr(pos), n_train, ptr(ex_pos), ptr(ex_trow), n_ex, N, Np, float(w_train), float(w_ex), Bp,
             ptr(lab), ptr(ncol), ptr(wrow), ptr(trow), st)
        if teacher is not None and n_ex > 0:
            self._teacher_lse(teacher, Np)
            tlse.zero_()
            tlse[n_train:n_train + n_ex] = self._tlse_all[trow[n_train:n_train + n_ex].long()]
            tptr, ldt = ptr(teacher), teacher.stride(0)
        else:
            tlse.zero_()
            tptr, ldt = None, 0
        return Bp, (ptr(lab), ptr(ncol), ptr(wrow), ptr(trow), ptr(tlse), tptr, ldt)

    def _teacher_lse(self, teacher, Np):
        """Natural log-sum-exp of every stored teacher row over its Np columns (self._tlse_all [E_all]).  The teacher logits of an
        exemplar are fixed for a whole period, so this runs once per teacher tensor and is gathered per step."""
        key = (teacher.data_ptr(), tuple(teacher.shape), teacher._version)
        if getattr(self, "_tlse_key", None) != key:
            E_all = teacher.shape[0]
            allrows = torch.arange(E_all, dtype=torch.int32, device=self.device)
            self._tlse_all = torch.empty(E_all, dtype=torch.float32, device=self.device)
            call("ader_row_lse", ptr(teacher), teacher.stride(0), Np, ptr(allrows), E_all, ptr(self._tlse_all), self._stream())
            self._tlse_key = key
        return self._tlse_all

    # ---------------------------------------------------------------------------------------- train step
    def loss_and_grad(self, seq, pos, max_item, *, ex_pos=None, teacher=None, ex_trow=None, lambda_=0.0, rate=0.0,
                      n_train_global=None, n_ex_global=None, _defer_table=False):
        """Forward + backward of one step (no optimiser).  seq [B,T] holds the train rows first and the exemplar rows
        after (main.py:229); pos [n_train]; exemplars are either distilled (teacher [*,Np] + ex_trow [n_ex] row indices,
        ADER.py:132-137) or one-hot (ex_pos [n_ex], ADER.py:126-131).  Leaves the loss in self.loss (device scalar) and
        the gradient of every parameter in self.grad.  (_defer_table, the fused-update form train_step uses: the loss scalar is
        summed beside the table update, so self.loss is final only after _fused_table_adam -- or the next call here.)"""
        self._refresh_stream()
        if self._pending_loss is not None:
            # a deferred step whose fused update never ran (an exception between the two calls, or loss_and_grad(_defer_table=True)
            # used on its own): its loss sum is still owed -- settle it before the row losses are overwritten
            call("ader_lbf_sum", ptr(self._pending_loss[0]), self._pending_loss[1], ptr(self.loss), self._stream())
            self._pending_loss = None
        seq = self._seq_in(seq)
        pos = self._dev_i32(pos)
        B, T, H, L = seq.shape[0], self.T, self.H, self.L
        cap = self.MAX_ROWS_FAST if self.lfast else self.MAX_ROWS
        _check(seq.dim() == 2 and seq.shape[1] == T, "input_seq must be [rows, maxlen = %d] (got %s)" % (T, tuple(seq.shape)))
        _check(B <= cap, "at most %d rows per step with logits_dtype=%s (got %d)" % (cap, self.logits_dtype, B))
        n_train = pos.shape[0]
        n_ex = B - n_train
        N = int(max_item)
        _check(1 <= N <= self.item_num, "max_item must be in [1, item_num = %d] (got %d)" % (self.item_num, N))
        _check(0 <= n_train <= B, "pos has %d rows but input_seq only %d" % (n_train, B))
        Np = 0
        if n_ex > 0:
            if teacher is not None:
                ex_trow = self._dev_i32(ex_trow if ex_trow is not None else np.arange(n_ex))
                Np = teacher.shape[1]
                _check(teacher.dtype == torch.float32 and teacher.stride(1) == 1 and Np <= N,
                       "exemplar_logits must be float32 [*, Np <= max_item], unit stride along the items")
            else:
                ex_pos = self._dev_i32(ex_pos)
                _check(ex_pos.shape[0] == n_ex, "exemplar_pos has %d rows, the batch %d exemplar rows" % (ex_pos.shape[0], n_ex))
        w_train = 1.0 / float(n_train_global if n_train_global is not None else max(n_train, 1))
        w_ex = (lambda_ / float(n_ex_global if n_ex_global is not None else n_ex)) if n_ex > 0 else 0.0
        step = self.global_step
        st = self._stream()
        rows = B * T
        # distilled steps with a bf16 shadow: the train rows take the bf16 flash path and the fused table update, the (few)
        # exemplar rows the exact-f32 kernels; their table gradient enters the fused update as a dense extra term
        split_kd = bool(self.lfast and teacher is not None and n_ex > 0 and n_train > 0 and _defer_table
                        and N >= self._grad_hi and self.dp_world == 1 and self.kd_split)
        # ... or (default, bf16 and x3 modes): ALL rows on the flash path -- the exemplar rows as their own 128-row chunks whose softmax runs
        # over the first Np items, with the teacher term as a second readout (forward) and a subtraction inside the fused update
        kd_rows_fit = ((n_train + 127) // 128 + (n_ex + 127) // 128) * 128 <= cap
        kd_fast = bool(split_kd and self.kd_fast and kd_rows_fit)
        # the same forward without the fused update (data-parallel ranks, or no optimiser step): the table gradient is written
        # out (ader_tab_grad_kd) and takes the dense exchange
        kd_fast_unfused = bool(not kd_fast and self.lfast and teacher is not None and n_ex > 0 and n_train > 0 and self.kd_fast
                               and kd_rows_fit and N >= self._grad_hi and (self.dp_world > 1 or not _defer_table))
        if kd_fast or kd_fast_unfused:
            split_kd = False
        use_bf16 = self.lfast and (teacher is None or split_kd or kd_fast)
        defer = bool(_defer_table and use_bf16 and N >= self._grad_hi)
        self._deferred = None
        # data-parallel shard with exemplar rows: its train rows and its exemplar rows sit at different global positions
        self.split_rows = n_train if (n_ex > 0 and getattr(self, "_ex_row0_set", False)) else None
        if use_bf16 and not (kd_fast or kd_fast_unfused):
            # the row descriptors of the flash logit kernels depend on the labels only: built BEFORE the forward stack (round 3: a
            # 5 us launch + a launch gap between k_seq_fwd and the logit forward, on the critical path of every step)
            Bb = n_train if split_kd else B            # rows of the bf16 / x3 path
            Bp = (Bb + 127) // 128 * 128
            lab, ncol = self.buf("ri_lab", (Bp,), torch.int32), self.buf("ri_ncol", (Bp,), torch.int32)
            wrow, trow = self.buf("ri_w", (Bp,)), self.buf("ri_trow", (Bp,), torch.int32)
            # (tried in round 4: the same launch on the side stream beside k_seq_fwd -- the cross-stream wait that then precedes the logit
            #  forward costs 12 us against 6 for the launch itself: profiles/r4x_timeline.txt)
            call("ader_build_rowinfo", ptr(pos), n_train, None if split_kd else ptr(ex_pos), None, 0 if split_kd else n_ex, N, 0,
                 float(w_train), float(w_ex), Bp, ptr(lab), ptr(ncol), ptr(wrow), ptr(trow), st)
        with self._sec("blocks_fwd"):
            rep = self.forward(seq, training=True, rate=rate, step=step, save=True)
        if kd_fast or kd_fast_unfused:
            return self._loss_and_grad_kd_fast(seq, pos, rep, n_train, n_ex, N, Np, teacher, ex_trow, w_train, w_ex, fused=kd_fast)
        if defer and self.dp_world == 1:
            # the id-bucketed lists of the fused table update need only the inputs: build them on a side stream, under the
            # logit kernels (the one-launch forward owns every CU's LDS; the logit kernels leave room for it)
            labs = pos if (n_ex == 0 or split_kd) else torch.cat([pos, ex_pos])
            self._lists_async(seq, labs, N)
        A = self._act
        emb = self._pp["emb"]
        demb = self.gradient("emb")
        if N < self._grad_hi:   # catalog shrank (never in the reference flow): clear stale rows
            demb[N + 1:self._grad_hi + 1].zero_()
        self._grad_hi = max(self._grad_hi, N)
        drep = self.buf("drep", (B, H))
        extra = None
        if use_bf16:
            R = call("ader_lbf_ranges", N, Bp)
            rep_bf = self.buf("lbf_rep", (Bp * 168,), torch.bfloat16)
            rep_lo = self.buf("lbf_rep_lo", (Bp * 168,), torch.bfloat16) if self.lx3 else None
            pm, pl = self.buf("lbf_pm", (R * Bp,)), self.buf("lbf_pl", (R * Bp,))
            pO = self.buf("lbf_pO", (R * Bp * 160,))
            lse, off, rowloss = self.buf("lg_lse", (Bp,)), self.buf("lbf_off", (Bp,)), self.buf("lg_rowloss", (Bp,))
            with self._sec("logits_fwd"):
                if self.lx3:
                    # the loss scalar feeds nothing in the backward pass: with the fused update deferred, its (single-workgroup) sum
                    # leaves the critical path and runs beside the table update
                    late_loss = bool(defer and self.dp_world == 1 and not split_kd and self.late_side_stream and self.seq_fused)
                    # (the operand images of the fused update are cut by the same launch as the operand planes)
                    img = (self.buf("lbf_rep_img", (call("ader_x3_rep_image_bytes", Bp),), torch.uint8, zero=True)
                           if (defer and self.x3_update == "tab16" and not split_kd) else None)
                    lnf = self._lnf_desc(B) if (Bb == B) else None      # (split_kd: the exemplar rows' dRep comes from other kernels)
                    call("ader_lx3_fwd_img_lnf", ptr(rep), emb, self.item_num, Bb, Bp, H, N, ptr(lab), ptr(wrow), ptr(rep_bf), ptr(rep_lo),
                         ptr(pm), ptr(pl), ptr(pO), ptr(lse), ptr(off), ptr(rowloss), None if late_loss else ptr(self.loss), ptr(drep),
                         ptr(img), lnf, st)
                    self._img_ready = img is not None
                    self._pending_loss = (rowloss, Bb) if late_loss else None
                else:
                    call("ader_lbf_fwd", ptr(rep), ptr(self.shadow), self.item_num, Bb, Bp, H, N, ptr(lab), ptr(wrow), ptr(rep_bf),
                         ptr(pm), ptr(pl), ptr(pO), ptr(lse), ptr(off), ptr(rowloss), ptr(self.loss), ptr(drep), st)
            if split_kd:
                rep_x, drep_x = rep[n_train:], drep[n_train:]
                Bpx, rix = self._rowinfo(n_ex, None, 0, None, ex_trow, N, Np, w_train, w_ex, teacher, tag="kd_")
                parts = call("ader_logits_parts", N)
                part = self.buf("lg_part", (parts * Bpx * 3,))
                lse_x, rowloss_x = self.buf("kd_lse", (Bpx,)), self.buf("kd_rowloss", (Bpx,))
                loss_x = self.buf("kd_loss", (1,))
                with self._sec("kd_rows"):
                    call("ader_logits_loss_fwd", ptr(rep_x), emb, n_ex, Bpx, H, N, *rix, ptr(part), ptr(lse_x), ptr(rowloss_x),
                         ptr(loss_x), st)
                    ranges = call("ader_logits_ranges", N, Bpx)
                    slab = self.buf("lg_slab", (ranges * Bpx * 160,))
                    call("ader_logits_bwd_drep", ptr(rep_x), emb, n_ex, Bpx, H, N, *rix, ptr(lse_x), ptr(slab), ptr(drep_x), st)
                    call("ader_logits_bwd_demb", ptr(rep_x), emb, n_ex, Bpx, H, N, *rix, ptr(lse_x), ptr(demb), st)
                    self.loss.add_(loss_x)
                extra = demb
            if not defer:
                with self._sec("logits_bwd_demb"):
                    call("ader_tab_grad", ptr(rep_bf), ptr(rep_lo), emb, self.item_num, B, Bp, H, N, ptr(lab), ptr(wrow),
                         ptr(off), ptr(demb), st)
        else:
            Bp, ri = self._rowinfo(B, pos, n_train, ex_pos if teacher is None else None, ex_trow if teacher is not None else None,
                                   N, Np, w_train, w_ex, teacher)
            parts = call("ader_logits_parts", N)
            part = self.buf("lg_part", (parts * Bp * 3,))
            lse, rowloss = self.buf("lg_lse", (Bp,)), self.buf("lg_rowloss", (Bp,))
            with self._sec("logits_fwd"):
                call("ader_logits_loss_fwd", ptr(rep), emb, B, Bp, H, N, *ri, ptr(part), ptr(lse), ptr(rowloss), ptr(self.loss), st)
            ranges = call("ader_logits_ranges", N, Bp)
            slab = self.buf("lg_slab", (ranges * Bp * 160,))
            with self._sec("logits_bwd_drep"):
                call("ader_logits_bwd_drep", ptr(rep), emb, B, Bp, H, N, *ri, ptr(lse), ptr(slab), ptr(drep), st)
            with self._sec("logits_bwd_demb"):
                call("ader_logits_bwd_demb", ptr(rep), emb, B, Bp, H, N, *ri, ptr(lse), ptr(demb), st)
        self._early = None
        if not defer and self.grad_early_hook is not None:
            self._early = self.grad_early_hook(self, N)       # async all-reduce of demb: overlaps the blocks backward below
        dx = self._blocks_backward(seq, drep, defer, demb)
        if self._early is not None:
            self._dp_rows = (seq, dx)                         # per-position input-gradient rows: exchanged and scattered in the hook
        if defer:
            self._deferred = dict(seq=seq, g=dx, B=(n_train if split_kd else B), Bp=Bp, N=N, rep_bf=rep_bf, rep_lo=rep_lo, off=off,
                                  lab=lab, wrow=wrow, extra=extra)
        return self.loss

    def _loss_and_grad_kd_fast(self, seq, pos, rep, n_train, n_ex, N, Np, teacher, ex_trow, w_train, w_ex, fused=True):
        """Distilled step (ADER.py:108-137) entirely on the bf16 flash kernels.  Rows are laid out [train rows padded to 128 |
        exemplar rows padded to 128]; ader_lbf_fwd_kd gives the student log-sum-exp of every row (exemplar rows: over the first Np
        items), the softmax-weighted readout O1 and, for exemplar rows, the teacher readout O2 = sum_j softmax(t)_j E_j, from which
        loss = w (lse - rep.O2) and dRep = w (O1/l - O2); the table gradient w (softmax(s) - softmax(t))^T rep is formed inside the
        fused update (ader_tab_update_sh_kd), which reads the teacher tile a second time.  Nothing [rows, N]-sized is materialised."""
        st = self._stream()
        H = self.H
        B = n_train + n_ex
        Bt, Bk = (n_train + 127) // 128 * 128, (n_ex + 127) // 128 * 128
        Bp = Bt + Bk
        tlse_all = self._teacher_lse(teacher, Np)
        lab, trow = self.buf("kf_lab", (Bp,), torch.int32), self.buf("kf_trow", (Bp,), torch.int32)
        wrow, tlse2 = self.buf("kf_w", (Bp,)), self.buf("kf_tlse2", (Bp,))
        if self.lx3:      # the x3 teacher readout is a launch of its own, with its own item ranges
            R, R2 = call("ader_lbf_ranges", N, Bp), call("ader_lx3_readout_ranges", Np, Bk)
        else:
            R, R2 = call("ader_lbf_ranges_kd", N, Bp, Bt), call("ader_lbf_readout_ranges", N, Bp, Bt)
        rep_bf = self.buf("lbf_rep", (Bp * 168,), torch.bfloat16)
        rep_lo = self.buf("lbf_rep_lo", (Bp * 168,), torch.bfloat16) if self.lx3 else None
        pm, pl = self.buf("lbf_pm", (R * Bp,)), self.buf("lbf_pl", (R * Bp,))
        pO, pO2 = self.buf("lbf_pO", (R * Bp * 160,)), self.buf("lbf_pO2", (R2 * Bk * 160,))
        lse, off, rowloss = self.buf("lg_lse", (Bp,)), self.buf("lbf_off", (Bp,)), self.buf("lg_rowloss", (Bp,))
        drep = self.buf("drep", (B, H))
        with self._sec("logits_fwd"):
            if self.lx3:
                # (as in the vanilla step: the loss scalar feeds nothing in the backward pass -- summed beside the table update)
                late_loss = bool(fused and self.dp_world == 1 and self.late_side_stream and self.seq_fused)
                # (... and the operand images of the fused update are cut by the launch that cuts the operand planes)
                img = (self.buf("lbf_rep_img", (call("ader_x3_rep_image_bytes", Bp),), torch.uint8, zero=True)
                       if (fused and self.x3_update == "tab16") else None)
                call("ader_lx3_fwd_kd_lnf", ptr(rep), self._pp["emb"], self.item_num, n_train, n_ex, Bt, Bp, H, N, Np, ptr(pos),
                     ptr(ex_trow), ptr(teacher), teacher.stride(0), ptr(tlse_all), float(w_train), float(w_ex), ptr(lab), ptr(wrow),
                     ptr(trow), ptr(tlse2), ptr(rep_bf), ptr(rep_lo), ptr(pm), ptr(pl), ptr(pO), ptr(pO2), ptr(lse), ptr(off),
                     ptr(rowloss), None if late_loss else ptr(self.loss), ptr(drep), ptr(img), self._lnf_desc(B), st)
                self._img_ready = img is not None
                self._pending_loss = (rowloss, Bp) if late_loss else None
            else:
                call("ader_lbf_fwd_kd", ptr(rep), ptr(self.shadow), self.item_num, n_train, n_ex, Bt, Bp, H, N, Np, ptr(pos),
                     ptr(ex_trow), ptr(teacher), teacher.stride(0), ptr(tlse_all), float(w_train), float(w_ex), ptr(lab), ptr(wrow),
                     ptr(trow), ptr(tlse2), ptr(rep_bf), ptr(pm), ptr(pl), ptr(pO), ptr(pO2), ptr(lse), ptr(off), ptr(rowloss),
                     ptr(self.loss), ptr(drep), st)
        self._grad_hi = max(self._grad_hi, N)
        if not fused:
            demb = self.gradient("emb")
            with self._sec("logits_bwd_demb"):
                call("ader_tab_grad_kd", ptr(rep_bf), ptr(rep_lo), self._pp["emb"], self.item_num, Bp, Bt, H, N, Np, ptr(lab), ptr(wrow),
                     ptr(off), ptr(teacher), teacher.stride(0), ptr(trow), ptr(tlse2), ptr(demb), st)
            self._early = None
            if self.grad_early_hook is not None:
                self._early = self.grad_early_hook(self, N)
            dx = self._blocks_backward(seq, drep, False, demb)
            if self._early is not None:
                self._dp_rows = (seq, dx)
            return self.loss
        self._lists_async(seq, lab, N)            # one-hot targets in the padded row numbering (label 0 = none)
        dx = self._blocks_backward(seq, drep, True, None)
        self._deferred = dict(seq=seq, g=dx, B=Bp, Bp=Bp, N=N, rep_bf=rep_bf, rep_lo=rep_lo, off=off, lab=lab, wrow=wrow, extra=None,
                              kd=dict(row0=Bt, Np=Np, teacher=teacher, trow=trow, tlse2=tlse2))
        return self.loss

    def _blocks_backward(self, seq, drep, defer, demb):
        """Backward of the final LayerNorm, the blocks and the prologue from drep [B,H] (gradient of the loss w.r.t. the
        representation).  Fills the gradients of every non-table parameter; the table's sparse term goes into demb (dense
        path) or, with `defer`, stays as per-position rows in the returned dx [B*T,H] for the fused table update."""
        A = self._act
        B, T, H, L = A["B"], self.T, self.H, self.L
        rows = B * T
        st = self._stream()
        tb = self._sec("blocks_bwd")
        tb.__enter__()
        wslab = self.buf("w_slab", (max(call("ader_gemm_atb_slabs", rows) * 160 * 160, call("ader_ln_bwd_slabs", rows) * 2 * H),))
        pp, gp = self._pp, self._gp
        xL = A["xL"]
        pk = A.get("pack")
        if pk is not None:
            # packed tiles: block-to-block gradients in tile order; the rows of the input embeddings leave by position (dx_emb)
            dx = self.buf("pdx_a", (pk["rows"], H), zero=True)
            dxn = self.buf("pdx_b", (pk["rows"], H), zero=True)
            dx_emb = self.buf("dx_emb", (rows, H), zero=True)
        else:
            dx = self.buf("dx_a", (rows, H), zero=True)
            dxn = self.buf("dx_b", (rows, H), zero=True)
        self._late_on = bool(defer and (self.dp_world == 1 or self._late_force) and self.seq_fused and self.late_side_stream)
        lnf_done, self._lnf_done = self._lnf_done, None
        if self.prune_last and lnf_done is not None and lnf_done[2] == B:
            # the merge launch of the logit forward has already written dx of the final LayerNorm and the per-row gamma / beta partials
            dxl, fslab = lnf_done[0], lnf_done[1]
            self._late_call("ader_reduce_slabs", ptr(fslab), 2 * H, B, H, 1, H, gp["lnf_g"], gp["lnf_b"])
        elif self.prune_last:
            dxl = self.buf("dx_L", (B, H))        # gradient of the final block's output row T-1 (compact)
            if self._late_on:       # gamma / beta partials reduced later, beside the table update (their own slab buffer)
                G = call("ader_ln_bwd_slabs", B)
                fslab = self.buf("lnf_slab", (G * 2 * H,))
                call("ader_ln_bwd", ptr(drep), H, ptr(xL), H, pp["lnf_g"], ptr(A["meanf"]), ptr(A["stdf"]), None, 0, ptr(dxl), H,
                     ptr(fslab), None, None, B, H, st)
                self._late_call("ader_reduce_slabs", ptr(fslab), 2 * H, G, H, 1, H, gp["lnf_g"], gp["lnf_b"])
            else:
                call("ader_ln_bwd", ptr(drep), H, ptr(xL), H, pp["lnf_g"], ptr(A["meanf"]), ptr(A["stdf"]), None, 0, ptr(dxl), H,
                     ptr(wslab), gp["lnf_g"], gp["lnf_b"], B, H, st)
        else:
            dx.zero_()
            call("ader_ln_bwd", ptr(drep), H, ptr(xL.view(B, T, H)[:, T - 1, :]), T * H, pp["lnf_g"], ptr(A["meanf"]),
                 ptr(A["stdf"]), None, 0, ptr(dx.view(B, T, H)[:, T - 1, :]), T * H, ptr(wslab), gp["lnf_g"], gp["lnf_b"], B, H, st)
        last_map = (T, T - 1)
        fused_emb = False
        for l in reversed(range(L)):
            p = "b%d." % l
            S = A[l]
            W = lambda s: pp[p + s]      # noqa: E731
            G = lambda s: gp[p + s]      # noqa: E731
            if S["pruned"]:
                M, rmap, dxo = B, last_map, dxl
            else:
                M, rmap, dxo = rows, (1, 0), dx
            if pk is not None:
                emb_bwd = l == 0
                self._bwd_block_packed(l, S, pk, dxo, dx_emb if emb_bwd else dxn, B, emb_bwd, A["d_emb"])
                fused_emb = fused_emb or emb_bwd
                dx, dxn = (dx_emb, dx) if emb_bwd else (dxn, dx)
                continue
            if self.seq_fused:
                emb_bwd = l == 0        # (block 0's chain applies the prologue mask / dropout to the rows it writes)
                self._bwd_block_fused(l, S, seq, dxo, dxn, M, B, emb_bwd, A["d_emb"])
                fused_emb = fused_emb or emb_bwd
                dx, dxn = dxn, dx
                continue
            tg = "L" if S["pruned"] else ""
            g = self.buf("bw_g" + tg, (M, H))
            dh2 = self.buf("bw_dh2%d" % l, (M, H))       # the weight-gradient operands stay alive until _atb_flush
            da_ = self.buf("bw_da%d" % l, (M, H))
            dy = self.buf("bw_dy" + tg, (M, H))
            dx1 = self.buf("bw_dx1" + tg, (M, H))
            dQ = self.buf("bw_dQ%d" % l, (M, H))
            dqin = self.buf("bw_dqin" + tg, (M, H))
            dK, dV = self.buf("bw_dK%d" % l, (rows, H)), self.buf("bw_dV%d" % l, (rows, H))
            call("ader_mask_dropgrad", ptr(dxo), ptr(seq), ptr(g), ptr(dh2), M, H, rmap[0], rmap[1], *S["d2"].args(), st)
            self._gemm(dh2, p + "w2", None, da_, S["h1d"], None, M, EPI_RELUDROPGRAD, trans=1, drop=S["d1"])
            self._gemm(da_, p + "w1", None, dy, g, None, M, EPI_ADD, trans=1)
            self._atb(S["h1d"], dh2, p + "w2", p + "b2", wslab, M)
            self._atb(S["y"], da_, p + "w1", p + "b1", wslab, M)
            call("ader_ln_bwd", ptr(dy), H, ptr(S["x1"]), H, W("ln2_g"), ptr(S["mean2"]), ptr(S["std2"]), None, 0, ptr(dx1), H,
                 ptr(wslab), G("ln2_g"), G("ln2_b"), M, H, st)
            if S["pruned"]:
                call("ader_attn_last_bwd", ptr(dx1), ptr(S["Q"]), ptr(S["K"]), ptr(S["V"]), ptr(S["P"]), ptr(S["kmask"]),
                     ptr(S["qmask"]), ptr(dQ), ptr(dK), ptr(dV), B, T, H, self.heads, *S["da"].args(), st)
            else:
                call("ader_attn_x3_bwd" if self.attn_x3 else "ader_attn_bwd", ptr(dx1), ptr(S["Q"]), ptr(S["K"]), ptr(S["V"]),
                     ptr(S["P"]), ptr(S["kmask"]), ptr(S["qmask"]), ptr(dQ), ptr(dK), ptr(dV), B, T, H, self.heads,
                     *S["da"].args(), st)
            self._gemm(dQ, p + "wq", None, dqin, dx1, None, M, EPI_ADD, trans=1)
            if S["pruned"]:
                # LN1 backward on row T-1 only; dK/dV reach every row through the K/V projections
                dql = self.buf("bw_dxq", (B, H))
                call("ader_ln_bwd", ptr(dqin), H, ptr(S["x"].view(B, T, H)[:, T - 1, :]), T * H, W("ln1_g"), ptr(S["mean1"]),
                     ptr(S["std1"]), None, 0, ptr(dql), H, ptr(wslab), G("ln1_g"), G("ln1_b"), B, H, st)
                self._gemm(dK, p + "wk", None, dxn, None, None, rows, EPI_BIAS, trans=1)
                self._gemm(dV, p + "wv", None, dxn, dxn, None, rows, EPI_ADD, trans=1)
                call("ader_add_rows", ptr(dql), ptr(dxn), B, H, T, T - 1, st)
            else:
                call("ader_ln_bwd", ptr(dqin), H, ptr(S["x"]), H, W("ln1_g"), ptr(S["mean1"]), ptr(S["std1"]), None, 0, ptr(dxn), H,
                     ptr(wslab), G("ln1_g"), G("ln1_b"), rows, H, st)
                self._gemm(dK, p + "wk", None, dxn, dxn, None, rows, EPI_ADD, trans=1)
                self._gemm(dV, p + "wv", None, dxn, dxn, None, rows, EPI_ADD, trans=1)
            self._atb(S["q_in"], dQ, p + "wq", p + "bq", wslab, M)
            self._atb(S["x"], dK, p + "wk", p + "bk", wslab, rows)
            self._atb(S["x"], dV, p + "wv", p + "bv", wslab, rows)
            dx, dxn = dxn, dx
        if not self._late_on:
            self._atb_flush()
        self._last_g = dx       # per-position gradient rows of the input embeddings (tests: column-sum checks)
        if pk is not None:
            # (the packed chain wrote the REAL positions of dx only: the positional gradient sums those; every other consumer
            #  addresses dx through the id lists, which leave the padding out)
            self._late_call("ader_pos_grad_packed", ptr(dx), ptr(pk["slen"]), gp["pos"], B, T, H)
            if not defer and self._early is None:
                lab0 = self.buf("dp_lab0", (1,), torch.int32, zero=True)
                ids_s, order, sp_start, _, _, _, _ = self._sparse_lists(seq, lab0, self.item_num)
                call("ader_scatter_rows_ordered", ptr(ids_s), ptr(order), ptr(sp_start), sp_start.numel() - 1, ptr(dx), H, self.V,
                     float(np.sqrt(np.float32(H))), ptr(demb), st)
        elif defer:
            # (block 0's ader_seq_bwd_qkv has already applied the prologue mask / dropout to the rows: seq = NULL)
            if fused_emb:
                self._late_call("ader_embed_bwd_rows", None, ptr(dx), gp["pos"], B, T, H, self.V, *A["d_emb"].args())
            else:
                call("ader_embed_bwd_rows", ptr(seq), ptr(dx), gp["pos"], B, T, H, self.V, *A["d_emb"].args(), st)
        elif self._early is not None:
            # the table gradient is being all-reduced: leave the masked rows in dx (scattered for all ranks after the reduction)
            call("ader_embed_bwd_rows", None if fused_emb else ptr(seq), ptr(dx), gp["pos"], B, T, H, self.V, *A["d_emb"].args(), st)
        else:
            # unfused single-process path (exact-f32 logits, EWC, loss_and_grad without the optimiser): mask / dropout on the rows,
            # then the rows are added into the table gradient bucket by bucket in position order -- no float atomics, so this path
            # is bitwise reproducible too (the reference sets TF_DETERMINISTIC_OPS, main.py:121-122)
            call("ader_embed_bwd_rows", None if fused_emb else ptr(seq), ptr(dx), gp["pos"], B, T, H, self.V, *A["d_emb"].args(), st)
            lab0 = self.buf("dp_lab0", (1,), torch.int32, zero=True)
            ids_s, order, sp_start, _, _, _, _ = self._sparse_lists(seq, lab0, self.item_num)
            call("ader_scatter_rows_ordered", ptr(ids_s), ptr(order), ptr(sp_start), sp_start.numel() - 1, ptr(dx), H, self.V,
                 float(np.sqrt(np.float32(H))), ptr(demb), st)
        tb.__exit__(None, None, None)
        self._late_on = False
        return dx

    def _bwd_block_fused(self, l, S, seq, dxo, dxn, M, B, emb_bwd, d_emb):
        """Backward of block l with the session-tiled chains (seq_bwd.hip) around the attention backward; queues the five
        weight-gradient products.  dxo: gradient of the block output ([B*T,H], or [B,H] for the pruned last block);
        dxn [B*T,H] receives the gradient of the block input."""
        T, H = self.T, self.H
        rows = B * T
        st = self._stream()
        p = "b%d." % l
        pp, gp = self._pp, self._gp
        pruned = 1 if S["pruned"] else 0
        wp = lambda w: self.wbf.data_ptr() + self._widx[p + w] * self._wplane     # noqa: E731
        dh2, da_ = self.buf("bw_dh2%d" % l, (M, H), zero=True), self.buf("bw_da%d" % l, (M, H), zero=True)
        dx1, dQ = self.buf("bw_dx1%d" % l, (M, H), zero=True), self.buf("bw_dQ%d" % l, (M, H), zero=True)
        dK, dV = self.buf("bw_dK%d" % l, (rows, H), zero=True), self.buf("bw_dV%d" % l, (rows, H), zero=True)
        slab2, slab1 = self.buf("ln_slab%d_2" % l, (B * 2 * H,)), self.buf("ln_slab%d_1" % l, (B * 2 * H,))
        f = _lib.AderSeqBwdFfn()
        f.seq, f.dx2, f.h1d, f.x1, f.mean2, f.std2 = ptr(seq), ptr(dxo), ptr(S["h1d"]), ptr(S["x1"]), ptr(S["mean2"]), ptr(S["std2"])
        f.ln2_g, f.w2, f.w1 = pp[p + "ln2_g"], wp("w2"), wp("w1")
        f.dh2, f.da, f.dx1, f.slab = ptr(dh2), ptr(da_), ptr(dx1), ptr(slab2)
        f.d_ffn1, f.d_ffn2 = S["d1"].c, S["d2"].c
        f.B, f.T, f.H, f.pruned = B, T, H, pruned
        call("ader_seq_bwd_ffn", ctypes.byref(f), st)
        self._late_call("ader_reduce_slabs", ptr(slab2), 2 * H, B, H, 1, H, gp[p + "ln2_g"], gp[p + "ln2_b"])
        wslab = self._ws["w_slab"]
        self._atb(S["h1d"], dh2, p + "w2", p + "b2", wslab, M)
        self._atb(S["y"], da_, p + "w1", p + "b1", wslab, M)
        if pruned:
            call("ader_attn_last_bwd", ptr(dx1), ptr(S["Q"]), ptr(S["K"]), ptr(S["V"]), ptr(S["P"]), ptr(S["kmask"]),
                 ptr(S["qmask"]), ptr(dQ), ptr(dK), ptr(dV), B, T, H, self.heads, *S["da"].args(), st)
        else:
            call("ader_attn_x3_bwd" if self.attn_x3 else "ader_attn_bwd", ptr(dx1), ptr(S["Q"]), ptr(S["K"]), ptr(S["V"]),
                 ptr(S["P"]), ptr(S["kmask"]), ptr(S["qmask"]), ptr(dQ), ptr(dK), ptr(dV), B, T, H, self.heads,
                 *S["da"].args(), st)
        q = _lib.AderSeqBwdQkv()
        q.seq, q.dQ, q.dx1, q.dK, q.dV, q.x = ptr(seq), ptr(dQ), ptr(dx1), ptr(dK), ptr(dV), ptr(S["x"])
        q.mean1, q.std1, q.ln1_g = ptr(S["mean1"]), ptr(S["std1"]), pp[p + "ln1_g"]
        q.wq, q.wk, q.wv = wp("wq"), wp("wk"), wp("wv")
        q.dx, q.slab = ptr(dxn), ptr(slab1)
        q.d_emb = d_emb.c
        q.B, q.T, q.H, q.pruned, q.emb_bwd = B, T, H, pruned, 1 if emb_bwd else 0
        call("ader_seq_bwd_qkv", ctypes.byref(q), st)
        self._late_call("ader_reduce_slabs", ptr(slab1), 2 * H, B, H, 1, H, gp[p + "ln1_g"], gp[p + "ln1_b"])
        self._atb(S["q_in"], dQ, p + "wq", p + "bq", wslab, M)
        self._atb(S["x"], dK, p + "wk", p + "bk", wslab, rows)
        self._atb(S["x"], dV, p + "wv", p + "bv", wslab, rows)

    def _bwd_block_packed(self, l, S, pk, dxo, dxn, B, emb_bwd, d_emb):
        """_bwd_block_fused on packed tiles (seqp_bwd.hip): dxo = gradient of the block output (tile order, or compact [B,H] for the
        pruned last block); dxn receives the gradient of the block input (tile order; block 0: the session-indexed [B*T,H] rows)."""
        T, H = self.T, self.H
        rows, mt = pk["rows"], pk["max_tiles"]
        st = self._stream()
        p = "b%d." % l
        pp, gp = self._pp, self._gp
        pruned = 1 if S["pruned"] else 0
        M = B if pruned else rows
        mpk = None if pruned else pk           # the compact tensors of a pruned block are plain [B, H]
        # (cached only on the default training path: small launches queued for the side stream, weight gradients batched)
        cacheable = bool(self.cache_descriptors and self._late_on and self.gemm_x3 and self.atb_batch)
        ck = ("bwdp", l, B, ptr(dxo), ptr(dxn), bool(emb_bwd))
        ent = self._dc(ck) if cacheable else None
        if ent is not None and ent[0] is S and ent[1] is pk and ent[2] is d_emb:
            # same saved-activation dict (a cached forward's): the descriptors stand, the dropout keys are this step's
            _, _, _, f, q, attn_name, attn_args, late2, late1, atbs2, atbs1 = ent
            f.d_ffn1, f.d_ffn2 = S["d1"].c, S["d2"].c
            q.d_emb = d_emb.c
            call("ader_seqp_bwd_ffn", ctypes.byref(f), pk["ref"], mt, st)
            self._late.append(late2)
            for it in atbs2:
                self._atb_q.append(it)
                if len(self._atb_q) == 16:
                    self._atb_flush()
            call(attn_name, *attn_args, st)
            call("ader_seqp_bwd_qkv", ctypes.byref(q), pk["ref"], mt, st)
            self._late.append(late1)
            for it in atbs1:
                self._atb_q.append(it)
                if len(self._atb_q) == 16:
                    self._atb_flush()
            return
        wp = lambda w: self.wbf.data_ptr() + self._widx[p + w] * self._wplane     # noqa: E731
        sfx = "L" if pruned else ""
        dh2, da_ = self.buf("pbw_dh2%d%s" % (l, sfx), (M, H), zero=True), self.buf("pbw_da%d%s" % (l, sfx), (M, H), zero=True)
        dx1, dQ = self.buf("pbw_dx1%d%s" % (l, sfx), (M, H), zero=True), self.buf("pbw_dQ%d%s" % (l, sfx), (M, H), zero=True)
        dK, dV = self.buf("pbw_dK%d" % l, (rows, H), zero=True), self.buf("pbw_dV%d" % l, (rows, H), zero=True)
        slab2, slab1 = self.buf("pln_slab%d_2" % l, (mt * 2 * H,)), self.buf("pln_slab%d_1" % l, (mt * 2 * H,))
        f = _lib.AderSeqBwdFfn()
        f.seq, f.dx2, f.h1d, f.x1, f.mean2, f.std2 = None, ptr(dxo), ptr(S["h1d"]), ptr(S["x1"]), ptr(S["mean2"]), ptr(S["std2"])
        f.ln2_g, f.w2, f.w1 = pp[p + "ln2_g"], wp("w2"), wp("w1")
        f.dh2, f.da, f.dx1, f.slab = ptr(dh2), ptr(da_), ptr(dx1), ptr(slab2)
        f.d_ffn1, f.d_ffn2 = S["d1"].c, S["d2"].c
        f.B, f.T, f.H, f.pruned = B, T, H, pruned
        call("ader_seqp_bwd_ffn", ctypes.byref(f), pk["ref"], mt, st)
        self._late_call("ader_reduce_slabs", ptr(slab2), 2 * H, mt, H, 1, H, gp[p + "ln2_g"], gp[p + "ln2_b"])
        wslab = self._ws["w_slab"]
        self._atb(S["h1d"], dh2, p + "w2", p + "b2", wslab, M, mpk)
        self._atb(S["y"], da_, p + "w1", p + "b1", wslab, M, mpk)
        attn_args = (ptr(dx1), ptr(S["Q"]), ptr(S["K"]), ptr(S["V"]), ptr(S["P"]), ptr(S["kmask"]), ptr(S["qmask"]), ptr(dQ), ptr(dK),
                     ptr(dV), B, T, H, *S["da"].args(), pk["ref"]) + (() if pruned else (mt,))
        attn_name = "ader_attnp_last_bwd" if pruned else "ader_attnp_bwd"
        call(attn_name, *attn_args, st)
        q = _lib.AderSeqBwdQkv()
        q.seq, q.dQ, q.dx1, q.dK, q.dV, q.x = None, ptr(dQ), ptr(dx1), ptr(dK), ptr(dV), ptr(S["x"])
        q.mean1, q.std1, q.ln1_g = ptr(S["mean1"]), ptr(S["std1"]), pp[p + "ln1_g"]
        q.wq, q.wk, q.wv = wp("wq"), wp("wk"), wp("wv")
        q.dx, q.slab = ptr(dxn), ptr(slab1)
        q.d_emb = d_emb.c
        q.B, q.T, q.H, q.pruned, q.emb_bwd = B, T, H, pruned, 1 if emb_bwd else 0
        call("ader_seqp_bwd_qkv", ctypes.byref(q), pk["ref"], mt, st)
        self._late_call("ader_reduce_slabs", ptr(slab1), 2 * H, mt, H, 1, H, gp[p + "ln1_g"], gp[p + "ln1_b"])
        self._atb(S["q_in"], dQ, p + "wq", p + "bq", wslab, M, mpk)
        self._atb(S["x"], dK, p + "wk", p + "bk", wslab, rows, pk)
        self._atb(S["x"], dV, p + "wv", p + "bv", wslab, rows, pk)
        if cacheable:
            late2 = ("ader_reduce_slabs", (ptr(slab2), 2 * H, mt, H, 1, H, gp[p + "ln2_g"], gp[p + "ln2_b"]))
            late1 = ("ader_reduce_slabs", (ptr(slab1), 2 * H, mt, H, 1, H, gp[p + "ln1_g"], gp[p + "ln1_b"]))
            atbs2 = [(S["h1d"], dh2, gp[p + "w2"], gp[p + "b2"], M, mpk), (S["y"], da_, gp[p + "w1"], gp[p + "b1"], M, mpk)]
            atbs1 = [(S["q_in"], dQ, gp[p + "wq"], gp[p + "bq"], M, mpk), (S["x"], dK, gp[p + "wk"], gp[p + "bk"], rows, pk),
                     (S["x"], dV, gp[p + "wv"], gp[p + "bv"], rows, pk)]
            self._dc_put(ck, (S, pk, d_emb, f, q, attn_name, attn_args, late2, late1, atbs2, atbs1))

    def _lr_t(self, lr):
        return float(np.float32(lr) * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p))

    def _advance_adam(self):
        self.refresh_weights()
        self.b1p = np.float32(self.b1p * np.float32(self.beta1))
        self.b2p = np.float32(self.b2p * np.float32(self.beta2))
        self.global_step += 1

    def adam(self, lr):
        """tf.train.AdamOptimizer step on every variable (dense, incl. the whole table; ADER.py:96)."""
        self._refresh_stream()
        self._gather_if_sharded()
        with self._sec("adam"):
            call("ader_adam_step", ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), ptr(self.grad), self.P, self._lr_t(lr),
                 self.beta1, self.beta2, self.eps, ptr(self.shadow), self.V * self.H, self.H, self._stream())
        self._advance_adam()

    def _sparse_lists(self, seq, lab, N):
        """Bucketed lists of the sparse table-gradient terms (input positions, one-hot targets) for the fused table update:
        (ids, positions, bucket starts) x 2 -- buckets of 64 ids in id order, inside a bucket in position order, so the
        contributions to a table row are added in position order (deterministic) -- and, in x3 mode, the per-tile records of the
        64-row kernel.  Built by ader_sparse_lists (csrc/index_prep.hip) into persistent buffers."""
        seq, lab = seq.reshape(-1), lab.reshape(-1)
        if seq.dtype != torch.int32:
            seq = seq.to(torch.int32)
        if lab.dtype != torch.int32:
            lab = lab.to(torch.int32)
        n_sp, n_tg = seq.numel(), lab.numel()
        nb1 = call("ader_sparse_lists_starts", N)
        i32 = torch.int32
        ids, order = self.buf("sl_ids", (n_sp,), i32), self.buf("sl_rows", (n_sp,), i32)
        tids, torder = self.buf("sl_tids", (n_tg,), i32), self.buf("sl_trows", (n_tg,), i32)
        sp_start, tg_start = self.buf("sl_sps", (nb1,), i32), self.buf("sl_tgs", (nb1,), i32)
        scratch = self.buf("sl_scratch", (call("ader_sparse_lists_scratch_n", n_sp, n_tg, N),), i32)
        meta = None
        if self.lx3 or self.bf16_update == "resident":        # per-tile list records of the 64-row update kernel (table_update.hip)
            meta = self.buf("sl_meta", (call("ader_tab_meta_ints", N),), torch.int32)
        # (lists + records: ONE launch on the catalogs of the shipped datasets, the chain of launches on large ones)
        call("ader_sparse_lists_meta", ptr(seq.contiguous()), n_sp, ptr(lab.contiguous()), n_tg, N, ptr(scratch), ptr(ids), ptr(order),
             ptr(sp_start), ptr(tids), ptr(torder), ptr(tg_start), ptr(meta), self._stream())
        return ids, order, sp_start, tids, torder, tg_start, meta

    def _lists_async(self, seq, lab, N):
        main = self._main
        # flat, contiguous int32 copies are made HERE, by torch on the main stream and before the side stream is told to wait for it:
        # a strided view (the catalog-sharded step passes ids_g[:, :n_pos]) would otherwise be materialised inside _sparse_lists --
        # a torch kernel on the main stream that the list kernels on the side stream do not wait for
        seq, lab = seq.reshape(-1).to(torch.int32).contiguous(), lab.reshape(-1).to(torch.int32).contiguous()
        if not self.lists_side_stream:
            self._lists = self._sparse_lists(seq, lab, N)
            return
        if getattr(self, "_side", None) is None:
            self._side = side_stream(self.device, main)
        self._side.wait_stream(main)         # inputs ready; also orders reuse of last step's list memory after its reader
        with Engine._OnStream(self, self._side):
            self._lists = self._sparse_lists(seq, lab, N)
        self._lists_seq = (seq, lab)         # keep the inputs alive until the side stream has consumed them

    def _lists_wait(self):
        out, self._lists = self._lists, None
        if self.lists_side_stream:
            self._main.wait_stream(self._side)       # (the lists live in persistent workspace buffers: no record_stream needed)
        return out

    def _fused_table_adam(self, lr):
        """Table rows 1..N: gradient GEMM + sparse terms + Adam in one pass (ader_lbf_bwd_adam); all other parameters:
        the flat Adam kernel on the tail of the buffer.  Rows 0 and > N have zero gradient and zero Adam state (the
        catalog only grows), so leaving them untouched equals the dense update."""
        D = self._deferred
        st = self._stream()
        H, T = self.H, self.T
        lr_t = self._lr_t(lr)
        ids, order, sp_start, tids, torder, tg_start, tmeta = self._lists_wait()
        span = self.layout["pos"][0]

        def small_update():     # everything that feeds / is the update of the non-table parameters
            pl_ = self._pending_loss
            if pl_ is not None:
                call("ader_lbf_sum", ptr(pl_[0]), pl_[1], ptr(self.loss), self._stream())
                self._pending_loss = None
            self._flush_late()
            self._atb_flush()
            with self._sec("adam"):
                call("ader_adam_step", self.theta.data_ptr() + 4 * span, self.adam_m.data_ptr() + 4 * span,
                     self.adam_v.data_ptr() + 4 * span, self.grad.data_ptr() + 4 * span, self.P - span, lr_t, self.beta1,
                     self.beta2, self.eps, None, 0, H, self._stream())
            self._advance_adam()

        main = self._main
        overlap = bool(self._late or self._atb_q) and self.late_side_stream
        if overlap:
            # weight-gradient products, LayerNorm / positional reductions, small Adam and the bf16 weight planes are compute /
            # latency bound and independent of the table: a side stream runs them under the HBM-bound table update.  The side
            # stream's wait is placed HERE (behind the backward chain), the update is ENQUEUED FIRST and the small launches after
            # it: on the small catalogs of the shipped datasets the host is only a launch or two ahead of the GPU, and with the eight
            # small launches enqueued first the update reached the queue 100 us after the backward chain had finished
            # (profiles/r5_packed/timeline_cfgY_update_late.txt)
            if getattr(self, "_side", None) is None:
                self._side = side_stream(self.device, main)
            self._side.wait_stream(main)
        with self._sec("logits_bwd_adam"):
            if self.lx3:        # operand rows as the LDS images k_tab16x3 streams by LDS-DMA
                img = self.buf("lbf_rep_img", (call("ader_x3_rep_image_bytes", D["Bp"]),), torch.uint8, zero=True)
                if not self._img_ready:
                    call("ader_x3_rep_image", ptr(D["rep_bf"]), ptr(D["rep_lo"]), D["Bp"], ptr(img), st)
                self._img_ready = False
            if self.lx3 and D.get("kd"):
                K = D["kd"]
                call("ader_tab_update_x3_kd" if self.x3_update == "tab16" else "ader_tab_update_kd", ptr(D["rep_bf"]), ptr(D["rep_lo"]),
                     *((ptr(img),) if self.x3_update == "tab16" else ()), self.item_num, D["Bp"], K["row0"], H, D["N"], K["Np"],
                     ptr(D["off"]), ptr(ids), ptr(order), ids.numel(), ptr(D["g"]), float(np.sqrt(np.float32(H))), ptr(tids),
                     ptr(torder), tids.numel(), ptr(tmeta), ptr(D["wrow"]), ptr(K["teacher"]), K["teacher"].stride(0), ptr(K["trow"]),
                     ptr(K["tlse2"]), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t, self.beta1, self.beta2, self.eps, st)
            elif self.lx3 and self.x3_update == "tab16":
                call("ader_tab_update_x3", ptr(D["rep_bf"]), ptr(D["rep_lo"]), ptr(img), self.item_num, D["B"], D["Bp"], H, D["N"],
                     ptr(D["off"]), ptr(ids), ptr(order), ids.numel(), ptr(D["g"]), float(np.sqrt(np.float32(H))), ptr(tids),
                     ptr(torder), tids.numel(), ptr(tmeta), ptr(D["wrow"]), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t,
                     self.beta1, self.beta2, self.eps, 0, -1, ptr(D.get("extra")), st)
            elif self.lx3 or (self.bf16_update == "resident" and not D.get("kd")):
                call("ader_tab_update", ptr(D["rep_bf"]), ptr(D["rep_lo"]), ptr(self.shadow), self.item_num, D["B"], D["Bp"], H, D["N"],
                     ptr(D["off"]), ptr(ids), ptr(order), ids.numel(), ptr(D["g"]), float(np.sqrt(np.float32(H))), ptr(tids),
                     ptr(torder), tids.numel(), ptr(tmeta), ptr(D["wrow"]), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t,
                     self.beta1, self.beta2, self.eps, 0, -1, ptr(D.get("extra")), st)
            elif D.get("kd"):
                K = D["kd"]
                call("ader_tab_update_sh_kd", ptr(D["rep_bf"]), ptr(self.shadow), self.item_num, D["Bp"], K["row0"], H, D["N"], K["Np"],
                     ptr(D["off"]), ptr(ids), ptr(order), ptr(sp_start), ids.numel(), ptr(D["g"]), float(np.sqrt(np.float32(H))),
                     ptr(tids), ptr(torder), ptr(tg_start), tids.numel(), ptr(D["wrow"]), ptr(K["teacher"]), K["teacher"].stride(0),
                     ptr(K["trow"]), ptr(K["tlse2"]), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t, self.beta1, self.beta2,
                     self.eps, st)
            else:
                call("ader_tab_update_sh", ptr(D["rep_bf"]), ptr(self.shadow), self.item_num, D["B"], D["Bp"], H, D["N"],
                     ptr(D["off"]), ptr(ids), ptr(order), ptr(sp_start), ids.numel(), ptr(D["g"]), float(np.sqrt(np.float32(H))),
                     ptr(tids), ptr(torder), ptr(tg_start), tids.numel(), ptr(D["wrow"]), ptr(self.theta), ptr(self.adam_m),
                     ptr(self.adam_v), lr_t, self.beta1, self.beta2, self.eps, 0, -1, ptr(D.get("extra")), st)
        if overlap:
            with Engine._OnStream(self, self._side):
                small_update()
            main.wait_stream(self._side)
        else:
            small_update()
        self._deferred = None

    def _fused_table_adam_sharded(self, lr):
        """Data-parallel table update without the dense gradient exchange (SURVEY 8e "ZeRO-1 style"): instead of
        SUM-reducing the 600 MB table gradient, every rank all-gathers the INPUTS of the table-gradient product (bf16
        representations, per-row exponent offsets/labels/weights, and the sparse input-embedding gradient rows: ~16 MB per
        rank), runs the fused gradient+Adam kernel for the GLOBAL batch on ITS shard of table rows, and the updated rows
        are all-gathered.  Mathematically the same update as the dense all-reduce (sum over all rows of the global batch);
        Adam m/v of the table stay sharded.  The small parameters use a plain all-reduce."""
        import torch.distributed as dist
        self._refresh_stream()
        D = self._deferred
        st = self._stream()
        W, r, grp = self.dp_world, self.dp_rank, self.dp_group
        H, B, Bp, N = self.H, D["B"], D["Bp"], D["N"]
        lr_t = self._lr_t(lr)

        def ag(t):                      # [W, *t.shape]; moved as raw bytes (any dtype, any backend)
            t = t.contiguous()
            out = torch.empty((W,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self._guard("sharded-update:inputs", "all_gather", t.shape, t.dtype)
            dist.all_gather_into_tensor(out.view(torch.uint8).view(-1), t.view(torch.uint8).view(-1), group=grp)
            return out

        with self._sec("grad_exchange"):
            rep_g, off_g = ag(D["rep_bf"]), ag(D["off"])
            lab_g, w_g = ag(D["lab"]), ag(D["wrow"])
            seq_g, g_g = ag(D["seq"]), ag(D["g"])
            span = self.layout["pos"][0]
            self._guard("sharded-update:small-gradients", "all_reduce", (self.P - span,), self.grad.dtype)
            dist.all_reduce(self.grad[span:], group=grp)
            self._guard("sharded-update:loss", "all_reduce", self.loss.shape, self.loss.dtype)
            dist.all_reduce(self.loss, group=grp)
        tiles = self.shard_items // 128
        ids, order, sp_start, tids, torder, tg_start, tmeta = self._sparse_lists(seq_g, lab_g, N)
        tiles = self.shard_items // 128
        with self._sec("logits_bwd_adam"):
            call("ader_tab_update_sh", ptr(rep_g), ptr(self.shadow), self.item_num, W * Bp, W * Bp, H, N, ptr(off_g),
                 ptr(ids), ptr(order), ptr(sp_start), ids.numel(), ptr(g_g), float(np.sqrt(np.float32(H))), ptr(tids),
                 ptr(torder), ptr(tg_start), tids.numel(), ptr(w_g), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t,
                 self.beta1, self.beta2, self.eps, r * tiles, tiles, None, st)
        with self._sec("param_allgather"):
            S = self.shard_items * H
            table = self.theta[H:H + W * S]                       # rows 1 .. W*shard_items
            own = table[r * S:(r + 1) * S].clone()
            self._guard("sharded-update:table-rows", "all_gather", own.shape, own.dtype)
            dist.all_gather_into_tensor(table, own, group=grp)
            if self.shadow is not None:                           # bf16 shadow rows of the other shards
                call("ader_lbf_shadow_refresh", self._pp["emb"], ptr(self.shadow), self.V, H, st)
        with self._sec("adam"):
            call("ader_adam_step", self.theta.data_ptr() + 4 * span, self.adam_m.data_ptr() + 4 * span,
                 self.adam_v.data_ptr() + 4 * span, self.grad.data_ptr() + 4 * span, self.P - span, lr_t, self.beta1, self.beta2,
                 self.eps, None, 0, H, st)
        self._deferred = None
        self._mv_sharded = True
        self._advance_adam()

    # ---------------------------------------------------------------------------------------- catalog-sharded data parallelism
    def _ag(self, t, site=None):
        """all-gather -> [W, *t.shape]; moved as raw bytes (any dtype, any backend)."""
        import torch.distributed as dist
        t = t.contiguous()
        self._guard(site or "Engine._ag", "all_gather", t.shape, t.dtype)
        out = torch.empty((self.dp_world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(torch.uint8).view(-1), t.view(torch.uint8).view(-1), group=self.dp_group)
        return out

    def _guard(self, site, kind, shape, dtype, splits=None):
        """dist.CollectiveGuard hook: announce the collective about to be issued (a no-op unless the guard is on)."""
        from . import dist as adist
        if adist.guard.on:
            import sys
            f = sys._getframe(1)
            while f.f_back is not None and f.f_code.co_name in ("_guard", "_ag", "_a2a", "_a2a_rows"):
                f = f.f_back
            adist.guard.check("%s@%s:%d" % (site, os.path.basename(f.f_code.co_filename), f.f_lineno), kind, shape, dtype, splits)

    def _a2a(self, t):
        """t [W, ...]: slice j goes to rank j; returns [W, ...] with slice i received from rank i."""
        import torch.distributed as dist
        t = t.contiguous()
        self._guard("Engine._a2a", "all_to_all", t.shape, t.dtype)
        if dist.get_backend(self.dp_group) == "nccl":
            out = torch.empty_like(t)
            dist.all_to_all_single(out.view(torch.uint8).view(-1), t.view(torch.uint8).view(-1), group=self.dp_group)
            return out
        adist_on = self._guard_off()
        try:
            return self._ag(t)[:, self.dp_rank].contiguous()    # backends without all-to-all on device tensors (tests)
        finally:
            self._guard_on(adist_on)

    def _guard_off(self):
        """the gloo stand-ins of the all-to-alls are built from an all-gather: announced once, as the all-to-all they stand for"""
        from . import dist as adist
        was, adist.guard.on = adist.guard.on, False
        return was

    def _guard_on(self, was):
        from . import dist as adist
        adist.guard.on = was

    def _a2a_rows(self, rows, counts):
        """Uneven all-to-all of rows [K, H]: counts [W, W] (host ints), counts[i][j] = rows rank i sends to rank j; the local
        rows are ordered by destination.  Returns the received rows ordered by source."""
        import torch.distributed as dist
        W, r = self.dp_world, self.dp_rank
        ins = [int(c) for c in counts[r]]
        outs = [int(counts[i][r]) for i in range(W)]
        self._guard("Engine._a2a_rows", "all_to_all(uneven)", rows.shape[1:], rows.dtype, (ins, outs))
        out = torch.empty((sum(outs),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        if dist.get_backend(self.dp_group) == "nccl":
            dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=outs, input_split_sizes=ins, group=self.dp_group)
            return out
        # backends without all-to-all on device tensors (tests): padded all-gather, then cut my segments out
        kmax = max(int(sum(counts[i])) for i in range(W))
        pad = torch.zeros((kmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        pad[:rows.shape[0]] = rows
        adist_on = self._guard_off()
        try:
            allr = self._ag(pad)
        finally:
            self._guard_on(adist_on)
        segs = []
        for i in range(W):
            o = int(sum(counts[i][:r]))
            segs.append(allr[i, o:o + outs[i]])
        return torch.cat(segs) if segs else out

    def _train_step_catalog(self, seq, pos, max_item, lr, rate=0.0, n_train_global=None, ids_host=None, pack_counts=None,
                            teacher=None, ex_trow=None, lambda_=0.0, n_ex_global=None, **_unused):
        """Vanilla train step with the item catalog sharded across the ranks (dense Adam touches every table row
        every step, so a replicated table costs (W-1)/W x 600 MB of xGMI traffic per rank per step; a sharded one costs only
        the rows the inputs touch).  Rank r owns table rows [1 + r*S, (r+1)*S]: theta / m / v / shadow of other rows are
        not maintained locally.  Per step:
          1. ids of all ranks are all-gathered; every rank gathers the fp32 rows it owns for everybody's input positions and
             labels and an all-to-all delivers them (each position has exactly one owner); they are written into the local
             table so the unchanged forward kernel can read them;
          2. forward on the local rows; representations all-gathered (bf16 operand rows);
          3. softmax partials {max, sum, weighted row sum} of ALL global rows over the local item shard
             (ader_lbf_fwd_shard), exchanged so every rank merges the W partials of its own rows -> loss, dRep, offsets;
          4. local backward; per-position gradient rows, labels, weights and offsets all-gathered;
          5. fused gradient + Adam + shadow on the local shard for the global batch (ader_lbf_bwd_adam); nothing is sent back.
        Same update as a single process on the global batch (sum over rows; tests/test_gpu_dp.py).
        DISTILLED steps (float32 grade; ADER.py:132-137, main.py:223-256: `teacher` [*, Np] replicated on every rank, `ex_trow` the
        teacher rows of this rank's exemplar rows, which follow its train rows in `seq`): the exemplar rows of all ranks form a
        second block of the global batch ([all train rows | all exemplar rows]); their student softmax runs over the first Np items
        only (ader_lx3_fwd_shard with N = Np), the teacher readout O2 = sum_j softmax(t)_j E_j is summed shard by shard
        (ader_lx3_readout_shard), ader_lx3_merge_parts_kd turns the W partials into loss = w (lse - rep.O2) and dRep = w (O1/l - O2),
        and the fused update subtracts the teacher term for its item range (ader_tab_update_x3_kd_range) -- nothing proportional to
        the table is exchanged either (rounds 1-3: distilled steps fell back to the replicated table and a dense all-reduce)."""
        import torch.distributed as dist
        self._refresh_stream()
        W, r, grp = self.dp_world, self.dp_rank, self.dp_group
        seq, pos = self._seq_in(seq), self._dev_i32(pos)
        B_all, T, H, S = seq.shape[0], self.T, self.H, self.shard_items
        B = pos.shape[0]                                                   # train rows (the exemplar rows follow them in seq)
        n_ex = B_all - B
        kd = teacher is not None and n_ex > 0
        N = int(max_item)
        Bp = (B + 127) // 128 * 128
        Bk = (n_ex + 127) // 128 * 128 if kd else 0
        _check((n_ex == 0 or kd) and 1 <= B and B_all <= self.MAX_ROWS and 1 <= N <= self.item_num,
               "catalog-sharded step: 1 <= train rows, input rows <= %d, 1 <= max_item <= item_num; rows beyond the labels need "
               "exemplar_logits" % self.MAX_ROWS)
        Np = 0
        if kd:
            _check(self.lx3, "distilled catalog-sharded steps run at float32 grade (logits_dtype='x3')")
            ex_trow = self._dev_i32(ex_trow if ex_trow is not None else np.arange(n_ex))
            Np = teacher.shape[1]
            _check(teacher.dtype == torch.float32 and teacher.stride(1) == 1 and Np <= N and ex_trow.shape[0] == n_ex,
                   "exemplar_logits must be float32 [*, Np <= max_item] with one teacher row index per exemplar row")
        st = self._stream()
        step = self.global_step
        n_pos, n_all = B_all * T, B_all * T + B
        pack = self.dp_pack
        with self._sec("grad_exchange"):
            ids_l = torch.cat([seq.reshape(-1), pos])                          # my input positions, then my labels
            ids_g = self._ag(ids_l)                                            # [W, n_all]
            e_lab = self.buf("cs_elab", (B, H))
            if not pack:
                # dense exchange: every rank sends a full [n_all, H] block to every peer (zeros where it is not the owner)
                send = self.buf("cs_send", (W, n_all, H))
                call("ader_gather_owned", self._pp["emb"], ptr(ids_g), W * n_all, H, r * S, (r + 1) * S, ptr(send), st)
                recv = self._a2a(send)                                         # slice i: rows rank i owns among MY positions
                call("ader_scatter_owned", ptr(recv), ptr(ids_l), n_all, n_pos, H, S, W, self._pp["emb"], ptr(e_lab), st)
            else:
                # packed exchange: only owned rows travel.  ONE launch (csrc/pack_plan.hip) derives, from the gathered ids, the
                # [owner, destination] row counts -- the same matrix on every rank -- and every index list of the exchange.  The
                # counts are the split sizes of the uneven all-to-alls, which torch wants as host integers: when the caller knows
                # the global batch on the host (`ids_host`, [W, n_all], or `pack_counts`: bench.py precomputes the counts outside its
                # timed region, tests/test_gpu_dp.py passes ids_host; main.py runs the replicated scheme -- every rank builds the same
                # batches), they are computed there and the step has NO host synchronisation; otherwise they are read back (one sync).
                i64 = torch.int64
                cnt = self.buf("pk_cnt", (2, W, W), torch.int32)
                send_id, ids_bk = self.buf("pk_send", (W * n_all,), i64), self.buf("pk_back", (W * n_all,), i64)
                perm, bsrc = self.buf("pk_perm", (n_all,), i64), self.buf("pk_bsrc", (max(n_pos, 1),), i64)
                call("ader_pack_plan", ptr(ids_g), W, n_all, n_pos, r, S, ptr(cnt), ptr(send_id), ptr(ids_bk), ptr(perm), ptr(bsrc), st)
                if pack_counts is not None:                                     # (C_all, C_pos) prepared by the caller
                    C_all, C_pos = pack_counts
                    _check(all(len(C) == W and all(len(row) == W for row in C) for C in (C_all, C_pos)),
                           "pack_counts must be two %d x %d [owner][destination] count matrices" % (W, W))
                    self.comm_syncs = 0
                elif ids_host is not None:
                    # (the layout dist.global_ids_host gives: per rank its input positions, then its labels.  A distilled step's rows
                    #  are [train | exemplar] x T positions followed by the TRAIN labels only: n_all as computed above)
                    _check(tuple(np.asarray(ids_host).shape) == (W, n_all),
                           "ids_host must be [world = %d, %d] (rows * T input positions, then the labels, per rank); got %s"
                           % (W, n_all, tuple(np.asarray(ids_host).shape)))
                    C_all, C_pos = pack_counts_host(ids_host, n_pos, S)
                    self.comm_syncs = 0
                else:
                    C = cnt.cpu().tolist()
                    C_all, C_pos = C[0], C[1]
                    self.comm_syncs = 1
                if self.check_pack_counts and self.comm_syncs == 0 and self.global_step % self.check_pack_counts == 0:
                    # opt-in cross-check of the caller's host-side counts against the device plan (one host synchronisation): a
                    # mismatch would misplace rows in the exchange below or hang the uneven all-to-all
                    Cd = cnt.cpu().tolist()
                    _check(Cd[0] == [list(r_) for r_ in C_all] and Cd[1] == [list(r_) for r_ in C_pos],
                           "packed catalog exchange: the host-side split sizes (ids_host / pack_counts) differ from the device plan "
                           "-- the ranks did not build the same global batch")
                table = self.theta[:self.V_alloc * H].view(self.V_alloc, H)
                K = sum(C_all[r])                                               # rows I send, ordered by (destination, position)
                rows = table.index_select(0, send_id[:K])
                got = self._a2a_rows(rows, C_all)                              # ordered by owner, then by my position index
                n_pad = n_all - got.shape[0]                                   # my padding positions (id 0) come first in perm
                full = self.buf("cs_full", (n_all, H))
                full.zero_()
                full.index_copy_(0, perm[n_pad:], got)
                # (padding positions carry id 0 and zero rows: written over row 0, which the gather never reads -- it treats id 0 as
                #  the zero row, modules.py:124-126 -- and which is restored right away to stay bit-identical with the other modes)
                row0 = table[0].clone()
                table.index_copy_(0, ids_l[:n_pos].long(), full[:n_pos])
                table[0].copy_(row0)
                e_lab.copy_(full[n_pos:])
                # the gradient rows that travel back after the backward pass: ids of the rows I will receive (my owned entries among
                # everybody's INPUT positions, in (source, position) order) and my input positions grouped by owner
                Kb = sum(C_pos[r])
                ids_back = ids_bk[:Kb].to(torch.int32)
                back_src = bsrc[:sum(C_pos[o][r] for o in range(W))]
        self._table_stale = True
        lab_all = self.buf("cs_lab_all", (W, Bp), torch.int32)                 # labels in the padded row numbering of rep_g
        lab_all.zero_()
        lab_all[:, :B] = ids_g[:, n_pos:]
        # id-sorted lists of the sparse terms of the GLOBAL batch (side stream): positions of the all-gathered gradient rows, or,
        # packed, of the rows this rank will receive
        self._lists_async(ids_back if pack else ids_g[:, :n_pos], lab_all, N)
        # (a shard's train rows and exemplar rows sit at different global rows: two dropout counter segments, as in loss_and_grad)
        self.split_rows = B if (n_ex > 0 and getattr(self, "_ex_row0_set", False)) else None
        with self._sec("blocks_fwd"):
            rep = self.forward(seq, training=True, rate=rate, step=step, save=True)
        rep_bf = None if self.lx3 else self.buf("lbf_rep", (Bp * 168,), torch.bfloat16)
        w_row = 1.0 / float(n_train_global if n_train_global is not None else B)
        meta = self.buf("cs_meta", (3, Bp), torch.int32)                       # rows: off (f32 bits), wrow (f32 bits), label
        off, wrow, lab = meta[0].view(torch.float32), meta[1].view(torch.float32), meta[2]
        wrow.zero_()
        wrow[:B] = (pos > 0).to(torch.float32) * w_row          # label 0 = padding row of an equal-size shard: weight 0
        lab.zero_()
        lab[:B] = pos
        drep = self.buf("drep", (B_all, H))
        rl_all = self.buf("lg_rowloss", (Bp + Bk,))
        lse, rowloss = self.buf("lg_lse", (Bp,)), rl_all[:Bp]
        Bg = W * (Bp + Bk)                                                     # rows of the global batch: [all train | all exemplar]
        with self._sec("logits_fwd"):
            n_part = call("ader_lbf_ranges", S, W * Bp) * W * Bp               # range partials: rows x item ranges of the larger block
            if kd:
                n_part = max(n_part, call("ader_lbf_ranges", S, W * Bk) * W * Bk)
            pm, pl = self.buf("lbf_pm", (n_part,)), self.buf("lbf_pl", (n_part,))
            pO = self.buf("lbf_pO", (n_part * 160,))
            part = self.buf("lbf_part", (W * Bp * 152,))
            if self.lx3:
                # float32 grade: the fp32 representations travel (W * Bp * H floats), every rank cuts the hi / lo operand planes of
                # the GLOBAL batch itself and streams the fp32 rows of ITS shard
                # (pad rows [B, Bp) must be finite: they are all-gathered and enter every rank's S = rep.E^T as real rows, silenced
                #  only by off = -inf -- exp2(NaN - inf) is NaN and would poison the whole shard: cleared at allocation, and rows
                #  a larger earlier batch left behind are finite representations)
                rep_pad = self.buf("cs_rep_pad", (Bp, H), zero=True)
                rep_pad[:B].copy_(rep[:B])
                rep_f = self._ag(rep_pad)                                      # [W, Bp, H]
                if kd:
                    rep_pad_k = self.buf("cs_rep_pad_k", (Bk, H), zero=True)
                    rep_pad_k[:n_ex].copy_(rep[B:])
                    rep_f = torch.cat([rep_f.view(W * Bp, H), self._ag(rep_pad_k).view(W * Bk, H)])
                rep_g = self.buf("cs_rep_hi", (Bg * 168,), torch.bfloat16)
                rep_lo_g = self.buf("cs_rep_lo", (Bg * 168,), torch.bfloat16)
                call("ader_lx3_prep", ptr(rep_f), ptr(rep_g), ptr(rep_lo_g), Bg, Bg, H, st)
                call("ader_lx3_fwd_shard", ptr(rep_g), ptr(rep_lo_g), self._pp["emb"], self.item_num, W * Bp, H, N, r * S, S,
                     ptr(pm), ptr(pl), ptr(pO), ptr(part), st)
                pr = self._a2a(part.view(W, Bp, 152))                          # partials of MY rows from every rank
                call("ader_lx3_merge_parts", ptr(pr), W, Bp, B, H, ptr(e_lab), ptr(rep), ptr(wrow), ptr(lse), ptr(off),
                     ptr(rowloss), ptr(self.loss), ptr(drep), st)
                if kd:
                    # the exemplar block: per-row info of MY rows, gathered for the readout (teacher row, its log2-domain lse) ...
                    w_ex = float(lambda_) / float(n_ex_global if n_ex_global is not None else n_ex)
                    tl_all = self._teacher_lse(teacher, Np)
                    metak = self.buf("cs_metak", (4, Bk), torch.int32)         # rows: off, wrow, teacher row, tlse2 (f32 bits)
                    off_k, w_k, tr_k, tl2_k = (metak[0].view(torch.float32), metak[1].view(torch.float32), metak[2],
                                               metak[3].view(torch.float32))
                    tr_k.fill_(-1)
                    tr_k[:n_ex] = ex_trow
                    w_k.zero_()
                    w_k[:n_ex] = (ex_trow >= 0).to(torch.float32) * w_ex
                    tl2_k.zero_()
                    tl2_k[:n_ex] = tl_all[ex_trow.clamp(min=0).long()] * 1.4426950408889634
                    tinfo = self._ag(torch.stack([tr_k, metak[3]]))            # [W, 2, Bk]
                    tr_g = tinfo[:, 0].contiguous().view(-1)
                    tl2_g = tinfo[:, 1].contiguous().view(torch.float32).view(-1)
                    # ... student partials over MY items below Np and the teacher readout over the same items, for ALL exemplar rows
                    kd_off = W * Bp * 168
                    part_k, part_t = self.buf("lbf_part_k", (W * Bk * 152,)), self.buf("lbf_part_t", (W * Bk * 152,))
                    call("ader_lx3_fwd_shard", rep_g.data_ptr() + 2 * kd_off, rep_lo_g.data_ptr() + 2 * kd_off, self._pp["emb"],
                         self.item_num, W * Bk, H, Np, r * S, S, ptr(pm), ptr(pl), ptr(pO), ptr(part_k), st)
                    R2 = call("ader_lx3_readout_ranges", S, W * Bk)
                    pO2 = self.buf("lbf_pO2", (R2 * W * Bk * 160,))
                    call("ader_lx3_readout_shard", self._pp["emb"], self.item_num, W * Bk, H, Np, r * S, S, ptr(teacher),
                         teacher.stride(0), ptr(tr_g), ptr(tl2_g), ptr(pO2), ptr(part_t), st)
                    pr_k, pr_t = self._a2a(part_k.view(W, Bk, 152)), self._a2a(part_t.view(W, Bk, 152))
                    lse_k = self.buf("lg_lse_k", (Bk,))
                    call("ader_lx3_merge_parts_kd", ptr(pr_k), ptr(pr_t), W, Bk, n_ex, H, rep.data_ptr() + 4 * B * H, ptr(w_k),
                         ptr(lse_k), ptr(off_k), rl_all.data_ptr() + 4 * Bp, drep.data_ptr() + 4 * B * H, st)
                    call("ader_lbf_sum", ptr(rl_all), Bp + Bk, ptr(self.loss), st)
            else:
                call("ader_lbf_prep", ptr(rep), ptr(rep_bf), B, Bp, H, st)
                rep_g = self._ag(rep_bf)                                       # [W, Bp*168]
                call("ader_lbf_fwd_shard", ptr(rep_g), ptr(self.shadow), self.item_num, W * Bp, H, N, r * S, S, ptr(pm), ptr(pl),
                     ptr(pO), ptr(part), st)
                pr = self._a2a(part.view(W, Bp, 152))                          # partials of MY rows from every rank
                call("ader_lbf_merge_parts", ptr(pr), W, Bp, B, H, ptr(e_lab), ptr(rep_bf), ptr(wrow), ptr(lse), ptr(off),
                     ptr(rowloss), ptr(self.loss), ptr(drep), st)
        self._late_force = True            # weight-gradient products and small reductions are queued ...
        try:
            dx = self._blocks_backward(seq, drep, True, None)
        finally:
            self._late_force = False
        main = self._main
        if self._late or self._atb_q:      # ... and run on the side stream under the row exchange below (the CUs are idle there)
            if getattr(self, "_side", None) is None:
                self._side = side_stream(self.device, main)
            self._side.wait_stream(main)
            with Engine._OnStream(self, self._side):
                self._flush_late()
                self._atb_flush()
        lr_t = self._lr_t(lr)
        span = self.layout["pos"][0]
        with self._sec("grad_exchange"):
            meta_g = self._ag(meta)                                            # [W,3,Bp]
            if pack:    # gradient rows go only to the owner of their id: counts transposed with respect to the fetch
                g_g = self._a2a_rows(dx[back_src], [[C_pos[j][i] for j in range(W)] for i in range(W)])
            else:
                g_g = self._ag(dx)                                             # [W,B*T,H]
            off_g = meta_g[:, 0].contiguous().view(torch.float32)
            w_g = meta_g[:, 1].contiguous().view(torch.float32)
            if kd:      # the exemplar block behind the train block: offsets, weights, teacher rows, teacher lse of every rank's rows
                mk_g = self._ag(metak)                                         # [W,4,Bk]
                zt = torch.zeros(W * Bp, dtype=torch.int32, device=self.device)
                off_g = torch.cat([off_g.view(-1), mk_g[:, 0].contiguous().view(torch.float32).view(-1)])
                w_g = torch.cat([w_g.view(-1), mk_g[:, 1].contiguous().view(torch.float32).view(-1)])
                trow_g = torch.cat([zt - 1, mk_g[:, 2].contiguous().view(-1)])
                tlse2_g = torch.cat([zt.view(torch.float32), mk_g[:, 3].contiguous().view(torch.float32).view(-1)])
            main.wait_stream(self._side) if getattr(self, "_side", None) is not None else None   # small gradients complete
            self._guard("catalog:small-gradients", "all_reduce", (self.P - span,), self.grad.dtype)
            dist.all_reduce(self.grad[span:], group=grp)
            self._guard("catalog:loss", "all_reduce", self.loss.shape, self.loss.dtype)
            dist.all_reduce(self.loss, group=grp)
        ids, order, sp_start, tids, torder, tg_start, tmeta = self._lists_wait()
        tiles = S // 128
        with self._sec("logits_bwd_adam"):
            if self.lx3:
                img = self.buf("lbf_rep_img", (call("ader_x3_rep_image_bytes", Bg),), torch.uint8, zero=True)
                call("ader_x3_rep_image", ptr(rep_g), ptr(rep_lo_g), Bg, ptr(img), st)
                if kd:
                    call("ader_tab_update_x3_kd_range", ptr(rep_g), ptr(rep_lo_g), ptr(img), self.item_num, Bg, W * Bp, H, N, Np,
                         ptr(off_g), ptr(ids), ptr(order), ids.numel(), ptr(g_g), float(np.sqrt(np.float32(H))), ptr(tids),
                         ptr(torder), tids.numel(), ptr(tmeta), ptr(w_g), ptr(teacher), teacher.stride(0), ptr(trow_g), ptr(tlse2_g),
                         ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t, self.beta1, self.beta2, self.eps, r * tiles,
                         tiles, st)
                else:
                    call("ader_tab_update_x3", ptr(rep_g), ptr(rep_lo_g), ptr(img), self.item_num, W * Bp, W * Bp, H, N, ptr(off_g),
                         ptr(ids), ptr(order), ids.numel(), ptr(g_g), float(np.sqrt(np.float32(H))), ptr(tids), ptr(torder),
                         tids.numel(), ptr(tmeta), ptr(w_g), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t, self.beta1,
                         self.beta2, self.eps, r * tiles, tiles, None, st)
            else:
                call("ader_tab_update_sh", ptr(rep_g), ptr(self.shadow), self.item_num, W * Bp, W * Bp, H, N, ptr(off_g),
                     ptr(ids), ptr(order), ptr(sp_start), ids.numel(), ptr(g_g), float(np.sqrt(np.float32(H))), ptr(tids),
                     ptr(torder), ptr(tg_start), tids.numel(), ptr(w_g), ptr(self.theta), ptr(self.adam_m), ptr(self.adam_v), lr_t,
                     self.beta1, self.beta2, self.eps, r * tiles, tiles, None, st)
        with self._sec("adam"):
            call("ader_adam_step", self.theta.data_ptr() + 4 * span, self.adam_m.data_ptr() + 4 * span,
                 self.adam_v.data_ptr() + 4 * span, self.grad.data_ptr() + 4 * span, self.P - span, lr_t, self.beta1, self.beta2,
                 self.eps, None, 0, H, st)
        self._mv_sharded = True
        self._advance_adam()
        return self.loss

    def sync_table(self):
        """Catalog-sharded mode: make the parameter rows of the whole table (and their bf16 shadow) valid on every rank again
        -- before evaluation, herding, checkpointing or a step that needs the replicated table."""
        if not self._table_stale:
            return
        import torch.distributed as dist
        H, S = self.H, self.shard_items * self.H
        tab = self.theta[H:H + self.dp_world * S]
        own = tab[self.dp_rank * S:(self.dp_rank + 1) * S].clone()
        self._guard("sync_table", "all_gather", own.shape, own.dtype)
        dist.all_gather_into_tensor(tab, own, group=self.dp_group)
        self._table_stale = False
        self.refresh_shadow()

    def gather_table_state(self):
        """Sharded mode: make adam_m / adam_v of the table complete on every rank (before checkpointing)."""
        if self.dp_world > 1 and (self.dp_sharded or self.dp_mode == "catalog"):
            import torch.distributed as dist
            self.sync_table()
            H, S = self.H, self.shard_items * self.H
            for buf in (self.adam_m, self.adam_v):
                table = buf[H:H + self.dp_world * S]
                own = table[self.dp_rank * S:(self.dp_rank + 1) * S].clone()
                self._guard("gather_table_state", "all_gather", own.shape, own.dtype)
                dist.all_gather_into_tensor(table, own, group=self.dp_group)

    def train_step(self, seq, pos, max_item, lr, **kw):
        """One `sess.run(train_op)` (main.py:233-256): forward, loss, backward, [gradient exchange], Adam.
        Returns the loss as a 1-element device tensor (no host sync)."""
        self._refresh_stream()
        self._in_step = True
        try:
            return self._train_step(seq, pos, max_item, lr, **kw)
        finally:
            self._in_step = False

    def _train_step(self, seq, pos, max_item, lr, **kw):
        if (self.dp_world > 1 and self.dp_mode == "catalog" and (self.shadow is not None or self.lx3) and self.seq_fused
                and kw.get("ex_pos") is None and (kw.get("teacher") is None or self.lx3)):
            return self._train_step_catalog(seq, pos, max_item, lr, **kw)
        kw.pop("ids_host", None)           # (host-side knowledge of the global batch: only the packed catalog exchange uses it)
        kw.pop("pack_counts", None)
        self.sync_table()
        sharded = self.dp_world > 1 and self.dp_sharded and self.shadow is not None      # (x3 / f32 logits: dense exchange)
        fuse = self.fuse_adam and (self.grad_hook is None or sharded)
        if self.ewc is not None and self.ewc["lam"] != 0.0:
            fuse, sharded = False, False         # the penalty's gradient lives in the dense gradient buffer
        loss = self.loss_and_grad(seq, pos, max_item, _defer_table=fuse, **kw)
        if self._deferred is not None:
            if sharded:
                self._fused_table_adam_sharded(lr)
            else:
                self._fused_table_adam(lr)
            return loss
        if self.grad_hook is not None:
            with self._sec("grad_exchange"):
                self.grad_hook(self)
        if self.ewc is not None and self.ewc["lam"] != 0.0:
            # loss += lambda/2 sum F (theta - theta_prev)^2 and its gradient (EWC.py:121-124); every rank holds the same F / prev
            call("ader_ewc_penalty", ptr(self.theta), ptr(self.ewc["prev"]), ptr(self.ewc["F"]), ptr(self.grad), self.P,
                 float(self.ewc["lam"]), ptr(self.buf("ewc_part", (1024,))), ptr(self.loss), self._stream())
        self.adam(lr)
        return loss

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise _lib.AderHipError("device status %d: item id outside [0, item_num] in input_seq" % s)

    # ---------------------------------------------------------------------------------------- inference paths
    def encode(self, seq):
        """Eval-mode representation (is_training=False): rep [n,H] for any n (chunks of MAX_ROWS)."""
        self._refresh_stream()
        self.sync_table()
        seq = self._seq_in(seq)
        n = seq.shape[0]
        out = torch.empty((n, self.H), dtype=torch.float32, device=self.device)
        for s in range(0, n, self.MAX_ROWS):
            e = min(n, s + self.MAX_ROWS)
            out[s:e] = self.forward(seq[s:e], training=False)
        return out

    def _ncol_all(self, Bp, B, N):
        t = self.buf("ncol_all", (Bp,), torch.int32)
        t.zero_()
        t[:B] = N
        return t

    def logits_from_rep(self, rep, max_item, out=None):
        """Dense logits [n, N] = rep . E[1..N]^T  (ADER.py:92)."""
        self._refresh_stream()
        self.sync_table()
        n, N = rep.shape[0], int(max_item)
        if out is None:
            # row stride padded to 16 bytes: the teacher readout of a distilled step streams these rows 16 bytes at a time (k_lx3r);
            # with an odd stride -- max_item is whatever the previous period's catalog was -- it falls back to the slower kernel
            out = torch.empty((n, (N + 3) // 4 * 4), dtype=torch.float32, device=self.device)[:, :N]
        for s in range(0, n, self.MAX_ROWS):
            e = min(n, s + self.MAX_ROWS)
            B = e - s
            Bp = (B + 63) // 64 * 64
            r = rep[s:e].contiguous()
            call("ader_logits_store", ptr(r), self._pp["emb"], B, Bp, self.H, N, ptr(self._ncol_all(Bp, B, N)),
                 ptr(out[s:e]), out.stride(0), self._stream())
        return out

    def logits(self, seq, max_item):
        return self.logits_from_rep(self.encode(seq), max_item)

    def teacher_logits(self, seq, max_item):
        self._refresh_stream()
        return self.logits(seq, max_item)

    def rank_targets(self, seq, pos, max_item):
        """0-based rank of pos[b] among items 1..N for every row (Evaluator path, util.py:323-325) -> int32 numpy [n]."""
        self._refresh_stream()
        self.sync_table()
        seq = self._seq_in(seq)
        pos = self._dev_i32(pos)
        n, N = seq.shape[0], int(max_item)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        for s in range(0, n, self.MAX_ROWS):
            e = min(n, s + self.MAX_ROWS)
            B = e - s
            Bp = (B + 63) // 64 * 64
            rep = self.forward(seq[s:e], training=False)
            tl = self.buf("rk_tl", (Bp,))
            rk = self.buf("rk_rank", (Bp,), torch.int32)
            tgt = self.buf("rk_tgt", (Bp,), torch.int32)
            tgt.zero_()
            tgt[:B] = pos[s:e]
            call("ader_rank_targets", ptr(rep), self._pp["emb"], B, Bp, self.H, N, ptr(tgt), ptr(self._ncol_all(Bp, B, N)),
                 ptr(tl), ptr(rk), self._stream())
            out[s:e] = rk[:B]
        return out.cpu().numpy()

    # ---------------------------------------------------------------------------------------- EWC baseline (EWC.py:115-164)
    def ewc_snapshot(self):
        """variables_prev = sess.run(model.variables) (main.py:260,321): the parameters the penalty pulls towards."""
        self._refresh_stream()
        if self.ewc is None:
            self.ewc = {"F": torch.zeros(self.P, dtype=torch.float32, device=self.device), "lam": 0.0}
        self.ewc["prev"] = self.theta.detach().clone()

    def compute_fisher(self, seq, pos, max_item):
        """Diagonal Fisher information of EWC.py:126-164: the mean over the n given sub-sequences of the SQUARED per-sample
        gradient of the eval-mode cross entropy (batch of one, dropout off) w.r.t. every parameter -> self.ewc["F"] (flat, the
        parameter layout).  One forward / backward per sample like the reference (n <= --ewc_sample_num = 1000).
        Deviation, on purpose: the reference densifies the position table's IndexedSlices gradient with `dense[idx] = value`
        (EWC.py:153-157), an ASSIGNMENT -- for a row that occurs more than once in a sample only the last slice survives -- whereas
        the SUMMED gradient is squared here (the mathematical Fisher diagonal).  Only rows repeated inside one sample differ
        (tests/test_gpu_extras.py pins the summed semantics against the oracle)."""
        self._refresh_stream()
        self.sync_table()
        if self.ewc is None:
            self.ewc_snapshot()
        seq, pos = self._seq_in(seq), self._dev_i32(pos)
        n = seq.shape[0]
        F = self.ewc["F"]
        F.zero_()
        hook, early = self.grad_hook, self.grad_early_hook
        self.grad_hook = self.grad_early_hook = None
        step = self.global_step
        try:
            for i in range(n):
                # (loss_and_grad overwrites only the table rows <= max_item; rows above were never touched and are zero)
                self.loss_and_grad(seq[i:i + 1], pos[i:i + 1], max_item, rate=0.0)
                call("ader_sq_accum", ptr(self.grad), ptr(F), self.P, 1.0 / n, self._stream())
        finally:
            self.grad_hook, self.grad_early_hook = hook, early
            self.global_step = step
        return F

    def row_losses(self, seq, pos, max_item):
        """Per-row cross entropy -log softmax(logits)[label] in eval mode (the quantity the reference's `loss` exemplar selector
        means to rank by, util.py:463-495; its graph fetches the batch MEAN, see ExemplarGenerator.loss_selection) -> float32 [n]
        device tensor.  Exact-f32 logit kernels, chunks of MAX_ROWS rows."""
        self._refresh_stream()
        self.sync_table()
        seq, pos = self._seq_in(seq), self._dev_i32(pos)
        n, N = seq.shape[0], int(max_item)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        st = self._stream()
        parts = call("ader_logits_parts", N)
        scr = torch.empty(1, dtype=torch.float32, device=self.device)
        for s in range(0, n, self.MAX_ROWS):
            e = min(n, s + self.MAX_ROWS)
            B = e - s
            rep = self.forward(seq[s:e], training=False)
            Bp, ri = self._rowinfo(B, pos[s:e].contiguous(), B, None, None, N, 0, 1.0, 0.0, None, tag="rl_")
            part = self.buf("lg_part", (parts * Bp * 3,))
            lse, rowloss = self.buf("rl_lse", (Bp,)), self.buf("rl_rowloss", (Bp,))
            call("ader_logits_loss_fwd", ptr(rep), self._pp["emb"], B, Bp, self.H, N, *ri, ptr(part), ptr(lse), ptr(rowloss),
                 ptr(scr), st)
            out[s:e] = rowloss[:B]
        return out

    def herding_select(self, seq_rows, offs, quota, max_item):
        """Segmented herding over label groups (util.py:436-461).  seq_rows [n,T] candidates in group order, offs [G+1],
        quota [G] = min(m, n_g).  Returns (sel [n] local indices per group span, sel_cnt [G]) as numpy."""
        self._refresh_stream()
        from .exemplar import herding_max_steps
        rep = self.encode(seq_rows)
        n, G = rep.shape[0], len(quota)
        seg = torch.as_tensor(np.asarray(offs, dtype=np.int64)).to(self.device)
        q = torch.as_tensor(np.asarray(quota, dtype=np.int32)).to(self.device)
        ms = torch.as_tensor(np.array([herding_max_steps(int(m)) for m in quota], dtype=np.int32)).to(self.device)
        D = torch.empty(n * self.H + G + 64, dtype=torch.float32, device=self.device)     # normalised columns + the device-built work list
        chosen = torch.empty(max(n, 1), dtype=torch.uint8, device=self.device)
        sel = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        cnt = torch.zeros(max(G, 1), dtype=torch.int32, device=self.device)
        call("ader_herding_select", ptr(rep), ptr(seg), ptr(q), ptr(ms), G, n, self.H, ptr(D), ptr(chosen), ptr(sel), ptr(cnt),
             None, self._stream())
        return sel.cpu().numpy().astype(np.int64), cnt.cpu().numpy()
