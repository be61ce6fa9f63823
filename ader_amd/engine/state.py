"""Engine state: the flat parameter / Adam / gradient buffers of the reference graph's variables, persistent workspaces,
streams and checkpoints."""
import collections
import math
import os

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr
from .common import _NULL, _check, param_layout, side_stream


class _State:
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
        self._side, self._side_for = None, None
        # native step driver (plan.py): steps of a plannable form are recorded once per (shape, mode) and replayed by one C call
        self.native_step = os.environ.get("ADER_NATIVE_STEP", "1") != "0"
        self.plan_verify = False           # tests: every replayable step runs through Python again and is compared with its plan
        self.plan_hits = self.plan_misses = self.plan_verified = 0
        self.plan_errors = []
        self._plans, self._plans_gen, self._keep_density, self._held, self._density_now = None, -1, False, None, None
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

    def _side_lane(self):
        """The engine's second lane for the CURRENT main stream (common.side_stream probes the hardware queues once per (device, main
        stream) of the process; resolved again when the caller moves the engine to another main stream)."""
        m = self._main
        if self._side is None or (self._side_for is not None and self._side_for != m.cuda_stream):      # (a lane set by hand stays)
            self._side, self._side_for = side_stream(self.device, m), m.cuda_stream
        return self._side

    def warm_up(self):
        """Everything a first train step would otherwise do inside the step (and inside a timed region): the side-stream probe -- a
        64 MB scratch tensor, eight streams and ~16 device synchronisations."""
        self._refresh_stream()
        self._side_lane()
        return self

    def _edge(self, waiter, other):
        """Stream `waiter` waits for everything enqueued on stream `other` so far (an event edge; recorded into a launch plan)."""
        waiter.wait_stream(other)
        rec = _lib.recorder
        if rec is not None:
            rec.wait(waiter.cuda_stream, other.cuda_stream)

    def _dev_i32(self, x):
        """int32 device tensor of a host array / tensor.  Host arrays go through a small ring of pinned staging buffers and an
        asynchronous copy: a pageable .to(device) would block the host until the GPU has drained (one sync per step)."""
        if isinstance(x, torch.Tensor):
            if x.device == self.device and x.dtype == torch.int32 and x.is_contiguous():
                return x
            return x.to(device=self.device, dtype=torch.int32).contiguous()
        a = np.ascontiguousarray(x, dtype=np.int32)
        if a.nbytes > (1 << 20):          # a whole evaluation set, not a step's batch: one plain copy (no ring of pinned buffers per shape)
            return torch.from_numpy(a).to(self.device)
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

    def check_status(self):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise _lib.AderHipError("device status %d: item id outside [0, item_num] in input_seq" % s)
