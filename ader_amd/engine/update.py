"""Optimiser side of the step (ADER.py:96, main.py:233-256): sparse lists, fused table update + Adam, train_step dispatch, EWC."""
import numpy as np
import torch

from .._lib import call, ptr
from .common import IN_LR, StepF


class _Update:
    def _lr_t(self, lr):
        """Adam's step size lr * sqrt(1 - beta2^t) / (1 - beta1^t) in float32 (tf.train.AdamOptimizer; the one launcher argument that
        changes every step: marked for the launch-plan recorder)."""
        return StepF(np.float32(lr) * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p), IN_LR)

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
        self._edge(self._side_lane(), main)      # inputs ready; also orders reuse of last step's list memory after its reader
        with self._OnStream(self, self._side):
            self._lists = self._sparse_lists(seq, lab, N)
            pl_ = self._pending_loss
            if pl_ is not None:
                # the loss scalar feeds nothing in this step: its (single-workgroup) sum rides behind the lists on the side lane -- the
                # row losses are complete (the edge above follows the logit forward) -- instead of heading the side lane's chain beside
                # the table update, which at the shipped datasets' shapes ends a few microseconds after the update does
                call("ader_lbf_sum", ptr(pl_[0]), pl_[1], ptr(self.loss), self._stream())
                self._pending_loss = None
        self._lists_seq = (seq, lab)         # keep the inputs alive until the side stream has consumed them

    def _lists_wait(self):
        out, self._lists = self._lists, None
        if self.lists_side_stream:
            self._edge(self._main, self._side)       # (the lists live in persistent workspace buffers: no record_stream needed)
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
            self._edge(self._side_lane(), main)
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
            with self._OnStream(self, self._side):
                small_update()
            self._edge(main, self._side)
        else:
            small_update()
        self._deferred = None

    def train_step(self, seq, pos, max_item, lr, **kw):
        """One `sess.run(train_op)` (main.py:233-256): forward, loss, backward, [gradient exchange], Adam.
        Returns the loss as a 1-element device tensor (no host sync)."""
        self._refresh_stream()
        self._in_step = True
        try:
            out = self._native_step(seq, pos, max_item, lr, kw)        # one C call per step when a launch plan exists (plan.py)
            return out if out is not None else self._train_step(seq, pos, max_item, lr, **kw)
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
