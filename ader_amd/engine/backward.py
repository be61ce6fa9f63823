"""Loss (ADER.py:87-137) and backward of one step: logit kernels, block backward, weight-gradient products."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr
from .common import EPI_ADD, EPI_BIAS, EPI_RELUDROPGRAD, _check


class _Backward:
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
            if n_ex == 0 or split_kd:
                labs = pos
            else:       # one-hot replay: train labels, then exemplar labels (a launcher, not torch.cat: recordable, no allocation)
                labs = self.buf("labs_cat", (n_train + n_ex,), torch.int32)
                call("ader_concat_i32", ptr(pos), n_train, ptr(ex_pos), n_ex, ptr(labs), st)
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
