"""Forward of the session stack (modules.py:118-266, ADER.py:41-85): per-op, one-launch and packed-tile forms."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr
from .common import EPI_BIAS, EPI_BIAS_DROP_RES_MASK, EPI_BIAS_RELU_DROP, SITE_EMB, _Drop, dropout_key, site_attn, site_ffn1, site_ffn2


class _Forward:
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
            if not self._keep_density:         # (a step being recorded re-enters with the device copy of a batch whose density is known)
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
        # rows expected to exist (how the weight-gradient workgroups are shared out): this batch's own density when it came from the host,
        # else the feeder's announcement, else -- nothing known -- the densest batch "auto" would still pack (a low guess would share a
        # dense batch's products over too few workgroups: correct, but silently slow)
        d = getattr(self, "_density_now", None)
        if d is None:
            d = self.pack_density if self.pack_density is not None else self.PACK_DENSITY_MAX
        est = int(min(n, max(64, 1.25 * d * B * T + 64)))
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
