"""Inference paths: encode, logits, ranks (util.py:323-325), row losses, herding selection (util.py:436-461)."""
import numpy as np
import torch

from .._lib import call, ptr


class _Infer:
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
        from ..exemplar import herding_max_steps
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
