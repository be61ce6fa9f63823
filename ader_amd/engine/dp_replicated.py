"""Data parallel, replicated table: row-sharded fused table update."""
import numpy as np
import torch

from .._lib import call, ptr


class _DpReplicated:
    def _gather_if_sharded(self):
        """Data-parallel sharded / catalog steps update Adam m/v (and, in catalog mode, theta) of the rank's OWN table rows only.
        Before anything that needs the whole optimiser state on every rank -- a checkpoint, or a dense Adam step (distilled
        periods) -- the shards are all-gathered.  A collective: every rank reaches these points together (state_dict / adam)."""
        if self.dp_world > 1 and self._mv_sharded:
            self.gather_table_state()
            self._mv_sharded = False

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
