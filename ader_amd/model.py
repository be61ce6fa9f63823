"""`Ader` with the reference's call surface (ADER.py:13-150) over the MI355X engine, plus `Session` / `Saver`
shims so `main.py` / `util.py`-style code (``sess.run(model.train_op, feed_dict)``, ``model.predict(sess, ...)``,
``saver.save/restore``) runs unchanged.

The reference builds a TF1 graph once (main.py:143-144) and drives it with ``sess.run``.  Here the "graph handles"
(`input_seq`, `pos`, `is_training`, `max_item`, `exemplar_logits`, `exemplar_pos`, `dropout_rate`, `lr`, `test_item`,
`rep`, `logits`, `loss`, `train_op`, `pred_last`) are inert tokens; `Session.run` dispatches on the fetched token and
reads the feeds by token.  Next to the shim are the fast paths that take device tensors and never leave the GPU:
`train_step`, `encode`, `rank_targets`.
"""
import numpy as np
import torch

from .engine import Engine


class _Handle:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<ader handle %s>" % self.name


class Ader:
    def __init__(self, item_num, args, reuse=None, device="cuda:0", logits_dtype=None, dp_rank=0, dp_world=1):
        self.args = args
        self.item_num = item_num
        ld = logits_dtype or getattr(args, "logits_dtype", "x3")
        self.engine = Engine(item_num, maxlen=args.maxlen, hidden_units=args.hidden_units, num_blocks=args.num_blocks,
                             num_heads=args.num_heads, seed=args.random_seed, device=device, logits_dtype=ld,
                             dp_rank=dp_rank, dp_world=dp_world)
        # session kernels on the real positions only (packed tiles): "auto" goes by the density of the batches (engine.py)
        self.engine.pack_sessions = {"auto": "auto", "on": True, "off": False}[getattr(args, "pack_sessions", "auto")]
        for n in ("is_training", "input_seq", "pos", "exemplar_logits", "exemplar_pos", "max_item", "lr", "dropout_rate",
                  "test_item", "rep", "logits", "loss", "train_op", "pred_last", "exemp_loss"):
            setattr(self, n, _Handle(n))
        self._loss_mode = "vanilla"
        self._lambda = 0.0

    # ---- loss selection (reference: re-issues optimizer.minimize on a different loss tensor, ADER.py:105-138)
    def set_vanilla_loss(self):
        self._loss_mode, self._lambda = "vanilla", 0.0

    def update_loss(self, lambda_):
        self._loss_mode = "onehot" if getattr(self.args, "disable_distillation", False) else "kd"
        self._lambda = float(lambda_)

    # ---- fast paths (device tensors / numpy in, no per-step host round trip)
    def train_step(self, seq, pos, max_item, lr, dropout_rate, *, ex_pos=None, teacher=None, ex_trow=None, **kw):
        """One optimisation step.  Exemplar rows (if any) are the last rows of `seq` (main.py:229)."""
        e = self.engine
        if self._loss_mode == "vanilla":
            return e.train_step(seq, pos, max_item, lr, rate=dropout_rate, **kw)
        if self._loss_mode == "onehot":
            return e.train_step(seq, pos, max_item, lr, rate=dropout_rate, ex_pos=ex_pos, lambda_=self._lambda, **kw)
        return e.train_step(seq, pos, max_item, lr, rate=dropout_rate, teacher=teacher, ex_trow=ex_trow, lambda_=self._lambda, **kw)

    def train_step_fed(self, feed, max_item, lr, dropout_rate, teacher=None):
        """One optimisation step whose batch the engine cuts on the device from the Samplers' GPU-resident rows (Engine.train_step_fed;
        feed as described there).  The loss follows set_vanilla_loss / update_loss like train_step."""
        e = self.engine
        if self._loss_mode == "vanilla":
            return e.train_step_fed(feed, max_item, lr, dropout_rate)
        if self._loss_mode == "onehot":
            return e.train_step_fed(feed, max_item, lr, dropout_rate, lambda_=self._lambda, onehot=True)
        return e.train_step_fed(feed, max_item, lr, dropout_rate, teacher=teacher, lambda_=self._lambda)

    def encode(self, seq):
        return self.engine.encode(seq)

    def rank_targets(self, seq, pos, max_item):
        return self.engine.rank_targets(seq, pos, max_item)

    def predict(self, sess, seq, item_idx):
        """Rank of every candidate item (ADER.py:140-150).  Kept for API parity; the Evaluator uses rank_targets."""
        return sess.run(self.pred_last, {self.input_seq: seq, self.test_item: item_idx, self.is_training: False,
                                         self.dropout_rate: self.args.dropout_rate})


class Ewc(Ader):
    """EWC baseline (reference EWC.py:14-164) on the same engine: the SASRec graph with the cross-entropy loss plus
    lambda/2 * sum_v F_v (theta_v - theta_prev_v)^2.  `variables_prev` and `F_accum` live on the device in the flat parameter
    layout (the reference holds them as numpy arrays baked into the graph when update_loss is called, EWC.py:115-124)."""

    def update_loss(self, lambda_):
        """EWC.py:115-124: from now on train with the penalty, using the Fisher information and the snapshot taken LAST (the
        reference bakes the arrays of the moment of this call into the graph: later compute_fisher calls do not change the loss
        of the running period)."""
        e = self.engine
        if e.ewc is None or "prev" not in e.ewc:
            e.ewc_snapshot()
        e.ewc["F_live"] = e.ewc["F"].clone()
        e.ewc["prev_live"] = e.ewc["prev"].clone()
        e.ewc["lam"] = float(lambda_)
        self._loss_mode, self._lambda = "vanilla", 0.0

    def set_vanilla_loss(self):
        if self.engine.ewc is not None:
            self.engine.ewc["lam"] = 0.0
        super().set_vanilla_loss()

    def train_step(self, seq, pos, max_item, lr, dropout_rate, **kw):
        e = self.engine
        if e.ewc is not None and e.ewc["lam"] != 0.0:
            live = {"F": e.ewc["F_live"], "prev": e.ewc["prev_live"], "lam": e.ewc["lam"]}
            saved, e.ewc = e.ewc, live
            try:
                return e.train_step(seq, pos, max_item, lr, rate=dropout_rate, **{k: v for k, v in kw.items() if k.startswith("n_")})
            finally:
                e.ewc = saved
        return e.train_step(seq, pos, max_item, lr, rate=dropout_rate, **{k: v for k, v in kw.items() if k.startswith("n_")})

    def snapshot_variables(self):
        """model.variables_prev = sess.run(model.variables) (main.py:260, 321)."""
        self.engine.ewc_snapshot()

    def compute_fisher(self, sess, data, batch_size, max_item, dry_run=False):
        """EWC.py:126-164 over the sub-sequences `data` (lists [items..., label]); `batch_size` only shapes the reference's
        sampler batches (every sample is differentiated on its own), but the sampler's shuffle consumes the `random` stream and
        fixes the sample order, so it is kept.  dry_run: consume the random stream only (see main.py)."""
        from .data import Sampler
        smp = Sampler(data, self.args.maxlen, batch_size, is_subseq=True)
        if dry_run:
            # the reference draws batch_num() batches (EWC.py:139-141): the last draw wraps around and reshuffles the index list
            # (util.py:232-237), a second consumption of the `random` stream that a dry run must reproduce as well
            for _ in range(smp.batch_num()):
                smp._next_indices()
            return
        seqs, poss = [], []
        for _ in range(smp.batch_num()):
            s, p = smp.next_batch()
            seqs.append(np.asarray(s.cpu() if hasattr(s, "cpu") else s))
            poss.append(np.asarray(p.cpu() if hasattr(p, "cpu") else p))
        if not seqs:
            return
        self.engine.compute_fisher(np.concatenate(seqs), np.concatenate(poss), max_item)


class Session:
    """Minimal `tf.Session` stand-in: context manager + run(fetches, feed_dict)."""

    def __init__(self, model=None, config=None):
        self.model = model

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    @staticmethod
    def _model_of(fetches, feed):
        raise RuntimeError("Session needs the model: construct Session(model)")

    def run(self, fetches, feed_dict=None):
        m = self.model or self._model_of(fetches, feed_dict)
        feed = {k.name: v for k, v in (feed_dict or {}).items()}
        single = not isinstance(fetches, (list, tuple))
        names = [fetches.name] if single else [f.name for f in fetches]
        e = m.engine
        if names == ["train_op"]:
            seq = np.asarray(feed["input_seq"], dtype=np.int32)
            pos = np.asarray(feed["pos"], dtype=np.int32).reshape(-1)
            kw = {}
            if "exemplar_logits" in feed and m._loss_mode == "kd":
                rows = feed["exemplar_logits"]
                if len(rows) and isinstance(rows[0], torch.Tensor):
                    kw["teacher"] = torch.stack([r.to(e.device) for r in rows]).float().contiguous()
                else:
                    kw["teacher"] = torch.as_tensor(np.asarray(rows, dtype=np.float32)).to(e.device)
            elif "exemplar_pos" in feed and m._loss_mode == "onehot":
                kw["ex_pos"] = np.asarray(feed["exemplar_pos"], dtype=np.int32).reshape(-1)
            rate = float(feed.get("dropout_rate", 0.0)) if feed.get("is_training", True) else 0.0
            m.train_step(seq, pos, int(feed["max_item"]), float(feed["lr"]), rate, **kw)
            return None
        seq = np.asarray(feed["input_seq"], dtype=np.int32)
        out = []
        rep = None
        for n in names:
            if n == "rep":
                rep = e.encode(seq) if rep is None else rep
                out.append(rep.cpu().numpy())
            elif n == "logits":
                rep = e.encode(seq) if rep is None else rep
                out.append(e.logits_from_rep(rep, int(feed["max_item"])).cpu().numpy())
            elif n == "pred_last":
                # pred_last = argsort(argsort(-test_logits)) (ADER.py:99-103): the 0-based rank of EVERY item, ties -> lower index
                # first.  rank(i) = #{j: l_j > l_i} + #{j < i: l_j == l_i}: from one sort and its run starts, no second sort.
                items = np.asarray(feed["test_item"], dtype=np.int64)
                N = int(items.max()) if items.size else 0
                if items.size == 0 or not np.array_equal(items, np.arange(1, N + 1)):
                    raise RuntimeError("predict(): test_item must be the contiguous catalog 1..N (the reference's only call site, "
                                       "util.py:323)")
                rep = e.encode(seq) if rep is None else rep
                lg = e.logits_from_rep(rep, N)
                order = torch.argsort(lg, dim=-1, descending=True, stable=True)          # ties keep index order
                rk = torch.empty_like(order, dtype=torch.int32)
                rk.scatter_(1, order, torch.arange(N, device=lg.device, dtype=torch.int32).expand_as(order))
                out.append(rk.cpu().numpy())
            else:
                raise KeyError("unsupported fetch %r" % n)
        return out[0] if single else out


class Saver:
    """`tf.train.Saver(max_to_keep=1)` stand-in: parameters + Adam slots + beta powers + global_step (main.py:209-213,
    280, 283 save/restore ALL global variables, so optimiser state carries across periods)."""

    def __init__(self, model=None, max_to_keep=1):
        self.model = model

    def save(self, sess, path):
        torch.save((self.model or sess.model).engine.state_dict(to_cpu=True), path)
        return path

    def restore(self, sess, path):
        (self.model or sess.model).engine.load_state_dict(torch.load(path, weights_only=True))
