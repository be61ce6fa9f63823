"""Wall-clock breakdown of a continual-learning run (dev tool): train steps / evaluation / exemplar selection / everything else.
usage: python tools/e2e_breakdown.py name:flag=value,flag=value ...   (flags of ader_amd/main.py without the leading --)
The sections are cut at synchronised marks (torch.cuda.synchronize at every mark): the time up to the entry of Evaluator.evaluate
is the epoch's train steps (+ feeder), the time inside it the evaluation, the time inside ExemplarGenerator's selectors the
exemplar selection (batched encode + herding + teacher logits), the rest host work between them (data loading, checkpoints)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd import main as M
from ader_amd import data as D
from ader_amd import exemplar as X

acc, last = {}, [0.0]


def mark(cat):
    torch.cuda.synchronize()
    now = time.perf_counter()
    acc[cat] = acc.get(cat, 0.0) + now - last[0]
    last[0] = now


def wrap(cls, name, before, inside):
    f = getattr(cls, name)

    def g(*a, **k):
        mark(before)
        try:
            return f(*a, **k)
        finally:
            mark(inside)
    setattr(cls, name, g)


wrap(D.Evaluator, "evaluate", "train", "eval")
M.Evaluator = D.Evaluator
for nm in ("herding_selection", "loss_selection", "randomly_selection"):
    wrap(X.ExemplarGenerator, nm, "other", "selection")

for spec in sys.argv[1:]:
    name, _, fl = spec.partition(":")
    argv = []
    for f in filter(None, fl.split(",")):
        k, _, v = f.partition("=")
        argv += ["--" + k, v]
    with tempfile.TemporaryDirectory() as d:
        acc.clear()
        torch.cuda.synchronize()
        t0 = last[0] = time.perf_counter()
        args = M.build_parser().parse_args(argv + ["--results_root", d, "--save_dir", name])
        out = M.run(args, log=lambda s="": None)
        mark("other")
        a = out["average"]
        tot = time.perf_counter() - t0
        # (the time between an evaluation and the next one is train steps, except the stretch after a period's test evaluation:
        #  state restore, data loading of the next period -- it is inside "train" of the next period's first epoch and in "other")
        print("%-24s Recall@20 %.2f MRR@20 %.2f | %.1f s: %s" % (
            name, 100 * a["recall20"], 100 * a["mrr20"], tot,
            "  ".join("%s %.1f" % (k, v) for k, v in sorted(acc.items(), key=lambda kv: -kv[1]))), flush=True)
