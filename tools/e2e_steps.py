"""Step and epoch counts of a continual-learning run beside its wall time (dev tool): python tools/e2e_steps.py name:flag=value,... ..."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd import main as M
from ader_amd import model as MD
from ader_amd import data as D

cnt = {"steps": 0, "evals": 0}
_ts = MD.Ader.train_step


def ts(self, *a, **k):
    cnt["steps"] += 1
    return _ts(self, *a, **k)


MD.Ader.train_step = ts
_tsf = MD.Ader.train_step_fed


def tsf(self, *a, **k):
    cnt["steps"] += 1
    return _tsf(self, *a, **k)


MD.Ader.train_step_fed = tsf
PLANS = {}
_ei0 = MD.Engine.__init__


def eng_init(self, *a, **k):
    _ei0(self, *a, **k)
    PLANS.clear()
    PLANS["engine"] = self


MD.Engine.__init__ = eng_init


class _P(dict):
    def __str__(self):
        e = self.get("engine")
        return "-" if e is None else "%d replayed, %d recorded, %d errors" % (e.plan_hits, e.plan_misses, len(e.plan_errors))


PLANS = _P()
_ev = D.Evaluator.evaluate


def ev(self, *a, **k):
    cnt["evals"] += 1
    return _ev(self, *a, **k)


D.Evaluator.evaluate = ev
M.Evaluator = D.Evaluator
_ei = D.Evaluator.__init__
NOCACHE = [False]


def ei(self, *a, **k):
    if NOCACHE[0]:
        D.Evaluator._packed.clear()          # (A/B: the evaluator packs its session list again every epoch, as before round 5)
    return _ei(self, *a, **k)


D.Evaluator.__init__ = ei
for spec in sys.argv[1:]:
    name, _, fl = spec.partition(":")
    NOCACHE[0] = name.endswith("_noevalcache")
    argv = []
    for f in filter(None, fl.split(",")):
        k, _, v = f.partition("=")
        argv += ["--" + k, v]
    with tempfile.TemporaryDirectory() as d:
        cnt["steps"] = cnt["evals"] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        args = M.build_parser().parse_args(argv + ["--results_root", d, "--save_dir", name])
        out = M.run(args, log=lambda s="": None)
        torch.cuda.synchronize()
        a = out["average"]
        print("%-22s Recall@20 %.2f MRR@20 %.2f | %.1f s, %d train steps, %d evaluations | plans: %s" % (
            name, 100 * a["recall20"], 100 * a["mrr20"], time.perf_counter() - t0, cnt["steps"], cnt["evals"], PLANS), flush=True)
