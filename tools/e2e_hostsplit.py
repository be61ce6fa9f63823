"""Where the HOST time of the training loop goes (dev tool): unsynchronised perf_counter accumulators around the feeder calls and
the train step of ader_amd/main.py, beside the loop's wall time.  usage: python tools/e2e_hostsplit.py [flag=value ...]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd import main as M
from ader_amd import data as D
from ader_amd import model as MD

acc, cnt = {}, {}


def wrap(cls, name, cat):
    f = getattr(cls, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[cat] = acc.get(cat, 0.0) + time.perf_counter() - t0
            cnt[cat] = cnt.get(cat, 0) + 1
    setattr(cls, name, g)


wrap(D.Sampler, "next_batch", "next_batch")
wrap(D.Sampler, "next_exemplar_batch", "next_exemplar_batch")
wrap(MD.Ader, "train_step", "train_step (host enqueue)")
wrap(D.Evaluator, "evaluate", "evaluate")
wrap(D.Sampler, "_repack", "sampler repack")
_sh = random.shuffle


def shuffle(x):
    t0 = time.perf_counter()
    _sh(x)
    acc["random.shuffle"] = acc.get("random.shuffle", 0.0) + time.perf_counter() - t0
    cnt["random.shuffle"] = cnt.get("random.shuffle", 0) + 1


random.shuffle = shuffle
argv = ["--dataset", "DIGINETICA", "--max_periods", "4", "--results_root", "/tmp/e2e_hostsplit"]
for f in sys.argv[1:]:
    k, _, v = f.partition("=")
    argv += ["--" + k, v]
args = M.build_parser().parse_args(argv)
torch.cuda.synchronize()
t0 = time.perf_counter()
M.run(args, log=lambda *a: None)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("total %.2f s" % tot)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %7.3f s  %7d calls  %8.1f us/call" % (k, v, cnt[k], v / cnt[k] * 1e6))
