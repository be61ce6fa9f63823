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
wrap(MD.Ader, "train_step_fed", "train_step_fed (host enqueue)")
wrap(D.Sampler, "next_index_slice", "next_index_slice")
wrap(D.Sampler, "to_device", "Sampler.to_device")
wrap(D.Sampler, "add_exemplar", "Sampler.add_exemplar")
wrap(D.Evaluator, "evaluate", "evaluate")
wrap(D.Sampler, "_repack", "sampler repack")
wrap(D.Sampler, "__init__", "Sampler.__init__ (incl. repack)")
wrap(D.Sampler, "split_data", "split_data")
wrap(D.DataLoader, "train_loader", "DataLoader.train_loader")
wrap(D.DataLoader, "evaluate_loader", "DataLoader.evaluate_loader")
wrap(D.Evaluator, "__init__", "Evaluator.__init__")
from ader_amd import exemplar as X
from ader_amd import engine as EN
wrap(X.ExemplarGenerator, "__init__", "ExemplarGenerator.__init__ (group_by_label, quotas)")
for nm in ("herding_selection", "loss_selection", "randomly_selection"):
    wrap(X.ExemplarGenerator, nm, "exemplar selection (encode + herding + teacher logits)")
wrap(EN.Engine, "load_state_dict", "Engine.load_state_dict")
wrap(EN.Engine, "state_dict", "Engine.state_dict")
wrap(EN.Engine, "init_params", "Engine.init_params")
wrap(EN.Engine, "check_status", "Engine.check_status (sync)")
wrap(EN.Engine, "__init__", "Engine.__init__")
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
