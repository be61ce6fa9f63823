"""Second pin of the MODEL-MATH ORACLE against values the reference publishes (build container only; ~1.5 CPU-hours on 8 threads).

tests/golden/make_oracle_period1.py pins the oracle on the figure's period-1 points.  This script runs the ORACLE ITSELF
(oracle/ader_ref_cpu.py: torch-CPU float32, autograd, TF-Adam) through the reference's whole continual-learning loop in its simplest
configuration -- the Finetune baseline (main.py:141-146: dropout 0, no exemplars; every period restores the previous period's best
state, main.py:209-213, trains with early stopping on valid Recall@20, patience 5, main.py:271-280, restores the best epoch and
evaluates the next period's test sessions, main.py:283-292) -- on DIGINETICA, periods 1..16, with the host feeders pinned bit-exactly
by tests/golden/{sampler,split,dataloader}.*.  It writes the 16 per-period test metrics to tests/golden/oracle_finetune16.json;
tests/test_oracle_model.py asserts them against the Finetune curve of the reference's published figure (results.svg ->
tests/golden/results_svg_curves.json): the 16-period average and the per-period deviation.

    python tests/golden/make_oracle_finetune16.py [--periods 16] [--threads N]
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ader_amd.data import DataLoader, Evaluator, Sampler  # noqa: E402  (numpy-only host feeders)
from oracle import ader_ref_cpu as R  # noqa: E402
from make_oracle_period1 import ITEM_NUM, T, H, L, HEADS, OracleModel  # noqa: E402


def snapshot(params, opt):
    return ({k: v.clone() for k, v in params.items()}, {k: v.clone() for k, v in opt.m.items()},
            {k: v.clone() for k, v in opt.v.items()}, opt.b1p, opt.b2p, opt.t)


def restore(params, opt, st):
    with torch.no_grad():
        for k in params:
            params[k].copy_(st[0][k]); opt.m[k].copy_(st[1][k]); opt.v[k].copy_(st[2][k])
    opt.b1p, opt.b2p, opt.t = st[3], st[4], st[5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--periods", type=int, default=16)
    ap.add_argument("--max-epochs", type=int, default=100)
    ap.add_argument("--threads", type=int, default=8,
                    help="pinned: the record depends on the thread count (threaded float32 sums); not bitwise reproducible either "
                         "way -- re-runs land within +-0.25 Recall@20 (VERDICT r4)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    np.random.seed(0)
    random.seed(0)
    torch.manual_seed(0)                                   # main.py:123-125 (random_seed 0)
    dl = DataLoader("DIGINETICA")
    params = R.init_params(ITEM_NUM, T, H, L, seed=0)
    opt = R.TFAdam(params)
    model = OracleModel(params)
    path = os.path.join(ROOT, "tests", "golden", "oracle_finetune16.json")
    best_state, step, periods, t_all = None, 0, [], time.time()
    for period in range(1, args.periods + 1):
        t0 = time.time()
        train_sess, _ = dl.train_loader(period - 1)
        smp = Sampler(train_sess, T, 256)
        valid_subseq, _train_subseq = smp.split_data(valid_portion=0.1, return_train=True)
        batch_num = smp.batch_num()
        test_sess, _ = dl.evaluate_loader(period)
        max_item = dl.max_item()
        if period > 1:
            restore(params, opt, best_state)                # saver.restore(previous best): variables AND Adam slots / beta powers
        best, best_epoch, stop, period_best, epochs = 0.0, 1, 0, None, 0
        for epoch in range(1, args.max_epochs + 1):
            for _ in range(batch_num):
                seq, pos = smp.next_batch()
                R.train_step(params, opt, seq, pos, max_item, L, HEADS, 5e-4, training=True, rate=0.0, seed=0, step=step)
                step += 1
            epochs = epoch
            ev = Evaluator(valid_subseq, True, T, 1024, max_item, "valid", model, None)
            ev.evaluate(epoch)
            perf = ev.results()[1]
            if best >= perf:                                # main.py:271-280
                stop += 1
                if stop >= 5:
                    break
            else:
                stop, best_epoch, best = 0, epoch, perf
                period_best = best_state = snapshot(params, opt)
        if period_best is None:
            best_state = snapshot(params, opt)
        restore(params, opt, best_state)                    # main.py:283
        ev = Evaluator(test_sess, False, T, 1024, max_item, "test", model, None)
        ev.evaluate(best_epoch)
        r = ev.results()
        periods.append({"period": period, "max_item": max_item, "batch_num": batch_num, "best_epoch": best_epoch, "epochs_run": epochs,
                        "mrr20": r[0], "recall20": r[1], "mrr10": r[2], "recall10": r[3], "minutes": round((time.time() - t0) / 60.0, 1)})
        print(json.dumps(periods[-1]), flush=True)
        out = {"dataset": "DIGINETICA", "config": "Finetune baseline (dropout 0, no exemplars), random_seed 0", "periods": periods,
               "average": {k: float(np.mean([p[k] for p in periods])) for k in ("mrr20", "recall20", "mrr10", "recall10")},
               "steps": step, "torch": torch.__version__, "threads": args.threads, "minutes": round((time.time() - t_all) / 60.0, 1)}
        json.dump(out, open(path, "w"), indent=1)           # (written after every period: a partial run is still a record)
    print(json.dumps(out["average"]))


if __name__ == "__main__":
    main()
