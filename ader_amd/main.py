#!/usr/bin/env python
"""Continual-learning driver with the reference's flags and log format (reference main.py:68-336), running the
MI355X hot path.  `python -m ader_amd.main --dataset DIGINETICA [--finetune=True] ...`

Differences from the reference that do not change results: booleans are parsed with str2bool (the reference's
`type=bool` turns "--finetune=False" into True, main.py:83-91); `stop_counter` is initialised (main.py:272-273);
batches and teacher logits stay on the GPU (the reference feeds Python float lists every step, util.py:254);
the best-epoch checkpoint is kept in memory and written to disk only with --save_ckpt.
Extra flags: --logits_dtype, --max_periods, --data_root, --save_ckpt, --eval_batch (--test_batch is accepted and ignored:
ranks do not depend on the evaluation batch size), --dist_backend, --device_feed.

Data parallel: launched as `python -m torch.distributed.run --nproc-per-node W -m ader_amd.main ...` every
rank builds the same batches from the same RNG streams, trains on its slice of the train rows and of the exemplar rows
(loss terms scaled by the global sub-batch sizes, gradients exchanged by ader_amd.dist), evaluates every W-th batch and
runs herding on its chunk of label groups; ranks / selected indices are exchanged so every rank holds the reference's
metrics and exemplar set.  Rank 0 writes the logs.
"""
import argparse
import math
import os
import random
import time

import numpy as np
import torch

from .data import DataLoader, Evaluator, Sampler
from .exemplar import ExemplarGenerator
from .model import Ader, Ewc, Saver, Session


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


# The reference's command line (main.py:76-107): (flag, default, type).  Booleans go through str2bool.
_REFERENCE_FLAGS = (
    ("dataset", "DIGINETICA", str), ("save_dir", "ADER", str), ("exemplar_size", 30000, int), ("lambda_", 0.8, float),
    ("finetune", False, bool), ("dropout", False, bool), ("joint", False, bool), ("selection", "herding", str),
    ("disable_distillation", False, bool), ("equal_exemplar", False, bool), ("fix_lambda", False, bool),
    ("num_epochs", 100, int), ("batch_size", 256, int), ("test_batch", 64, int), ("device_num", 0, int), ("lr", 0.0005, float),
    ("num_blocks", 2, int), ("num_heads", 1, int), ("stop", 5, int), ("random_seed", 0, int), ("hidden_units", 150, int),
    ("maxlen", 50, int), ("dropout_rate", 0.3, float), ("l2_emb", 0.0, float),
    ("ewc", False, bool), ("ewc_sample_num", 1000, int),
)
# Flags of this build only: (flag, default, type, help / choices)
_BUILD_FLAGS = (
    ("logits_dtype", "x3", str, ("f32", "bf16", "x3")), ("max_periods", 0, int, None), ("data_root", None, str, None),
    ("results_root", "results", str, None), ("save_ckpt", False, bool, None),
    ("dist_backend", "nccl", str, "torch.distributed backend when WORLD_SIZE > 1"),
    ("device_feed", True, bool, "keep the packed training rows on the GPU and gather batches there"),
    ("eval_batch", 1024, int, "rows per evaluation launch (results do not depend on it)"),
    ("pack_sessions", "auto", str, ("auto", "on", "off")),
    ("dp_mode", "auto", str, ("auto", "replicated", "catalog")),
    ("fed_steps", True, bool, "single GPU with --device_feed and --fixed_batches: the batch is cut on the GPU by the step's first launch "
                              "and the step is one native call (Engine.train_step_fed); False: batches assembled by torch, as data-parallel runs do"),
    ("fixed_batches", True, bool, "pad every train / exemplar batch to its nominal row count with weight-0 rows: the feeder drops "
                                  "invalid sub-sequences, so the row count wanders by a few rows from step to step (251..256 on "
                                  "DIGINETICA) and every new count re-allocates and clears the engine's ~45 activation buffers"),
)


def build_parser():
    p = argparse.ArgumentParser()
    _help = {"test_batch": "accepted for compatibility with the reference's command lines (README.md:77); evaluation launches take "
                           "--eval_batch rows -- the ranks, hence the metrics, do not depend on the evaluation batch size"}
    for name, default, typ in _REFERENCE_FLAGS:
        p.add_argument("--" + name, default=default, type=str2bool if typ is bool else typ, help=_help.get(name))
    for name, default, typ, extra in _BUILD_FLAGS:
        kw = {"default": default, "type": str2bool if typ is bool else typ}
        if isinstance(extra, tuple):
            kw["choices"] = extra
        elif extra:
            kw["help"] = extra
        p.add_argument("--" + name, **kw)
    return p


ITEM_NUM = {'DIGINETICA': 43136, 'YOOCHOOSE': 25958}     # main.py:133-138


class _NullLog:
    def write(self, *_):
        pass

    flush = close = write


def run(args, log=print):
    from . import dist as adist
    rank, world, local = adist.init(getattr(args, "dist_backend", "nccl"))
    shard = (rank, world)
    if rank != 0:
        log = lambda *_: None      # noqa: E731
    out_dir = os.path.join(args.results_root, args.dataset + '-' + args.save_dir)
    os.makedirs(out_dir, exist_ok=True)
    logs = open(os.path.join(out_dir, 'Training_logs.txt'), mode='w') if rank == 0 else _NullLog()
    logs.write('\n'.join([str(k) + ',' + str(v) for k, v in sorted(vars(args).items(), key=lambda x: x[0])]))
    np.random.seed(args.random_seed)
    random.seed(args.random_seed)
    torch.manual_seed(args.random_seed)
    if args.dataset not in ITEM_NUM:
        raise ValueError('Invalid dataset name')
    item_num = ITEM_NUM[args.dataset]
    args.dropout_rate = 0 if (args.ewc or args.finetune) else args.dropout_rate      # main.py:141
    dev_index = (local % max(torch.cuda.device_count(), 1)) if world > 1 else args.device_num
    model = (Ewc if args.ewc else Ader)(item_num, args, device="cuda:%d" % dev_index, dp_rank=rank, dp_world=world)   # main.py:144
    model.engine.warm_up()          # the side-stream probe (a scratch tensor, eight streams, ~16 synchronisations): here, not inside the first step
    dp = adist.DataParallel(model.engine, rank, world)
    if world > 1:
        # data-parallel scheme: "replicated" = every rank holds the table, dense gradient all-reduce (overlapped
        # with backward when the table outweighs the per-position rows, dist.DataParallel.early_pays); "catalog" = every rank owns 1/W
        # of the rows.  auto: by catalog size -- the shipped datasets (26-43 k items: a 15-26 MB table) stay replicated: ONE all-reduce
        # per step against the catalog scheme's ~10 small collectives on a 0.4 ms step
        model.engine.dp_mode = ("catalog" if item_num >= 200_000 else "replicated") if args.dp_mode == "auto" else args.dp_mode
    # the cyclic collector walks everything torch imported (~40 ms per full pass, tens of times per period once the loop below
    # churns Python lists): freeze what exists now, so later passes see only the loop's own objects
    import gc
    gc.collect()
    gc.freeze()
    baseline = args.finetune or args.dropout or args.joint
    dataloader = DataLoader(args.dataset, root=args.data_root)
    n_periods = dataloader.num_periods() - 1
    periods = range(1, n_periods + 1)
    if args.max_periods:
        periods = range(1, min(n_periods, args.max_periods) + 1)
    log('Continue Learning: number of periods is %d.' % len(periods))
    logs.write('Continue Learning: number of periods is %d.\n' % len(periods))
    item_num_prev = 0
    t_start = time.time()
    MRR_20, Recall_20, MRR_10, Recall_10 = [], [], [], []
    best_state, store = None, None
    saver = Saver(model)
    summary = []
    for period in periods:
        log('Period %d:' % period)
        logs.write('Period %d:\n' % period)
        best_performance, performance = 0, 0
        Evaluator.clear_cache()
        train_sess, info = dataloader.train_loader(period - 1)
        logs.write(info + '\n')
        if args.joint and period > 1:
            for p in range(1, period):
                pre, info = dataloader.train_loader(p - 1)
                logs.write(info + '\n')
                train_sess.extend(pre)
        train_sampler = Sampler(train_sess, args.maxlen, args.batch_size)
        valid_subseq, train_subseq = train_sampler.split_data(valid_portion=0.1, return_train=True)
        if args.device_feed:
            train_sampler.to_device(model.engine.device)
            model.engine.pack_density = train_sampler.density        # device batches: the feeder announces how sparse they are
        batch_num = train_sampler.batch_num()
        test_sess, info = dataloader.evaluate_loader(period)
        logs.write(info + '\n')
        max_item = dataloader.max_item()
        use_ex = period > 1 and not baseline
        if use_ex:
            # (the reference flattens {item: [[session, logits], ...]} into a list, main.py:54-65,176-190; the store holds the same
            #  exemplars in the same order as packed rows + one logits tensor, and the Sampler takes them as they are)
            exemplar_size = len(store)
            exemplar_subseq = store.sessions()
            exemplar_batch = int(exemplar_size / batch_num)             # main.py:187
            exemplar_sampler = Sampler([], args.maxlen, exemplar_batch)
            exemplar_sampler.add_exemplar(store)
            if args.device_feed:
                exemplar_sampler.to_device(model.engine.device)
            if args.ewc or args.fix_lambda:                              # main.py:196
                lambda_ = args.lambda_
            else:                                                        # main.py:200
                lambda_ = args.lambda_ * math.sqrt((item_num_prev / max_item) * (exemplar_size / train_sampler.data_size()))
            model.update_loss(lambda_=lambda_)
        else:
            exemplar_subseq = []
            model.set_vanilla_loss()
        with Session(model) as sess:
            if period > 1 and not args.joint:
                model.engine.load_state_dict(best_state)                # saver.restore(prev best), main.py:211
            else:
                model.engine.init_params(args.random_seed)              # global_variables_initializer, main.py:213
            best_epoch, stop_counter, period_best = 1, 0, None
            for epoch in range(1, args.num_epochs + 1):
                # device-fed steps: the batch (train rows, padding, exemplar rows) is cut on the GPU from the Samplers' resident rows by
                # the step's first launch and the whole step is one native call (Engine.train_step_fed) -- same batches, same padding,
                # same dropout counters as the branch below, which stays for data-parallel runs, the EWC baseline and host-fed batches
                fed = bool(args.fed_steps and args.device_feed and world == 1 and args.fixed_batches and not args.ewc)
                for _ in range(batch_num if fed else 0):
                    idx_t, o_t, n_t = train_sampler.next_index_slice()
                    if use_ex:
                        idx_e, o_e, n_e = exemplar_sampler.next_index_slice()
                        feed = (train_sampler.rows_dev(), idx_t, o_t, n_t, train_sampler.batch_size,
                                exemplar_sampler.rows_dev(), idx_e, o_e, n_e, exemplar_sampler.batch_size)
                    else:
                        feed = (train_sampler.rows_dev(), idx_t, o_t, n_t, train_sampler.batch_size, None, None, 0, 0, 0)
                    model.train_step_fed(feed, max_item, args.lr, args.dropout_rate, teacher=store.logits if use_ex else None)
                for _ in range(0 if fed else batch_num):
                    seq, pos = train_sampler.next_batch()
                    kw = {}
                    if world > 1:
                        # this rank's train rows, padded to ceil(n / W) rows with id-0 sessions of label 0 (weight 0 in the loss):
                        # every rank issues collectives of identical sizes, and no rank ever has an empty batch
                        lo, _ = adist.shard_bounds(len(pos), world, rank)
                        seq_t, pos_t = adist.shard_rows(seq, world, rank), adist.shard_rows(pos, world, rank)
                        kw.update(n_train_global=len(pos))
                        dp.set_rows(lo, max_item)
                    else:
                        seq_t, pos_t = seq, pos
                        if args.fixed_batches and 0 < len(pos) < train_sampler.batch_size:
                            # same step, same loss (the padding rows have weight 0 and the means divide by the real count), one shape
                            seq_t, pos_t = adist.pad_rows(seq, train_sampler.batch_size), adist.pad_rows(pos, train_sampler.batch_size)
                            kw.update(n_train_global=len(pos))
                        dp.set_rows(0, max_item)
                    if use_ex and not args.ewc:                                   # main.py:225
                        ex_seq, ex_pos, idx = exemplar_sampler.next_exemplar_batch()
                        idx = np.asarray(idx, dtype=np.int32)
                        idx_dev = getattr(exemplar_sampler, "last_idx_dev", None) if world == 1 else None   # (same indices, already on the device)
                        if world == 1 and args.fixed_batches and (len(pos_t) > len(pos) or 0 < len(ex_seq) < exemplar_sampler.batch_size):
                            kw.update(n_ex_global=len(ex_seq))
                            dp.set_rows(0, max_item, ex_row0=len(pos))             # the exemplar rows keep the dropout counters of the unpadded batch
                            if 0 < len(ex_seq) < exemplar_sampler.batch_size:
                                ex_seq, ex_pos = (adist.pad_rows(ex_seq, exemplar_sampler.batch_size),
                                                  adist.pad_rows(ex_pos, exemplar_sampler.batch_size))
                                idx = adist.pad_rows(idx, exemplar_sampler.batch_size, fill=-1)
                                idx_dev = None
                        if world > 1:                                            # ... and exemplar rows (main.py:229 order kept)
                            kw.update(n_ex_global=len(ex_seq))
                            elo, _ = adist.shard_bounds(len(ex_seq), world, rank)
                            dp.set_rows(lo, max_item, ex_row0=len(pos) + elo)      # dropout counters keyed by the global row
                            ex_seq, ex_pos = adist.shard_rows(ex_seq, world, rank), adist.shard_rows(ex_pos, world, rank)
                            idx = adist.shard_rows(idx, world, rank, fill=-1)
                        if len(ex_seq):
                            cat = torch.cat if isinstance(seq_t, torch.Tensor) else np.concatenate
                            seq_l = cat([seq_t, ex_seq])
                        else:
                            seq_l = seq_t
                        if args.disable_distillation:
                            model.train_step(seq_l, pos_t, max_item, args.lr, args.dropout_rate, ex_pos=ex_pos, **kw)
                        else:
                            model.train_step(seq_l, pos_t, max_item, args.lr, args.dropout_rate, teacher=store.logits,
                                             ex_trow=idx if idx_dev is None else idx_dev, **kw)
                    else:
                        model.train_step(seq_t, pos_t, max_item, args.lr, args.dropout_rate, **kw)
                model.engine.check_status()
                if use_ex and args.ewc:                                           # main.py:258-262
                    # The reference re-takes the snapshot and the Fisher information after every epoch, but its loss tensor was built
                    # from the arrays of the moment update_loss() was called (EWC.py:115-124), so these per-epoch values never reach
                    # the running period and are overwritten at its end: only their consumption of the `random` stream is kept.
                    random_exemplar = random.sample(exemplar_subseq, min(len(exemplar_subseq), args.ewc_sample_num))
                    model.compute_fisher(sess, random_exemplar, 50, max_item, dry_run=True)
                valid_evaluator = Evaluator(valid_subseq, True, args.maxlen, args.eval_batch, max_item, 'valid', model, sess, shard)
                info = valid_evaluator.evaluate(epoch)
                logs.write(info + '\n')
                performance = valid_evaluator.results()[1]
                if best_performance >= performance:                      # early stop, main.py:271-280
                    stop_counter += 1
                    if stop_counter >= args.stop:
                        break
                else:
                    stop_counter = 0
                    best_epoch = epoch
                    best_performance = performance
                    period_best = best_state = model.engine.state_dict()
                    if args.save_ckpt and rank == 0:
                        d = os.path.join(out_dir, 'model', 'period%d' % period)
                        os.makedirs(d, exist_ok=True)
                        saver.save(sess, os.path.join(d, 'epoch=%d.ckpt' % epoch))
            if period_best is None:                                      # no epoch improved on 0: keep the last state
                best_state = model.engine.state_dict()
            model.engine.load_state_dict(best_state)                    # saver.restore(best), main.py:283
            test_evaluator = Evaluator(test_sess, False, args.maxlen, args.eval_batch, max_item, 'test', model, sess, shard)
            info = test_evaluator.evaluate(best_epoch)
            logs.write(info + '\n')
            r = test_evaluator.results()
            MRR_20.append(r[0]); Recall_20.append(r[1]); MRR_10.append(r[2]); Recall_10.append(r[3])
            summary.append({"period": period, "best_epoch": best_epoch, "mrr20": r[0], "recall20": r[1], "mrr10": r[2],
                            "recall10": r[3], "max_item": max_item})
            if not baseline:                                             # exemplar selection, main.py:294-313
                exemplar_candidate = train_subseq
                exemplar_candidate.extend(valid_subseq)
                exemplar_candidate.extend(exemplar_subseq)
                exemplar = ExemplarGenerator(exemplar_candidate, args.exemplar_size, args.equal_exemplar, args.batch_size,
                                             args.maxlen, args.dropout_rate, max_item, shard)
                if args.selection == 'herding':
                    saved_num = exemplar.herding_selection(sess, model)
                elif args.selection == 'loss':
                    saved_num = exemplar.loss_selection(sess, model)
                elif args.selection == 'loss_ref':                           # the reference's executed behaviour (util.py:488)
                    saved_num = exemplar.loss_selection(sess, model, first_only=True)
                elif args.selection == 'random':
                    saved_num = exemplar.randomly_selection(sess, model)
                else:
                    raise ValueError("Invalid exemplar selection method")
                info = 'Total saved exemplar: %d' % saved_num
                log(info)
                logs.write(info + '\n')
                store = exemplar.store                                   # (exemplar.exemplars: the reference's {item: [[session, logits]]} view, on demand)
                if args.save_ckpt and rank == 0:                         # (the reference keeps exemplars in memory only)
                    d = os.path.join(out_dir, 'model', 'period%d' % period)
                    os.makedirs(d, exist_ok=True)
                    store.save(os.path.join(d, 'exemplars.pt'))
                del exemplar
            item_num_prev = max_item
            if args.ewc:                                                 # main.py:319-323: Fisher information for the next period
                exemplar_subseq = store.sessions()
                model.snapshot_variables()
                random_exemplar = random.sample(exemplar_subseq, min(len(exemplar_subseq), args.ewc_sample_num))
                model.compute_fisher(sess, random_exemplar, 50, max_item)
        logs.flush()
    res = (np.array(MRR_20).mean(), np.array(Recall_20).mean(), np.array(MRR_10).mean(), np.array(Recall_10).mean())
    info = 'Average: (MRR@20: %.4f, RECALL@20: %.4f, MRR@10: %.4f, RECALL@10: %.4f)' % res
    log(info)
    logs.write(info + '\n')
    log('Total time: %.2f minutes.' % ((time.time() - t_start) / 60.0))
    logs.write('Total time: %.2f minutes\nDone.' % ((time.time() - t_start) / 60.0))
    logs.close()
    log('Done.')
    return {"average": dict(zip(("mrr20", "recall20", "mrr10", "recall10"), map(float, res))), "periods": summary}


def main(argv=None):
    args = build_parser().parse_args(argv)
    return run(args)


if __name__ == '__main__':
    main()
