#!/usr/bin/env python
"""Benchmark of the ADER / SASRec training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic input already resident in HBM: embedding gather ->
causal self-attention blocks -> full-catalog logits + softmax CE -> backward -> [RCCL gradient all-reduce] -> dense
Adam (the `sess.run(train_op)` of reference main.py:233-256).  Workload = BASELINE.json configs[4]: synthetic 1M-item
catalog, seq_len 50, batch 512 per GPU (weak scaling), dense regime (every position a real item, ids ~ U[1,N]),
dropout 0.3, lr 5e-4, hidden 150, 2 blocks, 1 head (reference defaults main.py:98-107).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on
the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle timed on this host's cores).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0


def synth_batch(B, T, N, seed, device, regime="dense"):
    """dense: every row full length, ids ~ U[1,N] (BASELINE.json configs[4], the headline).  realistic: session
    length ~ 1 + Geometric(0.2) clipped to [1,T] (mean ~5, left-padded with 0), ids ~ Zipf(1.05) over a random permutation
    of [1,N]; labels follow the same id law."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if regime == "dense":
        seq = torch.randint(1, N + 1, (B, T), generator=g, dtype=torch.int32)
        pos = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32)
        return seq.to(device), pos.to(device)
    rs = np.random.RandomState(seed)
    perm = np.random.RandomState(12345).permutation(N).astype(np.int64) + 1
    w = 1.0 / np.arange(1, N + 1, dtype=np.float64) ** 1.05
    cdf = np.cumsum(w / w.sum())

    def ids(n):
        return perm[np.minimum(np.searchsorted(cdf, rs.rand(n)), N - 1)]
    seq = np.zeros((B, T), dtype=np.int32)
    ln = np.clip(rs.geometric(0.2, size=B), 1, T)
    for b in range(B):
        seq[b, T - ln[b]:] = ids(ln[b])
    pos = ids(B).astype(np.int32)
    return torch.from_numpy(seq).to(device), torch.from_numpy(pos).to(device)


def cpu_baseline(N, B, T, H, L, heads, rate, lr, E=0, Np=0):
    """Reference-equivalent CPU step (oracle/ader_ref_cpu.py: materialised [B,N] logits, one-hot CE [+ distillation against E
    teacher rows], autograd, dense TF-style Adam) timed on this host's cores.  Bounded sample: full steps of the same workload
    after a tiny warm-up step (thread pool / allocator) -- ONE step at the 1M-item catalog (~1 minute), up to 20 steps / ~20 s at
    the small catalogs; if the host has little memory the row count is reduced and stated."""
    import psutil
    from oracle import ader_ref_cpu as R
    cores = os.cpu_count() or 1
    if N < 200_000:
        cores = min(cores, 32)      # small catalogs: more threads only add synchronisation (256 threads: 49 s per step, measured)
    torch.set_num_threads(cores)
    avail = psutil.virtual_memory().available / 2 ** 30
    Bs = B if (avail > 48 or N < 200_000) else max(32, B // 4)
    params = R.init_params(N, T, H, L, seed=0)
    opt = R.TFAdam(params)
    rs = np.random.RandomState(0)
    seq = rs.randint(1, N + 1, size=(Bs + E, T)).astype(np.int64)
    pos = rs.randint(1, N + 1, size=Bs).astype(np.int64)
    kd = dict(ex_logits=torch.randn(E, Np, generator=torch.Generator().manual_seed(7)), lambda_=0.8) if E else {}
    kd4 = dict(ex_logits=kd["ex_logits"][:2], lambda_=0.8) if E else {}
    R.train_step(params, opt, seq[list(range(4)) + list(range(Bs, Bs + (2 if E else 0)))], pos[:4], N, L, heads, lr, training=True,
                 rate=rate, seed=0, step=0, **kd4)                                                   # warm-up (a few rows)
    n, t0 = 0, time.perf_counter()
    while True:
        R.train_step(params, opt, seq, pos, N, L, heads, lr, training=True, rate=rate, seed=0, step=1 + n, **kd)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 20.0 or n >= 20:
            break
    out = {"value": Bs * n / dt, "unit": "sessions/s", "cores": cores, "kind": "port",
           "sample": "%d step(s) of B=%d train%s rows x T=%d at the full N=%d catalog (%.1f s), torch-CPU float32 restatement, %d threads"
                     % (n, Bs, " + %d distilled" % E if E else "", T, N, dt, cores)}
    # single-thread figure on a smaller sample of the same workload: 16 rows at the full catalog
    try:
        torch.set_num_threads(1)
        B1 = 16
        t0 = time.perf_counter()
        R.train_step(params, opt, seq[:B1], pos[:B1], N, L, heads, lr, training=True, rate=rate, seed=0, step=2)
        dt1 = time.perf_counter() - t0
        out["value_1thread"] = B1 / dt1
        out["sample_1thread"] = "1 step of B=%d rows at N=%d (%.1f s), 1 thread" % (B1, N, dt1)
    finally:
        torch.set_num_threads(cores)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--logits", choices=["bf16", "x3", "f32"], default="bf16",
                    help="operand type of the logit GEMMs (fp32 master table, fp32 accumulate/softmax either way)")
    ap.add_argument("--regime", choices=["dense", "realistic"], default="dense",
                    help="synthetic id/length law (SURVEY 8d); the headline number is the dense regime")
    ap.add_argument("--exemplars", type=int, default=0,
                    help="ADER-mode variant: append this many exemplar rows distilled against N(0,1) teacher logits over 0.9 N items")
    ap.add_argument("--dp-mode", choices=["replicated", "catalog"], default="replicated",
                    help="N > 1: 'replicated' (headline; the north-star scheme: every rank holds the table, the table update is "
                         "row-sharded reduce-scatter / all-gather style and the updated rows are all-gathered over RCCL) or 'catalog' "
                         "(named variant: each rank OWNS 1/N of the table rows, only touched rows travel)")
    ap.add_argument("--workload", choices=["cfgS", "cfgD", "cfgY", "cfgF"], default="cfgS",
                    help="cfgS: BASELINE configs[4] (the metric's configuration).  Step-shape variants of the real-data configs "
                         "(SURVEY 8a): cfgD = DIGINETICA ADER last period (N 43,105, 256 train + 143 distilled rows), cfgY = YOOCHOOSE "
                         "ADER last period (N 25,750, 512 + 102 rows), cfgF = DIGINETICA finetune baseline (BASELINE configs[0]: N 43,105, "
                         "batch 128, no exemplars, dropout 0); synthetic ids of those shapes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32grade", action="store_true", help="skip the float32-grade (x3 logits) companion run")
    ap.add_argument("--pmc-json", default=None,
                    help="rocprofv3 PMC summary of THIS command (tools/summarize_profiles.py) to quote roofline.traffic from; "
                         "without it traffic is null (bench.py never pairs live timings with counters of another run)")
    ap.add_argument("--no-sections", action="store_true", help="do not record per-kernel HIP events in the timed region")
    args = ap.parse_args()

    from ader_amd import dist as adist
    from ader_amd.engine import Engine, SectionTimer
    import torch.distributed as dist

    # ADER_DIST_BACKEND=gloo lets several ranks share one GPU (functional check of the N > 1 path without an 8-GPU node)
    rank, world, local = adist.init(os.environ.get("ADER_DIST_BACKEND", "nccl"))
    local = local % max(torch.cuda.device_count(), 1)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.workload == "cfgD":
        args.items, args.batch, args.exemplars = 43105, 256, 143
    elif args.workload == "cfgY":
        args.items, args.batch, args.exemplars = 25750, 512, 102
    elif args.workload == "cfgF":
        args.items, args.batch, args.exemplars = 43105, 128, 0
    N, B, T, H, L, heads, rate, lr = args.items, args.batch, 50, 150, 2, 1, (0.0 if args.workload == "cfgF" else 0.3), 5e-4
    eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=L, num_heads=heads, seed=0, device=dev, logits_dtype=args.logits,
                 dp_rank=rank, dp_world=world)
    dp = adist.DataParallel(eng, rank, world)
    if world > 1 and args.logits == "bf16" and not args.exemplars:
        eng.dp_mode = args.dp_mode
    dp.set_rows(rank * B, N)
    nbatch = 4
    E = args.exemplars
    batches = [synth_batch(B + E, T, N, 1000 * s + rank, dev, args.regime) for s in range(nbatch)]   # resident in HBM before timing
    kw = dict(rate=rate, n_train_global=B * world)
    if E:
        Np = int(0.9 * N)
        teacher = torch.randn(E, Np, generator=torch.Generator().manual_seed(7)).to(dev)     # resident teacher logits [E,Np]
        kw.update(teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8, n_ex_global=E * world)
        batches = [(sq, ps[:B]) for sq, ps in batches]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-section HIP events (on the launch stream) cost stream time: the full breakdown is taken on the last warmup steps,
    # and the timed region records only the dominant section (every 8th step), whose duration feeds the roofline line.
    sections_all = {}
    for i in range(5):        # engine initialisation (workspace allocation, kernel attributes, side streams): never timed
        seq, pos = batches[i % nbatch]
        eng.train_step(seq, pos, N, lr, **kw)
    for i in range(args.warmup):
        if not args.no_sections and i == max(0, args.warmup - 3):
            eng.timer = SectionTimer()
        seq, pos = batches[i % nbatch]
        eng.train_step(seq, pos, N, lr, **kw)
    eng.check_status()
    if not args.no_sections:
        if eng.timer is not None:
            sections_all = eng.timer.collect()
        skip = ("grad_exchange", "param_allgather")
        dom_names = [k for k in sections_all if k not in skip]
        # (a timing event pair around the kernel breaks its overlap with the side stream and costs ~60 us of the step: the
        #  dominant kernel is therefore timed on every 8th step of the timed region)
        eng.timer = SectionTimer(only={max(dom_names, key=lambda k: sections_all[k])} if dom_names else None, every=8)
    # The interpreter's cyclic collector walks every object torch has imported (~40 ms per full pass): a pass that lands in the
    # timed region is host noise of the same size as the region.  Collect now and freeze what exists (the steps themselves create
    # no cycles); a training loop does the same once at start-up (ader_amd/main.py).
    gc.collect()
    gc.freeze()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        seq, pos = batches[i % nbatch]
        eng.train_step(seq, pos, N, lr, **kw)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(eng.loss.item())
    sections = eng.timer.collect() if eng.timer is not None else {}
    eng.timer = None
    sections = {**sections_all, **sections}          # dominant section: timed-region average; the rest: warmup steps

    # ---- standalone embedding gather (north_star: "rocprof HBM GB/s on the gather"): in the step it is fused into the one-launch
    # forward, so it is timed here as its own kernel on the same batch (ader_embed_fwd: ids -> x0 = drop(E[ids]*sqrt(H) + P) * mask)
    gather = None
    if rank == 0 and not args.no_sections:
        from ader_amd._lib import call, ptr
        seq0 = batches[0][0][:B].contiguous()
        x0 = torch.empty(B * T, H, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import ctypes
        from ader_amd._lib import AderDrop
        dd = AderDrop(0x1234, int(0.3 * 2 ** 24), 1.0 / 0.7, 0, 0xFFFFFFFF, 0)
        for it in range(25):
            if it == 5:
                a.record()
            call("ader_embed_fwd", ptr(seq0), eng._pp["emb"], eng._pp["pos"], ptr(x0), B * T, T, H, eng.V, ctypes.byref(dd),
                 ptr(eng.status), st)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        gbytes = B * T * (4 + 2 * H * 4) / 1e9                    # ids + one table row read + one row written
        gather = {"kernel": "k_embed_fwd (standalone)", "us": round(us, 2), "bytes": B * T * (4 + 2 * H * 4),
                  "GBps": round(gbytes / (us * 1e-6), 1), "frac_hbm": round(gbytes / (us * 1e-6) / HBM_PEAK_GBS, 4),
                  "note": "25,600 random 600-B rows of a 600 MB table + 15.4 MB written: latency / launch bound at this size"}

    # ---- float32-grade companion (the reference's arithmetic is fp32, ADER.py:91-93): the same step with logits_dtype="x3"
    f32g = None
    if world == 1 and args.logits == "bf16" and not args.no_f32grade and not E:
        del eng
        torch.cuda.empty_cache()
        eng3 = Engine(N, maxlen=T, hidden_units=H, num_blocks=L, num_heads=heads, seed=0, device=dev, logits_dtype="x3")
        for i in range(5 + args.warmup):
            seq, pos = batches[i % nbatch]
            eng3.train_step(seq, pos, N, lr, **kw)
        eng3.timer = SectionTimer(only={"logits_bwd_adam", "logits_fwd"}, every=8)
        gc.collect()
        gc.freeze()
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            seq, pos = batches[i % nbatch]
            eng3.train_step(seq, pos, N, lr, **kw)
        sync()
        dt3 = time.perf_counter() - t0
        sec3 = eng3.timer.collect()
        eng3.timer = None
        f32g = {"ms_per_step": dt3 / args.steps * 1e3, "value": B * args.steps / dt3, "final_loss": float(eng3.loss.item()),
                "sections_ms": {k: round(v, 4) for k, v in sorted(sec3.items())}}
        eng = eng3

    if rank == 0:
        ms = dt / args.steps * 1e3
        P = eng.P
        lpeak = BF16_MFMA_PEAK_TFLOPS if args.logits in ("bf16", "x3") else F32_MFMA_PEAK_TFLOPS
        # algorithmic work per launch (DESIGN.md "roofline accounting").  In bf16 mode the forward launch also
        # produces dRep (flash-style readout), so it is credited both GEMMs; recomputation is never credited.
        fwd_flops = 2.0 * B * N * H * (2 if args.logits in ("bf16", "x3") else 1)
        work = {
            "logits_fwd": ("mfma", fwd_flops, lpeak),
            "logits_bwd_drep": ("mfma", 2.0 * B * N * H, lpeak),
            "logits_bwd_demb": ("mfma", 2.0 * B * N * H, lpeak),
            "blocks_fwd": ("mfma", L * 2.0 * B * T * (5 * H * H + 2 * T * H), F32_MFMA_PEAK_TFLOPS),
            "blocks_bwd": ("mfma", 2 * L * 2.0 * B * T * (5 * H * H + 2 * T * H), F32_MFMA_PEAK_TFLOPS),
            "adam": ("hbm", 7.0 * P * 4, HBM_PEAK_GBS),
            # fused table update: theta/m/v of rows 1..N in and out (+ bf16 shadow row in and out in bf16 mode; x3 mode has no shadow
            # and counts ONE theta read although the kernel reads theta twice); dE never hits memory
            # (catalog-sharded N > 1: a rank updates only its N / world rows)
            "logits_bwd_adam": ("hbm", (6.0 * N * H * 4 + (2.0 * N * 336 if args.logits == "bf16" else 0.0)) / (world if eng.dp_mode == "catalog" else 1), HBM_PEAK_GBS),
            # distilled exemplar rows on the exact-f32 kernels: logits + dRep + dE over the 0.9 N teacher columns
            "kd_rows": ("mfma", 3 * 2.0 * E * int(0.9 * N) * H, F32_MFMA_PEAK_TFLOPS),
            "grad_exchange": ("hbm", 0.0, HBM_PEAK_GBS),
            "param_allgather": ("hbm", 0.0, HBM_PEAK_GBS),
        }
        # HBM traffic of the dominant kernel: only from a PMC summary of THIS command passed with --pmc-json (rocprofv3 separate
        # --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, gfx950 x2 read correction applied: tools/profile_round.sh); otherwise null
        pmc_kernel = {"logits_bwd_adam": "k_tab16<" if args.logits == "bf16" else "k_tab_upd<true, true",
                      "logits_fwd": "k_lbf_fwd" if args.logits == "bf16" else "k_lx3_fwd", "adam": "k_adam"}
        pmc, pmc_src, prof_us = {}, None, None
        pmc_path = args.pmc_json
        if pmc_path is None and N == 1_000_000 and B == 512 and not E and world == 1 and args.regime == "dense":
            # default: the committed end-of-round profile of THIS command (profiles/CURRENT.json names it and the commit it was
            # taken at); quoted with its provenance, never silently
            try:
                cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
                tag = cur["x3" if args.logits == "x3" else "bf16"]
                pmc_path = os.path.join(ROOT, "profiles", tag + "_pmc_hbm.json")
                pmc_src = {"file": "profiles/%s_pmc_hbm.json" % tag, "kernel_stats": "profiles/%s_kernel_stats.csv" % tag,
                           "profiled_at_commit": cur.get("git_head"),
                           "note": "rocprofv3 passes of this same command, run separately (tools/profile_round.sh)"}
            except Exception:
                pmc_path, pmc_src = None, None
        elif pmc_path:
            pmc_src = {"file": os.path.relpath(os.path.abspath(pmc_path), ROOT)}
        if pmc_path:
            try:
                pmc = json.load(open(pmc_path))["kernels"]
            except Exception:
                pmc, pmc_src = {}, None
        roof = None
        if sections:
            if "logits_bwd_adam" in sections:     # the small-parameter Adam launch is not the 7*P*4-byte kernel any more
                work["adam"] = ("hbm", 7.0 * (P - eng.layout["pos"][0]) * 4, HBM_PEAK_GBS)
            dom = max((k for k in sections if k not in ("grad_exchange", "param_allgather")), key=lambda k: sections[k])
            bound, amount, peak = work[dom]
            sec = sections[dom] * 1e-3
            if bound == "mfma":
                ach, unit = amount / sec / 1e12, "TFLOP/s"
            else:
                ach, unit = amount / sec / 1e9, "GB/s"
            roof = {"kernel": dom, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                    "traffic": (next((v.get("hbm_bytes") for k_, v in pmc.items()
                                      if pmc_kernel.get(dom) and k_.startswith(pmc_kernel[dom])), None)
                                if unit == "GB/s" else None),
                    "traffic_source": pmc_src,
                    "frac_rocprof": None,
                    "ms": sections[dom],
                    # whole step against SURVEY 8(d)'s compulsory traffic (6.64 GB at cfg-S, unfused accounting) and the HBM peak
                    "step_bytes": 6.64e9 * (N / 1e6) if (B == 512 and not E) else None,
                    "step_frac": (6.64e9 * (N / 1e6) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (B == 512 and not E and world == 1) else None,
                    "gather": gather,
                    "sections_ms": {k: round(v, 4) for k, v in sorted(sections.items())}}
        if roof and pmc_src and pmc_src.get("kernel_stats") and roof["unit"] == "GB/s":
            try:
                import csv
                want = pmc_kernel.get(roof["kernel"])
                for r_ in csv.DictReader(open(os.path.join(ROOT, pmc_src["kernel_stats"]))):
                    nm = r_["Name"].replace("void ", "")
                    if want and nm.startswith(want):
                        us = float(r_["AverageNs"]) / 1e3
                        roof["ms_rocprof"] = us / 1e3
                        roof["frac_rocprof"] = work[roof["kernel"]][1] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                        break
            except Exception:
                pass
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N = 1 leg only
            try:
                cpu = cpu_baseline(N, B, T, H, L, heads, rate, lr, E, int(0.9 * N) if E else 0)
            except Exception as e:  # report, never fake
                cpu = {"value": None, "unit": "sessions/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        out = {
            "metric": "train sessions/sec at batch=512 seq=50", "value": B * world * args.steps / dt, "unit": "sessions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "x3": "bf16x3", "f32": "f32"}[args.logits], "data": "synthetic",
            "config": {"workload": ("synthetic %s-item catalog, seq_len=50, batch=%d/GPU, %s regime%s (BASELINE.json configs[4]%s)"
                                    % ("1M" if N == 1_000_000 else "%d" % N, B, args.regime,
                                       ", +%d distilled exemplar rows" % E if E else "",
                                       "" if (N == 1_000_000 and B == 512) else ", REDUCED SIZE")) if args.workload == "cfgS" else
                                   ("step shape of %s: N=%d items, %d train + %d distilled rows, synthetic ids (%s regime)"
                                    % ({"cfgD": "DIGINETICA ADER (BASELINE.json configs[1])", "cfgY": "YOOCHOOSE ADER (configs[2])",
                                        "cfgF": "DIGINETICA finetune baseline (configs[0], main.py --finetune=True)"}
                                       [args.workload], N, B, E, args.regime)),
                       "items": N, "batch_per_gpu": B, "global_batch": B * world, "seq_len": T, "hidden": H, "blocks": L, "heads": heads,
                       "dropout": rate, "optimizer": "dense TF-Adam",
                       "exchange": ("none" if world == 1 else
                                    ("catalog-sharded table: input rows all-to-all + softmax partials + gradient rows all-gather"
                                     if eng.dp_mode == "catalog" else
                                     ("row-sharded table update + all-gather" if (eng.dp_sharded and eng.shadow is not None)
                                      else "dense gradient all-reduce"))),
                       "precision": {"bf16": "logit GEMMs: bf16 operands, fp32 accumulate + softmax; block GEMMs and attention: bf16x3 "
                                             "(three bf16 MFMAs per product on hi/lo splits, ~2^-16 relative, fp32 accumulate); "
                                             "LayerNorm, softmax, optimizer, master weights: fp32",
                                     "x3": "every GEMM bf16x3 (three bf16 MFMAs per product on hi/lo splits, ~2^-16 relative, fp32 "
                                           "accumulate): float32-grade; LayerNorm, softmax, optimizer, master weights: fp32",
                                     "f32": "fp32 throughout (f32 MFMA)"}[args.logits],
                       "parallelism": "dp%d" % world, "final_loss": loss,
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1)},
            # float32-grade companion of the same step (logits_dtype = x3; reference arithmetic is fp32, ADER.py:91-93)
            "value_f32grade": f32g["value"] if f32g else None,
            "ms_per_step_f32grade": f32g["ms_per_step"] if f32g else None,
            "f32grade": f32g,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
