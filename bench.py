#!/usr/bin/env python
"""Benchmark of the ADER / SASRec training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: bench.py starts its own N ranks -- a child `python -m torch.distributed.run` -- before it touches
     the GPU, relays rank 0's JSON line and exits with the children's code; under torchrun it is one of the ranks.)

A step = one pass of the hot path over one batch of synthetic input already resident in HBM: embedding gather ->
causal self-attention blocks -> full-catalog logits + softmax CE -> backward -> [RCCL gradient all-reduce] -> dense
Adam (the `sess.run(train_op)` of reference main.py:233-256).  Workload = BASELINE.json configs[4]: synthetic 1M-item
catalog, seq_len 50, batch 512 per GPU (weak scaling), dense regime (every position a real item, ids ~ U[1,N]),
dropout 0.3, lr 5e-4, hidden 150, 2 blocks, 1 head (reference defaults main.py:98-107).

The headline is the float32-grade step (`--logits x3`: every GEMM as three bf16 MFMAs on hi/lo operand splits, ~2^-16 relative
per product, fp32 accumulate -- the reference's arithmetic is float32, ADER.py:91-93); the bf16-operand step is reported beside it
as `value_bf16`.  The timed region is repeated `--reps` times (K steps each) and `value` is the MEDIAN repetition (min / max in
`reps_ms`).  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the launch stream inside the
timed region; `logit_gemm`, `gather`, `herding` sub-blocks) and `cpu_baseline` (the CPU oracle timed on this host's cores).
"""
import subprocess
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
HBM_ACHIEVABLE_GBS = 6300.0    # MI355X_MICROARCH.md: what a pure streaming kernel reaches (frac_of_achievable)
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0
BF16_MFMA_SUSTAINED_TFLOPS = 1370.0   # what the chip holds under the logit kernels at its power-limited ~1.4 GHz (profiles/r5x_pmc_sq.json:
#                                       k_lx3p 0.91 MFMA-busy; the round-5 review's figure) -- the rate the shape floors below are priced at


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


REAL_SHAPES = {  # step shapes of the real-data configurations: items, train rows, distilled exemplar rows
    "cfgD": ("DIGINETICA ADER, last period (BASELINE.json configs[1])", 43105, 256, 143),
    "cfgY": ("YOOCHOOSE ADER, last period (configs[2])", 25750, 512, 102),
}
# ... and the ADER-mode step at the headline catalog (what every period > 1 runs, ADER.py:108-137): cfg-S + 128 distilled rows, dense regime
ADER128 = ("cfg-S + 128 distilled exemplar rows (ADER mode at the headline catalog, configs[4] + ADER.py:132-137)", 1_000_000, 512, 128)


def shape_roofline(N, B, E, Np, positions, ms, H=150, L=2, x3=True):
    """Work and compulsory traffic of ONE distilled train step of a shape, and the floor they imply (SURVEY 8d accounting; DESIGN.md
    "roofline of the real shapes").  positions = real (non-padding) positions of the batch; the last block is pruned to the B + E
    query rows (its K / V projections still run on every position).
      logit MFMA   credited: logits + softmax readout O1 (train rows over N, exemplar rows over Np), teacher readout O2 (E x Np),
                   table-gradient product dE (rows^T x items); executed: the same on rows padded to 128 and three bf16 passes per
                   product (the hi/lo split) when x3
      session MFMA credited: 5 HxH products per position and block forward, twice that backward (dX and dW); pruned last block:
                   K, V on every position, Q / W1 / W2 on the query rows; attention products are ~2 % and left out
      bytes        theta / Adam m / Adam v of the N rows read and written once (6 N H 4), the teacher rows read twice (readout,
                   update), the session activations written once and read once by the backward (~24 [positions, H] tensors)
      floor_ms     executed MFMA work at the sustained bf16 rate + bytes at the achievable HBM rate, summed: the phases of a step
                   depend on each other (forward -> logits -> backward -> update), they do not overlap"""
    rows, rp = B + E, ((B + 127) // 128 + (E + 127) // 128) * 128
    logit_cred = 2.0 * H * (2.0 * B * N + 2.0 * E * Np + 1.0 * E * Np + 1.0 * B * N + 1.0 * E * Np)
    pad = rp / float(rows)
    logit_exec = logit_cred * pad * (3 if x3 else 1)
    gemm_rows = (L - 1) * 5.0 * positions + 2.0 * positions + 3.0 * rows
    sess_cred = 3.0 * 2.0 * H * H * gemm_rows
    sess_exec = sess_cred * (3 if x3 else 1)
    nbytes = 6.0 * N * H * 4 + 2.0 * E * Np * 4 + 2.0 * 24 * positions * H * 4
    t_logit = logit_exec / (BF16_MFMA_SUSTAINED_TFLOPS * 1e12) * 1e3
    t_sess = sess_exec / (BF16_MFMA_SUSTAINED_TFLOPS * 1e12) * 1e3
    t_hbm = nbytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
    floor = t_logit + t_sess + t_hbm
    return {"logit_flops_credited": logit_cred, "logit_flops_executed": logit_exec, "session_flops_credited": sess_cred,
            "session_flops_executed": sess_exec, "compulsory_bytes": nbytes,
            "floor_parts_ms": {"logit_mfma": round(t_logit, 5), "session_mfma": round(t_sess, 5), "hbm": round(t_hbm, 5)},
            "floor_ms": round(floor, 5), "frac": round(floor / ms, 4),
            "priced_at": {"bf16_mfma_sustained_TFLOPs": BF16_MFMA_SUSTAINED_TFLOPS, "hbm_achievable_GBps": HBM_ACHIEVABLE_GBS},
            "credited_TFLOPs": round((logit_cred + sess_cred) / (ms * 1e-3) / 1e12, 2),
            "executed_frac_of_bf16_peak": round((logit_exec + sess_exec) / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "hbm_frac_of_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def real_shape_line(name, dev, seconds=1.2, empty_cache=True):
    """One real-data step shape on its own engine, as main.py runs it: synthetic ids of the realistic law (session length
    1 + Geometric(0.2): the shipped splits' ~90 % padding), the exemplar rows distilled against resident teacher logits over
    0.9 N items, dropout 0.3, float32 grade, packed session tiles by the engine's own rule.  Same protocol as the headline
    (warm-up, then K steps bracketed by synchronize; median of 5), sized to ~`seconds` of GPU time."""
    from ader_amd.engine import Engine, SectionTimer
    label, N, B, E = ADER128 if name == "ader128" else REAL_SHAPES[name]
    regime = "dense" if name == "ader128" else "realistic"
    T, lr, rate = 50, 5e-4, 0.3
    batches = [synth_batch(B + E, T, N, 1000 * s + 77, dev, regime) for s in range(4)]
    eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev).warm_up()
    eng.pack_density = float(np.mean([float((sq != 0).float().mean()) for sq, _ in batches]))
    Np = int(0.9 * N)
    teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]          # rows 16-byte aligned, as Engine.teacher_logits allocates them
    teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))
    kw = dict(rate=rate, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)

    def step(i):
        sq, ps = batches[i % 4]
        eng.train_step(sq, ps[:B], N, lr, **kw)
    for i in range(8):
        step(i)
    eng.timer = SectionTimer()
    for i in range(4):
        step(i)
    sections = eng.timer.collect()
    eng.timer = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    K = max(40, int(seconds / 5 / max((time.perf_counter() - t0) / 40, 1e-5)))
    dts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / K * 1e3)
    eng.check_status()
    ms = float(np.median(dts))
    # (these steps take 0.4 ms of GPU time against 0.3 ms of host enqueue, so anything on the host shows at once.  The 3x slow blocks
    #  this line showed in round 5 were NOT that: the engine's side stream had landed on the slow one of HIP's four high-priority
    #  hardware queues -- engine.side_stream() now probes, profiles/r5_packed/side_stream_queues.txt.  The spread of the repetitions
    #  is on the line, and a disturbed block is repeated once)
    if max(dts) > 1.5 * min(dts) or ms > 1.5 * min(dts):
        dts2 = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                step(i)
            torch.cuda.synchronize()
            dts2.append((time.perf_counter() - t0) / K * 1e3)
        if float(np.median(dts2)) < ms:
            dts, ms = dts2, float(np.median(dts2))
    # host side of a step: enqueue time of 8 steps on an idle GPU (nothing blocks), and how they were driven
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        step(i)
    host_ms = (time.perf_counter() - t0) / 8 * 1e3
    torch.cuda.synchronize()
    pk = eng._act.get("pack")
    positions = int(pk["hdr"][2].item()) if pk is not None else (B + E) * T
    out = {"workload": "step shape of %s: N=%d items, %d train + %d distilled rows, synthetic ids, %s" % (
               label, N, B, E, "realistic length law" if regime == "realistic" else "dense regime (every position real)"),
           "host_enqueue_ms": round(host_ms, 4),
           "driver": ("native launch plan, one C call per step (%d replayed / %d recorded)" % (eng.plan_hits, eng.plan_misses)
                      if eng.plan_hits else "Python-driven launches"),
           "roofline": shape_roofline(N, B, E, Np, positions, ms),
           "ms_per_step": ms, "sessions_per_s": B / ms * 1e3, "rows_per_s": (B + E) / ms * 1e3, "steps": K, "reps_ms": [round(x, 4) for x in dts],
           "ms_per_step_min": round(min(dts), 4),
           "sections_ms": {k: round(v, 4) for k, v in sections.items()}, "real_positions_fraction": round(eng.pack_density, 4),
           "session_tiles": "packed" if pk is not None else "one session per workgroup", "final_loss": float(eng.loss.item())}
    if pk is not None:
        out["tiles"] = int(pk["hdr"][0].item())
        out["positions"] = int(pk["hdr"][2].item())
    del eng, teacher
    if empty_cache:
        torch.cuda.empty_cache()
    return out


def dp_shape_legs(name, dev, rank, world, steps=40, warmup=8, reps=3):
    """BASELINE.json configs[3] on the SCALE line: the YOOCHOOSE ADER step shape (reference README.md:77 scaled out: global batch
    512 x W train rows + 102 x W distilled rows, rows of BOTH sub-batches sharded over the ranks as ader_amd/main.py shards them,
    teacher logits replicated), weak scaling, BOTH data-parallel schemes back to back: "replicated" (every rank holds the table:
    dense gradient all-reduce -- the plain one at this catalog size, dist.DataParallel.early_pays) and "catalog" (every rank owns 1/W
    of the rows).  Per scheme: one guarded first step (every collective announced and compared across the ranks before it is
    issued), then `reps` x `steps` steps bracketed by barrier + synchronize, MAX over ranks, median repetition."""
    import torch.distributed as dist
    from ader_amd import dist as adist
    from ader_amd.engine import Engine, pack_counts_host
    label, N, B, E = REAL_SHAPES[name]
    T, H, lr, rate = 50, 150, 5e-4, 0.3
    nb = 4
    batches = [synth_batch(B + E, T, N, 1000 * s + 77 + rank, dev, "realistic") for s in range(nb)]
    Np = int(0.9 * N)
    teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]
    teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))          # (replicated, as the exemplar store is)
    trow = torch.arange(E, dtype=torch.int32, device=dev)
    density = float(np.mean([float((sq != 0).float().mean()) for sq, _ in batches]))
    out = {"workload": "step shape of %s scaled out: %d ranks x (%d train + %d distilled rows), N=%d items, realistic length law "
                       "(BASELINE.json configs[3] at %d ranks)" % (label, world, B, E, N, world), "global_batch": B * world,
           "global_exemplar_rows": E * world, "scaling": "weak", "steps": steps, "schemes": {}}

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    for mode in ("replicated", "catalog"):
        eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, device=dev, dp_rank=rank, dp_world=world).warm_up()
        dp = adist.DataParallel(eng, rank, world)
        eng.dp_mode, eng.pack_density = mode, density
        dp.set_rows(rank * B, N, ex_row0=B * world + rank * E)
        kw = dict(rate=rate, teacher=teacher, ex_trow=trow, lambda_=0.8, n_train_global=B * world, n_ex_global=E * world)
        counts = [None] * nb
        if mode == "catalog" and eng.dp_pack:
            # split sizes of the packed row exchange from the host copy of the GLOBAL batch's ids (every rank can form it: the batches
            # are a function of (step, rank)); for this leg's four resident batches they are computed once, ahead of the timed region
            for s_ in range(nb):
                rows = []
                for r_ in range(world):
                    sq, ps = synth_batch(B + E, T, N, 1000 * s_ + 77 + r_, "cpu", "realistic")
                    rows.append(np.concatenate([sq.numpy().reshape(-1), ps.numpy()[:B]]).astype(np.int32))
                counts[s_] = pack_counts_host(np.stack(rows), (B + E) * T, eng.shard_items)

        def step(i):
            sq, ps = batches[i % nb]
            extra = {"pack_counts": counts[i % nb]} if counts[i % nb] is not None else {}
            eng.train_step(sq, ps[:B], N, lr, **kw, **extra)
        adist.guard.start()
        try:
            step(0)
            torch.cuda.synchronize()
        finally:
            log_ = adist.guard.stop()
        sys.stderr.write("\n".join(["[rank %d] real_shapes_dp %s dp_mode=%s world=%d: %d collectives per step"
                                     % (rank, name, mode, world, len(log_))] + adist.guard.describe(rank)) + "\n")
        sys.stderr.flush()
        for i in range(5 + warmup):
            step(i)
        eng.check_status()
        dts = []
        for _ in range(reps):
            sync()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts.append(float(t.item()) / steps * 1e3)
        ms = float(np.median(dts))
        # bytes RECEIVED per rank and step (model)
        W, P, span = world, eng.P, eng.layout["pos"][0]
        small = 2.0 * (W - 1) / W * ((P - span) * 4 + 4)
        rowsT = (B + E) * T
        if mode == "replicated":
            early = dp.early_pays(eng, N)
            xb = 2.0 * (W - 1) / W * (N + 1) * H * 4 + small + ((W - 1) * rowsT * (H + 1) * 4 if early else 0.0)
            how = ("dense all-reduce of the table gradient started under the blocks backward + per-position rows all-gathered"
                   if early else "rows scattered locally, then ONE dense all-reduce of the whole gradient (table " "%.1f MB + %.1f MB of "
                   "block parameters)" % ((N + 1) * H * 4 / 1e6, (P - span) * 4 / 1e6))
        else:
            Bp, Bk = (B + 127) // 128 * 128, (E + 127) // 128 * 128
            n_all = rowsT + B
            f = (W - 1) / W if eng.dp_pack else (W - 1)
            xb = ((W - 1) * n_all * 4 + f * n_all * H * 4 + (W - 1) * (Bp + Bk) * H * 4 + (W - 1) * (Bp + 2 * Bk) * 152 * 4
                  + (W - 1) * (3 * Bp + 6 * Bk) * 4 + f * rowsT * H * 4 + small)
            how = "catalog-sharded table: owned rows, representations, softmax partials and gradient rows travel; nothing table-sized"
        out["schemes"][mode] = {"ms_per_step": ms, "reps_ms": [round(x, 4) for x in dts], "sessions_per_s": B * world / ms * 1e3,
                                "collectives_per_step": len(log_), "exchange_bytes_per_step": int(xb), "exchange": how,
                                "packed_rows": bool(eng.dp_pack) if mode == "catalog" else None,
                                "host_syncs_per_step": (eng.comm_syncs if mode == "catalog" else 0), "final_loss": float(eng.loss.item()),
                                "collectives": [{"site": a_, "kind": b_, "shape": list(c_), "dtype": d_} for a_, b_, c_, d_, _ in log_]}
        del eng, dp
        torch.cuda.empty_cache()
    best = min(out["schemes"], key=lambda k: out["schemes"][k]["ms_per_step"])
    out["faster_scheme"] = best
    return out


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
                     % (n, Bs, " + %d distilled" % E if E else "", T, N, dt, cores)
                     + ("; ONE step on purpose: it is already about twice the 10-30 s of CPU work this leg is bounded to, and a median "
                        "of three would put three minutes of host time into every default run (the warm-up step above has paid for "
                        "the thread pool and the allocator)" if n == 1 else "")}
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
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region (K steps each); value = median")
    ap.add_argument("--logits", choices=["bf16", "x3", "f32"], default="x3",
                    help="operand type of the logit GEMMs (fp32 master table, fp32 accumulate/softmax either way)")
    ap.add_argument("--regime", choices=["dense", "realistic"], default="dense",
                    help="synthetic id/length law (SURVEY 8d); the headline number is the dense regime")
    ap.add_argument("--exemplars", type=int, default=0,
                    help="ADER-mode variant: append this many exemplar rows distilled against N(0,1) teacher logits over 0.9 N items")
    ap.add_argument("--dp-mode", choices=["replicated", "catalog"], default="catalog",
                    help="N > 1: 'catalog' (default: each rank OWNS 1/N of the table rows -- parameters and Adam state --, only the rows "
                         "the inputs touch, the representations and per-row softmax partials travel; DESIGN.md section 5 has the "
                         "byte model) or 'replicated' (every rank holds the table: dense all-reduce of the table gradient overlapped "
                         "with backward at float32 grade, row-sharded update + all-gather with bf16 logits)")
    ap.add_argument("--workload", choices=["cfgS", "cfgD", "cfgY", "cfgF"], default="cfgS",
                    help="cfgS: BASELINE configs[4] (the metric's configuration).  Step-shape variants of the real-data configs "
                         "(SURVEY 8a): cfgD = DIGINETICA ADER last period (N 43,105, 256 train + 143 distilled rows), cfgY = YOOCHOOSE "
                         "ADER last period (N 25,750, 512 + 102 rows), cfgF = DIGINETICA finetune baseline (BASELINE configs[0]: N 43,105, "
                         "batch 128, no exemplars, dropout 0); synthetic ids of those shapes")
    ap.add_argument("--pack", choices=["auto", "on", "off"], default="auto",
                    help="packed session tiles (csrc/seqp_*.hip: the session kernels on the real positions only).  auto: the engine's "
                         "rule on the density of the synthetic batches (packed in the realistic regime, not in the dense one)")
    ap.add_argument("--pack-window", default=None, help="w1_min,w1_max,target of the packing plan (tuning; default: the engine's)")
    ap.add_argument("--no-real-shapes", action="store_true",
                    help="skip the real_shapes block (cfgD / cfgY step shapes with the realistic length law and ADER rows, ~2 s each)")
    ap.add_argument("--no-other-dp-leg", action="store_true",
                    help="N > 1: skip the leg of the OTHER data-parallel scheme (comm.replicated_ms_per_step beside the catalog default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32grade", "--no-companion", dest="no_companion", action="store_true",
                    help="skip the companion run of the other logits type (bf16 beside the x3 headline)")
    ap.add_argument("--no-herding", action="store_true", help="skip the exemplar-selection measurement (YOOCHOOSE period 1)")
    ap.add_argument("--pmc-json", default=None,
                    help="rocprofv3 PMC summary of THIS command (tools/summarize_profiles.py) to quote roofline.traffic from; "
                         "without it traffic is null (bench.py never pairs live timings with counters of another run)")
    ap.add_argument("--no-sections", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="after the median-of-reps block: ONE more timed region of this many steps of the same kernels (~4 s), so the "
                         "line also says what a power-limited chip sustains (0 = skip)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the ranks as children BEFORE anything touches the GPU (never re-exec a GPU process)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if r.returncode != 0 or not lines:
            sys.stderr.write(r.stdout[-4000:] + r.stderr[-4000:])
            sys.exit(r.returncode or 1)
        sys.stderr.write("\n".join(ln for ln in r.stderr.splitlines() if ln.startswith("[rank ")) + "\n")     # the ranks' collective lists
        print(lines[-1])
        sys.exit(0)

    from ader_amd import dist as adist
    from ader_amd.engine import Engine, SectionTimer
    import torch.distributed as dist

    # ADER_DIST_BACKEND=gloo lets several ranks share one GPU (functional check of the N > 1 path without an 8-GPU node)
    rank, world, local = adist.init(os.environ.get("ADER_DIST_BACKEND", "nccl"))
    local = local % max(torch.cuda.device_count(), 1)
    if world != args.gpus:
        raise RuntimeError("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.workload == "cfgD":
        args.items, args.batch, args.exemplars = 43105, 256, 143
    elif args.workload == "cfgY":
        args.items, args.batch, args.exemplars = 25750, 512, 102
    elif args.workload == "cfgF":
        args.items, args.batch, args.exemplars = 43105, 128, 0
    N, B, T, H, L, heads, rate, lr = args.items, args.batch, 50, 150, 2, 1, (0.0 if args.workload == "cfgF" else 0.3), 5e-4
    E = args.exemplars
    nbatch = 4
    batches = [synth_batch(B + E, T, N, 1000 * s + rank, dev, args.regime) for s in range(nbatch)]   # resident in HBM before timing
    kw = dict(rate=rate, n_train_global=B * world)
    if E:
        Np = int(0.9 * N)
        teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]      # resident teacher logits [E,Np]; rows 16-byte aligned, as
        teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))     # Engine.teacher_logits allocates them
        kw.update(teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8, n_ex_global=E * world)
        batches = [(sq, ps[:B]) for sq, ps in batches]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_engine(logits, dp_mode=None):
        eng_ = Engine(N, maxlen=T, hidden_units=H, num_blocks=L, num_heads=heads, seed=0, device=dev, logits_dtype=logits,
                      dp_rank=rank, dp_world=world).warm_up()
        dp_ = adist.DataParallel(eng_, rank, world)
        if world > 1 and (logits == "x3" or (logits == "bf16" and not E)):     # (distilled rows on the sharded table: float32 grade)
            eng_.dp_mode = dp_mode or args.dp_mode
        dp_.set_rows(rank * B, N)
        # the feeder's announcement: fraction of real positions of the batches (what Sampler.to_device tells the engine in main.py)
        eng_.pack_density = float(np.mean([float((sq != 0).float().mean()) for sq, _ in batches]))
        eng_.pack_sessions = {"auto": "auto", "on": True, "off": False}[args.pack]
        if args.pack_window:
            eng_.pack_window = tuple(int(v) for v in args.pack_window.split(","))
        return eng_

    # packed catalog exchange (the default from 8 ranks on): its all-to-all split sizes come from the ids of the GLOBAL batch, which
    # every rank can form on the host (the synthetic batches of all ranks are a function of (s, rank); a training loop's feeder
    # holds the rows on the host too: Sampler._rows).  The counts themselves are computed INSIDE the timed region, per step, as a
    # training loop has to -- by a one-step-ahead worker thread (numpy releases the interpreter lock), so no device-to-host
    # synchronisation and no host work on the step's critical path
    pack_ids = [None] * nbatch
    pack_state = {"eng": None, "ex": None, "fut": None, "fut_i": -1}

    def prepare_pack_counts(eng_):
        pack_state.update(eng=None, fut=None, fut_i=-1)
        if world > 1 and eng_.dp_mode == "catalog" and eng_.dp_pack and not E:
            import concurrent.futures
            for s_ in range(nbatch):
                if pack_ids[s_] is None:
                    rows = []
                    for r_ in range(world):
                        sq, ps = synth_batch(B, T, N, 1000 * s_ + r_, "cpu", args.regime)
                        rows.append(np.concatenate([sq.numpy().reshape(-1), ps.numpy()]).astype(np.int32))
                    pack_ids[s_] = np.stack(rows)
            if pack_state["ex"] is None:
                pack_state["ex"] = concurrent.futures.ThreadPoolExecutor(1)
            pack_state["eng"] = eng_

    def pack_for(i):
        eng_ = pack_state["eng"]
        if eng_ is None:
            return {}
        from ader_amd.engine import pack_counts_host
        ex = pack_state["ex"]
        fut = pack_state["fut"] if pack_state["fut_i"] == i else ex.submit(pack_counts_host, pack_ids[i % nbatch], B * T, eng_.shard_items)
        pack_state["fut"], pack_state["fut_i"] = ex.submit(pack_counts_host, pack_ids[(i + 1) % nbatch], B * T, eng_.shard_items), i + 1
        return {"pack_counts": fut.result()}

    guard_logs = {}

    def timed(eng_, reps, sections=True):
        """5 untimed initialisation steps, W warm-up steps (per-kernel HIP events on the last three), then `reps` repetitions of
        EXACTLY K steps, each bracketed by a barrier + synchronize on both sides; per repetition the MAX over ranks."""
        sec_all = {}
        prepare_pack_counts(eng_)
        if world > 1:
            # first contact (the multi-GPU node is the driver's): ONE guarded step before anything is timed -- every collective is
            # announced and compared across the ranks before it is issued (ader_amd/dist.py: CollectiveGuard), a mismatch raises with
            # the rank and the call site instead of hanging in RCCL -- and every rank prints what it issued
            adist.guard.start()
            try:
                eng_.train_step(*batches[0], N, lr, **kw, **pack_for(0))
                torch.cuda.synchronize()
            finally:
                log_ = adist.guard.stop()
            sys.stderr.write("\n".join(["[rank %d] dp_mode=%s world=%d backend=%s: %d collectives per step"
                                         % (rank, eng_.dp_mode, world, dist.get_backend(), len(log_))] + adist.guard.describe(rank)) + "\n")
            sys.stderr.flush()
            guard_logs[eng_.dp_mode] = [{"site": a_, "kind": b_, "shape": list(c_), "dtype": d_,
                                         "splits": None if e_ is None else {"send": list(e_[0]), "recv": list(e_[1])}}
                                        for a_, b_, c_, d_, e_ in log_]
        for i in range(5):        # engine initialisation (workspace allocation, kernel attributes, side streams): never timed
            eng_.train_step(*batches[i % nbatch], N, lr, **kw, **pack_for(i))
        for i in range(args.warmup):
            if sections and i == max(0, args.warmup - 3):
                eng_.timer = SectionTimer()
            eng_.train_step(*batches[i % nbatch], N, lr, **kw, **pack_for(i))
        eng_.check_status()
        if sections:
            if eng_.timer is not None:
                sec_all = eng_.timer.collect()
            # (a timing event pair around a kernel breaks its overlap with the side stream and costs ~60 us of the step: in the timed
            #  region only the two logit kernels are timed, on every 8th step)
            eng_.timer = SectionTimer(only={"logits_bwd_adam", "logits_fwd"}, every=8)
        # The interpreter's cyclic collector walks every object torch has imported (~40 ms per full pass): a pass that lands in the
        # timed region is host noise of the same size as the region.  Collect now and freeze what exists (the steps themselves
        # create no cycles); a training loop does the same once at start-up (ader_amd/main.py).
        gc.collect()
        gc.freeze()
        dts = []
        for _ in range(max(1, reps)):
            sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                eng_.train_step(*batches[i % nbatch], N, lr, **kw, **pack_for(i))
            sync()
            dt_ = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt_], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.item())
            dts.append(dt_)
        sec = eng_.timer.collect() if eng_.timer is not None else {}
        eng_.timer = None
        return dts, {**sec_all, **sec}, float(eng_.loss.item())

    eng = make_engine(args.logits)
    dts, sections, loss = timed(eng, args.reps, not args.no_sections)
    session_tiles = "packed" if eng._act.get("pack") is not None else "one session per workgroup"
    dt = float(np.median(dts))
    value_dp_mode = eng.dp_mode if world > 1 else None
    # the other data-parallel scheme in the same invocation (north_star words the replicated one: dense all-reduce over xGMI overlapped
    # with backward; `value` is the catalog-sharded default): one repetition of the same K steps
    other_leg = None
    if world > 1 and not args.no_other_dp_leg and args.logits == "x3" and not E:
        other = "replicated" if eng.dp_mode == "catalog" else "catalog"
        eng_o = make_engine(args.logits, dp_mode=other)
        dts_o, _, loss_o = timed(eng_o, 1, False)
        other_leg = {"dp_mode": other, "ms_per_step": dts_o[0] / args.steps * 1e3, "value": B * world * args.steps / dts_o[0],
                     "final_loss": loss_o, "steps": args.steps}
        del eng_o
        torch.cuda.empty_cache()

    # ---- configs[3] (YOOCHOOSE ADER, global batch 512 x N) in the same multi-GPU invocation, both schemes: every rank takes part
    real_dp = None
    if world > 1 and not args.no_real_shapes and args.workload == "cfgS" and not E and args.logits == "x3":
        try:
            real_dp = {"cfgY": dp_shape_legs("cfgY", dev, rank, world, steps=max(10, min(40, args.steps)))}
        except Exception as e:          # (collectives inside: a failure on one rank is fatal for all, so it is re-raised after the note)
            sys.stderr.write("[rank %d] real_shapes_dp failed: %r\n" % (rank, e))
            raise

    # ---- sustained figure: the K-step repetitions above are ~40 ms bursts on a chip that clocks down under the logit kernels
    # (MI355X_MICROARCH.md, DVFS): one multi-second region of the SAME steps, no per-kernel events, same barrier + synchronize
    sustained = None
    if args.sustained_steps > 0 and args.workload == "cfgS":
        eng.timer = None
        sync()
        t0 = time.perf_counter()
        for i in range(args.sustained_steps):
            eng.train_step(*batches[i % nbatch], N, lr, **kw, **pack_for(i))
        sync()
        dts_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dts_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts_ = float(t.item())
        eng.check_status()
        sustained = {"steps": args.sustained_steps, "seconds": round(dts_, 3), "ms_per_step": dts_ / args.sustained_steps * 1e3,
                     "value": B * world * args.sustained_steps / dts_,
                     "clock_note": "one uninterrupted region after the median-of-reps block (the chip has been under load for its whole "
                                   "length: steady-state clocks); `value` above is the median of short bursts"}

    # ---- standalone embedding gather (north_star: "rocprof HBM GB/s on the gather"): in the step it is fused into the one-launch
    # forward, so it is timed here as its own kernel on the same batch (ader_embed_fwd: ids -> x0 = drop(E[ids]*sqrt(H) + P) * mask)
    gather = None
    if rank == 0 and not args.no_sections:
        import ctypes
        from ader_amd._lib import AderDrop, call, ptr
        seq0 = batches[0][0][:B].contiguous()
        x0 = torch.empty(B * T, H, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
                  "note": "25,600 random 600-B rows of a 600 MB table + 15.4 MB written (one launch of ~7 us: launch / latency bound at this size); H = 150: rows as packed 8-byte pieces, nontemporal stores"}
        if N >= 100_000:
            # the same kernel on 16 batches' worth of ids (409,600 random rows, 493 MB): what the gather reaches when it is not launch-bound
            Bl = 16 * B
            seql = torch.randint(1, N + 1, (Bl, T), generator=torch.Generator().manual_seed(7), dtype=torch.int32).to(dev)
            xl = torch.empty(Bl * T, H, device=dev)
            for it in range(12):
                if it == 2:
                    a.record()
                call("ader_embed_fwd", ptr(seql), eng._pp["emb"], eng._pp["pos"], ptr(xl), Bl * T, T, H, eng.V, ctypes.byref(dd),
                     ptr(eng.status), st)
            b.record()
            torch.cuda.synchronize()
            usl = a.elapsed_time(b) / 10 * 1e3
            gl = Bl * T * (4 + 2 * H * 4) / 1e9
            for it in range(12):            # ... and without the dropout hash (evaluation / herding forwards): the bare gather
                if it == 2:
                    a.record()
                call("ader_embed_fwd", ptr(seql), eng._pp["emb"], eng._pp["pos"], ptr(xl), Bl * T, T, H, eng.V, None, ptr(eng.status), st)
            b.record()
            torch.cuda.synchronize()
            use = a.elapsed_time(b) / 10 * 1e3
            gather["large"] = {"rows": Bl * T, "us": round(usl, 2), "bytes": Bl * T * (4 + 2 * H * 4), "GBps": round(gl / (usl * 1e-6), 1),
                               "frac_hbm": round(gl / (usl * 1e-6) / HBM_PEAK_GBS, 4), "us_no_dropout": round(use, 2),
                               "GBps_no_dropout": round(gl / (use * 1e-6), 1), "frac_hbm_no_dropout": round(gl / (use * 1e-6) / HBM_PEAK_GBS, 4),
                               "note": "training prologue = gather + counter-hash dropout; 16 batches of ids back to back"}
            del seql, xl

    P, span = eng.P, eng.layout["pos"][0]
    dp_mode, dp_pack = eng.dp_mode, eng.dp_pack
    comm_syncs = eng.comm_syncs if eng.dp_pack else 0
    exchange = ("none" if world == 1 else
                ("catalog-sharded table: input rows all-to-all + representations all-gather + softmax partials all-to-all + gradient "
                 "rows all-gather" if (dp_mode == "catalog" and (args.logits == "x3" or (args.logits == "bf16" and not E))) else
                 ("row-sharded table update + all-gather of the updated rows" if (eng.dp_sharded and eng.shadow is not None)
                  else "dense gradient all-reduce (table part started right after the logits backward, under the blocks backward)")))

    # ---- companion run of the other logits type (bf16 operands beside the float32-grade headline, or the reverse)
    comp, comp_name = None, {"x3": "bf16", "bf16": "x3"}.get(args.logits)
    if world == 1 and comp_name and not args.no_companion and not E:
        del eng
        torch.cuda.empty_cache()
        engc = make_engine(comp_name)
        dtc, secc, lossc = timed(engc, 1, not args.no_sections)
        comp = {"logits": comp_name, "ms_per_step": dtc[0] / args.steps * 1e3, "value": B * args.steps / dtc[0], "final_loss": lossc,
                "sections_ms": {k: round(v, 4) for k, v in sorted(secc.items())}}
        del engc
        torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        x3 = args.logits == "x3"
        lpeak = BF16_MFMA_PEAK_TFLOPS if args.logits in ("bf16", "x3") else F32_MFMA_PEAK_TFLOPS
        cat = world if (world > 1 and dp_mode == "catalog" and (args.logits == "x3" or (args.logits == "bf16" and not E))) else 1
        # algorithmic work per launch (DESIGN.md "roofline accounting").  In the flash modes the forward launch also
        # produces dRep (softmax-weighted readout), so it is credited both GEMMs; recomputation and the 3x of the hi/lo split are
        # never credited.  Catalog-sharded N > 1: a rank streams N / world items for world * B rows -- the same products.
        fwd_flops = 2.0 * B * N * H * (2 if args.logits in ("bf16", "x3") else 1)
        work = {
            "logits_fwd": ("mfma", fwd_flops, lpeak),
            "logits_bwd_drep": ("mfma", 2.0 * B * N * H, lpeak),
            "logits_bwd_demb": ("mfma", 2.0 * B * N * H, lpeak),
            "blocks_fwd": ("mfma", L * 2.0 * B * T * (5 * H * H + 2 * T * H), F32_MFMA_PEAK_TFLOPS),
            "blocks_bwd": ("mfma", 2 * L * 2.0 * B * T * (5 * H * H + 2 * T * H), F32_MFMA_PEAK_TFLOPS),
            "adam": ("hbm", 7.0 * (P - span if "logits_bwd_adam" in sections else P) * 4, HBM_PEAK_GBS),
            # fused table update: theta/m/v of rows 1..N in and out (+ bf16 shadow row in and out in bf16 mode; x3 mode has no shadow
            # and counts ONE theta read although the kernel reads theta twice); dE never hits memory
            "logits_bwd_adam": ("hbm", (6.0 * N * H * 4 + (2.0 * N * 336 if args.logits == "bf16" else 0.0)) / cat, HBM_PEAK_GBS),
            # distilled exemplar rows on the exact-f32 kernels: logits + dRep + dE over the 0.9 N teacher columns
            "kd_rows": ("mfma", 3 * 2.0 * E * int(0.9 * N) * H, F32_MFMA_PEAK_TFLOPS),
        }
        comm_names = ("grad_exchange", "param_allgather")
        # ---- counters of the dominant kernels: from a PMC summary of THIS command (--pmc-json), or the committed end-of-round
        # profile IF it was taken from the kernel sources that are running now (profiles/CURRENT.json: sha of ader_amd/csrc);
        # a profile of other kernels is named but never quoted
        pmc_kernel = {"logits_bwd_adam": {"bf16": "k_tab16<", "x3": "k_tab32x3<"}.get(args.logits, "k_tab_upd"),
                      "logits_fwd": {"bf16": "k_lbf_fwd", "x3": "k_lx3p<"}.get(args.logits, "k_logits"), "adam": "k_adam"}
        pmc, sq, pmc_src = {}, {}, None
        std = N == 1_000_000 and B == 512 and not E and world == 1 and args.regime == "dense"
        try:
            if args.pmc_json:
                pmc = json.load(open(args.pmc_json))["kernels"]
                pmc_src = {"file": os.path.relpath(os.path.abspath(args.pmc_json), ROOT), "stale": False}
            elif std:
                cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
                tag = cur[args.logits]
                stale = cur.get("src_sha16") != csrc_sha16()
                pmc_src = {"file": "profiles/%s_pmc_hbm.json" % tag, "kernel_stats": "profiles/%s_kernel_stats.csv" % tag,
                           "sq": "profiles/%s_pmc_sq.json" % tag, "profiled_at_commit": cur.get("git_head"), "stale": stale,
                           "note": "rocprofv3 passes of this same command, run separately (tools/profile_round.sh)"
                                   + ("; taken from OTHER kernel sources than the ones running: not quoted" if stale else "")}
                if not stale:
                    pmc = json.load(open(os.path.join(ROOT, pmc_src["file"])))["kernels"]
                    sq = json.load(open(os.path.join(ROOT, pmc_src["sq"])))["kernels"]
        except Exception:
            pmc, sq, pmc_src = {}, {}, None

        def pick(table, prefix):
            return next((v for k_, v in table.items() if prefix and k_.replace("void ", "").startswith(prefix)), None)

        def rocprof_ms(prefix):
            if not (pmc_src and pmc_src.get("kernel_stats") and not pmc_src.get("stale")):
                return None
            try:
                import csv
                for r_ in csv.DictReader(open(os.path.join(ROOT, pmc_src["kernel_stats"]))):
                    if r_["Name"].replace("void ", "").startswith(prefix):
                        return float(r_["AverageNs"]) / 1e6
            except Exception:
                pass
            return None

        roof = None
        if sections:
            dom = max((k for k in sections if k not in comm_names and k in work), key=lambda k: sections[k])
            bound, amount, peak = work[dom]
            sec = sections[dom] * 1e-3
            ach, unit = (amount / sec / 1e12, "TFLOP/s") if bound == "mfma" else (amount / sec / 1e9, "GB/s")
            hb = pick(pmc, pmc_kernel.get(dom))
            rms = rocprof_ms(pmc_kernel.get(dom, "?"))
            roof = {"kernel": dom + " (" + pmc_kernel.get(dom, "").rstrip("<") + ")", "bound": bound, "achieved": ach, "peak": peak,
                    "unit": unit, "frac": ach / peak,
                    "frac_of_achievable": (ach / HBM_ACHIEVABLE_GBS) if unit == "GB/s" else None,
                    "traffic": (hb.get("hbm_bytes") if (hb and unit == "GB/s") else None),
                    "traffic_source": pmc_src, "ms": sections[dom], "ms_rocprof": rms,
                    "frac_rocprof": (amount / (rms * 1e-3) / (1e9 if unit == "GB/s" else 1e12) / peak) if rms else None,
                    "algorithmic": amount,
                    # whole step against SURVEY 8(d)'s compulsory traffic (6.64 GB at cfg-S, unfused accounting) and the HBM peak
                    "step_bytes": 6.64e9 * (N / 1e6) if (B == 512 and not E) else None,
                    "step_frac": (6.64e9 * (N / 1e6) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (B == 512 and not E and world == 1) else None,
                    "sections_ms": {k: round(v, 4) for k, v in sorted(sections.items())}}
            # the logit GEMM (north_star: "MFMA utilisation on the logit GEMM against gfx950 peak"): the flash forward launch
            if "logits_fwd" in sections and args.logits in ("bf16", "x3"):
                fms = sections["logits_fwd"]
                q = pick(sq, pmc_kernel["logits_fwd"])
                # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1,024 SIMDs; busy fraction = cycles / SIMDs / (kernel time x the clock the
                # profiled pass held: SQ_WAVE_CYCLES counts 4-clock units per resident wave)
                busy = None
                frm = rocprof_ms(pmc_kernel["logits_fwd"])
                if q and frm and q.get("SQ_VALU_MFMA_BUSY_CYCLES") and q.get("SQ_WAVE_CYCLES"):
                    waves = 8 * 256 if x3 else 8 * 256
                    clocks = 4.0 * q["SQ_WAVE_CYCLES"] / waves                  # clocks the launch lasted (resident-wave average)
                    busy = q["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / clocks
                roof["logit_gemm"] = {"kernel": pmc_kernel["logits_fwd"].rstrip("<") + " (logits + softmax + readout)", "ms": fms,
                                      "flops_credited": fwd_flops, "flops_executed": fwd_flops * (3 if x3 else 1),
                                      "TFLOPs_credited": fwd_flops / (fms * 1e-3) / 1e12,
                                      "frac_of_bf16_peak_credited": fwd_flops / (fms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                      "frac_of_bf16_peak_executed": fwd_flops * (3 if x3 else 1) / (fms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                      "mfma_busy_profiled": busy}
            roof["gather"] = gather
        comm = None
        if world > 1:
            # bytes RECEIVED per rank and step (model; W = ranks): see DESIGN.md section 5
            W, n_all, Bp = world, B * T + B, (B + 127) // 128 * 128
            small = 2.0 * (W - 1) / W * (P - span) * 4
            if cat > 1:
                rows = (W - 1) * n_all * H * 4 if not dp_pack else (W - 1) / W * n_all * H * 4
                rb = ((W - 1) * n_all * 4 + rows + (W - 1) * Bp * (H * 4 if x3 else 336) + (W - 1) * Bp * 152 * 4
                      + (W - 1) * 3 * Bp * 4 + ((W - 1) * B * T * H * 4 if not dp_pack else (W - 1) / W * B * T * H * 4) + small)
            elif eng_sharded(args.logits):
                rb = (W - 1) * (Bp * 336 + 3 * Bp * 4 + B * T * 4 + B * T * H * 4) + (W - 1) / W * N * H * 4 + small
            else:
                rb = 2.0 * (W - 1) / W * (N + 1) * H * 4 + (W - 1) * (B * T * 4 + B * T * H * 4) + small
            comm = {"dp_mode": dp_mode if cat > 1 or eng_sharded(args.logits) else "dense all-reduce", "packed_rows": bool(dp_pack),
                    "value_is": "the %s scheme" % (value_dp_mode,),
                    ("%s_ms_per_step" % (other_leg["dp_mode"] if other_leg else "other")): (other_leg["ms_per_step"] if other_leg else None),
                    "other_leg": other_leg,
                    # what rank 0 announced in the guarded first step of each scheme (every rank prints its own list to stderr)
                    "collectives": guard_logs,
                    "split_sizes": ("computed per step INSIDE the timed region from the host copy of the global batch's ids "
                                    "(engine.pack_counts_host on a one-step-ahead worker thread): no device-to-host synchronisation"
                                    if (cat > 1 and dp_pack) else None),
                    # per step of the catalog-sharded scheme: collectives issued, launches of exchange bookkeeping, host syncs
                    "collectives_per_step": (8 if cat > 1 else None),
                    "bookkeeping_launches_per_step": ((1 + 7) if (cat > 1 and dp_pack) else (2 if cat > 1 else None)),
                    "host_syncs_per_step": (comm_syncs if cat > 1 else None),
                    "comm_ms": round(sum(sections.get(k, 0.0) for k in comm_names), 4),
                    "comm_ms_note": "host-side sections around the collectives in the warm-up steps (they include the kernels that pack / "
                                    "unpack the exchanged rows); the dense all-reduce overlaps the blocks backward",
                    "exchange_bytes_per_step": int(rb)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N = 1 leg only
            try:
                cpu = cpu_baseline(N, B, T, H, L, heads, rate, lr, E, int(0.9 * N) if E else 0)
            except Exception as e:  # report, never fake
                cpu = {"value": None, "unit": "sessions/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        herd = None
        if world == 1 and not args.no_herding and args.workload == "cfgS" and not E:
            try:        # exemplar selection of YOOCHOOSE period 1 (groups up to 1,710 rows, SURVEY 8d "Herding measurement")
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import contextlib
                import bench_herding
                torch.cuda.empty_cache()
                with contextlib.redirect_stdout(sys.stderr):        # (the data loader prints the reference's log lines)
                    herd = bench_herding.measure("YOOCHOOSE", 1, cpu=not args.no_cpu_baseline)
            except Exception as e:
                herd = {"failed": repr(e)}
        if roof is not None:
            roof["herding"] = herd
        real = None
        if world == 1 and not args.no_real_shapes and args.workload == "cfgS" and not E and args.regime == "dense":
            # the workloads the reference trains on (BASELINE.json configs[1], [2]) on the driver-run line, after everything else
            real = {}
            for nm in list(REAL_SHAPES) + ["ader128"]:
                try:
                    real[nm] = real_shape_line(nm, dev, seconds=2.0 if nm == "ader128" else 1.2)
                except Exception as e:
                    real[nm] = {"failed": repr(e)}
        prec = {"bf16": "logit GEMMs: bf16 operands, fp32 accumulate + softmax; block GEMMs and attention: bf16x3 (three bf16 MFMAs "
                        "per product on hi/lo splits, ~2^-16 relative, fp32 accumulate); LayerNorm, softmax, optimizer, master "
                        "weights: fp32",
                "x3": "float32 grade: every GEMM bf16x3 (three bf16 MFMAs per product on hi/lo operand splits, ~2^-16 relative per "
                      "product, fp32 accumulate); LayerNorm, softmax, optimizer, master weights: fp32",
                "f32": "fp32 throughout (f32 MFMA)"}
        out = {
            "metric": "train sessions/sec at batch=512 seq=50", "value": B * world * args.steps / dt, "unit": "sessions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "x3": "bf16x3 (2^-16/product, fp32 accumulate)", "f32": "f32"}[args.logits], "data": "synthetic",
            "reps": len(dts), "reps_ms": {"median": ms, "min": min(dts) / args.steps * 1e3, "max": max(dts) / args.steps * 1e3,
                                          "all": [round(x / args.steps * 1e3, 4) for x in dts]},
            "config": {"workload": ("synthetic %s-item catalog, seq_len=50, batch=%d/GPU, %s regime%s (BASELINE.json configs[4]%s)"
                                    % ("1M" if N == 1_000_000 else "%d" % N, B, args.regime,
                                       ", +%d distilled exemplar rows" % E if E else "",
                                       "" if (N == 1_000_000 and B == 512) else ", REDUCED SIZE")) if args.workload == "cfgS" else
                                   ("step shape of %s: N=%d items, %d train + %d distilled rows, synthetic ids (%s regime)"
                                    % ({"cfgD": "DIGINETICA ADER (BASELINE.json configs[1])", "cfgY": "YOOCHOOSE ADER (configs[2])",
                                        "cfgF": "DIGINETICA finetune baseline (configs[0], main.py --finetune=True)"}
                                       [args.workload], N, B, E, args.regime)),
                       "items": N, "batch_per_gpu": B, "global_batch": B * world, "seq_len": T, "hidden": H, "blocks": L, "heads": heads,
                       "dropout": rate, "optimizer": "dense TF-Adam", "exchange": exchange, "precision": prec[args.logits],
                       "parallelism": "dp%d" % world, "dp_mode_of_value": value_dp_mode, "final_loss": loss,
                       "session_tiles": session_tiles,
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1)},
            "sustained": sustained,
            "comm": comm,
            # companion of the same step with the other logits type (bf16 operands: narrower than the reference's float32)
            ("value_" + (comp_name or "companion")): comp["value"] if comp else None,
            ("ms_per_step_" + (comp_name or "companion")): comp["ms_per_step"] if comp else None,
            "companion": comp,
            "roofline": roof, "cpu_baseline": cpu,
            "real_shapes": real,
            "real_shapes_dp": real_dp,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def eng_sharded(logits):
    """replicated mode: the row-sharded table update exists for bf16 logits (it rebuilds the shadow rows it all-gathers)"""
    return logits == "bf16"


def csrc_sha16():
    """sha256 (first 16 hex digits) of the kernel sources: ties a committed profile to the kernels it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ader_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()
