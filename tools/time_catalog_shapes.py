"""Per-rank kernel times of the catalog-sharded step at W = 2, 4, 8 shapes, emulated on one GPU without communication:
ader_lbf_fwd_shard over N/W items for W*512 batch rows, ader_tab_update over the N/W-row shard for W*512 rows (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd._lib import call, ptr
from ader_amd.engine import Engine
N, H, T = 1_000_000, 150, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="bf16")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for W in (1, 2, 4, 8):
    B = 512 * W
    S = -(-N // (128 * W)) * 128
    rep = (torch.randn(B, H, generator=g) * 0.5).to(dev)
    rep_bf = torch.zeros(B * 168, dtype=torch.bfloat16, device=dev)
    call("ader_lbf_prep", ptr(rep), ptr(rep_bf), B, B, H, st)
    R = call("ader_lbf_ranges", S, B)
    pm, pl, pO = torch.empty(R * B, device=dev), torch.empty(R * B, device=dev), torch.empty(R * B * 160, device=dev)
    part = torch.empty(B * 152, device=dev)
    t_fwd = timed(lambda: call("ader_lbf_fwd_shard", ptr(rep_bf), ptr(eng.shadow), N, B, H, N, 0, S, ptr(pm), ptr(pl), ptr(pO), ptr(part), st))
    off = torch.full((B,), -20.0, device=dev)
    wrow = torch.full((B,), 1.0 / B, device=dev)
    seq = torch.randint(1, N + 1, (B * T // W,), generator=g, dtype=torch.int32).to(dev)     # rows this shard receives
    lab = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    gsrc = torch.randn(seq.numel(), H, generator=g).to(dev) * 1e-3
    ids, order, sp_start, tids, torder, tg_start, tmeta = eng._sparse_lists(seq, lab, N)
    tiles = S // 128
    t_upd = timed(lambda: call("ader_tab_update_sh", ptr(rep_bf), ptr(eng.shadow), N, B, B, H, N, ptr(off), ptr(ids), ptr(order),
                               ptr(sp_start), ids.numel(), ptr(gsrc), 12.2, ptr(tids), ptr(torder), ptr(tg_start), tids.numel(),
                               ptr(wrow), ptr(eng.theta), ptr(eng.adam_m), ptr(eng.adam_v), 1e-4, 0.9, 0.999, 1e-8, 0, tiles, None, st))
    print("W=%d  rows %4d  shard %7d items:  logits fwd %7.1f us   table update %7.1f us" % (W, B, S, t_fwd, t_upd))
