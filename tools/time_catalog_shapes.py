"""Per-rank kernel times of the catalog-sharded step at W = 1, 2, 4, 8 shapes, emulated on one GPU without communication (dev tool):
the flash forward over N/W items for W*512 batch rows and the fused table update over the N/W-row shard for W*512 rows, for both
logits types.  Feeds the predicted multi-GPU step times of DESIGN.md section 5.   python tools/time_catalog_shapes.py [x3|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd._lib import call, ptr
from ader_amd.engine import Engine
N, H, T = 1_000_000, 150, 50
mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype=mode)
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
    R = call("ader_lbf_ranges", S, B)
    pm, pl, pO = torch.empty(R * B, device=dev), torch.empty(R * B, device=dev), torch.empty(R * B * 160, device=dev)
    part = torch.empty(B * 152, device=dev)
    off = torch.full((B,), -20.0, device=dev)
    wrow = torch.full((B,), 1.0 / B, device=dev)
    seq = torch.randint(1, N + 1, (B * T // W,), generator=g, dtype=torch.int32).to(dev)     # rows this shard receives
    lab = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    gsrc = torch.randn(seq.numel(), H, generator=g).to(dev) * 1e-3
    ids, order, sp_start, tids, torder, tg_start, tmeta = eng._sparse_lists(seq, lab, N)
    tiles = S // 128
    if mode == "x3":
        rep_hi = torch.zeros(B * 168, dtype=torch.bfloat16, device=dev)
        rep_lo = torch.zeros(B * 168, dtype=torch.bfloat16, device=dev)
        call("ader_lx3_prep", ptr(rep), ptr(rep_hi), ptr(rep_lo), B, B, H, st)
        img = torch.zeros(call("ader_x3_rep_image_bytes", B), dtype=torch.uint8, device=dev)
        call("ader_x3_rep_image", ptr(rep_hi), ptr(rep_lo), B, ptr(img), st)
        t_fwd = timed(lambda: call("ader_lx3_fwd_shard", ptr(rep_hi), ptr(rep_lo), eng._pp["emb"], N, B, H, N, 0, S, ptr(pm), ptr(pl),
                                   ptr(pO), ptr(part), st))
        t_upd = timed(lambda: call("ader_tab_update_x3", ptr(rep_hi), ptr(rep_lo), ptr(img), N, B, B, H, N, ptr(off), ptr(ids),
                                   ptr(order), ids.numel(), ptr(gsrc), 12.2, ptr(tids), ptr(torder), tids.numel(), ptr(tmeta),
                                   ptr(wrow), ptr(eng.theta), ptr(eng.adam_m), ptr(eng.adam_v), 1e-4, 0.9, 0.999, 1e-8, 0, tiles,
                                   None, st))
    else:
        rep_bf = torch.zeros(B * 168, dtype=torch.bfloat16, device=dev)
        call("ader_lbf_prep", ptr(rep), ptr(rep_bf), B, B, H, st)
        t_fwd = timed(lambda: call("ader_lbf_fwd_shard", ptr(rep_bf), ptr(eng.shadow), N, B, H, N, 0, S, ptr(pm), ptr(pl), ptr(pO),
                                   ptr(part), st))
        t_upd = timed(lambda: call("ader_tab_update_sh", ptr(rep_bf), ptr(eng.shadow), N, B, B, H, N, ptr(off), ptr(ids), ptr(order),
                                   ptr(sp_start), ids.numel(), ptr(gsrc), 12.2, ptr(tids), ptr(torder), ptr(tg_start), tids.numel(),
                                   ptr(wrow), ptr(eng.theta), ptr(eng.adam_m), ptr(eng.adam_v), 1e-4, 0.9, 0.999, 1e-8, 0, tiles,
                                   None, st))
    print("%s  W=%d  rows %4d  shard %7d items:  logits fwd %7.1f us   table update %7.1f us" % (mode, W, B, S, t_fwd, t_upd), flush=True)
