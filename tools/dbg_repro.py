"""Is the headline step bitwise reproducible between two engines of one process?  (dev tool)  usage: dbg_repro.py [flags...]
flags: nolists (lists on the main stream), nolate (small launches on the main stream), nocache.  (This tool found the rare wrong
vectors of the removed k_tabp kernel: the first engine of a cold process differed from the next three in 3-8 table rows.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
N, B, T, H = 1_000_000, 512, 50, 150
flags = set(sys.argv[1:])
g = torch.Generator().manual_seed(5)
batches = []
for s in range(2):
    seq = torch.randint(1, N + 1, (B, T), generator=g, dtype=torch.int32)
    seq[:50, :30] = 0
    seq[60:120, -3:] = 777
    pos = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32)
    batches.append((seq.numpy(), pos.numpy()))
_lib.load()
outs = []
NE = 8 if "many" in flags else 4
for rep in range(NE):
    eng = Engine(N + 50, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, logits_dtype="x3")
    if "nolists" in flags:
        eng.lists_side_stream = False
    if "nolate" in flags:
        eng.late_side_stream = False
    if "nocache" in flags:
        eng.cache_descriptors = False
    snaps = []
    for seq, pos in batches:
        eng.train_step(seq, pos, N, 5e-4, rate=0.3)
        torch.cuda.synchronize()
        snaps.append((eng.theta.clone(), eng._act["rep"].clone(), eng._ws["drep"].clone(), float(eng.loss)))
    outs.append(snaps)
    del eng
    torch.cuda.empty_cache()
ref = outs[0]
for r in range(1, NE):
    msg = []
    for s in range(2):
        msg.append("step %d: theta %s rep %s drep %s loss %s" % (s, torch.equal(ref[s][0], outs[r][s][0]), torch.equal(ref[s][1], outs[r][s][1]),
                                                                torch.equal(ref[s][2], outs[r][s][2]), ref[s][3] == outs[r][s][3]))
    print(sorted(flags), "engine", r, "vs 0 |", " | ".join(msg), flush=True)
