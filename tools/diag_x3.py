import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import ader_ref_cpu as R
import test_gpu_parity as TP

cfg = TP.CFGS[2]
item_num, T, H, L, heads, B, N = cfg
res = {}
for gemm in ("f32", "x3"):
    eng = TP._engine(item_num, T, H, L, heads, seed=3, gemm=gemm)
    rs = np.random.RandomState(2)
    seq = TP._seqs(rs, B, T, N)
    pos = rs.randint(1, N + 1, size=B).astype(np.int32)
    eng.global_step = 4
    rate = float(os.environ.get("RATE", "0.3"))
    loss = eng.loss_and_grad(seq, pos, N, rate=rate)
    torch.cuda.synchronize()
    if gemm == "f32":
        oloss, og = R.loss_and_grads(TP._params(eng, torch.float64), seq, pos, N, L, heads, training=True, rate=rate, seed=3, step=4)
    print(gemm, "loss", float(loss.item()), float(oloss))
    for k in eng.layout:
        g = eng.gradient(k).cpu().numpy()
        e = TP.nerr(g, og[k].numpy(), floor=1e-4)
        res.setdefault(k, []).append(e)
for k, v in res.items():
    print("%-10s f32 %.2e   x3 %.2e" % (k, v[0], v[1]))
# where is the emb error?
