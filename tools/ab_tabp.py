"""A/B of the x3 table update kernels on the bench workload (dev tool): python tools/ab_tabp.py <pipelined 0|1> [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd import _lib
from ader_amd.engine import Engine, SectionTimer
from bench import synth_batch
m = int(sys.argv[1])
_lib.load(); _lib.call("ader_x3_update_pipelined", m)
N, B, T = 1_000_000, 512, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="x3")
bs = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]
for i in range(8): eng.train_step(*bs[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for i in range(n): eng.train_step(*bs[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
eng.timer = SectionTimer(only={"logits_bwd_adam", "logits_fwd"}, every=4)
for i in range(40): eng.train_step(*bs[i % 4], N, 5e-4, rate=0.3)
sec = eng.timer.collect()
print("pipelined=%d ko=%s ms/step %.4f" % (m, os.environ.get("ADER_TP_KO", "-"), ms), {k: round(v, 4) for k, v in sec.items()}, flush=True)
