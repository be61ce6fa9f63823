"""Herding measurement: exemplar selection of one DIGINETICA period on the GPU (one batched encode + one
segmented herding launch over all label groups) against the C restatement of the reference loop (oracle/herding_ref.c,
single thread) on the same representations; also checks the selections are identical.  Measurement tool (bench-side use of
the oracle).    python tools/bench_herding.py [period]"""
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ader_amd.data import DataLoader, Sampler  # noqa: E402
from ader_amd.engine import Engine  # noqa: E402
from ader_amd.exemplar import ExemplarGenerator, herding_max_steps  # noqa: E402


def main():
    period = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    np.random.seed(0)
    random.seed(0)
    dl = DataLoader("DIGINETICA")
    sess, _ = dl.train_loader(period - 1)
    dl.evaluate_loader(period)
    N = dl.max_item()
    smp = Sampler(sess, 50, 256)
    valid, train = smp.split_data(valid_portion=0.1, return_train=True)
    cand = train + valid
    eng = Engine(43136, maxlen=50, hidden_units=150, num_blocks=2, num_heads=1, seed=0)
    gen = ExemplarGenerator(cand, 30000, False, 256, 50, 0.3, N)
    labels, offs, quota, rows = gen._segments()
    for _ in range(2):      # warm-up + timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sel, cnt = eng.herding_select(rows[:, :50], offs, quota, N)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
    rep = eng.encode(rows[:, :50]).cpu().numpy()
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libherding_ref.so"))
    lib.herding_ref.restype = ctypes.c_int
    lib.herding_ref.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    t0 = time.perf_counter()
    same, iters = 0, 0
    for g in range(len(labels)):
        lo, hi = int(offs[g]), int(offs[g + 1])
        r = np.ascontiguousarray(rep[lo:hi])
        out = np.zeros(max(hi - lo, 1), dtype=np.int32)
        steps = ctypes.c_int(0)
        k = lib.herding_ref(r.ctypes.data, hi - lo, 150, int(quota[g]), out.ctypes.data, ctypes.byref(steps))
        iters += steps.value
        same += int(k == int(cnt[g]) and np.array_equal(out[:k], sel[lo:lo + k]))
    t_cpu = time.perf_counter() - t0
    n_sel = int(cnt.sum())
    print(json.dumps({"period": period, "candidates": int(len(rows)), "label_groups": len(labels), "selected": n_sel,
                      "loop_iterations": iters, "gpu_seconds_encode_plus_select": round(t_gpu, 4),
                      "gpu_selections_per_s": round(n_sel / t_gpu, 1), "cpu_c_restatement_seconds_select_only": round(t_cpu, 3),
                      "cpu_selections_per_s": round(n_sel / t_cpu, 1), "groups_identical_to_cpu": same,
                      "max_steps_rule": "ceil(1.1*m) = %d for m = 10" % herding_max_steps(10)}))


if __name__ == "__main__":
    main()
