"""The engine's side stream is probed, not taken blindly from torch's pool: one of HIP's four high-priority hardware queues
answers cross-stream dependencies ~5x slower on MI355X (profiles/r5_packed/side_stream_queues.txt), and the 4th / 5th / 9th engine
of a process used to step 3x slower because of it."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_every_engine_of_a_process_steps_at_the_same_speed():
    import bench
    from ader_amd.engine import Engine, side_stream
    dev = torch.device("cuda:0")
    _, N, B, E = bench.REAL_SHAPES["cfgY"]
    T = 50
    seq, pos = bench.synth_batch(B, T, N, 5, dev, "realistic")
    ms = []
    for k in range(6):
        torch.cuda.Stream(device=dev, priority=-1)            # (advance torch's round-robin pool, as other users of it would)
        eng = Engine(N, maxlen=T, seed=0, device=dev)
        eng.pack_density = 0.1
        for i in range(10):
            eng.train_step(seq, pos, N, 5e-4, rate=0.3)
        best = None
        for rep in range(3):                                   # (best of three: the host of a shared box is not quiet)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(150):
                eng.train_step(seq, pos, N, 5e-4, rate=0.3)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 150 * 1e3
            best = dt if best is None else min(best, dt)
        ms.append(best)
        assert eng._side is side_stream(dev, torch.cuda.current_stream())
        eng.check_status()
    assert max(ms) < 1.6 * min(ms), ms                         # (an engine on the slow hardware queue stepped 3.2x slower)
