"""Is the 3x slow block of bench.real_shape_line process state or the box?  Runs the two shapes alternately in one process and prints,
per call, the wall figure, the process-CPU / wall ratio of the call and the host's load average; `--keep-cache` skips empty_cache."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

keep = "--keep-cache" in sys.argv
dev = torch.device("cuda:0")
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "keep_cache", keep, flush=True)
for i in range(10):
    name = ("cfgY", "cfgD")[i % 2]
    w0, c0 = time.perf_counter(), time.process_time()
    o = bench.real_shape_line(name, dev, empty_cache=not keep)
    w1, c1 = time.perf_counter(), time.process_time()
    print("%2d %s median %.4f min %.4f reps %s | call %.1f s wall, cpu/wall %.2f | load %s" % (
        i, name, o["ms_per_step"], o["ms_per_step_min"], o["reps_ms"], w1 - w0, (c1 - c0) / (w1 - w0),
        " ".join("%.1f" % x for x in os.getloadavg())), flush=True)
