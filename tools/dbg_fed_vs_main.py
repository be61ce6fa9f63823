"""dev: per-period metrics of the fed and the torch-assembled main loop side by side (first periods)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ader_amd import main as M
base = sys.argv[1:] or ["--dataset", "YOOCHOOSE", "--lambda_", "1.0", "--max_periods", "3"]
for fed in ("True", "False"):
    with tempfile.TemporaryDirectory() as d:
        args = M.build_parser().parse_args(base + ["--fed_steps", fed, "--results_root", d])
        out = M.run(args, log=lambda s="": None)
        print("fed=%s" % fed, [(p["period"], p["best_epoch"], round(100 * p["recall20"], 3), round(100 * p["mrr20"], 3)) for p in out["periods"]], flush=True)
