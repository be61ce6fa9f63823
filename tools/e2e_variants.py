"""Run several flag sets of the continual-learning driver back to back and print their `Average` metrics (dev tool).
usage: python tools/e2e_variants.py name1:flag,flag,... name2:...   (flags without the leading --, '=' between name and value)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ader_amd import main as M

for spec in sys.argv[1:]:
    name, _, fl = spec.partition(":")
    argv = []
    for f in filter(None, fl.split(",")):
        k, _, v = f.partition("=")
        argv += ["--" + k, v]
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        args = M.build_parser().parse_args(argv + ["--results_root", d, "--save_dir", name])
        out = M.run(args, log=lambda s="": None)
        a = out["average"]
        print("%-28s Recall@20 %.2f  MRR@20 %.2f  Recall@10 %.2f  MRR@10 %.2f  (%.0f s)" % (
            name, 100 * a["recall20"], 100 * a["mrr20"], 100 * a["recall10"], 100 * a["mrr10"], time.time() - t0), flush=True)
