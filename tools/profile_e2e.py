"""Host-side profile of one continual-learning period on DIGINETICA (dev tool): where does the wall time go?"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ader_amd import main as M
args = M.build_parser().parse_args(["--dataset", "DIGINETICA", "--max_periods", "3", "--logits_dtype", "bf16", "--results_root", "gpurun_out/prof_e2e"])
pr = cProfile.Profile()
pr.enable()
M.run(args, log=lambda *a: None)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
