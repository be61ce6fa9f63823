"""Host-side profile of a few continual-learning periods on DIGINETICA (dev tool): where does the wall time go?
usage: python tools/profile_e2e.py [flag=value ...]   (flags of ader_amd/main.py without the leading --)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ader_amd import main as M
argv = ["--dataset", "DIGINETICA", "--max_periods", "3", "--results_root", "gpurun_out/prof_e2e"]
for f in sys.argv[1:]:
    k, _, v = f.partition("=")
    argv += ["--" + k, v]
args = M.build_parser().parse_args(argv)
pr = cProfile.Profile()
pr.enable()
M.run(args, log=lambda *a: None)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("PROF_SORT", "cumulative")).print_stats(45)
print(s.getvalue()[:8000])
