"""python tools/run_main.py <ader_amd.main flags>: the continual-learning driver, runnable from any working directory
(rocprofv3 wants the program itself after `--`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from ader_amd.main import main  # noqa: E402

main(sys.argv[1:])
