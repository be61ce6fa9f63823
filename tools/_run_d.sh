python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops_autograd.py -x -q -m gpu -k "kd or distill" 2>&1 | tail -3 > gpurun_out/r4w.txt
python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "kd or distilled" 2>&1 | tail -3 >> gpurun_out/r4w.txt
A="--exemplars 128 --steps 20 --warmup 5 --no-cpu-baseline --no-companion --no-herding --sustained-steps 0"
python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ex128', round(d['ms_per_step'],4), d['roofline']['sections_ms'], d['config']['final_loss'])" >> gpurun_out/r4w.txt
