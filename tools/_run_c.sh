A="--exemplars 128 --steps 20 --warmup 5 --no-cpu-baseline --no-companion --no-herding --sustained-steps 0"
for v in base roko1 roko2 roko3; do
  if [ $v = base ]; then L=""; else L="ader_amd/variants/libader_hip_$v.so"; fi
  ADER_HIP_LIB=$L python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['roofline']['sections_ms'])"
done > gpurun_out/r4v_roko.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kd" 2>&1 | tail -3 >> gpurun_out/r4v_roko.txt
