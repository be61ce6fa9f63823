for v in f g; do ADER_X3_FWD=$v timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3 or kd_fast or more_than_1024" 2>&1 | tail -2; done
for r in 1 2; do
for v in old f g; do
ADER_X3_FWD=$v python3 bench.py --logits x3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fwd=$v', round(d['ms_per_step'],4), d['roofline']['sections_ms']['logits_fwd'], d['config']['final_loss'])"
done
done
ADER_HIP_LIB=$PWD/ader_amd/variants/libader_hip_g3stamp.so ADER_X3_FWD=g python3 tools/stamp_x3.py 2>&1 | tail -9
