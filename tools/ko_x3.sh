timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 60 -k "x3 or fused or hot_item or reproducible or fullsize or kd_fast" 2>&1 | tail -8
for r in 1 2; do
for ko in 0 64; do
ADER_X3_KO=$ko timeout 120 python3 bench.py --logits x3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ko=$ko', round(d['ms_per_step'],4), d['roofline']['sections_ms'], d['config']['final_loss'])"
done
done
ADER_HIP_LIB=$PWD/ader_amd/variants/libader_hip_t3stamp.so timeout 120 python3 tools/stamp_t3.py 2>&1 | tail -13
