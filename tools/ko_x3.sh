timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 60 -k "x3 or fused or hot_item or reproducible or fullsize or kd_fast" 2>&1 | tail -3
for r in 1 2 3; do
for ko in 0 32; do
ADER_X3_KO=$ko timeout 120 python3 bench.py --logits x3 --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --no-companion --no-herding 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ko=$ko', round(d['ms_per_step'],4), d['roofline']['sections_ms']['logits_bwd_adam'])"
done
done
