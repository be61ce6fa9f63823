timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3 or fused or hot_item or reproducible or kd_fast" 2>&1 | tail -3
for r in 1 2; do
for ko in 0 1 2 4 5; do
ADER_X3_KO=$ko python3 bench.py --logits x3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ko=$ko', round(d['ms_per_step'],4), d['roofline']['sections_ms']['logits_bwd_adam'])"
done
done
