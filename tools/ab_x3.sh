# A/B of the x3 table update (ADER_X3_UPDATE=old|new) on one box: parity subset, then alternating bench runs.  Dev tool.
set -x
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "x3 or fused or hot_item or reproducible or fullsize or kd_fast" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for r in 1 2; do
for v in tab32 tab16; do
ADER_X3_UPDATE=$v python3 bench.py --logits x3 --steps 40 --warmup 5 --no-cpu-baseline 2>$OUT/err_$v.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['roofline']['sections_ms'])"
done
done
