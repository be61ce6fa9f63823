for s in 1 0 1 0; do
ADER_LISTS_SIDE=$s python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side $s', round(d['ms_per_step'],4), d['roofline']['sections_ms'], round(d['roofline']['ms'],4))"
done
