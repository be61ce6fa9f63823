# Per-shape evidence beside the round bundle: bench line + rocprofv3 kernel stats + one-step timeline for the ADER-mode headline and
# the two real-data step shapes.  usage (GPU box, repo root): bash tools/profile_shapes.sh <tag>   -> gpurun_out/<tag>_shapes/
set -e
TAG=${1:-rX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench arguments
    NAME=$1; shift
    python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-herding --no-real-shapes --sustained-steps 0 --steps 40 --warmup 5 "$@" \
        > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err || true
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$NAME -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline \
        --no-sections --no-companion --no-herding --no-real-shapes --sustained-steps 0 --steps 20 --warmup 3 "$@" > $OUT/tr_$NAME.log 2>&1 || true
    F=$(find $OUT/tr_$NAME -name 'k_kernel_trace.csv' | head -n 1)
    S=$(find $OUT/tr_$NAME -name 'k_kernel_stats.csv' | head -n 1)
    [ -n "$F" ] && python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $F > $OUT/timeline_$NAME.txt || true
    [ -n "$S" ] && cp $S $OUT/kernel_stats_$NAME.csv || true
    rm -rf $OUT/tr_$NAME
}
run cfgS
run ader128 --exemplars 128
run cfgD --workload cfgD --regime realistic
run cfgY --workload cfgY --regime realistic
ls $OUT
