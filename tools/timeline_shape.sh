# one-step kernel timeline of a step shape (GPU box, repo root): bash tools/timeline_shape.sh <tag> <workload> [env...]  -> gpurun_out/<tag>/timeline_<workload>.txt
set -e
TAG=$1; W=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$W -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline \
    --no-sections --no-companion --no-herding --no-real-shapes --sustained-steps 0 --steps 20 --warmup 3 --workload $W --regime realistic > $OUT/tr_$W.log 2>&1 || true
F=$(find $OUT/tr_$W -name 'k_kernel_trace.csv' | head -n 1)
S=$(find $OUT/tr_$W -name 'k_kernel_stats.csv' | head -n 1)
[ -n "$F" ] && python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $F > $OUT/timeline_$W.txt
[ -n "$S" ] && cp $S $OUT/kernel_stats_$W.csv
rm -rf $OUT/tr_$W
cat $OUT/timeline_$W.txt
