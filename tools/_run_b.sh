python3 bench.py --exemplars 128 --steps 20 --warmup 5 --no-cpu-baseline --no-companion --no-herding --sustained-steps 0 > gpurun_out/r4u_bench_ex128.json 2> gpurun_out/r4u_bench_ex128.err
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4u
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --exemplars 128 --no-cpu-baseline --no-sections --no-companion --no-herding --sustained-steps 0 --steps 20 --warmup 3 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py $OUT/stats/k_kernel_trace.csv > gpurun_out/r4u_timeline_ex128.txt 2>&1
cp $OUT/stats/k_kernel_stats.csv gpurun_out/r4u_kernel_stats_ex128.csv
rm -rf $OUT
