# Round profile bundle: bench line, rocprofv3 kernel stats, separate PMC passes (HBM bytes; SQ busy counters).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [extra bench.py arguments, e.g. --logits x3]
set -e
TAG=${1:-rX}
shift || true
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 40 --warmup 5 $EXTRA > $OUT/bench.json 2> $OUT/bench.err || true
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sections --no-companion --no-herding --no-real-shapes --sustained-steps 0 $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $B --steps 20 --warmup 3 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B --steps 6 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B --steps 6 --warmup 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $B --steps 6 --warmup 2 > $OUT/pmc_sq.log 2>&1
# where the L2's read requests go: all requests vs the ones routed to the memory controllers (separate passes: TCC counters are few)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc1 -o p -- $B --steps 6 --warmup 2 > /dev/null 2>&1 || true
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc2 -o p -- $B --steps 6 --warmup 2 > /dev/null 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc3 -o p -- $B --steps 6 --warmup 2 > /dev/null 2>&1 || true
ls $OUT/*
