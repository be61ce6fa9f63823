# SQ counters of the x3 forward variants (own PMC passes).  Dev tool.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_x3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --logits x3 --no-cpu-baseline --no-sections --steps 4 --warmup 1"
for v in old f g; do
export ADER_X3_FWD=$v
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/a_$v -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $OUT/b_$v -o p -- $B > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_x3"
for d in sorted(glob.glob(out+"/*")):
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: print(d,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "lx3" not in k or "prep" in k: continue
        k=k[:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc: print(os.path.basename(d), k, {c: round(v/len(n[k])/1e6,1) for c,v in acc[k].items()})
PY
