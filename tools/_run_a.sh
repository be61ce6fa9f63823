ADER_HIP_LIB=ader_amd/variants/libader_hip_sfstamp.so python tools/stamp_sf.py > gpurun_out/r4_sf_stamps.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sections --no-companion --no-herding --sustained-steps 0"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4t
mkdir -p $OUT
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc1 -o p -- $B --steps 6 --warmup 2 > $OUT/t1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc2 -o p -- $B --steps 6 --warmup 2 > $OUT/t2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc3 -o p -- $B --steps 6 --warmup 2 > $OUT/t3.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, json, os, re
out = {}
for sub in ("pmc_tcc1", "pmc_tcc2", "pmc_tcc3"):
    for root, _, files in os.walk("gpurun_out/r4t/" + sub):
        for f in files:
            if f.endswith("counter_collection.csv"):
                acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
                for r in csv.DictReader(open(os.path.join(root, f))):
                    k = re.sub(r"^void ", "", r["Kernel_Name"])[:40]
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
                for k in acc:
                    out.setdefault(k, {}).update({c: v / len(n[k]) for c, v in acc[k].items()})
json.dump(out, open("gpurun_out/r4t_tcc.json", "w"), indent=1)
PY
tail -3 gpurun_out/r4t/t1.log
rm -rf gpurun_out/r4t/pmc_tcc*
