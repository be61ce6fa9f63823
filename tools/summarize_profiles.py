"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_hbm.json (HBM bytes per launch: FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
separate passes), profiles/<tag>_pmc_sq.json (per-launch SQ counters) and profiles/bench_<tag>.json.
usage: python tools/summarize_profiles.py r1g [bf16|x3]   (the second argument makes it the default profile bench.py quotes)"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(launches[k]), 1) for c, v in cs.items()} for k, cs in acc.items()}


shutil.copy(os.path.join(src, "stats", "k_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_%s.json" % tag))
fetch = counters(os.path.join(src, "pmc_fetch", "p_counter_collection.csv"))
write = counters(os.path.join(src, "pmc_write", "p_counter_collection.csv"))
hbm = {}
for k in sorted(set(fetch) | set(write)):
    if not (k.startswith("k_") or "k_" in k):
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
    w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    hbm[k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w, "hbm_bytes": 2 * f + w}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on `python3 bench.py "
                   "--steps 6 --warmup 2 --no-cpu-baseline --no-sections`; per-launch averages; counter unit KB (x1024 bytes); "
                   "gfx950: FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> "
                   "fetch_bytes_corrected = 2 x raw", "kernels": hbm},
          open(os.path.join(dst, "%s_pmc_hbm.json" % tag), "w"), indent=1)
sq = {k: v for k, v in counters(os.path.join(src, "pmc_sq", "p_counter_collection.csv")).items() if "k_" in k}
json.dump({"note": "rocprofv3 --pmc SQ_* (own pass, tools/profile_round.sh); per-launch sums over all waves.  SQ_WAVE_CYCLES, "
                   "SQ_WAIT_*, SQ_ACTIVE_INST_* are in units of 4 clocks; SQ_VALU_MFMA_BUSY_CYCLES in clocks summed over SIMDs.",
           "kernels": sq}, open(os.path.join(dst, "%s_pmc_sq.json" % tag), "w"), indent=1)
tcc = collections.defaultdict(dict)
for sub in ("pmc_tcc1", "pmc_tcc2", "pmc_tcc3"):
    f = os.path.join(src, sub, "p_counter_collection.csv")
    if os.path.exists(f):
        for k, v in counters(f).items():
            if "k_" in k:
                tcc[k].update(v)
if tcc:
    json.dump({"note": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum / TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum / "
                       "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum (three own passes); per-launch sums over the L2 channels.  RDREQ = read "
                       "requests the L2 sends to the fabric, RDREQ_DRAM = those routed to the memory controllers (the Infinity Cache "
                       "sits behind that interface: its hits are not separable at the L2)", "kernels": tcc},
              open(os.path.join(dst, "%s_pmc_tcc.json" % tag), "w"), indent=1)
# profiles/CURRENT.json: which summaries bench.py may quote by default (with provenance)
import subprocess
cur_p = os.path.join(dst, "CURRENT.json")
cur = json.load(open(cur_p)) if os.path.exists(cur_p) else {}
if len(sys.argv) > 2 and sys.argv[2] in ("bf16", "x3"):
    cur[sys.argv[2]] = tag
    try:
        cur["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"]).decode().strip()
    except Exception:
        pass
    sys.path.insert(0, ROOT)
    import bench
    cur["src_sha16"] = bench.csrc_sha16()        # bench.py quotes these counters only while the kernel sources are unchanged
    json.dump(cur, open(cur_p, "w"), indent=1)
print("wrote profiles/%s_*" % tag)
