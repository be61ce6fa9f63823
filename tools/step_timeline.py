"""Timeline of one train step from a rocprofv3 kernel trace (dev tool): python tools/step_timeline.py <k_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = next(k for k in ('k_feed_step', 'k_plan_len', 'k_seq_fwd') if any(r['Kernel_Name'].startswith(k) for r in rows))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(first)]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    print("%8.1f %8.1f %7.1f  q%s %s" % (s / 1000, e / 1000, (e - s) / 1000, r.get('Queue_Id', ''), r['Kernel_Name'][:70]))
print("# step period (start of the step's first session kernel to the next one): %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1000))
