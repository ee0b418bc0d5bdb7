"""Turn rocprofv3's rocpd sqlite outputs (gpurun_out/prof_*/..._results.db) into the small CSV summaries kept under profiles/.
usage: python tools/rocpd_summary.py stats <db> <out.csv> | pmc <db> <out.csv> | bygrid <db> <out.csv>
(bygrid: average duration per (kernel, grid size) - tells the launches of one kernel on differently sized problems apart)"""
import csv
import sqlite3
import sys

mode, db, out = sys.argv[1:4]
c = sqlite3.connect(db)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    if mode == "stats":
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            w.writerow([name.split("(")[0][:80], calls, "%.1f" % total, "%.2f" % avg, "%.3f" % pct])
    elif mode == "bygrid":
        w.writerow(["kernel", "grid_size", "workgroup_size", "dispatches", "avg_us", "min_us", "max_us"])
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        gcol = "grid_size" if "grid_size" in cols else ("grid_x" if "grid_x" in cols else cols[0])
        wcol = "workgroup_size" if "workgroup_size" in cols else ("workgroup_x" if "workgroup_x" in cols else gcol)
        dur = "duration" if "duration" in cols else '("end" - start)'
        q = "select name,%s,%s,count(*),avg(%s),min(%s),max(%s) from kernels group by 1,2,3 order by 5*4 desc" % (gcol, wcol, dur, dur, dur)
        for name, gs, ws, n, a, lo, hi in c.execute(q):
            w.writerow([name.split("(")[0][:80], gs, ws, n, "%.2f" % (a / 1e3), "%.2f" % (lo / 1e3), "%.2f" % (hi / 1e3)])
    else:
        w.writerow(["kernel", "grid_size", "counter", "dispatches", "avg_value", "min_value", "max_value", "avg_duration_us"])
        q = ("select kernel_name,grid_size,counter_name,count(*),avg(value),min(value),max(value),avg(duration) from counters_collection "
             "group by 1,2,3 order by 1,2,3")
        for name, gs, cn, n, a, lo, hi, d in c.execute(q):
            w.writerow([name.split("(")[0][:80], gs, cn, n, "%.1f" % a, "%.1f" % lo, "%.1f" % hi, "%.1f" % (d / 1e3)])
