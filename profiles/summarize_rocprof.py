#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the small text summaries
committed under profiles/.

    python profiles/summarize_rocprof.py kernel  <kt_results.db>   # --kernel-trace --stats run
    python profiles/summarize_rocprof.py pmc     <pmc_results.db>  # --pmc run
"""
import sqlite3
import sys


def kernel_summary(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    print("%-110s %8s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-110s %8d %14.3f %14.3f %8.3f" % (name[:110], calls, total / 1e3, avg / 1e3, pct))
    print()
    print("per-dispatch (vpt kernels): name grid wg vgpr agpr sgpr lds scratch duration_us")
    q = ("select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, duration "
         "from kernels where name like '%vpt::%' order by start")
    for r in cur.execute(q):
        print("  %-70s %9d %4d %4d %4d %4d %6d %4d %12.3f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8] / 1e3))


def pmc_summary(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
         "where kernel_name like '%vpt::%' group by kernel_name, counter_name order by kernel_name, counter_name")
    last = None
    for k, c, v, n, d in cur.execute(q):
        if k != last:
            print("\n%s   (dispatches: %d, avg duration %.3f us)" % (k, n, d / 1e3))
            last = k
        print("  %-28s %18.0f   per-dispatch %16.1f" % (c, v, v / n))


if __name__ == "__main__":
    {"kernel": kernel_summary, "pmc": pmc_summary}[sys.argv[1]](sys.argv[2])
