#!/usr/bin/env python3
"""Summarise rocprofv3's rocpd SQLite output (bench_results.db): per-kernel calls / total / average duration, and,
when the run collected counters (--pmc), the per-kernel counter sums.  Used to write profiles/*.txt.

    python tools/rocpd_summary.py <results.db> [--pmc]"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("mnc::", "")
    return name.split("(")[0][:48]


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-50s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, c, s, a, mn, mx in rows:
        print("%-50s %7d %12.1f %12.2f %12.2f %12.2f %6.2f" % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    if "--pmc" in sys.argv:
        cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
        print("\ncounters_collection columns:", cols)
        q = ("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name")
        try:
            for n, cn, c, s, a in cur.execute(q):
                print("%-50s %-28s n=%-6d sum=%-16.6g avg=%-14.6g" % (short(n), cn, c, s, a))
        except sqlite3.OperationalError as e:
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
