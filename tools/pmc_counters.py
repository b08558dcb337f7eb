#!/usr/bin/env python3
"""Print per-kernel averages of whatever counters a rocprofv3 --pmc pass collected (rocpd SQLite), normalised per CU.
    python tools/pmc_counters.py <results.db> [<results.db> ...]"""
import sqlite3
import sys


def short(n):
    return n.replace("void ", "").replace("mnc::", "").split("(")[0][:40]


def main():
    rows = {}
    for db in sys.argv[1:]:
        con = sqlite3.connect(db)
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        for k, c, n, v, d in con.execute(q):
            rows.setdefault(short(k), {})[c] = (n, v, d)
    names = sorted({c for r in rows.values() for c in r})
    print("%-40s %8s " % ("kernel", "avg_us") + " ".join("%22s" % c[:22] for c in names))
    for k, r in sorted(rows.items(), key=lambda kv: -max(v[2] * v[0] for v in kv[1].values())):
        dur = max(v[2] for v in r.values())
        print("%-40s %8.1f " % (k, dur / 1e3) + " ".join("%22.4g" % (r[c][1] if c in r else float("nan")) for c in names))


if __name__ == "__main__":
    main()
