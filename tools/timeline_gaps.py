#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of one stream from a rocprofv3 --kernel-trace rocpd database: where the
difference between sum(kernel time) and wall time goes.

    python tools/timeline_gaps.py <results.db> [--top 25]"""
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").replace("mnc::", "").split("(")[0][:40]


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [c[1] for c in con.execute("pragma table_info('kernels')")]
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    print("columns:", cols)
    print("%d kernels, busy %.3f ms, span %.3f ms" % (len(rows), sum(r[2] - r[1] for r in rows) / 1e6,
                                                       (rows[-1][2] - rows[0][1]) / 1e6))
    gaps = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        key = (short(n0), short(n1))
        a = gaps.setdefault(key, [0, 0.0, 0.0, []])
        a[0] += 1
        a[1] += g
        a[2] = max(a[2], g)
        a[3].append(g)
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    print("%-40s -> %-40s %6s %10s %10s %10s" % ("after", "before", "n", "median_us", "avg_us", "max_us"))
    med = lambda v: sorted(v)[len(v) // 2]
    for (a, b), (n, tot, mx, vals) in sorted(gaps.items(), key=lambda kv: -med(kv[1][3]) * kv[1][0])[:top]:
        print("%-40s -> %-40s %6d %10.2f %10.2f %10.2f" % (a, b, n, med(vals) / 1e3, tot / n / 1e3, mx / 1e3))
    small = sum(v[1] for v in gaps.values() if v[1] / v[0] < 20e3)
    print("sum of gaps with avg < 20 us: %.3f ms over %d transitions" % (small / 1e6, sum(v[0] for v in gaps.values() if v[1] / v[0] < 20e3)))


if __name__ == "__main__":
    main()
