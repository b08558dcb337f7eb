#!/usr/bin/env python3
"""Several images in flight on one GPU, from a rocprofv3 --kernel-trace rocpd database: how busy the GPU is (union of the kernel
intervals over the span), how much of the time 1, 2, 3 ... kernels run side by side, and how much longer each kernel takes than
in a one-image-at-a-time run of the same build (a second database) -- where the distance between the sum of an image's kernels
and the step time goes.

    python tools/overlap_report.py <in_flight.db> [<serial.db>] [--skip-ms 400]"""
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").replace("mnc::", "").split("(")[0][:44]


def load(db, skip_ns):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    t0 = rows[0][1] + skip_ns                       # skip weight packing / warm-up
    return [(short(n), s, e) for n, s, e in rows if s >= t0]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = float(sys.argv[sys.argv.index("--skip-ms") + 1]) * 1e6 if "--skip-ms" in sys.argv else 0.0
    rows = load(args[0], skip)
    span = max(e for _, _, e in rows) - rows[0][1]
    ev = sorted([(s, 1) for _, s, _ in rows] + [(e, -1) for _, _, e in rows])
    depth, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    print("%d kernels over %.2f ms; sum of kernel time %.2f ms" % (len(rows), span / 1e6, sum(e - s for _, s, e in rows) / 1e6))
    for k in sorted(hist):
        print("  %d kernel(s) running: %7.2f ms  %5.1f %%" % (k, hist[k] / 1e6, 100.0 * hist[k] / span))
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    ser = {}
    if len(args) > 1:
        for n, s, e in load(args[1], skip):
            a = ser.setdefault(n, [0, 0])
            a[0] += 1
            a[1] += e - s
    print("%-46s %7s %10s %10s %8s" % ("kernel", "calls", "avg_us", "serial_us", "ratio"))
    tot_in = tot_ser = 0.0
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        s = ser.get(n)
        sa = s[1] / s[0] / 1e3 if s else float("nan")
        print("%-46s %7d %10.2f %10.2f %8.2f" % (n, c, t / c / 1e3, sa, (t / c / 1e3) / sa if s else float("nan")))
        if s:
            tot_in += t / c * (s[0] and 1) * 1.0
            tot_ser += s[1] / s[0]


if __name__ == "__main__":
    main()
