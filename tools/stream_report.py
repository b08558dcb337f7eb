#!/usr/bin/env python3
"""Several images in flight on one GPU: what each stream's kernels do to each other, from a rocprofv3 --kernel-trace database.

    python tools/stream_report.py <results.db> [--images A B]      (window: from the A-th to the B-th prep_image_kernel of the trace)

Prints: the hardware queues / streams the dispatches came through; how long 0, 1, 2 ... kernels run side by side; per kernel class
(the large MFMA kernels, everything else) the mean duration alone vs with another LARGE kernel running beside it; the time a
stream's next kernel waits after its predecessor ended (the stream's own gaps); and the images per second of the traced span."""
import sqlite3
import sys

BIG = ("fc_mfma_dma16_kernel", "conv3x3_wino4_kernel", "fc_lowp_dma_kernel", "conv3x3_sw_kernel", "fc_x3_kernel<10", "fc_x3_kernel<5")


def short(name):
    return name.replace("void ", "").replace("mnc::", "").split("(")[0][:40]


def main():
    db = sys.argv[1]
    ia, ib = (int(sys.argv[sys.argv.index("--images") + 1]), int(sys.argv[sys.argv.index("--images") + 2])) if "--images" in sys.argv else (30, 130)
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + (", %s" % qcol if qcol else ", 0") + (", %s" % scol if scol else ", 0")
    rows = con.execute("select %s from kernels order by start" % sel).fetchall()
    preps = [s for n, s, e, q, st in rows if "prep_image_kernel" in n]
    t0, t1 = preps[min(ia, len(preps) - 2)], preps[min(ib, len(preps) - 1)]
    rows = [(short(n), s, e, q, st) for n, s, e, q, st in rows if t0 <= s < t1]
    span = max(r[2] for r in rows) - rows[0][1]
    nimg = sum(1 for r in rows if r[0].startswith("prep_image_kernel"))
    print("%d kernels over %.1f ms, %d images -> %.1f images/s in the traced span" % (len(rows), span / 1e6, nimg, nimg / (span / 1e9)))
    print("columns of `kernels`:", ", ".join(cols))
    for label, idx in (("queue", 3), ("stream", 4)):
        cnt = {}
        for r in rows:
            cnt[r[idx]] = cnt.get(r[idx], 0) + 1
        print("dispatches per %s: %s" % (label, ", ".join("%s: %d" % (k, v) for k, v in sorted(cnt.items(), key=lambda kv: str(kv[0])))))
    # concurrency histogram
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    depth, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    print("kernels running side by side: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    # big kernels: alone vs overlapped with another big kernel (by start-to-end interval intersection > 20 % of own duration)
    big = [r for r in rows if r[0].startswith(BIG)]
    big.sort(key=lambda r: r[1])
    stats = {}
    j0 = 0
    for i, r in enumerate(big):
        while j0 < len(big) and big[j0][2] <= r[1]:
            j0 += 1
        ov = 0
        for o in big[j0:]:
            if o[1] >= r[2]:
                break
            if o is r:
                continue
            ov += max(0, min(o[2], r[2]) - max(o[1], r[1]))
        d = r[2] - r[1]
        key = (r[0], "beside another large kernel" if ov > 0.2 * d else "alone")
        a = stats.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += d
    print("%-42s %-28s %7s %10s" % ("large kernel", "", "calls", "avg_us"))
    for (n, k), (c, t) in sorted(stats.items()):
        print("%-42s %-28s %7d %10.1f" % (n, k, c, t / c / 1e3))
    tot_big = sum(r[2] - r[1] for r in big)
    tot_all = sum(r[2] - r[1] for r in rows)
    print("sum of kernel durations per image: large %.3f ms, others %.3f ms; wall per image %.3f ms" %
          (tot_big / 1e6 / max(nimg, 1), (tot_all - tot_big) / 1e6 / max(nimg, 1), span / 1e6 / max(nimg, 1)))
    # per-stream gaps: time between a kernel's end and the next kernel's start on the same stream (or queue)
    idx = 4 if scol else 3
    by = {}
    for r in rows:
        by.setdefault(r[idx], []).append(r)
    gap_tot, gap_n, wait_big = 0.0, 0, 0.0
    for k, lst in by.items():
        lst.sort(key=lambda r: r[1])
        for a, b in zip(lst, lst[1:]):
            g = b[1] - a[2]
            if 0 < g < 2e6:
                gap_tot += g
                gap_n += 1
    print("gaps between consecutive kernels of one %s: %d gaps, %.3f ms per image" % ("stream" if scol else "queue", gap_n,
                                                                                      gap_tot / 1e6 / max(nimg, 1)))


if __name__ == "__main__":
    main()
