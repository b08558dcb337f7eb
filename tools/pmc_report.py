#!/usr/bin/env python3
"""Derive per-kernel figures from rocprofv3 --pmc passes (rocpd SQLite):
   MfmaUtil%  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE_per_XCD * 1024 SIMDs)      (rocprofv3's own MfmaUtil formula;
               this build reports GRBM_GUI_ACTIVE summed over the 8 XCDs, hence the /8)
   clock GHz  = GRBM_GUI_ACTIVE / 8 / duration
   HBM bytes  = FETCH_SIZE KB * 1024 (x2 for wide coalesced reads on gfx950, MI355X_MICROARCH.md "HBM") + WRITE_SIZE KB * 1024
 usage: pmc_report.py <mfma.db> <fetch.db> <write.db> [--json out.json] [--build HASH] [--cycle KERNEL_SUBSTRING=N ...]
   --build HASH      recorded in the JSON as "_build" (python -m mnc_amd._build --hash): bench.py reports counter traffic only when
                     the profile was taken on the build it runs
   --cycle NAME=N    the kernel whose name contains NAME is launched N times per image in a fixed order (the InnerProduct kernel:
                     fc6_maskest, fc6, fc7, fc6_mask, fc7_mask x 2 stages = 10): its dispatches are also averaged per position in
                     that cycle ("by_position"), so that shapes sharing one template instantiation get their own traffic"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? "
         "group by kernel_name")
    return {r[0]: (r[1], r[2], r[3]) for r in con.execute(q, (counter,))}


def per_position(db, counter, name_part, period):
    """[avg value of dispatch position p in the kernel's cycle of `period` launches]  (dispatch order = dispatch id order)"""
    con = sqlite3.connect(db)
    cols = [c[1] for c in con.execute("pragma table_info('counters_collection')")]
    order = next((c for c in ("dispatch_id", "start", "start_timestamp", "id") if c in cols), None)
    q = ("select kernel_name, value, duration from counters_collection where counter_name=? and kernel_name like ?"
         + (" order by %s" % order if order else ""))
    rows = list(con.execute(q, (counter, "%" + name_part + "%")))
    if not rows or len(rows) % period:
        return None
    out = [[0.0, 0.0, 0] for _ in range(period)]
    for i, (_, v, d) in enumerate(rows):
        o = out[i % period]
        o[0] += v; o[1] += d; o[2] += 1
    return [(a / n, b / n) for a, b, n in out]


def short(n):
    return n.replace("void ", "").replace("mnc::", "").split("(")[0]


def main():
    mfma_db, fetch_db, write_db = sys.argv[1:4]
    busy = per_kernel(mfma_db, "SQ_VALU_MFMA_BUSY_CYCLES")
    act = per_kernel(mfma_db, "GRBM_GUI_ACTIVE")
    conf = per_kernel(mfma_db, "SQ_LDS_BANK_CONFLICT")
    fetch = per_kernel(fetch_db, "FETCH_SIZE")
    write = per_kernel(write_db, "WRITE_SIZE")
    out = {}
    print("%-34s %6s %10s %9s %9s %12s %12s %12s" % ("kernel", "calls", "avg_us", "MfmaUtil%", "clk_GHz", "fetch_MB", "write_MB",
                                                       "ldsconf/CU"))
    for k in sorted(act, key=lambda k: -act[k][0] * act[k][2]):
        n, a, dur = act[k]
        b = busy.get(k, (0, 0, 0))[1]
        util = 100.0 * b / (a / 8.0 * 1024.0) if a else 0.0
        clk = a / 8.0 / dur if dur else 0.0
        f = fetch.get(k, (0, 0, 0))[1] * 1024 / 1e6
        w = write.get(k, (0, 0, 0))[1] * 1024 / 1e6
        c = conf.get(k, (0, 0, 0))[1] / 256.0
        print("%-34s %6d %10.1f %9.1f %9.2f %12.2f %12.2f %12.0f" % (short(k)[:34], n, dur / 1e3, util, clk, f, w, c))
        out[short(k)] = {"calls": n, "avg_us": dur / 1e3, "mfma_util_pct": util, "clock_ghz": clk,
                         "fetch_bytes_raw": f * 1e6, "write_bytes": w * 1e6,
                         "hbm_bytes_corrected": 2 * f * 1e6 + w * 1e6}
    for arg in sys.argv:
        if arg.startswith("--cycle="):
            name_part, period = arg[len("--cycle="):].split("=")
            period = int(period)
            f = per_position(fetch_db, "FETCH_SIZE", name_part, period)
            w = per_position(write_db, "WRITE_SIZE", name_part, period)
            b = per_position(mfma_db, "SQ_VALU_MFMA_BUSY_CYCLES", name_part, period)
            a = per_position(mfma_db, "GRBM_GUI_ACTIVE", name_part, period)
            if not (f and w and b and a):
                print("# --cycle %s=%d: launch count is not a multiple of the period; no per-position table" % (name_part, period))
                continue
            print("# per position in the %d-launch cycle of *%s*:" % (period, name_part))
            pos = []
            for i in range(period):
                util = 100.0 * b[i][0] / (a[i][0] / 8.0 * 1024.0) if a[i][0] else 0.0
                print("#   %2d  avg_us %8.1f  MfmaUtil %5.1f %%  fetch %8.2f MB  write %8.2f MB"
                      % (i, a[i][1] / 1e3, util, f[i][0] * 1024 / 1e6, w[i][0] * 1024 / 1e6))
                pos.append({"avg_us": a[i][1] / 1e3, "mfma_util_pct": util,
                            "hbm_bytes_corrected": 2 * f[i][0] * 1024 + w[i][0] * 1024})
            for k in out:
                if name_part in k:
                    out[k]["by_position"] = pos
    if "--build" in sys.argv:
        out["_build"] = sys.argv[sys.argv.index("--build") + 1]
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
