#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table: calls, total, avg, min, max, pct.
Usage: python tools/rocpd_kernel_stats.py gpurun_out/<dir>/<name>_results.db "header line" > profiles/<file>.txt"""
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True,
                             text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select k.kernel_name, d.end - d.start from %s d join %s k on d.kernel_id = k.id" % (kd, ks)).fetchall()
    agg = {}
    dm = demangle(sorted(set(r[0] for r in rows)))
    for name, ns in rows:
        name = dm.get(name, name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += ns
        a[2] = min(a[2], ns)
        a[3] = max(a[3], ns)
    tot = sum(a[1] for a in agg.values()) or 1
    for h in sys.argv[2:]:
        print("# " + h)
    print("%-64s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-64s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%" % (name[:64], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3,
                                                                a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == "__main__":
    main()
