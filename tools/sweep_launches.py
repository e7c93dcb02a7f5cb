#!/usr/bin/env python
"""Per-launch durations of the sweep kernel in a rocprofv3 rocpd DB (last frame only): grid, microseconds."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_sweep"
    per_frame = int(sys.argv[3]) if len(sys.argv) > 3 else 134
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s k on "
                       "d.kernel_id = k.id order by d.start" % (kd, ks)).fetchall()
    sw = [(r[3] // r[4], (r[2] - r[1]) / 1e3) for r in rows if pat in r[0]][-per_frame:]
    tot = 0.0
    for i in range(0, len(sw), 2):
        g, a = sw[i]
        b = sw[i + 1][1] if i + 1 < len(sw) else 0.0
        tot += a + b
        print("%6d wgs  fwd %9.1f us  bwd %9.1f us" % (g, a, b))
    print("total %.2f ms" % (tot / 1e3))


if __name__ == "__main__":
    main()
