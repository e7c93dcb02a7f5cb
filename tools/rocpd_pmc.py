#!/usr/bin/env python
"""Per-kernel sums of one derived PMC counter from a rocprofv3 rocpd DB (run with --kernel-trace --pmc NAME).
Usage: python tools/rocpd_pmc.py <db> <COUNTER> [kernel-substring]"""
import sqlite3
import subprocess
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    counter = sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    rows = cur.execute(
        "select k.kernel_name, sum(e.value), d.id from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id "
        "join %s k on d.kernel_id = k.id where p.name = ? group by d.id" % (pe, ip, kd, ks), (counter,)).fetchall()
    agg = {}
    for name, val, _ in rows:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += val
    names = sorted(agg)
    try:
        dm = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                                            capture_output=True, text=True, check=True).stdout.split("\n")))
    except Exception:
        dm = {n: n for n in names}
    print("%-70s %8s %16s %16s" % ("kernel", "launches", counter + " total", "per launch"))
    for n, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if pat and pat not in n:
            continue
        print("%-70s %8d %16.1f %16.2f" % (dm[n][:70], cnt, tot, tot / cnt))


if __name__ == "__main__":
    main()
