#!/usr/bin/env python
"""How much of the pole-flow sweep launches' time other kernels run beside them (rocprofv3 --kernel-trace rocpd DB of the
DEFAULT bench command: 2 contexts in flight). VERDICT r05 item 3 asks whether the 0.31-VALU-busy pole sweeps
(k_sweep_quad<true, 4>) need to be interleaved with the side sweeps inside one context: this is the trace that says how
much of them already runs beside another context's kernels, and beside which.
For every dispatch of the kernels matching <pattern> inside the timed part of the trace (the last <frac> of the trace's
time, default 0.5: past the check-free warm-up): the share of its duration during which >= 1 dispatch of ANOTHER queue
is executing, split by that other kernel's family.
Usage: python tools/sweep_overlap.py <results.db> ["k_sweep_quad<true, 4>"] [frac]"""
import sqlite3
import subprocess
import sys


def family(name):
    for key, fam in (("k_sweep_quad<true, 3>", "side sweep"), ("k_sweep_quad<true, 4>", "pole sweep"), ("k_sweep_lock", "latency sweep"),
                     ("k_median5", "median"), ("k_sepblur", "blur / diffusion / gradients"), ("k_resize", "resize / pyramid"),
                     ("k_remap", "remap"), ("k_iir", "sharpen"), ("k_novel_view", "novel view")):
        if key in name:
            return fam
    return "other"


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_sweep_quad<true, 4>"
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute("select k.kernel_name, d.start, d.end, %s from %s d join %s k on d.kernel_id = k.id order by d.start" % (
        "d." + qcol if qcol else "0", kd, ks)).fetchall()
    names = sorted(set(r[0] for r in rows))
    try:
        dm = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                                            capture_output=True, text=True, check=True).stdout.split("\n")))
    except Exception:  # noqa: BLE001
        dm = {n: n for n in names}
    rows = [(dm[n], s, e, q) for n, s, e, q in rows if "s360::" in dm[n]]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t1 - frac * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    targets = [r for r in rows if pat in r[0]]
    others = rows
    tot = covered = 0
    by_fam = {}
    j0 = 0
    for name, s, e, q in targets:
        tot += e - s
        # union of the other queues' dispatches over [s, e)
        while j0 < len(others) and others[j0][2] <= s and others[j0][1] < s - 50_000_000:
            j0 += 1
        iv = []
        for n2, s2, e2, q2 in others[j0:]:
            if s2 >= e:
                break
            if q2 == q or e2 <= s:
                continue
            a, b = max(s, s2), min(e, e2)
            iv.append((a, b))
            by_fam[family(n2)] = by_fam.get(family(n2), 0) + (b - a)
        iv.sort()
        end = s
        for a, b in iv:
            if b > end:
                covered += b - max(a, end)
                end = b
    print("# %s: %d launches, %.1f ms in the last %.0f %% of the trace; queue column: %s" % (pat, len(targets), tot / 1e6, 100 * frac, qcol))
    print("share of their time with >= 1 kernel of another queue executing beside them: %.3f" % (covered / max(tot, 1)))
    print("kernel-time of the other queues inside those launches, by family (ms; families overlap each other, so the sum may exceed the launches' time):")
    for fam, ns in sorted(by_fam.items(), key=lambda kv: -kv[1]):
        print("  %-30s %10.1f   (%.2f of the launches' time)" % (fam, ns / 1e6, ns / max(tot, 1)))


if __name__ == "__main__":
    main()
