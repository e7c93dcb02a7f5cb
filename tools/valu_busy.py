#!/usr/bin/env python
"""Per kernel: how busy the vector ALUs were — SQ_ACTIVE_INST_VALU against the kernel's own duration — from ONE rocprofv3
--kernel-trace --pmc pass over the SQ block (tools/gpu_round.sh's `sq` database). A kernel whose HBM fraction is low but whose
VALUs are busy is shown AT its bound instead of asserted to be there.
  VALU busy = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x duration x clock)      (rocprof's derived VALUBusy with the dispatch's own
  duration as the denominator; SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs: x 4 = cycles)
Usage: python tools/valu_busy.py <sq_results.db> [--json] [--clock-ghz 2.4] [--simds 1024]"""
import json
import sqlite3
import subprocess
import sys


def main():
    args = sys.argv[1:]
    as_json = "--json" in args
    clock = float(args[args.index("--clock-ghz") + 1]) if "--clock-ghz" in args else 2.4
    simds = int(args[args.index("--simds") + 1]) if "--simds" in args else 1024
    db = sqlite3.connect(args[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    rows = cur.execute(
        "select k.kernel_name, p.name, sum(e.value), d.id, d.end - d.start from %s e join %s p on e.pmc_id = p.id "
        "join %s d on e.event_id = d.event_id join %s k on d.kernel_id = k.id group by d.id, p.name" % (pe, ip, kd, ks)).fetchall()
    agg = {}
    seen = set()
    for name, ctr, val, did, ns in rows:
        a = agg.setdefault(name, {"launches": 0, "ns": 0})
        a[ctr] = a.get(ctr, 0.0) + val
        if did not in seen:
            seen.add(did)
            a["launches"] += 1
            a["ns"] += ns
    names = sorted(agg)
    try:
        dm = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names),
                                            capture_output=True, text=True, check=True).stdout.split("\n")))
    except Exception:  # noqa: BLE001
        dm = {n: n for n in names}
    out = {}
    for n, a in agg.items():
        if ("SQ_ACTIVE_INST_VALU" not in a and "SQ_INSTS_VALU" not in a) or a["ns"] <= 0:
            continue
        cyc = a["ns"] * clock  # ns x GHz = cycles
        rec = {"launches": a["launches"], "ms": round(a["ns"] / 1e6, 3)}
        if "SQ_ACTIVE_INST_VALU" in a:
            rec["valu_busy"] = round(a["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cyc), 4)
        if "SQ_INSTS_VALU" in a:  # executed wave-level VALU instructions
            rec["valu_insts_per_launch"] = round(a["SQ_INSTS_VALU"] / a["launches"], 1)
            for extra in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if extra in a:
                    rec[extra.lower()[3:] + "_per_launch"] = round(a[extra] / a["launches"], 1)
        if "SQ_ACTIVE_INST_LDS" in a:
            rec["lds_busy"] = round(a["SQ_ACTIVE_INST_LDS"] * 4.0 / (simds * cyc), 4)
        if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"] > 0:
            rec["wave_time_waiting"] = round(a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"], 3)
            rec["wave_time_issuing"] = round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"], 3)
        out[dm[n][:90]] = rec
    if as_json:
        print(json.dumps({"clock_ghz_assumed": clock, "simds": simds, "kernels": out}, indent=1))
        return
    print("# VALU busy = SQ_ACTIVE_INST_VALU x 4 / (%d SIMDs x duration x %.1f GHz); one SQ pass" % (simds, clock))
    print("%-72s %8s %10s %9s %9s %8s %8s" % ("kernel", "launches", "ms", "VALU busy", "LDS busy", "waiting", "issuing"))
    for n, r in sorted(out.items(), key=lambda kv: -kv[1]["ms"]):
        print("%-72s %8d %10.3f %9s %9s %8s %8s" % (n[:72], r["launches"], r["ms"], r.get("valu_busy", "-"), r.get("lds_busy", "-"),
                                                  r.get("wave_time_waiting", "-"), r.get("wave_time_issuing", "-")))


if __name__ == "__main__":
    main()
