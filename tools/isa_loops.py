#!/usr/bin/env python3
"""Static look at the loops of a gfx950 kernel: `hipcc -S` output in, one line per loop out.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S sweep_quad.hip -o sweep_quad.s
    python tools/isa_loops.py sweep_quad.s [--kernel k_sweep_quad] [--min-insts 100]

A loop is a backward branch (s_cbranch_* / s_branch to a label that was defined earlier in the same function); its body is the
text between the label and the branch. Per loop: instructions, VALU / SALU / LDS / global / scratch instructions, and the
`s_waitcnt vmcnt(..)` it contains — the throughput sweep kernel's steady steps must have none (loads and stores retire through one in-order
counter on gfx950: a wait inside the step waits for the next chunk's prefetches), and no scratch access.
Per kernel: VGPRs, scratch and LDS bytes, waves per SIMD from the metadata the compiler wrote. tests/test_cpu_isa.py holds k_sweep_quad to that (the latency kernel, k_sweep_lock, gathers from
global memory inside its step by design: its compute waves do wait there).
"""
import argparse
import re
import sys


def parse_functions(text):
    """-> {name: [lines]} for every function (kernel) of the device assembly."""
    funcs, cur, name = {}, None, None
    for ln in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not m.group(1).startswith(".L"):
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is not None:
            if ln.strip().startswith(".Lfunc_end") or ln.strip().startswith(".end_amdhsa_kernel"):
                cur = None
                continue
            cur.append(ln)
    return funcs


def classify(op):
    if op.startswith(("v_", "v_pk")):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def loops_of(lines):
    labels, insts = {}, []  # label -> index into insts; insts = (op, rest)
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", "//", ".")) and not s.startswith(".L"):
            continue
        m = re.match(r"^(\.L[\w$.]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        insts.append((parts[0], parts[1] if len(parts) > 1 else ""))
    out = []
    for i, (op, rest) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = rest.strip()
            if tgt in labels and labels[tgt] <= i:
                body = insts[labels[tgt]:i + 1]
                cnt = {"valu": 0, "salu": 0, "lds": 0, "global": 0, "scratch": 0, "other": 0}
                waits = []
                for bop, brest in body:
                    cnt[classify(bop)] += 1
                    if bop == "s_waitcnt" and "vmcnt" in brest:
                        waits.append(re.search(r"vmcnt\((\d+)\)", brest).group(1))
                out.append({"label": tgt, "start": labels[tgt], "end": i, "insts": len(body), **cnt, "vmcnt_waits": waits})
    # mark the loops that contain another loop
    for a in out:
        a["contains_loop"] = any(b is not a and a["start"] <= b["start"] and b["end"] <= a["end"] for b in out)
    return out


def kernel_meta(text):
    """-> {kernel symbol: {field: value}} from the .amdhsa_ directives / the resource-usage comments."""
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel\s+(\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        d = {}
        for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size", "accum_offset"):
            mm = re.search(r"\.amdhsa_%s\s+(\d+)" % k, m.group(2))
            if mm:
                d[k] = int(mm.group(1))
        meta[m.group(1)] = d
    # the comment block the compiler writes behind every kernel
    for m in re.finditer(r"\.Lfunc_end\d+:\s*\n\s*\.size\s+(\S+),.*?; Occupancy: (\d+)", text, re.S):
        meta.setdefault(m.group(1), {})["occupancy"] = int(m.group(2))
    for name in list(meta):
        blk = re.search(re.escape(name) + r".*?; codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+)", text, re.S)
        if blk:
            meta[name].update(code_bytes=int(blk.group(1)), vgprs=int(blk.group(2)), scratch=int(blk.group(3)))
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default="", help="substring of the (mangled) kernel names to report")
    ap.add_argument("--min-insts", type=int, default=60, help="loops shorter than this are not listed")
    args = ap.parse_args()
    text = open(args.asm).read()
    meta = kernel_meta(text)
    for name, lines in parse_functions(text).items():
        if args.kernel not in name or name not in meta:
            continue
        print("%s\n  %s" % (name, meta[name]))
        for lp in loops_of(lines):
            if lp["insts"] < args.min_insts:
                continue
            print("  loop %-10s insts %5d  valu %5d salu %4d lds %3d global %3d scratch %3d  vmcnt waits %s%s" % (
                lp["label"], lp["insts"], lp["valu"], lp["salu"], lp["lds"], lp["global"], lp["scratch"],
                lp["vmcnt_waits"] or "-", "  (contains a loop)" if lp["contains_loop"] else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
