#!/usr/bin/env python
"""profiles/valu_busy.json from tools/valu_busy.py --json outputs: per kernel FAMILY of bench.py (the names of its
`kernel_ms_per_frame` / `warp_blend_roofline` keys) the SQ "VALU busy" figure (the family's kernels weighted by their time) and,
when the pass counted SQ_INSTS_VALU, the executed VALU wave-instructions per launch of each kernel — what bench.py turns into
`valu_roof_frac` (executed VALU instructions of the family per frame / (SIMDs x measured issue rate x the family's time)).
Usage: python tools/valu_families.py <tag> <busy.json> [<insts.json>] > profiles/valu_busy.json"""
import json
import sys

RATE_CLASS = {  # what the family's arithmetic is made of (which issue rate is its roof)
    "project_side": "int", "project_pole": "int", "pole_warp": "int", "novel_view": "int", "flow_median": "int",
    "flatten": "int", "assemble_pano": "int", "flow_sweep": "fma", "flow_diffusion": "fma", "flow_blur15": "fma", "flow_upscale": "fma",
    "flow_gradients": "fma"}
FAMILIES = {  # family -> substrings of the (demangled) kernel names it launches
    "project_side": ["k_remap_cubic_u8c4_packed<s360::MapFromBuffer, 0"],
    "project_pole": ["k_remap_cubic_u8c4_packed<s360::MapFromBuffer, 1", "k_remap_cubic_u8c4_packed<s360::MapFromBuffer, 2"],
    "novel_view": ["k_novel_view"],
    "pole_warp": ["k_remap_cubic_u8c4_packed<s360::MapFromPoleFlow", "k_remap_pack<s360::MapFromPoleFlow", "k_pole_finish"],
    "flatten": ["k_flatten"],
    "assemble_pano": ["k_assemble_pano"],
    "flow_median": ["k_median5"],
    "flow_diffusion": ["k_sepblur<7, 2, 1,"],
    "flow_blur15": ["k_sepblur<7, 2, 2,", "k_sepblur<7, 2, 3,"],
    "flow_upscale": ["k_resize_cubic_f32c2"],
    "flow_gradients": ["k_sepblur<1, 2, 0, 1", "k_sepblur<1, 2, 0, 2"],
    "flow_sweep": ["k_sweep_quad"],
}


def main():
    tag = sys.argv[1]
    busy = json.load(open(sys.argv[2]))
    insts = json.load(open(sys.argv[3]))["kernels"] if len(sys.argv) > 3 else {}
    frames = sum(r["launches"] for n, r in insts.items() if "k_novel_view" in n) or None  # one launch per rendered frame
    out = {"issue_rate": {"clock_ghz": 2.4, "simds": busy["simds"],
                          "valu_inst_per_cycle_per_simd": {"fma": 0.43, "int": 0.25},
                          "source": "profiles/r05_v5_issue_rate_per_opcode.txt (tools/issue_rate, 8 waves per SIMD, every CU busy, "
                                    "wave-instructions per nominal cycle and SIMD): v_fma / v_add / v_mul_f32 0.43 (class fma: the doubled "
                                    "FP32 pipes), v_perm_b32 / v_dot2c_i32_i16 / v_min / v_max_f32 / integer / DPP 0.24-0.26 (class int)"},
           "source": "profiles/%s_valu_busy.txt (tools/valu_busy.py: rocprofv3 --pmc SQ passes of bench.py --inflight 1 --slots 22 "
                     "--no-extras; VALU busy = SQ_ACTIVE_INST_VALU x 4 / (%d SIMDs x the kernel's own duration x %.1f GHz); a "
                     "family's kernels weighted by their time; valu_insts_per_launch = SQ_INSTS_VALU (executed wave-level VALU "
                     "instructions) / launches)" % (tag, busy["simds"], busy["clock_ghz_assumed"]),
           "families": {}}
    for fam, pats in FAMILIES.items():
        ks = {n: r for n, r in busy["kernels"].items() if any(p in n for p in pats)}
        if not ks:
            continue
        ms = sum(r["ms"] for r in ks.values())
        rec = {"valu_busy": round(sum(r["valu_busy"] * r["ms"] for r in ks.values()) / ms, 3), "rate_class": RATE_CLASS.get(fam, "fma"),
               "kernels": sorted(ks)}
        # per instantiation: what the pass measured for that kernel alone (bench.py's roofline record shows k_sweep_quad's two)
        rec["per_kernel"] = {n: {k: r[k] for k in ("launches", "ms", "valu_busy", "lds_busy", "wave_time_waiting", "wave_time_issuing") if k in r}
                             for n, r in ks.items()}
        vi = {n: insts[n]["valu_insts_per_launch"] for n in ks if n in insts and "valu_insts_per_launch" in insts[n]}
        if vi:
            rec["valu_insts_per_launch"] = vi
            rec["launches_profiled"] = {n: insts[n]["launches"] for n in vi}
            if frames:
                rec["valu_insts_per_frame"] = round(sum(vi[n] * insts[n]["launches"] for n in vi) / frames, 1)
                rec["frames_profiled"] = frames
        out["families"][fam] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
