"""What the throughput sweep kernel's speed rests on and only the ISA shows, checked without a GPU: hipcc cross-compiles sweep_quad.hip with
the product's flags, tools/isa_loops.py reads the loops out of the assembly.
  * The steady step of the throughput kernel (two loops per build: steps 0..11 of a chunk and its last four) holds no wait on
    the memory counter — loads and stores retire in order through one counter on gfx950, so a wait there waits for the next
    chunk's prefetches (the build profiled as r03_v8 did, at 10 % of the sweep time) — and no scratch access.
  * The builds do not spill and keep two waves per SIMD. (A build of the same text held to three waves per SIMD by the register
    allocator — 168 VGPRs, its spills outside the steady loops — was measured in round 4 and removed: DESIGN.md section 5,
    profiles/r04_v5_*.)
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def quad_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "sweep_quad.s")
    # (the flags of surround360_amd/csrc/Makefile)
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "surround360_amd", "csrc", "sweep_quad.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernels(text, needle):
    import isa_loops
    meta = isa_loops.kernel_meta(text)
    return {name: (meta[name], isa_loops.loops_of(lines)) for name, lines in isa_loops.parse_functions(text).items()
            if needle in name and name in meta}


def _steady(loops):
    """innermost loops of the size of a step that read the window twice (two ds_read2_b64 per round, the records, the ring)"""
    return [lp for lp in loops if not lp["contains_loop"] and lp["insts"] >= 200 and lp["lds"] >= 8 and lp["global"] <= 1]


def test_steady_steps_without_memory_counter_waits(quad_asm):
    ks = _kernels(quad_asm, "k_sweep_quad")
    fast = {n: v for n, v in ks.items() if "ILb1E" in n}  # FAST = true: what the product launches
    assert len(fast) == 2, sorted(ks)  # <true, 3>, <true, 4>
    for name, (meta, loops) in fast.items():
        st = _steady(loops)
        assert len(st) == 2, (name, [(lp["label"], lp["insts"]) for lp in st])
        for lp in st:
            assert lp["vmcnt_waits"] == [] and lp["scratch"] == 0, (name, lp)
            assert lp["insts"] <= (320 if "Li3E" in name else 290), (name, lp)  # (279 / 304 when this was written)


def test_register_and_lds_budgets(quad_asm):
    for name, (meta, _) in _kernels(quad_asm, "k_sweep_quad").items():
        assert meta["private_segment_fixed_size"] == 0, (name, meta)  # nothing in scratch memory
        assert meta["next_free_vgpr"] <= 256 and meta["occupancy"] == 2, (name, meta)
        assert meta["group_segment_fixed_size"] <= 160 * 1024 // 8, (name, meta)  # eight waves per CU


def test_latency_kernel_keeps_nothing_in_scratch(tmp_path):
    """k_sweep_lock (round 5): the texel-independent part of errorFunction is pinned in front of the barrier through the barrier's asm
    operands. Passed as members of a struct behind a reference those eight operands became eight scratch slots — spill loads and
    stores inside the steady step; as scalar locals they are registers. Nothing of the latency kernel may live in scratch memory,
    and the pinned square root must sit between the step's gathers and its barrier."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / "sweep_lock.s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "surround360_amd", "csrc", "sweep_lock.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    ks = _kernels(text, "k_sweep_lockILi2ELb1")
    assert len(ks) == 1, sorted(ks)
    (name, (meta, loops)), = ks.items()
    assert meta["private_segment_fixed_size"] == 0 and all(lp["scratch"] == 0 for lp in loops), (name, meta)
    # between a pair of bilinear gathers (two global_load_dwordx4) and the next s_barrier of the steady step: a v_sqrt_f32
    import isa_loops
    lines = isa_loops.parse_functions(text)[name]
    ops = [ln.split()[0] for ln in lines if ln.strip() and not ln.strip().startswith((";", "."))]
    found = False
    for i in range(len(ops) - 1):
        if ops[i] == "global_load_dwordx4" and "global_load_dwordx4" in ops[i + 1:i + 3]:
            window = ops[i:i + 90]
            if "s_barrier" in window and "v_sqrt_f32_e32" in window[:window.index("s_barrier")]:
                found = True
                break
    assert found, "no square root between the gathers and the barrier of a step"
