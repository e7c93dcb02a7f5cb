"""The N>1 path without GPUs, through the NATIVE exchange code (surround360_amd/csrc/comm.cpp), not a Python twin:

* the policy layer (surround360_amd/parallel.py: pair blocks, pole-unit owners, which rank assembles which eye);
* host/TestRenderStereoPanorama --num_gpus 2 / 3 / 8 linked against the emulated library (tools/emu/, ranks = host
  threads, the RCCL stand-in of tools/hip_wave_shim/rccl_emu.cpp): s360_frame_exchange_strips with unequal and empty
  blocks, pole units spread over the ranks, s360_frame_gather_pole_layers, temporal state per owner over two chained
  frames — every file equal to the REFERENCE program's (tests/golden/refprogram_golden.json);
* world_size 2 and 3 as separate PROCESSES under torch.distributed (`gloo`, 127.0.0.1), the way bench.py runs on a
  node: tests/mp_sharded_frame.py, the emulated RCCL talking through files."""
import json
import os
import subprocess
import sys

import pytest

import refprog
import rigutil
from surround360_amd import parallel

ROOT = refprog.ROOT


def test_partition_pairs():
    assert parallel.partition_pairs(14, 8) == [0, 2, 4, 6, 8, 10, 12, 13, 14]
    assert parallel.partition_pairs(14, 1) == [0, 14]
    assert parallel.partition_pairs(14, 4) == [0, 4, 8, 11, 14]
    b = parallel.partition_pairs(14, 16)  # more ranks than pairs: trailing ranks get nothing
    assert b[-1] == 14 and all(0 <= b[i + 1] - b[i] <= 1 for i in range(16))


def test_pole_unit_assignment_matches_survey_8e():
    assert parallel.pole_owners(8) == [0, 1, 2, 3]  # "pole unit u -> devices 0-3"
    assert parallel.pole_owners(4, pole_removal=True) == [0, 1, 2, 2]  # the merged bottom image is computed once
    assert parallel.pole_owners(2) == [0, 0, 1, 1] and parallel.pole_owners(3) == [0, 0, 1, 1]
    assert parallel.pole_owners(1) == [0, 0, 0, 0]
    assert parallel.pole_owners(8, enable_top=False) == [-1, -1, 2, 3]
    own = parallel.pole_owners(8)
    assert parallel.unit_masks(own, 8) == [1, 2, 4, 8, 0, 0, 0, 0]
    assert parallel.strip_needs(own, 8) == [3, 2, 1, 2, 0, 0, 0, 0]  # the root composites both eyes; unit u reads eye u & 1
    assert parallel.strip_needs(parallel.pole_owners(2), 2) == [3, 3]


@pytest.fixture(scope="module")
def emu_programs():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "libs360_emu.so", "emu_programs"])
    return os.path.join(ROOT, "tools", "emu")


@pytest.mark.parametrize("name,gpus", [("two_frames", 2), ("two_frames", 3), ("two_frames", 8), ("pole_removal", 4)])
def test_sharded_program_writes_what_the_reference_program_writes(tmp_path, emu_programs, name, gpus):
    """--num_gpus G: pairs in blocks (8 ranks: 2,2,2,2,2,2,1,1), the strips exchanged once, the pole units on their owners
    (with their temporal state across the chained frames), the warped layers gathered, the composite on rank 0 — the 140
    (134) files of the case, digest for digest the reference program's. The RCCL stand-in runs in its STRICT mode
    (EMU_RCCL_STRICT=1: no buffering — a send completes only against a receive its peer has posted in a group executing at the same
    time, so ranks whose groups are ordered differently time out here instead of hanging on a real node; the probe below shows
    that it tells the two apart)."""
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  refprog.CAM / 2048.0)
    out = refprog.run_case(os.path.join(emu_programs, "TestRenderStereoPanorama"), str(tmp_path), rig, name,
                           more_args=["--num_gpus", str(gpus)], env={"EMU_DEVICES": str(gpus), "EMU_RCCL_STRICT": "1"})
    got = refprog.digests(out, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    assert sorted(got) == sorted(golden)
    differing = sorted(k for k in golden if got[k] != golden[k])
    assert not differing, "%d of %d files differ from the reference program's: %s" % (len(differing), len(golden), differing[:12])


def test_rccl_stand_in_strict_mode_detects_mismatched_group_order(tmp_path):
    """tests/probes/rccl_strict_probe.cpp: two ranks that each send in one group and receive in the next pass the buffered mailbox —
    and hang on real RCCL; the strict mode reports them. Send and receive in one group pass both."""
    exe = str(tmp_path / "probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "tools", "hip_wave_shim"),
                           os.path.join(ROOT, "tests", "probes", "rccl_strict_probe.cpp"),
                           os.path.join(ROOT, "tools", "hip_wave_shim", "rccl_emu.cpp"), "-o", exe])
    env = {k: v for k, v in os.environ.items() if not k.startswith("EMU_RCCL")}
    strict = dict(env, EMU_RCCL_STRICT="1", EMU_RCCL_STRICT_SECONDS="1")
    assert subprocess.run([exe, "good"], env=env).returncode == 0
    assert subprocess.run([exe, "bad"], env=env).returncode == 0       # the buffered stand-in cannot see it
    assert subprocess.run([exe, "good"], env=strict).returncode == 0
    assert subprocess.run([exe, "bad"], env=strict).returncode != 0    # the strict one does


@pytest.mark.parametrize("world", [3])  # (world 2 as processes: test_bench_script_two_ranks_dry_run below)
def test_sharded_frame_across_processes_gloo(tmp_path, emu_programs, world):
    port = 29500 + os.getpid() % 2000 + world
    env = dict(os.environ, EMU_RCCL_DIR=str(tmp_path), EMU_RCCL_STRICT="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "mp_sharded_frame.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "SHARDED_FRAME_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _check_two_rank_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert "dry_run" in d and "errors" not in d
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1 and d["warmup"] == 0
    assert d["checked"] is True
    assert d["checked_frames"] == 2 and d["mismatching_frames_all_ranks"] == 0  # 1 frame in flight per rank, both ranks counted
    # configs[3] as a first-class key; rccl_ranks is what the library's communicator says (ncclCommCount through s360_comm_size,
    # here read back from the RCCL stand-in), and config.rccl_ranks is that number, not WORLD_SIZE
    sf = d["sharded_frame"]
    assert "error" not in sf and sf["checked"] is True and sf["equals_single_gpu_frame"] is True
    assert sf["rccl_ranks"] == 2 and d["config"]["rccl_ranks"] == 2 and sf["ms"] > 0
    for key in ("exchange_strips", "exchange_pole_layers"):
        x = sf[key]
        assert x["bytes_moved"] > 0 and len(x["ms_per_rank"]) == 2 and x["xgmi_link_peak_GBps"] == 153.0
        assert sum(x["bytes_sent_per_rank"]) == sum(x["bytes_received_per_rank"]) == x["bytes_moved"]
    # two ranks: each assembles one eye of the poles -> rank 0 sends its block of the right eye, rank 1 its block of both
    assert sf["exchange_pole_layers"]["bytes_sent_per_rank"][0] == 0 and sf["exchange_pole_layers"]["bytes_received_per_rank"][1] == 0
    # configs[4] on N GPUs: one stream per rank
    v = d["video_stream"]
    assert "error" not in v and v["streams"] == 2 and len(v["ms_per_frame_of_each_stream"]) == 2 and v["frames_per_s"] > 0
    return d


BENCH_2 = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--slots", "1", "--inflight", "1", "--video-frames", "14"]


def _bench_env(tmp_path):
    return dict(os.environ, EMU_RCCL_DIR=str(tmp_path), EMU_RCCL_STRICT="1", EMU_DEVICES="2", S360_TEST_EMULATED_LIB="1",
                S360_BENCH_BACKEND="gloo", S360_BENCH_DEVICE="0", OMP_NUM_THREADS="1")


def test_bench_script_starts_itself_for_two_gpus(tmp_path, emu_programs):
    """`python bench.py --gpus 2` with NO launcher (the form of the driver's recorded command, BENCH_rNN.json "cmd"): the script
    re-executes itself under torch.distributed.run with one rank per GPU and rank 0 prints the one line. On the emulated library
    (S360_TEST_EMULATED_LIB=1: NOT a measurement; gloo for the timing reductions, the RCCL stand-in between the two processes)."""
    env = _bench_env(tmp_path)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_2, capture_output=True, text=True, env=env,
                       timeout=1500, cwd=ROOT)
    _check_two_rank_line(r)


@pytest.mark.skipif(os.environ.get("S360_RUN_SLOW") != "1", reason="the same line under an explicit launcher: opt-in (S360_RUN_SLOW=1)")
def test_bench_script_two_ranks_dry_run(tmp_path, emu_programs):
    """bench.py the way the contract launches it for N > 1 — `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2`: the same line as the self-started form above."""
    port = 31500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + BENCH_2, capture_output=True, text=True,
                       env=_bench_env(tmp_path), timeout=1500, cwd=ROOT)
    _check_two_rank_line(r)


def test_two_streams_on_two_gpus_equal_two_single_runs(tmp_path, emu_programs):
    """BASELINE configs[4] on N GPUs, the honest form: a stream cannot use more than one GPU, N streams use
    N. host/TestRenderStereoPanorama --num_frames 3 --num_streams 2 on two (emulated) devices renders frames 7-8 as a stream
    on device 0 and frame 9 as a stream of its own on device 1 — and writes, file for file, what two separate invocations
    with those frame ranges write: equirects of all three frames, the state files behind the last frame of each stream."""
    refprog.check_two_streams(os.path.join(emu_programs, "TestRenderStereoPanorama"), tmp_path, dict(os.environ, EMU_DEVICES="2"))


@pytest.mark.parametrize("streams", [2] + ([3] if os.environ.get("S360_RUN_SLOW") == "1" else []))
def test_streams_sharing_a_gpu_are_frame_slots_of_one_context(tmp_path, emu_programs, streams):
    """host/TestRenderStereoPanorama --num_streams S --stream_gpus 1: the streams are the frame slots of ONE context, frame k of
    every stream that still has one in one launch sequence (s360_frame_render_slots) with each stream's own device-resident
    temporal state — the reference's real workload (every preset chains frames, batch_process_video.py:157-158) as a batch. File
    for file what S separate --num_frames invocations write; 2 streams of unequal length (the second step renders a subset of the
    slots), 3 streams of one frame."""
    refprog.check_two_streams(os.path.join(emu_programs, "TestRenderStereoPanorama"), tmp_path, dict(os.environ, EMU_DEVICES="2"),
                              more_args=["--stream_gpus", "1"], streams=streams)


def test_hardware_day_checklist_walks_on_the_emulation(emu_programs):
    """tools/gpu_multi.sh's checklist for an N-GPU node (tools/multi_gpu_check.py: which librccl, comm_init_all + loopback on every
    rank, a two-frame sharded render against the reference program's digests; on hardware also bench.py --gpus N), its first
    three steps on two emulated devices with the strict RCCL stand-in: the script itself must not be what fails on the day."""
    import sys
    e = dict(os.environ, S360_TEST_EMULATED_LIB="1", EMU_DEVICES="2", EMU_RCCL_STRICT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_gpu_check.py"), "2"], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and "all steps passed on 2 GPUs" in r.stdout and "0 differ" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
