"""One rank of a frame sharded over N processes, on the CPU (launched by tests/test_cpu_parallel.py, one process per
rank, `gloo`): the product's own multi-GPU path — s360_comm_init_rank, s360_frame_render_pairs,
s360_frame_exchange_strips, s360_frame_pole_units, s360_frame_gather_pole_layers, s360_frame_composite — through the
emulated library (tools/libs360_emu.so; its RCCL stand-in talks between processes through files in EMU_RCCL_DIR).
torch.distributed (gloo) does what it does in bench.py: it carries the communicator id. Rank 0 compares the sharded
frame with one context rendering the whole frame and prints SHARDED_FRAME_OK."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from surround360_amd import _capi
    _capi.LIB_PATH = os.path.join(ROOT, "tools", "libs360_emu.so")
    from surround360_amd import parallel, render as R
    import rigutil
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cam = 160
    rig_path = os.path.join(os.environ["EMU_RCCL_DIR"], "rig_%d.json" % rank)
    rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), rig_path, cam / 2048.0)
    rig = R.RigDescription(rig_path)
    flags = dict(eqr_width=504, eqr_height=252, enable_top=1, enable_bottom=1, final_eqr_width=480, final_eqr_height=480,
                 sharpening=0.25)
    frames = [rigutil.frame_inputs(rig_path, cam, yaw_deg=0.7 * k) for k in range(2)]  # the same inputs on every rank
    P = rig.get_side_camera_count()
    bounds = parallel.partition_pairs(P, world)
    owner = parallel.pole_owners(world)
    masks, need = parallel.unit_masks(owner, world), parallel.strip_needs(owner, world)
    ctx = R.Context(rig, R.make_params(**flags))
    ids = [R.Context.comm_get_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.comm_init_rank(ids[0], rank, world)
    ctx.set_partition(bounds[rank], bounds[rank + 1])
    for k, (side, top, bottom) in enumerate(frames):  # second frame: temporal state stays on the rank that owns the unit
        ctx.upload_frame(side, top if masks[rank] & 3 else None, bottom if masks[rank] & 12 else None)
        ctx.render_pairs(bounds[rank], bounds[rank + 1], k > 0)
        ctx.exchange_strips(bounds, need)
        if masks[rank] or rank == 0:
            ctx.pole_units(masks[rank], k > 0)
        ctx.gather_pole_layers(owner, 0)
        if rank == 0:
            ctx.composite(15)
    ok = True
    if rank == 0:
        got = ctx.download_equirect()
        one = R.Context(rig, R.make_params(**flags))
        for k, (side, top, bottom) in enumerate(frames):
            one.upload_frame(side, top, bottom)
            one.render(k > 0)
        want = one.download_equirect()
        ok = bool(np.array_equal(got, want)) and got.std() > 5
        one.close()
    dist.barrier()
    ctx.comm_destroy()
    ctx.close()
    if rank == 0:
        print("SHARDED_FRAME_OK" if ok else "SHARDED_FRAME_DIFFERS", "world", world, "bounds", bounds, "owners", owner)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
