"""Multi-GPU sharding of one frame: the 14 adjacent side-camera pairs are independent units,
pair p lives on rank `p mod world` in contiguous blocks, and ONE exchange moves the rendered strips
(2 eyes x [camH][stripW] BGRA per pair) to the root, which assembles the panoramas and runs the pole
units. torch.distributed is only plumbing here: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""
import torch
import torch.distributed as dist


def partition_pairs(num_pairs, world):
    """Contiguous, balanced blocks: rank r renders pairs [bounds[r], bounds[r+1]). 14 pairs on 8 ranks ->
    2,2,2,2,2,2,1,1. Temporal state of a pair never migrates between ranks."""
    base, extra = divmod(num_pairs, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def gather_strips(strips, bounds, rank, world, root=0):
    """strips: tensor [2][P][camH][stripW][4] uint8 on every rank; rank r holds valid data for its own pairs.
    After the call the root's tensor is complete. Point-to-point sends of unequal block sizes, issued as one
    batch so that RCCL drives all of the root's inbound xGMI links concurrently."""
    if world == 1:
        return
    ops = []
    if rank == root:
        for r in range(world):
            if r == root or bounds[r + 1] == bounds[r]:
                continue
            for eye in range(2):
                ops.append(dist.P2POp(dist.irecv, strips[eye, bounds[r]:bounds[r + 1]], r))
    else:
        if bounds[rank + 1] > bounds[rank]:
            for eye in range(2):
                ops.append(dist.P2POp(dist.isend, strips[eye, bounds[rank]:bounds[rank + 1]], root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class _DevArray:
    """Zero-copy view of a HIP device pointer owned by libs360 for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def strips_tensor(ctx, device):
    """torch uint8 tensor [2][P][camH][stripW][4] aliasing the context's strip buffers."""
    g = ctx.geometry
    P = ctx.rig.get_side_camera_count()
    strip_w = ctx.params.eqr_width // P
    p0, _ = ctx.strip_ptr(0)
    return torch.as_tensor(_DevArray(p0, (2, P, g.cam_image_height, strip_w, 4)), device=device)
