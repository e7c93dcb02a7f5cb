"""Multi-GPU sharding of one frame, the policy side. The data path is native: the ranks' strips and
warped pole layers travel through libs360's own grouped ncclSend / ncclRecv exchanges (surround360_amd/csrc/comm.cpp:
s360_frame_exchange_strips, s360_frame_gather_pole_layers); torch.distributed only carries the 128-byte communicator id
in bench.py. What lives here is who does what — the same assignment host/TestRenderStereoPanorama --num_gpus makes:
  * the 14 adjacent side-camera pairs in contiguous balanced blocks (temporal state of a pair never migrates);
  * the four pole units (0 top_left, 1 top_right, 2 bottom_left, 3 bottom_right; TRSP:811-860 runs them as four
    threads) on ranks 0-3, or the two poles on ranks 0 / 1 when there are fewer than four ranks;
  * per rank, the eyes whose complete strips it has to assemble (the root: both; a pole unit's owner: that unit's eye).
"""


def partition_pairs(num_pairs, world):
    """Contiguous, balanced blocks: rank r renders pairs [bounds[r], bounds[r+1]). 14 pairs on 8 ranks ->
    2,2,2,2,2,2,1,1. Temporal state of a pair never migrates between ranks."""
    base, extra = divmod(num_pairs, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def pole_owners(world, enable_top=True, enable_bottom=True, pole_removal=False):
    """owner[u] = rank that runs pole unit u, -1 for a unit that is not enabled. 4+ ranks: unit u on rank u (with pole
    removal both bottom units on rank 2, which then merges the two bottom cameras once); 2-3 ranks: top units on rank 0,
    bottom units on rank 1; 1 rank: everything on it."""
    owner = []
    for u in range(4):
        if not (enable_top if u < 2 else enable_bottom):
            owner.append(-1)
        elif world >= 4:
            owner.append(2 if (pole_removal and u == 3) else u)
        elif world >= 2:
            owner.append(0 if u < 2 else 1)
        else:
            owner.append(0)
    return owner


def unit_masks(owner, world):
    """Per rank, the bit mask of the pole units it runs (argument of s360_frame_pole_units)."""
    masks = [0] * world
    for u, r in enumerate(owner):
        if r >= 0:
            masks[r] |= 1 << u
    return masks


def strip_needs(owner, world, root=0):
    """Per rank, the eyes (bit 0 left, bit 1 right) whose complete strips it assembles (need_mask of
    s360_frame_exchange_strips): poleToSideFlowThread reads the whole side panorama of its eye (TRSP:388-398), the root
    composites both."""
    need = [0] * world
    for u, r in enumerate(owner):
        if r >= 0:
            need[r] |= 1 << (u & 1)
    need[root] = 3
    return need


class _DevArray:
    """Zero-copy view of a HIP device pointer owned by libs360 for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def strips_tensor(ctx, device):
    """torch uint8 tensor [2][P][camH][stripW][4] aliasing the context's strip buffers."""
    import torch
    g = ctx.geometry
    P = ctx.rig.get_side_camera_count()
    strip_w = ctx.params.eqr_width // P
    p0, _ = ctx.strip_ptr(0)
    return torch.as_tensor(_DevArray(p0, (2, P, g.cam_image_height, strip_w, 4)), device=device)
