"""Frame-sharded multi-GPU fusion: shard frames across ranks, fuse locally with zero communication,
merge the per-GPU voxel-block hashes once at the end with a single RCCL reduce.

This is the distributed form of CubeHandler::Merge (/root/reference/src/Integration/CubeHandler.h:
145-167): key union + per-voxel weighted mean.  The reference has no communication layer; the
exchange below is designed for xGMI:

  1. all_gather the per-rank block counts and (padded) int32x3 key arrays    -- 12 B / block
  2. every rank builds the same sorted union of keys (deterministic, no communication)
  3. each rank packs its blocks into union order in SUM form [w*sdf, w, w*c0, w*c1, w*c2]
     (HIP kernel k_pack_sum; zeros where the rank has no data)               -- 10 KiB / block
  4. ONE reduce(SUM, fp32) to the root over RCCL                              -- the only bulk transfer
  5. the root normalises back to mean form (HIP kernel k_unpack_sum).

With 2 ranks step 3-5 evaluate exactly TSDFVoxel::operator+ ((w1*s1 + w2*s2)/(w1+w2)); with more
ranks the summation order differs from a sequential Merge chain only in fp32 rounding (weights and
keys stay exact).  ICP does not shard (one pose chain): replicas only.

The collective logic is backend-neutral: `ops` supplies the three device steps.  HipVolumeOps is
the product implementation (C-ABI kernels); tests drive the same merge_volumes() over gloo with a
CPU stand-in to cover the world_size > 1 control flow without GPUs.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def shard_range(n_total, rank, world):
    """Contiguous frame chunk of `rank` (keeps each GPU's volume spatially compact)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipVolumeOps:
    """Device steps of the merge for an integration.CubeHandler (torch tensors on its GPU)."""

    def __init__(self, handler, device):
        self.h = handler
        self.device = device

    def keys(self):
        import torch
        n = self.h.BlockCount()
        t = torch.empty((max(n, 1), 3), dtype=torch.int32, device=self.device)
        got = C.c_size_t(0)
        L.check(L.load().op_volume_keys_device(self.h._h, C.c_void_p(t.data_ptr()), n, C.byref(got)))
        return t[:n]

    def pack_sum(self, union_keys):
        import torch
        n = union_keys.shape[0]
        out = torch.empty((max(n, 1), 5, 512), dtype=torch.float32, device=self.device)
        L.check(L.load().op_volume_pack_sum(self.h._h, C.c_void_p(union_keys.data_ptr()), n, C.c_void_p(out.data_ptr())))
        return out[:n]

    def unpack_sum(self, union_keys, summed):
        n = union_keys.shape[0]
        L.check(L.load().op_volume_unpack_sum(self.h._h, C.c_void_p(union_keys.data_ptr()), n, C.c_void_p(summed.data_ptr())))

    def unpack_begin(self, union_keys):
        L.check(L.load().op_volume_unpack_sum_begin(self.h._h, C.c_void_p(union_keys.data_ptr()), union_keys.shape[0]))

    def unpack_chunk(self, first, summed_chunk):
        L.check(L.load().op_volume_unpack_sum_chunk(self.h._h, first, summed_chunk.shape[0], C.c_void_p(summed_chunk.data_ptr())))


def _forced():
    import os
    return os.environ.get("ONEPIECE_MERGE_FORCE") == "1"


def merge_volumes(ops, root=0, group=None, chunk_blocks=32768):
    """Merge every rank's volume into `root`'s.  Returns the number of union blocks.

    The one reduce is issued in slices of `chunk_blocks` union blocks (320 MB each): while slice i is on the wire (RCCL's
    own stream), slice i+1 is packed on the volume's stream and slice i-1 is normalised on the root -- the device steps
    (~1 ms per GB) hide behind the transfer (~6 ms per GB over xGMI's point-to-point links).
    With world_size == 1 there is nothing to merge (ONEPIECE_MERGE_FORCE=1 runs the exchange anyway: a one-rank
    all_gather + reduce, which is how the RCCL path is exercised on a single-GPU box).
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _forced()):
        return int(ops.keys().shape[0])
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    keys = ops.keys().contiguous()
    dev = keys.device
    # 1. counts, then padded keys
    cnt = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, 3), dtype=torch.int32, device=dev)
    padded[:keys.shape[0]] = keys
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    allk = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
    # 2. identical sorted union on every rank: pack each key into one int64 (3 x 21 bits, the same
    #    packing the device hash table uses) so that the union is a 1-D radix sort instead of the much
    #    slower row-wise torch.unique(dim=0); the packed order is the lexicographic (x, y, z) order
    if allk.shape[0]:
        off = 1 << 20
        k64 = allk.to(torch.int64) + off
        packed_keys = torch.unique((k64[:, 0] << 42) | (k64[:, 1] << 21) | k64[:, 2])
        union = torch.stack([(packed_keys >> 42) - off, ((packed_keys >> 21) & 0x1FFFFF) - off, (packed_keys & 0x1FFFFF) - off],
                            dim=1).to(torch.int32).contiguous()
    else:
        union = allk
    n_union = int(union.shape[0])
    if n_union == 0:
        return 0
    # `union` was produced by torch ops queued on torch's stream, the pack / unpack kernels run on the volume's own
    # stream: make the keys final before those kernels read them.
    sync = (lambda: torch.cuda.current_stream(union.device).synchronize()) if union.is_cuda else (lambda: None)
    sync()
    is_root = rank == root
    chunk_blocks = max(1, int(chunk_blocks))
    spans = [(lo, min(lo + chunk_blocks, n_union)) for lo in range(0, n_union, chunk_blocks)]
    chunked = hasattr(ops, "unpack_begin") and len(spans) > 1
    if not chunked:
        # 3.-5. sum-form pack, one reduce to the root, normalise on the root
        packed = ops.pack_sum(union).contiguous()
        dist.reduce(packed, dst=root, op=dist.ReduceOp.SUM, group=group)
        sync()  # the collective is ordered on torch's stream; the volume's kernels run on the volume's own stream
        if is_root:
            ops.unpack_sum(union, packed)
        return n_union
    # 3.-5. pipelined over slices of the union
    return _merge_pipelined(ops, dist, union, spans, root, group, is_root, sync)


def _merge_pipelined(ops, dist, union, spans, root, group, is_root, sync):
    n_union = int(union.shape[0])
    # The root's volume is both a source (its blocks are summed like everybody's) and the destination.  Its slices are
    # therefore ALL packed before the first one is normalised into it: pack(i) happens at step i, unpack(i) at step
    # i + 2 at the earliest, and unpack_begin (which clears the volume) only after the last pack.
    bufs = [None] * len(spans)
    works = [None] * len(spans)
    for i, (lo, hi) in enumerate(spans):
        bufs[i] = ops.pack_sum(union[lo:hi]).contiguous()
        works[i] = dist.reduce(bufs[i], dst=root, op=dist.ReduceOp.SUM, group=group, async_op=True)
        if not is_root and i >= 2:                   # non-root ranks only need their send buffers until the slice is through
            works[i - 2].wait(); sync(); bufs[i - 2] = None
    if is_root:
        ops.unpack_begin(union)
    for i, (lo, hi) in enumerate(spans):
        if works[i] is not None:
            works[i].wait()
            sync()
        if is_root:
            ops.unpack_chunk(lo, bufs[i])
        bufs[i] = None
    return n_union
