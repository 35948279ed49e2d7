"""Frame-sharded multi-GPU fusion: shard frames across ranks, fuse locally with zero communication,
merge the per-GPU voxel-block hashes once at the end over RCCL.

This is the distributed form of CubeHandler::Merge (/root/reference/src/Integration/CubeHandler.h:
145-167): key union + per-voxel weighted mean -- and, like the reference's Merge, it only touches blocks
somebody HOLDS.  The reference has no communication layer; the exchange is designed for xGMI,
which is point to point (7 links per GPU).  Two algorithms, the same two as op_volume_merge_rccl (csrc/merge_rccl.hip):

  "owner" (default) -- every block key has an owner rank (a hash of the key mod the number of ranks):
    1. a rank sorts its keys by owner and packs its blocks in that order in SUM form [w*sdf, w, w*c0, w*c1, w*c2]
       (HIP kernel k_pack_sum)                                                                    -- 10 KiB / block HELD
    2. all_gather of the world x world matrix of counts, then one batch of point-to-point sends / receives: each rank
       sends every other rank the keys and blocks it holds of that rank's partition -- (world - 1) / world of what it
       holds crosses the wire, over its links to all peers at once
    3. the owner builds the sorted union of its partition and adds the received blocks into it source by source in
       rank order (deterministic)
    4. root given: the owners send their summed partitions to the root, which normalises the whole map into its volume
       (HIP kernel k_unpack_sum); root None: every rank's volume becomes its owned, merged partition.
  "dense" (rounds 1-4, the fallback) -- all_gather the padded key arrays, the same sorted union on every rank, every rank
    packs the WHOLE union (zeros where it holds nothing), ONE sliced reduce(SUM, fp32) to the root, normalise.

With 2 ranks both evaluate exactly TSDFVoxel::operator+ ((w1*s1 + w2*s2)/(w1+w2)); with more
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


last_stats = {}   # what the last merge_volumes call of this process moved (bench.py's multi_gpu object)

_BLOCK_BYTES = 10240 + 8   # a sum-form block + its packed key


def _pack64(keys):
    """int32 x 3 block ids -> one int64 each (3 x 21 bits: the packing of the device hash table; the packed order is the lexicographic one)."""
    import torch
    k = keys.to(torch.int64) + (1 << 20)
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def _unpack64(packed):
    import torch
    off = 1 << 20
    return torch.stack([(packed >> 42) - off, ((packed >> 21) & 0x1FFFFF) - off, (packed & 0x1FFFFF) - off], dim=1).to(torch.int32).contiguous()


def owner_of(packed, world):
    """Owner rank of a block: a finalising mix of its packed id, mod the number of ranks (= owner_of of csrc/merge_rccl.hip; int64 arithmetic
    wraps like uint64's, the shifts are made logical by masking)."""
    p = packed.clone()
    p = p ^ ((p >> 33) & 0x7FFFFFFF)
    p = p * (-49064778989728563)            # 0xff51afd7ed558ccd
    p = p ^ ((p >> 33) & 0x7FFFFFFF)
    p = p * (-4265267296055464877)          # 0xc4ceb9fe1a85ec53
    p = p ^ ((p >> 33) & 0x7FFFFFFF)
    # unsigned p mod world from the signed representation: p_u = 2 * (p >>> 1) + (p & 1)
    half = (p >> 1) & 0x7FFFFFFFFFFFFFFF
    return ((half % world) * 2 + (p & 1)) % world


def _merge_owner_exchange(ops, dist, root, group, world, rank):
    import torch
    keys = ops.keys().contiguous()
    dev = keys.device
    n_local = int(keys.shape[0])
    packed = _pack64(keys) if n_local else torch.zeros((0,), dtype=torch.int64, device=dev)
    own = owner_of(packed, world) if n_local else packed
    order = torch.argsort(own, stable=True)
    packed = packed[order].contiguous()
    counts = torch.bincount(own, minlength=world).to(torch.int64) if n_local else torch.zeros((world,), dtype=torch.int64, device=dev)
    # 1. my blocks in owner order, sum form
    sync = (lambda: torch.cuda.current_stream(dev).synchronize()) if keys.is_cuda else (lambda: None)
    skeys = _unpack64(packed) if n_local else keys
    sync()
    send = ops.pack_sum(skeys).contiguous() if n_local else torch.zeros((0, 5, 512), dtype=torch.float32, device=dev)
    # 2. the matrix of counts
    rows = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(rows, counts, group=group)
    matrix = torch.stack(rows).cpu().numpy()                 # matrix[s][d]: blocks s holds of d's partition
    bounds = np.concatenate([[0], np.cumsum(matrix[rank])]).astype(np.int64)
    roff = np.concatenate([[0], np.cumsum(matrix[:, rank])]).astype(np.int64)
    n_recv = int(roff[-1])
    rkeys = torch.empty((n_recv,), dtype=torch.int64, device=dev)
    rpay = torch.empty((n_recv, 5, 512), dtype=torch.float32, device=dev)
    # 3. the exchange: one batch of point-to-point operations over all pairs
    p2p, sent, received = [], 0, 0
    for p in range(world):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        rlo, rhi = int(roff[p]), int(roff[p + 1])
        if p == rank:
            if hi > lo:
                rkeys[rlo:rhi] = packed[lo:hi]; rpay[rlo:rhi] = send[lo:hi]
            continue
        if hi > lo:
            p2p += [dist.P2POp(dist.isend, packed[lo:hi], p, group), dist.P2POp(dist.isend, send[lo:hi], p, group)]
            sent += (hi - lo) * _BLOCK_BYTES
        if rhi > rlo:
            p2p += [dist.P2POp(dist.irecv, rkeys[rlo:rhi], p, group), dist.P2POp(dist.irecv, rpay[rlo:rhi], p, group)]
            received += (rhi - rlo) * _BLOCK_BYTES
    sync()   # the pack kernel ran on the volume's stream; torch's collectives are ordered on torch's
    if p2p:
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
    # 4. the partition's sorted union, and the sum over the sources in rank order
    uni, inv = torch.unique(rkeys, sorted=True, return_inverse=True) if n_recv else (rkeys, rkeys)
    n_own = int(uni.shape[0])
    acc = torch.zeros((n_own, 5, 512), dtype=torch.float32, device=dev)
    for s_ in range(world):
        lo, hi = int(roff[s_]), int(roff[s_ + 1])
        if hi > lo:
            acc.index_add_(0, inv[lo:hi], rpay[lo:hi])       # a source holds a key once: no two rows of one call meet
    del rpay, send
    # 5. sizes of the partitions
    mine = torch.tensor([n_own], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    owned = [int(x.item()) for x in sizes]
    n_union = sum(owned)
    if root is None:
        sync()
        ops.unpack_sum(_unpack64(uni) if n_own else keys[:0], acc)      # my volume = my owned, merged partition
    else:
        goff = np.concatenate([[0], np.cumsum(owned)]).astype(np.int64)
        p2p = []
        if rank == root:
            gk = torch.empty((n_union,), dtype=torch.int64, device=dev)
            gp = torch.empty((n_union, 5, 512), dtype=torch.float32, device=dev)
            for r_ in range(world):
                lo, hi = int(goff[r_]), int(goff[r_ + 1])
                if hi == lo:
                    continue
                if r_ == rank:
                    gk[lo:hi] = uni; gp[lo:hi] = acc
                else:
                    p2p += [dist.P2POp(dist.irecv, gk[lo:hi], r_, group), dist.P2POp(dist.irecv, gp[lo:hi], r_, group)]
                    received += (hi - lo) * _BLOCK_BYTES
        elif n_own:
            p2p += [dist.P2POp(dist.isend, uni.contiguous(), root, group), dist.P2POp(dist.isend, acc, root, group)]
            sent += n_own * _BLOCK_BYTES
        if p2p:
            for w in dist.batch_isend_irecv(p2p):
                w.wait()
        if rank == root:
            allk = _unpack64(gk) if n_union else keys[:0]
            sync()
            ops.unpack_sum(allk, gp)
    last_stats.clear()
    last_stats.update({"algorithm": "owner", "ranks": world, "rank": rank, "held_blocks": n_local, "owned_blocks": n_own, "union_blocks": n_union,
                       "wire_bytes_sent": int(sent), "wire_bytes_received": int(received)})
    return n_union


def merge_volumes(ops, root=0, group=None, chunk_blocks=32768, algorithm="owner"):
    """Merge every rank's volume into `root`'s (algorithm "owner" with root=None: into a distributed map, every rank keeping its owned
    partition).  Returns the number of union blocks; `last_stats` says what crossed the wire.

    "owner": the owner-partitioned exchange of the module docstring.  "dense": the one reduce of the whole union, below.

    The one reduce is issued in slices of `chunk_blocks` union blocks (320 MB each): while slice i is on the wire (RCCL's
    own stream), slice i+1 is packed on the volume's stream and slice i-1 is normalised on the root -- the device steps
    (~1 ms per GB) hide behind the transfer (~6 ms per GB over xGMI's point-to-point links).
    With world_size == 1 there is nothing to merge (ONEPIECE_MERGE_FORCE=1 runs the exchange anyway: a one-rank
    all_gather + reduce, which is how the RCCL path is exercised on a single-GPU box).
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _forced()):
        n = int(ops.keys().shape[0])   # one rank: it holds and owns the whole map, nothing crosses a wire
        last_stats.clear()
        last_stats.update({"algorithm": "none", "ranks": 1, "rank": 0, "held_blocks": n, "owned_blocks": n, "union_blocks": n, "wire_bytes_sent": 0, "wire_bytes_received": 0})
        return n
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if algorithm == "owner":
        return _merge_owner_exchange(ops, dist, root, group, world, rank)
    if root is None:
        raise ValueError("the dense reduce needs a root")
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
    last_stats.clear()
    last_stats.update({"algorithm": "dense", "ranks": world, "rank": rank, "held_blocks": int(keys.shape[0]), "owned_blocks": n_union if rank == root else 0,
                       "union_blocks": n_union, "wire_bytes_sent": (n_union * 10240 if rank != root and world > 1 else 0) + (world - 1) * mx * 12,
                       # the reduced payload ends on the root only (what a ring's inner ranks forward is RCCL's business); everybody receives the gathered keys
                       "wire_bytes_received": (n_union * 10240 if rank == root and world > 1 else 0) + (world - 1) * mx * 12})
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


# ---- the product path: op_volume_merge_rccl on a communicator of the caller's own ------------------------------------------------------
class RcclCommunicator:
    """An ncclComm_t for op_volume_merge_rccl (csrc/merge_rccl.hip), made the way any RCCL host program makes one: rank 0 draws a unique id,
    the caller carries its 128 bytes to the other ranks (`exchange`: bytes-or-None -> bytes; over torch.distributed's store, MPI, a file ...),
    every rank calls ncclCommInitRank on the device that is current.  The library named here is also the one the merge binds
    (op_runtime_set_rccl_library), so that communicator and collectives come from ONE RCCL -- in a torch process, torch's own copy.
    """

    def __init__(self, rank, world, exchange, library=None):
        if library is None:
            library = default_rccl_library()
        self.library = library
        self._rccl = C.CDLL(library, mode=C.RTLD_GLOBAL)

        class _UniqueId(C.Structure):
            _fields_ = [("internal", C.c_ubyte * 128)]   # ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)

        uid = _UniqueId()
        self._rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        self._rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        self._rccl.ncclGetErrorString.restype = C.c_char_p
        if rank == 0:
            self._check(self._rccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = exchange(bytes(uid.internal) if rank == 0 else None)
        if len(raw) != 128:
            raise ValueError("the exchanged ncclUniqueId has %d bytes, not 128" % len(raw))
        C.memmove(C.byref(uid), raw, 128)
        rc = L.load().op_runtime_set_rccl_library(library.encode())
        if rc != L.OP_OK and not getattr(RcclCommunicator, "_bound", None) == library:   # already bound by an earlier merge of this process: fine if it is the same library
            L.check(rc)
        RcclCommunicator._bound = library
        self.comm = C.c_void_p()
        self._check(self._rccl.ncclCommInitRank(C.byref(self.comm), int(world), uid, int(rank)), "ncclCommInitRank")
        self.rank, self.world = int(rank), int(world)

    def _check(self, r, what):
        if r != 0:
            raise RuntimeError("%s failed: %s" % (what, self._rccl.ncclGetErrorString(r).decode()))

    def count(self):
        """ncclCommCount: the ranks RCCL itself says this communicator has."""
        n = C.c_int(0)
        self._check(self._rccl.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        return n.value

    def destroy(self):
        if self.comm:
            self._rccl.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def default_rccl_library():
    """torch's own librccl.so when torch is importable (a second copy of RCCL in one process is asking for trouble), else the system's."""
    import os
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            return cand
    except ImportError:
        pass
    return "librccl.so.1"


def torch_exchange(dist, group=None):
    """`exchange` for RcclCommunicator over an initialised torch.distributed process group (any backend)."""
    def ex(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    return ex


_ALGORITHMS = {"owner": L.OP_MERGE_OWNER_EXCHANGE, "dense": L.OP_MERGE_DENSE_REDUCE}


def merge_volumes_rccl(handler, comm, root=0, algorithm="owner", force_single_rank=False):
    """CubeHandler::MergeAcrossRanks: ONE library call, op_volume_merge_rccl_stats, on `comm` (an RcclCommunicator or a raw ncclComm_t
    address) -- the C++ class surface's path (host/one_piece/src/CubeHandler.cpp) and what bench.py --gpus N times.  root=None: no
    gather, every rank keeps its owned partition.  Returns the union's block count; `last_stats` says what crossed the wire."""
    lib = L.load()
    L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_MERGE_ALGORITHM, _ALGORITHMS[algorithm]))
    L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK, 1 if force_single_rank else 0))
    st = L.MergeStats()
    n = C.c_size_t(0)
    raw = comm.comm if isinstance(comm, RcclCommunicator) else C.c_void_p(comm)
    rc = lib.op_volume_merge_rccl_stats(handler._h, raw, -1 if root is None else int(root), C.byref(n), C.byref(st))
    last_stats.clear()
    last_stats.update({"algorithm": "owner" if st.algorithm == L.OP_MERGE_OWNER_EXCHANGE else "dense", "impl": "op_volume_merge_rccl", "ranks": st.ranks, "rank": st.rank,
                       "held_blocks": int(st.held_blocks), "owned_blocks": int(st.owned_blocks), "union_blocks": int(st.union_blocks),
                       "wire_bytes_sent": int(st.wire_bytes_sent), "wire_bytes_received": int(st.wire_bytes_received),
                       "prepare_ms": st.prepare_ms, "transfer_ms": st.transfer_ms, "total_ms": st.total_ms, "slices": int(st.slices)})
    L.check(rc)
    return int(n.value)
