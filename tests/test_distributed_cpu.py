"""world_size-2 gloo coverage of the frame-sharded merge (onepiece_amd.distributed.merge_volumes).

The collective control flow (count/key all_gather, deterministic union, sum-form pack, ONE reduce,
normalise on the root) is backend-neutral; here its three device steps are stood in for by numpy
over oracle volumes so the N > 1 path runs on CPU.  The result must equal CubeHandler::Merge of the
two shards (oracle) -- keys and weights exactly, sdf/colour within 1e-4 (sum-form rounding).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleVolumeOps:
    """CPU stand-in for distributed.HipVolumeOps (same three steps, numpy arithmetic)."""

    def __init__(self, vol):
        self.vol = vol

    def keys(self):
        import torch
        k, _ = self.vol.export()
        return torch.from_numpy(k.copy())

    def pack_sum(self, union_keys):
        import torch
        k, v = self.vol.export()
        index = {tuple(r): i for i, r in enumerate(k.tolist())}
        out = np.zeros((union_keys.shape[0], 5, 512), np.float32)
        for j, r in enumerate(union_keys.tolist()):
            i = index.get(tuple(r))
            if i is None:
                continue
            w = v[i, :, 1]
            m = w > 0
            out[j, 1, m] = w[m]
            out[j, 0, m] = w[m] * v[i, m, 0]
            for c in range(3):
                out[j, 2 + c, m] = w[m] * v[i, m, 2 + c]
        return torch.from_numpy(out)

    def unpack_sum(self, union_keys, summed):
        s = summed.numpy()
        n = s.shape[0]
        vox = np.empty((n, 512, 5), np.float32)
        w = s[:, 1, :]
        m = w > 0
        safe = np.where(m, w, 1).astype(np.float32)
        vox[:, :, 1] = np.where(m, w, 0)
        vox[:, :, 0] = np.where(m, s[:, 0, :] / safe, 999)
        for c in range(3):
            vox[:, :, 2 + c] = np.where(m, s[:, 2 + c, :] / safe, -1)
        self.vol.clear()
        self.vol.load(union_keys.numpy().astype(np.int32), vox)

    # the sliced form (op_volume_unpack_sum_begin / _chunk)
    def unpack_begin(self, union_keys):
        self._union = union_keys.numpy().astype(np.int32)
        self.vol.clear()

    def unpack_chunk(self, first, summed_chunk):
        keep, self.vol = self.vol, _Appender(self.vol)
        try:
            self.unpack_sum_into(self._union[first:first + summed_chunk.shape[0]], summed_chunk)
        finally:
            self.vol = keep

    def unpack_sum_into(self, keys, summed):
        import torch
        tmp = OracleVolumeOps(self.vol)
        tmp.unpack_sum(torch.from_numpy(keys), summed)


class _Appender:
    """Lets unpack_sum's clear() + load() append a slice instead of replacing the volume."""

    def __init__(self, vol):
        self._vol = vol

    def clear(self):
        pass

    def load(self, keys, vox):
        self._vol.load(keys, vox)


def _worker(rank, world, port, outdir, chunk_blocks=32768, algorithm="dense", root=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle as O
    from onepiece_amd import distributed as D, synthetic as S
    from helpers import small_camera
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam = small_camera(4)
    frames = [0, 25, 50, 75, 100, 125]
    lo, hi = D.shard_range(len(frames), rank, world)
    vol = O.Volume(O.make_camera(*cam), voxel_res=0.02)
    for i in frames[lo:hi]:
        pose = S.room_pose(i)
        d, c = S.room_render(pose, width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
        vol.integrate(d, c, pose)
    k, v = vol.export()
    np.savez(os.path.join(outdir, "shard%d.npz" % rank), k=k, v=v)
    n_union = D.merge_volumes(OracleVolumeOps(vol), root=root, chunk_blocks=chunk_blocks, algorithm=algorithm)
    import json
    json.dump(dict(D.last_stats), open(os.path.join(outdir, "stats%d.json" % rank), "w"))
    if root is None:                                          # a distributed map: every rank keeps its owned partition
        k, v = vol.export()
        np.savez(os.path.join(outdir, "part%d.npz" % rank), k=k, v=v, n_union=n_union)
    elif rank == root:
        k, v = vol.export()
        np.savez(os.path.join(outdir, "merged.npz"), k=k, v=v, n_union=n_union)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from onepiece_amd import distributed as D
    for n, w in [(8000, 8), (10, 3), (2, 4), (0, 2)]:
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_gloo_merge_equals_reference_merge(oracle, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from helpers import small_camera
    cam = oracle.make_camera(*small_camera(4))
    a, b = oracle.Volume(cam, voxel_res=0.02), oracle.Volume(cam, voxel_res=0.02)
    s0, s1 = np.load(tmp_path / "shard0.npz"), np.load(tmp_path / "shard1.npz")
    a.load(s0["k"], s0["v"]); b.load(s1["k"], s1["v"])
    assert a.merge(b) == 0  # CubeHandler::Merge (CubeHandler.h:145-167)
    rk, rv = a.export()
    m = np.load(tmp_path / "merged.npz")
    assert int(m["n_union"]) == len(rk)
    assert np.array_equal(m["k"], rk), "merged key set differs from CubeHandler::Merge"
    assert np.array_equal(m["v"][:, :, 1], rv[:, :, 1]), "weights must be exact"
    seen = rv[:, :, 1] > 0
    assert np.abs(m["v"][:, :, 0] - rv[:, :, 0])[seen].max() <= 1e-4 * 0.1
    assert np.abs(m["v"][:, :, 2:] - rv[:, :, 2:])[seen].max() <= 1e-4
    assert np.array_equal(m["v"][~seen], rv[~seen])  # untouched voxels keep the {999, 0, -1} sentinel
    # both shards really contributed and really overlapped
    k0 = {tuple(r) for r in s0["k"].tolist()}; k1 = {tuple(r) for r in s1["k"].tolist()}
    assert k0 - k1 and k1 - k0 and k0 & k1


def test_four_rank_gloo_sliced_merge_equals_sequential_merge_chain(oracle, tmp_path):
    """World size 4, the reduce issued in slices of 64 union blocks (pack / reduce / normalise pipelined, the root's volume
    being source AND destination): rank 0's merged volume equals CubeHandler::Merge applied shard after shard -- keys and
    weights exactly, sdf / colour to fp32 summation order."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(4, port, str(tmp_path), 64), nprocs=4, join=True)
    from helpers import small_camera
    cam = oracle.make_camera(*small_camera(4))
    vols = []
    for r in range(4):
        sh = np.load(tmp_path / ("shard%d.npz" % r))
        v = oracle.Volume(cam, voxel_res=0.02)
        if len(sh["k"]):
            v.load(sh["k"], sh["v"])
        vols.append(v)
    for v in vols[1:]:
        assert vols[0].merge(v) == 0
    rk, rv = vols[0].export()
    m = np.load(tmp_path / "merged.npz")
    assert int(m["n_union"]) == len(rk) > 3 * 64        # several slices were in flight
    assert np.array_equal(m["k"], rk) and np.array_equal(m["v"][:, :, 1], rv[:, :, 1])
    seen = rv[:, :, 1] > 0
    assert np.abs(m["v"][:, :, 0] - rv[:, :, 0])[seen].max() <= 1e-4 * 0.1 and np.abs(m["v"][:, :, 2:] - rv[:, :, 2:])[seen].max() <= 1e-4
    assert np.array_equal(m["v"][~seen], rv[~seen])


def _chain(oracle, tmp_path, world):
    from helpers import small_camera
    cam = oracle.make_camera(*small_camera(4))
    vols, shards = [], []
    for r in range(world):
        sh = np.load(tmp_path / ("shard%d.npz" % r))
        v = oracle.Volume(cam, voxel_res=0.02)
        if len(sh["k"]):
            v.load(sh["k"], sh["v"])
        vols.append(v); shards.append(sh)
    for v in vols[1:]:
        assert vols[0].merge(v) == 0
    return vols[0].export(), shards


def _same_as_chain(k, v, rk, rv):
    assert np.array_equal(k, rk) and np.array_equal(v[:, :, 1], rv[:, :, 1])
    seen = rv[:, :, 1] > 0
    assert np.abs(v[:, :, 0] - rv[:, :, 0])[seen].max() <= 1e-4 * 0.1 and np.abs(v[:, :, 2:] - rv[:, :, 2:])[seen].max() <= 1e-4
    assert np.array_equal(v[~seen], rv[~seen])


@pytest.mark.parametrize("world,root", [(2, 0), (4, 2)])
def test_gloo_owner_exchange_equals_sequential_merge_chain_and_moves_only_held_blocks(oracle, tmp_path, world, root):
    """The default algorithm, the owner-partitioned exchange, over gloo: every rank sends the blocks it HOLDS to their owners (point to
    point, all pairs in one batch), the owners' summed partitions are gathered on the root.  The root's volume = the sequential Merge
    chain (keys and weights exactly); the bytes on the wire are what the held blocks say."""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 32768, "owner", root), nprocs=world, join=True)
    (rk, rv), _sh = _chain(oracle, tmp_path, world)
    m = np.load(tmp_path / "merged.npz")
    assert int(m["n_union"]) == len(rk)
    _same_as_chain(m["k"], m["v"], rk, rv)
    st = [json.load(open(tmp_path / ("stats%d.json" % r))) for r in range(world)]
    assert all(s["algorithm"] == "owner" and s["union_blocks"] == len(rk) for s in st) and sum(s["owned_blocks"] for s in st) == len(rk)
    B = 10240 + 8
    held = sum(s["held_blocks"] for s in st)
    total_sent = sum(s["wire_bytes_sent"] for s in st)
    assert total_sent == sum(s["wire_bytes_received"] for s in st)
    exchange = total_sent - sum(s["owned_blocks"] * B for r, s in enumerate(st) if r != root)     # minus the gather to the root
    assert 0 < exchange <= held * B and abs(exchange / (held * B) - (world - 1) / world) < 0.1
    assert st[root]["wire_bytes_sent"] == exchange - sum(s["wire_bytes_sent"] - s["owned_blocks"] * B for r, s in enumerate(st) if r != root)


def test_gloo_owner_exchange_without_a_root_leaves_a_distributed_map(oracle, tmp_path):
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 32768, "owner", None), nprocs=world, join=True)
    (rk, rv), _sh = _chain(oracle, tmp_path, world)
    parts = [np.load(tmp_path / ("part%d.npz" % r)) for r in range(world)]
    keys = np.concatenate([p["k"] for p in parts]); vox = np.concatenate([p["v"] for p in parts])
    assert len(np.unique(keys, axis=0)) == len(keys) == len(rk) and all(int(p["n_union"]) == len(rk) for p in parts)
    o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    _same_as_chain(keys[o], vox[o], rk, rv)
    from onepiece_amd import distributed as D
    import torch
    for r, p in enumerate(parts):                            # every block sits on its owner
        if len(p["k"]):
            assert (D.owner_of(D._pack64(torch.from_numpy(p["k"])), world) == r).all()


def test_frame_prefetcher_yields_in_order(tmp_path):
    """Harness code: threaded PNG prefetch returns exactly what sequential imread returns."""
    import numpy as np
    from onepiece_amd import sequence as Q
    rng = np.random.default_rng(3)
    n, h, w = 12, 24, 32
    depths = [rng.uniform(0.5, 4.0, (h, w)).astype(np.float32) for _ in range(n)]
    rgbs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]
    poses = [np.eye(4, dtype=np.float32) for _ in range(n)]
    Q.WriteImageSequence(str(tmp_path), depths, rgbs, poses)
    rgb_files, depth_files = Q.ReadImageSequence(str(tmp_path))
    got = list(Q.FramePrefetcher(rgb_files, depth_files, indices=range(0, n, 2), workers=4, ahead=3))
    assert [g[0] for g in got] == list(range(0, n, 2))
    for i, rgb, depth in got:
        assert np.array_equal(rgb, Q.imread(rgb_files[i])) and np.array_equal(rgb, rgbs[i])
        assert np.array_equal(depth, Q.imread(depth_files[i], unchanged=True))
        assert depth.dtype == np.uint16 and np.array_equal(depth, np.round(depths[i].astype(np.float64) * 1000).astype(np.uint16))
