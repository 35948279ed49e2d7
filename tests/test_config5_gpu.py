"""BASELINE.json configs[4] -- the 8-GPU frame-sharded ImageSequenceIntegration with the RCCL voxel-block hash merge -- as far as ONE
MI355X can run it.

1. The workload at its size: 8 shards x 1000 synthetic 640x480 frames, 5 mm voxels, each shard fused into its own volume on the one
   GPU, then merged through exactly the device steps the multi-GPU merge wraps around its one reduce (op_volume_keys_device ->
   sorted union -> k_pack_sum -> sum over "ranks" -> k_unpack_sum) and compared with the reference semantics, a sequential
   CubeHandler::Merge chain (/root/reference/src/Integration/CubeHandler.h:145-167; op_volume_merge).
2. op_volume_merge_rccl ITSELF with 2 ... 8 ranks, both algorithms (the owner-partitioned exchange and the dense reduce): the real RCCL refuses
   two ranks on one device, so the library's run-time binding (op_runtime_set_rccl_library) is pointed at tests/cpp/librccl_double.so -- ranks = host threads of tests/cpp/merge_world.bin, all on
   device 0, collectives through host memory.  What executes is the product's merge: padded key all-gather, the ~0 sentinel of the
   union, the three agreement points, the sliced reduce, root-only unpack -- with uneven shards, an empty rank, a root that is not
   rank 0, several reduce slices, and a rank that enters with a failed volume (every rank must return an error, nobody may hang).
"""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from onepiece_amd import integration as I, synthetic as S, distributed as D
from helpers import small_camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
LIB = os.path.join(ROOT, "onepiece_amd")
DOUBLE = os.path.join(CPP, "librccl_double.so")
WORLD = os.path.join(CPP, "merge_world.bin")


def build_double_and_driver():
    """The two test-only artefacts (tests/cpp/Makefile; also built by __graft_entry__.build(), so that they travel to the GPU box prebuilt)."""
    subprocess.check_call(["make", "-C", CPP, "-s"])


def _sorted_map(hv):
    k, v = hv.GetCubeMap()
    o = np.lexsort(k.T[::-1])
    return k[o], v[o]


def _assert_merged(got, want, tol=1e-6):
    """keys = the union, weights = exact sums, sdf / colour within fp32 summation order of the sequential Merge chain."""
    (gk, gv), (wk, wv) = got, want
    assert gk.shape == wk.shape and np.array_equal(gk, wk), "key sets differ"
    assert np.array_equal(gv[..., 1], wv[..., 1]), "weights differ"
    obs = wv[..., 1] > 0
    assert np.abs(gv[..., 0] - wv[..., 0])[obs].max(initial=0) <= tol and np.abs(gv[..., 2:] - wv[..., 2:])[obs].max(initial=0) <= tol
    assert np.array_equal(gv[~obs].view(np.uint32), wv[~obs].view(np.uint32)), "unobserved voxels must keep the sentinel"


def test_config5_8_shards_of_1000_frames_merge_on_one_gpu(hip):
    """configs[4] at its size, minus the wire: 8000 frames in 8 contiguous shards (distributed.shard_range), 8 volumes, the sum-form
    merge against the sequential Merge chain."""
    import torch
    dev = torch.device("cuda:0")
    ranks, per = 8, 1000
    vols, ops, local = [], [], []
    for r in range(ranks):
        lo, hi = D.shard_range(ranks * per, r, ranks)
        assert hi - lo == per
        depth, rgb, poses = S.room_sequence_torch(lo, per, dev)
        torch.cuda.synchronize()
        hv = I.CubeHandler()
        hv.SetVoxelResolution(0.005)
        hv.IntegrateSequence(depth, rgb, poses)
        hv.Synchronize()
        assert hv.Stats()["frames"] == per
        local.append(hv.BlockCount())
        vols.append(hv); ops.append(D.HipVolumeOps(hv, dev))
        del depth, rgb
    assert min(local) > 100_000
    # the exchange's device steps, the reduce replaced by a sum over the packed buffers in rank order
    allk = torch.cat([o.keys() for o in ops])
    off = 1 << 20
    k64 = allk.to(torch.int64) + off
    pk = torch.unique((k64[:, 0] << 42) | (k64[:, 1] << 21) | k64[:, 2])
    union = torch.stack([(pk >> 42) - off, ((pk >> 21) & 0x1FFFFF) - off, (pk & 0x1FFFFF) - off], 1).to(torch.int32).contiguous()
    torch.cuda.synchronize()
    n_union = union.shape[0]
    assert max(local) < n_union < sum(local)
    acc = None
    for o in ops:
        p = o.pack_sum(union)
        torch.cuda.synchronize()
        acc = p if acc is None else acc.add_(p)
        del p
    torch.cuda.synchronize()
    root = I.CubeHandler()
    root.SetVoxelResolution(0.005)
    D.HipVolumeOps(root, dev).unpack_sum(union, acc)
    del acc
    got = _sorted_map(root)
    del root
    # reference semantics: rank 0's volume += every other rank's, one after the other
    for v in vols[1:]:
        vols[0].Merge(v)
    want = _sorted_map(vols[0])
    assert len(want[0]) == n_union
    _assert_merged(got, want)
    # every in-band observation of every frame of every shard is in the merged weights (integers: exact in any order)
    assert int(want[1][..., 1].astype(np.float64).sum()) == sum(v.Stats()["voxels_updated"] for v in vols)


def _write_frames(path, frames, cam):
    fx, fy, cx, cy, w, h, _ = cam
    out = []
    with open(path, "wb") as f:
        np.array([len(frames), w, h], np.int32).tofile(f)
        for i in frames:
            pose = S.room_pose(i)
            d, c = S.room_render(pose, width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy)
            pose.astype(np.float32).tofile(f); d.astype(np.float32).tofile(f); c.astype(np.uint8).tofile(f)
            out.append((d, c, pose))
    return out


def _expected(frames, shards, cam, voxel, root):
    hcam = I.PinholeCamera()
    hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    vols = []
    for lo, hi in shards:
        hv = I.CubeHandler(hcam)
        hv.SetVoxelResolution(voxel)
        for d, c, p in frames[lo:hi]:
            hv.IntegrateImage(d, c, p)
        vols.append(hv)
    order = [root] + [r for r in range(len(shards)) if r != root]
    for r in order[1:]:
        vols[root].Merge(vols[r])
    return _sorted_map(vols[root]), hcam, [v.BlockCount() for v in vols[:root]] + [None] + [v.BlockCount() for v in vols[root + 1:]]


def _run_world(tmp_path, shards, voxel, root=0, fail_rank=None, slice_blocks=None, algorithm="owner", timeout=300, fault=None):
    build_double_and_driver()
    mp = str(tmp_path / "merged.map")
    cmd = [WORLD, str(tmp_path / "frames.bin"), mp, "--shards", ",".join("%d-%d" % s for s in shards), "--voxel", str(voxel), "--root", str(root),
           "--rccl-library", DOUBLE, "--algorithm", algorithm]
    if slice_blocks:
        cmd += ["--slice-blocks", str(slice_blocks)]
    if fail_rank is not None:
        cmd += ["--fail-rank", str(fail_rank)]
    if fault:
        cmd += ["--fault", str(fault)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)   # a rank left waiting in a collective = a timeout here
    assert run.stdout.strip(), run.stderr
    return run, json.loads(run.stdout.strip().splitlines()[-1]), mp


SHARD_CASES = [
    ([(0, 6), (6, 12)], 0, None),                                           # two equal ranks, one slice
    ([(0, 5), (5, 6), (6, 6), (6, 14)], 2, 700),                            # uneven, an EMPTY rank that is also the root, several slices
    ([(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 14), (14, 16)], 5, 500),   # eight ranks as in configs[4], root 5
    ([(0, 4), (4, 9), (9, 13)], 1, 900),                                    # three ranks: the owner hash must spread keys for a rank count that is no power of two
]


def _read_map(hcam, voxel, path):
    hv = I.CubeHandler(hcam)
    hv.SetVoxelResolution(voxel)
    hv.ReadFromFile(path)
    return hv


@pytest.mark.parametrize("shards,root,slice_blocks", SHARD_CASES)
def test_merge_rccl_dense_reduce_with_several_ranks_on_one_gpu(hip, tmp_path, shards, root, slice_blocks):
    """OP_MERGE_DENSE_REDUCE (the fallback): key all-gather, the union everywhere, one sliced ncclReduce of the whole union."""
    cam = small_camera(4)
    voxel = 0.02
    frames = _write_frames(str(tmp_path / "frames.bin"), [100 + 7 * i for i in range(shards[-1][1])], cam)
    want, hcam, _ = _expected(frames, shards, cam, voxel, root)
    run, r, mp = _run_world(tmp_path, shards, voxel, root=root, slice_blocks=slice_blocks, algorithm="dense")
    assert run.returncode == 0 and r["ok"] is True, run.stdout + run.stderr
    world = len(shards)
    assert [p["rccl_rank"] for p in r["per_rank"]] == list(range(world)) and all(p["rccl_ranks"] == world and p["status"] == 0 and p["algorithm"] == 1 for p in r["per_rank"])
    n_union = len(want[0])
    assert r["root_blocks"] == n_union and r["per_rank"][root]["union_blocks"] == n_union
    assert all(p["bytes"] == n_union * 10240 for p in r["per_rank"])
    if slice_blocks:
        assert all(p["slices"] == -(-n_union // slice_blocks) >= 2 for p in r["per_rank"])
    for p, (lo, hi) in zip(r["per_rank"], shards):
        assert (p["local_blocks"] == 0) == (lo == hi)
    _assert_merged(_sorted_map(_read_map(hcam, voxel, mp)), want)


@pytest.mark.parametrize("shards,root,slice_blocks", SHARD_CASES)
def test_merge_rccl_owner_exchange_with_several_ranks_on_one_gpu(hip, tmp_path, shards, root, slice_blocks):
    """OP_MERGE_OWNER_EXCHANGE (the default): every rank sends only the blocks it HOLDS, to their owners; the owners' partitions are
    gathered on the root.  (a) the merged volume = the sequential Merge chain; (b) the bytes on the wire are what the blocks held say:
    per rank, sent = (blocks held of other ranks' partitions) x 10 248 B + its summed partition (unless it is the root); summed over the
    ranks, the exchange moves held x 10 248 x (world - 1) / world on average -- never the union's zeros."""
    del slice_blocks
    cam = small_camera(4)
    voxel = 0.02
    frames = _write_frames(str(tmp_path / "frames.bin"), [100 + 7 * i for i in range(shards[-1][1])], cam)
    want, hcam, _ = _expected(frames, shards, cam, voxel, root)
    run, r, mp = _run_world(tmp_path, shards, voxel, root=root)
    assert run.returncode == 0 and r["ok"] is True, run.stdout + run.stderr
    world = len(shards)
    P = r["per_rank"]
    assert [p["rccl_rank"] for p in P] == list(range(world)) and all(p["rccl_ranks"] == world and p["status"] == 0 and p["algorithm"] == 0 for p in P)
    n_union = len(want[0])
    assert r["root_blocks"] == n_union and all(p["union_blocks"] == n_union for p in P)
    assert sum(p["owned_blocks"] for p in P) == n_union                      # the partitions are disjoint and cover the union
    assert all(p["held_blocks"] == p["local_blocks"] for p in P)
    B = 10240 + 8
    held = sum(p["held_blocks"] for p in P)
    exchange_sent = sum(p["wire_bytes_sent"] for p in P) - sum(p["owned_blocks"] * B for k, p in enumerate(P) if k != root)   # minus the gather
    assert sum(p["wire_bytes_sent"] for p in P) == sum(p["wire_bytes_received"] for p in P)
    assert 0 <= exchange_sent <= held * B and exchange_sent % B == 0
    # the owner hash spreads the keys evenly: the exchange moves (world - 1) / world of what is held, within a few per cent on these ~1e3-block volumes
    assert abs(exchange_sent / (held * B) - (world - 1) / world) < 0.08
    # a rank never sends more than what it holds plus its summed partition (the dense reduce sends the whole union from every rank but the root, through one link;
    # with shards that overlap almost completely -- these -- the totals are close, what differs is that the exchange uses every link at once)
    assert all(p["wire_bytes_sent"] <= (p["held_blocks"] + p["owned_blocks"]) * B for p in P)
    _assert_merged(_sorted_map(_read_map(hcam, voxel, mp)), want)


def test_merge_rccl_owner_exchange_without_a_gather_leaves_every_rank_its_partition(hip, tmp_path):
    """root = -1: a distributed map -- every rank's volume is replaced by its owned, merged partition; their disjoint union is the Merge chain's result."""
    shards = [(0, 4), (4, 5), (5, 5), (5, 12)]
    cam = small_camera(4)
    voxel = 0.02
    frames = _write_frames(str(tmp_path / "frames.bin"), [100 + 7 * i for i in range(shards[-1][1])], cam)
    want, hcam, _ = _expected(frames, shards, cam, voxel, 0)
    run, r, mp = _run_world(tmp_path, shards, voxel, root=-1)
    assert run.returncode == 0 and r["ok"] is True, run.stdout + run.stderr
    parts = [_sorted_map(_read_map(hcam, voxel, "%s.rank%d" % (mp, k))) for k in range(len(shards))]
    assert [len(p[0]) for p in parts] == [p["owned_blocks"] for p in r["per_rank"]]
    keys = np.concatenate([p[0] for p in parts]); vox = np.concatenate([p[1] for p in parts])
    assert len(np.unique(keys, axis=0)) == len(keys) == len(want[0])       # disjoint, complete
    o = np.lexsort(keys.T[::-1])
    _assert_merged((keys[o], vox[o]), want)
    assert sum(p["wire_bytes_sent"] for p in r["per_rank"]) == sum(p["wire_bytes_received"] for p in r["per_rank"])


@pytest.mark.parametrize("algorithm", ["owner", "dense"])
def test_merge_rccl_rank_with_a_failed_volume_fails_everywhere_without_hanging(hip, tmp_path, algorithm):
    cam = small_camera(4)
    frames = _write_frames(str(tmp_path / "frames.bin"), [100 + 7 * i for i in range(9)], cam)
    del frames
    run, r, _ = _run_world(tmp_path, [(0, 3), (3, 6), (6, 9)], 0.02, root=0, fail_rank=1, algorithm=algorithm, timeout=120)
    assert run.returncode != 0 and r["ok"] is False
    assert all(p["status"] != 0 for p in r["per_rank"]), r          # nobody "succeeds" with a partial merge
    assert "bounding box" in r["per_rank"][1]["error"]               # the failing rank reports ITS failure ...
    assert all("rank 1 entered the merge with a failed volume" in r["per_rank"][k]["error"] for k in (0, 2))   # ... the others whose it was


@pytest.mark.parametrize("stage,rank,root,whose", [(1, 2, 0, "rank 2 could not sum its partition"), (2, 1, 1, "the root could not allocate the gathered map")])
def test_merge_rccl_allocation_failure_after_the_exchange_fails_everywhere_without_hanging(hip, tmp_path, stage, rank, root, whose):
    """The two allocations whose size is only known AFTER the exchange -- an owner's partition sums (n_own x 10 KiB) and the root's gather
    buffers (n_union x 10 KiB, the largest of the merge) -- are agreed on before the next transfer: a rank that cannot get them (injected:
    OP_RUNTIME_OPT_MERGE_FAULT) makes EVERY rank return an error instead of leaving its peers inside ncclSend / the all-gather."""
    cam = small_camera(4)
    frames = _write_frames(str(tmp_path / "frames.bin"), [100 + 7 * i for i in range(9)], cam)
    del frames
    run, r, _ = _run_world(tmp_path, [(0, 3), (3, 6), (6, 9)], 0.02, root=root, timeout=120, fault=stage * 1024 + rank + 1)
    assert run.returncode != 0 and r["ok"] is False
    assert all(p["status"] != 0 for p in r["per_rank"]), r
    assert "(injected)" in r["per_rank"][rank]["error"]
    assert all(whose in r["per_rank"][k]["error"] for k in range(3) if k != rank), r
