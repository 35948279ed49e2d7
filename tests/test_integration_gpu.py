"""Parity of the HIP TSDF-integration path (through the C-ABI) against the CPU oracle.

Bar (BASELINE.json north_star): bit-exact voxel-block keys + 64-bit hashes; TSDF within 1e-4
relative.  In practice the kernels keep the reference's operation order, so voxel values are
compared for exact equality first and the tolerance is only the documented fallback.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from onepiece_amd import integration as I, synthetic as S
from helpers import small_camera

TOL = 1e-4  # north_star: "TSDF ... within 1e-4 relative" (relative to the truncation distance / to 1 for colour)


def _mk(oracle, res=0.005, cam=None, **kw):
    ocam = oracle.make_camera() if cam is None else oracle.make_camera(*cam)
    hcam = I.PinholeCamera()
    if cam is not None:
        hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    ov = oracle.Volume(ocam, voxel_res=res)
    hv = I.CubeHandler(hcam, **kw)
    hv.SetVoxelResolution(res)
    return ov, hv


def _compare(oracle, ov, hv, exact=True):
    ok, ovx = ov.export()
    hk, hvx = hv.GetCubeMap()
    assert hk.shape == ok.shape and np.array_equal(hk, ok), "block key sets differ"
    oh = [oracle.hash_key(*k) for k in ok[:: max(1, len(ok) // 512)]]
    hh = [I.hash_key(*k) for k in hk[:: max(1, len(hk) // 512)]]
    assert oh == hh
    assert np.array_equal(hvx[:, :, 1], ovx[:, :, 1]), "weights must be exactly equal"
    if exact:
        assert np.array_equal(hvx.view(np.uint32), ovx.view(np.uint32)), "voxel payload not bit-identical"
    else:
        trunc = 0.1
        assert np.abs(hvx[:, :, 0] - ovx[:, :, 0])[ovx[:, :, 1] > 0].max(initial=0) <= TOL * trunc
        assert np.abs(hvx[:, :, 2:] - ovx[:, :, 2:]).max(initial=0) <= TOL
    return ok, ovx


def test_wall_scene_5mm_bit_exact(oracle):
    ov, hv = _mk(oracle, 0.005)
    tot_upd = 0
    for i in range(3):
        d, rgb, pose = S.wall_frame(i)
        n, nvis, nupd = ov.integrate(d, rgb, pose)
        hv.IntegrateImage(d, rgb, pose)
        tot_upd += nupd
    st = hv.Stats()
    assert st["frames"] == 3 and st["voxels_updated"] == tot_upd
    _compare(oracle, ov, hv)


def test_room_scene_rotating_poses(oracle):
    ov, hv = _mk(oracle, 0.005)
    sel = upd = 0
    for i in (0, 7, 14, 140):
        d, rgb, pose = S.room_frame(i)
        n, nvis, nupd = ov.integrate(d, rgb, pose)
        hv.IntegrateImage(d, rgb, pose)
        sel += n
        upd += nupd
    st = hv.Stats()
    assert st["blocks_selected"] == sel and st["voxels_visited"] == 512 * sel and st["voxels_updated"] == upd
    _compare(oracle, ov, hv)


def test_prepare_cubes_order_and_bounding(oracle):
    ov, hv = _mk(oracle, 0.01)
    d, rgb, pose = S.room_frame(33)
    oids, ocand = ov.prepare_cubes(d, pose)
    hids, hcand = hv.PrepareCubes(d, pose, return_candidates=True)
    assert ocand == hcand
    assert np.array_equal(oids, hids), "cube_id_list differs (content or loop order)"
    omx, omn, oin = oracle.compute_bounding(ov.cam, d, pose)
    hmx, hmn, hin = hv.ComputeBounding(d, pose)
    assert oin == hin and np.array_equal(omx, hmx) and np.array_equal(omn, hmn)
    assert hv.BlockCount() == ov.block_count() == len(oids)
    assert hv.HasCube(oids[0]) and not hv.HasCube((10 ** 6, 0, 0))


def test_uint16_depth_and_holes(oracle):
    ov, hv = _mk(oracle, 0.01)
    d, rgb, pose = S.room_frame(5)
    d16 = np.round(d * 1000.0).astype(np.uint16)
    d16[100:200, 300:420] = 0          # a hole
    d16[::7, ::5] = 0                   # ragged validity
    ov.integrate(d16, rgb, pose)
    hv.IntegrateImage(d16, rgb, pose)
    df = d.copy()
    df[50:60, :] = np.nan               # NaN depth is skipped by both comparisons
    df[300:, 600:] = -1.0
    ov.integrate(df, rgb, S.room_pose(9))
    hv.IntegrateImage(df, rgb, S.room_pose(9))
    _compare(oracle, ov, hv)


def test_empty_and_tiny_inputs(oracle):
    cam = (128.7, 128.8, 79.7, 59.6, 160, 120, 1000.0)
    ov, hv = _mk(oracle, 0.02, cam=cam)
    z = np.zeros((120, 160), np.float32)
    rgb = np.zeros((120, 160, 3), np.uint8)
    assert ov.integrate(z, rgb, np.eye(4))[0] == 0
    hv.IntegrateImage(z, rgb, np.eye(4))
    assert hv.BlockCount() == 0
    d = np.full((120, 160), 1.5, np.float32)
    d[:, 80:] = 7.0  # beyond the far plane: outside the frustum for bounding, still integrated if selected
    rgb[:] = (10, 200, 90)
    pose = S.room_pose(77)
    ov.integrate(d, rgb, pose)
    hv.IntegrateImage(d, rgb, pose)
    _compare(oracle, ov, hv)


def test_merge_upload_download(oracle):
    ov1, hv1 = _mk(oracle, 0.01)
    ov2, hv2 = _mk(oracle, 0.01)
    for i in (0, 20):
        d, rgb, pose = S.room_frame(i)
        ov1.integrate(d, rgb, pose); hv1.IntegrateImage(d, rgb, pose)
    for i in (10, 30, 60):
        d, rgb, pose = S.room_frame(i)
        ov2.integrate(d, rgb, pose); hv2.IntegrateImage(d, rgb, pose)
    # SetCubeMap / GetCubeMap round trip
    k2, v2 = hv2.GetCubeMap()
    hv3 = I.CubeHandler(); hv3.SetVoxelResolution(0.01); hv3.SetCubeMap(k2, v2)
    k3, v3 = hv3.GetCubeMap()
    assert np.array_equal(k2, k3) and np.array_equal(v2.view(np.uint32), v3.view(np.uint32))
    # Merge == CubeHandler::Merge
    assert ov1.merge(ov2) == 0
    hv1.Merge(hv2)
    _compare(oracle, ov1, hv1)
    # resolution mismatch: warning, untouched (CubeHandler.h:147-151)
    hv4 = I.CubeHandler(); hv4.SetVoxelResolution(0.02)
    before = hv1.BlockCount()
    hv1.Merge(hv4)
    assert hv1.BlockCount() == before


def test_volume_grows_like_the_reference_map(oracle):
    """CubeMap is a std::unordered_map that grows with the scene (CubeHandler.h:22, CubeHandler.cpp:181-190).  A volume
    created with room for 1024 blocks fuses the 5 mm wall scene (23 706 blocks): the batch that exhausts the pool poisons
    the stream on the device (nothing is fused from it on), the host grows pool + hash table and replays from that batch --
    several times here -- and the result is bit-equal to the oracle's: no frame lost, none applied twice."""
    hv = I.CubeHandler(max_blocks=1024)
    hv.SetVoxelResolution(0.005)
    ov = oracle.Volume(voxel_res=0.005)
    for i in range(5):
        d, rgb, pose = S.wall_frame(i)
        hv.IntegrateImage(d, rgb, pose)
        ov.integrate(d, rgb, pose)
    _compare(oracle, ov, hv)
    anchor = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_wall_anchor.json")))
    assert hv.BlockCount() == anchor["blocks"]
    st = hv.Stats()
    assert st["frames"] == 5 and st["voxels_updated"] == int(ov.export()[1][..., 1].astype(np.float64).sum())
    # the same through the device-resident sequence path, growth in the middle of a 40-frame call (several batches in flight)
    import torch
    dev = torch.device("cuda:0")
    depth, rgb, poses = S.room_sequence_torch(0, 40, dev)
    torch.cuda.synchronize()
    hs = I.CubeHandler(max_blocks=2048); hs.SetVoxelResolution(0.01)
    hs.IntegrateSequence(depth, rgb, poses)
    os_ = oracle.Volume(voxel_res=0.01)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    upd = 0
    for k in range(40):
        upd += os_.integrate(dn[k], cn[k], poses[k])[2]
    _compare(oracle, os_, hs)
    assert hs.Stats()["voxels_updated"] == upd and hs.Stats()["frames"] == 40
    # PrepareCubes, AddCubes, Merge and Transform outgrow their pools too
    hp = I.CubeHandler(max_blocks=512); hp.SetVoxelResolution(0.005)
    d, rgb, pose = S.wall_frame(0)
    ids = hp.PrepareCubes(d, pose)
    ref_ids = oracle.Volume(voxel_res=0.005).prepare_cubes(d, pose)
    ref_ids = ref_ids[0] if isinstance(ref_ids, tuple) else ref_ids
    assert np.array_equal(np.asarray(ids[0] if isinstance(ids, tuple) else ids), ref_ids)
    hm = I.CubeHandler(max_blocks=256); hm.SetVoxelResolution(0.01)
    hm.Merge(hs)
    _compare(oracle, os_, hm)
    hu = I.CubeHandler(max_blocks=256); hu.SetVoxelResolution(0.01)
    hu.SetCubeMap(*hs.GetCubeMap())
    _compare(oracle, os_, hu)


def test_device_resident_sequence_matches_per_frame(oracle):
    import torch
    dev = torch.device("cuda:0")
    depth, rgb, poses = S.room_sequence_torch(100, 6, dev)
    torch.cuda.synchronize()
    hv_a = I.CubeHandler(); hv_a.SetVoxelResolution(0.005)
    hv_b = I.CubeHandler(); hv_b.SetVoxelResolution(0.005)
    hv_a.IntegrateSequence(depth, rgb, poses)
    ov = oracle.Volume(voxel_res=0.005)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    for k in range(6):
        hv_b.IntegrateImage(dn[k], cn[k], poses[k])
        ov.integrate(dn[k], cn[k], poses[k])
    ka, va = hv_a.GetCubeMap(); kb, vb = hv_b.GetCubeMap()
    assert np.array_equal(ka, kb) and np.array_equal(va.view(np.uint32), vb.view(np.uint32))
    _compare(oracle, ov, hv_a)


def test_sum_form_merge_kernels_match_reference_merge(oracle):
    """The device steps of the multi-GPU merge (keys_device / pack_sum / unpack_sum) on one GPU:
    two shards packed in union order, summed (what the RCCL reduce does), normalised on the 'root'.
    Must equal CubeHandler::Merge of the shards: keys + weights exact, values within 1e-4."""
    import torch
    from onepiece_amd import distributed as D
    dev = torch.device("cuda:0")
    ov1, hv1 = _mk(oracle, 0.01)
    ov2, hv2 = _mk(oracle, 0.01)
    for i in (0, 15, 30):
        d, rgb, pose = S.room_frame(i)
        ov1.integrate(d, rgb, pose); hv1.IntegrateImage(d, rgb, pose)
    for i in (45, 60):
        d, rgb, pose = S.room_frame(i)
        ov2.integrate(d, rgb, pose); hv2.IntegrateImage(d, rgb, pose)
    o1, o2 = D.HipVolumeOps(hv1, dev), D.HipVolumeOps(hv2, dev)
    k1, k2 = o1.keys(), o2.keys()
    assert k1.shape[0] == hv1.BlockCount() and k2.shape[0] == hv2.BlockCount()
    union = torch.unique(torch.cat([k1, k2]), dim=0).contiguous()
    total = o1.pack_sum(union) + o2.pack_sum(union)
    torch.cuda.synchronize()
    o1.unpack_sum(union, total.contiguous())
    assert ov1.merge(ov2) == 0
    _compare(oracle, ov1, hv1, exact=False)
    # world_size == 1 path of merge_volumes is a no-op that reports the block count
    assert D.merge_volumes(o2) == hv2.BlockCount()


def test_batched_sequence_over_batch_boundary_matches_oracle(oracle):
    """37 device-resident frames = a batch of 32 + a remainder of 5 that waits in the queue (the second pass joins it: 5 + 27, then 10
    launched by the accessor); a voxel sees its frames in order inside a batch, so the result must stay bit-identical to the
    oracle's frame-by-frame fusion.  Includes frames that revisit the same blocks."""
    import torch
    dev = torch.device("cuda:0")
    depth, rgb, poses = S.room_sequence_torch(990, 37, dev)   # crosses the orbit wrap-around (frames 990..1026)
    torch.cuda.synchronize()
    ov, hv = _mk(oracle, 0.01)
    hv.IntegrateSequence(depth, rgb, poses)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    sel = upd = 0
    for k in range(37):
        n, _vis, nu = ov.integrate(dn[k], cn[k], poses[k])
        sel += n; upd += nu
    st = hv.Stats()
    assert st["frames"] == 37 and st["blocks_selected"] == sel and st["voxels_updated"] == upd
    _compare(oracle, ov, hv)
    # a second pass over the same frames (every voxel already valid) stays exact too
    hv.IntegrateSequence(depth, rgb, poses)
    for k in range(37):
        ov.integrate(dn[k], cn[k], poses[k])
    _compare(oracle, ov, hv)


def test_prepare_cubes_leaves_no_pending_selection(oracle):
    """PrepareCubes allocates blocks but must not leak its selection into the next IntegrateImage."""
    ov, hv = _mk(oracle, 0.01)
    d0, c0, p0 = S.room_frame(0)
    d1, c1, p1 = S.room_frame(400)
    ov.prepare_cubes(d0, p0); hv.PrepareCubes(d0, p0)
    ov.integrate(d1, c1, p1); hv.IntegrateImage(d1, c1, p1)
    _compare(oracle, ov, hv)


def _bench_line(out):
    """The LAST stdout line must be the result object, strict JSON, under 8 KB (the driver parses exactly that line)."""
    import json
    line = out.stdout.rstrip("\n").splitlines()[-1]
    assert line.startswith("{") and len(line) < 8192, (len(line), line[:200])
    return json.loads(line)


def test_bench_multi_rank_flow_on_one_gpu(oracle, tmp_path):
    """bench.py --gpus 2 end to end (frame sharding, per-rank fusion, key all_gather, union, sum-form
    pack, the exchange, normalise on rank 0, max-over-ranks timing) with both ranks on cuda:0 and gloo
    standing in for RCCL (RCCL refuses two ranks on one device): the torch.distributed mirror of the merge."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ONEPIECE_BENCH_SINGLE_DEVICE="1", ONEPIECE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames-per-step", "20", "--no-icp", "--detail-file", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _bench_line(out)
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    mg = r["multi_gpu"]
    assert mg["merge_impl"] == "torch" and mg["backend"] == "gloo" and mg["ranks"] == 2
    assert mg["union_blocks"] and mg["union_blocks"] >= r["per_frame"]["blocks_selected"]
    # rank 0's volume now holds the union
    assert r["per_frame"]["final_blocks_rank0"] == mg["union_blocks"]
    assert 0 < r["roofline"]["frac"] <= 1 and r["roofline"]["bound"] == "hbm"
    full = json.load(open(tmp_path / "detail.json"))
    assert full["merge_union_blocks"] == mg["union_blocks"] and full["value"] == pytest.approx(r["value"], rel=1e-3)


@pytest.mark.parametrize("algorithm", ["owner", "dense"])
def test_bench_distributed_path_over_rccl_with_one_rank(tmp_path, algorithm):
    """The WHOLE distributed path of bench.py on real RCCL with the one rank a single-GPU box can host (ONEPIECE_BENCH_FORCE_DIST=1; two ranks
    on one device are refused by RCCL): process group over backend "nccl", the bench's OWN communicator (ncclGetUniqueId -> carried over the
    process group -> ncclCommInitRank in torch's librccl), and the PRODUCT's merge on it -- op_volume_merge_rccl (csrc/merge_rccl.hip): the
    collectives of either algorithm (owner: count matrix all-gather, partition sums, gather; dense: all-gather of counts and keys, the
    reduce issued in slices), normalisation on the root.  The merged volume keeps every block."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ONEPIECE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--frames-per-step", "40", "--no-icp", "--no-tracking", "--no-cpu-baseline", "--merge-algorithm", algorithm, "--detail-file", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _bench_line(out)
    assert r["n_gpus"] == 1 and r["value"] > 0
    mg = r["multi_gpu"]
    assert mg["merge_impl"] == "cabi" and mg["merge_fallback"] is None and mg["rccl_ranks"] == 1          # the library call ran, on a communicator RCCL itself counts one rank in
    assert mg["union_blocks"] == r["per_frame"]["final_blocks_rank0"] > 32768     # more than one slice went through the reduce
    assert mg["backend"] == "nccl" and mg["merge_algorithm"] == algorithm
    full = json.load(open(tmp_path / "detail.json"))
    pr = full["multi_gpu"]["per_rank"][0]
    assert pr["held_blocks"] == mg["union_blocks"] and pr["owned_blocks"] == mg["union_blocks"]
    assert pr["wire_bytes_sent"] == 0        # one rank owns (owner) / is the root of (dense) everything it holds: nothing leaves the device
    assert len(full["multi_gpu"]["per_rank"]) == 1 and pr["frames"] == 80 and pr["merge_ms"] > 0 and pr["fusion_ms"] > 0 and pr["transfer_ms"] > 0


def test_bench_torch_mirror_over_rccl_with_one_rank(tmp_path):
    """--merge-impl torch on real RCCL: the torch.distributed mirror (what a failing library merge falls back to) keeps working."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ONEPIECE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--frames-per-step", "40", "--no-icp", "--no-tracking", "--no-cpu-baseline", "--merge-impl", "torch", "--detail-file", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _bench_line(out)
    mg = r["multi_gpu"]
    assert mg["merge_impl"] == "torch" and mg["backend"] == "nccl" and mg["rccl_ranks"] is None
    assert mg["union_blocks"] == r["per_frame"]["final_blocks_rank0"] > 32768


@pytest.mark.parametrize("ranks,algorithm", [(2, "owner"), (4, "owner"), (3, "dense")])
def test_bench_runs_the_library_merge_with_several_ranks_through_the_multi_process_double(tmp_path, ranks, algorithm):
    """bench.py --gpus N with the PRODUCT's merge (--merge-impl auto -> cabi): N processes, each makes its ncclComm_t the way the 8-GPU run will (ncclGetUniqueId on
    rank 0 -> carried over the process group -> ncclCommInitRank) and ends its timed region in op_volume_merge_rccl_stats (csrc/merge_rccl.hip).  RCCL refuses several
    ranks on one device, so the library named to the bench is the multi-process double (tests/cpp/rccl_double_mp.cpp: collectives through files under /tmp), all ranks
    on cuda:0, the process group over gloo.  Rank 0 ends up with the union; what the ranks report adds up."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp"), "-s"])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ONEPIECE_BENCH_SINGLE_DEVICE="1", ONEPIECE_BENCH_BACKEND="gloo",
               ONEPIECE_BENCH_RCCL_LIBRARY=os.path.join(root, "tests", "cpp", "librccl_double_mp.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--frames-per-step", "20", "--no-icp", "--no-tracking", "--merge-algorithm", algorithm, "--detail-file", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _bench_line(out)
    mg = r["multi_gpu"]
    assert r["n_gpus"] == ranks and mg["merge_impl"] == "cabi" and mg["merge_fallback"] is None and mg["rccl_ranks"] == ranks and mg["merge_algorithm"] == algorithm
    assert mg["union_blocks"] == r["per_frame"]["final_blocks_rank0"] > 0
    full = json.load(open(tmp_path / "detail.json"))
    P = full["multi_gpu"]["per_rank"]
    assert [p["rank"] for p in P] == list(range(ranks)) and all(p["frames"] == 40 and p["local_blocks"] > 0 and p["rccl_ranks"] == ranks for p in P)
    assert mg["union_blocks"] >= max(p["local_blocks"] for p in P)
    if algorithm == "owner":
        assert sum(p["owned_blocks"] for p in P) == mg["union_blocks"]                                   # the partitions cover the union
        assert sum(p["wire_bytes_sent"] for p in P) == sum(p["wire_bytes_received"] for p in P)          # what was sent was received
        assert all(p["held_blocks"] == p["local_blocks"] for p in P)


def test_bench_strong_scaling_four_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 4 --scaling strong (BASELINE configs[4]'s shape: the total is fixed, each rank fuses a quarter) with
    all ranks on cuda:0 over gloo: the job's frame count does not depend on N and rank 0 ends up with the union."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ONEPIECE_BENCH_SINGLE_DEVICE="1", ONEPIECE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--frames-per-step", "40", "--scaling", "strong", "--no-icp", "--no-tracking", "--detail-file", str(tmp_path / "detail.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = _bench_line(out)
    assert r["n_gpus"] == 4 and r["scaling"] == "strong" and r["config"]["frames_per_gpu"] == 20
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] - 80) < 0.1            # 2 x 40 frames in total (the line rounds to 4 digits)
    full = json.load(open(tmp_path / "detail.json"))
    mg = full["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["ranks_in_process_group"] == 4 and [p["rank"] for p in mg["per_rank"]] == [0, 1, 2, 3]
    assert all(p["frames"] == 20 and p["local_blocks"] > 0 for p in mg["per_rank"]) and full["merge_union_blocks"] >= max(p["local_blocks"] for p in mg["per_rank"])
    assert full["per_frame"]["final_blocks_rank0"] == full["merge_union_blocks"] == r["multi_gpu"]["union_blocks"]
    # the merge is the owner-partitioned exchange: the partitions cover the union, what was sent was received, and the exchange moved about 3/4 of what the ranks held
    P = mg["per_rank"]
    assert mg["merge_algorithm"] == "owner" and sum(p["owned_blocks"] for p in P) == full["merge_union_blocks"]
    assert sum(p["wire_bytes_sent"] for p in P) == sum(p["wire_bytes_received"] for p in P)
    B = 10248
    exchange = sum(p["wire_bytes_sent"] for p in P) - sum(p["owned_blocks"] * B for p in P[1:])
    assert abs(exchange / (sum(p["held_blocks"] for p in P) * B) - 0.75) < 0.05


def test_survey_reference_run_anchor_on_the_gpu():
    """The HIP path against committed data only (no live oracle): the statistics the survey recorded from
    the REFERENCE ITSELF on the 5-frame wall scene (tests/golden/survey_wall_anchor.json)."""
    import json, os
    a = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_wall_anchor.json")))
    hv = I.CubeHandler(); hv.SetVoxelResolution(a["voxel_res"])
    d = S.wall_depth(); rgb = np.full((S.H, S.W, 3), 128, np.uint8)
    for i in range(a["frames"]):
        pose = np.eye(4, dtype=np.float32); pose[0, 3] = np.float32(0.01 * i)
        hv.IntegrateImage(d, rgb, pose)
    keys, vox = hv.GetCubeMap()
    w = vox[:, :, 1]
    x = 0
    for k in keys:
        x ^= I.hash_key(*k)
    assert len(keys) == a["blocks"] and int((w > 0).sum()) == a["voxels_with_weight"]
    assert int(w.sum(dtype=np.float64)) == a["weight_sum"] and x == int(a["xor_of_key_hashes"], 16)
    for kx, ky, kz, h in a["hash_known_answers"]:
        assert I.hash_key(kx, ky, kz) == h


def test_far_from_origin_negative_block_ids_and_tum_camera(oracle):
    """Block ids around (-1250, 625, -375) exercise sign extension in the 64-bit hash and the 21-bit key
    packing; the TUM preset exercises cy = 255.3 / depth_scale 5000 (Camera.h:78-92) with uint16 depth."""
    cam = (517.3, 516.5, 318.6, 255.3, 640, 480, 5000.0)
    ov, hv = _mk(oracle, 0.01, cam=cam)
    shift = np.eye(4, dtype=np.float32); shift[:3, 3] = (-100.0, 50.0, -30.0)
    for i in (3, 9):
        d, rgb, pose = S.room_frame(i)
        d16 = np.round(d * 5000.0).astype(np.uint16)
        pose = (shift @ pose).astype(np.float32)
        ov.integrate(d16, rgb, pose); hv.IntegrateImage(d16, rgb, pose)
    ok, _ = _compare(oracle, ov, hv)
    assert ok[:, 0].max() < -1200 and ok[:, 1].min() > 600


def test_large_image_mi_dataset_camera(oracle):
    """1440x1080 MI_DATASET intrinsics (Camera.h:106-120): more pixels than one KA workgroup row, cx = 756.2."""
    cam = (2209.84366, 2210.23057, 756.24762, 530.00418, 1440, 1080, 1000.0)
    ov, hv = _mk(oracle, 0.02, cam=cam)
    pose = S.room_pose(50)
    d, rgb = S.room_render(pose, width=1440, height=1080, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
    ov.integrate(d, rgb, pose); hv.IntegrateImage(d, rgb, pose)
    _compare(oracle, ov, hv)
    omx, omn, oin = oracle.compute_bounding(ov.cam, d, pose)
    hmx, hmn, hin = hv.ComputeBounding(d, pose)
    assert oin == hin and np.array_equal(omx, hmx) and np.array_equal(omn, hmn)


def test_queued_frames_survive_setters_and_mixed_formats(oracle):
    """Single-frame calls are queued; a setter, a format change or an accessor must flush with the OLD
    settings first -- so the sequence below equals the oracle's strictly sequential semantics."""
    ov, hv = _mk(oracle, 0.01)
    d0, c0, p0 = S.room_frame(0); d1, c1, p1 = S.room_frame(8); d2, c2, p2 = S.room_frame(16)
    hv.IntegrateImage(d0, c0, p0); ov.integrate(d0, c0, p0)
    hv.SetTruncation(0.05)                                  # flushes frame 0 under trunc 0.1
    ov2 = oracle.Volume(ov.cam, voxel_res=0.01, trunc=0.05)
    k, v = ov.export(); ov2.load(k, v)
    d1u = np.round(d1 * 1000).astype(np.uint16)
    hv.IntegrateImage(d1u, c1, p1); ov2.integrate(d1u, c1, p1)   # u16 after f32: separate batch
    hv.IntegrateImage(d2, c2, p2); ov2.integrate(d2, c2, p2)
    assert hv.Stats()["frames"] == 3
    _compare(oracle, ov2, hv)


def test_host_buffers_are_only_borrowed_for_the_call(oracle):
    """SURVEY 8b "Ownership": IntegrateImage borrows the cv::Mat buffers for the call only.  Frames are
    queued inside the library, so the caller scribbling over its buffers right after the call returns
    must not change the result."""
    ov, hv = _mk(oracle, 0.01)
    buf_d = np.empty((S.H, S.W), np.float32)
    buf_c = np.empty((S.H, S.W, 3), np.uint8)
    for i in (0, 12, 24, 36, 48):
        d, c, p = S.room_frame(i)
        ov.integrate(d, c, p)
        buf_d[:] = d; buf_c[:] = c
        hv.IntegrateImage(buf_d, buf_c, p)
        buf_d[:] = -7.0; buf_c[:] = 255          # the caller reuses its buffers immediately
    _compare(oracle, ov, hv)


def test_full_size_invariants_without_the_oracle():
    """BASELINE sizes (640x480, 5 mm, 160 frames resident in HBM): properties that hold at any size and
    need no CPU run -- weight checksum, allocated-set = union of the per-frame selections,
    shard-and-merge against sequential fusion, upload/download idempotence."""
    import torch
    n = 160
    dev = torch.device("cuda", 0)
    depth, rgb, poses = S.room_sequence_torch(40, n, dev)
    hv = I.CubeHandler(max_blocks=1 << 18); hv.SetVoxelResolution(0.005)
    hv.IntegrateSequence(depth, rgb, poses)
    st = hv.Stats()
    keys, vox = hv.GetCubeMap()
    assert st["frames"] == n and st["voxels_visited"] == 512 * st["blocks_selected"]
    # every update adds weight 1 (TSDFVoxel::operator+, w + 1): the weight plane is a checksum of the updates
    assert int(vox[:, :, 1].astype(np.float64).sum()) == st["voxels_updated"]
    assert np.all(vox[:, :, 1] == np.round(vox[:, :, 1])) and vox[:, :, 1].max() <= n
    observed = vox[:, :, 1] > 0
    assert np.all(np.abs(vox[:, :, 0][observed]) < 0.1 + 1e-6) and np.all(vox[:, :, 0][~observed] == 999)
    assert np.all((vox[:, :, 2:][observed] >= 0) & (vox[:, :, 2:][observed] <= 1))
    # allocated block set == union over frames of PrepareCubes' lists (CubeHandler.cpp:181-190), sampled frames
    probe = I.CubeHandler(max_blocks=1 << 18); probe.SetVoxelResolution(0.005)
    keyset = set(map(tuple, keys.tolist()))
    for i in (0, 57, n - 1):
        ids = probe.PrepareCubes(depth[i], poses[i])
        assert set(map(tuple, np.asarray(ids).tolist())) <= keyset
    # shard-and-merge (two contiguous halves, CubeHandler::Merge) vs sequential: same keys, same weights exactly,
    # sdf / colour within 1e-4 (mean of means vs running mean)
    a = I.CubeHandler(max_blocks=1 << 18); a.SetVoxelResolution(0.005)
    b = I.CubeHandler(max_blocks=1 << 18); b.SetVoxelResolution(0.005)
    a.IntegrateSequence(depth[:n // 2], rgb[:n // 2], poses[:n // 2])
    b.IntegrateSequence(depth[n // 2:], rgb[n // 2:], poses[n // 2:])
    a.Merge(b)
    mk, mv = a.GetCubeMap()
    assert np.array_equal(mk, keys) and np.array_equal(mv[:, :, 1], vox[:, :, 1])
    assert np.abs(mv[:, :, 0] - vox[:, :, 0])[observed].max() <= TOL * 0.1
    assert np.abs(mv[:, :, 2:] - vox[:, :, 2:])[observed].max() <= TOL
    # upload/download idempotence
    c = I.CubeHandler(max_blocks=1 << 18); c.SetVoxelResolution(0.005); c.SetCubeMap(keys, vox)
    ck, cv = c.GetCubeMap()
    assert np.array_equal(ck, keys) and np.array_equal(cv.view(np.uint32), vox.view(np.uint32))


def test_random_frames_fuzz(oracle):
    """Unstructured inputs: random depth (incl. zeros, negatives, NaN, huge values), random colours, random rigid poses
    (large rotations, camera inside/outside the volume) -- bit-identical to the oracle frame after frame."""
    rng = np.random.default_rng(20260928)
    cam = (120.0, 118.0, 79.3, 60.7, 160, 120, 1000.0)
    ov, hv = _mk(oracle, 0.02, cam, max_blocks=1 << 16)
    for k in range(12):
        d = rng.uniform(0.3, 4.5, (120, 160)).astype(np.float32)
        d[rng.random((120, 160)) < 0.1] = 0.0
        d[rng.random((120, 160)) < 0.02] = -1.0
        d[rng.random((120, 160)) < 0.01] = np.nan
        d[rng.random((120, 160)) < 0.01] = 1e6
        if k % 3 == 0:                                   # smooth surface so that many voxels actually update
            u, v = np.meshgrid(np.arange(160), np.arange(120))
            d = (1.5 + 0.5 * np.sin(u / 17.0 + k) * np.cos(v / 13.0)).astype(np.float32)
        c = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
        x = np.concatenate([rng.uniform(-0.3, 0.3, 3), rng.uniform(-1.0, 1.0, 3)]).astype(np.float32)
        pose = oracle.se3_exp(x)
        ov.integrate(d, c, pose)
        hv.IntegrateImage(d, c, pose)
        if k % 4 == 3:
            _compare(oracle, ov, hv)
    _compare(oracle, ov, hv)


def test_projection_with_shared_reciprocal_equals_plain_division():
    """The kernels evaluate (fx*X)/Z and (fy*Y)/Z with one refined reciprocal (csrc/volume_core.hpp: project_pixel).  On the device,
    over dense random operands, operands a few ulp around every kind of rounding boundary, and specials (0, inf, NaN,
    denormals, |Z| outside the fast window), the resulting pixel equals the reference formula with the plain IEEE division
    whenever either of them is a pixel an image could contain (except for Z = +-inf, where no sdf can be finite)."""
    import ctypes as C
    from onepiece_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(7)
    fx, fy, cx, cy = np.float32(514.817), np.float32(515.375), np.float32(318.771), np.float32(238.447)

    def run(X, Y, Z, fx=fx, fy=fy, cx=cx, cy=cy):
        X, Y, Z = (np.ascontiguousarray(a, np.float32) for a in (X, Y, Z))
        out = np.empty((len(X), 4), np.int32)
        vp = lambda a: C.c_void_p(a.ctypes.data)
        L.check(lib.op_debug_project_uv(float(fx), float(fy), float(cx), float(cy), vp(X), vp(Y), vp(Z), len(X), 0, vp(out)))
        return out

    def agree(out, limits=(640, 480)):
        # the kernels' projection reports the pixel only when BOTH coordinates are inside the (640 x 480) image, INT_MIN
        # otherwise; the reference formula's pair is inside exactly then, and equal
        r_in = (out[:, 2] >= 0) & (out[:, 2] < limits[0]) & (out[:, 3] >= 0) & (out[:, 3] < limits[1])
        f_in = out[:, 0] != np.iinfo(np.int32).min
        bad = (r_in != f_in) | (r_in & ((out[:, 0] != out[:, 2]) | (out[:, 1] != out[:, 3])))
        assert not bad.any(), (np.flatnonzero(bad)[:5], out[bad][:5])
        return r_in.mean()

    n = 1 << 21
    sign = lambda m: np.where(rng.random(m) < 0.5, -1.0, 1.0)
    # 1. camera-like operands: Z in [0.05, 20] m, X / Y within a few field-of-views
    Z = rng.uniform(0.05, 20.0, n) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
    frac = agree(run(rng.uniform(-2, 2, n) * np.abs(Z), rng.uniform(-2, 2, n) * np.abs(Z), Z))
    assert frac > 0.05
    # 2. log-uniform magnitudes across and beyond the fast window, every sign
    Z = sign(n) * np.exp2(rng.uniform(-80, 80, n))
    agree(run(sign(n) * np.exp2(rng.uniform(-149, 120, n)), sign(n) * np.exp2(rng.uniform(-149, 120, n)), Z))
    q = np.exp2(rng.uniform(-60, 20, n))                         # quotients from far below a pixel to far beyond an image
    agree(run(sign(n) * q * np.abs(Z) / fx, sign(n) * q * np.abs(Z) / fy, Z))
    # 3. a few ulp around the rounding boundaries: (fx*X)/Z = k + 0.5 - frac(cx) etc.
    for c, f, col in ((cx, fx, 0), (cy, fy, 1)):
        k = rng.integers(-700, 700, n).astype(np.float64)
        tgt = (k + rng.choice([-0.5, 0.5, 1.5], n) - (np.float64(c) - np.floor(np.float64(c)))).astype(np.float32)
        Z = (sign(n) * rng.uniform(0.3, 6.0, n)).astype(np.float32)
        X = (tgt.astype(np.float64) * Z.astype(np.float64) / np.float64(f)).astype(np.float32)
        X = (X.view(np.int32) + rng.integers(-3, 4, n).astype(np.int32)).view(np.float32)
        other = rng.uniform(-1, 1, n).astype(np.float32)
        agree(run(X, other, Z) if col == 0 else run(other, X, Z))
    # 4. specials in every position
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3e38, -3e38, 1.0, -1.0, 2.0 ** -60, 2.0 ** 60,
                   2.0 ** -61, 2.0 ** 59, 0.5, 1234.5], np.float32)
    g = np.stack(np.meshgrid(sp, sp, sp, indexing="ij"), -1).reshape(-1, 3)
    with np.errstate(all="ignore"):
        out = run(g[:, 0], g[:, 1], g[:, 2])
    fin = ~np.isinf(g[:, 2])
    agree(out[fin])
    # Z = +-inf is the one operand where the two differ: the plain quotient is +-0 (the principal point), the shared
    # reciprocal gives NaN (no pixel).  Neither can matter: the voxel's sdf is d - Z = -+inf, never inside the truncation.
    assert np.all(out[~fin][:, :2] == np.iinfo(np.int32).min)
    # 5. an intrinsic whose fractional part is 0.5 takes the double-precision rounding path
    agree(run(rng.uniform(-2, 2, 4096), rng.uniform(-2, 2, 4096), rng.uniform(0.2, 5, 4096), cx=np.float32(320.5), cy=np.float32(240.5)))


def test_torch_tensors_are_final_before_library_kernels_read_them():
    """The library's objects own their HIP streams; a torch tensor handed to them may still be the target of work queued
    on torch's stream.  Here the depth frames are written by a copy that sits behind ~30 ms of queued matrix products:
    the mirrors must drain torch's stream before the fusion kernels read the tensor (L.torch_ready), not fuse zeros."""
    import torch
    dev = torch.device("cuda", 0)
    depth, rgb, poses = S.room_sequence_torch(300, 6, dev)
    torch.cuda.synchronize()
    ref = I.CubeHandler(max_blocks=1 << 17); ref.SetVoxelResolution(0.01)
    ref.IntegrateSequence(depth, rgb, poses)
    want = ref.Stats()
    late = torch.zeros_like(depth)
    junk = torch.randn(4096, 4096, device=dev)
    for _ in range(30):
        junk = (junk @ junk) * 1e-4
    late.copy_(depth)                     # queued behind the products, not done yet
    hv = I.CubeHandler(max_blocks=1 << 17); hv.SetVoxelResolution(0.01)
    hv.IntegrateSequence(late, rgb, poses)
    got = hv.Stats()
    got.pop("integrate_shader_cycles"); want.pop("integrate_shader_cycles")   # a duration, not a result
    assert got == want and got["voxels_updated"] > 0
    del junk


def test_integrate_cubes_is_integrate_image_without_the_selection(oracle):
    """Integrator::IntegrateImage for a caller-chosen cube list (op_volume_integrate_cubes): with the list PrepareCubes
    returns it IS CubeHandler::IntegrateImage -- bit-equal to the oracle; a cube outside the selection is allocated and
    keeps its default voxels; listing a cube twice fuses it once; a second frame through the list of ITS selection
    continues the running mean bit-exactly."""
    cam = (S.FX / 2, S.FY / 2, S.CX / 2, S.CY / 2, S.W // 2, S.H // 2, 1000.0)
    ov, hv = _mk(oracle, 0.01, cam)
    for i in (20, 24):
        pose = S.room_pose(i)
        d, rgb = S.room_render(pose, width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
        ids = hv.PrepareCubes(d, pose)                     # allocates the selection, fuses nothing
        ov.integrate(d, rgb, pose)
        hv.IntegrateCubes(d, rgb, pose, np.concatenate([ids, ids[:5]]))
        _compare(oracle, ov, hv)
    n = hv.BlockCount()
    far = np.array([[900, 900, 900]], np.int32)
    hv.IntegrateCubes(d, rgb, pose, far)                   # not in view: allocated, every voxel stays at the default
    assert hv.BlockCount() == n + 1 and hv.HasCube((900, 900, 900))
    hk, hvx = hv.GetCubeMap()
    blk = hvx[np.where((hk == far[0]).all(1))[0][0]]
    assert np.all(blk[:, 0] == 999) and np.all(blk[:, 1] == 0) and np.all(blk[:, 2:] == -1)
    hv.IntegrateCubes(d, rgb, pose, np.zeros((0, 3), np.int32))   # empty list: nothing happens
    assert hv.BlockCount() == n + 1


def test_launch_level_counters(oracle):
    """op_volume_stats_launches: blocks read and voxels written per k_integrate launch (the batch-level byte model of the
    roofline).  One frame per launch: blocks read = len(cube_id_list), voxels written = voxels updated (each changed
    voxel is written once); a 3-frame batch reads the union of the three lists once and writes every changed voxel once."""
    ov, hv = _mk(oracle, 0.01)
    frames = [S.room_frame(i) for i in (0, 2, 4)]
    sel = upd = 0
    for d, rgb, pose in frames:
        n, _vis, nupd = ov.integrate(d, rgb, pose)
        hv.IntegrateImage(d, rgb, pose); hv.Synchronize()
        sel += n; upd += nupd
    st = hv.Stats()
    assert st["launches"] == 3 and st["blocks_read"] == sel == st["blocks_selected"] and st["voxels_written"] == upd == st["voxels_updated"]
    assert 3 * 5e3 < st["integrate_shader_cycles"] < 3 * 5e6      # three launches of 5 us .. 2 ms each at ~2 GHz
    import torch
    hv.Clear()
    dev = torch.device("cuda", 0)
    dd = torch.from_numpy(np.stack([f[0] for f in frames])).to(dev); cc = torch.from_numpy(np.stack([f[1] for f in frames])).to(dev)
    hv.IntegrateSequence(dd, cc, np.stack([f[2] for f in frames]))
    st = hv.Stats()
    keys, vox = ov.export()
    assert st["launches"] == 1 and st["frames"] == 3 and st["blocks_selected"] == sel and st["voxels_updated"] == upd
    assert st["blocks_read"] == len(keys) and st["voxels_written"] == int((vox[:, :, 1] > 0).sum())
    _compare(oracle, ov, hv)


def test_fusion_after_upload_mixes_general_and_fast_updates_bit_exactly(oracle):
    """After SetCubeMap the blocks that existed take the general update (the reference's two-branch form with true divisions), blocks
    allocated later keep the shared-reciprocal one -- chosen per block inside one launch.  The uploaded blocks here carry data the fast
    path may not see (fractional and huge weights, sdf exactly 1, denormal colours): the result equals the oracle's bit for bit."""
    cam = (S.FX / 2, S.FY / 2, S.CX / 2, S.CY / 2, S.W // 2, S.H // 2, 1000.0)
    ov, hv = _mk(oracle, 0.01, cam)
    frames = []
    for i in (0, 6, 12, 18, 24, 30):
        pose = S.room_pose(i)
        d, rgb = S.room_render(pose, width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
        frames.append((d, rgb, pose))
    for d, rgb, pose in frames[:2]:
        ov.integrate(d, rgb, pose); hv.IntegrateImage(d, rgb, pose)
    keys, vox = hv.GetCubeMap()
    rng = np.random.default_rng(7)
    odd = vox.copy()
    pick = rng.random(odd.shape[:2]) < 0.05
    odd[..., 1][pick] = rng.choice(np.array([0.37, 2.5, 3.0e7, 1.0e-3], np.float32), pick.sum())
    odd[..., 0][rng.random(odd.shape[:2]) < 0.01] = 1.0
    odd[..., 2][rng.random(odd.shape[:2]) < 0.01] = 1e-41
    hv.SetCubeMap(keys, odd)
    ov.load(keys, odd)
    n_before = hv.BlockCount()
    for d, rgb, pose in frames[2:]:
        ov.integrate(d, rgb, pose); hv.IntegrateImage(d, rgb, pose)
    assert hv.BlockCount() > n_before + 200          # later frames allocate blocks of their own (fast update) next to the uploaded ones
    _compare(oracle, ov, hv)


def test_coarse_selection_test_never_drops_a_block(oracle):
    """k_select rejects whole 4 x 4 x 4 groups of blocks against the depth range of the image tiles they project into before it runs
    the reference's per-block test.  The selected list must stay the reference's, entry for entry, on the inputs that stress that
    shortcut: image sizes that are no multiple of the tile / of KA's rectangle, slanted and stepped surfaces (wide depth range per tile),
    surfaces closer than the near plane and within centimetres of the camera, holes / NaN / negative / huge depths, 16-bit depth,
    cameras whose principal point lies outside the image, random rigid poses."""
    rng = np.random.default_rng(20260929)
    sizes = [(160, 120), (37, 29), (65, 17), (16, 16), (129, 97), (200, 3), (1, 1), (64, 16), (63, 47)]
    n_sel = 0
    for case in range(27):
        w, h = sizes[case % len(sizes)]
        f = rng.uniform(0.6, 1.4) * w
        cx, cy = (w * rng.uniform(0.3, 0.7), h * rng.uniform(0.3, 0.7)) if case % 5 else (-0.3 * w, 1.4 * h)
        cam = (float(f), float(f * rng.uniform(0.9, 1.1)), float(cx), float(cy), w, h, 1000.0)
        res = float(rng.choice([0.004, 0.01, 0.02]))
        ov, hv = _mk(oracle, res, cam=cam, max_blocks=1 << 15)
        u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
        kind = case % 6
        if kind == 0:    # slanted plane
            d = 1.0 + 0.004 * u * rng.uniform(-1, 1) + 0.006 * v * rng.uniform(-1, 1)
        elif kind == 1:  # steps: foreground object in front of a wall
            d = np.where((u // max(1, w // 5) + v // max(1, h // 3)) % 2 == 0, 0.7, 2.6)
        elif kind == 2:  # within centimetres of the camera .. near plane
            d = np.where((u + v) % 5 == 0, rng.uniform(0.02, 0.3), 0.55 + 2.5 * u / max(w, 1) + 0.05 * np.sin(v / 7.0))
        elif kind == 3:  # noise
            d = rng.uniform(0.3, 4.5, (h, w))
        elif kind == 4:  # bowl
            d = 1.2 + 0.8 * (((u - w / 2) / max(w, 1)) ** 2 + ((v - h / 2) / max(h, 1)) ** 2)
        else:            # far wall beyond the far plane next to a near one
            d = np.where(u < w / 2, 6.5, 0.9)
        d = np.broadcast_to(np.asarray(d, np.float32), (h, w)).copy()
        d[rng.random((h, w)) < 0.05] = 0.0
        d[rng.random((h, w)) < 0.01] = np.nan
        d[rng.random((h, w)) < 0.01] = -2.0
        d[rng.random((h, w)) < 0.005] = 1e7
        x = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-1.5, 1.5, 3)]).astype(np.float32)
        pose = oracle.se3_exp(x) if case % 4 else np.eye(4, dtype=np.float32)
        if case % 7 == 3:
            d = np.round(np.nan_to_num(np.clip(d, 0, 60.0)) * 1000.0).astype(np.uint16)
        oids, ocand = ov.prepare_cubes(d, pose)
        hids, hcand = hv.PrepareCubes(d, pose, return_candidates=True)
        assert ocand == hcand, "case %d: candidate count" % case
        assert np.array_equal(oids, hids), "case %d (%dx%d, %s voxel): cube_id_list differs" % (case, w, h, res)
        n_sel += len(oids)
        omx, omn, oin = oracle.compute_bounding(ov.cam, d, pose)
        hmx, hmn, hin = hv.ComputeBounding(d, pose)
        assert oin == hin and np.array_equal(omx, hmx) and np.array_equal(omn, hmn), "case %d: bounding" % case
        # and the fused result (KA's wide / narrow loads, tile outputs, KB, KC) on a second view of the same surface
        c = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ov.integrate(d, c, pose)
        hv.IntegrateImage(d, c, pose)
        _compare(oracle, ov, hv)
    assert n_sel > 20000, "the cases must select something (%d)" % n_sel


def test_sequence_calls_share_one_queue_with_single_frames(oracle):
    """op_volume_integrate_sequence and op_volume_integrate fill the same queue: calls of 3, 1 (single frame), 40, 7 frames, an explicit
    Flush in between and a format change give launches of every size from 1 to 32 -- the volume equals frame-by-frame fusion."""
    import torch
    dev = torch.device("cuda:0")
    n = 3 + 1 + 40 + 7 + 33
    depth, rgb, poses = S.room_sequence_torch(100, n, dev)
    torch.cuda.synchronize()
    ov, hv = _mk(oracle, 0.01)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    k = 0
    hv.IntegrateSequence(depth[k:k + 3], rgb[k:k + 3], poses[k:k + 3]); k += 3
    hv.IntegrateImage(depth[k], rgb[k], poses[k]); k += 1                      # device tensors, single-frame call: same queue
    hv.IntegrateSequence(depth[k:k + 40], rgb[k:k + 40], poses[k:k + 40]); k += 40
    hv.Flush()                                                                 # 12 queued frames go now
    hv.IntegrateSequence(depth[k:k + 7], rgb[k:k + 7], poses[k:k + 7]); k += 7
    d16 = (depth[k:k + 33] * 1000.0).round().to(torch.int16)                   # format change flushes the 7
    hv.IntegrateSequence(d16, rgb[k:k + 33], poses[k:k + 33])
    for i in range(k):
        ov.integrate(dn[i], cn[i], poses[i])
    d16n = d16.cpu().numpy().view(np.uint16)
    for i in range(33):
        ov.integrate(d16n[i], cn[k + i], poses[k + i])
    st = hv.Stats()
    assert st["frames"] == n and st["launches"] == 5, st   # 32 + 12 | 7 | 32 + 1
    _compare(oracle, ov, hv)


def test_poses_whose_bottom_row_is_not_0001(oracle):
    """TransformPoints divides by w (Geometry.cpp:24-26); k_prepare_frames skips the division while every lane's w is exactly 1.  Poses with a
    scaled or projective bottom row take the dividing path and must agree with the oracle bit for bit as well (bounding, list, volume)."""
    cam = (128.7, 128.8, 79.7, 59.6, 160, 120, 1000.0)
    d, rgb, _ = S.room_frame(12)
    d = np.ascontiguousarray(d[::4, ::4]); rgb = np.ascontiguousarray(rgb[::4, ::4])
    base = S.room_pose(12).astype(np.float32)
    for bottom in ((0.0, 0.0, 0.0, 2.0), (1e-3, -2e-3, 5e-4, 1.0), (0.0, 0.0, 0.0, 1.0)):
        pose = base.copy(); pose[3] = bottom
        ov, hv = _mk(oracle, 0.02, cam=cam)
        omx, omn, oin = oracle.compute_bounding(ov.cam, d, pose)
        hmx, hmn, hin = hv.ComputeBounding(d, pose)
        assert oin == hin and np.array_equal(omx, hmx) and np.array_equal(omn, hmn), bottom
        oids, ocand = ov.prepare_cubes(d, pose)
        hids, hcand = hv.PrepareCubes(d, pose, return_candidates=True)
        assert ocand == hcand and np.array_equal(oids, hids), bottom
        ov.integrate(d, rgb, pose); hv.IntegrateImage(d, rgb, pose)
        _compare(oracle, ov, hv)


def test_first_host_frame_after_queued_device_frames(oracle):
    """A volume's FIRST host image allocates the staging ring, which launches whatever is queued; the new frame must then take position 0 of
    the next batch, not the stale position (it used to be dropped and the previous batch's frame 0 fused twice).  Device frames from a sequence
    call (their remainder stays queued), then host frames, then device frames again."""
    import torch
    dev = torch.device("cuda:0")
    n = 11
    depth, rgb, poses = S.room_sequence_torch(40, n, dev)
    torch.cuda.synchronize()
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    ov, hv = _mk(oracle, 0.01)
    hv.IntegrateSequence(depth[:5], rgb[:5], poses[:5])        # 5 device frames queued, nothing launched
    hv.IntegrateImage(dn[5], cn[5], poses[5])                  # first host frame: ring allocation flushes the 5
    hv.IntegrateImage(dn[6], cn[6], poses[6])
    hv.IntegrateSequence(depth[7:9], rgb[7:9], poses[7:9])     # device frames join the batch the host frames started
    hv.IntegrateImage(dn[9], cn[9], poses[9])
    hv.IntegrateImage(depth[10], rgb[10], poses[10])
    for i in range(n):
        ov.integrate(dn[i], cn[i], poses[i])
    st = hv.Stats()
    assert st["frames"] == n, st
    _compare(oracle, ov, hv)


def test_sum_form_update_matches_the_reference_to_tolerance(oracle):
    """OP_VOLUME_UPDATE_SUM_FORM (opt-in): a batch's observations of a voxel are summed and averaged into it once per batch.  Same blocks, same
    pixels, same (integer) weights; sdf within 1e-4 of the truncation distance and colour within 1e-4 of the oracle's frame-by-frame running
    mean (north_star's bar; measured: a few 1e-7).  Full batches, a partial one, re-fusion into existing voxels, uint16 depth with holes."""
    import torch
    dev = torch.device("cuda:0")
    n = 75
    depth, rgb, poses = S.room_sequence_torch(200, n, dev)
    d16 = (depth * 1000.0).round().to(torch.int16)
    d16[:, 100:160, 200:300] = 0
    torch.cuda.synchronize()
    dn, cn = d16.cpu().numpy().view(np.uint16), rgb.cpu().numpy()
    ov, hv = _mk(oracle, 0.01)
    hv.SetUpdateMode("sum_form")
    hv.IntegrateSequence(d16[:70], rgb[:70], poses[:70])       # 32 + 32 + 6
    hv.Synchronize()
    hv.IntegrateSequence(d16[70:], rgb[70:], poses[70:])       # a short batch into the same voxels
    upd = 0
    for i in range(n):
        upd += ov.integrate(dn[i], cn[i], poses[i])[2]
    st = hv.Stats()
    assert st["frames"] == n and st["voxels_updated"] == upd      # the same observations
    ok, ovx = _compare(oracle, ov, hv, exact=False)
    hk, hvx = hv.GetCubeMap()
    obs = ovx[:, :, 1] > 0
    e_sdf = np.abs(hvx[:, :, 0] - ovx[:, :, 0])[obs].max() / 0.1
    e_col = np.abs(hvx[:, :, 2:] - ovx[:, :, 2:])[obs].max()
    assert e_sdf <= 1e-5 and e_col <= 1e-5, (e_sdf, e_col)       # far inside the bar: float rounding only
    assert np.array_equal(hvx[~obs].view(np.uint32), ovx[~obs].view(np.uint32))   # unobserved voxels untouched
    # back to the exact update: later frames are applied the reference's way to the sum-form volume (still within tolerance of the oracle)
    hv.SetUpdateMode("exact")
    d, c, p = S.room_frame(200 + n)
    ov.integrate(d, c, p); hv.IntegrateImage(d, c, p)
    _compare(oracle, ov, hv, exact=False)


def test_sum_form_update_after_upload_and_on_the_wall_scene(oracle):
    """The sum form's once-per-batch mean is the GENERAL TSDFVoxel::operator+ (true divisions, IsValid test): it also applies to voxels that
    came from an upload; and the survey's 5-frame wall scene keeps its block and weight statistics."""
    ov, hv = _mk(oracle, 0.005)
    hv.SetUpdateMode("sum_form")
    for i in range(5):
        d, rgb, pose = S.wall_frame(i)
        ov.integrate(d, rgb, pose)
        hv.IntegrateImage(d, rgb, pose)
    ok, ovx = _compare(oracle, ov, hv, exact=False)
    assert len(ok) == 23706 and int((ovx[:, :, 1] > 0).sum()) == 8671233 and int(ovx[:, :, 1].astype(np.float64).sum()) == 37033130
    k, v = hv.GetCubeMap()
    ov2, hv2 = _mk(oracle, 0.005)
    hv2.SetUpdateMode("sum_form")
    hv2.SetCubeMap(k, v); ov2.load(k, v)
    for i in range(5, 8):
        d, rgb, pose = S.wall_frame(i)
        ov2.integrate(d, rgb, pose)
        hv2.IntegrateImage(d, rgb, pose)
    _compare(oracle, ov2, hv2, exact=False)


def test_sum_form_with_a_truncation_of_one_or_more_takes_the_exact_update(oracle):
    """TSDFVoxel::IsValid is sdf < 1 (TSDFVoxel.h:75-78) and the reference re-tests it before EVERY frame: with truncation >= 1 an observation can itself
    be "invalid" and is then REPLACED by the next one (weight 1 again) -- which a once-per-batch mean cannot reproduce.  The sum-form option therefore
    only applies below 1; above, the volume is the exact one bit for bit, option or not."""
    cam = small_camera(4)
    frames = []
    for i in range(6):
        pose = S.room_pose(40 + i)
        d, c = S.room_render(pose, width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
        frames.append((d, c, pose))
    vols = {}
    for mode in ("exact", "sum_form"):
        hcam = I.PinholeCamera()
        hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
        hv = I.CubeHandler(hcam, max_blocks=1 << 16)
        hv.SetVoxelResolution(0.08); hv.SetTruncation(1.5)
        hv.SetUpdateMode(mode)
        for d, c, p in frames:
            hv.IntegrateImage(d, c, p)
        vols[mode] = hv.GetCubeMap()
    ov = oracle.Volume(oracle.make_camera(*cam), voxel_res=0.08, trunc=1.5)
    for d, c, p in frames:
        ov.integrate(d, c, p)
    ok, ox = ov.export()
    for mode in vols:
        k, v = vols[mode]
        assert np.array_equal(k, ok) and np.array_equal(v.view(np.uint32), ox.view(np.uint32)), mode
    assert (ox[..., 0][ox[..., 1] > 0] >= 1).any()        # the case exists in this volume: stored observations that IsValid rejects


def test_batch64_build_on_the_largest_image_it_admits(tmp_path):
    """The documented build option -DOP_MAX_BATCH=64 (64-bit batch masks).  k_integrate addresses a batch's packed frames with 32-bit byte offsets,
    so the 64-frame build admits images of up to 2^23 pixels (the default: 2^24).  On a 4096 x 2048 image -- its limit -- 70 frames fused in
    64-frame launches give bit for bit the volume the default build fuses in 32-frame launches (frames 32..63 of a batch are the ones a wrapped
    offset would have read from the wrong frame), and a 4096 x 4096 camera is refused by the one and admitted by the other."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b64 = os.path.join(root, "onepiece_amd", "libonepiece_hip_b64.so")
    if not os.path.exists(b64):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "onepiece_amd", "csrc"), "b64"])
    script = os.path.join(root, "tests", "tools", "batch_build_check.py")
    res = {}
    for name, lib in (("b32", ""), ("b64", b64)):
        env = dict(os.environ)
        env.pop("ONEPIECE_HIP_LIBRARY", None)
        if lib:
            env["ONEPIECE_HIP_LIBRARY"] = lib
        run = subprocess.run([sys.executable, script, "4096", "2048", "70", "0.04", "4096", "4096"], capture_output=True, text=True, env=env, timeout=900)
        assert run.returncode == 0, run.stdout + run.stderr
        res[name] = json.loads(run.stdout.strip().splitlines()[-1])
    a, b = res["b32"], res["b64"]
    assert a["library"] == "libonepiece_hip.so" and b["library"] == "libonepiece_hip_b64.so"
    assert a["frames"] == b["frames"] == 70 and a["launches"] == 3 and b["launches"] == 2, (a, b)      # 32 + 32 + 6 | 64 + 6
    assert a["blocks"] == b["blocks"] > 200 and a["voxels_updated"] == b["voxels_updated"] > 10 ** 6   # 32 cm blocks: a few hundred hold the room
    assert a["keys_sha"] == b["keys_sha"] and a["voxels_sha"] == b["voxels_sha"]
    assert a["probe_admitted"] is True and b["probe_admitted"] is False and "pixels" in b["probe_error"]


def _fuse(oracle_mod, depth, rgb, poses, mode, res=0.01, chunks=None):
    """The frames through IntegrateSequence with the given selection mode; `chunks`: the sizes of the calls (default: one call)."""
    _ov, hv = _mk(oracle_mod, res)
    hv.SetSelectMode(mode)
    k = 0
    for n in (chunks or [len(poses)]):
        hv.IntegrateSequence(depth[k:k + n], rgb[k:k + n], poses[k:k + n])
        hv.Stats()   # launches what is queued: the next call starts a new batch
        k += n
    assert k == len(poses)
    return hv


def test_selection_forms_select_the_same_blocks(oracle):
    """Round 4: batches of >= 4 frames record their selections per super-block (k_select_vote) and one pass claims every block once
    (k_select_merge); frames whose range is too large for that claim directly, as every frame did before.  All forms -- and every mix of them
    inside one batch (a limit of n super-blocks splits the frames of a batch by the size of their range) -- must give the oracle's volume bit for bit:
    same keys, same voxels, same counters."""
    import torch
    dev = torch.device("cuda:0")
    depth, rgb, poses = S.room_sequence_torch(300, 37, dev)
    torch.cuda.synchronize()
    ov, _ = _mk(oracle, 0.01)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    sel = upd = 0
    for k in range(37):
        n, _vis, nu = ov.integrate(dn[k], cn[k], poses[k])
        sel += n; upd += nu
    for mode in ("auto", "direct", 1, 1500, 2200, 3000, 6000):
        hv = _fuse(oracle, depth, rgb, poses, mode)
        st = hv.Stats()
        assert st["frames"] == 37 and st["blocks_selected"] == sel and st["voxels_updated"] == upd, (mode, st)
        _compare(oracle, ov, hv)
    # short batches: with an explicit limit every batch of >= 2 frames records and merges; "auto" sends them through the direct form
    for mode in (1 << 18, 2200, "auto"):
        hv = _fuse(oracle, depth, rgb, poses, mode, chunks=[4, 5, 7, 1, 3, 6, 2, 9])
        assert hv.Stats()["blocks_selected"] == sel
        _compare(oracle, ov, hv)


def test_selection_with_frames_far_apart(oracle):
    """One batch whose frames look at places 60 m apart: the bounding range of their candidate ranges is mostly empty, so the merge step walks the
    frames' words instead of the bounding range (and a block is handled by the lowest frame that selected anything in its super-block).  Also: two
    frames of the batch see nothing at all."""
    import torch
    dev = torch.device("cuda:0")
    depth, rgb, poses = S.room_sequence_torch(40, 12, dev)
    poses = poses.copy()
    for k in range(12):
        poses[k, :3, 3] += np.float32(60.0 * (k % 3)) * np.array([1.0, -0.5, 0.25], np.float32)   # three sites
    depth[5] = 0.0
    depth[9] = 0.0
    torch.cuda.synchronize()
    ov, _ = _mk(oracle, 0.01)
    dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
    for k in range(12):
        ov.integrate(dn[k], cn[k], poses[k])
    for mode in ("auto", "direct", 2000):
        hv = _fuse(oracle, depth, rgb, poses, mode)
        _compare(oracle, ov, hv)


def test_released_pools_are_kept_for_reuse_up_to_the_configured_limit():
    """A destroyed volume's block pool (2.7 GB at the default capacity) stays with the library for the next volume -- the reference's
    drivers make a CubeHandler per submap and per Transform -- until op_release_cached_memory, or not at all with the limit at 0
    (OP_RUNTIME_OPT_CACHE_DEVICE_BYTES)."""
    import torch
    from onepiece_amd import _lib as L
    lib = L.load()
    pool = (1 << 18) * 10240
    L.check(lib.op_release_cached_memory())
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    hv = I.CubeHandler(); del hv
    import gc; gc.collect()
    kept = free0 - torch.cuda.mem_get_info(0)[0]
    assert kept >= pool                                   # the pool (and the small tables) are still allocated
    hv = I.CubeHandler(); del hv; gc.collect()
    assert free0 - torch.cuda.mem_get_info(0)[0] <= kept + (64 << 20)   # the second volume took the first one's buffers
    L.check(lib.op_release_cached_memory())
    assert free0 - torch.cuda.mem_get_info(0)[0] < pool // 2
    try:
        L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_CACHE_DEVICE_BYTES, 0))
        hv = I.CubeHandler(); del hv; gc.collect()
        assert free0 - torch.cuda.mem_get_info(0)[0] < pool // 2
    finally:
        L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_CACHE_DEVICE_BYTES, 32 << 30))
        L.check(lib.op_release_cached_memory())
