"""BASELINE.json's configurations at their FULL sizes, in the suite the driver runs.

configs[2]: the whole 1000-frame 640x480 / 5 mm sequence through the HIP path and through the CPU oracle, every block key
and every voxel compared bit for bit.  The oracle's three fusion loops run on up to 32 host threads for this (their
units are independent -- a pixel, a candidate block, an exclusively owned selected block -- so its results do not
depend on the thread count: tests/test_oracle_golden.py::test_fusion_threads_do_not_change_results).
configs[3]: a 60-frame tracked pose chain against the oracle's chain, pose by pose, and the 2000-frame tracking + fusion
run with size-independent property checks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from onepiece_amd import integration as I, synthetic as S
from helpers import rel_err


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch.device("cuda:0")


def test_config3_1000_frames_bit_equal(oracle, torch_dev):
    """example/ImageSequenceIntegration at BASELINE's size: 1000 frames, 640x480, 5 mm voxels, every frame fused."""
    import torch
    n = 1000
    depth, rgb, poses = S.room_sequence_torch(0, n, torch_dev)
    torch.cuda.synchronize()
    hv = I.CubeHandler()
    hv.SetVoxelResolution(0.005)
    hv.IntegrateSequence(depth, rgb, poses)
    hv.Synchronize()
    st = hv.Stats()
    hk, hx = hv.GetCubeMap()
    del hv
    oracle.set_fusion_threads(None)
    try:
        ov = oracle.Volume(voxel_res=0.005)
        upd = vis = 0
        chunk = 100
        for s in range(0, n, chunk):                      # bounded host copies of the frames
            dn, cn = depth[s:s + chunk].cpu().numpy(), rgb[s:s + chunk].cpu().numpy()
            for k in range(dn.shape[0]):
                r = ov.integrate(dn[k], cn[k], poses[s + k])
                vis += r[1]; upd += r[2]
    finally:
        oracle.set_fusion_threads(1)
    ok, ox = ov.export()
    assert len(ok) == len(hk) > 150_000
    assert np.array_equal(hk, ok), "block key sets differ"
    assert st["voxels_updated"] == upd and st["voxels_visited"] == vis and st["frames"] == n
    assert np.array_equal(hx.view(np.uint32), ox.view(np.uint32)), "voxel payload differs"


def _chains(oracle, depth, rgb, poses, sums="fp64"):
    """GPU DenseSlam chain and the oracle's chain over the same frames (DenseSlam.cpp:21-33)."""
    from onepiece_amd import dense_slam as DS
    n = depth.shape[0]
    slam = DS.DenseSlam(I.PinholeCamera("OPEN3D_DATASET"))
    slam.rgbd_odometry.SetSums(sums)
    for i in range(n):
        assert slam.UpdateFrame(rgb[i], depth[i])
    hd, hc = depth.cpu().numpy(), rgb.cpu().numpy()
    ocam = oracle.make_camera()
    ref = [np.eye(4, dtype=np.float32)]
    pair_err = []
    for i in range(1, n):
        r = oracle.dense_tracking(ocam, hc[i - 1], hc[i], hd[i - 1], hd[i], (4, 8, 16), 0)
        assert r["tracking_success"]
        ref.append(DS._mat4_mul_f32(ref[-1], oracle.mat4_inverse(r["T"])))
        gp = np.linalg.inv(np.asarray(slam.global_poses[i - 1], np.float64)) @ np.asarray(slam.global_poses[i], np.float64)
        pair_err.append(rel_err(gp, np.linalg.inv(r["T"].astype(np.float64))))
    g0 = np.linalg.inv(poses[0].astype(np.float64))
    drift = lambda chain: np.array([np.abs(np.asarray(chain[i], np.float64) - g0 @ poses[i].astype(np.float64))[:3, 3].max() for i in range(n)])
    return slam, ref, np.array(pair_err), drift(slam.global_poses), drift(ref)


def test_config4_tracked_pose_chain_60_frames(oracle, torch_dev):
    """example/DenseFusion's tracking over 60 consecutive 640x480 frames in the DEFAULT summation mode (fp64 reduction on the device, no host
    round trip): every pose of the GPU chain against the CPU oracle's chain.  This mode does NOT meet north_star's 1e-4 pose bar on every
    pair, and the bars below are not that bar: they are regression bars set at what was measured (tests/tools/measure_chain.py, DESIGN.md
    section 7) -- per pair <= 5e-4 (measured worst 4.5e-4), <= 1e-4 on at least 90 % of the pairs (measured 55 of 59), chained pose <= 3e-3.
    The reference sums JTJ/JTr sequentially in float32 and its own result moves by up to 2e-4 per pair when it sums in double instead;
    the mode that DOES meet the bar on every pair is OP_TRACK_SUMS_REFERENCE_F32 (next test: 0.0).  The drift of the GPU chain from the
    ground-truth trajectory stays within 5 mm of the drift of the oracle's own chain at every frame (both are printed)."""
    import torch
    n = 60
    depth, rgb, poses = S.room_sequence_torch(0, n, torch_dev)
    torch.cuda.synchronize()
    slam, ref, pair_err, dg, dr = _chains(oracle, depth, rgb, poses)
    chain_err = np.array([rel_err(slam.global_poses[i], ref[i]) for i in range(n)])
    print("\nconfig 4 pose chain, %d frames: per-pair rel err max %.2e median %.2e (<= 1e-4 on %d/%d); chain rel err max %.2e;"
          " drift vs ground truth: gpu max %.4f m final %.4f m | oracle max %.4f m final %.4f m"
          % (n, pair_err.max(), np.median(pair_err), (pair_err <= 1e-4).sum(), len(pair_err), chain_err.max(), dg.max(), dg[-1], dr.max(), dr[-1]))
    assert all(slam.tracking_success) and slam.last_tracking_frame_id == n - 1
    # north_star's bar for this row is 1e-4 relative on the pose.  This mode is documented NOT to meet it on every pair; the assertion below states
    # that in executable form (should it ever hold, the documentation and bench.py's choice of config 4's mode are out of date and must be revisited)
    meets_north_star_bar = bool(pair_err.max() <= 1e-4)
    assert not meets_north_star_bar, "the fp64 mode now meets 1e-4 on every pair: update DESIGN.md section 7 and benchparts/dense_fusion.py"
    assert pair_err.max() <= 5e-4
    assert (pair_err <= 1e-4).mean() >= 0.90
    assert chain_err.max() <= 3e-3
    assert np.abs(dg - dr).max() <= 0.005


def test_config4_tracked_pose_chain_60_frames_reference_order_sums(oracle, torch_dev):
    """The same 60-frame chain with OP_TRACK_SUMS_REFERENCE_F32 -- every iteration's rows and NormalizeIntensity's means summed sequentially
    in float32 in raster order like the reference, on the device (k_seq_sums): every pair and every chained pose agrees with the CPU path to
    1e-5 -- north_star's 1e-4 with a decade to spare, on all 59 pairs (measured: 0.0).  What the default mode differs by (previous test) is
    the order of summation and nothing else.  50-140 tracks/s depending on the pair (a sequential float32 sum costs 8.25 shader cycles per
    term on this chip; 2-4 million terms per track)."""
    import torch
    n = 60
    depth, rgb, poses = S.room_sequence_torch(0, n, torch_dev)
    torch.cuda.synchronize()
    slam, ref, pair_err, dg, dr = _chains(oracle, depth, rgb, poses, sums="reference_f32")
    chain_err = np.array([rel_err(slam.global_poses[i], ref[i]) for i in range(n)])
    print("\nconfig 4 pose chain, reference-order sums: per-pair rel err max %.2e, chain rel err max %.2e" % (pair_err.max(), chain_err.max()))
    assert all(slam.tracking_success)
    assert pair_err.max() <= 1e-5 and chain_err.max() <= 1e-5
    assert np.abs(dg - dr).max() <= 1e-5


def test_config4_2000_frames_tracking_and_fusion_properties(torch_dev):
    """example/DenseFusion at BASELINE's size: 2000 synthetic 640x480 frames tracked frame to frame (4 pairs in flight) and
    fused at 5 mm with the TRACKED poses.  No oracle at this size (the CPU path needs ~10 minutes); size-independent
    properties instead: every frame tracks, every chained pose is a rigid transform, the odometry drift against the
    ground-truth trajectory stays bounded, the pipelined result equals the sequential loop on a prefix, and the fused
    volume is a consistent TSDF (weights are integers in [0, frames], observed sdf inside the truncation band, the
    update counter equals the sum of the weights)."""
    import torch
    from onepiece_amd import dense_slam as DS
    n, chunk = 2000, 250
    vol = I.CubeHandler()                                  # default pool; grows as the trajectory covers the room
    vol.SetVoxelResolution(0.005)
    fused = []
    slam = DS.DenseSlam(I.PinholeCamera("OPEN3D_DATASET"), pipeline=4,
                        on_tracked=lambda fid, c, d, T: (fused.append(fid), vol.IntegrateImage(d, c, T)))
    gt = np.empty((n, 4, 4), np.float32)
    keep = []
    for s in range(0, n, chunk):
        depth, rgb, poses = S.room_sequence_torch(s, chunk, torch_dev)
        gt[s:s + chunk] = poses
        keep.append((depth, rgb))                          # frames are used in place until resolved
        for k in range(chunk):
            slam.UpdateFrame(rgb[k], depth[k])
        if len(keep) > 2:
            slam.Finish(); vol.Synchronize(); keep.pop(0)
    slam.Finish()
    st = vol.Stats()
    assert fused == list(range(n)) and all(slam.tracking_success)
    P = np.asarray(slam.global_poses, np.float64)
    Rm = P[:, :3, :3]
    assert np.abs(Rm @ Rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4 and np.all(np.linalg.det(Rm) > 0.999)
    assert np.array_equal(P[:, 3], np.tile([0, 0, 0, 1.0], (n, 1)))
    g0 = np.linalg.inv(gt[0].astype(np.float64))
    drift = np.abs(P - g0 @ gt.astype(np.float64))[:, :3, 3].max(1)
    print("\nconfig 4, %d frames tracked + fused: drift vs ground truth max %.3f m, final %.3f m; %d blocks" % (n, drift.max(), drift[-1], vol.BlockCount()))
    # pure frame-to-frame odometry over two orbits of the room, no loop closure (measured: 0.77 m after 2000 frames, i.e.
    # 0.4 mm per frame): the drift stays of that order and accumulates smoothly -- no frame jumps -- within an orbit (the
    # synthetic trajectory itself steps by 5 cm where one orbit ends and the next, wider one begins: S.LOOP)
    step = np.abs(np.diff(drift))
    step[S.LOOP - 1::S.LOOP] = 0
    assert drift.max() < 1.5 and step.max() < 0.01
    # the pipelined chain equals the sequential loop (first 40 frames)
    depth, rgb, _ = S.room_sequence_torch(0, 40, torch_dev)
    seq = DS.DenseSlam(I.PinholeCamera("OPEN3D_DATASET"))
    for k in range(40):
        seq.UpdateFrame(rgb[k], depth[k])
    assert np.array_equal(np.asarray(seq.global_poses), np.asarray(slam.global_poses[:40]))
    # the volume
    assert st["frames"] == n
    hk, hx = vol.GetCubeMap()
    w, sdf = hx[..., 1], hx[..., 0]
    obs = w > 0
    assert np.array_equal(w, np.round(w)) and w.min() >= 0 and w.max() <= n
    assert int(w.astype(np.float64).sum()) == st["voxels_updated"]
    assert np.all(np.abs(sdf[obs]) < 0.1 + 1e-6) and np.all(sdf[~obs] == 999.0)
    assert np.all((hx[..., 2:][obs] >= 0) & (hx[..., 2:][obs] <= 1))
