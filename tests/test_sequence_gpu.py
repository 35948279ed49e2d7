"""The reference's sequence format end to end: synthetic frames -> associate.txt +
trajectory.txt + 16-bit depth PNGs -> reader -> fusion, HIP vs oracle on exactly the decoded data."""
import numpy as np
import pytest

from onepiece_amd import sequence as Q, synthetic as S


def _write(tmp_path, n=6):
    frames = [S.room_frame(20 * i) for i in range(n)]
    Q.WriteImageSequence(str(tmp_path), [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], 1000.0)
    return frames


def test_sequence_files_round_trip(tmp_path):
    frames = _write(tmp_path, 3)
    rgb_files, depth_files, poses = Q.ReadImageSequenceWithPose(str(tmp_path))
    assert len(rgb_files) == len(depth_files) == len(poses) == 3
    for i, (d, c, p) in enumerate(frames):
        assert np.array_equal(poses[i], p)                                   # "%.9g" round-trips float32
        assert np.array_equal(Q.imread(rgb_files[i]), c)                     # B,G,R as cv::imread would give
        d16 = Q.imread(depth_files[i], unchanged=True)
        assert d16.dtype == np.uint16 and np.abs(d16.astype(np.float64) / 1000.0 - d).max() <= 0.0005 + 1e-9
        f32 = Q.ConvertDepthTo32F(d16, 1000.0)
        assert f32.dtype == np.float32 and np.array_equal(f32, d16.astype(np.float32) / np.float32(1000.0))
    line = open(tmp_path / "associate.txt").readline().split()
    assert len(line) == 4 and line[1].startswith("rgb/") and line[3].startswith("depth/")


@pytest.mark.gpu
def test_fusing_the_decoded_sequence_matches_oracle(oracle, tmp_path):
    from onepiece_amd import integration as I
    _write(tmp_path, 6)
    rgb_files, depth_files, poses = Q.ReadImageSequenceWithPose(str(tmp_path))
    hv16 = I.CubeHandler(); hv16.SetVoxelResolution(0.00625)
    hv32 = I.CubeHandler(); hv32.SetVoxelResolution(0.00625)
    ov = oracle.Volume(voxel_res=0.00625)
    for i in range(len(poses)):
        rgb = Q.imread(rgb_files[i]); d16 = Q.imread(depth_files[i], unchanged=True)
        hv16.IntegrateImage(d16, rgb, poses[i])                              # CV_16UC1 path (Integrator.cpp:29)
        hv32.IntegrateImage(Q.ConvertDepthTo32F(d16, hv32.camera.depth_scale), rgb, poses[i])   # the examples' path
        ov.integrate(d16, rgb, poses[i])
    ok, ox = ov.export()
    for hv in (hv16, hv32):
        hk, hx = hv.GetCubeMap()
        assert np.array_equal(ok, hk) and np.array_equal(ox.view(np.uint32), hx.view(np.uint32))
