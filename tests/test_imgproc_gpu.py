"""tool::ConvertDepthTo32F + tool::BilateralFilter on the GPU (op_bilateral_filter_depth) vs the oracle's restatement of the
same definition.  Floating point: the taps are summed in the same order; the kernel evaluates the two Gaussian factors as
one hardware base-2 exponential (weights within ~1e-6 relative of the oracle's expf product), which moves a weighted mean of
nearby depths by far less than the bar of 2e-6 relative to the image's largest depth (the reference's own OpenCV call is unpinned, see include/onepiece_hip.h)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from onepiece_amd import integration as I, synthetic as S, tool as T, _lib as L
from helpers import small_camera

TOL = 2e-6


def _noisy_room(i, scale=4, seed=0, holes=True):
    cam = small_camera(scale)
    d, c = S.room_render(S.room_pose(i), width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
    rng = np.random.default_rng(seed)
    d = (d + rng.normal(0, 0.004, d.shape)).astype(np.float32)
    if holes:
        d[rng.random(d.shape) < 0.03] = 0
        d[10:14, 20:40] = 0
    return d, c


def _close(a, b, ref):
    assert a.shape == b.shape and a.dtype == b.dtype == np.float32
    assert np.abs(a - b).max() <= TOL * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("scale", [4, 1])
def test_float_depth_matches_oracle(oracle, scale):
    d, _ = _noisy_room(30, scale)
    out = T.BilateralFilter(d)
    _close(out, oracle.bilateral_filter(d), d)
    # invalid pixels (0) see no valid neighbour within 3 sigma_color and stay invalid; valid ones stay in range
    assert np.all(out[d == 0] < 1e-6) and np.all(out[d > 0] > 0.4)
    assert np.abs(out - d)[d > 0].max() < 0.05 and np.abs(out - d)[d > 0].mean() > 1e-4     # it does smooth


def test_u16_input_folds_the_conversion(oracle):
    d, _ = _noisy_room(5)
    d16 = np.clip(np.round(d * 1000), 0, 65535).astype(np.uint16)
    out = T.BilateralFilter(d16, depth_scale=1000.0)
    _close(out, oracle.bilateral_filter(d16, depth_scale=1000.0), d)
    assert np.array_equal(out, T.BilateralFilter(T.ConvertDepthTo32F(d16, 1000.0)))          # the two calls of the drivers
    out5k = T.BilateralFilter(d16, depth_scale=5000.0)
    _close(out5k, oracle.bilateral_filter(d16, depth_scale=5000.0), d)


@pytest.mark.parametrize("rng_d", [3, 5, 9, 12, 0, 31])
def test_other_diameters(oracle, rng_d):
    d, _ = _noisy_room(12, seed=rng_d)
    _close(T.BilateralFilter(d, range=rng_d), oracle.bilateral_filter(d, d=rng_d), d)


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (3, 2), (5, 70), (17, 64), (33, 129)])
def test_small_and_ragged_sizes(oracle, shape):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    d = (2.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
    _close(T.BilateralFilter(d), oracle.bilateral_filter(d), d)
    _close(T.BilateralFilter(d, range=15), oracle.bilateral_filter(d, d=15), d)


def test_constant_and_empty(oracle):
    d = np.full((48, 64), 1.75, np.float32)
    assert np.abs(T.BilateralFilter(d) - d).max() <= 6e-7      # 29 rounded products and sums
    z = np.zeros((48, 64), np.float32)
    assert np.array_equal(T.BilateralFilter(z), z)
    assert T.BilateralFilter(np.zeros((0, 48, 64), np.float32)).shape == (0, 48, 64)


def test_device_batch_on_the_volume_stream_feeds_fusion(oracle):
    """[n,h,w] tensors in HBM, filtered on the CubeHandler's own stream and fused from there without a host sync: the
    volume equals the one fused from the downloaded filtered images, and the images equal the per-image host calls."""
    import torch
    cam = small_camera(4)
    hcam = I.PinholeCamera()
    hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    frames = [_noisy_room(i, holes=False, seed=i) for i in (0, 8, 16, 24)]
    poses = np.stack([S.room_pose(i) for i in (0, 8, 16, 24)])
    depth = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    rgb = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    torch.cuda.synchronize()
    hv = I.CubeHandler(hcam, max_blocks=1 << 16); hv.SetVoxelResolution(0.02)
    filt = T.BilateralFilter(depth, stream=hv.Stream())
    hv.IntegrateSequence(filt, rgb, poses)
    hv.Synchronize()
    f_host = filt.cpu().numpy()
    for k, (d, _c) in enumerate(frames):
        assert np.array_equal(f_host[k], T.BilateralFilter(d))
        _close(f_host[k], oracle.bilateral_filter(d), d)
    ov = oracle.Volume(oracle.make_camera(*cam), voxel_res=0.02)
    for k in range(len(frames)):
        ov.integrate(f_host[k], frames[k][1], poses[k])
    ok, ox = ov.export()
    hk, hx = hv.GetCubeMap()
    assert np.array_equal(ok, hk) and np.array_equal(ox.view(np.uint32), hx.view(np.uint32))
    # without a stream the call is synchronous and gives the same bits; `out` is honoured
    out = torch.empty_like(depth)
    assert T.BilateralFilter(depth, out=out) is out and torch.equal(out, filt)


def test_argument_errors():
    lib = L.load()
    d = np.zeros((8, 8), np.float32); o = np.zeros((8, 8), np.float32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    call = lambda *a: lib.op_bilateral_filter_depth(*a)
    assert call(None, 0, 1000.0, 8, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 0, None, vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 7, 1000.0, 8, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 0, None, vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 1, 0.0, 8, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 0, None, vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 0, 1000.0, 0, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 0, None, vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 0, 1000.0, 8, 8, 1, 33, 0.03, 4.5, L.OP_MEM_HOST, 0, None, vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 0, 1000.0, 8, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 0, C.c_void_p(1), vp(o)) == L.OP_ERR_INVALID
    assert call(vp(d), 0, 1000.0, 8, 8, 1, 7, 0.03, 4.5, L.OP_MEM_HOST, 99, None, vp(o)) == L.OP_ERR_INVALID
    with pytest.raises(ValueError):
        T.BilateralFilter(np.zeros((2, 2, 2, 2), np.float32))
