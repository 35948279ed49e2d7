"""Long fusion fuzz (one-off confidence run, not part of the suites): many seeds of tests/test_integration_gpu.py's
random-frame fuzz, HIP path vs oracle bit for bit, in every form of the selection step (24 frames per seed = one batch that records and merges
in "auto"; a limit of n super-blocks splits the batch's frames between the recording and the direct form).  usage: fuzz_fusion_long.py [seeds=20]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import integration as I
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cam = (120.0, 118.0, 79.3, 60.7, 160, 120, 1000.0)
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(1000 + seed)
    hcam = I.PinholeCamera(); hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    res = float(rng.choice([0.02, 0.03, 0.04]))
    ov = oracle.Volume(oracle.make_camera(*cam), voxel_res=res)
    modes = ("auto", "direct", int(rng.integers(50, 4000)))
    hvs = []
    for m in modes:
        hv = I.CubeHandler(hcam, max_blocks=1 << 19); hv.SetVoxelResolution(res); hv.SetSelectMode(m)
        hvs.append(hv)
    for k in range(24):
        d = rng.uniform(0.05, 6.0, (120, 160)).astype(np.float32)
        d[rng.random((120, 160)) < 0.1] = 0.0
        d[rng.random((120, 160)) < 0.02] = -1.0
        d[rng.random((120, 160)) < 0.01] = np.nan
        d[rng.random((120, 160)) < 0.01] = 1e6
        if k % 2 == 0:
            u, v = np.meshgrid(np.arange(160), np.arange(120))
            d = (1.5 + 0.5 * np.sin(u / 17.0 + k + seed) * np.cos(v / 13.0)).astype(np.float32)
        c = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
        c[rng.random((120, 160)) < 0.2] = 0                         # black pixels: zero numerators in the colour update
        x = np.concatenate([rng.uniform(-0.3, 0.3, 3), rng.uniform(-1.5, 1.5, 3)]).astype(np.float32)
        pose = oracle.se3_exp(x)
        ov.integrate(d, c, pose)
        for hv in hvs:
            hv.IntegrateImage(d, c, pose)
    ok, ox = ov.export()
    same = True
    for hv in hvs:
        hk, hx = hv.GetCubeMap()
        same = same and np.array_equal(ok, hk) and np.array_equal(ox.view(np.uint32), hx.view(np.uint32))
    bad += not same
    print("seed %d res %.2f, selection %s: %d blocks %s" % (seed, res, modes, len(ok), "bit-equal" if same else "DIFFERENT"), flush=True)
print("%d of %d seeds differ" % (bad, n_seeds))
sys.exit(1 if bad else 0)
