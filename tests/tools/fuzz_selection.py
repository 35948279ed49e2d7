"""Selection fuzz (one-off confidence run, not part of the suites): PrepareCubes of the HIP path -- coarse 4x4x4 group test against the tile
depth ranges, then the reference's per-block test on the survivors -- against the oracle's plain per-block loop, list for list, over random
cameras, image sizes, voxel sizes, surfaces (smooth, stepped, noisy, with holes / NaN / negative / huge values, 16-bit) and rigid poses.
usage: fuzz_selection.py [cases=300]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import integration as I
oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = n_sel = n_cand = 0
for case in range(n_cases):
    rng = np.random.default_rng(77000 + case)
    w, h = int(rng.integers(1, 330)), int(rng.integers(1, 250))
    if case % 3 == 0: w, h = 320, 240
    f = rng.uniform(0.5, 1.6) * max(w, h)
    cam = (float(f), float(f * rng.uniform(0.9, 1.1)), float(w * rng.uniform(-0.2, 1.2)), float(h * rng.uniform(-0.2, 1.2)), w, h, 1000.0)
    res = float(rng.choice([0.004, 0.005, 0.008, 0.0125, 0.02]))
    hcam = I.PinholeCamera(); hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    ov = oracle.Volume(oracle.make_camera(*cam), voxel_res=res)
    hv = I.CubeHandler(hcam, max_blocks=1 << 16); hv.SetVoxelResolution(res)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d = rng.uniform(0.55, 3.0) + rng.uniform(0, 0.6) * np.sin(u / rng.uniform(3, 60) + rng.uniform(0, 6)) * np.cos(v / rng.uniform(3, 60))
    d = d + rng.uniform(-0.004, 0.004) * u + rng.uniform(-0.004, 0.004) * v
    if case % 4 == 1: d = np.where(((u // rng.integers(3, 40)) + (v // rng.integers(3, 40))) % 2 == 0, d, d + rng.uniform(0.2, 2.5))   # steps
    if case % 4 == 2: d = d + rng.normal(0, rng.uniform(0.001, 0.2), (h, w))                                                        # noise
    if case % 7 == 3: d = np.where((u + 2 * v) % 11 == 0, rng.uniform(0.01, 0.4), d)                                                # speckles close to the camera
    d = np.asarray(d, np.float32)
    d[rng.random((h, w)) < rng.uniform(0, 0.2)] = 0.0
    d[rng.random((h, w)) < 0.01] = np.nan
    d[rng.random((h, w)) < 0.01] = -1.0
    d[rng.random((h, w)) < 0.005] = 1e6
    x = np.concatenate([rng.uniform(-1.0, 1.0, 3), rng.uniform(-3.0, 3.0, 3)]).astype(np.float32)
    pose = oracle.se3_exp(x) if case % 5 else np.eye(4, dtype=np.float32)
    if case % 6 == 4:
        d = np.round(np.nan_to_num(np.clip(d, 0, 60.0)) * 1000.0).astype(np.uint16)
    oids, ocand = ov.prepare_cubes(d, pose)
    hids, hcand = hv.PrepareCubes(d, pose, return_candidates=True)
    same = ocand == hcand and np.array_equal(oids, hids)
    bad += not same
    n_sel += len(oids); n_cand += ocand
    if not same or case % 25 == 0:
        print("case %d %dx%d voxel %.4f: %d candidates, %d selected %s" % (case, w, h, res, ocand, len(oids), "equal" if same else "DIFFERENT (hip %d)" % len(hids)), flush=True)
print("%d of %d cases differ; %d candidates, %d selected in total" % (bad, n_cases, n_cand, n_sel))
sys.exit(1 if bad else 0)
