"""Wide fusion fuzz (one-off confidence run, not part of the suites): where fuzz_fusion_long.py varies the frames, this one varies the
CONFIGURATION -- image sizes that are not multiples of 4 / 16 / 64 (narrow loads, partial tiles), intrinsics, uint16 and float32 depth,
resolution, truncation, near / far planes, a starting pool of a few hundred blocks (growth + replay in the middle of a batch), frame counts
that end batches at every fill level -- changes the parameters of a volume in use (truncation, planes, camera incl. its image size, resolution) and interleaves the other writers of a volume between fusions (Merge of a second volume, upload through
SetCubeMap, Clear, AddCube), which switch the update between its plain and general forms.  HIP path vs oracle, keys and voxels bit for bit
after every phase.  With FUZZ_VIEWS=1 every phase and every operation is also followed by a raycast of the volume from the last frame's pose (depth bit for bit against
the restatement): views BETWEEN fusions prune by the block summaries k_integrate keeps current.  usage: fuzz_fusion_wide.py [seeds=20] [first_seed=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import integration as I
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def frame(rng, w, h, k, u16, scale):
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    if k % 3 == 0:
        d = rng.uniform(0.05, 6.0, (h, w))
    else:
        d = 1.2 + 0.6 * np.sin(u / (7.0 + k % 5) + k) * np.cos(v / (5.0 + k % 7)) + 0.3 * (k % 4)
    d = d.astype(np.float32)
    d[rng.random((h, w)) < 0.08] = 0.0
    if u16:
        d = np.clip(np.round(d * scale), 0, 65535).astype(np.uint16)
    else:
        d[rng.random((h, w)) < 0.01] = -1.0
        d[rng.random((h, w)) < 0.005] = np.nan
        d[rng.random((h, w)) < 0.005] = np.inf
    c = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    c[rng.random((h, w)) < 0.15] = 0
    c[rng.random((h, w)) < 0.05] = 255
    x = np.concatenate([rng.uniform(-0.25, 0.25, 3), rng.uniform(-1.0, 1.0, 3)]).astype(np.float32)
    return d, c, oracle.se3_exp(x)


def same(ov, hv, what, seed):
    ok, ox = ov.export()
    hk, hx = hv.GetCubeMap()
    good = np.array_equal(ok, hk) and np.array_equal(ox.view(np.uint32), hx.view(np.uint32))
    if not good:
        detail = "keys differ (%d vs %d blocks)" % (len(ok), len(hk)) if not np.array_equal(ok, hk) else "%d voxels differ" % int((ox.view(np.uint32) != hx.view(np.uint32)).any(axis=2).sum())
        print("seed %d: DIFFERENT after %s: %s" % (seed, what, detail), flush=True)
    return good


VIEWS = os.environ.get("FUZZ_VIEWS") == "1"
pruned_total = 0


def view(ov, hv, pose, hcam, ocam, what, seed):
    """depth images of the volume as it is now, HIP raycaster vs the CPU restatement"""
    global pruned_total
    hd, _hn, _hc = hv.Raycast(pose, hcam)
    od, _on, _oc = ov.raycast(pose, ocam)
    pruned_total += hv.RaycastStats()["dropped_unloaded"]
    good = np.array_equal(hd.view(np.uint32), od.view(np.uint32))
    if not good:
        print("seed %d: VIEW DIFFERENT after %s: %d pixels" % (seed, what, int((hd.view(np.uint32) != od.view(np.uint32)).sum())), flush=True)
    return good


bad = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(77000 + seed)
    w = int(rng.choice([37, 63, 64, 65, 101, 127, 130, 160, 161, 200]))
    h = int(rng.choice([29, 31, 48, 50, 97, 120]))
    f = float(rng.uniform(0.6, 1.3)) * w
    scale = float(rng.choice([1000.0, 5000.0]))
    cam = (f, f * float(rng.uniform(0.97, 1.03)), w / 2 + float(rng.uniform(-3, 3)), h / 2 + float(rng.uniform(-3, 3)), w, h, scale)
    res = float(rng.choice([0.015, 0.02, 0.03, 0.05]))
    trunc = float(rng.choice([0.06, 0.1, 0.2, 1.5]))   # 1.5: observations can be "invalid" voxels themselves (TSDFVoxel::IsValid)
    near, far = float(rng.choice([0.1, 0.5, 0.9])), float(rng.choice([2.5, 5.0, 8.0]))
    u16 = bool(rng.random() < 0.5)
    hcam = I.PinholeCamera(); hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    ocam = oracle.make_camera(*cam)
    pool = int(rng.choice([64, 300, 2000, 1 << 16]))
    ov = oracle.Volume(ocam, voxel_res=res, trunc=trunc, far=far, near=near)
    hv = I.CubeHandler(hcam, max_blocks=pool)
    hv.SetVoxelResolution(res); hv.SetTruncation(trunc); hv.SetNearPlane(near); hv.SetFarPlane(far)
    hv.SetSelectMode(rng.choice(["auto", "direct"]) if rng.random() < 0.7 else int(rng.integers(1, 500)))
    good = True
    k = 0
    log = []
    for phase in range(int(rng.integers(2, 5))):
        n = int(rng.choice([1, 2, 7, 19, 20, 31, 32, 33, 45]))
        for _ in range(n):
            d, c, pose = frame(rng, w, h, k, u16, scale); k += 1
            ov.integrate(d, c, pose)
            hv.IntegrateImage(d, c, pose)
        log.append("fuse %d" % n)
        good = good and same(ov, hv, " / ".join(log), seed)
        if VIEWS: good = view(ov, hv, pose, hcam, ocam, " / ".join(log), seed) and good
        op = rng.choice(["none", "merge", "upload", "clear", "addcube", "reconfigure"], p=[0.2, 0.25, 0.15, 0.1, 0.1, 0.2])
        if op == "reconfigure":  # the setters of a volume in use (CubeHandler.h:36-39,137-144,339-346): parameters change, the content stays
            what = rng.choice(["truncation", "planes", "camera", "resolution"])
            if what == "truncation": trunc = float(rng.choice([0.06, 0.1, 0.2, 1.5])); hv.SetTruncation(trunc)
            elif what == "planes": near, far = float(rng.choice([0.1, 0.5, 0.9])), float(rng.choice([2.5, 5.0, 8.0])); hv.SetNearPlane(near); hv.SetFarPlane(far)
            elif what == "resolution": res = float(rng.choice([0.015, 0.02, 0.03, 0.05])); hv.SetVoxelResolution(res)
            else:
                w = int(rng.choice([37, 63, 64, 65, 101, 127, 130, 160, 161, 200])); h = int(rng.choice([29, 31, 48, 50, 97, 120]))
                f = float(rng.uniform(0.6, 1.3)) * w
                cam = (f, f * float(rng.uniform(0.97, 1.03)), w / 2 + float(rng.uniform(-3, 3)), h / 2 + float(rng.uniform(-3, 3)), w, h, scale)
                hcam = I.PinholeCamera(); hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
                ocam = oracle.make_camera(*cam)
                hv.SetCamera(hcam)
            keys, vox = ov.export(sort=False)
            ov = oracle.Volume(ocam, voxel_res=res, trunc=trunc, far=far, near=near)
            ov.load(keys, vox)
            op = "set " + str(what)
        if op == "merge":
            ov2 = oracle.Volume(ocam, voxel_res=res, trunc=trunc, far=far, near=near)
            hv2 = I.CubeHandler(hcam, max_blocks=pool); hv2.SetVoxelResolution(res); hv2.SetTruncation(trunc); hv2.SetNearPlane(near); hv2.SetFarPlane(far)
            for _ in range(int(rng.integers(1, 6))):
                d, c, pose = frame(rng, w, h, k, u16, scale); k += 1
                ov2.integrate(d, c, pose); hv2.IntegrateImage(d, c, pose)
            ov.merge(ov2); hv.Merge(hv2)
        elif op == "upload":
            keys, vox = ov.export(sort=False)
            ov.clear(); ov.load(keys, vox)
            hk, hx = hv.GetCubeMap(sort=False)
            hv.SetCubeMap(hk, hx)
        elif op == "clear":
            ov.clear(); hv.Clear()
        elif op == "addcube" and ov.block_count():
            keys, _ = ov.export()
            cid = (keys[0] + np.array([40, 0, 0])).astype(np.int32)
            blank = np.zeros((1, 512, 5), np.float32); blank[..., 0] = 999.0; blank[..., 2:] = -1.0
            if not (keys == cid).all(axis=1).any(): ov.load(cid.reshape(1, 3), blank)   # CubeHandler::AddCube leaves a block that is there alone
            hv.AddCube(cid)
        log.append(op)
        if op != "none":
            good = good and same(ov, hv, " / ".join(log), seed)
            if VIEWS: good = view(ov, hv, pose, hcam, ocam, " / ".join(log), seed) and good
    bad += not good
    print("seed %d: %dx%d %s res %.3f trunc %.2f near %.1f far %.1f pool %d: %s -> %d blocks %s" % (seed, w, h, "u16" if u16 else "f32", res, trunc, near, far, pool, ", ".join(log), ov.block_count(),
                                                                                              "bit-equal" if good else "DIFFERENT"), flush=True)
print("%d of %d seeds differ" % (bad, n_seeds) + ("; views between the phases dropped %d blocks unloaded in total" % pruned_total if VIEWS else ""))
sys.exit(1 if bad else 0)
