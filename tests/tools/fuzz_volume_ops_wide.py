"""Wide fuzz of the volume operations (one-off confidence run, not part of the suites): volumes fused from random frames under random
configurations (image size, intrinsics, resolution, truncation, near / far, uint16 / float depth; some pass through Merge / SetCubeMap first,
so that they are no longer "plain"), then Transform / TransformNearest by random rigid motions (small, large, 90-degree turns, pure
translations by whole blocks), GetPointCloud, ExtractTriangleMesh with the default tables, the .map round trip in both directions and
raycasts from random poses with random cameras -- HIP path vs oracle: volumes, clouds and triangle soups bit for bit, raycast depth bit for
bit.  usage: fuzz_volume_ops_wide.py [seeds=30] [first_seed=0]"""
import os, sys, tempfile, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import integration as I
from helpers import triangle_soup
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
host = C.CDLL(os.path.join(ROOT, "host", "one_piece", "libone_piece_hip_host.so"))
TRI = np.zeros((256, 16), np.int32); EDGES = np.zeros((12, 2), np.int32)
host.op_host_generate_mc_tables(TRI.ctypes.data_as(C.POINTER(C.c_int)), EDGES.ctypes.data_as(C.POINTER(C.c_int)))


def frame(rng, w, h, k, u16, scale):
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    d = (1.3 + 0.5 * np.sin(u / (9.0 + k % 5) + k) * np.cos(v / (6.0 + k % 7)) + 0.2 * (k % 3)).astype(np.float32)
    d[rng.random((h, w)) < 0.05] = 0.0
    if u16:
        d = np.clip(np.round(d * scale), 0, 65535).astype(np.uint16)
    c = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    x = np.concatenate([rng.uniform(-0.15, 0.15, 3), rng.uniform(-0.4, 0.4, 3)]).astype(np.float32)
    return d, c, oracle.se3_exp(x)


def volumes_equal(ov, hv):
    ok, ox = ov.export(); hk, hx = hv.GetCubeMap()
    return np.array_equal(ok, hk) and np.array_equal(ox.view(np.uint32), hx.view(np.uint32))


def canon(p, c):
    a = np.concatenate([p, c], axis=1)
    return a[np.lexsort(a.T[::-1])]


bad = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(55000 + seed)
    w, h = int(rng.choice([48, 63, 80, 101, 128])), int(rng.choice([36, 47, 60, 75]))
    f = float(rng.uniform(0.7, 1.2)) * w
    scale = 1000.0
    cam = (f, f, w / 2 - 0.4, h / 2 + 0.3, w, h, scale)
    res = float(rng.choice([0.01, 0.02, 0.03]))
    trunc = float(rng.choice([0.05, 0.1, 0.15]))
    near, far = float(rng.choice([0.3, 0.5])), float(rng.choice([3.0, 5.0]))
    u16 = bool(rng.random() < 0.4)
    hcam = I.PinholeCamera(); hcam.fx, hcam.fy, hcam.cx, hcam.cy, hcam.width, hcam.height, hcam.depth_scale = cam
    ocam = oracle.make_camera(*cam)
    mk_o = lambda: oracle.Volume(ocam, voxel_res=res, trunc=trunc, far=far, near=near)
    def mk_h():
        v = I.CubeHandler(hcam, max_blocks=int(rng.choice([256, 1 << 15]))); v.SetVoxelResolution(res); v.SetTruncation(trunc); v.SetNearPlane(near); v.SetFarPlane(far)
        return v
    ov, hv = mk_o(), mk_h()
    k = 0
    for _ in range(int(rng.integers(2, 9))):
        d, c, pose = frame(rng, w, h, k, u16, scale); k += 1
        ov.integrate(d, c, pose); hv.IntegrateImage(d, c, pose)
    how = "plain"
    if rng.random() < 0.4:  # a second volume merged in: the volume is no longer "plain" (general update, weight plane read by the raycaster)
        ov2, hv2 = mk_o(), mk_h()
        for _ in range(int(rng.integers(1, 4))):
            d, c, pose = frame(rng, w, h, k, u16, scale); k += 1
            ov2.integrate(d, c, pose); hv2.IntegrateImage(d, c, pose)
        ov.merge(ov2); hv.Merge(hv2); how = "merged"
    problems = []
    if not volumes_equal(ov, hv): problems.append("fusion")
    # -- resampling
    kind = str(rng.choice(["small", "large", "quarter_turn", "block_shift"]))
    if kind == "small": T = oracle.se3_exp((0.05 * rng.standard_normal(6)).astype(np.float32))
    elif kind == "large": T = oracle.se3_exp(np.concatenate([rng.uniform(-1.5, 1.5, 3), rng.uniform(-2, 2, 3)]).astype(np.float32))
    elif kind == "quarter_turn":
        T = np.eye(4, dtype=np.float32); T[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32); T[:3, 3] = rng.uniform(-1, 1, 3)
    else:
        T = np.eye(4, dtype=np.float32); T[:3, 3] = 8 * res * rng.integers(-5, 6, 3)
    T = np.ascontiguousarray(T, np.float32)
    for nearest in (True, False):
        ot = ov.transform(T, nearest=nearest)
        ht = hv.TransformNearest(T, max_blocks=64) if nearest else hv.Transform(T, max_blocks=64)
        if not volumes_equal(ot, ht): problems.append("transform(%s, %s)" % (kind, "nearest" if nearest else "trilinear"))
    # -- point cloud and mesh
    op, oc = ov.point_cloud(); hp, hc = hv.GetPointCloud()
    if op.shape != hp.shape or not np.array_equal(canon(op, oc).view(np.uint32), canon(hp, hc).view(np.uint32)): problems.append("point cloud")
    om, omc = ov.extract_mesh(TRI, EDGES); hm, hmc = hv.ExtractTriangleMesh(TRI, EDGES)
    if om.shape != hm.shape or not np.array_equal(triangle_soup(om, omc).view(np.uint32), triangle_soup(hm, hmc).view(np.uint32)): problems.append("mesh")
    # -- files
    with tempfile.TemporaryDirectory() as tmp:
        hv.WriteToFile(os.path.join(tmp, "h.map")); ov.write_file(os.path.join(tmp, "o.map"))
        o2 = mk_o(); h2 = mk_h()
        o2.read_file(os.path.join(tmp, "h.map")); h2.ReadFromFile(os.path.join(tmp, "o.map"))
        if not volumes_equal(o2, h2): problems.append("files")
    # -- raycasts
    for view in range(3):
        rw, rh = int(rng.choice([33, 64, 97])), int(rng.choice([25, 48, 61]))
        rf = float(rng.uniform(0.6, 1.4)) * rw
        rc = (rf, rf * 1.02, rw / 2 + float(rng.uniform(-4, 4)), rh / 2 + float(rng.uniform(-4, 4)), rw, rh, 1000.0)
        rcam = I.PinholeCamera(); rcam.fx, rcam.fy, rcam.cx, rcam.cy, rcam.width, rcam.height, rcam.depth_scale = rc
        pose = oracle.se3_exp(np.concatenate([rng.uniform(-0.4, 0.4, 3), rng.uniform(-0.8, 0.8, 3)]).astype(np.float32))
        hd, hn, hcol = hv.Raycast(pose, rcam); od, on, ocol = ov.raycast(pose, oracle.make_camera(*rc))
        if not np.array_equal(hd.view(np.uint32), od.view(np.uint32)): problems.append("raycast depth (view %d: %d pixels)" % (view, int((hd != od).sum())))
        else:
            hit = od > 0
            if hit.any() and (np.abs(hn - on)[hit].max() > 1e-3 or np.abs(hcol - ocol)[hit].max() > 1e-5): problems.append("raycast normals / colours (view %d)" % view)
            if hn[~hit].any() or hcol[~hit].any(): problems.append("raycast: values at pixels without a hit")
    bad += bool(problems)
    print("seed %d: %dx%d %s res %.2f trunc %.2f %s, %d blocks, %s, %d points, %d triangles -> %s" % (seed, w, h, "u16" if u16 else "f32", res, trunc, how, ov.block_count(), kind,
          len(op), len(om) // 3, "equal" if not problems else "DIFFERENT: " + "; ".join(problems)), flush=True)
print("%d of %d seeds differ" % (bad, n_seeds))
sys.exit(1 if bad else 0)
