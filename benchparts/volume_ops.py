"""The volume operations next to fusion -- GetPointCloud, ExtractTriangleMesh, Transform / TransformNearest, EstimateNormals (G2),
BilateralFilter (N5) -- as KERNEL times with a byte model each: tools/ops_driver.bin (torch-free) under rocprofv3 --kernel-trace --stats on a dump of
this run's first frames.

One section of bench.py's JSON line (bench.py builds the context `c` and calls run(c, out))."""
import csv
import glob
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np


def _kernel_rows(d):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            rows[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
    return rows


def run(c, out):
    depth, rgb, poses, n_local, ROOT, W, H, HBM = c.depth, c.rgb, c.poses, c.n_local, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    driver = os.path.join(ROOT, "tools", "ops_driver.bin")
    if not os.path.exists(driver) or shutil.which("rocprofv3") is None:
        out["volume_ops"] = {"error": "tools/ops_driver.bin or rocprofv3 missing"}
        return
    nf = min(96, n_local)
    td = tempfile.mkdtemp(prefix="opv_", dir="/tmp")
    try:
        fname = os.path.join(td, "frames.bin")
        with open(fname, "wb") as tf:
            np.array([nf, W, H], np.int32).tofile(tf)
            dh, ch = depth[:nf].cpu().numpy(), rgb[:nf].cpu().numpy()
            for i in range(nf):
                poses[i].astype(np.float32).tofile(tf); dh[i].tofile(tf); ch[i].tofile(tf)
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(td, "prof"), "-o", "p", "--",
               driver, fname, repr(float(c.args.voxel)), "3", "pointcloud", "mesh", "transform", "transform_nn", "normals", "bilateral"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        txt = p.stdout
        k = _kernel_rows(os.path.join(td, "prof"))
    except Exception as e:  # a failing profiler must not take the bench line down
        out["volume_ops"] = {"error": repr(e)[:300]}
        return
    finally:
        shutil.rmtree(td, ignore_errors=True)
    num = lambda pat, g=1: (float(re.search(pat, txt).group(g)) if re.search(pat, txt) else None)
    blocks = num(r"fused \d+ frames .*?: (\d+) blocks")
    points = num(r"pointcloud: (\d+) points")
    tris = num(r"mesh: (\d+) triangles")
    t_out = num(r"transform: \d+ -> (\d+) blocks")
    tn_out = num(r"transform_nn: \d+ -> (\d+) blocks")
    n_pts = num(r"normals: (\d+) points")

    def entry(kernel, bytes_per_launch, model, note=None):
        r = k.get(kernel)
        if not r or not bytes_per_launch:
            return None
        gbs = bytes_per_launch / (r["avg_us"] * 1e-6) / 1e9
        e = {"kernel": kernel, "launches": r["calls"], "avg_us": r["avg_us"], "bytes_per_launch": bytes_per_launch, "gbs": gbs, "frac_of_hbm_peak": gbs / HBM, "byte_model": model}
        if note:
            e["note"] = note
        return e

    res = {"frames": nf, "blocks": blocks, "source": "rocprofv3 --kernel-trace --stats of tools/ops_driver.bin on the first %d frames of this run (5 mm volume); "
                                                     "the same run on the 164 k-block room volume with FETCH_SIZE / WRITE_SIZE / TCC passes: profiles/r05_volume_ops.*" % nf,
           "driver_wall_clock": [l for l in txt.splitlines() if l and not l.startswith("fused")]}
    if blocks:
        res["point_cloud"] = entry("k_point_cloud", 4096.0 * blocks + 12.0 * (points or 0), "per pass (count, then emit): the sdf + weight planes of every block (4 KB) [+ 24 B per point written by the emit pass, halved over the two passes]")
        res["mesh"] = entry("k_mesh", 4096.0 * blocks + (36.0 + 36.0) * (tris or 0), "per pass (count, then emit): the sdf + weight planes of every block (4 KB), staged once in LDS "
                            "[+ per triangle, in the emit pass: 72 B of colour gathers and 72 B written, halved over the two passes]",
                            "the +x/+y/+z layers of the 7 neighbour blocks are re-reads that mostly hit in L2; 3.3 ms per pass before round 5 (every voxel loaded 8 corners x 5 planes)")
        if t_out:
            res["transform"] = entry("k_transform_fill_wave", 10240.0 * (t_out + blocks), "10 KB written per result block + every source block read at least once",
                                     "eight trilinear taps x five planes per voxel: a gather kernel (40 scattered 4-byte loads per voxel), not a streaming one; round 5: the source "
                                     "blocks of a result block are looked up once per block, and one wave fills a block (no barriers, more blocks in flight): 4.8 -> 2.6 ms on the "
                                     "164 k-block volume; what was measured and did not move it (LDS staging, 8-byte tap pairs, Morton order): DESIGN.md section 8")
            res["transform_alloc"] = entry("k_transform_alloc<false>", 512.0 * 4.0 * blocks, "nominal: one 4-byte key component per source voxel -- the kernel is hash-table work (claims), not traffic",
                                           "the result blocks a source block's voxels land in are gathered in an LDS set and claimed once per workgroup since round 5 (3.4 -> 1.5 ms)")
        if tn_out:
            res["transform_nearest"] = entry("k_transform_fill<true>", 10240.0 * (tn_out + blocks), "10 KB written per result block + every source block read at least once")
    if n_pts:
        res["estimate_normals"] = entry("k_estimate_normals", 24.0 * n_pts, "12 B read + 12 B written per point",
                                        "exact 30-nearest-neighbour search over the cell grid + PCA per point: bound by the search's arithmetic, the byte model only says how little memory it must move")
    bil = [v for kk, v in k.items() if kk.startswith("k_bilateral")]
    if bil:
        kk = [n for n in k if n.startswith("k_bilateral")][0]
        res["bilateral_filter"] = entry(kk, 8.0 * W * H * nf, "4 B read + 4 B written per pixel, %d images per launch" % nf, "37 taps with two exponentials each per pixel: arithmetic-bound")
    out["volume_ops"] = res
