"""Wide fuzz of the dense RGB-D tracker (one-off confidence run, not part of the suites): Odometry::DenseTracking end to end (image preparation,
pyramids, the coarse-to-fine loop) on rendered room pairs under random configurations -- image sizes (odd, not multiples of 16 or 64), 1-4 pyramid
levels, iteration counts, the three residual terms, uint16 / float depth with holes, frame distances from adjacent to far apart (lost tracks), an
initial guess -- with the reference-order float32 sums, against the oracle: iteration count, success flag, pixel correspondences identical, pose to
1e-6.  usage: fuzz_tracking_wide.py [seeds=40] [first_seed=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import integration as I, odometry as O, synthetic as S
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
worst = 0.0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(33000 + seed)
    w = int(rng.choice([64, 81, 96, 127, 160, 161, 200, 320])); h = int(rng.choice([48, 61, 75, 96, 120, 121, 240]))
    sc = 640.0 / w
    fx, fy, cx, cy = S.FX / sc, S.FY / sc, S.CX / sc, S.CY * (h / 480.0)
    levels = int(rng.choice([1, 2, 3, 4]))
    while (w >> (levels - 1)) < 8 or (h >> (levels - 1)) < 8: levels -= 1
    iters = [int(rng.choice([1, 2, 4, 8, 16])) for _ in range(levels)]
    term = int(rng.choice([0, 1, 2]))
    u16 = bool(rng.random() < 0.5)
    a = int(rng.integers(0, 900)); b = a + int(rng.choice([1, 1, 2, 5, 20, 120]))
    frames = []
    for k in (a, b):
        d, c = S.room_render(S.room_pose(k), width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy)
        d = d.copy()
        for _ in range(int(rng.integers(0, 4))):  # holes
            y0, x0 = int(rng.integers(0, h - 4)), int(rng.integers(0, w - 4))
            d[y0:y0 + int(rng.integers(2, h // 3 + 3)), x0:x0 + int(rng.integers(2, w // 3 + 3))] = 0.0
        if u16: d = np.clip(np.round(d * 1000.0), 0, 65535).astype(np.uint16)
        frames.append((d, c))
    T0 = oracle.se3_exp((0.01 * rng.standard_normal(6)).astype(np.float32)) if rng.random() < 0.4 else None
    cam = I.PinholeCamera("OPEN3D_DATASET"); cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height = fx, fy, cx, cy, w, h
    odo = O.Odometry(cam); odo.SetSums("reference_f32"); odo.SetMultiScale(levels); odo.iter_count_per_level = list(iters)
    ocam = oracle.make_camera(fx, fy, cx, cy, w, h)
    ref = oracle.dense_tracking(ocam, frames[1][1], frames[0][1], frames[1][0], frames[0][0], tuple(iters), term, init_T=T0)
    tag = "seed %d: %dx%d %s levels %d iters %s term %d frames %d/%d%s" % (seed, w, h, "u16" if u16 else "f32", levels, iters, term, a, b, " T0" if T0 is not None else "")
    try:
        got = odo.DenseTracking(frames[1][1], frames[0][1], frames[1][0], frames[0][0], T0, term)
    except Exception as e:  # noqa
        print("%s: hip refuses (%s); oracle iterations %d -> DIFFERENT" % (tag, e, ref["iterations"]), flush=True); bad += 1; continue
    fin = np.isfinite(ref["T"]).all() and np.isfinite(got.T).all()
    rel = float(np.linalg.norm(got.T - ref["T"]) / max(np.linalg.norm(ref["T"]), 1e-30)) if fin else (0.0 if np.array_equal(np.isnan(got.T), np.isnan(ref["T"])) else float("inf"))
    ok = got.iterations == ref["iterations"] and got.tracking_success == ref["tracking_success"] and np.array_equal(got.pixel_correspondence_set, ref["pixel_correspondences"]) and rel <= 1e-6
    worst = max(worst, rel)
    print("%s: iterations %d, success %s, %d pairs -> %s" % (tag, ref["iterations"], ref["tracking_success"], len(ref["pixel_correspondences"]),
          "identical (pose rel %.1e)" % rel if ok else "DIFFERENT (iterations %d/%d success %s/%s pairs %s pose rel %.2e)" % (got.iterations, ref["iterations"], got.tracking_success, ref["tracking_success"],
          np.array_equal(got.pixel_correspondence_set, ref["pixel_correspondences"]), rel)), flush=True)
    bad += not ok
print("%d of %d seeds differ; worst pose rel %.2e" % (bad, n_seeds, worst))
sys.exit(1 if bad else 0)
