"""Dense RGB-D tracking.

One section of bench.py's JSON line (bench.py builds the context `c` -- the fused volume, the frames in HBM, the timed region's counters -- and calls run(c, out))."""
import json
import os
import sys
import time

import numpy as np


# ---- dense RGB-D tracking (Odometry::DenseTracking's coarse-to-fine loop); rank 0 reports
def run(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    import ctypes as C
    from onepiece_amd import odometry as OD, _lib as L
    lib = L.load()
    odo = OD.Odometry(hv.camera, device=local_rank)
    odo.SetSums("fp64")   # (a new tracker starts in the reference-order mode, OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS: the order-free fp64 reduction is measured first)
    # frame 1 -> frame 0 of this rank's shard.  (a) from the raw frames, end to end (op_tracker_dense_tracking);
    # (b) the loop alone on the pyramids (a) built, resident in HBM (boundary = MultiScaleComputing's inputs)
    full = lambda: odo.DenseTracking(rgb[1], rgb[0], depth[1], depth[0], None, 0, want_correspondences=False)
    for _ in range(3):
        full()
    n_full = 100
    t = time.perf_counter()
    for _ in range(n_full):
        fres = full()
    full_s = n_full / (time.perf_counter() - t)
    levels = odo.PreparedLevels()
    dev_levels = []
    for lv in levels:
        d = dict(lv)
        for k in OD.TRACK_IMAGES:
            d[k] = torch.from_numpy(np.ascontiguousarray(lv[k])).to(dev)
        dev_levels.append(d)
    arr, mem, _keep = OD._levels_arg(dev_levels)
    it3 = np.array(odo.iter_count_per_level, np.int32)
    T0 = np.eye(4, dtype=np.float32).reshape(16)
    tres = L.TrackResult()
    run = lambda: L.check(lib.op_tracker_track(odo._h, arr, 3, it3.ctypes.data_as(L._ip), W, H, 0, T0.ctypes.data_as(L._fp), mem,
                                               C.byref(tres), None, None, 0, None, None))
    for _ in range(5):
        run()
    n_tr = 100
    t = time.perf_counter()
    for _ in range(n_tr):
        run()
    tr_s = n_tr / (time.perf_counter() - t)
    odo.SetSums("reference_f32")       # the reference's sums: rows of every iteration summed sequentially in float32 in raster order, by one wave on the device
    for _ in range(2):
        run()
    t = time.perf_counter()
    for _ in range(20):
        run()
    tr_ref_s = 20 / (time.perf_counter() - t)
    odo.SetSums("reference_f32_host")  # the same sums on one host thread (all rows cross PCIe every iteration): the cross-check variant
    run()
    t = time.perf_counter()
    for _ in range(5):
        run()
    tr_ref_host_s = 5 / (time.perf_counter() - t)
    odo.SetSums("fp64")
    # headline = the default mode (reference-order float32 sums: every pair within 1e-4 of the CPU path); the fp64 reduction is the opt-in fast mode
    out["tracking"] = {"tracks_per_s": tr_ref_s, "mode": "OP_TRACK_SUMS_REFERENCE_F32 (default)", "fp64_mode_tracks_per_s": tr_s, "fp64_mode_ms_per_track": 1e3 / tr_s, "reference_order_tracks_per_s": tr_ref_s, "reference_order_host_sums_tracks_per_s": tr_ref_host_s,
                       "from_raw_frames_tracks_per_s": full_s, "levels": 3, "iters_per_level": [4, 8, 16],
                       "iterations_executed": int(tres.iterations), "term": "hybrid", "resolution": [W, H],
                       "correspondences": int(tres.n_correspondences), "tracking_success": bool(tres.tracking_success),
                       "input": "pyramids resident in HBM (boundary = Odometry::MultiScaleComputing inputs)"}
    if world == 1 and not args.no_cpu_baseline:
        O = c.oracle   # the CPU oracle, imported by bench.py for its cpu_baseline leg (the only place that does)
        O.dense_track(levels, (4, 8, 16), term=0)
        t = time.perf_counter()
        for _ in range(5):
            ref = O.dense_track(levels, (4, 8, 16), term=0)
        out["cpu_baseline"]["tracks_per_s"] = 5 / (time.perf_counter() - t)
        O.lib().orc_set_accumulate_double(1)
        ref_d = O.dense_track(levels, (4, 8, 16), term=0)
        O.lib().orc_set_accumulate_double(0)
        rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
        g = np.array(tres.T, np.float64).reshape(4, 4)
        out["tracking"]["parity"] = {"pose_rel_err_vs_cpu": rel(g, ref["T"]), "pose_rel_err_vs_cpu_double_sums": rel(g, ref_d["T"]),
                                     "cpu_float_vs_double_sums": rel(ref["T"], ref_d["T"]),
                                     "iterations": {"gpu": int(tres.iterations), "cpu": int(ref["iterations"])},
                                     "correspondences": {"gpu": int(tres.n_correspondences), "cpu": int(len(ref["pixel_correspondences"]))}}
    del odo
