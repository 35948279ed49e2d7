"""Dense-tracker timing: tracks/s at 640x480, 3 levels {4,8,16}, hybrid term, pyramids resident in HBM.
Also used under rocprofv3 --kernel-trace --stats (tools/profile_track.sh)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import torch
from onepiece_amd import odometry as O, integration as I, _lib as L
from onepiece_amd import synthetic as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sums = sys.argv[2] if len(sys.argv) > 2 else "fp64"   # fp64 | reference_f32 | reference_f32_host (OP_TRACK_OPT_SUMS)
lib = L.load()
odo = O.Odometry(I.PinholeCamera("OPEN3D_DATASET"))
odo.SetSums(sums)
for (i, j, term, name) in [(300, 301, 0, "hybrid 300->301"), (100, 102, 0, "hybrid 100->102"), (300, 301, 2, "depth 300->301")]:
    # pyramids built by the library itself from the two raw frames (DenseTracking), then read back
    di, ci, _ = S.room_frame(i)
    dj, cj, _ = S.room_frame(j)
    odo.SetMultiScale(3); odo.iter_count_per_level = [4, 8, 16]
    odo.DenseTracking(ci, cj, di, dj, None, 0, want_correspondences=False)
    levels = odo.PreparedLevels()
    dev = []
    for lv in levels:
        d = dict(lv)
        for k in O.TRACK_IMAGES:
            d[k] = torch.from_numpy(np.ascontiguousarray(lv[k])).cuda()
        dev.append(d)
    arr, mem, keep = O._levels_arg(dev)
    iters = np.array([4, 8, 16], np.int32)
    T0 = np.eye(4, dtype=np.float32).reshape(16)
    res = L.TrackResult()
    call = lambda: L.check(lib.op_tracker_track(odo._h, arr, 3, iters.ctypes.data_as(L._ip), 640, 480, term, T0.ctypes.data_as(L._fp),
                                                mem, C.byref(res), None, None, 0, None, None))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        call()
    dt = time.perf_counter() - t
    print("%-18s device pyramids: %.1f tracks/s  %.3f ms/track  iterations %d  n %d" % (name, reps / dt, dt / reps * 1e3, res.iterations, res.n_correspondences))
    # host pyramids (PCIe-inclusive)
    arr_h, mem_h, keep_h = O._levels_arg(levels)
    call_h = lambda: L.check(lib.op_tracker_track(odo._h, arr_h, 3, iters.ctypes.data_as(L._ip), 640, 480, term, T0.ctypes.data_as(L._fp),
                                                  mem_h, C.byref(res), None, None, 0, None, None))
    call_h()
    t = time.perf_counter()
    for _ in range(max(reps // 10, 5)):
        call_h()
    dt = time.perf_counter() - t
    print("%-18s host pyramids  : %.1f tracks/s  %.3f ms/track" % (name, max(reps // 10, 5) / dt, dt / max(reps // 10, 5) * 1e3))
