"""Tracking + fusion in the reference-order mode: frames/s by pairs in flight, with the trackers' sequential sums taken per tracker (OP_RUNTIME_OPT_TRACKER_BATCH_SUMS = 0)
or together in one launch per round once twelve or more trackers run (= 1).  python tools/track_depth_probe.py [frames=200] [reference_f32|fp64]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, dense_slam as DS, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sums = sys.argv[2] if len(sys.argv) > 2 else "reference_f32"   # or fp64
dev = torch.device("cuda", 0)
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
cam = I.PinholeCamera("OPEN3D_DATASET")
lib = L.load()


def run(pipe):
    vol = I.CubeHandler(cam); vol.SetVoxelResolution(0.005)
    slam = DS.DenseSlam(cam, pipeline=pipe, on_tracked=lambda fid, c, d, T: vol.IntegrateImage(d, c, T))
    slam.SetSums(sums)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        slam.UpdateFrame(rgb[i], depth[i])
    slam.Finish(); vol.BlockCount()
    return n / (time.perf_counter() - t), np.array(slam.global_poses)


ref = None
for batch in (0, 1):
    L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_TRACKER_BATCH_SUMS, batch))
    for pipe in (4, 8, 12, 16, 24):
        run(pipe)
        r = max(run(pipe), run(pipe), key=lambda x: x[0])
        if ref is None:
            ref = r[1]
        print("batch_sums %d  pairs in flight %2d: %7.1f frames/s  poses identical to the first run: %s" % (batch, pipe, r[0], bool(np.array_equal(r[1], ref))), flush=True)
