"""Timing of the CubeHandler rows around fusion on a 5 mm volume fused from N room frames:
Transform (trilinear), TransformNearest, GetPointCloud, Merge, WriteToFile / ReadFromFile -- HIP path, and with a
second argument the CPU oracle on the same volume (one core, like the reference).
usage: volume_ops_bench.py [frames=100] [cpu]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
cam = I.PinholeCamera("OPEN3D_DATASET")
vol = I.CubeHandler(cam, max_blocks=1 << 19); vol.SetVoxelResolution(0.005)
vol.IntegrateSequence(depth, rgb, poses); vol.Synchronize()
other = I.CubeHandler(cam, max_blocks=1 << 19); other.SetVoxelResolution(0.005)
d2, c2, p2 = S.room_sequence_torch(500, n, dev)
other.IntegrateSequence(d2, c2, p2); other.Synchronize()
T = np.array(poses[n // 2], np.float32)
tmp = tempfile.mkdtemp()


def tm(f, reps=3):
    best = 1e9; r = None
    for _ in range(reps):
        t = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t)
    return best * 1e3, r


print("volume: %d blocks (%d frames at 5 mm); second volume %d blocks" % (vol.BlockCount(), n, other.BlockCount()))
ms, tn = tm(lambda: (lambda v: (v.BlockCount(), v)[1])(vol.TransformNearest(T, max_blocks=1 << 19)))
print("GPU TransformNearest : %8.2f ms -> %d blocks" % (ms, tn.BlockCount()))
ms, tt = tm(lambda: (lambda v: (v.BlockCount(), v)[1])(vol.Transform(T, max_blocks=1 << 19)))
print("GPU Transform        : %8.2f ms -> %d blocks" % (ms, tt.BlockCount()))
ms, pc = tm(lambda: vol.GetPointCloud())
print("GPU GetPointCloud    : %8.2f ms -> %d points (incl. download)" % (ms, len(pc[0])))
k0, v0 = vol.GetCubeMap()


def merge_once():
    m = I.CubeHandler(cam, max_blocks=1 << 19); m.SetVoxelResolution(0.005); m.SetCubeMap(k0, v0); m.Synchronize()
    t = time.perf_counter(); m.Merge(other); n_after = m.BlockCount(); return time.perf_counter() - t, n_after


best = min(merge_once() for _ in range(3))
print("GPU Merge            : %8.2f ms -> %d blocks" % (best[0] * 1e3, best[1]))
path = os.path.join(tmp, "v.map")
ms, _ = tm(lambda: vol.WriteToFile(path), reps=2)
print("GPU WriteToFile      : %8.2f ms (%.0f MB)" % (ms, os.path.getsize(path) / 1e6))


def read_once():
    m = I.CubeHandler(cam, max_blocks=1 << 19); m.SetVoxelResolution(0.005)
    t = time.perf_counter(); m.ReadFromFile(path); nb = m.BlockCount(); return time.perf_counter() - t, nb


best = min(read_once() for _ in range(2))
print("GPU ReadFromFile     : %8.2f ms -> %d blocks" % (best[0] * 1e3, best[1]))
if len(sys.argv) > 2:
    from oracle import oracle as O
    ov = O.Volume(voxel_res=0.005); ov.load(k0, v0)
    k1, v1 = other.GetCubeMap()
    oo = O.Volume(voxel_res=0.005); oo.load(k1, v1)
    t = time.perf_counter(); r = ov.transform(T, nearest=True); print("CPU TransformNearest : %8.2f ms -> %d blocks" % ((time.perf_counter() - t) * 1e3, r.block_count()))
    t = time.perf_counter(); r = ov.transform(T, nearest=False); print("CPU Transform        : %8.2f ms -> %d blocks" % ((time.perf_counter() - t) * 1e3, r.block_count()))
    t = time.perf_counter(); p = ov.point_cloud(); print("CPU GetPointCloud    : %8.2f ms -> %d points" % ((time.perf_counter() - t) * 1e3, len(p[0])))
    t = time.perf_counter(); ov.write_file(os.path.join(tmp, "o.map")); print("CPU WriteToFile      : %8.2f ms" % ((time.perf_counter() - t) * 1e3))
    o2 = O.Volume(voxel_res=0.005)
    t = time.perf_counter(); o2.read_file(os.path.join(tmp, "o.map")); print("CPU ReadFromFile     : %8.2f ms" % ((time.perf_counter() - t) * 1e3))
    t = time.perf_counter(); ov.merge(oo); print("CPU Merge            : %8.2f ms -> %d blocks" % ((time.perf_counter() - t) * 1e3, ov.block_count()))
