"""How long the HOST spends inside IntegrateSequence (enqueue only) vs. how long the GPU needs: 20 steps of 100 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onepiece_amd import integration as I, synthetic as S
dev = torch.device("cuda:0")
depth, rgb, poses = S.room_sequence_torch(0, 2000, dev)
torch.cuda.synchronize()
hv = I.CubeHandler(device=0); hv.SetVoxelResolution(0.005)
for rep in range(3):
    hv.Clear(); hv.Synchronize()
    calls = []
    t0 = time.perf_counter()
    for k in range(20):
        t = time.perf_counter()
        hv.IntegrateSequence(depth[k * 100:(k + 1) * 100], rgb[k * 100:(k + 1) * 100], poses[k * 100:(k + 1) * 100])
        calls.append(time.perf_counter() - t)
    t1 = time.perf_counter()
    hv.Synchronize()
    t2 = time.perf_counter()
    print("rep %d: enqueue loop %.2f ms (calls min %.3f median %.3f max %.3f ms), final sync %.2f ms, total %.2f ms -> %.0f frames/s" % (
        rep, (t1 - t0) * 1e3, min(calls) * 1e3, sorted(calls)[10] * 1e3, max(calls) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, 2000 / (t2 - t0)), hv.GrowthStats())
