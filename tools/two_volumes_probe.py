"""How much of the chip does one fusion stream leave idle?  Two independent volumes fuse the same 1000 frames on their own HIP streams,
concurrently, vs one after the other: if the concurrent run is faster than 2x the single one, kernels of different batches overlap
usefully (the headroom a KA/KB-behind-KC pipeline inside ONE volume could reach)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onepiece_amd import integration as I, synthetic as S
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
torch.cuda.synchronize()
vols = [I.CubeHandler(device=0) for _ in range(2)]
for v in vols:
    v.SetVoxelResolution(0.005); v.IntegrateSequence(depth[:32], rgb[:32], poses[:32]); v.Synchronize()
def run(which, chunk=100):
    for v in which: v.Clear()
    for v in which: v.Synchronize()
    t = time.perf_counter()
    for s in range(0, n, chunk):
        for v in which: v.IntegrateSequence(depth[s:s + chunk], rgb[s:s + chunk], poses[s:s + chunk])
    for v in which: v.Synchronize()
    return time.perf_counter() - t
for rep in range(3):
    t1 = run(vols[:1]); t2 = run(vols)
    print("rep %d: one volume %.2f ms (%.0f frames/s); two concurrent volumes %.2f ms (%.0f frames/s aggregate, %.2fx one volume's rate)" % (
        rep, t1 * 1e3, n / t1, t2 * 1e3, 2 * n / t2, (2 * n / t2) / (n / t1)))
