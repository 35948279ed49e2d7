"""Early timing probe (not the bench contract): per-frame time of IntegrateSequence."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
torch.cuda.synchronize()
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # initial pool capacity in blocks (0 = library default; the pool grows on demand)
hv = I.CubeHandler(max_blocks=mb); hv.SetVoxelResolution(0.005)
if os.environ.get("QB_UPDATE"):   # QB_UPDATE=sum_form: the opt-in once-per-batch update
    hv.SetUpdateMode(os.environ["QB_UPDATE"])
hv.IntegrateSequence(depth[:10], rgb[:10], poses[:10]); hv.Synchronize()
for rep in range(3):
    hv.Clear()
    t = time.time(); hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize(); dt = time.time() - t
    st = hv.Stats()
    print("rep", rep, "frames", n, "ms/frame %.4f" % (dt / n * 1e3), "fps %.1f" % (n / dt), st, "blocks", hv.BlockCount())
    bytes_ = 40 * st["voxels_updated"] + 7 * 307200 * st["frames"]
    print("   algorithmic GB/s over whole path: %.1f" % (bytes_ / dt / 1e9))
hv.Clear(); hv.ProfileEnable(1)
hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize()
print(hv.ProfileRead())
