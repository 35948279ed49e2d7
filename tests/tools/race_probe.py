"""One-off: does fusing torch-generated frames WITHOUT draining torch's stream first race?  (It did once the library
stopped synchronising the device in hipMalloc / hipFree; the Python mirrors now drain torch's stream, L.torch_ready.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, _lib as L
dev = torch.device("cuda", 0)
ready = L.torch_ready
def run(drain):
    L.torch_ready = ready if drain else (lambda t: t)
    depth, rgb, poses = S.room_sequence_torch(40, 160, dev)
    hv = I.CubeHandler(max_blocks=1 << 18); hv.SetVoxelResolution(0.005)
    hv.IntegrateSequence(depth, rgb, poses)
    st = hv.Stats()
    return st["voxels_updated"], st["blocks_selected"]
ref = run(True)
for drain in (True, False):
    bad = sum(run(drain) != ref for _ in range(8))
    print("drain torch's stream first: %s -> %d of 8 runs differ from the reference" % (drain, bad))
