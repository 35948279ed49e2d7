"""Device time of tool::BilateralFilter (op_bilateral_filter_depth) on batches of 640x480 depth images resident in HBM,
alone and in front of CubeHandler::IntegrateSequence on the volume's stream (the fusion drivers' per-frame front end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, tool as T

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
d16 = (depth * 1000.0).round().clamp(0, 65535).to(torch.uint16)
out = torch.empty_like(depth)
torch.cuda.synchronize()
for name, src in (("float32", depth), ("uint16", d16)):
    T.BilateralFilter(src, out=out)
    t = time.perf_counter()
    for _ in range(5):
        T.BilateralFilter(src, out=out)
    dt = (time.perf_counter() - t) / 5
    byts = src.numel() * src.element_size() + out.numel() * 4
    print("BilateralFilter %s: %.1f us / image, %.0f images/s, %.0f GB/s of image traffic" % (name, dt / n * 1e6, n / dt, byts / dt / 1e9))
hv = I.CubeHandler(max_blocks=1 << 19); hv.SetVoxelResolution(0.005)
for filt in (False, True):
    for rep in range(2):
        hv.Clear(); hv.Synchronize()
        t = time.perf_counter()
        for s in range(0, n, 100):
            d = depth[s:s + 100]
            if filt:
                d = T.BilateralFilter(d, stream=hv.Stream(), out=out[s:s + 100])
            hv.IntegrateSequence(d, rgb[s:s + 100], poses[s:s + 100])
        hv.Synchronize()
        dt = time.perf_counter() - t
    print("fusion %s the bilateral filter: %.0f frames/s" % ("with" if filt else "without", n / dt))
