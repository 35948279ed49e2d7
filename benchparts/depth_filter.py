"""The K steps behind the drivers' depth front end.

One section of bench.py's JSON line (bench.py builds the context `c` -- the fused volume, the frames in HBM, the timed region's counters -- and calls run(c, out))."""
import json
import os
import sys
import time

import numpy as np


# ---- the same K steps behind the drivers' depth front end (tool::ConvertDepthTo32F + tool::BilateralFilter,
# ImageSequenceIntegration.cpp:36-38) from raw 16-bit depth, filter enqueued on the volume's stream.  Supplementary:
# the filter is OpenCV's in the reference (unpinned), so the headline `value` above stays without it.
def run(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    from onepiece_amd import tool as T
    d16 = (depth * 1000.0).round().clamp(0, 65535).to(torch.uint16)
    fbuf = torch.empty_like(depth)
    torch.cuda.synchronize()
    best = None
    for rep in range(2):
        hv.Clear(); hv.Synchronize()
        t = time.perf_counter()
        for k in range(K):
            s = k * F
            T.BilateralFilter(d16[s:s + F], depth_scale=1000.0, stream=hv.Stream(), out=fbuf[s:s + F])
            hv.IntegrateSequence(fbuf[s:s + F], rgb[s:s + F], poses[s:s + F])
        hv.Synchronize()
        dtf = time.perf_counter() - t
        best = dtf if best is None else min(best, dtf)
    T.BilateralFilter(d16[:F], depth_scale=1000.0, out=fbuf[:F])   # creates the library's stream for stream-less calls
    t = time.perf_counter()
    T.BilateralFilter(d16, depth_scale=1000.0, out=fbuf)            # all K*F images in one call, final on return
    t_filter = time.perf_counter() - t
    out["with_depth_filter"] = {"frames_per_s": n_local / best, "filter_us_per_image": t_filter / n_local * 1e6,
                                "filter": "ConvertDepthTo32F + BilateralFilter(d=7, 0.03, 4.5) from uint16 depth, k_bilateral on the volume's stream",
                                "blocks": hv.BlockCount()}
    del d16, fbuf
