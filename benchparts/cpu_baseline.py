"""CPU baseline (the oracle, one host core) and parity on a bounded sample of the same frames.

One section of bench.py's JSON line (bench.py builds the context `c` -- the fused volume, the frames in HBM, the timed region's counters -- and calls run(c, out))."""
import json
import os
import sys
import time

import numpy as np


# ---- CPU baseline + parity on a bounded sample of the SAME frames (rank 0, N = 1 only)
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    O = c.oracle   # the CPU oracle, imported by bench.py for its cpu_baseline leg (the only place that does)
    ns = min(args.cpu_sample_frames, n_local)
    dn, cn = depth[:ns].cpu().numpy(), rgb[:ns].cpu().numpy()
    ov = O.Volume(voxel_res=args.voxel)
    t = time.perf_counter()
    for i in range(ns):
        ov.integrate(dn[i], cn[i], poses[i])
    cpu_dt = time.perf_counter() - t
    out["cpu_baseline"] = {"value": ns / cpu_dt, "unit": "frames/s", "cores": 1, "kind": "port",
                           "sample": "first %d frames of the same sequence fused by oracle/onepiece_oracle.c "
                                     "(the reference's integrate path is serial), host has %d cores" % (ns, os.cpu_count()),
                           "host_cores": os.cpu_count(), "cpu_model": _cpu_model()}
    # parity at the benchmark's own sizes: same sample through the HIP path, compared bit for bit
    hv2 = I.CubeHandler(device=local_rank)
    hv2.SetVoxelResolution(args.voxel)
    hv2.IntegrateSequence(depth[:ns], rgb[:ns], poses[:ns])
    hk, hvx = hv2.GetCubeMap()
    ok, ovx = ov.export()
    keys_equal = hk.shape == ok.shape and bool(np.array_equal(hk, ok))
    out["parity"] = {"sample_frames": ns, "blocks": int(len(ok)), "keys_equal": keys_equal,
                     "voxels_bit_equal": bool(keys_equal and np.array_equal(hvx.view(np.uint32), ovx.view(np.uint32)))}
    del hv2
