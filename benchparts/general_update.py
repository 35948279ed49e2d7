"""Fusing into a volume that other writers touched (general update form).

One section of bench.py's JSON line (bench.py builds the context `c` -- the fused volume, the frames in HBM, the timed region's counters -- and calls run(c, out))."""
import json
import os
import sys
import time

import numpy as np


# ---- fusing into a volume that was NOT written by the integrate kernel alone (after SetCubeMap / ReadFromFile / Merge -- the reference's
# MergeMultipleSubmaps / FBAFusion pattern): the update then takes the general form (two branches, four true divisions per voxel)
def run(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    ng = min(400, n_local)
    rates = {}
    for name in ("plain", "after_upload"):
        hv.Clear()
        hv.IntegrateSequence(depth[:20], rgb[:20], poses[:20])
        if name == "after_upload":
            k_, v_ = hv.GetCubeMap(sort=False)
            hv.SetCubeMap(k_, v_)                       # same content, but now "foreign" data: k_integrate<., PLAIN=false>
            del k_, v_
        hv.Synchronize()
        t = time.perf_counter()
        hv.IntegrateSequence(depth[20:ng], rgb[20:ng], poses[20:ng])
        hv.Synchronize()
        rates[name] = (ng - 20) / (time.perf_counter() - t)
    out["general_update_path"] = {"frames": ng - 20, "plain_frames_per_s": rates["plain"], "after_upload_frames_per_s": rates["after_upload"],
                                  "note": "frames/s of IntegrateSequence into a volume holding 20 fused frames: as fused (shared-reciprocal update) vs after the "
                                          "same content went through GetCubeMap / SetCubeMap (general update with IEEE divisions; results identical)"}
