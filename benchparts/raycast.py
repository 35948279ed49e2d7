"""The raycaster north_star names (the reference has none, F2): rays/s for 640 x 480 views of the volume the timed region fused.

One section of bench.py's JSON line (bench.py builds the context `c` and calls run(c, out))."""
import time

import numpy as np


def run(c, out):
    torch, dev, hv, poses, n_local, W, H, HBM_PEAK_GBS = c.torch, c.dev, c.hv, c.poses, c.n_local, c.W, c.H, c.HBM_PEAK_GBS
    npx = W * H
    d = torch.empty((H, W), dtype=torch.float32, device=dev)
    nr = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    cl = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    views = [poses[(k * n_local) // 8] for k in range(8)]           # eight views spread over the sequence the volume was fused from
    blocks = hv.BlockCount()
    res = {"rays_per_view": npx, "views": len(views), "volume_blocks": int(blocks), "voxel_m": float(c.args.voxel),
           "definition": "fixed lattice t_k = near + k * res along every pixel ray, trilinear sdf over observed voxels, first + -> - lattice pair, linear "
                         "interpolation (include/onepiece_hip.h: op_volume_raycast); outputs stay in HBM; ms = call to completion (op_volume_raycast returns "
                         "when the images are final)"}
    for name, want in (("depth_only", False), ("depth_normals_colours", True)):
        for mode, prune in (("cold", False), ("warm", True)):
            # cold: every view loads every visible block -- what the first view after a change of the volume costs; warm: later views of the
            # unchanged volume drop blocks by what earlier views learnt about them (the default, OP_VOLUME_OPT_RAYCAST_PRUNE)
            hv.SetRaycastPrune(prune)
            ts, st = [], {"visible_blocks": 0, "dropped_unloaded": 0, "loaded_blocks": 0, "marched_blocks": 0}
            hits = 0
            for rep in range(4):
                for p in views:
                    t = time.perf_counter()
                    hv.RaycastDevice(p, d.data_ptr(), nr.data_ptr() if want else 0, cl.data_ptr() if want else 0)
                    dt = time.perf_counter() - t
                    if rep == 0:
                        hits += int((d > 0).sum())
                        continue                                       # (warm-up round; with pruning on, the round that learns)
                    ts.append(dt)
                    s = hv.RaycastStats()
                    for k in st:
                        st[k] += s[k]
            n = len(ts)
            mean = float(np.mean(ts))
            loaded = st["loaded_blocks"] / n
            alg = loaded * 4096.0 + 4.0 * npx      # the sdf + weight planes (2 x 2 KB) of every block loaded, once, + the depth image
            tile = loaded * 1331 * 8.0             # what the kernel asks for: an 11^3 tile per loaded block (own voxels + the shell of the 26 neighbours), 8 B per voxel
            res.setdefault(name, {})[mode] = {
                "ms_per_view": mean * 1e3, "best_ms": float(np.min(ts)) * 1e3, "rays_per_s": npx / mean, "hit_fraction": hits / float(npx * len(views)),
                "per_view": {k: v / n for k, v in st.items()},
                "algorithmic_bytes_per_view": alg, "algorithmic_gbs": alg / mean / 1e9, "frac_of_hbm_peak": alg / mean / 1e9 / HBM_PEAK_GBS,
                "tile_bytes_per_view": tile, "tile_gbs": tile / mean / 1e9}
    hv.SetRaycastPrune(True)
    # A live pipeline (track against the model, fuse, view again) changes the volume between any two views.  The exact update restates the summaries of
    # the blocks it changes (k_integrate), so the view after a fusion still prunes: one frame fused (untimed), then one view (timed), eight times over.
    try:
        depth, rgb = c.depth, c.rgb
        ts, st = [], {"visible_blocks": 0, "dropped_unloaded": 0, "loaded_blocks": 0, "marched_blocks": 0}
        for rep in range(3):
            for k, p in enumerate(views):
                i = (k * n_local) // 8
                hv.IntegrateSequence(depth[i:i + 1], rgb[i:i + 1], poses[i:i + 1])
                hv.Synchronize()
                t = time.perf_counter()
                hv.RaycastDevice(p, d.data_ptr(), 0, 0)
                dt = time.perf_counter() - t
                if rep == 0:
                    continue
                ts.append(dt)
                s = hv.RaycastStats()
                for kk in st:
                    st[kk] += s[kk]
        n = len(ts)
        res["depth_only"]["view_after_fusing_a_frame"] = {"ms_per_view": float(np.mean(ts)) * 1e3, "best_ms": float(np.min(ts)) * 1e3, "per_view": {k: v / n for k, v in st.items()},
                                                          "note": "between any two of these views a frame was fused with the default (exact) update: the blocks it changed had their "
                                                                  "summaries restated by k_integrate, the others kept theirs -- dropped_unloaded stays where the warm views have it "
                                                                  "(before the end of round 5 every fusion invalidated all summaries and such a view cost what `cold` costs)"}
    except Exception as e:  # noqa
        res["depth_only"]["view_after_fusing_a_frame"] = {"error": repr(e)[:200]}
    res["before_round_5"] = {"ms_per_view_depth_only": 1.121, "ms_per_view_depth_normals_colours": 1.239, "volume": "the same scene from 250 frames (164 k blocks)",
                             "evidence": "profiles/r05_raycast_before.driver.txt / .kernel_stats.csv (one thread per ray, one-voxel steps through the hash)"}
    res["bound"] = ("k_rc_march (89 us): ~60 % pixel march = VALU issue (33.7 M wave-instructions, ~85 per sample, 45-50 % lane utilisation), ~40 % tile loading = latency (waves wait on "
                    "memory 47 % of their cycles; 135 MB per launch over the fabric = 1.5 TB/s, 0.19 of the HBM peak; L2 hit rate 52 %): profiles/r05_raycast.*.pmc.csv, DESIGN.md section 3.5")
    res["evidence"] = "profiles/r05_raycast.driver.txt, .kernel_stats.csv, .FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum_TCC_MISS_sum / SQ_* .pmc.csv (tools/profile_volume_ops.sh), profiles/r05_ab_raycast.txt"
    out["raycast"] = res
    del d, nr, cl
