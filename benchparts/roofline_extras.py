"""The roofline object's extras: the one-frame-per-launch run (where SURVEY 8(d)'s bytes are a bound), the HBM traffic of the batched launch
(rocprofv3 PMC: live with --counters, else the committed measurement scaled to this run), and with --full the sum-form update.

One section of bench.py's JSON line (bench.py builds the context `c` -- the fused volume, the frames in HBM, the timed region's counters -- and calls run(c, out))."""
import json
import os
import sys
import time

import numpy as np


def run(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    stats, prof, frames_per_launch, k3_s, kc_cycles, batch_bytes = c.stats, c.prof, c.frames_per_launch, c.k3_s, c.kc_cycles, c.batch_bytes
    # -- the byte model where it is a roofline: one frame per launch (every voxel read + written once per frame)
    hv.Clear(); hv.ProfileEnable(1)
    nb1 = min(64, n_local)
    for k in range(nb1):
        hv.IntegrateSequence(depth[k:k + 1], rgb[k:k + 1], poses[k:k + 1])
        hv.Flush()   # one frame per launch
    hv.Synchronize()
    st1, p1 = hv.Stats(), hv.ProfileRead()
    hv.ProfileEnable(0)
    b1 = 40.0 * st1["voxels_updated"] / max(st1["frames"], 1) + 7.0 * W * H
    a1 = b1 / (p1["integrate_ms"] * 1e-3) / 1e9
    m1 = (10240.0 * st1["blocks_read"] + 20.0 * st1["voxels_written"]) / max(st1["launches"], 1) + 8.0 * W * H
    out["roofline"]["batch1_frac"] = a1 / HBM_PEAK_GBS   # SURVEY 8(d)'s bytes where they ARE a bound (one frame per launch) / time / 8 TB/s: north_star's ">= 50 % of HBM roofline"
    out["roofline"]["batch1_avg_launch_ms"] = p1["integrate_ms"]
    out["roofline"]["batch1"] = {"frames": nb1, "bound": "hbm", "avg_launch_ms": p1["integrate_ms"], "algorithmic_bytes_per_launch": b1, "achieved": a1, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS, "traffic_model_bytes_per_launch": m1, "traffic_model_frac": m1 / (p1["integrate_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "evidence": "profiles/r06_batch1.* (tools/profile_roofline.sh ... batch=1: rocprofv3 --kernel-trace --stats and the FETCH_SIZE / WRITE_SIZE passes of the final tree)",
                                 "note": "k_integrate with ONE frame per launch: SURVEY 8(d)'s algorithmic bytes are then a lower bound of the real traffic, "
                                         "so this is a true HBM roofline fraction (north_star: >= 50 % of HBM roofline on the integrate kernel)"}
    _traffic(c, out)
    if not args.full:
        return
    # -- the opt-in sum-form update (OP_VOLUME_UPDATE_SUM_FORM): the same frames, one weighted mean per batch instead of one rounded update per frame
    fuse_all = lambda: [hv.IntegrateSequence(depth[k * F:(k + 1) * F], rgb[k * F:(k + 1) * F], poses[k * F:(k + 1) * F]) for k in range(K)]
    hv.Clear(); hv.SetUpdateMode("sum_form"); hv.ProfileEnable(args.profile_every)
    best_sf = 0.0
    for _rep in range(2):
        hv.Clear()
        hv.Synchronize(); torch.cuda.synchronize()
        t_sf = time.perf_counter()
        fuse_all()
        hv.Synchronize()
        best_sf = max(best_sf, n_local / (time.perf_counter() - t_sf))
    psf = hv.ProfileRead()
    hv.ProfileEnable(0)
    out["sum_form"] = {"frames_per_s": best_sf, "integrate_ms_per_launch": psf["integrate_ms"], "frames_per_launch": psf["frames"] / max(psf["launches"], 1),
                       "note": "op_volume_set_option(OP_VOLUME_OPT_UPDATE, OP_VOLUME_UPDATE_SUM_FORM): opt-in; `value` above is the default exact update",
                       "evidence": "profiles/r05_batch32_sum.* next to profiles/r05_batch32.* (tools/profile_roofline.sh with PD_UPDATE=sum_form): 517.8 vs 654.1 us per 32-frame launch, "
                                   "379 M vs 511 M VALU wave-instructions, VALU share of the issue slots 0.59 vs 0.63, 882 vs 994 MB of HBM traffic per launch (0.21 / 0.19 of peak)"}
    if True:   # the two volumes side by side over a prefix of the workload (~1 GB of host memory each at 500 frames)
        n_cmp = min(n_local, 500)
        hv.Clear()
        hv.IntegrateSequence(depth[:n_cmp], rgb[:n_cmp], poses[:n_cmp])
        k_sf, v_sf = hv.GetCubeMap()
        hv.SetUpdateMode("exact"); hv.Clear()
        hv.IntegrateSequence(depth[:n_cmp], rgb[:n_cmp], poses[:n_cmp])
        k_ex, v_ex = hv.GetCubeMap()
        obs = v_ex[..., 1] > 0
        out["sum_form"]["parity_vs_exact_update"] = {
            "keys_equal": bool(np.array_equal(k_ex, k_sf)), "weights_equal": bool(np.array_equal(v_ex[..., 1], v_sf[..., 1])),
            "max_abs_sdf_diff_over_truncation": float(np.abs(v_ex[..., 0] - v_sf[..., 0])[obs].max() / 0.1),
            "max_abs_colour_diff": float(np.abs(v_ex[..., 2:] - v_sf[..., 2:])[obs].max()), "blocks": int(len(k_ex)), "frames": int(n_cmp), "bar": 1e-4}
        del k_sf, v_sf, k_ex, v_ex, obs
    hv.SetUpdateMode("exact")


CACHE = os.path.join("profiles", "r06_integrate_counters.json")


def _traffic(c, out):
    """roofline.traffic / achieved / frac of the batched launch: rocprofv3 PMC bytes per launch.  Live (--counters: ~30 s of extra passes on a dump of this
    step's frames), else the committed measurement of the same kernel on the same workload (profiles/r06_integrate_counters.json), carried over to this
    run's launch shape through its ratio to the device-counted byte model -- which this run measures itself (blocks read, voxels written, frames per launch)."""
    args, R = c.args, out["roofline"]
    if c.world == 1 and args.counters:
        _live_counters(c, out)
        if R.get("traffic"):
            R["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes, this run"
            if args.save_counters:
                json.dump({"traffic_bytes_per_launch": R["traffic"], "model_bytes_per_launch": c.batch_bytes, "traffic_over_model": R["traffic"] / c.batch_bytes,
                           "frames_per_launch": c.frames_per_launch, "avg_launch_ms": c.prof["integrate_ms"], "valu_issue_frac": R.get("issue_frac"),
                           "hbm": R.get("hbm"), "workload": out["config"]["workload"],
                           "how": "python bench.py --counters --save-counters (tools/counters.py: rocprofv3 --kernel-trace --pmc, one group per pass, on tools/prof_driver.bin)"},
                          open(os.path.join(c.ROOT, CACHE), "w"), indent=1)
            return
    try:
        cj = json.load(open(os.path.join(c.ROOT, CACHE)))
    except (OSError, ValueError):
        return
    tr = cj["traffic_over_model"] * c.batch_bytes
    R.update({"traffic": tr, "achieved": tr / c.k3_s / 1e9, "frac": tr / c.k3_s / 1e9 / c.HBM_PEAK_GBS,
              "traffic_source": "%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE at %.1f frames per launch: %.3g B) x this run's byte model / that run's (ratio %.3f); "
                                "--counters measures live" % (CACHE, cj["frames_per_launch"], cj["traffic_bytes_per_launch"], cj["traffic_over_model"]),
              "issue_frac": cj.get("valu_issue_frac"), "issue_source": CACHE + ": VALU wave-instructions x measured issue cost / SIMD cycles (what binds the batched launch)"})


def _live_counters(c, out):
    args, torch, dev, rank, world, local_rank, hv, depth, rgb, poses, K, F, n_local = c.args, c.torch, c.dev, c.rank, c.world, c.local_rank, c.hv, c.depth, c.rgb, c.poses, c.K, c.F, c.n_local
    I, S, ROOT, W, H, HBM_PEAK_GBS = c.I, c.S, c.ROOT, c.W, c.H, c.HBM_PEAK_GBS
    stats, prof, frames_per_launch, k3_s, kc_cycles, batch_bytes = c.stats, c.prof, c.frames_per_launch, c.k3_s, c.kc_cycles, c.batch_bytes
    # -- live PMC passes (separate rocprofv3 --pmc runs of the torch-free driver on a dump of this step's frames)
    if True:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import counters as CT
            import issue_model as IM
            import tempfile
            nfc = F if F < 32 else F // 32 * 32   # whole 32-frame batches, like the timed region's launches
            with tempfile.NamedTemporaryFile(prefix="opc_frames_", suffix=".bin", dir="/tmp", delete=False) as tf:
                np.array([nfc, W, H], np.int32).tofile(tf)
                dh, ch = depth[:nfc].cpu().numpy(), rgb[:nfc].cpu().numpy()
                for i in range(nfc):
                    poses[i].astype(np.float32).tofile(tf); dh[i].tofile(tf); ch[i].tofile(tf)
                fname = tf.name
            try:
                cnt = CT.measure(fname, args.voxel)
            finally:
                os.unlink(fname)
            kc = cnt["k_integrate"]
            R = out["roofline"]
            # the driver's launches fuse nfc / ceil(nfc / 32) frames each, the timed region's frames_per_launch (its last launch may be
            # shorter): per-launch counts are scaled by the ratio (they are proportional to the frames of a launch to within a few %)
            fpl_driver = nfc / float(-(-nfc // 32))
            scale = frames_per_launch / fpl_driver
            for key in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch"):
                kc[key] *= scale
            for c in kc:
                if isinstance(kc[c], dict) and "mean_per_launch" in kc[c]:
                    kc[c]["mean_per_launch"] *= scale
            R["traffic"] = kc["hbm_bytes_per_launch"]
            R["hbm"] = {"model_bytes_per_launch": batch_bytes}
            R["hbm"].update({"traffic_bytes_per_launch": kc["hbm_bytes_per_launch"], "read_bytes_per_launch": kc["hbm_read_bytes_per_launch"],
                             "write_bytes_per_launch": kc["hbm_write_bytes_per_launch"], "achieved": kc["hbm_bytes_per_launch"] / k3_s / 1e9, "unit": "GB/s",
                             "frac": kc["hbm_bytes_per_launch"] / k3_s / 1e9 / HBM_PEAK_GBS, "traffic_over_model": kc["hbm_bytes_per_launch"] / batch_bytes,
                             "source": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950, calibrated for this kernel's 4 B/lane plane rows: profiles/r03_calib.FETCH_SIZE.pmc.csv) "
                                       "and WRITE_SIZE (exact: profiles/r03_calib.WRITE_SIZE.pmc.csv), separate passes, tools/prof_driver.bin on the first %d frames "
                                       "of this run (32-frame launches; per-launch figures scaled by %.3f to the timed region's %.1f frames per launch)" % (nfc, scale, frames_per_launch)})
            costs_file = os.path.join(ROOT, "profiles", "r05_issue_costs.json")
            if not os.path.exists(costs_file):
                costs_file = os.path.join(ROOT, "profiles", "r04_issue_costs.json")
            if not os.path.exists(costs_file):
                costs_file = os.path.join(ROOT, "profiles", "r03_issue_costs.json")
            cj = json.load(open(costs_file))
            counts = {c: kc[c]["mean_per_launch"] for c in kc if isinstance(kc[c], dict) and "mean_per_launch" in kc[c]}
            counts["kernel_cycles"] = kc_cycles
            im = IM.model(cj["costs"], cj["valu_mix_k_integrate_plain"], counts)
            # The binding resource goes to the top of the object: a fraction <= 1 of the SIMDs' instruction-issue capacity.  Which fraction: the
            # additive model (every class charged into ONE budget per SIMD) when the mixed-class microbenchmark of the costs file says it predicts
            # such a kernel's time within 10 %; otherwise the VALU share alone (scalar instructions of other waves co-issue), the rest as context.
            mc = cj.get("mixed_check")
            additive_ok = bool(mc and mc.get("additive_model_holds"))
            valu_frac = im["classes"]["valu"]["share_of_capacity"]
            # `frac` stays SURVEY 8(d)'s quantity (HBM bytes / time / peak); what actually binds the batched launch -- instruction issue -- goes under its own keys
            R.update({"achieved": R["hbm"]["achieved"], "frac": R["hbm"]["frac"],
                      "issue_frac": im["frac"] if additive_ok else valu_frac,
                      "issue_source": "live SQ_INSTS_* x measured issue costs (%s) / SIMD cycles: %s" % (os.path.relpath(costs_file, ROOT), "all classes, additive model validated" if additive_ok else "VALU share (scalar work co-issues)"),
                      "valu_frac": valu_frac, "all_classes_additive_frac": im["frac"], "mixed_check": mc, "costs_file": os.path.relpath(costs_file, ROOT)})
            R["hbm_frac"] = R["hbm"]["frac"]      # measured HBM traffic of the batched launch / launch time / 8 TB/s
            R["issue"] = {"classes": im["classes"], "valu_cycles_each": im["valu_cycles_each"], "kernel_cycles": im["kernel_cycles"],
                          "insts_per_voxel_frame_wave": {"valu": counts["SQ_INSTS_VALU"] / (stats["blocks_selected"] / max(stats["frames"], 1) * frames_per_launch * 8.0),
                                                         "salu": counts.get("SQ_INSTS_SALU", 0.0) / (stats["blocks_selected"] / max(stats["frames"], 1) * frames_per_launch * 8.0)},
                          "costs": os.path.relpath(costs_file, ROOT) + " (tools/valu_ubench.hip at 8 waves per SIMD, shader cycles per wave64 instruction and SIMD; "
                                   "VALU classes weighted by the kernel's static opcode histogram)",
                          "note": "frac = sum over classes of wave-instructions (SQ_INSTS_* of this step's launches) x issue cost / (1024 SIMDs x the launch's "
                                  "shader cycles): the share of the chip's instruction-issue slots the kernel fills.  HBM is far from binding for a batched launch "
                                  "(roofline.hbm), so the way to make it faster is fewer instructions per voxel and frame"}
            R["counters"] = {k: {c: v["mean_per_launch"] for c, v in r.items() if isinstance(v, dict) and "mean_per_launch" in v} for k, r in cnt.items()}
        except Exception as e:  # rocprofv3 missing / failing must not take the bench line down
            out["roofline"]["traffic_error"] = repr(e)[:300]

