#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X TSDF-fusion + ICP hot path.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched by torch.distributed.run, one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], "ImageSequenceIntegration"): every rank fuses K steps x F
frames (default 10 x 100 = the 1000-frame sequence) of the synthetic 640x480 room sequence into
its own 5 mm voxel-block-hashed TSDF volume, frames already resident in HBM, starting from an
empty volume.  A "step" is one pass of the hot path (ComputeBounding -> PrepareCubes -> Integrate
for each frame) over one batch of F frames.  For N > 1 frames are sharded contiguously (rank r
fuses global frames [r*K*F, (r+1)*K*F)), there is no communication during fusion, and the timed
region ends with the single RCCL reduce that merges the per-GPU block hashes (weak scaling).
value = frames fused by all ranks / max-over-ranks wall time.

Extra objects on the JSON line: roofline (integrate kernel, HIP-event timed on the volume's own
stream, against 8 TB/s HBM), cpu_baseline (the CPU oracle = port of the reference path, timed on
this box's host cores on a bounded sample of the same frames; N=1, rank 0 only), parity (the GPU
volume for that sample compared bit-for-bit with the oracle's), icp (iterations/s at 307 200 points).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The tracking + fusion pipeline keeps four frame pairs in flight, each tracker on its own HIP stream, next to the volume's streams.  The
# runtime maps streams onto 4 hardware queues by default, so two of those streams would share a queue and serialise (2.6 k instead of 4.5 k
# frames/s in tools/prof_driver.bin track=4); with more than four ACTIVE queues the rate collapses again, so the pipeline depth stays 4.
# Read once, when the HIP runtime initialises -- hence here, before anything touches the GPU.  No effect on the fusion / ICP figures.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
W, H = 640, 480


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=100)
    ap.add_argument("--voxel", type=float, default=0.005)
    ap.add_argument("--cpu-sample-frames", type=int, default=40, help="frames of the workload the CPU baseline fuses (~0.25 s each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-icp", action="store_true")
    ap.add_argument("--no-tracking", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank fuses K x F frames; strong: the K x F frames are split over the ranks (BASELINE configs[4] as "
                         "written: `--scaling strong --steps 80` = 8000 frames in total at any N)")
    ap.add_argument("--timed-only", action="store_true", help="only the warm-up and the timed region (for rocprofv3 --kernel-trace --stats runs: "
                    "every k_integrate launch in the trace then has the timed region's batch shape)")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 PMC passes behind roofline.traffic / roofline.valu (~30 s)")
    ap.add_argument("--profile-every", type=int, default=1, help="HIP-event sample rate for the roofline (every k-th launch group)")
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu_baseline = args.no_icp = args.no_tracking = args.no_counters = True

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch N>1 with torch.distributed.run" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback path")
    # Test hooks (not used by the driver): ONEPIECE_BENCH_SINGLE_DEVICE=1 maps every rank to cuda:0 and
    # ONEPIECE_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N > 1 control flow can be exercised on a
    # one-GPU box (RCCL refuses two ranks on one device).
    if os.environ.get("ONEPIECE_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ONEPIECE_BENCH_FORCE_DIST=1: run the whole distributed path (process group over RCCL, key all_gather, sliced reduce,
    # normalisation) even with ONE rank -- how the RCCL merge is exercised on a single-GPU box.
    force_dist = os.environ.get("ONEPIECE_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["ONEPIECE_MERGE_FORCE"] = "1"
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ONEPIECE_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from onepiece_amd import integration as I, synthetic as S, distributed as D

    K, Wm, F = args.steps, args.warmup, args.frames_per_step
    if args.scaling == "strong":
        if F % world:
            raise SystemExit("--scaling strong needs --frames-per-step (%d) divisible by the number of GPUs (%d)" % (F, world))
        F //= world     # the step's frames are split: the job fuses K x frames_per_step frames in total at any N
    n_local = K * F
    first = rank * n_local  # contiguous shard of the global sequence
    # ---- inputs: generated straight into HBM, not timed
    depth, rgb, poses = S.room_sequence_torch(first, n_local, dev)
    torch.cuda.synchronize()

    hv = I.CubeHandler(device=local_rank)  # default pool (2^18 blocks, 2.7 GB); it grows on demand like the reference's map
    hv.SetVoxelResolution(args.voxel)
    ops = D.HipVolumeOps(hv, dev)

    def barrier():
        hv.Synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- warmup: W untimed steps (+ one merge so RCCL is initialised), then start from empty
    for w in range(Wm):
        s = (w % K) * F
        hv.IntegrateSequence(depth[s:s + F], rgb[s:s + F], poses[s:s + F])
    hv.Synchronize()
    if world > 1 or force_dist:
        D.merge_volumes(ops, root=0)
    hv.Clear()
    hv.ProfileEnable(args.profile_every)

    # ---- timed region: exactly K steps (+ the final merge for N > 1)
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        s = k * F
        hv.IntegrateSequence(depth[s:s + F], rgb[s:s + F], poses[s:s + F])
    hv.Synchronize()
    t_fuse = time.perf_counter() - t0
    stats = hv.Stats()  # per-rank counters, read before the merge rewrites the root volume
    growth = hv.GrowthStats()
    n_union = None
    local_blocks = hv.BlockCount()
    t_m0 = time.perf_counter()
    if world > 1 or force_dist:
        n_union = D.merge_volumes(ops, root=0)
        hv.Synchronize(); torch.cuda.synchronize()
    t_merge = time.perf_counter() - t_m0
    barrier()
    dt = time.perf_counter() - t0
    prof = hv.ProfileRead()
    hv.ProfileEnable(0)

    tmax = torch.tensor([dt, t_fuse], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max, t_fuse_max = float(tmax[0]), float(tmax[1])
    total_frames = n_local * world
    # what every rank did, so that a 1 -> 8 GPU curve decomposes into fusion and merge (gathered outside the timed region)
    mine = {"rank": rank, "frames": n_local, "fusion_ms": t_fuse * 1e3, "merge_ms": t_merge * 1e3, "local_blocks": int(local_blocks)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    out = None
    if rank == 0:
        n_upd_frame = stats["voxels_updated"] / max(stats["frames"], 1)
        alg_bytes_frame = 40.0 * n_upd_frame + 7.0 * W * H  # B_frame = 40*N_upd + 7*W*H
        frames_per_launch = prof["frames"] / max(prof["launches"], 1)  # k_integrate fuses a batch of frames per launch
        alg_bytes = alg_bytes_frame * frames_per_launch             # SURVEY 8d's figure for the units one launch processes
        k3_s = prof["integrate_ms"] * 1e-3
        n_launch = max(stats["launches"], 1)
        kc_cycles = stats["integrate_shader_cycles"] / n_launch     # shader cycles per launch (s_memtime span of the resident workgroups)
        # what a batched launch MUST move through HBM whatever the number of frames it fuses: every selected block read once
        # (20 B x 512 voxels), every voxel that changed written once (20 B), every packed {depth, rgba} image read once (8 B/px)
        batch_bytes = 10240.0 * stats["blocks_read"] / n_launch + 20.0 * stats["voxels_written"] / n_launch + 8.0 * W * H * frames_per_launch
        out = {
            "metric": "RGB-D frames/sec fused (640x480, 5 mm voxel TSDF)",
            "value": total_frames / dt_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": dt_max / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ImageSequenceIntegration: %d-frame synthetic 640x480 room sequence per GPU, %.4g m voxel, "
                                   "trunc 0.1 m, frames resident in HBM" % (n_local, args.voxel),
                       "frames_per_step": F, "frames_per_gpu": n_local, "sharding": "contiguous frames per GPU, one RCCL reduce at end",
                       "voxel_m": args.voxel},
            "fusion_only_frames_per_s": total_frames / t_fuse_max,
            "pool": growth,   # the pool starts at 2^18 blocks and grows on demand INSIDE the timed region (grows / replayed batches since create)
            "merge_union_blocks": n_union,
            "multi_gpu": {"ranks_in_process_group": (dist.get_world_size() if (world > 1 or force_dist) else 1),
                          "backend": (dist.get_backend() if (world > 1 or force_dist) else None),
                          "merge_bytes_per_rank": (int(n_union) * 10240 if n_union else 0), "merge_slices": (-(-int(n_union) // 32768) if n_union else 0),
                          "per_rank": per_rank,
                          "note": "weak scaling: every rank fuses its own frames without communication (fusion_ms), then ONE merge: all_gather of the block keys, "
                                  "sum-form pack, sliced reduce(SUM) to rank 0 over RCCL, normalisation (merge_ms, inside the timed region)"},
            "per_frame": {"blocks_selected": stats["blocks_selected"] / max(stats["frames"], 1),
                          "voxels_visited": stats["voxels_visited"] / max(stats["frames"], 1),
                          "voxels_updated": n_upd_frame, "final_blocks_rank0": hv.BlockCount()},
            "kernels_ms_per_launch": {"prepare_frames": prof["prepare_ms"], "select": prof["select_ms"], "integrate": prof["integrate_ms"],
                                      "event_sampled_launches": prof["launches"], "frames_per_launch": frames_per_launch},
            # The integrate kernel with full batches is bound by instruction ISSUE: `frac` is filled in below from the SQ counters of this
            # very step (tools/issue_model.py).  Until then (--no-counters, N > 1) the object carries the HBM view, which is a true
            # fraction too: the bytes a batched launch must move / launch time / 8 TB/s.
            "roofline": {"kernel": "k_integrate (Integrator::IntegrateImage, %.1f frames per launch)" % frames_per_launch,
                         "bound": "hbm", "achieved": batch_bytes / k3_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": batch_bytes / k3_s / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": prof["integrate_ms"], "traffic": None,
                         "hbm": {"bound": "hbm", "model_bytes_per_launch": batch_bytes, "model_gbs": batch_bytes / k3_s / 1e9, "peak": HBM_PEAK_GBS,
                                 "model_frac": batch_bytes / k3_s / 1e9 / HBM_PEAK_GBS,
                                 "blocks_read_per_launch": stats["blocks_read"] / n_launch, "voxels_written_per_launch": stats["voxels_written"] / n_launch,
                                 "note": "model = 10 240 B x blocks read + 20 B x voxels written + 8 B x W*H x frames per launch (counted on the device, "
                                         "op_volume_stats_launches): a LOWER bound of the launch's HBM traffic"},
                         "algorithmic_model": {"bytes_per_frame": alg_bytes_frame, "bytes_per_launch": alg_bytes, "gbs": alg_bytes / k3_s / 1e9,
                                               "ratio_to_hbm_peak": alg_bytes / k3_s / 1e9 / HBM_PEAK_GBS,
                                               "note": "SURVEY 8(d): 40 B per updated voxel per FRAME + images, / launch time.  NOT a bound for a batched launch "
                                                       "(it touches each voxel once per batch of up to 32 frames, so this ratio may exceed 1); it is one for "
                                                       "a one-frame launch: see batch1"},
                         "shader_cycles_per_launch": kc_cycles, "shader_clock_ghz": kc_cycles / k3_s / 1e9 if k3_s > 0 else None},
        }
    if rank == 0 and not args.timed_only:
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
        out["roofline"]["hbm_frac"] = None                    # measured traffic of the batched launch / time / 8 TB/s; filled in by the counter passes below
        out["roofline"]["batch1"] = {"frames": nb1, "bound": "hbm", "avg_launch_ms": p1["integrate_ms"], "algorithmic_bytes_per_launch": b1, "achieved": a1, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS, "traffic_model_bytes_per_launch": m1, "traffic_model_frac": m1 / (p1["integrate_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "evidence": "profiles/r03g_batch1.kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/prof_driver.bin ... batch=1), "
                                                 "profiles/r03g_batch1.FETCH_SIZE.pmc.csv / WRITE_SIZE.pmc.csv: 395 MB per launch measured = 0.63 of the 8 TB/s peak, the rate the "
                                                 "read-modify-write calibration kernel of the same shape reaches (profiles/r03_calib.timing.txt: 5.0 TB/s)",
                                     "note": "k_integrate with ONE frame per launch: SURVEY 8(d)'s algorithmic bytes are then a lower bound of the real traffic, "
                                             "so this is a true HBM roofline fraction (north_star: >= 50 % of HBM roofline on the integrate kernel)"}
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
                           "note": "op_volume_set_option(OP_VOLUME_OPT_UPDATE, OP_VOLUME_UPDATE_SUM_FORM): opt-in; `value` above is the default exact update"}
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
        # -- live PMC passes (separate rocprofv3 --pmc runs of the torch-free driver on a dump of this step's frames)
        if world == 1 and not args.no_counters:
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
                R["hbm"].update({"traffic_bytes_per_launch": kc["hbm_bytes_per_launch"], "read_bytes_per_launch": kc["hbm_read_bytes_per_launch"],
                                 "write_bytes_per_launch": kc["hbm_write_bytes_per_launch"], "achieved": kc["hbm_bytes_per_launch"] / k3_s / 1e9, "unit": "GB/s",
                                 "frac": kc["hbm_bytes_per_launch"] / k3_s / 1e9 / HBM_PEAK_GBS, "traffic_over_model": kc["hbm_bytes_per_launch"] / batch_bytes,
                                 "source": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950, calibrated for this kernel's 4 B/lane plane rows: profiles/r03_calib.FETCH_SIZE.pmc.csv) "
                                           "and WRITE_SIZE (exact: profiles/r03_calib.WRITE_SIZE.pmc.csv), separate passes, tools/prof_driver.bin on the first %d frames "
                                           "of this run (32-frame launches; per-launch figures scaled by %.3f to the timed region's %.1f frames per launch)" % (nfc, scale, frames_per_launch)})
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
                R.update({"bound": "issue", "achieved": im["issue_cycles_per_launch"] if additive_ok else im["classes"]["valu"]["issue_cycles"],
                          "peak": im["simd_cycles_per_launch"], "unit": "SIMD issue cycles per launch (shader clock)",
                          "frac": im["frac"] if additive_ok else valu_frac,
                          "frac_definition": ("sum over all instruction classes x measured issue cost / SIMD cycles (additive model, validated on a mixed-class "
                                              "microbenchmark: predicted / measured = %.3f)" % mc["additive_over_measured"]) if additive_ok else
                                             "VALU wave-instructions x measured issue cost / SIMD cycles (the additive all-class model is NOT validated%s: scalar work co-issues)"
                                             % ((": it predicts %.2f x the mixed microbenchmark's time" % mc["additive_over_measured"]) if mc else ""),
                          "valu_frac": valu_frac, "all_classes_additive_frac": im["frac"], "mixed_check": mc, "costs_file": os.path.relpath(costs_file, ROOT)})
                R["hbm_frac"] = R["hbm"]["frac"]      # measured HBM traffic of the batched launch / launch time / 8 TB/s
                R["issue"] = {"classes": im["classes"], "valu_cycles_each": im["valu_cycles_each"], "kernel_cycles": im["kernel_cycles"],
                              "insts_per_voxel_frame_wave": {"valu": counts["SQ_INSTS_VALU"] / (stats["blocks_selected"] / max(stats["frames"], 1) * frames_per_launch * 8.0),
                                                             "salu": counts.get("SQ_INSTS_SALU", 0.0) / (stats["blocks_selected"] / max(stats["frames"], 1) * frames_per_launch * 8.0)},
                              "costs": "profiles/r03_issue_costs.json (tools/valu_ubench.hip at 8 waves per SIMD, shader cycles per wave64 instruction and SIMD; "
                                       "VALU classes weighted by the kernel's static opcode histogram)",
                              "note": "frac = sum over classes of wave-instructions (SQ_INSTS_* of this step's launches) x issue cost / (1024 SIMDs x the launch's "
                                      "shader cycles): the share of the chip's instruction-issue slots the kernel fills.  HBM is far from binding for a batched launch "
                                      "(roofline.hbm), so the way to make it faster is fewer instructions per voxel and frame"}
                R["counters"] = {k: {c: v["mean_per_launch"] for c, v in r.items() if isinstance(v, dict) and "mean_per_launch" in v} for k, r in cnt.items()}
            except Exception as e:  # rocprofv3 missing / failing must not take the bench line down
                out["roofline"]["traffic_error"] = repr(e)[:300]


    # ---- the reference's own call pattern: one CubeHandler::IntegrateImage(cv::Mat depth, cv::Mat rgb, pose) per frame with
    # PAGEABLE host images (CubeHandler.cpp:197-210).  PCIe-inclusive, never the headline `value`: each call copies its two
    # images into the pinned staging ring (caller thread + 2 helper threads), the DMA runs on a copy stream and overlaps the
    # previous batch's kernels, frames are fused up to 32 per launch group.  (From C++ -- tools/prof_driver.cpp "host" -- the same
    # loop reaches ~13 k frames/s; here the Python interpreter sits in the loop.)
    if rank == 0 and world == 1 and not args.timed_only:
        nh = min(300, n_local)
        dn, cn = depth[:nh].cpu().numpy(), rgb[:nh].cpu().numpy()
        d16h = np.clip(np.round(dn * 1000.0), 0, 65535).astype(np.uint16)
        rates = {}
        for name, dsrc in (("float32_depth", dn), ("uint16_depth", d16h)):
            best = None
            for rep in range(3):
                hv.Clear(); hv.Synchronize()
                t = time.perf_counter()
                for k in range(nh):
                    hv.IntegrateImage(dsrc[k], cn[k], poses[k])
                hv.Synchronize()
                dth = time.perf_counter() - t
                best = dth if best is None else min(best, dth)
            rates[name] = nh / best
        out["host_images_frames_per_s"] = rates["float32_depth"]
        out["host_images"] = {"frames": nh, "float32_depth_frames_per_s": rates["float32_depth"], "uint16_depth_frames_per_s": rates["uint16_depth"],
                              "call_pattern": "one IntegrateImage(depth, rgb, pose) per frame, pageable numpy buffers, Python loop; pinned staging ring + copy stream"}
        # (a) the same loop from C++ (tools/prof_driver.bin host): no interpreter between the calls -- the reference's actual call pattern
        try:
            import subprocess, tempfile, re as _re
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import counters as CT
            CT.build_driver()
            with tempfile.NamedTemporaryFile(prefix="opc_host_", suffix=".bin", dir="/tmp", delete=False) as tf:
                np.array([nh, W, H], np.int32).tofile(tf)
                for i in range(nh):
                    poses[i].astype(np.float32).tofile(tf); dn[i].tofile(tf); cn[i].tofile(tf)
                hname = tf.name
            try:
                txt = subprocess.run([CT.DRIVER, hname, "3", repr(float(args.voxel)), "host"], capture_output=True, text=True, timeout=300).stdout
            finally:
                os.unlink(hname)
            best_cpp = {}
            for m in _re.finditer(r"host images, (float32|uint16) depth: \d+ frames, ([\d.]+) frames/s", txt):
                best_cpp[m.group(1)] = max(best_cpp.get(m.group(1), 0.0), float(m.group(2)))
            out["host_images"]["cpp_float32_depth_frames_per_s"] = best_cpp.get("float32")
            out["host_images"]["cpp_uint16_depth_frames_per_s"] = best_cpp.get("uint16")
            out["host_images"]["cpp_driver"] = "tools/prof_driver.bin <frames> 3 <voxel> host: one op_volume_integrate per frame from C++ with pageable images, best of 3"
        except Exception as e:
            out["host_images"]["cpp_error"] = repr(e)[:200]
        del dn, cn, d16h

    # ---- fusing into a volume that was NOT written by the integrate kernel alone (after SetCubeMap / ReadFromFile / Merge -- the reference's
    # MergeMultipleSubmaps / FBAFusion pattern): the update then takes the general form (two branches, four true divisions per voxel)
    if rank == 0 and world == 1 and not args.timed_only:
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

    # ---- the same K steps behind the drivers' depth front end (tool::ConvertDepthTo32F + tool::BilateralFilter,
    # ImageSequenceIntegration.cpp:36-38) from raw 16-bit depth, filter enqueued on the volume's stream.  Supplementary:
    # the filter is OpenCV's in the reference (unpinned), so the headline `value` above stays without it.
    if rank == 0 and world == 1 and not args.timed_only:
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

    # ---- CPU baseline + parity on a bounded sample of the SAME frames (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
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

    # ---- ICP iterations/s (second half of BASELINE.json's metric; configs[1]); replicas only, rank 0 reports
    if rank == 0 and not args.no_icp:
        from onepiece_amd import registration as R
        import ctypes as C
        from onepiece_amd import _lib as L
        lib = L.load()
        cam = hv.camera
        d0, d1 = depth[0].cpu().numpy(), depth[1].cpu().numpy()
        tgt_pc = R.PointCloud.LoadFromDepth(d0, cam, device=local_rank)
        src = R.PointCloud.LoadFromDepth(d1, cam, device=local_rank).points
        tgt_pc.EstimateNormals(0.1, 30, device=local_rank)  # warm-up (ICPTest.cpp:24: EstimateNormals before PointToPlane)
        t = time.perf_counter()
        tgt_pc.EstimateNormals(0.1, 30, device=local_rank)
        normals_s = time.perf_counter() - t
        tgt, nrm = tgt_pc.points, tgt_pc.normals
        h = C.c_void_p()
        L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), 0.01, L.OP_MEM_HOST, local_rank, C.byref(h)))
        L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
        res = L.IcpResult()
        T0 = np.eye(4, dtype=np.float32).reshape(16)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        L.check(lib.op_icp_run(h, 1, fp(T0), 5, C.byref(res), None, 0, None, None))  # warm
        iters = 60
        t = time.perf_counter()
        L.check(lib.op_icp_run(h, 1, fp(T0), iters, C.byref(res), None, 0, None, None))
        gpu_it_s = iters / (time.perf_counter() - t)
        # the same call with the order-free fp64 finish: what is left is the iteration loop itself (the default finish --
        # the reference's sequential float32 Kabsch over ~3e5 pairs on one host thread -- is ~0.8 ms per CALL, not per iteration)
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_FINISH, L.OP_ICP_FINISH_FP64))
        res64 = L.IcpResult()
        t = time.perf_counter()
        L.check(lib.op_icp_run(h, 1, fp(T0), iters, C.byref(res64), None, 0, None, None))
        loop_it_s = iters / (time.perf_counter() - t)
        # the price of exactness: the validation mode that sums every iteration's rows in the reference's sequential float32 order on the
        # host (identical per-iteration inlier counts and pairs at this size, tests/test_icp_gpu.py)
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_FINISH, L.OP_ICP_FINISH_REFERENCE))
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_SUMS, L.OP_ICP_SUMS_REFERENCE_F32))
        res_ref = L.IcpResult()
        it_ref = 20
        t = time.perf_counter()
        L.check(lib.op_icp_run(h, 1, fp(T0), it_ref, C.byref(res_ref), None, 0, None, None))
        ref_it_s = it_ref / (time.perf_counter() - t)
        lib.op_icp_destroy(h)
        # what a caller of registration::PointToPlane pays: the one-shot entry point builds the search grid, uploads both
        # clouds, runs ICPTest's 30 iterations, forms the reference-order result and drops everything again
        reg_ms = []
        for _ in range(6):
            r1 = L.IcpResult()
            t = time.perf_counter()
            L.check(lib.op_icp_register(1, fp(src.reshape(-1)), len(src), fp(tgt.reshape(-1)), fp(nrm.reshape(-1)), len(tgt), fp(T0), 30, 0.01, local_rank,
                                        C.byref(r1), None, 0))
            reg_ms.append((time.perf_counter() - t) * 1e3)
        out["icp"] = {"iters_per_s": gpu_it_s, "loop_only_iters_per_s": loop_it_s, "reference_order_sums_iters_per_s": ref_it_s, "iterations_per_call": iters, "points": int(len(src)),
                      "mode": "point-to-plane, threshold 0.01 (ICPTest.cpp:31)",
                      "final_inliers": int(res.n_inliers), "estimate_normals_s": normals_s,
                      "register_call_ms": float(np.median(reg_ms[1:])), "register_call_iterations": 30,
                      # 36 B per source point per iteration (source + matched target + normal); the kernel is
                      # a latency-bound gather (27-cell scan), so this is far from the HBM roof by construction
                      "algorithmic_gbs": 36.0 * len(src) * gpu_it_s / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            t = time.perf_counter()
            ref = O.icp(src, tgt, nrm, None, 10, 0.01, True)
            cpu_it_s = 10 / (time.perf_counter() - t)
            out["cpu_baseline"]["icp_iters_per_s"] = cpu_it_s
            out["cpu_baseline"]["icp_threads"] = os.cpu_count()
            t = time.perf_counter()
            O.estimate_normals(tgt, 0.1, 30)
            out["cpu_baseline"]["estimate_normals_s"] = time.perf_counter() - t
            # pose parity of the timed configuration (same clouds, same normals, 10 iterations)
            chk = L.IcpResult()
            h2 = C.c_void_p()
            L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), 0.01, L.OP_MEM_HOST, local_rank, C.byref(h2)))
            L.check(lib.op_icp_set_source(h2, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
            L.check(lib.op_icp_run(h2, 1, fp(T0), 10, C.byref(chk), None, 0, None, None))
            chk64 = L.IcpResult()
            L.check(lib.op_icp_set_option(h2, L.OP_ICP_OPT_FINISH, L.OP_ICP_FINISH_FP64))
            L.check(lib.op_icp_run(h2, 1, fp(T0), 10, C.byref(chk64), None, 0, None, None))
            lib.op_icp_destroy(h2)
            g64 = np.array(chk64.T, np.float64).reshape(4, 4)
            g = np.array(chk.T, np.float64).reshape(4, 4)
            gl = np.array(chk.last_T, np.float64).reshape(4, 4)
            rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
            # float64 Kabsch over the CPU path's final inlier pairs = the exact answer both paths approximate
            ps, pt = src[ref["pairs"][:, 0]].astype(np.float64), tgt[ref["pairs"][:, 1]].astype(np.float64)
            ms, mt = ps.mean(0), pt.mean(0)
            U, _sv, Vt = np.linalg.svd((ps - ms).T @ (pt - mt))
            Rm = Vt.T @ U.T
            if np.linalg.det(Rm) < 0:
                Vt[2] *= -1
                Rm = Vt.T @ U.T
            T64 = np.eye(4); T64[:3, :3] = Rm; T64[:3, 3] = mt - Rm @ ms
            out["icp"]["parity_10_iterations"] = {
                "accumulated_pose_rel_err_vs_cpu": rel(gl, ref["last_T"]),
                "returned_T_rel_err_vs_cpu": rel(g, ref["T"]),
                "returned_T_rel_err_vs_float64_kabsch": {"gpu": rel(g, T64), "gpu_fp64_finish": rel(g64, T64), "cpu": rel(ref["T"], T64)},
                "inliers": {"gpu": int(chk.n_inliers), "cpu": int(len(ref["pairs"]))},
                "note": "RegistrationResult::T is a Kabsch fit whose sums the reference accumulates sequentially in float32 over ~3e5 "
                        "near-planar pairs (Geometry.cpp:117-133).  The default finish (OP_ICP_FINISH_REFERENCE) reproduces that order on "
                        "the compacted inlier pairs, so returned_T agrees with the CPU path; gpu_fp64_finish is the order-free variant"}
            out["icp"]["note"] = "cpu oracle (kd-tree NN, OpenMP over %d threads) timed on the same clouds" % os.cpu_count()

    # ---- dense RGB-D tracking (Odometry::DenseTracking's coarse-to-fine loop); rank 0 reports
    if rank == 0 and not args.no_tracking:
        import ctypes as C
        from onepiece_amd import odometry as OD, _lib as L
        lib = L.load()
        odo = OD.Odometry(hv.camera, device=local_rank)
        # frame 1 -> frame 0 of this rank's shard.  (a) from the raw frames, end to end (op_tracker_dense_tracking);
        # (b) the loop alone on the pyramids (a) built, resident in HBM (boundary = MultiScaleComputing's inputs)
        full = lambda: odo.DenseTracking(rgb[1], rgb[0], depth[1], depth[0], None, 0, want_correspondences=False)
        for _ in range(3):
            full()
        n_full = 100
        t = time.perf_counter()
        for _ in range(n_full):
            fres = full()
        full_s = n_full / (time.perf_counter() - t)
        levels = odo.PreparedLevels()
        dev_levels = []
        for lv in levels:
            d = dict(lv)
            for k in OD.TRACK_IMAGES:
                d[k] = torch.from_numpy(np.ascontiguousarray(lv[k])).to(dev)
            dev_levels.append(d)
        arr, mem, _keep = OD._levels_arg(dev_levels)
        it3 = np.array(odo.iter_count_per_level, np.int32)
        T0 = np.eye(4, dtype=np.float32).reshape(16)
        tres = L.TrackResult()
        run = lambda: L.check(lib.op_tracker_track(odo._h, arr, 3, it3.ctypes.data_as(L._ip), W, H, 0, T0.ctypes.data_as(L._fp), mem,
                                                   C.byref(tres), None, None, 0, None, None))
        for _ in range(5):
            run()
        n_tr = 100
        t = time.perf_counter()
        for _ in range(n_tr):
            run()
        tr_s = n_tr / (time.perf_counter() - t)
        odo.SetSums("reference_f32")       # the reference's sums: rows of every iteration summed sequentially in float32 in raster order, by one wave on the device
        for _ in range(2):
            run()
        t = time.perf_counter()
        for _ in range(20):
            run()
        tr_ref_s = 20 / (time.perf_counter() - t)
        odo.SetSums("reference_f32_host")  # the same sums on one host thread (all rows cross PCIe every iteration): the cross-check variant
        run()
        t = time.perf_counter()
        for _ in range(5):
            run()
        tr_ref_host_s = 5 / (time.perf_counter() - t)
        odo.SetSums("fp64")
        out["tracking"] = {"tracks_per_s": tr_s, "ms_per_track": 1e3 / tr_s, "reference_order_tracks_per_s": tr_ref_s, "reference_order_host_sums_tracks_per_s": tr_ref_host_s,
                           "from_raw_frames_tracks_per_s": full_s, "levels": 3, "iters_per_level": [4, 8, 16],
                           "iterations_executed": int(tres.iterations), "term": "hybrid", "resolution": [W, H],
                           "correspondences": int(tres.n_correspondences), "tracking_success": bool(tres.tracking_success),
                           "input": "pyramids resident in HBM (boundary = Odometry::MultiScaleComputing inputs)"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            O.dense_track(levels, (4, 8, 16), term=0)
            t = time.perf_counter()
            for _ in range(5):
                ref = O.dense_track(levels, (4, 8, 16), term=0)
            out["cpu_baseline"]["tracks_per_s"] = 5 / (time.perf_counter() - t)
            O.lib().orc_set_accumulate_double(1)
            ref_d = O.dense_track(levels, (4, 8, 16), term=0)
            O.lib().orc_set_accumulate_double(0)
            rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
            g = np.array(tres.T, np.float64).reshape(4, 4)
            out["tracking"]["parity"] = {"pose_rel_err_vs_cpu": rel(g, ref["T"]), "pose_rel_err_vs_cpu_double_sums": rel(g, ref_d["T"]),
                                         "cpu_float_vs_double_sums": rel(ref["T"], ref_d["T"]),
                                         "iterations": {"gpu": int(tres.iterations), "cpu": int(ref["iterations"])},
                                         "correspondences": {"gpu": int(tres.n_correspondences), "cpu": int(len(ref["pixel_correspondences"]))}}
        del odo

    # ---- config 4 (BASELINE configs[3]): tracking + fusion, frames resident in HBM; rank 0 reports
    if rank == 0 and not args.no_tracking:
        from onepiece_amd import dense_slam as DS
        n_df = min(100, n_local)

        def dense_fusion_pass(pipe):
            vol = I.CubeHandler(hv.camera, device=local_rank)
            vol.SetVoxelResolution(0.005)
            slam = DS.DenseSlam(hv.camera, device=local_rank, pipeline=pipe,
                                on_tracked=lambda fid, c, d, T: vol.IntegrateImage(d, c, T))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n_df):
                slam.UpdateFrame(rgb[i], depth[i])
            slam.Finish()
            nb = vol.BlockCount()           # flushes the pending batch and synchronises
            return slam, nb, time.perf_counter() - t0

        dense_fusion_pass(4)                # warm-up
        _s1, _nb1, dt_seq = dense_fusion_pass(1)
        slam, nb, dt = dense_fusion_pass(4)
        if world == 1 and not args.no_cpu_baseline:
            # the same pipeline on one host core (the reference's tracker and integrator are serial): 4 frames fused, then tracking alone
            # over a 24-frame prefix for the pose-chain parity
            from oracle import oracle as O
            ocam = O.make_camera()
            ovol = O.Volume(ocam, voxel_res=0.005)
            n_par = min(24, n_df)
            hd, hc = depth[:n_par].cpu().numpy(), rgb[:n_par].cpu().numpy()
            t0 = time.perf_counter()
            gp = np.eye(4, dtype=np.float32)
            ovol.integrate(hd[0], hc[0], gp)
            ref_chain = [gp]
            for i in range(1, 4):
                r = O.dense_tracking(ocam, hc[i - 1], hc[i], hd[i - 1], hd[i], (4, 8, 16), 0)
                gp = DS._mat4_mul_f32(gp, O.mat4_inverse(r["T"]))
                ref_chain.append(gp)
                ovol.integrate(hd[i], hc[i], gp)
            out["cpu_baseline"]["dense_fusion_frames_per_s"] = 4 / (time.perf_counter() - t0)
            for i in range(4, n_par):
                r = O.dense_tracking(ocam, hc[i - 1], hc[i], hd[i - 1], hd[i], (4, 8, 16), 0)
                ref_chain.append(DS._mat4_mul_f32(ref_chain[-1], O.mat4_inverse(r["T"])))
            rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
            pair = lambda chain, i: np.linalg.inv(np.asarray(chain[i - 1], np.float64)) @ np.asarray(chain[i], np.float64)
            g0p = np.linalg.inv(poses[0].astype(np.float64))
            drift_of = lambda chain: [float(np.abs(np.asarray(chain[i], np.float64) - g0p @ poses[i].astype(np.float64))[:3, 3].max()) for i in range(n_par)]
            par = {"frames": n_par, "oracle_chain_max_translation_drift_m": max(drift_of(ref_chain))}
            for mode_name in ("fp64", "reference_f32"):
                chk = DS.DenseSlam(hv.camera, device=local_rank)
                chk.rgbd_odometry.SetSums(mode_name)
                chk.UpdateFrame(rgb[0], depth[0]); chk.UpdateFrame(rgb[1], depth[1])      # (first call: workspace allocation)
                torch.cuda.synchronize(dev)
                t_par = time.perf_counter()
                for i in range(2, n_par):
                    chk.UpdateFrame(rgb[i], depth[i])
                t_par = time.perf_counter() - t_par
                pe = [rel(pair(chk.global_poses, i), pair(ref_chain, i)) for i in range(1, n_par)]
                ce = [rel(chk.global_poses[i], ref_chain[i]) for i in range(n_par)]
                par[mode_name] = {"pair_rel_err_max": max(pe), "pair_rel_err_median": float(np.median(pe)), "pairs_within_1e-4": int(sum(e <= 1e-4 for e in pe)),
                                  "pairs": len(pe), "chain_rel_err_max": max(ce), "max_translation_drift_m": max(drift_of(chk.global_poses)),
                                  "tracks_per_s": (n_par - 2) / t_par}   # one pair at a time, from raw frames (image preparation included)
            out["dense_fusion_parity"] = par
        g0 = np.linalg.inv(poses[0].astype(np.float64))
        drift = max(float(np.abs(np.asarray(slam.global_poses[i], np.float64) - g0 @ poses[i].astype(np.float64))[:3, 3].max())
                    for i in range(n_df))
        par_summary = None
        if "dense_fusion_parity" in out:   # the pose error of the mode the rates below are quoted in, next to them (north_star's bar: 1e-4 relative)
            pf, pr = out["dense_fusion_parity"]["fp64"], out["dense_fusion_parity"]["reference_f32"]
            par_summary = {"default_mode_fp64": {"pair_rel_err_max_vs_cpu": pf["pair_rel_err_max"], "pairs_within_1e-4": pf["pairs_within_1e-4"], "pairs": pf["pairs"],
                                                 "meets_1e-4_on_every_pair": pf["pairs_within_1e-4"] == pf["pairs"]},
                           "reference_order_f32": {"pair_rel_err_max_vs_cpu": pr["pair_rel_err_max"], "pairs_within_1e-4": pr["pairs_within_1e-4"], "pairs": pr["pairs"],
                                                   "meets_1e-4_on_every_pair": pr["pairs_within_1e-4"] == pr["pairs"], "tracks_per_s": pr["tracks_per_s"]},
                           "note": "the rates of this object are the DEFAULT mode's (fp64 reduction, no host round trip); the reference's own float32 summation order "
                                   "(OP_TRACK_SUMS_REFERENCE_F32, sums by one wave on the device) follows the CPU path step for step at tracks_per_s"}
        out["dense_fusion"] = {"pose_parity": par_summary, "frames_per_s": n_df / dt, "one_pair_at_a_time_frames_per_s": n_df / dt_seq, "pairs_in_flight": 4,
                               "frames": n_df, "tracked": int(sum(slam.tracking_success)),
                               "blocks": int(nb), "voxel_m": 0.005, "max_translation_drift_m": drift,
                               "pipeline": "per frame: Odometry::DenseTracking(prev, cur, I) on the GPU (image preparation, 3 levels x "
                                           "{4,8,16}), pose chaining on the host, CubeHandler::IntegrateImage with the TRACKED pose; "
                                           "no submap registration / BA (out of scope).  pairs_in_flight independent frame pairs are tracked "
                                           "concurrently on separate HIP streams (speculating on the success flag, resolved in order): "
                                           "identical poses, the latency-bound tracker no longer leaves the chip idle"}
        # the same pipeline from C++ over the C-ABI (tools/prof_driver.bin track=4: op_tracker_dense_tracking_enqueue / op_tracker_wait on four
        # trackers, op_volume_integrate with the chained pose): the interpreter's ~250 us per frame are what limits the figure above
        if world == 1:
            try:
                import subprocess, tempfile, re as _re
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import counters as CT
                CT.build_driver()
                n_cpp = min(200, n_local)
                with tempfile.NamedTemporaryFile(prefix="opc_track_", suffix=".bin", dir="/tmp", delete=False) as tf:
                    np.array([n_cpp, W, H], np.int32).tofile(tf)
                    dh, ch = depth[:n_cpp].cpu().numpy(), rgb[:n_cpp].cpu().numpy()
                    for i in range(n_cpp):
                        poses[i].astype(np.float32).tofile(tf); dh[i].tofile(tf); ch[i].tofile(tf)
                    tname = tf.name
                try:
                    rates_cpp = {}
                    for k in (1, 4):
                        txt = subprocess.run([CT.DRIVER, tname, "3", "0.005", "track=%d" % k], capture_output=True, text=True, timeout=300).stdout
                        m = _re.findall(r"tracked (\d+)/(\d+) frames, ([\d.]+) frames/s", txt)
                        rates_cpp[k] = (max(float(x[2]) for x in m), int(m[-1][0]), int(m[-1][1])) if m else None
                finally:
                    os.unlink(tname)
                if rates_cpp.get(4):
                    out["dense_fusion"].update({"cpp_frames_per_s": rates_cpp[4][0], "cpp_one_pair_at_a_time_frames_per_s": rates_cpp[1][0] if rates_cpp.get(1) else None,
                                                "cpp_tracked": rates_cpp[4][1], "cpp_frames": rates_cpp[4][2],
                                                "cpp_driver": "tools/prof_driver.bin <frames> 3 0.005 track=4: the same pipeline over the C-ABI without the interpreter, "
                                                              "best of 3; every tracker stream has a hardware queue of its own (the library asks for 8 when it is loaded)"})
            except Exception as e:
                out["dense_fusion"]["cpp_error"] = repr(e)[:200]

    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
