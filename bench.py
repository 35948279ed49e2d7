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
    ap.add_argument("--merge-algorithm", choices=["owner", "dense"], default="owner",
                    help="the N > 1 merge: the owner-partitioned exchange (default) or the dense reduce of the whole union (rounds 1-4)")
    ap.add_argument("--timed-only", action="store_true", help="only the warm-up and the timed region (for rocprofv3 --kernel-trace --stats runs: "
                    "every k_integrate launch in the trace then has the timed region's batch shape)")
    ap.add_argument("--summaries", action="store_true", help="A/B aid: raycast the empty volume once before the warm-up, so that the raycaster's block summaries exist and k_integrate "
                                                              "keeps them current inside the timed region (what a tracking-against-the-model pipeline does)")
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

    if args.summaries:
        hv.Raycast(poses[0])
    # ---- warmup: W untimed steps (+ one merge so RCCL is initialised), then start from empty
    for w in range(Wm):
        s = (w % K) * F
        hv.IntegrateSequence(depth[s:s + F], rgb[s:s + F], poses[s:s + F])
    hv.Synchronize()
    if world > 1 or force_dist:
        D.merge_volumes(ops, root=0, algorithm=args.merge_algorithm)
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
        n_union = D.merge_volumes(ops, root=0, algorithm=args.merge_algorithm)
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
    if world > 1 or force_dist:   # what this rank put on the wire (onepiece_amd.distributed.last_stats)
        mine.update({k: D.last_stats.get(k) for k in ("algorithm", "held_blocks", "owned_blocks", "wire_bytes_sent", "wire_bytes_received")})
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
                       "voxel_m": args.voxel,
                       "update_mode": "exact: `value` is measured with the default update, whose voxels are bit-identical to the reference's CPU path; the opt-in sum form "
                                      "(within 2e-6 of it, two decades inside north_star's 1e-4) is faster and reported under sum_form -- both answer north_star, only this one bit for bit"},
            "fusion_only_frames_per_s": total_frames / t_fuse_max,
            "pool": growth,   # the pool starts at 2^18 blocks and grows on demand INSIDE the timed region (grows / replayed batches since create)
            "merge_union_blocks": n_union,
            "multi_gpu": {"ranks_in_process_group": (dist.get_world_size() if (world > 1 or force_dist) else 1),
                          "backend": (dist.get_backend() if (world > 1 or force_dist) else None),
                          "merge_algorithm": per_rank[0].get("algorithm") if (world > 1 or force_dist) else None,
                          "dense_reduce_bytes_per_rank_for_comparison": (int(n_union) * 10240 if n_union else 0),
                          "wire_bytes_sent_per_rank": [p.get("wire_bytes_sent") for p in per_rank] if (world > 1 or force_dist) else None,
                          "per_rank": per_rank,
                          "note": "weak scaling: every rank fuses its own frames without communication (fusion_ms), then ONE merge (merge_ms, inside the timed region): the "
                                  "owner-partitioned exchange -- every rank sends the blocks it HOLDS, in sum form, to their owner ranks (all pairs at once over xGMI's "
                                  "point-to-point links), the owners add them up and send their partitions to rank 0, which normalises the whole map (the reference's Merge "
                                  "semantics).  wire_bytes_sent = (held blocks of other ranks' partitions + the rank's own summed partition) x 10 248 B; the dense "
                                  "reduce of rounds 1-4 put union x 10 240 B on every rank's link"},
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
    if rank == 0:
        # ---- everything below is supplementary (rank 0; most of it N = 1 only): one module per section under benchparts/
        import types
        from benchparts import roofline_extras, raycast, volume_ops, host_images, general_update, depth_filter, cpu_baseline, icp, tracking, dense_fusion
        c = types.SimpleNamespace(args=args, torch=torch, dev=dev, rank=rank, world=world, local_rank=local_rank, hv=hv, depth=depth, rgb=rgb, poses=poses, K=K, F=F,
                                  n_local=n_local, I=I, S=S, ROOT=ROOT, W=W, H=H, HBM_PEAK_GBS=HBM_PEAK_GBS, stats=stats, prof=prof,
                                  frames_per_launch=frames_per_launch, k3_s=k3_s, kc_cycles=kc_cycles, batch_bytes=batch_bytes)
        single = world == 1
        c.oracle = None
        if single and not args.no_cpu_baseline:
            from oracle import oracle as _oracle   # the cpu_baseline leg: the CPU oracle as the timed baseline and the parity checker, never on the measured path
            c.oracle = _oracle
        if single and not args.timed_only:
            raycast.run(c, out)            # first: the volume still holds the timed region's frames
        if not args.timed_only:
            roofline_extras.run(c, out)
        if single and not args.timed_only:
            volume_ops.run(c, out)
            host_images.run(c, out)
            general_update.run(c, out)
            depth_filter.run(c, out)
        if single and not args.no_cpu_baseline:
            cpu_baseline.run(c, out)
        if not args.no_icp:
            icp.run(c, out)
        if not args.no_tracking:
            tracking.run(c, out)
            dense_fusion.run(c, out)


    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
