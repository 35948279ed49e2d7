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

The LAST stdout line is a compact JSON object (< 8 KB, tests/test_bench_line_cpu.py): the contract's keys plus roofline (the integrate
kernel, HIP-event timed on the volume's own stream; frac = HBM bytes per launch / launch time / 8 TB/s), cpu_baseline (the CPU oracle =
port of the reference path, timed on this box's host cores on a bounded sample of the same frames; N=1, rank 0 only), parity (the GPU
volume for that sample compared bit-for-bit with the oracle's), icp (iterations/s at 307 200 points, the figure of the in-tolerance
mode next to the default's), tracking, multi_gpu.  Everything else the sections measure goes to bench_detail.json next to this file
(and to gpurun_out/ when that directory exists); `--full` adds the supplementary sections (raycast, volume ops, host images, sum form,
depth filter), `--counters` re-measures the HBM counters with rocprofv3 instead of taking the committed ones (profiles/).

N > 1: the merge inside the timed region is the PRODUCT's -- op_volume_merge_rccl (csrc/merge_rccl.hip) on an ncclComm_t this script
makes (ncclGetUniqueId on rank 0, carried over the process group, ncclCommInitRank); `--merge-impl torch` selects the torch.distributed
mirror of the same algorithm instead (onepiece_amd/distributed.py; what the gloo test hooks use).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# Kernels of different HIP streams only run side by side when the streams sit on different HARDWARE queues, and the runtime maps all streams onto
# GPU_MAX_HW_QUEUES of them (default 4; tools/queue_probe.hip: K one-workgroup kernels on K streams take ceil(K / queues) kernel times).  The tracking + fusion
# pipeline (a tracker stream per pair in flight) and the ICP replicas (a stream per context) want one each: 16 (profiles/r06_track_hw_queues.txt: 349 -> 489
# frames/s with 16 pairs in flight; with 32 the rates of this process collapse).  Read once, when the HIP runtime initialises -- hence here, before anything
# touches the GPU.  No effect on the fusion figure.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: without it RCCL fails with `hipIpcGetMemHandle: invalid argument`); the
# boxes export it already -- this only covers a shell that does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    ap.add_argument("--merge-impl", choices=["auto", "cabi", "torch"], default="auto",
                    help="the N > 1 merge: cabi = op_volume_merge_rccl on a communicator made here (the product's path; auto picks it on RCCL), "
                         "torch = the torch.distributed mirror of the same algorithm (auto picks it on the gloo test hook)")
    ap.add_argument("--full", action="store_true", help="also run the supplementary sections (raycast, volume ops, host images, general update, depth filter, sum form): "
                                                         "they go to bench_detail.json, never to the result line")
    ap.add_argument("--counters", action="store_true", help="re-measure roofline.traffic with rocprofv3 PMC passes (~30 s) instead of scaling the committed measurement "
                                                             "(profiles/r06_integrate_counters.json) to this run's launch shape")
    ap.add_argument("--save-counters", action="store_true", help="with --counters: rewrite profiles/r06_integrate_counters.json from this run")
    ap.add_argument("--no-counters", action="store_true", help=argparse.SUPPRESS)   # rounds 3-5: the passes used to be on by default
    ap.add_argument("--detail-file", default=None, help="where the full (uncompacted) object goes (default: bench_detail.json next to bench.py)")
    ap.add_argument("--profile-every", type=int, default=1, help="HIP-event sample rate for the roofline (every k-th launch group)")
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu_baseline = args.no_icp = args.no_tracking = True
        args.counters = args.full = False

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

    distributed = world > 1 or force_dist
    merge_impl, merge_fallback, comm = None, None, None
    # Test hook (not used by the driver): ONEPIECE_BENCH_RCCL_LIBRARY names the RCCL the communicator is made in and the merge binds -- the multi-process double
    # (tests/cpp/librccl_double_mp.so) lets N ranks run the LIBRARY's merge on a one-GPU box, next to ONEPIECE_BENCH_SINGLE_DEVICE=1 / ONEPIECE_BENCH_BACKEND=gloo.
    rccl_library = os.environ.get("ONEPIECE_BENCH_RCCL_LIBRARY") or None
    if distributed:
        merge_impl = args.merge_impl if args.merge_impl != "auto" else ("cabi" if (dist.get_backend() == "nccl" or rccl_library) else "torch")
    if merge_impl == "cabi":
        # the product's communicator: ncclGetUniqueId on rank 0 -> the process group carries the 128 bytes -> ncclCommInitRank (torch's own RCCL, the one the merge binds)
        comm = D.RcclCommunicator(rank, world, D.torch_exchange(dist), library=rccl_library)
        assert comm.count() == world

    def merge():
        if merge_impl == "cabi":
            return D.merge_volumes_rccl(hv, comm, root=0, algorithm=args.merge_algorithm, force_single_rank=force_dist)
        return D.merge_volumes(ops, root=0, algorithm=args.merge_algorithm)

    if args.summaries:
        hv.Raycast(poses[0])
    # ---- warmup: W untimed steps (+ one merge so RCCL is initialised), then start from empty
    for w in range(Wm):
        s = (w % K) * F
        hv.IntegrateSequence(depth[s:s + F], rgb[s:s + F], poses[s:s + F])
    hv.Synchronize()
    if distributed:
        # The warm-up merge must see what the timed one will: its temporaries (a rank's packed blocks, the receive buffers, the root's gathered map: GBs) come from
        # the library's buffer cache, and a cached buffer only serves requests of its size class -- a merge of W steps' blocks would leave the timed merge to
        # hipMalloc its buffers inside the timed region (allocations of that size have taken 0.4 - 1.5 s on this pool, profiles/r05_transform_pool.txt).  So the
        # shard is fused completely (K steps, ~25 ms per 1000 frames, untimed) before the warm-up merge.
        hv.Clear()
        hv.IntegrateSequence(depth[:n_local], rgb[:n_local], poses[:n_local])
        hv.Synchronize()
        err = None
        try:
            merge()
        except Exception as e:   # the library's merge returns its error on EVERY rank (agreement points): all ranks land here together
            if merge_impl != "cabi":
                raise
            err = repr(e)[:300]
        if merge_impl == "cabi":
            # one agreement over the process group: if the C-ABI merge failed anywhere, every rank takes the mirror -- loudly (multi_gpu.merge_fallback)
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                merge_fallback = err or "op_volume_merge_rccl failed on another rank"
                print("bench.py: op_volume_merge_rccl failed in the warm-up (%s): falling back to the torch.distributed mirror" % merge_fallback, file=sys.stderr, flush=True)
                merge_impl = "torch"
                hv.Clear()
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
    if distributed:
        n_union = merge()
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
    if distributed:   # what this rank put on the wire (onepiece_amd.distributed.last_stats: op_merge_stats of the library call, or the mirror's own count)
        mine.update({k: D.last_stats.get(k) for k in ("algorithm", "held_blocks", "owned_blocks", "wire_bytes_sent", "wire_bytes_received", "prepare_ms", "transfer_ms")})
        mine["rccl_ranks"] = comm.count() if comm is not None else None
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
                       "frames_per_step": F, "frames_per_gpu": n_local, "sharding": "contiguous frames per GPU, one RCCL merge at end",
                       "voxel_m": args.voxel, "update_mode": "exact (voxels bit-identical to the reference's CPU path)"},
            "fusion_only_frames_per_s": total_frames / t_fuse_max,
            "pool": growth,   # the pool starts at 2^18 blocks and grows on demand INSIDE the timed region (grows / replayed batches since create)
            "merge_union_blocks": n_union,
            "multi_gpu": {"ranks": world, "ranks_in_process_group": (dist.get_world_size() if distributed else 1),
                          "backend": (dist.get_backend() if distributed else None),
                          # which code merged: "cabi" = op_volume_merge_rccl (csrc/merge_rccl.hip) on this script's own ncclComm_t, "torch" = the torch.distributed mirror
                          "merge_impl": merge_impl, "merge_fallback": merge_fallback,
                          "rccl_ranks": per_rank[0].get("rccl_ranks") if distributed else None,    # ncclCommCount of the communicator the library call ran on
                          "merge_algorithm": per_rank[0].get("algorithm") if distributed else None,
                          "dense_reduce_bytes_per_rank_for_comparison": (int(n_union) * 10240 if n_union else 0),
                          "wire_bytes_sent_per_rank": [p.get("wire_bytes_sent") for p in per_rank] if distributed else None,
                          "per_rank": per_rank},
            "per_frame": {"blocks_selected": stats["blocks_selected"] / max(stats["frames"], 1),
                          "voxels_visited": stats["voxels_visited"] / max(stats["frames"], 1),
                          "voxels_updated": n_upd_frame, "final_blocks_rank0": hv.BlockCount()},
            "kernels_ms_per_launch": {"prepare_frames": prof["prepare_ms"], "select": prof["select_ms"], "integrate": prof["integrate_ms"],
                                      "event_sampled_launches": prof["launches"], "frames_per_launch": frames_per_launch},
            # SURVEY 8(d) + the round-5 review: `frac` = the launch's HBM bytes / launch time / 8 TB/s.  `traffic` = the PMC counters' bytes per launch
            # (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes: live with --counters, else the committed measurement scaled by this run's
            # own byte model, see benchparts/roofline_extras.py); until it is filled in, the model (a lower bound of the traffic) stands in.
            "roofline": {"kernel": "k_integrate<FAST,PLAIN,ZT=2> (Integrator::IntegrateImage), %.1f frames per launch" % frames_per_launch,
                         "bound": "hbm", "achieved": batch_bytes / k3_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": batch_bytes / k3_s / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": prof["integrate_ms"], "frames_per_launch": frames_per_launch, "traffic": None, "traffic_source": "none: byte model (lower bound) used for achieved / frac",
                         "model_bytes_per_launch": batch_bytes,
                         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_frame": alg_bytes_frame,
                         # 8(d)'s per-frame bytes x frames per launch / time / peak: NOT a bound for a batched launch (each voxel is loaded once, <= 32 frames are
                         # applied in registers, stored once), hence > 1; it is a bound with one frame per launch: batch1_frac
                         "algorithmic_ratio_to_peak": alg_bytes / k3_s / 1e9 / HBM_PEAK_GBS,
                         "shader_cycles_per_launch": kc_cycles, "shader_clock_ghz": kc_cycles / k3_s / 1e9 if k3_s > 0 else None},
        }
    sections = {}
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

        def section(name, fn):
            t_s = time.perf_counter()
            fn(c, out)
            sections[name] = round(time.perf_counter() - t_s, 2)

        if single and args.full:
            section("raycast", raycast.run)            # first: the volume still holds the timed region's frames
        if not args.timed_only:
            section("roofline_extras", roofline_extras.run)
        if single and args.full:
            section("volume_ops", volume_ops.run)
            section("host_images", host_images.run)
            section("general_update", general_update.run)
            section("depth_filter", depth_filter.run)
        if single and not args.no_cpu_baseline:
            section("cpu_baseline", cpu_baseline.run)
        # ICP / tracking do not shard (replicas only): they are one-GPU measurements and run at N = 1 (or with --full), so that an N > 1 run is the
        # sharded fusion + the merge and nothing else keeps the other ranks waiting at the final barrier
        if not args.no_icp and (single or args.full):
            section("icp", icp.run)
        if not args.no_tracking and (single or args.full):
            section("tracking", tracking.run)
            section("dense_fusion", dense_fusion.run)
        out["section_seconds"] = sections

    if rank == 0:
        # the whole object goes to a file; the LAST stdout line is its compact form (the driver parses that line: round 5's 28 KB line was not parseable)
        detail = args.detail_file or os.path.join(ROOT, "bench_detail.json")
        wrote = []
        for path in [detail] + ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not args.detail_file else []):
            try:
                with open(path, "w") as f:
                    json.dump(out, f, indent=1)
                wrote.append(os.path.relpath(path, ROOT))
            except OSError:
                pass
        line = json.dumps(compact(out, wrote), separators=(",", ":"))
        assert len(line) < 8192, "the result line must stay under 8 KB (it is %d bytes)" % len(line)
        sys.stdout.flush()
        print(line, flush=True)
    if comm is not None:
        comm.destroy()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _r(x, nd=4):
    """Numbers to `nd` significant digits (the detail file keeps them whole)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact(full, detail_files=()):
    """The result line: the contract's keys + roofline + cpu_baseline + parity + icp + tracking + multi_gpu, numbers only (prose lives in DESIGN.md and the detail file)."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: full["config"].get(k) for k in ("workload", "frames_per_step", "frames_per_gpu", "voxel_m", "update_mode")}
    out["fusion_only_frames_per_s"] = full.get("fusion_only_frames_per_s")
    R = full["roofline"]
    out["roofline"] = {k: R.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms", "frames_per_launch",
                                              "model_bytes_per_launch", "algorithmic_bytes_per_launch", "algorithmic_ratio_to_peak", "batch1_frac", "batch1_avg_launch_ms",
                                              "issue_frac", "issue_source", "shader_clock_ghz") if k in R}
    if "cpu_baseline" in full:
        out["cpu_baseline"] = {k: full["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample", "host_cores", "cpu_model", "icp_iters_per_s", "icp_threads",
                                                                        "tracks_per_s", "dense_fusion_frames_per_s") if k in full["cpu_baseline"]}
    if "parity" in full:
        out["parity"] = full["parity"]
    if "icp" in full:
        ic = full["icp"]
        out["icp"] = {k: ic.get(k) for k in ("iters_per_s", "mode", "in_tolerance_iters_per_s", "in_tolerance_mode", "fp64_mode_iters_per_s", "points", "register_call_ms", "algorithmic_gbs") if k in ic}
        if "replicas" in ic and "aggregate_iters_per_s" in ic["replicas"]:
            out["icp"]["replicas"] = {k: ic["replicas"].get(k) for k in ("aggregate_iters_per_s", "speedup_over_one_context", "fp64_mode_aggregate_iters_per_s", "fp64_mode_speedup_over_one_context", "submitter", "in_flight_results_identical_to_sequential")}
        pp = ic.get("pose_parity_over_pairs")
        if pp:
            out["icp"]["pose_parity"] = {k: pp.get(k) for k in ("pairs_checked", "bar", "default_mode_within_bar", "default_mode_max_rel_err", "fp64_mode_within_bar", "fp64_mode_max_rel_err", "fp64_mode_max_rel_err_vs_cpu_double_sums") if k in pp}
    if "tracking" in full:
        tr = full["tracking"]
        out["tracking"] = {k: tr.get(k) for k in ("tracks_per_s", "mode", "fp64_mode_tracks_per_s", "from_raw_frames_tracks_per_s") if k in tr}
    if "dense_fusion" in full:
        df = full["dense_fusion"]
        out["dense_fusion"] = {k: df.get(k) for k in ("frames_per_s", "frames_per_s_mode", "pairs_in_flight", "frames_per_s_by_pairs_in_flight", "one_pair_at_a_time_frames_per_s", "frames") if k in df}
        if isinstance(df.get("outside_tolerance_fp64_mode"), dict):
            out["dense_fusion"]["fp64_mode_frames_per_s"] = df["outside_tolerance_fp64_mode"].get("frames_per_s")
        pz = df.get("pose_parity")
        if pz:
            out["dense_fusion"]["pose_parity"] = {m: {k: v.get(k) for k in ("pair_rel_err_max_vs_cpu", "pairs_within_1e-4", "pairs")} for m, v in pz.items() if isinstance(v, dict)}
    mg = full["multi_gpu"]
    out["multi_gpu"] = {k: mg.get(k) for k in ("ranks", "backend", "merge_impl", "merge_fallback", "rccl_ranks", "merge_algorithm", "wire_bytes_sent_per_rank")}
    out["multi_gpu"]["union_blocks"] = full.get("merge_union_blocks")
    out["multi_gpu"]["per_rank"] = [{k: p.get(k) for k in ("rank", "frames", "fusion_ms", "merge_ms", "local_blocks", "owned_blocks")} for p in mg.get("per_rank", [])][:8]
    out["per_frame"] = full.get("per_frame")
    out["section_seconds"] = full.get("section_seconds")
    out["detail"] = list(detail_files)
    return _r(out)


if __name__ == "__main__":
    main()
