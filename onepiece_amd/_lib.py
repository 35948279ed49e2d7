"""ctypes loader for libonepiece_hip.so (the C-ABI of include/onepiece_hip.h).

The library is the product: there is no Python/CPU fallback.  `load()` raises if the shared
object is missing (run `python -c "import __graft_entry__ as g; g.build()"` or `make -C
onepiece_amd/csrc`), and every compute entry point of the library itself fails with
OP_ERR_NO_DEVICE when no gfx950 GPU is usable.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ONEPIECE_HIP_LIBRARY: another build of the same library (the 64-frames-per-batch variant, `make -C onepiece_amd/csrc b64`)
SO_PATH = os.environ.get("ONEPIECE_HIP_LIBRARY") or os.path.join(_HERE, "libonepiece_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class OnePieceHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("onepiece_hip error %d: %s" % (code, msg))
        self.code = code


class Camera(C.Structure):
    """op_camera == camera::PinholeCamera (Camera/Camera.h:13-131) as a POD."""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("depth_scale", C.c_float)]


class MergeStats(C.Structure):
    """op_merge_stats of op_volume_merge_rccl_stats."""
    _fields_ = [("ranks", C.c_int32), ("rank", C.c_int32), ("union_blocks", C.c_uint64), ("reduce_bytes", C.c_uint64), ("slices", C.c_uint64),
                ("prepare_ms", C.c_double), ("transfer_ms", C.c_double), ("total_ms", C.c_double),
                ("algorithm", C.c_int32), ("pad_", C.c_int32), ("held_blocks", C.c_uint64), ("owned_blocks", C.c_uint64),
                ("wire_bytes_sent", C.c_uint64), ("wire_bytes_received", C.c_uint64)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("last_T", C.c_float * 16), ("rmse", C.c_double),
                ("n_inliers", C.c_uint64), ("iterations", C.c_int32)]


class TrackLevel(C.Structure):
    """op_track_level: one pyramid level of Odometry::MultiScaleComputing's inputs."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)] + [
        (k, C.c_void_p) for k in ("source_color", "source_depth", "target_color", "target_depth",
                                  "target_color_dx", "target_color_dy", "target_depth_dx", "target_depth_dy")]


class TrackResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("rmse", C.c_double), ("n_correspondences", C.c_uint64),
                ("tracking_success", C.c_int32), ("iterations", C.c_int32)]


OP_TRACK_HYBRID, OP_TRACK_PHOTO, OP_TRACK_DEPTH = 0, 1, 2
OP_TRACK_OPT_SUMS, OP_TRACK_SUMS_FP64, OP_TRACK_SUMS_REFERENCE_F32, OP_TRACK_SUMS_REFERENCE_F32_HOST = 0, 0, 1, 2
OP_DEPTH_F32, OP_DEPTH_U16 = 0, 1
OP_VOLUME_OPT_UPDATE, OP_VOLUME_UPDATE_EXACT, OP_VOLUME_UPDATE_SUM_FORM = 0, 0, 1
OP_VOLUME_OPT_SELECT, OP_VOLUME_SELECT_AUTO, OP_VOLUME_SELECT_DIRECT = 1, 0, -1
OP_VOLUME_OPT_RAYCAST_PRUNE = 2
OP_RUNTIME_OPT_MERGE_ALGORITHM, OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS, OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK, OP_RUNTIME_OPT_TRACKER_GRAPH, OP_RUNTIME_OPT_COPY_THREADS, OP_RUNTIME_OPT_CACHE_DEVICE_BYTES, OP_RUNTIME_OPT_MERGE_FAULT, OP_RUNTIME_OPT_ICP_DEFAULT_SUMS, OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS, OP_RUNTIME_OPT_TRACKER_BATCH_SUMS, OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
OP_MERGE_OWNER_EXCHANGE, OP_MERGE_DENSE_REDUCE = 0, 1
OP_MEM_HOST, OP_MEM_DEVICE = 0, 1
OP_ICP_POINT_TO_POINT, OP_ICP_POINT_TO_PLANE = 0, 1
OP_ICP_OPT_FINISH, OP_ICP_OPT_SUMS, OP_ICP_OPT_TIES = 0, 1, 2
OP_ICP_TIES_LOWEST_INDEX, OP_ICP_TIES_REFERENCE = 0, 1
OP_ICP_FINISH_REFERENCE, OP_ICP_FINISH_FP64 = 0, 1
OP_ICP_SUMS_FP64, OP_ICP_SUMS_REFERENCE_F32 = 0, 1
OP_OK, OP_ERR_INVALID = 0, 1
OP_ERR_NO_DEVICE, OP_ERR_CAPACITY, OP_ERR_MISMATCH, OP_ERR_NO_NORMALS = 2, 3, 4, 5

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_szp = C.POINTER(C.c_size_t)
_u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every symbol include/onepiece_hip.h declares
SIGNATURES = {
    "op_abi_version": (C.c_int, []),
    "op_last_error": (C.c_char_p, []),
    "op_runtime_hw_queues": (C.c_int, [C.POINTER(C.c_int)]),
    "op_runtime_configure": (C.c_int, [C.c_int]),
    "op_runtime_set_option": (C.c_int, [C.c_int, C.c_longlong]),
    "op_runtime_set_rccl_library": (C.c_int, [C.c_char_p]),
    "op_device_alloc": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "op_device_write": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]),
    "op_device_upload": (C.c_int, [_vp, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "op_device_release": (C.c_int, [_vp, C.c_int]),
    "op_release_cached_memory": (C.c_int, []),
    "op_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "op_camera_preset": (C.c_int, [C.c_int, C.POINTER(Camera)]),
    "op_mat4_inverse": (C.c_int, [_fp, _fp]),
    "op_hash_key": (C.c_uint64, [C.c_int32, C.c_int32, C.c_int32]),
    "op_frustum_planes": (C.c_int, [C.POINTER(Camera), _fp, C.c_float, C.c_float, _fp]),
    "op_frustum_from_camera": (C.c_int, [C.POINTER(Camera), _fp, C.c_float, C.c_float, _fp, _fp]),
    "op_frustum_from_vectors": (C.c_int, [_fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp]),
    "op_get_sdf": (C.c_int, [C.POINTER(Camera), _fp, _fp, _fp, _vp, C.c_int, _fp]),
    "op_se3_exp": (C.c_int, [_fp, _fp]),
    "op_debug_project_px": (C.c_int, [C.c_float, C.c_float, C.c_int]),
    "op_debug_project_uv": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "op_volume_create": (C.c_int, [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_int, C.c_uint64, C.POINTER(_vp)]),
    "op_volume_destroy": (C.c_int, [_vp]),
    "op_volume_set_resolution": (C.c_int, [_vp, C.c_float]),
    "op_volume_set_truncation": (C.c_int, [_vp, C.c_float]),
    "op_volume_set_camera": (C.c_int, [_vp, C.POINTER(Camera)]),
    "op_volume_set_near_far": (C.c_int, [_vp, C.c_float, C.c_float]),
    "op_volume_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "op_volume_progress": (C.c_int, [_vp, _u64p, _u64p]),
    "op_volume_clear": (C.c_int, [_vp]),
    "op_volume_sync": (C.c_int, [_vp]),
    "op_volume_flush": (C.c_int, [_vp]),
    "op_volume_block_count": (C.c_int, [_vp, _szp]),
    "op_volume_has_cube": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int)]),
    "op_volume_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "op_volume_compute_bounding": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _fp, _fp, _fp, _szp]),
    "op_volume_prepare_cubes": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _fp, _fp, _ip, C.c_size_t,
                                          _szp, _szp]),
    "op_volume_integrate": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _fp, _fp]),
    "op_volume_integrate_sequence": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, _fp,
                                               C.c_size_t]),
    "op_volume_integrate_cubes": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _fp, _fp, _vp, C.c_size_t]),
    "op_volume_stats": (C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p]),
    "op_volume_stats_launches": (C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p]),
    "op_volume_growth_stats": (C.c_int, [_vp, _u64p, _u64p, _u64p]),
    "op_volume_profile_enable": (C.c_int, [_vp, C.c_int]),
    "op_volume_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), _u64p, _u64p]),
    "op_volume_download": (C.c_int, [_vp, _ip, _fp, C.c_size_t, _szp]),
    "op_volume_upload": (C.c_int, [_vp, _ip, _fp, C.c_size_t]),
    "op_volume_merge": (C.c_int, [_vp, _vp]),
    "op_volume_transform": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_uint64, C.POINTER(_vp)]),
    "op_volume_resolution": (C.c_int, [_vp, _fp]),
    "op_volume_point_cloud": (C.c_int, [_vp, _fp, _fp, C.c_size_t, _szp]),
    "op_volume_extract_mesh": (C.c_int, [_vp, _ip, _ip, _ip, _fp, _fp, C.c_size_t, _szp]),
    "op_volume_write_file": (C.c_int, [_vp, C.c_char_p]),
    "op_volume_read_file": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "op_volume_raycast": (C.c_int, [_vp, C.POINTER(Camera), _fp, _fp, _fp, _fp, C.c_int]),
    "op_volume_raycast_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "op_volume_keys_device": (C.c_int, [_vp, _vp, C.c_size_t, _szp]),
    "op_volume_pack_sum": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "op_volume_unpack_sum": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "op_volume_unpack_sum_begin": (C.c_int, [_vp, _vp, C.c_size_t]),
    "op_volume_unpack_sum_chunk": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp]),
    "op_volume_merge_rccl": (C.c_int, [_vp, _vp, C.c_int, _szp]),
    "op_volume_merge_rccl_stats": (C.c_int, [_vp, _vp, C.c_int, _szp, _vp]),
    "op_icp_create": (C.c_int, [_vp, _vp, C.c_size_t, C.c_double, C.c_int, C.c_int, C.POINTER(_vp)]),
    "op_icp_destroy": (C.c_int, [_vp]),
    "op_icp_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "op_icp_run_many": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _fp, C.c_int, _vp]),
    "op_icp_tie_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "op_icp_final_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "op_icp_set_source": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "op_icp_iterate": (C.c_int, [_vp, _fp, C.c_int, C.POINTER(C.c_double), _u64p,
                                 C.POINTER(C.c_double)]),
    "op_icp_run_enqueue": (C.c_int, [_vp, C.c_int, _fp, C.c_int, C.POINTER(IcpResult), _ip, C.c_size_t]),
    "op_icp_wait": (C.c_int, [_vp]),
    "op_icp_run": (C.c_int, [_vp, C.c_int, _fp, C.c_int, C.POINTER(IcpResult), _ip, C.c_size_t, _ip,
                             _fp]),
    "op_icp_register": (C.c_int, [C.c_int, _fp, C.c_size_t, _fp, _fp, C.c_size_t, _fp, C.c_int,
                                  C.c_double, C.c_int, C.POINTER(IcpResult), _ip, C.c_size_t]),
    "op_estimate_normals": (C.c_int, [_vp, C.c_size_t, C.c_float, C.c_int, C.c_int, C.c_int, _vp]),
    "op_estimate_rigid_point_to_plane": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, _fp]),
    "op_estimate_rigid_transformation": (C.c_int, [_vp, C.c_size_t, C.c_int, C.c_int, _fp]),
    "op_estimate_rigid_point_to_plane_ex": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp]),
    "op_estimate_rigid_transformation_ex": (C.c_int, [_vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp]),
    "op_points_from_rgbd": (C.c_int, [C.POINTER(Camera), _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _szp]),
    "op_bilateral_filter_depth": (C.c_int, [_vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _vp, _vp]),
    "op_points_from_depth": (C.c_int, [C.POINTER(Camera), _vp, C.c_int, C.c_int, C.c_int, _vp, _szp]),
    "op_tracker_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "op_tracker_destroy": (C.c_int, [_vp]),
    "op_tracker_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "op_tracker_track": (C.c_int, [_vp, C.POINTER(TrackLevel), C.c_int, _ip, C.c_int, C.c_int, C.c_int, _fp,
                                   C.c_int, C.POINTER(TrackResult), _vp, _vp, C.c_size_t, _vp, _vp]),
    "op_tracker_correspondences": (C.c_int, [_vp, C.POINTER(TrackLevel), _fp, C.c_int, _vp, C.c_size_t, _szp]),
    "op_tracker_dense_tracking": (C.c_int, [_vp, C.POINTER(Camera), C.c_int, _ip, _vp, _vp, _vp, _vp, C.c_int, _fp, C.c_int,
                                            C.c_int, C.POINTER(TrackResult), _vp, _vp, C.c_size_t]),
    "op_tracker_dense_tracking_enqueue": (C.c_int, [_vp, C.POINTER(Camera), C.c_int, _ip, _vp, _vp, _vp, _vp, C.c_int, _fp, C.c_int,
                                                    C.c_int, C.c_int]),
    "op_tracker_wait": (C.c_int, [_vp, C.POINTER(TrackResult), _vp, _vp, C.c_size_t]),
    "op_tracker_read_pyramid": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t]),
    "op_dense_track": (C.c_int, [C.POINTER(TrackLevel), C.c_int, _ip, C.c_int, C.c_int, C.c_int, _fp, C.c_int,
                                 C.c_int, C.POINTER(TrackResult), _vp, _vp, C.c_size_t]),
    "op_track_projection": (C.c_int, [_fp, _fp, _fp, _fp]),
    "op_ldlt_solve6": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), _fp]),
}

_lib = None


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(_HERE, "..", "include", "onepiece_hip.h")]
    if (not force and os.path.exists(SO_PATH)
            and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(s) for s in srcs)):
        return SO_PATH
    subprocess.check_call(["make", "-C", CSRC, "-B"])
    return SO_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise OnePieceHipError(-1, "libonepiece_hip.so is not built (%s); there is no fallback path. "
                               "Run __graft_entry__.build() or `make -C onepiece_amd/csrc`." % SO_PATH)
    try:
        # One HIP runtime per process: PyTorch bundles its own libamdhip64; if ours (from /opt/rocm)
        # were loaded first, torch would later fail with "No HIP GPUs are available".  Importing
        # torch first makes libonepiece_hip.so bind to the runtime torch uses, so device pointers
        # and streams are shared.  (C/C++ callers without torch simply use /opt/rocm's runtime.)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise OnePieceHipError(rc, load().op_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(load().op_device_count(C.byref(n)))
    return n.value


def torch_ready(t):
    """Makes a CUDA torch tensor's contents final before a kernel on one of the LIBRARY's streams reads it: the library's
    objects own their HIP streams and know nothing of torch's, so work still queued on torch's current stream (the
    tensor's producer) would race with them.  A query when torch's stream is idle (~1 us), a synchronise otherwise."""
    if getattr(t, "is_cuda", False):
        import torch
        s = torch.cuda.current_stream(t.device)
        if not s.query():
            s.synchronize()
    return t
