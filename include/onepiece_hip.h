/*
 * onepiece_hip.h -- C-ABI of the MI355X (gfx950) implementation of OnePiece's TSDF-fusion + ICP
 * hot path.  This is the drop-in boundary: the reference has no FFI layer (its boundary is the C++
 * header surface one_piece::integration::CubeHandler / one_piece::registration::PointToPlane), so
 * every entry point below names the reference declaration (file:line under /root/reference/src)
 * whose work it replaces; INTEGRATION.md shows the forwarding calls a maintainer adds inside the
 * reference's CubeHandler / ICP to bind them.
 *
 * Conventions
 *   - extern "C", POD only, caller-allocated outputs, int return: 0 = OP_OK, otherwise an
 *     op_status; op_last_error() returns a thread-local description of the last failure.
 *   - All 4x4 matrices are ROW-MAJOR float[16] (the shim transposes Eigen's column-major storage).
 *   - Images: depth is float32 metres (OP_DEPTH_F32, cv CV_32FC1) or uint16 / depth_scale
 *     (OP_DEPTH_U16, cv CV_16UC1), row-major width x height; colour is 3 bytes / pixel in stored
 *     channel order (cv CV_8UC3).  `mem` says where image/point buffers live: OP_MEM_HOST buffers
 *     are copied to the device by the call, OP_MEM_DEVICE buffers are used in place and must stay
 *     valid until op_volume_sync()/the next synchronising call.
 *   - A volume owns one HIP stream.  op_volume_integrate() only enqueues work; every accessor that
 *     returns data synchronises first (SURVEY.md 8b "Threading").
 *   - There is NO CPU fallback: every compute entry point fails with OP_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef ONEPIECE_HIP_H
#define ONEPIECE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OP_ABI_VERSION 1

typedef enum {
    OP_OK = 0,
    OP_ERR_INVALID = 1,      /* bad argument */
    OP_ERR_NO_DEVICE = 2,    /* no usable HIP device / HIP runtime error */
    OP_ERR_CAPACITY = 3,     /* block pool or hash table exhausted (create with more max_blocks) */
    OP_ERR_MISMATCH = 4,     /* e.g. Merge of volumes with different voxel resolution */
    OP_ERR_NO_NORMALS = 5,   /* PointToPlane without target normals (ICP.cpp:159-163) */
    OP_ERR_HIP = 6
} op_status;

enum { OP_DEPTH_F32 = 0, OP_DEPTH_U16 = 1 };
enum { OP_MEM_HOST = 0, OP_MEM_DEVICE = 1 };
enum { OP_ICP_POINT_TO_POINT = 0, OP_ICP_POINT_TO_PLANE = 1 };
enum { OP_TRACK_HYBRID = 0, OP_TRACK_PHOTO = 1, OP_TRACK_DEPTH = 2 }; /* DenseTracking term_type (Odometry.cpp:546-553) */

/* camera::PinholeCamera (Camera/Camera.h:13-131) as a POD. */
typedef struct {
    float fx, fy, cx, cy;
    int32_t width, height;
    float depth_scale;
} op_camera;

typedef struct op_volume op_volume; /* device-resident integration::CubeHandler state */
typedef struct op_icp op_icp;       /* device-resident target cloud + search grid */
typedef struct op_tracker op_tracker; /* dense RGB-D tracker workspace (stream, device state) */

/* ---- library ----------------------------------------------------------------------------- */
int op_abi_version(void);
const char *op_last_error(void);
/* ---- process-wide settings.  The library reads nothing from the environment and changes nothing in it on its own (rounds 1-4 set
 * GPU_MAX_HW_QUEUES from a load-time constructor and took test hooks from ONEPIECE_* variables): a host opts in through these calls.
 *
 * op_runtime_configure(hw_queues): every volume and every tracker owns a HIP stream, and streams that share a hardware queue serialise;
 *   the runtime maps streams onto 4 queues unless GPU_MAX_HW_QUEUES says otherwise, and reads that variable ONCE, at its first API call.
 *   This call sets it (never overwriting a value the caller has set) -- to be made before the process touches HIP; the C++ class surface
 *   (host/one_piece) makes it when its first device object is constructed, the Python package before it loads the library -- both ask for 16:
 *   K one-workgroup kernels on K streams take ceil(K / queues) kernel times (tools/queue_probe.hip), and the tracking + fusion pipeline's rate with
 *   16 pairs in flight goes from 349 to 489 frames/s between 8 and 16 queues (with 32 the rates of a process that also runs torch collapse).
 * op_runtime_hw_queues: what is in the variable now (4 = the runtime's default when unset); whether the runtime had already read it cannot be known. */
int op_runtime_configure(int hw_queues);
int op_runtime_hw_queues(int *requested);
/* op_runtime_set_option:
 *   OP_RUNTIME_OPT_MERGE_ALGORITHM        OP_MERGE_OWNER_EXCHANGE (default) / OP_MERGE_DENSE_REDUCE -- see op_volume_merge_rccl
 *   OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS     union blocks per reduce slice of the dense merge (0 = the built-in 32 768; tests force several slices)
 *   OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK  1: run the whole exchange with a one-rank communicator too (how a one-GPU box exercises the RCCL path)
 *   OP_RUNTIME_OPT_TRACKER_GRAPH          0: trackers created afterwards issue plain launches instead of replaying a captured hipGraph
 *   OP_RUNTIME_OPT_COPY_THREADS           0 .. 8 helper threads of the pageable -> pinned staging copies (before the first host image)
 *   OP_RUNTIME_OPT_CACHE_DEVICE_BYTES     bytes of RELEASED device buffers (block pools, cell tables, images) the library keeps per device for the next
 *                                         request of a similar size instead of returning them to the driver (default 32 GB; 0 = keep none; buffers
 *                                         already kept stay until op_release_cached_memory)
 *   OP_RUNTIME_OPT_MERGE_FAULT            TEST HOOK: stage * 1024 + rank + 1 makes that rank's allocation of that stage of the owner-exchange merge fail
 *                                         (stage 1: the partition's sums after the exchange; stage 2: the root's gather buffers) -- every rank must then
 *                                         return an error, nobody may wait inside RCCL (tests/test_config5_gpu.py); 0 = off (default)
 *   OP_RUNTIME_OPT_ICP_DEFAULT_SUMS       the OP_ICP_OPT_SUMS mode of ICP contexts created afterwards (op_icp_create, op_icp_register, hence
 *                                         registration::PointToPlane / PointToPoint of the class surface).  Default OP_ICP_SUMS_REFERENCE_F32: the reference's own
 *                                         sequential float32 sums, the mode whose pose is within 1e-4 of the CPU path's on EVERY pair (~0.6 k iterations/s at
 *                                         307 200 points); OP_ICP_SUMS_FP64 opts into the order-free fp64 reduction (~24 k iterations/s; equal to the CPU path
 *                                         with double sums, but up to 1e-2 from its float32 answer where J^T J is rank-deficient -- DESIGN.md section 5)
 *   OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS   likewise the OP_TRACK_OPT_SUMS mode of trackers created afterwards (Odometry::DenseTracking of the class surface): default
 *                                         OP_TRACK_SUMS_REFERENCE_F32 (every pair within 1e-4 of the CPU path: ~75 tracks/s alone, ~230 frames/s with four pairs in flight);
 *                                         OP_TRACK_SUMS_FP64 opts into the fp64 reduction (~2.2 k tracks/s; 20 of 23 pairs of the bench's chain within 1e-4, worst 4.4e-4)
 *   OP_RUNTIME_OPT_TRACKER_BATCH_SUMS     0 (default): every tracker launches its own one-workgroup sum kernel.  1: twelve or more trackers in the reference-order mode that
 *                                         run at the same time (pairs in flight of a pipeline) meet once per iteration and take their sequential sums in ONE launch, a
 *                                         workgroup per tracker (what op_icp_run_many does for nine or more reference-order ICP contexts).  Measured WITHOUT gain for the
 *                                         tracker (profiles/r06_track_depth_probe.txt): what its pipeline needs is a hardware queue per tracker stream
 *                                         (op_runtime_configure(16)).  Results do not depend on it.
 *   OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT     iterations op_icp_run_many keeps enqueued at a time, in turn over its fp64-mode contexts (default 4, 1 .. 1024)
 * op_runtime_set_rccl_library(path): the RCCL to bind at the first merge instead of "librccl.so.1" (a site build; the test suite names a
 *   host-memory double that runs several ranks on one device); NULL = the system's.  Fails once RCCL has been bound. */
#define OP_RUNTIME_OPT_MERGE_ALGORITHM 0
#define OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS 1
#define OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK 2
#define OP_RUNTIME_OPT_TRACKER_GRAPH 3
#define OP_RUNTIME_OPT_COPY_THREADS 4
#define OP_RUNTIME_OPT_CACHE_DEVICE_BYTES 5
#define OP_RUNTIME_OPT_MERGE_FAULT 6
#define OP_RUNTIME_OPT_ICP_DEFAULT_SUMS 7
#define OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS 8
#define OP_RUNTIME_OPT_TRACKER_BATCH_SUMS 9
#define OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT 10
#define OP_MERGE_OWNER_EXCHANGE 0
#define OP_MERGE_DENSE_REDUCE 1
int op_runtime_set_option(int option, long long value);
int op_runtime_set_rccl_library(const char *path);
/* Images that are used more than once -- a frame is tracked against twice and fused once (example/DenseFusion/DenseSlam.cpp:24-33,
 * DenseFusion.cpp:86-96) -- can be brought to the device ONCE and then handed to op_tracker_dense_tracking(_enqueue) /
 * op_volume_integrate with OP_MEM_DEVICE.
 *   op_device_alloc    a buffer from the library's buffer cache (hipMalloc synchronises the whole device: allocate slabs, not frames)
 *   op_device_write    one or two host arrays -> the buffer at the given byte offsets: staged through pinned memory by the caller's thread
 *                      and the library's copy helpers, one DMA; blocking (complete and visible to every stream on return)
 *   op_device_upload   alloc + write of one array
 *   op_device_release  returns a buffer of op_device_alloc / op_device_upload
 * The caller keeps a buffer alive until its last consumer has finished (op_tracker_wait; for a volume: op_volume_progress). */
int op_device_alloc(size_t bytes, int device, void **device_ptr);
int op_device_write(void *device_ptr, size_t n_parts, const void *const *parts, const size_t *bytes, const size_t *offsets, int device);
int op_device_upload(const void *host, size_t bytes, int device, void **device_ptr);
int op_device_release(void *device_ptr, int device);
/* Registration objects (op_icp, the contexts behind op_icp_register / op_estimate_normals / op_points_from_depth) are
 * created and dropped per call, as registration::PointToPlane builds and drops its kd-tree (Registration/ICP.cpp:
 * 166-170); their device and pinned buffers, streams and events are kept in a per-process cache when released and
 * handed out again, so that a steady stream of calls does not allocate.  This empties the cache (at most 8 GB of
 * device memory per GPU and 1 GB of pinned memory are ever held). */
int op_release_cached_memory(void);
int op_device_count(int *count);
/* OPEN3D_DATASET / TUM_DATASET presets, Camera/Camera.h:76-104.  type: 0 = TUM, 1 = OPEN3D. */
int op_camera_preset(int type, op_camera *out);

/* ---- host-side geometry used on the path (bit-faithful to the reference's Eigen build) ---- */
/* geometry::TransformationMatrix::inverse() as Eigen 3.3.7 evaluates it with -msse4.2
 * (Integrator.cpp:18,48; 3rdparty/Eigen/Eigen/src/LU/arch/Inverse_SSE.h:35-165). */
int op_mat4_inverse(const float m[16], float out[16]);
/* geometry::VoxelGridHasher::operator() (Geometry/Geometry.h:101-112). */
uint64_t op_hash_key(int32_t x, int32_t y, int32_t z);
/* Frustum::ComputeFromCamera (Integration/Frustum.cpp:7-46): planes top,left,right,bottom,near,far. */
int op_frustum_planes(const op_camera *cam, const float pose[16], float far_dist, float near_dist,
                      float planes[24]);
/* The whole integration::Frustum object (Integration/Frustum.h:10-105): the six planes as above plus, when
 * corners != NULL, the eight corner points in the order of the reference's public `corners` member
 * (Frustum.cpp:49-56: far top-left, far top-right, far bottom-left, far bottom-right, near bottom-right,
 * near top-left, near top-right, near bottom-left).  op_frustum_from_camera = Frustum::ComputeFromCamera
 * (Frustum.cpp:7-24), op_frustum_from_vectors = Frustum::ComputeFromVectors (:25-94).  Host arithmetic in
 * the reference's float order; no device needed. */
int op_frustum_from_camera(const op_camera *cam, const float pose[16], float far_dist, float near_dist,
                           float planes[24], float corners[24]);
int op_frustum_from_vectors(const float forward[3], const float position[3], const float right[3],
                            const float up[3], float far_dist, float near_dist, float fov, float aspect,
                            float planes[24], float corners[24]);
/* Integrator::GetSDF (Integrator.cpp:8-35) for one world point against a HOST depth image: 999 when the point projects
 * off the image or onto a pixel without depth, else depth - z.  pose_inv may be NULL (computed as op_mat4_inverse).
 * A host scalar in the reference's float / double order; the kernels evaluate the same expression for every probe. */
int op_get_sdf(const op_camera *cam, const float point[3], const float pose[16], const float *pose_inv,
               const void *depth, int depth_fmt, float *sdf);
/* Test hook: the pixel rounding of Integrator.cpp:20-21 applied to a = fx*X/Z and c = cx.
 * fast = 0: the reference's double formula; fast = 1: the fp32/integer evaluation the kernels use
 * (valid when |c| >= 1 or c == 0).  Both return INT_MIN for values no int can hold. */
int op_debug_project_px(float a, float c, int fast);
/* Test hook (device): the kernels' projection of n operand triples -- u = (fx*X)/Z + 0.5 + cx, v likewise, the two
 * quotients sharing one refined reciprocal (csrc/volume_core.hpp: project_pixel) -- next to the same formula with the plain
 * IEEE division.  out: n x {u, v, u_plain, v_plain}; INT_MIN stands for "no int can hold it". */
int op_debug_project_uv(float fx, float fy, float cx, float cy, const float *X, const float *Y, const float *Z,
                        size_t n, int device, int32_t *out);
/* geometry::Se3ToSE3 (Geometry/Geometry.cpp:9-13). */
int op_se3_exp(const float x[6], float T[16]);

/* ---- integration::CubeHandler (Integration/CubeHandler.h:24-366) ------------------------- */
/* CubeHandler(const PinholeCamera&) + SetVoxelResolution + SetTruncation + SetFar/NearPlane.
 * max_blocks sizes the device block pool (10 KiB per 8^3 block) and hash table; 0 = default. */
int op_volume_create(const op_camera *cam, float voxel_res, float truncation, float far_dist,
                     float near_dist, int device, uint64_t max_blocks, op_volume **out);
int op_volume_destroy(op_volume *v);
int op_volume_set_resolution(op_volume *v, float voxel_res);  /* CubeHandler.h:36-39 */
int op_volume_set_truncation(op_volume *v, float truncation); /* CubeHandler.h:141-144 */
int op_volume_set_camera(op_volume *v, const op_camera *cam); /* CubeHandler.h:137-140 */
int op_volume_set_near_far(op_volume *v, float near_dist, float far_dist); /* CubeHandler.h:339-346 */
/* OP_VOLUME_OPT_UPDATE: how the frames of a batch reach a voxel (TSDFVoxel::operator+, Integration/TSDFVoxel.h:24-39; Integrator.cpp:74-87).
 *   OP_VOLUME_UPDATE_EXACT (default): frame by frame in frame order, every product, sum and quotient rounded as the reference rounds it --
 *     voxels bit-identical to the reference's CPU path.
 *   OP_VOLUME_UPDATE_SUM_FORM: the observations a batch of frames (<= 32) makes of a voxel are summed and the weighted mean with the stored
 *     voxel is formed once per batch.  Same blocks, same pixels, same weights; sdf and colour agree with the reference to float rounding
 *     (<= 1e-6 of the truncation distance / of 1 in practice; the path's bar is 1e-4 relative).  The integrate kernel executes a third
 *     fewer instructions per voxel and frame.  Precondition: truncation < 1 -- the reference re-tests IsValid (sdf < 1) before every frame,
 *     the sum form once per batch (on the stored voxel), and the two only agree while no observation can itself be "invalid"; with
 *     truncation >= 1 the exact update runs whatever this option says.
 * Changing the option waits for the frames accepted so far (they are fused under the old setting). */
#define OP_VOLUME_OPT_UPDATE 0
#define OP_VOLUME_UPDATE_EXACT 0
#define OP_VOLUME_UPDATE_SUM_FORM 1
/* OP_VOLUME_OPT_SELECT: which form of the selection step (CubeHandler::PrepareCubes, CubeHandler.cpp:147-196) a batch takes.  The
 * selected set is the same either way (tests/test_integration_gpu.py); a tuning and test knob, not part of the reference's surface.
 *   OP_VOLUME_SELECT_AUTO (default): in batches of >= 20 frames the frames record their selections per super-block (4 x 4 x 4 blocks) and one pass
 *                                    claims every block once (faster from ~16 frames on); a frame whose candidate range exceeds 2^18 super-blocks,
 *                                    and every frame of a shorter batch, claims its blocks directly;
 *   OP_VOLUME_SELECT_DIRECT: every frame claims directly (the only form of rounds 1-3);
 *   n >= 1: every batch of >= 2 frames takes the recording form, with n super-blocks as the limit (forces every mix of the two on small scenes). */
#define OP_VOLUME_OPT_SELECT 1
#define OP_VOLUME_SELECT_AUTO 0
#define OP_VOLUME_SELECT_DIRECT -1
/* OP_VOLUME_OPT_RAYCAST_PRUNE (default 1): op_volume_raycast remembers, per block it loaded, whether the block holds an observed sdf <= 0 / > 0,
 * and later views use that to drop blocks in which no zero crossing can end before loading them.  Fusion with the default (exact) update keeps
 * the knowledge current -- the kernel that changes a block restates its summary -- so a view that follows a fused frame still prunes; every other
 * writer (sum-form fusion, upload, merge, resampling, clear, pool growth) invalidates what was remembered.  0: every view loads every visible block,
 * as the first view after such a change does.
 * The images are identical either way (tests/test_volume_ops_gpu.py); a measurement and test knob. */
#define OP_VOLUME_OPT_RAYCAST_PRUNE 2
int op_volume_set_option(op_volume *v, int option, int value);
/* How far the volume has got with the frames handed to op_volume_integrate / _sequence, WITHOUT waiting: frames accepted so far, and how many
 * of them belong to batches the device has reported complete (frames are queued up to 32 per launch and a launched batch may still be
 * replayed after a pool growth).  Device images of the first `frames_done` frames (OP_MEM_DEVICE, in call order) are
 * no longer read and may be released or overwritten; host images are only borrowed for the call anyway. */
int op_volume_progress(op_volume *v, uint64_t *frames_accepted, uint64_t *frames_done);
int op_volume_clear(op_volume *v);                            /* CubeHandler.h:133-136 */
int op_volume_sync(op_volume *v);
/* Launches the frames op_volume_integrate / op_volume_integrate_sequence have queued (a batch smaller than 32), without
 * waiting for them: for callers that want the GPU to start now although more frames will follow.  No reference counterpart
 * (the reference fuses inside the call). */
int op_volume_flush(op_volume *v);
int op_volume_block_count(op_volume *v, size_t *n);
int op_volume_has_cube(op_volume *v, int32_t x, int32_t y, int32_t z, int *present); /* :129-132 */
/* Raw HIP stream handle of the volume (hipStream_t), for callers that time with HIP events. */
int op_volume_stream(op_volume *v, void **stream);

/* CubeHandler::ComputeBounding (CubeHandler.cpp:116-145). */
int op_volume_compute_bounding(op_volume *v, const void *depth, int depth_fmt, int mem,
                               const float pose[16], float max_pos[3], float min_pos[3],
                               size_t *n_inside);
/* CubeHandler::PrepareCubes (CubeHandler.cpp:147-196): selects + allocates blocks; ids_xyz gets the
 * cube_id_list in the reference's i,j,k loop order (n x 3).  *n is the full list length even when
 * it exceeds cap.  pose_inv may be NULL (then op_mat4_inverse(pose) is used). */
int op_volume_prepare_cubes(op_volume *v, const void *depth, int depth_fmt, int mem,
                            const float pose[16], const float *pose_inv, int32_t *ids_xyz,
                            size_t cap, size_t *n, size_t *n_candidates);
/* CubeHandler::IntegrateImage(depth, rgb, pose) (CubeHandler.cpp:197-210 -> Integrator.cpp:36-94).
 * Asynchronous: frames are queued and fused up to 32 at a time (results identical to frame-by-frame
 * fusion); every accessor / setter / op_volume_sync flushes the queue first.  OP_MEM_HOST images are
 * copied before the call returns; OP_MEM_DEVICE images must stay valid until the next
 * synchronising call. */
int op_volume_integrate(op_volume *v, const void *depth, int depth_fmt, const uint8_t *rgb,
                        int mem, const float pose[16], const float *pose_inv);
/* Integrator::IntegrateImage(depth, rgb, pose, camera, voxel_cube, c_para) (Integrator.cpp:36-94), the reference's
 * public per-cube member, for a caller-chosen list of n cube ids: the frame is fused into exactly those cubes
 * (allocated when absent; a cube listed twice is fused once), without PrepareCubes' selection.  Synchronous. */
int op_volume_integrate_cubes(op_volume *v, const void *depth, int depth_fmt, const uint8_t *rgb, int mem,
                              const float pose[16], const float *pose_inv, const int32_t *keys_xyz, size_t n);
/* Multi-frame form of the same call for frames already resident on the device: frame f uses
 * depth + f*depth_stride_bytes, rgb + f*rgb_stride_bytes, poses + 16*f.  Results are bit-identical
 * to n_frames sequential op_volume_integrate calls -- the frames join the same queue: they are fused in batches of
 * up to 32 per kernel launch (inside a batch every voxel applies its frames in order in registers), a remainder
 * waits for the next frames or the next flushing call, and the images must stay valid until then. */
int op_volume_integrate_sequence(op_volume *v, const void *depth, size_t depth_stride_bytes,
                                 int depth_fmt, const uint8_t *rgb, size_t rgb_stride_bytes,
                                 const float *poses, size_t n_frames);
/* Counters since create/clear (synchronises): frames integrated, sum over frames of
 * len(cube_id_list), of voxels visited (512 x len) and of voxels that passed the update predicate
 * (Integrator.cpp:63,70,74). */
int op_volume_stats(op_volume *v, uint64_t *frames, uint64_t *blocks_selected, uint64_t *voxels_visited,
                    uint64_t *voxels_updated);
/* Launch-level counters since create/clear (synchronises; measurement hook, no reference counterpart):
 * Integrator::IntegrateImage runs as ONE kernel launch per batch of up to 32 frames that reads every
 * selected block once and writes every voxel that changed once, so the HBM traffic of a launch is
 * bounded below by 20 B x 512 x blocks_read + 20 B x voxels_written (+ the packed images), whatever
 * the number of frames in the batch.  launches = k_integrate launches that fused something; shader_cycles = their summed
 * durations in SHADER clock cycles (the longest s_memtime span of one of the launch's resident workgroups) --
 * the denominator of the kernel's instruction-issue utilisation, whatever the clock did while it ran. */
int op_volume_stats_launches(op_volume *v, uint64_t *launches, uint64_t *blocks_read, uint64_t *voxels_written,
                             uint64_t *shader_cycles);

/* The volume grows like the reference's std::unordered_map (CubeHandler.h:22): current capacity in blocks, how often the
 * pool has been enlarged since create, and how many batches were launched a second time because a batch had exhausted the
 * pool before the host noticed (results are unaffected; the replays cost time).  Does not synchronise. */
int op_volume_growth_stats(op_volume *v, uint64_t *max_blocks, uint64_t *grows, uint64_t *replayed_batches);
/* Measurement hook (no reference counterpart): with sample_every = k > 0, every k-th launch group
 * (one group = one op_volume_integrate call, or one batch of up to 32 frames of
 * op_volume_integrate_sequence) is bracketed by HIP events on the volume's stream.  profile_read
 * synchronises and returns the summed durations in ms of the three kernels -- [0] frame
 * preparation + ComputeBounding (KA), [1] PrepareCubes (KB), [2] Integrator::IntegrateImage (KC) --
 * over n_launches sampled groups covering n_frames frames. */
int op_volume_profile_enable(op_volume *v, int sample_every);
int op_volume_profile_read(op_volume *v, double ms_sum[3], uint64_t *n_launches, uint64_t *n_frames);

/* CubeHandler::GetCubeMap (CubeHandler.h:347-350): keys n x 3 int32; voxels n x 512 x 5 float in
 * the reference's TSDFVoxel member order {sdf, weight, color[0..2]}, voxel index x + 8y + 64z. */
int op_volume_download(op_volume *v, int32_t *keys_xyz, float *voxels_aos, size_t cap, size_t *n);
/* CubeHandler::SetCubeMap / AddCube + assignment (CubeHandler.h:185-198,351-356). */
int op_volume_upload(op_volume *v, const int32_t *keys_xyz, const float *voxels_aos, size_t n);
/* CubeHandler::Merge(const CubeHandler&) (CubeHandler.h:145-167), both volumes on one device. */
int op_volume_merge(op_volume *dst, op_volume *src);

/* CubeHandler::Transform (nearest = 0, trilinear ReadVoxelInterpolate, CubeHandler.h:242-298,
 * VoxelCube.cpp:6-50) / CubeHandler::TransformNearest (nearest = 1, CubeHandler.h:299-338) on the
 * device: returns a NEW volume on the same device (destroy it with op_volume_destroy).  T_inv may
 * be NULL.  The reference's quirk is kept: TransformNearest's result has the default voxel
 * resolution 0.01 and allocates its blocks with it.  max_blocks = 0 sizes the result automatically. */
int op_volume_transform(op_volume *src, const float T[16], const float *T_inv, int nearest,
                        uint64_t max_blocks, op_volume **out);
int op_volume_resolution(op_volume *v, float *voxel_res);
/* CubeHandler::GetPointCloud (CubeHandler.cpp:45-69): points and grey colours |sdf|/truncation
 * (n x 3 floats each, host).  *n is the full count; xyz/colors may be NULL to query it. */
int op_volume_point_cloud(op_volume *v, float *xyz, float *colors, size_t cap, size_t *n);
/* CubeHandler::ExtractTriangleMesh (Integration/CubeHandler.cpp:9-44) = GenerateMeshByCube (:70-114) over
 * every block + MarchingCube (MarchingCube.cpp:8-74), on the device.  tri_table (256 x 16 int32, rows of edge
 * ids terminated by -1) and edge_pairs (12 x 2 corner ids) are the CALLER'S tables -- in the reference they are
 * `MCLookTable` / `EdgeIndexPairs` of MarchingCubePredefined.h, which its shim passes straight through (the
 * library neither contains nor assumes a particular table).  Output: three unshared vertices per triangle
 * (triangle k = vertices 3k..3k+2), points and colours as n x 3 floats, exactly as MarchingCube() pushes them;
 * blocks in pool order (the reference's order is its unordered_map's), voxels in the x, y, z loop order.
 * only_block: NULL, or the one CubeID to mesh (= GenerateMeshByCube).  points/colors may be NULL to query
 * *n_vertices. */
int op_volume_extract_mesh(op_volume *v, const int32_t *tri_table, const int32_t *edge_pairs,
                           const int32_t *only_block, float *points, float *colors,
                           size_t cap_vertices, size_t *n_vertices);
/* CubeHandler::WriteToFile / ReadFromFile / ReadFromFileFloat (CubeHandler.h:40-128): the .map
 * float-stream format (VoxelCube.h:128-193).  Blocks are written in pool order. */
int op_volume_write_file(op_volume *v, const char *path);
int op_volume_read_file(op_volume *v, const char *path, int legacy_float_format);

/* TSDF ray casting (north_star "integrate/raycast").  NO reference counterpart: OnePiece has no
 * raycast (its closest relative is the trilinear gather of VoxelCube::ReadVoxelInterpolate,
 * Integration/VoxelCube.cpp:6-50), so the definition is this library's own, restated on the CPU in
 * the test suite's CPU checker and validated against the analytic synthetic scene.  For every pixel (u, v) of
 * `cam` (NULL = the volume's camera) at camera-to-world `pose`:
 *   ray      origin = the pose's translation, direction d = R ((u - cx) / fx, (v - cy) / fy, 1): the parameter t is the z-depth;
 *   lattice  t_k = near + k * res for k = 0, 1, ... while t_k <= far;  p_k = origin + t_k d;
 *   sample   s_k = trilinear sdf over the 8 voxel centres around p_k (g = p * (1 / res) - 0.5, base voxel floor(g)),
 *            valid when all 8 voxels are observed (weight > 0);
 *   hit      the smallest k >= 1 with s_(k-1) valid and > 0 and s_k valid and <= 0:
 *            depth = t_(k-1) + (t_k - t_(k-1)) * s_(k-1) / (s_(k-1) - s_k), in metres (0 = no hit).
 * Optional outputs (W*H*3 floats each, zero where there is no hit): world-frame normals = the normalised central difference
 * of the trilinear sdf at +-res/2 around the hit point, and the trilinear colour there -- each zero unless all its samples are valid.
 * The crossing test is local to a lattice pair, so the result does not depend on the order the volume is traversed in: the kernels
 * march block by block from an LDS tile per block (csrc/raycast.hip) and agree with the CPU restatement bit for bit (depth). */
int op_volume_raycast(op_volume *v, const op_camera *cam, const float pose[16], float *depth_out,
                      float *normals_out, float *colors_out, int mem);
/* Measurement hook: what the LAST op_volume_raycast call on this volume did -- blocks whose sample domain met the view frustum, how many
 * of those earlier views' summaries dropped before loading (OP_VOLUME_OPT_RAYCAST_PRUNE), how many were loaded into LDS and how many of
 * the loaded ones were marched (the others held no zero crossing). */
int op_volume_raycast_stats(op_volume *v, uint64_t *visible_blocks, uint64_t *dropped_unloaded, uint64_t *loaded_blocks, uint64_t *marched_blocks);

/* Frame-sharded multi-GPU merge (the distributed form of CubeHandler::Merge; DESIGN.md "Multi-GPU").
 * All pointers are DEVICE pointers on the volume's device.
 *   keys_device : copy this volume's block keys (n x 3 int32) into d_keys.
 *   pack_sum    : for the n_union keys (any order) write sum-form voxels
 *                 [w*sdf, w, w*c0, w*c1, w*c2] x 512 per block, SoA planes per block
 *                 (5 x 512 floats), zeros where this volume has no block / w == 0.
 *   unpack_sum  : replace the volume's content by the normalised reduction result. */
int op_volume_keys_device(op_volume *v, int32_t *d_keys, size_t cap, size_t *n);
int op_volume_pack_sum(op_volume *v, const int32_t *d_union_keys, size_t n_union, float *d_out);
int op_volume_unpack_sum(op_volume *v, const int32_t *d_union_keys, size_t n_union,
                         const float *d_sum);
/* The same in slices, so that the reduce of later slices overlaps the normalisation of earlier ones: _begin enters the
 * union keys (growing the pool if needed; the volume's own content is dropped only after validation), _chunk writes
 * union blocks [first, first + count) from a buffer holding just those (count x 5 x 512 floats). */
int op_volume_unpack_sum_begin(op_volume *v, const int32_t *d_union_keys, size_t n_union);
int op_volume_unpack_sum_chunk(op_volume *v, size_t first, size_t count, const float *d_sum_chunk);
/* The whole merge as ONE call for a C/C++ host: CubeHandler::Merge (CubeHandler.h:145-167) across the ranks of a
 * communicator.  nccl_comm is an ncclComm_t whose rank owns the volume's device; call from one host thread (or process) per rank.
 * *n_union (may be NULL) = number of blocks in the merged map.  RCCL is bound at run time (dlopen), so the library does not need it
 * until this entry point is used.  Two algorithms (OP_RUNTIME_OPT_MERGE_ALGORITHM):
 *   OP_MERGE_OWNER_EXCHANGE (default): every block key has an owner rank (a hash of the key); each rank sends the blocks it HOLDS, in sum
 *     form, to their owners (one group of ncclSend / ncclRecv over all pairs), the owners add them up in rank order; then
 *       root >= 0: the owners send their partitions to the root, which normalises the whole map into its volume (the others' volumes
 *                  are left untouched) -- the reference's Merge semantics;
 *       root == -1: no gather -- every rank's volume is REPLACED by its owned, merged partition (a distributed map).
 *   OP_MERGE_DENSE_REDUCE: all-gather of the keys, the same sorted union on every rank, every rank packs the whole union (zeros where it
 *     holds nothing), ONE sliced ncclReduce(float32, sum) to `root` (>= 0), normalisation on the root. */
int op_volume_merge_rccl(op_volume *v, void *nccl_comm, int root, size_t *n_union);
/* The same call, reporting what was moved and how the time was spent (for the N > 1 bench line): ranks / rank of the communicator, blocks
 * in the union, the number of reduce slices (dense) and the bytes this rank put into the reduce / the exchange, host wall time of the
 * preparation, of the transfer pipeline, and of the whole call in milliseconds; the algorithm that ran; the blocks this rank HELD, the blocks
 * it OWNS after the exchange (dense: the union on the root, 0 elsewhere) and the payload bytes it sent / received over the wire (keys included;
 * the few-word agreements excluded).  Owner exchange: sent = (blocks held of OTHER ranks' partitions) x 10 248 B (+ its summed partition
 * when a root gathers) -- on average held x 10 248 x (world - 1) / world. */
typedef struct op_merge_stats {
    int32_t ranks, rank;
    uint64_t union_blocks, reduce_bytes, slices;
    double prepare_ms, transfer_ms, total_ms;
    int32_t algorithm, pad_;
    uint64_t held_blocks, owned_blocks, wire_bytes_sent, wire_bytes_received;
} op_merge_stats;
int op_volume_merge_rccl_stats(op_volume *v, void *nccl_comm, int root, size_t *n_union, op_merge_stats *stats);

/* ---- registration (Registration/ICP.h:13-26, RegistrationResult.h:9-16) ------------------- */
typedef struct {
    float T[16];        /* RegistrationResult::T: Kabsch over the final inlier set (ICP.cpp:221) */
    float last_T[16];   /* accumulated start_T after the last iteration (ICP.cpp:198) */
    double rmse;        /* RegistrationResult::rmse (ICP.cpp:206) */
    uint64_t n_inliers; /* correspondence_set_index.size() */
    int32_t iterations;
} op_icp_result;

/* Builds the device search structure over the target cloud (replaces KDTree::BuildTree,
 * ICP.cpp:172-173).  threshold = ICPParameter::threshold (max correspondence distance). */
int op_icp_create(const float *tgt_xyz, const float *tgt_normals, size_t m, double threshold,
                  int mem, int device, op_icp **out);
int op_icp_destroy(op_icp *icp);
/* How the two sequential float32 accumulations of the reference are reproduced.
 *   OP_ICP_OPT_FINISH: the Kabsch fit that becomes RegistrationResult::T (ICP.cpp:215-221 -> Geometry.cpp:117-133).
 *     OP_ICP_FINISH_REFERENCE (default): the final inlier pairs are compacted on the device in ascending source
 *       index and summed by one host thread in float32, pair by pair, exactly like the reference's two loops --
 *       over 3e5 near-planar pairs their rounding is ~1e-3 of T, i.e. it is part of the reference's result.
 *     OP_ICP_FINISH_FP64: order-free fp64 reduction on the device (closer to the exact Kabsch of the pairs).
 *   OP_ICP_OPT_SUMS: the per-iteration sums (JTJ/JTr, ICP.cpp:121-136; the point-to-point Kabsch, :76-79).
 *     (a new context starts in the process-wide OP_RUNTIME_OPT_ICP_DEFAULT_SUMS mode: OP_ICP_SUMS_REFERENCE_F32 unless the host opted into the fp64 reduction)
 *     OP_ICP_SUMS_FP64: fp64 reduction on the device, no host round trip in the point-to-point loop: ~24 000 iterations/s at 307 200 points.  Equal to the
 *       CPU path WITH DOUBLE SUMS to 1e-7; within 1e-4 of the reference's float32 answer only where that answer is stable (2 of the bench's 4 pairs).
 *     OP_ICP_SUMS_REFERENCE_F32 (default): every iteration's inlier rows are put in inlier order on the device and summed
 *       sequentially in float32 as the reference does -- point-to-plane by one wave on the device (36 + 6 accumulators,
 *       one lane each; 42 numbers come back), point-to-point on one host thread after a transfer of the rows.  The
 *       per-iteration inlier counts and the poses then equal the CPU path's at any size and on every frame pair: where
 *       J^T J sits at JacobiSVD's rank threshold the reference's own float32 rounding decides which way its step goes
 *       (its pose moves by up to 5e-2 when the same sums are taken in double), and only this mode follows it there.
 *       ~600 iterations/s at 307 200 points against ~24 000 with the fp64 reduction.  No middle way exists: tests/tools/icp_sigma_probe.py
 *       (profiles/r06_icp_sigma_probe.txt) shows the smallest eigenvalue of the float32 J^T J is summation noise in EVERY iteration, so no test on the fp64
 *       sums can tell which iterations need the reference's order.
 *   OP_ICP_OPT_TIES: which target a source point is paired with when its nearest candidates are EXACTLY equidistant (duplicated
 *     target points, clouds on a lattice, quantised coordinates; on depth-derived clouds a chance event of ~1e-7 per query).
 *     OP_ICP_TIES_REFERENCE (default): the target the reference's kd-tree (nanoflann 1.3.2 as Geometry/KDTree.h:62-98,171-190 drives it)
 *       returns: the candidate its depth-first traversal meets first.  The search marks the tied queries (one compare and one select per
 *       candidate: +2-4 % on the iteration kernel); their number comes back with the sums and their records through host-mapped memory.
 *       Only for them does the host repeat the search, in the tree nanoflann would build (onepiece_nanotree.hpp; built on the first tie,
 *       ~1 us per query afterwards), and where the partner changes it exchanges the pair's contribution to the sums (same float
 *       expressions as the kernel's) and patches the stored correspondence; floods of ties (lattices) and the reference-order summation
 *       mode take the sums again on the device instead.  Measured at 307 200 points: a depth-derived pair has 0-3 tied queries per
 *       pass (two float32 squared distances equal in every bit); 60 iterations 2.32 ms against 2.25 ms.  op_icp_tie_stats counts them.
 *     OP_ICP_TIES_LOWEST_INDEX: the smallest target index among the equidistant ones -- what the grid search yields without the marking. */
#define OP_ICP_OPT_FINISH 0
#define OP_ICP_OPT_SUMS 1
#define OP_ICP_OPT_TIES 2
#define OP_ICP_FINISH_REFERENCE 0
#define OP_ICP_FINISH_FP64 1
#define OP_ICP_SUMS_FP64 0
#define OP_ICP_SUMS_REFERENCE_F32 1
#define OP_ICP_TIES_LOWEST_INDEX 0
#define OP_ICP_TIES_REFERENCE 1
int op_icp_set_option(op_icp *icp, int option, int value);
int op_icp_set_source(op_icp *icp, const float *src_xyz, size_t n, int mem);
/* OP_ICP_TIES_REFERENCE: queries with exactly equidistant nearest candidates seen since op_icp_create (summed over passes), and how many of
 * them the reference pairs with another target than the smallest index. */
int op_icp_tie_stats(op_icp *icp, uint64_t *tied_queries, uint64_t *changed);
/* The final CountInliers of a registration (ICP.cpp:206) measures the LAST search's correspondences with the pose the last solve produced.  The
 * grid search vouches for a correspondence only within one cell edge (a little over the threshold) of its query -- all that matters while search
 * and count share a pose -- so op_icp_run's final pass singles out the source points whose stored partner lies beyond that under the search's pose
 * AND that the last step moved far enough to bridge the gap, and re-decides exactly those in the tree the reference would search (none in a
 * converged registration; many when the loop is stopped after one or two large steps).  redecided: how many since op_icp_create. */
int op_icp_final_stats(op_icp *icp, uint64_t *redecided);
/* One loop body of ICP.cpp:177-199 without the solve: transform by T, 1-NN, CountInliers and the
 * normal-equation sums.  mode PLANE: sums[0..35] = JTJ (row-major 6x6), sums[36..41] = JTr.
 * mode POINT: sums[0..2] = sum s', [3..5] = sum t, [6..14] = sum s' t^T (row-major), s' = T*s. */
int op_icp_iterate(op_icp *icp, const float T[16], int mode, double sums[42], uint64_t *n_inliers,
                   double *sum_sq_err);
/* registration::PointToPlane / PointToPoint end to end (ICP.cpp:146-224 / :31-107).
 * pairs (cap x 2 int32: source id, target id, ascending source id), per_iter_inliers (max_iter)
 * and per_iter_T (max_iter x 16) may be NULL. */
int op_icp_run(op_icp *icp, int mode, const float init_T[16], int max_iteration,
               op_icp_result *result, int32_t *pairs, size_t pairs_cap, int32_t *per_iter_inliers,
               float *per_iter_T);
/* The same run split in two, so that several contexts work on independent frame pairs at once -- ICP shards only as REPLICAS,
 * and one 307 200-point problem cannot fill the chip (28 k iterations/s = 4 % of the HBM rate its 11 MB would allow).  The loop needs
 * the host after every iteration (the 6x6 solve), so an enqueued run proceeds on a host thread of the context's own; `result` and `pairs`
 * (may be NULL) must stay valid until op_icp_wait, which returns the run's status.  One run may be outstanding per context; results are
 * those of op_icp_run (each context is independent of the others). */
int op_icp_run_enqueue(op_icp *icp, int mode, const float init_T[16], int max_iteration, op_icp_result *result, int32_t *pairs, size_t pairs_cap);
int op_icp_wait(op_icp *icp);
/* K registrations on K contexts from ONE host thread (ICP does not shard -- replicas only).  Contexts in the fp64-reduction mode are pipelined by the
 * caller's thread: every context has one iteration in flight, the thread goes round waiting for a context's sums (host-mapped memory), solving, enqueueing its next
 * iteration; the finishes run side by side on helper threads.  Contexts in the reference-order mode get a submitter thread each (they synchronise every iteration).
 * init_T: K x 16 floats, or NULL for the identity everywhere; results: K entries; every context's result equals what op_icp_run gives it alone. */
int op_icp_run_many(op_icp *const *icps, int k, int mode, const float *init_T, int max_iteration, op_icp_result *results);

/* Convenience: create + set_source + run + destroy with host buffers. */
int op_icp_register(int mode, const float *src_xyz, size_t n, const float *tgt_xyz,
                    const float *tgt_normals, size_t m, const float init_T[16], int max_iteration,
                    double threshold, int device, op_icp_result *result, int32_t *pairs,
                    size_t pairs_cap);
/* registration::EstimateRigidTransformationPointToPlane (ICP.h:24-26, ICP.cpp:108-144): one Gauss-Newton
 * step over caller-supplied inliers (n x 2 int32: source id, target id); source = the ALREADY transformed
 * points, as the reference's loop passes them.  Sums on the device (fp64), JacobiSVD-semantics solve +
 * SE3 exp on the host. */
int op_estimate_rigid_point_to_plane(const float *source_xyz, size_t n_source, const float *target_xyz,
                                     const float *target_normals, size_t n_target,
                                     const int32_t *inliers, size_t n_inliers, int mem, int device,
                                     float T[16]);
/* The same with the accumulation chosen: sums = OP_ICP_SUMS_FP64 (what the plain entry does) or
 * OP_ICP_SUMS_REFERENCE_F32 (JTJ/JTr summed sequentially in float32 on one host thread, ICP.cpp:121-136). */
int op_estimate_rigid_point_to_plane_ex(const float *source_xyz, size_t n_source, const float *target_xyz,
                                        const float *target_normals, size_t n_target,
                                        const int32_t *inliers, size_t n_inliers, int mem, int device,
                                        int sums, float T[16]);
/* geometry::EstimateRigidTransformation (Geometry/Geometry.cpp:107-151): Kabsch fit of a correspondence set
 * given as n x 6 floats (source xyz, target xyz).  The plain entry sums as the reference does (sequential
 * float32 in pair order, OP_ICP_FINISH_REFERENCE); _ex selects OP_ICP_FINISH_FP64 (order-free device reduction). */
int op_estimate_rigid_transformation(const float *pairs_xyz6, size_t n_pairs, int mem, int device,
                                     float T[16]);
int op_estimate_rigid_transformation_ex(const float *pairs_xyz6, size_t n_pairs, int mem, int device,
                                        int finish, float T[16]);
/* tool::ConvertDepthTo32F + tool::BilateralFilter (Tool/ImageProcessing.cpp:68-91, 64-67; header default
 * range = 7, Tool/ImageProcessing.h:19), the depth preprocessing every fusion driver runs right before
 * IntegrateImage (example/ImageSequenceIntegration.cpp:36-38, DenseFusion/DenseFusion.cpp:92-94,
 * ImageIntegration.cpp:24-27): cv::bilateralFilter(src, dst, d, sigma_color = 0.03, sigma_space = 4.5) on
 * n_images row-major width x height depth images (OP_DEPTH_U16 input is first divided by depth_scale).
 * PARITY NOTE: OpenCV is not vendored by the reference; this is cv::bilateralFilter's documented CV_32FC1
 * definition, unpinned:  radius = d / 2 (d <= 0: round(1.5 sigma_space); at least 1), sigma <= 0 -> 1,
 *     out(p) = sum_q w(q) src(q) / sum_q w(q),  q = p + (i, j) with i*i + j*j <= radius^2, BORDER_REFLECT_101,
 *     w(q) = exp(-(i*i + j*j) / (2 sigma_space^2)) * exp(-(src(q) - src(p))^2 / (2 sigma_color^2)),
 * in float32, taps summed row by row.  (OpenCV evaluates the second factor through a 4096-bin linearly
 * interpolated table over the image's value range; this evaluates it directly.)
 * stream: NULL -> the call runs on the library's stream and returns when `out` is final; a hipStream_t
 * (OP_MEM_DEVICE only) -> only enqueued on that stream, e.g. op_volume_stream() so that the filtered depth
 * feeds op_volume_integrate without a host synchronisation. */
int op_bilateral_filter_depth(const void *depth, int depth_format, float depth_scale, int width, int height,
                              int n_images, int d, float sigma_color, float sigma_space, int mem, int device,
                              void *stream, float *out);

/* PointCloud::LoadFromDepth (Geometry/PointCloud.cpp:72-100) on the device: xyz_out (mem) gets the
 * compacted row-major-ordered points; *n the count. */
int op_points_from_depth(const op_camera *cam, const void *depth, int depth_fmt, int mem, int device,
                         float *xyz_out, size_t *n);

/* PointCloud::LoadFromRGBD (Geometry/PointCloud.cpp:17-48): the same compaction plus colours =
 * (b0,b1,b2)/255 in stored channel order (n x 3 floats). */
int op_points_from_rgbd(const op_camera *cam, const void *depth, int depth_fmt, const uint8_t *rgb,
                        int mem, int device, float *xyz_out, float *colors_out, size_t *n);

/* PointCloud::EstimateNormals(radius, knn) (Geometry/PointCloud.cpp:102-144) on the device: exact
 * knn nearest neighbours (knn <= 32), the prefix whose SQUARED distance is <= radius (the
 * reference's KnnRadiusSearch, KDTree.h:230-255), PCA plane fit (geometry::FitPlane,
 * Geometry.cpp:172-218).  normals_out: n x 3 floats in `mem`; the sign of each normal is
 * undetermined (as in the reference, whose disambiguation is #if 0'd out); < 3 neighbours -> 0. */
int op_estimate_normals(const float *xyz, size_t n, float radius, int knn, int mem, int device,
                        float *normals_out);

/* ---- dense RGB-D tracker (Odometry/Odometry.h:38-175) ------------------------
 * Boundary = the inputs of Odometry::MultiScaleComputing (Odometry.cpp:621-636): image pyramids
 * already built (cv::pyrDown / cv::Sobel / cv::GaussianBlur stay where they are in the reference,
 * Odometry.cpp:436-449,609-620).  All images are dense row-major float32 of width x height
 * (CV_32FC1); invalid depth is NaN (ConvertDepthTo32FNaN, DenseOdometryFunction.cpp:28-56); the
 * *_dx/_dy images are the raw cv::Sobel outputs (SOBEL_SCALE is applied inside).  The image-XYZ
 * pyramids of the reference are recomputed from the depth pyramids with TransformToMatXYZ's own
 * arithmetic (Geometry.cpp:72-106).  levels[0] is full resolution (camera_pyramid[0]). */
typedef struct {
    int32_t width, height;          /* camera_pyramid[l].GetWidth()/GetHeight() */
    float fx, fy, cx, cy;           /* Camera.h:38-42: halved per level */
    const float *source_color, *source_depth, *target_color, *target_depth;
    const float *target_color_dx, *target_color_dy, *target_depth_dx, *target_depth_dy;
} op_track_level;

typedef struct {
    float T[16];                 /* DenseTrackingResult::T (Odometry.h:32) */
    double rmse;                 /* ComputeReprojectionError3D(correspondence_set, T), Odometry.cpp:606 */
    uint64_t n_correspondences;  /* pixel_correspondence_set.size(): pairs of the LAST executed iteration */
    int32_t tracking_success;    /* ratio >= MIN_INLIER_RATIO_DENSE over the FULL-resolution pixel count */
    int32_t iterations;          /* iterations executed (early-out at ratio > MAX_INLIER_RATIO_DENSE) */
} op_track_result;

int op_tracker_create(int device, op_tracker **out);
int op_tracker_destroy(op_tracker *t);
/* OP_TRACK_OPT_SUMS: how an iteration's normal equations are summed (DenseOdometryFunction.cpp:297-381).
 *   (a new tracker starts in the process-wide OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS mode: OP_TRACK_SUMS_REFERENCE_F32 unless the host opted into the fp64 reduction)
 *   OP_TRACK_SUMS_FP64: fp64 reduction on the device, the whole coarse-to-fine loop without a host round trip.
 *   OP_TRACK_SUMS_REFERENCE_F32: the reference's own sums -- association, acceptance and Jacobian rows come from the same kernels, and every
 *     iteration's rows are summed in raster order in float32 exactly like the reference's loop: on the device, by one wave that owns the
 *     36 + 6 accumulators and walks the compacted rows (k_seq_sums; the other waves of its workgroup prepare the products), then the 42
 *     sums go to the host for the LDL^T solve / exp / pose update the reference-order mode shares with the CPU path.  A run follows the
 *     CPU path step for step: identical correspondence counts at every iteration, poses to 1e-6 (north_star's bar is 1e-4), at ~9 ms
 *     per 640x480 track.  NormalizeIntensity's two means likewise.
 *   OP_TRACK_SUMS_REFERENCE_F32_HOST: the same sums taken on ONE host thread after a transfer of all rows (17 MB per full-resolution
 *     iteration; ~30 tracks/s) -- the cross-check of the device sums: both give the same floats. */
#define OP_TRACK_OPT_SUMS 0
#define OP_TRACK_SUMS_FP64 0
#define OP_TRACK_SUMS_REFERENCE_F32 1
#define OP_TRACK_SUMS_REFERENCE_F32_HOST 2
int op_tracker_set_option(op_tracker *t, int option, int value);
/* Odometry::MultiScaleComputing + the result assembly of DenseTracking (Odometry.cpp:621-687,
 * :600-607).  iters_per_level[l] = iter_count_per_level[l] (Odometry.h:170, default {4,8,16});
 * levels are visited n_levels-1 .. 0.  full_width/full_height = camera.GetWidth()/GetHeight(), the
 * denominator of both inlier ratios at EVERY level (Odometry.cpp:635,669,686).
 * Optional outputs (may be NULL): pixel_corr (corr_cap x 4 int32 {v_s,u_s,v_t,u_t}, raster order of
 * the source pixel = the reference's push_back order), point_corr (corr_cap x 6 float: source xyz,
 * target xyz, both read at the SOURCE pixel of level 0 as Odometry.cpp:676-683 does),
 * per_iter_count (sum(iters) int32) and per_iter_T (sum(iters) x 16): the correspondence count used
 * by, and the pose after, each executed iteration. */
int op_tracker_track(op_tracker *t, const op_track_level *levels, int n_levels,
                     const int32_t *iters_per_level, int full_width, int full_height, int term_type,
                     const float init_T[16], int mem, op_track_result *result, int32_t *pixel_corr,
                     float *point_corr, size_t corr_cap, int32_t *per_iter_count, float *per_iter_T);
/* ComputeCorrespondencePixelWise (DenseOdometryFunction.cpp:72-128) alone, incl. its source-indexed
 * "z-buffer" (:9-27): what DenseTracking runs with the identity pose before NormalizeIntensity
 * (Odometry.cpp:543-544). */
int op_tracker_correspondences(op_tracker *t, const op_track_level *level, const float T[16], int mem,
                               int32_t *pixel_corr, size_t corr_cap, size_t *n);
/* Odometry::DenseTracking, cv::Mat overload (Odometry.cpp:463-524), end to end on the device:
 * InitializeRGBDDenseTracking (:609-620) for both frames, ComputeCorrespondencePixelWise at the
 * identity + NormalizeIntensity (:543-544, DenseOdometryFunction.cpp:129-145), CreatePyramidCameras,
 * CreateImagePyramid (:436-449), MultiScaleComputing.  rgb: 3 bytes / pixel in stored order; depth:
 * OP_DEPTH_F32 metres or OP_DEPTH_U16 raw (divided by cam->depth_scale).  n_levels = multi_scale_level.
 * PARITY NOTE: the reference's cvtColor / GaussianBlur / pyrDown / Sobel are OpenCV calls (not vendored);
 * this entry point implements their published definitions (BORDER_REFLECT_101, float) and is checked
 * against a restatement of those definitions, not against OpenCV.  A caller who needs the reference's
 * exact OpenCV pyramids builds them on the host and calls op_tracker_track. */
int op_tracker_dense_tracking(op_tracker *t, const op_camera *cam, int n_levels,
                              const int32_t *iters_per_level, const uint8_t *source_rgb,
                              const uint8_t *target_rgb, const void *source_depth,
                              const void *target_depth, int depth_fmt, const float init_T[16],
                              int term_type, int mem, op_track_result *result, int32_t *pixel_corr,
                              float *point_corr, size_t corr_cap);
/* The same call split in two so that several trackers (each owns a HIP stream) can work on independent
 * frame pairs concurrently -- a single 640x480 track is 28 strictly sequential small problems that leave
 * most of the 256 CUs idle.  _enqueue returns as soon as everything is on the tracker's stream (device
 * frames are used in place and must stay valid until the wait); op_tracker_wait synchronises that stream
 * and fills the result.  One enqueue may be outstanding per tracker.  With the reference-order sums
 * (OP_TRACK_SUMS_REFERENCE_F32*), which need the host after every iteration, the run proceeds on a host thread of
 * the tracker's own, so _enqueue still returns at once and trackers still overlap; HOST frames must then stay
 * valid until the wait as well. */
int op_tracker_dense_tracking_enqueue(op_tracker *t, const op_camera *cam, int n_levels,
                                      const int32_t *iters_per_level, const uint8_t *source_rgb,
                                      const uint8_t *target_rgb, const void *source_depth,
                                      const void *target_depth, int depth_fmt, const float init_T[16],
                                      int term_type, int mem, int want_point_corr);
int op_tracker_wait(op_tracker *t, op_track_result *result, int32_t *pixel_corr, float *point_corr,
                    size_t corr_cap);
/* Reads back an image prepared by the last op_tracker_dense_tracking call: frame 0 source / 1 target;
 * kind 0 colour, 1 depth, 2 colour_dx, 3 colour_dy, 4 depth_dx, 5 depth_dy (derivatives: target only). */
int op_tracker_read_pyramid(op_tracker *t, int frame, int kind, int level, float *out, size_t cap);
/* Convenience: create + track + destroy. */
int op_dense_track(const op_track_level *levels, int n_levels, const int32_t *iters_per_level,
                   int full_width, int full_height, int term_type, const float init_T[16], int mem,
                   int device, op_track_result *result, int32_t *pixel_corr, float *point_corr,
                   size_t corr_cap);
/* Host-side arithmetic of that path, exported so it can be checked without a GPU:
 * K*R*K.inverse() and K*t as Eigen 3.3.7 evaluates them (DenseOdometryFunction.cpp:82-87; cam4 =
 * {fx,fy,cx,cy}, row-major 3x3 out) and JTJ.ldlt().solve(-JTr) (:404; JTJ row-major 6x6). */
int op_track_projection(const float cam4[4], const float T[16], float KRK_inv[9], float Kt[3]);
int op_ldlt_solve6(const double JTJ[36], const double JTr[6], float x[6]);

#ifdef __cplusplus
}
#endif
#endif /* ONEPIECE_HIP_H */
