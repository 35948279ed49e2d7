// CubeHandler.cpp -- integration::CubeHandler over the C-ABI (include/onepiece_hip.h).  See the header for the contract.
#include "Integration/CubeHandler.h"

#include <cmath>
#include <unordered_set>

#include "Bridge.h"
#include "DeviceFrame.h"

namespace one_piece {
namespace integration {

using bridge::Failed;

void CubeHandler::Report(const char* where) { std::cout << RED << "[ERROR]::[CubeHandler::" << where << "]::" << op_last_error() << RESET << std::endl; }

bool CubeHandler::Ensure() const {
    if (vol) return true;
    const int rc = op_volume_create(&camera.Pod(), c_para.VoxelResolution, integrator.truncation, far, near, bridge::Device(), 0, &vol);
    if (rc != OP_OK) { Report("create"); vol = nullptr; return false; }
    return true;
}

CubeHandler::CubeHandler() : cube_map(this) { c_para.InitializeVoxelCube(); }
CubeHandler::CubeHandler(const camera::PinholeCamera& _camera) : camera(_camera), cube_map(this) { c_para.InitializeVoxelCube(); }

// ---- the protected cube_map mirror -----------------------------------------------------------------------------------
const CubeMap& CubeMapMirror::get() const {
    if (!edited && seen != owner->changes) { // stale: the volume changed on the device since the last look
        host = owner->GetCubeMap();          // (commits nothing: `edited` is false here)
        seen = owner->changes;
    }
    return host;
}
CubeMap& CubeMapMirror::edit() {
    get();
    edited = true;
    return host;
}
void CubeHandler::Pending() const {
    if (cube_map.edited) const_cast<CubeHandler*>(this)->CommitCubeMap();
}
void CubeHandler::CommitCubeMap() {
    if (!cube_map.edited) return;
    cube_map.edited = false; // first: the upload below goes through members that call Pending()
    if (!Ensure()) return;
    if (op_volume_clear(vol) != OP_OK) { Report("CommitCubeMap"); return; }
    std::vector<int32_t> keys;
    std::vector<float> vox;
    keys.reserve(3 * cube_map.host.size());
    vox.reserve(cube_map.host.size() * 512 * 5);
    for (CubeMap::const_iterator it = cube_map.host.begin(); it != cube_map.host.end(); ++it) {
        for (int k = 0; k < 3; ++k) keys.push_back(it->first(k));
        for (int v = 0; v < 512; ++v) {
            const TSDFVoxel& t = it->second.voxels[v];
            const float rec[5] = {t.sdf, t.weight, t.color(0), t.color(1), t.color(2)};
            vox.insert(vox.end(), rec, rec + 5);
        }
    }
    if (!keys.empty() && op_volume_upload(vol, keys.data(), vox.data(), keys.size() / 3) != OP_OK) Report("CommitCubeMap");
    Touch();
    cube_map.seen = changes; // the mirror IS the volume's content
}

CubeHandler::CubeHandler(op_volume* adopted, const CubeHandler& like, float resolution)
    : camera(like.camera), integrator(like.integrator), far(like.far), near(like.near), cube_map(this), vol(adopted) {
    c_para.VoxelResolution = resolution;
    c_para.InitializeVoxelCube();
}

// value semantics: the copy owns its own device volume with the same content (Merge into an empty volume copies
// every block verbatim)
CubeHandler::CubeHandler(const CubeHandler& other)
    : camera(other.camera), integrator(other.integrator), c_para(other.c_para), far(other.far), near(other.near), cube_map(this) {
    other.Pending();
    if (other.vol && Ensure() && op_volume_merge(vol, other.vol) != OP_OK) Report("copy");
}
CubeHandler& CubeHandler::operator=(const CubeHandler& other) {
    if (this == &other) return *this;
    other.Pending();
    cube_map.edited = false; // whatever was pending here is overwritten
    Touch();
    if (vol) { op_volume_destroy(vol); vol = nullptr; borrowed_.clear(); } // (the destroy synchronised: nothing reads the borrowed device frames any more)
    camera = other.camera; integrator = other.integrator; c_para = other.c_para; far = other.far; near = other.near;
    if (other.vol && Ensure() && op_volume_merge(vol, other.vol) != OP_OK) Report("assign");
    return *this;
}
CubeHandler::~CubeHandler() {
    if (vol) op_volume_destroy(vol);
}

op_volume* CubeHandler::Handle() const { Pending(); Touch(); return Ensure() ? vol : nullptr; } // (the caller may change the volume through the handle)

void CubeHandler::SetVoxelResolution(float resolution) {
    Pending();
    c_para.SetVoxelResolution(resolution);
    if (vol && op_volume_set_resolution(vol, resolution) != OP_OK) Report("SetVoxelResolution");
}
void CubeHandler::SetTruncation(float trunc) {
    integrator.SetTruncation(trunc);
    if (vol && op_volume_set_truncation(vol, trunc) != OP_OK) Report("SetTruncation");
}
void CubeHandler::SetCamera(const camera::PinholeCamera& _camera) {
    camera = _camera;
    if (vol && op_volume_set_camera(vol, &camera.Pod()) != OP_OK) Report("SetCamera");
}
void CubeHandler::SetFarPlane(float _far) {
    far = _far;
    if (vol && op_volume_set_near_far(vol, near, far) != OP_OK) Report("SetFarPlane");
}
void CubeHandler::SetNearPlane(float _near) {
    near = _near;
    if (vol && op_volume_set_near_far(vol, near, far) != OP_OK) Report("SetNearPlane");
}

void CubeHandler::Clear() {
    cube_map.edited = false; // pending edits would be wiped with the volume
    Touch();
    if (vol && op_volume_clear(vol) != OP_OK) Report("Clear");
}
bool CubeHandler::HasCube(const CubeID& cube_id) const {
    Pending();
    if (!vol) return false;
    int present = 0;
    if (op_volume_has_cube(vol, cube_id(0), cube_id(1), cube_id(2), &present) != OP_OK) Report("HasCube");
    return present != 0;
}
size_t CubeHandler::GetCubeCount() const {
    Pending();
    size_t n = 0;
    if (vol && op_volume_block_count(vol, &n) != OP_OK) Report("GetCubeCount");
    else ReleaseBorrowed(); // op_volume_block_count synchronises
    return n;
}
void CubeHandler::Synchronize() const {
    Pending();
    if (vol && op_volume_sync(vol) != OP_OK) Report("Synchronize");
    else ReleaseBorrowed();
}

void CubeHandler::AddCube(const CubeID& cube_id) {
    Pending();
    if (!Ensure() || HasCube(cube_id)) return;
    Touch();
    const int32_t key[3] = {cube_id(0), cube_id(1), cube_id(2)};
    std::vector<float> fresh(512 * 5);
    for (int v = 0; v < 512; ++v) { fresh[5 * v] = 999; fresh[5 * v + 1] = 0; fresh[5 * v + 2] = fresh[5 * v + 3] = fresh[5 * v + 4] = -1; }
    if (op_volume_upload(vol, key, fresh.data(), 1) != OP_OK) Report("AddCube");
}

// CubeHandler.h:199-241 of the reference: ALLOCATE (AddCube) the blocks that the voxel centres of `v_cube` land in after
// `trans` -- the eight trilinear neighbours of the moved centre, or the one voxel that contains it.  The arithmetic is the
// reference's (4x4 * (p,1) accumulated column by column, division by w, float division by the resolution, floor).
void CubeHandler::AddTransformedCubes(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans, bool nearest) {
    if (!Ensure()) return;
    const float res = c_para.VoxelResolution, half = res / 2;
    std::unordered_set<CubeID, CubeHasher> wanted;
    for (size_t voxel_id = 0; voxel_id != v_cube.voxels.size(); ++voxel_id) {
        geometry::Point3 p = geometry::TransformPoint(trans, c_para.GetGlobalPoint(v_cube.cube_id, static_cast<int>(voxel_id)));
        if (!nearest) p = p - geometry::Point3(half, half, half);
        const geometry::Point3i base(static_cast<int>(std::floor(p(0) / res)), static_cast<int>(std::floor(p(1) / res)), static_cast<int>(std::floor(p(2) / res)));
        for (int n = 0; n < (nearest ? 1 : 8); ++n)
            wanted.insert(c_para.GetCubeID(geometry::Point3i(base(0) + (n & 1), base(1) + ((n >> 1) & 1), base(2) + (n >> 2))));
    }
    for (std::unordered_set<CubeID, CubeHasher>::const_iterator it = wanted.begin(); it != wanted.end(); ++it) AddCube(*it);
}
void CubeHandler::AddTransformedCube(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans) { AddTransformedCubes(v_cube, trans, false); }
void CubeHandler::AddTransformedCubeNearest(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans) { AddTransformedCubes(v_cube, trans, true); }

void CubeHandler::ComputeBounding(const cv::Mat& depth, const geometry::TransformationMatrix& pose, geometry::Point3& max_pos, geometry::Point3& min_pos) {
    Pending();
    if (!Ensure()) return;
    float p[16], mx[3], mn[3];
    bridge::RowMajor(pose, p);
    size_t inside = 0;
    if (op_volume_compute_bounding(vol, depth.data, bridge::DepthFormat(depth), OP_MEM_HOST, p, mx, mn, &inside) != OP_OK) { Report("ComputeBounding"); return; }
    max_pos = geometry::Point3(mx[0], mx[1], mx[2]);
    min_pos = geometry::Point3(mn[0], mn[1], mn[2]);
}

void CubeHandler::PrepareCubes(const cv::Mat& depth, const geometry::TransformationMatrix& pose, std::vector<CubeID>& cube_id_list) {
    cube_id_list.clear();
    Pending();
    if (!Ensure()) return;
    Touch(); // PrepareCubes allocates the blocks it selects (CubeHandler.cpp:181-190)
    float p[16], pi[16];
    bridge::RowMajor(pose, p);
    bridge::RowMajor(pose.inverse(), pi); // the caller's own Eigen inverse when built with Eigen (Integrator.cpp:18)
    std::vector<int32_t> ids(3 * 65536);
    size_t n = 0;
    int rc = op_volume_prepare_cubes(vol, depth.data, bridge::DepthFormat(depth), OP_MEM_HOST, p, pi, ids.data(), ids.size() / 3, &n, nullptr);
    if (rc == OP_OK && n > ids.size() / 3) { // the blocks are allocated already: the second call only lists them
        ids.resize(3 * n);
        rc = op_volume_prepare_cubes(vol, depth.data, bridge::DepthFormat(depth), OP_MEM_HOST, p, pi, ids.data(), n, &n, nullptr);
    }
    if (rc != OP_OK) { Report("PrepareCubes"); return; }
    cube_id_list.reserve(n);
    for (size_t i = 0; i < n; ++i) cube_id_list.push_back(CubeID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
}

void CubeHandler::IntegrateImage(const cv::Mat& depth, const cv::Mat& rgb, const geometry::TransformationMatrix& pose) {
    Pending();
    if (!Ensure()) return;
    Touch();
    float p[16], pi[16];
    bridge::RowMajor(pose, p);
    bridge::RowMajor(pose.inverse(), pi);
    // the images are only borrowed for the call: the library copies them into its pinned staging ring before returning
    if (op_volume_integrate(vol, depth.data, bridge::DepthFormat(depth), rgb.data, OP_MEM_HOST, p, pi) != OP_OK) { Report("IntegrateImage"); return; }
    // the reference prints two lines per frame here (CubeHandler.cpp:202-209); at > 10 k frames/s that would be the
    // bottleneck, so the line is opt-in (ONEPIECE_HIP_VERBOSE=1) and says what is true: the frame is queued
    static const bool verbose = std::getenv("ONEPIECE_HIP_VERBOSE") != nullptr;
    if (verbose) std::cout << GREEN << "[IntegrateImage]::[Info]::Image queued for integration." << RESET << std::endl;
}
void CubeHandler::IntegrateImage(const geometry::RGBDFrame& rgbd, const geometry::TransformationMatrix& pose) {
    if (!rgbd.on_device) { IntegrateImage(rgbd.depth, rgbd.rgb, pose); return; }
    // the frame's images are on the device already (the tracker put them there): fused in place, nothing crosses PCIe again
    Pending();
    if (!Ensure()) return;
    Touch();
    std::shared_ptr<bridge::DeviceImages> d = std::static_pointer_cast<bridge::DeviceImages>(rgbd.on_device);
    // the device copy was made for the tracker's camera: it is fused in place only if it is what THIS volume reads -- its camera's size, its device
    if (d->width != camera.GetWidth() || d->height != camera.GetHeight() || d->device != bridge::Device()) {
        if (rgbd.depth.empty() || rgbd.rgb.empty()) {
            std::cout << RED << "[ERROR]::[IntegrateImage]::the frame's device copy (" << d->width << " x " << d->height << ", device " << d->device
                      << ") does not match the volume's camera (" << camera.GetWidth() << " x " << camera.GetHeight() << ") and the frame has no host images" << RESET << std::endl;
            return;
        }
        IntegrateImage(rgbd.depth, rgbd.rgb, pose);
        return;
    }
    float p[16], pi[16];
    bridge::RowMajor(pose, p);
    bridge::RowMajor(pose.inverse(), pi);
    // the volume reads the device images when their batch is launched (and again if it is replayed after a pool growth): they are held here
    // until the volume reports their batch complete (op_volume_progress: no waiting) -- about three batches' worth of frames in steady state
    uint64_t accepted = 0, done = 0;
    if (op_volume_progress(vol, &accepted, &done) != OP_OK) { Report("IntegrateImage"); return; }
    while (!borrowed_.empty() && borrowed_.front().first < done) borrowed_.pop_front();
    if (op_volume_integrate(vol, d->depth, d->depth_fmt, static_cast<const uint8_t*>(d->rgb), OP_MEM_DEVICE, p, pi) != OP_OK) { Report("IntegrateImage"); return; }
    borrowed_.push_back(std::make_pair(static_cast<unsigned long long>(accepted), rgbd.on_device)); // this frame is number `accepted` (0-based)
}

void CubeHandler::Merge(const CubeHandler& another) {
    if (c_para.VoxelResolution != another.c_para.VoxelResolution) {
        std::cout << YELLOW << "[Warning]::[MergeVoxelHash]::Voxel resolution is not identical." << RESET << std::endl;
        return;
    }
    Pending();
    another.Pending();
    if (!another.vol || !Ensure()) return;
    Touch();
    if (op_volume_merge(vol, another.vol) != OP_OK) Report("Merge");
}
void CubeHandler::Merge(const CubeHandler& another, const geometry::TransformationMatrix& trans) {
    if (c_para.VoxelResolution != another.c_para.VoxelResolution) {
        std::cout << YELLOW << "[Warning]::[MergeVoxelHash]::Voxel resolution is not identical." << RESET << std::endl;
        return;
    }
    std::shared_ptr<CubeHandler> moved = another.Transform(trans);
    if (moved) Merge(*moved);
}

std::shared_ptr<CubeHandler> CubeHandler::Transform(const geometry::TransformationMatrix& trans) const {
    Pending();
    if (!Ensure()) return std::shared_ptr<CubeHandler>();
    float T[16], Ti[16];
    bridge::RowMajor(trans, T);
    bridge::RowMajor(trans.inverse(), Ti);
    op_volume* out = nullptr;
    if (op_volume_transform(vol, T, Ti, 0, 0, &out) != OP_OK) { Report("Transform"); return std::shared_ptr<CubeHandler>(); }
    return std::shared_ptr<CubeHandler>(new CubeHandler(out, *this, c_para.VoxelResolution));
}
std::shared_ptr<CubeHandler> CubeHandler::TransformNearest(const geometry::TransformationMatrix& trans) {
    Pending();
    if (!Ensure()) return std::shared_ptr<CubeHandler>();
    float T[16], Ti[16];
    bridge::RowMajor(trans, T);
    bridge::RowMajor(trans.inverse(), Ti);
    op_volume* out = nullptr;
    if (op_volume_transform(vol, T, Ti, 1, 0, &out) != OP_OK) { Report("TransformNearest"); return std::shared_ptr<CubeHandler>(); }
    // the reference does not hand its c_para to the result (CubeHandler.h:299-305): it keeps the default resolution
    return std::shared_ptr<CubeHandler>(new CubeHandler(out, *this, CubePara().VoxelResolution));
}

std::shared_ptr<geometry::PointCloud> CubeHandler::GetPointCloud() const {
    std::shared_ptr<geometry::PointCloud> pcd = std::make_shared<geometry::PointCloud>();
    Pending();
    if (!vol) return pcd;
    size_t n = 0;
    if (op_volume_point_cloud(vol, nullptr, nullptr, 0, &n) != OP_OK) { Report("GetPointCloud"); return pcd; }
    pcd->points.resize(n);
    pcd->colors.resize(n);
    if (n && op_volume_point_cloud(vol, bridge::Floats(pcd->points), bridge::Floats(pcd->colors), n, &n) != OP_OK) { Report("GetPointCloud"); pcd->Reset(); }
    return pcd;
}

namespace {
void AppendMesh(op_volume* vol, const int32_t* only_block, geometry::TriangleMesh& mesh, const char* where) {
    const int *tri = nullptr, *edges = nullptr;
    GetMarchingCubeTables(&tri, &edges);
    size_t n = 0;
    if (Failed(op_volume_extract_mesh(vol, tri, edges, only_block, nullptr, nullptr, 0, &n), where) || n == 0) return;
    std::vector<float> p(3 * n), c(3 * n);
    if (Failed(op_volume_extract_mesh(vol, tri, edges, only_block, p.data(), c.data(), n, &n), where)) return;
    unsigned index = static_cast<unsigned>(mesh.points.size()); // triangles index the running vertex list (MarchingCube.cpp:39)
    for (size_t k = 0; k < n; ++k) {
        mesh.points.push_back(geometry::Point3(p[3 * k], p[3 * k + 1], p[3 * k + 2]));
        mesh.colors.push_back(geometry::Point3(c[3 * k], c[3 * k + 1], c[3 * k + 2]));
        if (k % 3 == 2) { mesh.triangles.push_back(geometry::Point3ui(index, index + 1, index + 2)); index += 3; }
    }
}
} // namespace

void CubeHandler::ExtractTriangleMesh(geometry::TriangleMesh& mesh) {
    mesh.Reset(); // CubeHandler.cpp:11
    Pending();
    if (!vol) return;
    AppendMesh(vol, nullptr, mesh, "ExtractTriangleMesh");
    std::cout << BLUE << "[ExtractTriangleMesh]::[INFO]::Finish Extracting Mesh, " << mesh.triangles.size() << " triangles." << RESET << std::endl;
}
void CubeHandler::GenerateMeshByCube(const CubeID& cube_id, geometry::TriangleMesh& mesh) {
    Pending();
    if (!vol) return;
    const int32_t only[3] = {cube_id(0), cube_id(1), cube_id(2)};
    AppendMesh(vol, only, mesh, "GenerateMeshByCube");
}

CubeMap CubeHandler::GetCubeMap() {
    Pending();
    CubeMap cube_map; // (a fresh copy BY VALUE, as the reference returns one; not the mirror member)
    if (!vol) return cube_map;
    size_t n = 0;
    if (op_volume_block_count(vol, &n) != OP_OK) { Report("GetCubeMap"); return cube_map; }
    std::vector<int32_t> keys(3 * n);
    std::vector<float> vox(n * 512 * 5);
    if (n && op_volume_download(vol, keys.data(), vox.data(), n, &n) != OP_OK) { Report("GetCubeMap"); return cube_map; }
    cube_map.reserve(n);
    for (size_t b = 0; b < n; ++b) {
        const CubeID id(keys[3 * b], keys[3 * b + 1], keys[3 * b + 2]);
        VoxelCube& cube = (cube_map[id] = VoxelCube(id));
        const float* src = &vox[b * 512 * 5];
        for (int v = 0; v < 512; ++v, src += 5) cube.voxels[v] = TSDFVoxel(src[0], src[1], geometry::Point3(src[2], src[3], src[4]));
    }
    return cube_map;
}

void CubeHandler::SetCubeMap(const CubeMap& _cube_map) {
    std::cout << YELLOW << "[WARNING]::[SetCubeMap]::Note that you are changing the hashing map directly." << RESET << std::endl;
    this->cube_map.edited = false; // replaced wholesale
    Touch();
    if (!Ensure()) return;
    if (op_volume_clear(vol) != OP_OK) { Report("SetCubeMap"); return; }
    std::vector<int32_t> keys;
    std::vector<float> vox;
    keys.reserve(3 * _cube_map.size());
    vox.reserve(_cube_map.size() * 512 * 5);
    for (CubeMap::const_iterator it = _cube_map.begin(); it != _cube_map.end(); ++it) {
        for (int k = 0; k < 3; ++k) keys.push_back(it->first(k));
        for (int v = 0; v < 512; ++v) {
            const TSDFVoxel& t = it->second.voxels[v];
            const float rec[5] = {t.sdf, t.weight, t.color(0), t.color(1), t.color(2)};
            vox.insert(vox.end(), rec, rec + 5);
        }
    }
    if (!keys.empty() && op_volume_upload(vol, keys.data(), vox.data(), keys.size() / 3) != OP_OK) Report("SetCubeMap");
}

bool CubeHandler::WriteToFile(const std::string& filename) const {
    Pending();
    if (Ensure() && op_volume_write_file(vol, filename.c_str()) != OP_OK) Report("WriteToFile");
    else std::cout << GREEN << "[CubeHandler]::[INFO]::Write TSDF field done!(To BinaryFile) " << RESET << std::endl;
    return true; // the reference's file calls always return true (SURVEY 8b "Errors")
}
bool CubeHandler::ReadFromFile(const std::string& filename) {
    cube_map.edited = false;
    Touch();
    if (Ensure() && op_volume_read_file(vol, filename.c_str(), 0) != OP_OK) Report("ReadFromFile");
    return true;
}
bool CubeHandler::ReadFromFileFloat(const std::string& filename) {
    cube_map.edited = false;
    Touch();
    if (Ensure() && op_volume_read_file(vol, filename.c_str(), 1) != OP_OK) Report("ReadFromFileFloat");
    return true;
}

bool CubeHandler::MergeAcrossRanks(void* nccl_comm, int root) {
    Pending();
    Touch();
    if (!Ensure()) return false;
    if (op_volume_merge_rccl(vol, nccl_comm, root, nullptr) != OP_OK) { Report("MergeAcrossRanks"); return false; }
    return true;
}

} // namespace integration
} // namespace one_piece
