// Integration/CubeHandler.h -- one_piece::integration::CubeHandler on an MI355X.
//
// Same class name, namespace, public signatures, default arguments and value semantics as the reference's
// src/Integration/CubeHandler.h:24-366, so that code written against it (example/ImageSequenceIntegration.cpp,
// example/DenseFusion) compiles and links against this library instead.  What differs is where the block hash lives:
// the reference holds a std::unordered_map<CubeID, VoxelCube> on the host; here the object owns a device-resident
// volume (op_volume, include/onepiece_hip.h) and every member forwards to the C-ABI:
//
//   IntegrateImage      -> op_volume_integrate   (enqueues; frames are fused in batches of up to 32 per launch)
//   every reader        -> flushes + synchronises first (GetCubeMap, HasCube, ExtractTriangleMesh, WriteToFile, ...),
//                          so the deferral cannot be observed (SURVEY 8b "Threading")
//   GetCubeMap()        -> downloads into a fresh CubeMap and returns it BY VALUE, as the reference does
//   copy construction / assignment -> deep copy on the device (a CubeHandler owns its volume; no sharing)
//   Transform* / GetPointCloud     -> std::shared_ptr to freshly made objects
//   errors              -> a coloured line on std::cout and an early return; nothing throws (reference behaviour)
//
// The reference's protected `CubeMap cube_map` member (CubeHandler.h:359) is a MIRROR here (CubeMapMirror below): a derived class
// that reads it gets a host copy refreshed from the device when the volume has changed since its last look, and what it writes
// through it is uploaded before the next member call touches the volume.  Device selection: environment variable
// ONEPIECE_HIP_DEVICE (default 0).
#pragma once
#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "Camera/Camera.h"
#include "Geometry/Geometry.h"
#include "Geometry/RGBDFrame.h"
#include "Geometry/TriangleMesh.h"
#include "Integration/Frustum.h" // the reference's CubeHandler.h brings it in (CubeHandler.h:5; example/ImageIntegration.cpp:30)
#include "Integration/Integrator.h"
#include "Integration/MarchingCube.h"
#include "Integration/VoxelCube.h"

#define TRUNCATED_DISTANCES 1.0

struct op_volume;

namespace one_piece {
namespace integration {

typedef std::unordered_map<CubeID, VoxelCube, CubeHasher> CubeMap;

class CubeHandler;

// The protected `cube_map` of the reference's CubeHandler (CubeHandler.h:359), for classes derived from it.  The volume itself lives in HBM; this
// object hands out a host-side CubeMap that is downloaded lazily -- on the first access after the device volume changed -- and uploaded
// lazily: access through a non-const member marks it edited, and the handler pushes it to the device before its next member call looks at
// or changes the volume (or at CommitCubeMap()).  The container members derived classes use are forwarded; anything else is reachable
// through get() / edit().  A full download / upload per change of ownership: meant for inspection and small edits, not for per-frame use.
class CubeMapMirror {
  public:
    typedef CubeMap::iterator iterator;
    typedef CubeMap::const_iterator const_iterator;
    typedef CubeMap::key_type key_type;
    typedef CubeMap::mapped_type mapped_type;
    typedef CubeMap::value_type value_type;
    typedef CubeMap::size_type size_type;

    const CubeMap& get() const; // refreshed from the device if stale
    CubeMap& edit();            // refreshed, and marked edited
    operator const CubeMap&() const { return get(); }

    size_type size() const { return get().size(); }
    bool empty() const { return get().empty(); }
    size_type count(const key_type& k) const { return get().count(k); }
    const_iterator find(const key_type& k) const { return get().find(k); }
    const_iterator begin() const { return get().begin(); }
    const_iterator end() const { return get().end(); }
    const_iterator cbegin() const { return get().begin(); }
    const_iterator cend() const { return get().end(); }
    const mapped_type& at(const key_type& k) const { return get().at(k); }
    iterator find(const key_type& k) { return edit().find(k); }
    iterator begin() { return edit().begin(); }
    iterator end() { return edit().end(); }
    mapped_type& at(const key_type& k) { return edit().at(k); }
    mapped_type& operator[](const key_type& k) { return edit()[k]; }
    std::pair<iterator, bool> insert(const value_type& v) { return edit().insert(v); }
    size_type erase(const key_type& k) { return edit().erase(k); }
    void clear() { edit().clear(); }
    void reserve(size_type n) { edit().reserve(n); }
    CubeMapMirror& operator=(const CubeMap& m) { edit() = m; return *this; }

  private:
    friend class CubeHandler;
    explicit CubeMapMirror(CubeHandler* o) : owner(o) {}
    CubeMapMirror(const CubeMapMirror&);            // a mirror belongs to one handler: not copyable
    CubeMapMirror& operator=(const CubeMapMirror&);
    CubeHandler* owner;
    mutable CubeMap host;
    mutable unsigned long long seen = 0; // the handler's change counter at the last download / upload (0 = never)
    mutable bool edited = false;
};

class CubeHandler {
  public:
    CubeHandler();
    CubeHandler(const camera::PinholeCamera& _camera);
    CubeHandler(const CubeHandler& other);
    CubeHandler& operator=(const CubeHandler& other);
    ~CubeHandler();

    void SetVoxelResolution(float resolution);
    bool ReadFromFile(const std::string& filename);
    bool ReadFromFileFloat(const std::string& filename);
    bool WriteToFile(const std::string& filename) const;
    bool HasCube(const CubeID& cube_id) const;
    void Clear();
    void SetCamera(const camera::PinholeCamera& _camera);
    void SetTruncation(float trunc);
    void Merge(const CubeHandler& another);
    void Merge(const CubeHandler& another, const geometry::TransformationMatrix& trans);
    void ComputeBounding(const cv::Mat& depth, const geometry::TransformationMatrix& pose, geometry::Point3& max_pos, geometry::Point3& min_pos);
    void PrepareCubes(const cv::Mat& depth, const geometry::TransformationMatrix& pose, std::vector<CubeID>& cube_id_list);
    void IntegrateImage(const cv::Mat& depth, const cv::Mat& rgb, const geometry::TransformationMatrix& pose);
    void IntegrateImage(const geometry::RGBDFrame& rgbd, const geometry::TransformationMatrix& pose);
    CubeID GetCubeID(const geometry::Point3& point) const { return c_para.GetCubeID(point); }
    void AddCube(const CubeID& cube_id);
    // allocate the blocks the voxel centres of v_cube land in after trans: their eight trilinear neighbours / the voxel
    // that contains them (the allocation passes of Transform / TransformNearest, CubeHandler.h:199-241)
    void AddTransformedCube(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans);
    void AddTransformedCubeNearest(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans);
    void ExtractTriangleMesh(geometry::TriangleMesh& mesh);
    void GenerateMeshByCube(const CubeID& cube_id, geometry::TriangleMesh& mesh);
    std::shared_ptr<geometry::PointCloud> GetPointCloud() const;
    std::shared_ptr<CubeHandler> Transform(const geometry::TransformationMatrix& trans) const;
    std::shared_ptr<CubeHandler> TransformNearest(const geometry::TransformationMatrix& trans);
    CubeMap GetCubeMap();
    void SetCubeMap(const CubeMap& _cube_map);
    void SetFarPlane(float _far);
    void SetNearPlane(float _near);

    // ---- beyond the reference's surface -------------------------------------------------------------------
    // blocks currently allocated (cube_map.size() in the reference); flushes
    size_t GetCubeCount() const;
    // waits until every queued frame has been fused
    void Synchronize() const;
    // frame-sharded fusion: merges the volumes of all ranks of an RCCL communicator into `root`'s with one
    // reduce; see op_volume_merge_rccl.  comm is an ncclComm_t.
    bool MergeAcrossRanks(void* nccl_comm, int root);
    // the C-ABI handle (created on first use), for callers that mix in direct op_volume_* calls
    op_volume* Handle() const;

    // pushes what a derived class wrote through `cube_map` to the device now (it happens by itself before the next member call)
    void CommitCubeMap();

  protected:
    camera::PinholeCamera camera;
    Integrator integrator;
    CubePara c_para;
    float far = 5.0;
    float near = 0.5;
    CubeMapMirror cube_map; // CubeHandler.h:359: see CubeMapMirror

  private:
    friend class CubeMapMirror;
    void Pending() const;                // uploads an edited mirror; first statement of every member that looks at or changes the volume
    void Touch() const { ++changes; }    // the device volume has (possibly) changed: the mirror is stale
    mutable unsigned long long changes = 1;
    bool Ensure() const;                 // creates the device volume on first use; false (after a message) without a GPU
    static void Report(const char* where);
    void AddTransformedCubes(const VoxelCube& v_cube, const geometry::TransformationMatrix& trans, bool nearest);
    mutable op_volume* vol = nullptr;
    // device images of frames that were fused in place (RGBDFrame::on_device): held until the volume is known to be done with them
    mutable std::deque<std::pair<unsigned long long, std::shared_ptr<void> > > borrowed_; // (position among the volume's accepted frames, images)
    void ReleaseBorrowed() const { borrowed_.clear(); } // call only right after a synchronising C-ABI call
    explicit CubeHandler(op_volume* adopted, const CubeHandler& like, float resolution);
};

} // namespace integration
} // namespace one_piece
