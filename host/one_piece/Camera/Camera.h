// Camera/Camera.h -- camera::PinholeCamera (reference: src/Camera/Camera.h:13-131), a thin class over the C-ABI's
// op_camera POD.  Same constructors, getters (GetWidth / GetHeight return float, as the reference's do, :61-62),
// SetPara, SetCameraType and presets; the preset intrinsics come from op_camera_preset where it has them.
#pragma once
#include "Geometry/Geometry.h"
#include "onepiece_hip.h"

namespace one_piece {
namespace camera {

enum CameraType { TUM_DATASET, OPEN3D_DATASET, MI_DATASET };

class PinholeCamera {
  public:
    PinholeCamera() { SetCameraType(OPEN3D_DATASET); }
    PinholeCamera(float _fx, float _fy, float _cx, float _cy, int _width, int _height, float _depth_scale, float* _distortion = nullptr) {
        SetPara(_fx, _fy, _cx, _cy, _width, _height, _depth_scale, _distortion);
        if (_distortion == nullptr) ClearDistortion();
    }
    PinholeCamera GenerateNextPyramid() const {
        return PinholeCamera(pod.fx / 2, pod.fy / 2, pod.cx / 2, pod.cy / 2, pod.width / 2, pod.height / 2, pod.depth_scale);
    }
    geometry::Matrix3 ToCameraMatrix() const {
        geometry::Matrix3 K = geometry::Matrix3::Zero();
        K(0, 0) = pod.fx; K(0, 2) = pod.cx; K(1, 1) = pod.fy; K(1, 2) = pod.cy; K(2, 2) = 1;
        return K;
    }
    float GetFx() const { return pod.fx; }
    float GetFy() const { return pod.fy; }
    float GetCx() const { return pod.cx; }
    float GetCy() const { return pod.cy; }
    float GetWidth() const { return static_cast<float>(pod.width); }
    float GetHeight() const { return static_cast<float>(pod.height); }
    float GetDepthScale() const { return pod.depth_scale; }
    void SetPara(float _fx, float _fy, float _cx, float _cy, int _width, int _height, float _depth_scale = -1, float* _distortion = nullptr) {
        pod.fx = _fx; pod.fy = _fy; pod.cx = _cx; pod.cy = _cy;
        pod.width = _width; pod.height = _height; pod.depth_scale = _depth_scale;
        if (_distortion != nullptr)
            for (int i = 0; i < 5; ++i) distortion_para[i] = _distortion[i];
    }
    void SetCameraType(const CameraType& type) {
        ClearDistortion();
        if (type == MI_DATASET) { // Camera.h:105-117
            SetPara(2209.84366f, 2210.23057f, 756.24762f, 530.00418f, 1440, 1080, 1000);
            return;
        }
        op_camera_preset(type == TUM_DATASET ? 0 : 1, &pod);
        if (type == TUM_DATASET) { // Camera.h:79-92
            const float d[5] = {0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f};
            for (int i = 0; i < 5; ++i) distortion_para[i] = d[i];
        }
    }
    // what crosses the C-ABI
    const op_camera& Pod() const { return pod; }

  protected:
    void ClearDistortion() { for (int i = 0; i < 5; ++i) distortion_para[i] = 0; }
    op_camera pod;
    float distortion_para[5];
};

} // namespace camera
} // namespace one_piece
