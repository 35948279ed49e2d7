// MiniCv.h -- the image container the hot-path surface is declared with, for builds WITHOUT OpenCV.
//
// The reference passes images as cv::Mat (CubeHandler.h:178-184, PointCloud.h:30-32).  Inside the reference tree
// (-DONEPIECE_HAVE_OPENCV) the real <opencv2/core/core.hpp> is used and this file is not included.  Elsewhere (this
// repository's tests; OpenCV is not in the image) the same declarations compile against the minimal container
// below: a reference-counted, continuous, row-major buffer with rows/cols/data/type()/depth()/channels()/create/
// clone/at<T>/ptr<T>, the CV_* type codes of the formats the path accepts, cv::Vec3b, and an imread() for the PNG
// files of the reference's sequence format (8-bit grey / RGB / RGBA and 16-bit grey, non-interlaced; inflated with
// zlib; colour is returned B,G,R like cv::imread).  It is a container, not an image-processing library.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)

namespace cv {

struct Vec3b {
    unsigned char val[3];
    unsigned char& operator[](int i) { return val[i]; }
    const unsigned char& operator[](int i) const { return val[i]; }
};

class Mat {
  public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // wraps caller-owned memory (no copy, no ownership), like cv::Mat(rows, cols, type, void*)
    Mat(int r, int c, int type, void* external) : rows(r), cols(c), data(static_cast<unsigned char*>(external)), flags_(type) {}

    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == flags_ && buf_ && buf_.use_count() == 1) return;
        rows = r; cols = c; flags_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>(static_cast<size_t>(r) * c * elemSize());
        data = buf_->empty() ? nullptr : buf_->data();
    }
    void release() { buf_.reset(); rows = cols = 0; data = nullptr; }
    Mat clone() const {
        Mat m(rows, cols, flags_);
        if (data && m.data) std::memcpy(m.data, data, total() * elemSize());
        return m;
    }
    int type() const { return flags_; }
    int depth() const { return flags_ & 7; }
    int channels() const { return (flags_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const {
        static const int bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        return static_cast<size_t>(bytes[depth()]) * channels();
    }
    size_t total() const { return static_cast<size_t>(rows) * cols; }
    bool empty() const { return data == nullptr || total() == 0; }
    bool isContinuous() const { return true; }

    template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[static_cast<size_t>(r) * cols + c]; }
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[static_cast<size_t>(r) * cols + c]; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + static_cast<size_t>(r) * cols * elemSize()); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + static_cast<size_t>(r) * cols * elemSize()); }

  private:
    int flags_ = 0;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

// PNG files only.  flags: -1 (IMREAD_UNCHANGED) keeps 16-bit grey as CV_16UC1; otherwise 3-channel B,G,R bytes.
// Returns an empty Mat when the file is missing or not one of the supported PNG kinds.
Mat imread(const std::string& filename, int flags = 1);

} // namespace cv
