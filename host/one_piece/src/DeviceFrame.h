// DeviceFrame.h -- device copies of an RGBDFrame's images (library-internal; see RGBDFrame::on_device).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "Bridge.h"
#include "Geometry/RGBDFrame.h"

namespace one_piece {
namespace bridge {

// Device memory for frames comes in slabs of kSlabFrames slots (hipMalloc synchronises the whole device: one call per 64 frames instead of two
// per frame keeps a pipelined tracker running); a slot goes back to its pool when the last copy of its frame lets go.  Slabs stay with the
// process (the buffer cache of the C-ABI library owns them at exit).
class FramePool {
  public:
    static FramePool& Get(int device, size_t slot_bytes) {
        static std::mutex mu;
        static std::map<std::pair<int, size_t>, FramePool*> pools;
        std::lock_guard<std::mutex> lock(mu);
        FramePool*& p = pools[std::make_pair(device, slot_bytes)];
        if (!p) p = new FramePool(device, slot_bytes);
        return *p;
    }
    void* Take(const char* where) {
        std::lock_guard<std::mutex> lock(mu_);
        if (free_.empty()) {
            void* slab = nullptr;
            if (Failed(op_device_alloc(slot_ * kSlabFrames, device_, &slab), where)) return nullptr;
            for (int i = kSlabFrames - 1; i >= 0; --i) free_.push_back(static_cast<char*>(slab) + static_cast<size_t>(i) * slot_);
        }
        void* p = free_.back();
        free_.pop_back();
        return p;
    }
    void Give(void* p) { std::lock_guard<std::mutex> lock(mu_); free_.push_back(p); }

  private:
    static constexpr int kSlabFrames = 64;
    FramePool(int device, size_t slot) : device_(device), slot_(slot) {}
    int device_;
    size_t slot_;
    std::mutex mu_;
    std::vector<void*> free_;
};

struct DeviceImages {
    void* rgb = nullptr;   // the slot: colour at its start, depth behind it
    void* depth = nullptr;
    int depth_fmt = 0, width = 0, height = 0, device = 0;
    FramePool* pool = nullptr;
    ~DeviceImages() { if (pool && rgb) pool->Give(rgb); }
};

// The frame's images on the device (uploaded at the first call, then shared by every copy of the frame), or null after a message when the
// images are not what the GPU path reads: 3 bytes of colour per pixel, depth as float32 or uint16, both continuous and of the same size.
inline std::shared_ptr<DeviceImages> OnDevice(const geometry::RGBDFrame& f, const char* where) {
    if (f.on_device) return std::static_pointer_cast<DeviceImages>(f.on_device);
    const bool depth_ok = f.depth.type() == CV_16UC1 || f.depth.type() == CV_32FC1;
    if (f.rgb.empty() || f.depth.empty() || f.rgb.type() != CV_8UC3 || !depth_ok || !f.rgb.isContinuous() || !f.depth.isContinuous() ||
        f.rgb.rows != f.depth.rows || f.rgb.cols != f.depth.cols) {
        std::cout << RED << "[ERROR]::[" << where << "]::the frame needs a continuous CV_8UC3 colour image and a CV_16UC1 / CV_32FC1 depth image of the same size" << RESET << std::endl;
        return std::shared_ptr<DeviceImages>();
    }
    std::shared_ptr<DeviceImages> d = std::make_shared<DeviceImages>();
    d->device = Device(); d->width = f.rgb.cols; d->height = f.rgb.rows; d->depth_fmt = DepthFormat(f.depth);
    const size_t npx = static_cast<size_t>(d->width) * d->height;
    const size_t rgb_bytes = npx * 3, depth_at = (rgb_bytes + 255) & ~static_cast<size_t>(255), depth_bytes = npx * (d->depth_fmt == OP_DEPTH_U16 ? 2 : 4);
    FramePool& pool = FramePool::Get(d->device, depth_at + ((npx * 4 + 255) & ~static_cast<size_t>(255))); // one slot size per image size, whatever the depth type
    void* slot = pool.Take(where);
    if (!slot) return std::shared_ptr<DeviceImages>();
    d->pool = &pool; d->rgb = slot; d->depth = static_cast<char*>(slot) + depth_at;
    const void* parts[2] = {f.rgb.data, f.depth.data};
    const size_t bytes[2] = {rgb_bytes, depth_bytes}, offsets[2] = {0, depth_at};
    if (Failed(op_device_write(slot, 2, parts, bytes, offsets, d->device), where)) return std::shared_ptr<DeviceImages>();
    f.on_device = d;
    return d;
}

} // namespace bridge
} // namespace one_piece
