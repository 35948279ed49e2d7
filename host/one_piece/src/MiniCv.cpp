// MiniCv.cpp -- cv::imread for builds without OpenCV: a small PNG decoder (zlib inflate + the five scanline filters).
// Supports what the reference's sequence format contains: 8-bit grey / RGB / RGBA / grey+alpha and 16-bit grey,
// non-interlaced.  Not compiled when the real OpenCV is used (-DONEPIECE_HAVE_OPENCV).
#ifndef ONEPIECE_HAVE_OPENCV
#include "compat/MiniCv.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>

namespace cv {

namespace {
unsigned BE32(const unsigned char* p) { return (static_cast<unsigned>(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
int Paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
} // namespace

Mat imread(const std::string& filename, int flags) {
    Mat empty;
    std::FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return empty;
    std::vector<unsigned char> file;
    unsigned char chunk[1 << 16];
    for (size_t got; (got = std::fread(chunk, 1, sizeof(chunk), f)) > 0;) file.insert(file.end(), chunk, chunk + got);
    std::fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return empty;
    unsigned width = 0, height = 0;
    int bit_depth = 0, colour = 0, interlace = 0;
    std::vector<unsigned char> idat;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const unsigned len = BE32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        const unsigned char* body = &file[pos + 8];
        if (pos + 12 + len > file.size()) return empty;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = BE32(body); height = BE32(body + 4);
            bit_depth = body[8]; colour = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    const int samples = colour == 0 ? 1 : colour == 2 ? 3 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!width || !height || interlace || !samples || !(bit_depth == 8 || (bit_depth == 16 && colour == 0))) return empty;
    const size_t bpp = static_cast<size_t>(samples) * bit_depth / 8, stride = bpp * width;
    std::vector<unsigned char> raw((stride + 1) * height);
    uLongf raw_len = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size()) return empty;
    // undo the scanline filters in place (each row: 1 filter byte + stride bytes)
    std::vector<unsigned char> img(stride * height);
    for (unsigned y = 0; y < height; ++y) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        const int filter = in[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[1 + x];
            switch (filter) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += Paeth(a, b, c); break;
                default: break;
            }
            cur[x] = static_cast<unsigned char>(v);
        }
    }
    if (bit_depth == 16 && flags < 0) { // IMREAD_UNCHANGED: big-endian samples -> native uint16
        Mat m(static_cast<int>(height), static_cast<int>(width), CV_16UC1);
        unsigned short* out = reinterpret_cast<unsigned short*>(m.data);
        for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) out[i] = static_cast<unsigned short>((img[2 * i] << 8) | img[2 * i + 1]);
        return m;
    }
    // everything else is delivered as 3-channel B,G,R bytes (what cv::imread's default flag does)
    Mat m(static_cast<int>(height), static_cast<int>(width), CV_8UC3);
    for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) {
        unsigned char r, g, b;
        if (bit_depth == 16) { r = g = b = img[2 * i]; }
        else if (samples <= 2) { r = g = b = img[bpp * i]; }
        else { r = img[bpp * i]; g = img[bpp * i + 1]; b = img[bpp * i + 2]; }
        m.data[3 * i] = b; m.data[3 * i + 1] = g; m.data[3 * i + 2] = r;
    }
    return m;
}

} // namespace cv

// C entry for tests: decode `path` with the reader above; returns 0 and fills rows / cols / type (+ the pixels when
// `out` holds at least rows * cols * elemSize bytes), 1 when the file cannot be decoded.
extern "C" int op_host_imread(const char* path, int flags, int* rows, int* cols, int* type, unsigned char* out, size_t cap) {
    const cv::Mat m = cv::imread(path, flags);
    if (m.empty()) return 1;
    *rows = m.rows; *cols = m.cols; *type = m.type();
    const size_t bytes = m.total() * m.elemSize();
    if (out && cap >= bytes) std::memcpy(out, m.data, bytes);
    return 0;
}
#endif
