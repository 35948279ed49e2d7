// MeshIO.cpp -- see MeshIO.h.
#include "MeshIO.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

namespace one_piece {
namespace meshio {
namespace {

enum Scalar { I8, U8, I16, U16, I32, U32, F32, F64, BAD };

Scalar ParseType(const std::string& t) {
    if (t == "char" || t == "int8") return I8;
    if (t == "uchar" || t == "uint8") return U8;
    if (t == "short" || t == "int16") return I16;
    if (t == "ushort" || t == "uint16") return U16;
    if (t == "int" || t == "int32") return I32;
    if (t == "uint" || t == "uint32") return U32;
    if (t == "float" || t == "float32") return F32;
    if (t == "double" || t == "float64") return F64;
    return BAD;
}
size_t SizeOf(Scalar s) { static const size_t n[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return n[s]; }

struct Property { std::string name; Scalar type; bool is_list; Scalar count_type; };
struct Element { std::string name; size_t count; std::vector<Property> props; };

// one scalar of the body, as double (ascii token or little-endian bytes)
struct Body {
    std::istream& is;
    bool ascii;
    bool ok;
    double Next(Scalar t) {
        if (ascii) { double v = 0; if (!(is >> v)) ok = false; return v; }
        unsigned char b[8] = {0};
        if (!is.read(reinterpret_cast<char*>(b), static_cast<std::streamsize>(SizeOf(t)))) { ok = false; return 0; }
        switch (t) {
            case I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
            case U8: return b[0];
            case I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
            case U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
            case I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
            case U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
            case F32: { float v; std::memcpy(&v, b, 4); return v; }
            case F64: { double v; std::memcpy(&v, b, 8); return v; }
            default: ok = false; return 0;
        }
    }
};

int Slot(const std::string& n) {
    static const char* names[9] = {"x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"};
    for (int i = 0; i < 9; ++i) if (n == names[i]) return i;
    return -1;
}

void Message(const char* what, const std::string& file) {
    std::cout << RED << "[ERROR]::[MeshIO]::" << what << " " << file << RESET << std::endl;
}

} // namespace

bool ReadPly(const std::string& file, geometry::Point3List& points, geometry::Point3List& normals, geometry::Point3List& colors,
             geometry::Point3uiList* triangles) {
    std::ifstream is(file.c_str(), std::ios::binary);
    if (!is) { Message("cannot open", file); return false; }
    std::string line;
    if (!std::getline(is, line) || line.compare(0, 3, "ply") != 0) { Message("not a PLY file:", file); return false; }
    bool ascii = false, known_format = false;
    std::vector<Element> elements;
    while (std::getline(is, line)) {
        if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "end_header") break;
        if (key == "format") {
            std::string f; ls >> f;
            ascii = f == "ascii";
            known_format = ascii || f == "binary_little_endian";
        } else if (key == "element") {
            Element e; ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (key == "property" && !elements.empty()) {
            Property p; p.is_list = false; p.count_type = BAD;
            std::string t; ls >> t;
            if (t == "list") { std::string ct, vt; ls >> ct >> vt >> p.name; p.is_list = true; p.count_type = ParseType(ct); p.type = ParseType(vt); }
            else { p.type = ParseType(t); ls >> p.name; }
            if (p.type == BAD || (p.is_list && p.count_type == BAD)) { Message("unsupported property type in", file); return false; }
            elements.back().props.push_back(p);
        }
    }
    if (!known_format) { Message("unsupported PLY format (ascii and binary_little_endian are read) in", file); return false; }
    // The header is not trusted: an element cannot have more entries than the rest of the file has bytes (every entry takes at least one),
    // list counts are bounded the same way, and face indices must name vertices that exist (ReadObj checks the same).
    const std::streampos body_at = is.tellg();
    is.seekg(0, std::ios::end);
    const size_t body_bytes = body_at < is.tellg() ? static_cast<size_t>(is.tellg() - body_at) : 0;
    is.seekg(body_at);
    for (size_t e = 0; e < elements.size(); ++e)
        if (!elements[e].props.empty() && elements[e].count > body_bytes) { Message("element count exceeds the file size in", file); return false; }
    Body body = {is, ascii, true};
    points.clear(); normals.clear(); colors.clear();
    if (triangles) triangles->clear();
    for (size_t e = 0; e < elements.size() && body.ok; ++e) {
        const Element& el = elements[e];
        if (el.name == "vertex") {
            bool has[9] = {false};
            bool color_is_byte = true;
            for (size_t k = 0; k < el.props.size(); ++k) {
                const int s = Slot(el.props[k].name);
                if (s >= 0 && !el.props[k].is_list) { has[s] = true; if (s >= 6 && (el.props[k].type == F32 || el.props[k].type == F64)) color_is_byte = false; }
            }
            const bool want_n = has[3] && has[4] && has[5], want_c = has[6] && has[7] && has[8];
            points.resize(el.count);
            if (want_n) normals.resize(el.count);
            if (want_c) colors.resize(el.count);
            for (size_t i = 0; i < el.count && body.ok; ++i) {
                float v[9] = {0};
                for (size_t k = 0; k < el.props.size(); ++k) {
                    const Property& p = el.props[k];
                    if (p.is_list) {
                        const double cnt = body.Next(p.count_type);
                        if (!(cnt >= 0) || cnt > static_cast<double>(body_bytes)) { body.ok = false; break; } // negative, NaN or larger than the file
                        const size_t n = static_cast<size_t>(cnt);
                        for (size_t j = 0; j < n && body.ok; ++j) body.Next(p.type);
                        continue;
                    }
                    const double d = body.Next(p.type);
                    const int s = Slot(p.name);
                    if (s >= 0) v[s] = static_cast<float>(d);
                }
                points[i] = geometry::Point3(v[0], v[1], v[2]);
                if (want_n) normals[i] = geometry::Point3(v[3], v[4], v[5]);
                if (want_c) colors[i] = color_is_byte ? geometry::Point3(v[6] / 255.0f, v[7] / 255.0f, v[8] / 255.0f) : geometry::Point3(v[6], v[7], v[8]);
            }
        } else {
            const bool faces = el.name == "face";
            for (size_t i = 0; i < el.count && body.ok; ++i)
                for (size_t k = 0; k < el.props.size(); ++k) {
                    const Property& p = el.props[k];
                    if (!p.is_list) { body.Next(p.type); continue; }
                    const double cnt = body.Next(p.count_type);
                    if (!(cnt >= 0) || cnt > static_cast<double>(body_bytes)) { body.ok = false; break; }
                    const size_t n = static_cast<size_t>(cnt);
                    const bool indices = faces && triangles && (p.name == "vertex_indices" || p.name == "vertex_index");
                    unsigned first = 0, prev = 0;
                    bool valid = true; // a face that names a vertex the file does not have is dropped
                    const size_t before = indices ? triangles->size() : 0;
                    for (size_t j = 0; j < n && body.ok; ++j) {
                        const double raw = body.Next(p.type);
                        if (!indices) continue;
                        if (!(raw >= 0) || raw >= static_cast<double>(points.size())) { valid = false; continue; }
                        const unsigned id = static_cast<unsigned>(raw);
                        if (j == 0) first = id;
                        else if (j >= 2 && valid) triangles->push_back(geometry::Point3ui(first, prev, id)); // fan
                        prev = id;
                    }
                    if (indices && !valid) triangles->resize(before);
                }
        }
    }
    if (!body.ok) { Message("truncated or malformed body in", file); return false; }
    return true;
}

bool WritePly(const std::string& file, const geometry::Point3List& pts, const geometry::Point3List& nrm, const geometry::Point3List& col,
              const geometry::Point3uiList* tri) {
    std::ofstream os(file.c_str(), std::ios::binary);
    if (!os) { Message("cannot open", file); return false; }
    const bool has_n = nrm.size() == pts.size() && !pts.empty(), has_c = col.size() == pts.size() && !pts.empty();
    os << "ply\nformat binary_little_endian 1.0\nelement vertex " << pts.size() << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (has_n) os << "property float nx\nproperty float ny\nproperty float nz\n";
    if (has_c) os << "property uchar red\nproperty uchar green\nproperty uchar blue\n";
    if (tri) os << "element face " << tri->size() << "\nproperty list uchar uint vertex_indices\n";
    os << "end_header\n";
    for (size_t i = 0; i < pts.size(); ++i) {
        os.write(reinterpret_cast<const char*>(pts[i].data()), 12);
        if (has_n) os.write(reinterpret_cast<const char*>(nrm[i].data()), 12);
        if (has_c) {
            unsigned char rgb[3];
            for (int k = 0; k < 3; ++k) {
                const float v = col[i](k) * 255.0f;
                rgb[k] = static_cast<unsigned char>(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            os.write(reinterpret_cast<const char*>(rgb), 3);
        }
    }
    if (tri)
        for (size_t i = 0; i < tri->size(); ++i) {
            const unsigned char three = 3;
            os.write(reinterpret_cast<const char*>(&three), 1);
            os.write(reinterpret_cast<const char*>((*tri)[i].data()), 12);
        }
    return static_cast<bool>(os);
}

bool ReadObj(const std::string& file, geometry::Point3List& points, geometry::Point3List& normals, geometry::Point3List& colors,
             geometry::Point3uiList* triangles) {
    std::ifstream is(file.c_str());
    if (!is) { Message("cannot open", file); return false; }
    points.clear(); normals.clear(); colors.clear();
    if (triangles) triangles->clear();
    geometry::Point3List vn;                 // normals in file order; attached to vertices through the faces' v//vn pairs
    std::vector<long> normal_of;             // per vertex: index into vn or -1
    bool all_colored = true;
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "v") {
            float v[6] = {0, 0, 0, 0, 0, 0};
            int n = 0;
            while (n < 6 && (ls >> v[n])) ++n;
            if (n < 3) { Message("malformed vertex in", file); return false; }
            points.push_back(geometry::Point3(v[0], v[1], v[2]));
            if (n == 6) colors.push_back(geometry::Point3(v[3], v[4], v[5])); else all_colored = false;
        } else if (key == "vn") {
            float v[3] = {0, 0, 0};
            ls >> v[0] >> v[1] >> v[2];
            vn.push_back(geometry::Point3(v[0], v[1], v[2]));
        } else if (key == "f") {
            std::vector<unsigned> ids;
            std::string tok;
            while (ls >> tok) {
                long vi = std::strtol(tok.c_str(), nullptr, 10), ni = 0;
                const size_t s1 = tok.find('/');
                if (s1 != std::string::npos) {
                    const size_t s2 = tok.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) ni = std::strtol(tok.c_str() + s2 + 1, nullptr, 10);
                }
                if (vi < 0) vi += static_cast<long>(points.size()) + 1;
                if (ni < 0) ni += static_cast<long>(vn.size()) + 1;
                if (vi < 1 || vi > static_cast<long>(points.size())) { Message("face index out of range in", file); return false; }
                ids.push_back(static_cast<unsigned>(vi - 1));
                if (ni >= 1 && ni <= static_cast<long>(vn.size())) {
                    if (normal_of.size() < points.size()) normal_of.resize(points.size(), -1);
                    normal_of[vi - 1] = ni - 1;
                }
            }
            if (triangles)
                for (size_t j = 2; j < ids.size(); ++j) triangles->push_back(geometry::Point3ui(ids[0], ids[j - 1], ids[j]));
        }
    }
    if (!all_colored || colors.size() != points.size()) colors.clear();
    if (vn.size() == points.size() && normal_of.empty()) normals = vn; // point clouds: one vn per v, in order
    else if (!normal_of.empty()) {
        normal_of.resize(points.size(), -1);
        normals.assign(points.size(), geometry::Point3(0, 0, 0));
        for (size_t i = 0; i < points.size(); ++i) if (normal_of[i] >= 0) normals[i] = vn[normal_of[i]];
    }
    return true;
}

bool WriteObj(const std::string& file, const geometry::Point3List& pts, const geometry::Point3List& nrm, const geometry::Point3List& col,
              const geometry::Point3uiList* tri) {
    std::FILE* f = std::fopen(file.c_str(), "w");
    if (!f) { Message("cannot open", file); return false; }
    const bool has_n = nrm.size() == pts.size() && !pts.empty(), has_c = col.size() == pts.size() && !pts.empty();
    for (size_t i = 0; i < pts.size(); ++i) {
        if (has_c) std::fprintf(f, "v %.9g %.9g %.9g %.9g %.9g %.9g\n", pts[i](0), pts[i](1), pts[i](2), col[i](0), col[i](1), col[i](2));
        else std::fprintf(f, "v %.9g %.9g %.9g\n", pts[i](0), pts[i](1), pts[i](2));
    }
    if (has_n) for (size_t i = 0; i < nrm.size(); ++i) std::fprintf(f, "vn %.9g %.9g %.9g\n", nrm[i](0), nrm[i](1), nrm[i](2));
    if (tri)
        for (size_t i = 0; i < tri->size(); ++i) {
            const unsigned a = (*tri)[i](0) + 1, b = (*tri)[i](1) + 1, c = (*tri)[i](2) + 1;
            if (has_n) std::fprintf(f, "f %u//%u %u//%u %u//%u\n", a, a, b, b, c, c);
            else std::fprintf(f, "f %u %u %u\n", a, b, c);
        }
    return std::fclose(f) == 0;
}

} // namespace meshio
} // namespace one_piece
