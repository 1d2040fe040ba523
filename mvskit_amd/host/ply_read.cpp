// See ply_read.hpp.  ASCII, binary_little_endian and binary_big_endian files; any scalar property type; elements other than
// "vertex" (faces, range grids) and properties other than x y z nx ny nz are walked over.
#include "ply_read.hpp"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>

namespace mvshost {
namespace {

enum Type { I8, U8, I16, U16, I32, U32, F32, F64, BAD };
Type type_of(const std::string& s) {
    if (s == "char" || s == "int8") return I8;
    if (s == "uchar" || s == "uint8") return U8;
    if (s == "short" || s == "int16") return I16;
    if (s == "ushort" || s == "uint16") return U16;
    if (s == "int" || s == "int32") return I32;
    if (s == "uint" || s == "uint32") return U32;
    if (s == "float" || s == "float32") return F32;
    if (s == "double" || s == "float64") return F64;
    return BAD;
}
int size_of(Type t) { static const int s[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return s[t]; }

struct Property { std::string name; bool list = false; Type count = U8, value = F32; };
struct Element { std::string name; long long n = 0; std::vector<Property> props; };

bool read_binary(std::istream& is, Type t, bool swap, double& out) {
    unsigned char b[8];
    const int n = size_of(t);
    if (!is.read((char*)b, n)) return false;
    if (swap) for (int i = 0; i < n / 2; ++i) { const unsigned char x = b[i]; b[i] = b[n - 1 - i]; b[n - 1 - i] = x; }
    switch (t) {
        case I8: out = (int8_t)b[0]; break;
        case U8: out = b[0]; break;
        case I16: { int16_t v; memcpy(&v, b, 2); out = v; break; }
        case U16: { uint16_t v; memcpy(&v, b, 2); out = v; break; }
        case I32: { int32_t v; memcpy(&v, b, 4); out = v; break; }
        case U32: { uint32_t v; memcpy(&v, b, 4); out = v; break; }
        case F32: { float v; memcpy(&v, b, 4); out = v; break; }
        case F64: { double v; memcpy(&v, b, 8); out = v; break; }
        default: return false;
    }
    return true;
}

// RPly hands an ASCII value on at the precision of the property's declared type (iascii_float32 & co. in rply.c)
bool read_ascii(std::istream& is, Type t, double& out) {
    double v;
    if (!(is >> v)) return false;
    auto in = [&](double lo, double hi) { return v >= lo && v <= hi; };  // rply.c rejects an ASCII integer outside its type's range
    switch (t) {
        case I8: if (!in(-128, 127)) return false; out = (int8_t)v; break;
        case U8: if (!in(0, 255)) return false; out = (uint8_t)v; break;
        case I16: if (!in(-32768, 32767)) return false; out = (int16_t)v; break;
        case U16: if (!in(0, 65535)) return false; out = (uint16_t)v; break;
        case I32: if (!in(-2147483648.0, 2147483647.0)) return false; out = (int32_t)v; break;
        case U32: if (!in(0, 4294967295.0)) return false; out = (uint32_t)v; break;
        case F32: out = (float)v; break;
        default: out = v; break;
    }
    return true;
}

}  // namespace

int readPlyVertices(const std::string& file, std::vector<double>& points, std::vector<double>* normals, std::string* error) {
    auto fail = [&](const std::string& m) { if (error) *error = m + ": " + file; return -1; };
    std::ifstream is(file.c_str(), std::ios::binary);
    if (!is.is_open()) return fail("cannot open");
    std::string line;
    if (!std::getline(is, line) || line.substr(0, 3) != "ply") return fail("not a PLY file");
    int format = -1;  // 0 ascii, 1 little endian, 2 big endian
    std::vector<Element> elements;
    bool ended = false;
    while (std::getline(is, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key)) continue;
        if (key == "format") {
            std::string f;
            ls >> f;
            format = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
        } else if (key == "element") {
            Element e;
            if (!(ls >> e.name >> e.n) || e.n < 0 || e.n > (1ll << 31)) return fail("bad element line in the PLY header");
            elements.push_back(e);
        } else if (key == "property") {
            if (elements.empty()) return fail("property before any element in the PLY header");
            Property p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, vt;
                ls >> ct >> vt >> p.name;
                p.list = true; p.count = type_of(ct); p.value = type_of(vt);
                if (p.count == BAD || p.count == F32 || p.count == F64) return fail("bad list count type in the PLY header");
            } else {
                p.value = type_of(t);
                ls >> p.name;
            }
            if (p.value == BAD || p.name.empty()) return fail("bad property line in the PLY header");
            elements.back().props.push_back(p);
        } else if (key == "end_header") { ended = true; break; }
        // comment / obj_info: ignored
    }
    if (!ended || format < 0) return fail("incomplete PLY header");
    const uint16_t probe = 1;
    const bool host_little = *(const unsigned char*)&probe == 1;
    const bool swap = format != 0 && ((format == 1) != host_little);
    points.clear();
    if (normals) normals->clear();
    for (const Element& e : elements) {
        const bool vertex = e.name == "vertex";
        int slot[6] = {-1, -1, -1, -1, -1, -1};  // property index of x y z nx ny nz
        if (vertex) {
            static const char* names[6] = {"x", "y", "z", "nx", "ny", "nz"};
            for (int k = 0; k < 6; ++k) for (size_t i = 0; i < e.props.size(); ++i) if (!e.props[i].list && e.props[i].name == names[k]) slot[k] = (int)i;
            if (slot[0] < 0 || slot[1] < 0 || slot[2] < 0) return fail("PLY vertex element without x y z");
            try {
                points.assign((size_t)e.n * 3, 0.0);
                if (normals && slot[3] >= 0 && slot[4] >= 0 && slot[5] >= 0) normals->assign((size_t)e.n * 3, 0.0);
            } catch (const std::bad_alloc&) { return fail("PLY vertex count does not fit in memory"); }
        }
        for (long long i = 0; i < e.n; ++i) {
            for (size_t pi = 0; pi < e.props.size(); ++pi) {
                const Property& p = e.props[pi];
                long long count = 1;
                double v = 0.0;
                if (p.list) {
                    if (format == 0 ? !read_ascii(is, p.count, v) : !read_binary(is, p.count, swap, v)) return fail("short PLY body");
                    count = (long long)v;
                    if (count < 0) return fail("negative list length in the PLY body");
                }
                for (long long c = 0; c < count; ++c)
                    if (format == 0 ? !read_ascii(is, p.value, v) : !read_binary(is, p.value, swap, v)) return fail("short PLY body");
                if (vertex && !p.list) {
                    for (int k = 0; k < 3; ++k) if (slot[k] == (int)pi) points[(size_t)i * 3 + k] = v;
                    if (normals && !normals->empty()) for (int k = 3; k < 6; ++k) if (slot[k] == (int)pi) (*normals)[(size_t)i * 3 + k - 3] = v;
                }
            }
        }
    }
    if (points.empty() && elements.empty()) return fail("PLY file without elements");
    return 0;
}

}  // namespace mvshost

extern "C" long long mvshost_ply_probe(const char* file, double* points, double* normals, long long capacity_vertices, int* has_normals) {
    std::vector<double> p, n;
    if (mvshost::readPlyVertices(file, p, &n) != 0) return -1;
    const long long nv = (long long)(p.size() / 3);
    if (has_normals) *has_normals = n.empty() ? 0 : 1;
    if (points && nv <= capacity_vertices) memcpy(points, p.data(), p.size() * sizeof(double));
    if (normals && !n.empty() && nv <= capacity_vertices) memcpy(normals, n.data(), n.size() * sizeof(double));
    return nv;
}
