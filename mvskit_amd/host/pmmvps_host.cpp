// pmmvps_host.cpp -- see pmmvps_host.hpp.  Host code around the C ABI; nothing here computes the hot path.
#include "pmmvps_host.hpp"
#include "jpeg_decode.hpp"
#include "ply_read.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

namespace mvshost {

using std::cerr;
using std::endl;
using std::ifstream;
using std::ofstream;

// ------------------------------------------------------------------ Option (pmmvps/option.cpp)
Option::Option() {  // option.cpp:19-33
    m_nimages = 0; m_nillums = 1;
    m_level = 1; m_csize = 2; m_wsize = 7; m_nccThreshold = 0.7f; m_minImageNum = 3; m_cpu = 4;
    m_setEdge = 0; m_useBound = 0; m_useVisData = 0; m_sequence = -1; m_flag = -10;
    m_maxAngleThreshold = (float)(10.0f * M_PI / 180.0f);
    m_quadThreshold = 2.5f;
}

void Option::init(const string prefix, const string option) {
    m_status = -1;  // option.cpp:35-149
    m_prefix = prefix; m_option = option;
    ifstream ifstr((prefix + option).c_str());
    if (!ifstr.is_open()) { cerr << "Cannot open option file: " << prefix + option << endl; return; }
    string name;
    while (ifstr >> name) {
        if (name[0] == '#') { string rest; std::getline(ifstr, rest); continue; }
        if (name == "image") ifstr >> m_nimages;
        else if (name == "illum") ifstr >> m_nillums;
        else if (name == "level") ifstr >> m_level;
        else if (name == "csize") ifstr >> m_csize;
        else if (name == "threshold") ifstr >> m_nccThreshold;
        else if (name == "wsize") ifstr >> m_wsize;
        else if (name == "minImageNum") ifstr >> m_minImageNum;
        else if (name == "CPU") ifstr >> m_cpu;
        else if (name == "setEdge") ifstr >> m_setEdge;
        else if (name == "useBound") ifstr >> m_useBound;
        else if (name == "useVisData") ifstr >> m_useVisData;
        else if (name == "sequence") ifstr >> m_sequence;
        else if (name == "maxAngle") { ifstr >> m_maxAngleThreshold; m_maxAngleThreshold *= (float)(M_PI / 180.0f); }
        else if (name == "quad") ifstr >> m_quadThreshold;
        else if (name == "images") {
            ifstr >> m_flag;
            if (m_flag == -1) {
                int first, last;
                ifstr >> first >> last;
                for (int i = first; i < last; ++i) m_images.push_back(i);
            } else if (0 < m_flag) {
                for (int i = 0; i < m_flag; ++i) { int idx; ifstr >> idx; m_images.push_back(idx); }
            } else { cerr << "flag is not valid: " << m_flag << endl; return; }
        } else { cerr << "Unrecognizable option: " << name << endl; return; }
    }
    if (m_flag == -10) { cerr << "m_flag not specified: " << m_flag << endl; return; }
    if (m_nimages == 0) m_nimages = (int)m_images.size();
    for (int i = 0; i < (int)m_images.size(); ++i) m_dict[m_images[i]] = i;
    initVisdata();
    m_status = 0;
}

void Option::initVisdata() {  // option.cpp:151-170
    if (m_useVisData != 0) return;
    const int n = (int)m_images.size();
    m_visdata2.assign(n, {});
    for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) if (x != y) m_visdata2[y].push_back(x);
}

// ------------------------------------------------------------------ Photo / PhotoSet
int Photo::initCamera(const string cname) {  // camera.cpp:27-63
    ifstream ifstr(cname.c_str());
    string header;
    if (!(ifstr >> header)) { cerr << "Cannot read camera file " << cname << endl; return -1; }
    if (header == "CONTOUR") m_txtType = 0;
    else if (header == "CONTOUR2") m_txtType = 2;
    else { cerr << "Unrecognizable text format" << endl; return -1; }
    float p[12];
    for (int i = 0; i < 12; ++i) if (!(ifstr >> p[i])) { cerr << "Short camera file " << cname << endl; return -1; }
    if (m_txtType == 0) {  // camera.cpp:110-116
        for (int i = 0; i < 12; ++i) m_projection[i] = p[i];
        return 0;
    }
    // CONTOUR2: K from (fx, fy, skew, cx, cy, -), Rt from Euler degrees + t (camera.cpp:117-134, quat2proj 241-261)
    const float a = (float)(p[6] * M_PI / 180.0), b = (float)(p[7] * M_PI / 180.0), g = (float)(p[8] * M_PI / 180.0);
    const float s1 = sinf(a), s2 = sinf(b), s3 = sinf(g), c1 = cosf(a), c2 = cosf(b), c3 = cosf(g);
    const float Rt[3][4] = {{c2 * c3, c3 * s2 * s1 - s3 * c1, c3 * s2 * c1 + s3 * s1, p[9]},
                            {s3 * c2, s3 * s2 * s1 + c3 * c1, s3 * s2 * c1 - c3 * s1, p[10]},
                            {-s2, c2 * s1, c2 * c1, p[11]}};
    for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x) m_R[3 * y + x] = Rt[y][x];
    const float K[3][3] = {{p[0], p[2], p[3]}, {0.0f, p[1], p[4]}, {0.0f, 0.0f, 1.0f}};
    for (int y = 0; y < 3; ++y) for (int x = 0; x < 4; ++x) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += K[y][k] * Rt[k][x];
        m_projection[4 * y + x] = s;
    }
    return 0;
}

static bool pnm_header(std::istream& is, const char* magic, int& w, int& h, int& maxv) {
    string m;
    is >> m;
    if (m != magic) return false;
    auto skip = [&]() { while (is >> std::ws && is.peek() == '#') { string l; std::getline(is, l); } };
    skip(); is >> w; skip(); is >> h; skip(); is >> maxv;
    is.get();
    return (bool)is && w > 0 && h > 0 && w <= 65535 && h <= 65535 && maxv == 255;
}
int Photo::readPpm(const string iname) {
    ifstream is(iname.c_str(), std::ios::binary);
    int w, h, maxv;
    if (!is.is_open() || !pnm_header(is, "P6", w, h, maxv)) return -1;
    m_width = w; m_height = h;
    m_image.resize((size_t)w * h * 3);
    is.read((char*)m_image.data(), (std::streamsize)m_image.size());
    return is ? 0 : -1;
}
int Photo::readJpeg(const string iname) {  // image.cpp:827-879
    int w = 0, h = 0, ch = 0;
    vector<unsigned char> px;
    string err;
    if (readJpegFile(iname, px, w, h, ch, &err) != 0) { cerr << "Couldn't read image " << iname << " (" << err << ")" << endl; return -1; }
    m_width = w; m_height = h;
    if (ch == 3) { m_image.swap(px); return 0; }
    m_image.resize((size_t)w * h * 3);  // image.cpp:850-858
    for (size_t i = 0; i < (size_t)w * h; ++i) m_image[3 * i] = m_image[3 * i + 1] = m_image[3 * i + 2] = px[i];
    return 0;
}
int Photo::readPgmMask(const string mname) {
    ifstream is(mname.c_str(), std::ios::binary);
    int w, h, maxv;
    if (!is.is_open() || !pnm_header(is, "P5", w, h, maxv) || w != m_width || h != m_height) return -1;
    m_mask.resize((size_t)w * h);
    is.read((char*)m_mask.data(), (std::streamsize)m_mask.size());
    for (auto& m : m_mask) m = m > 127 ? 255 : 0;
    return is ? 0 : -1;
}

// Image::readPBMImage, image.cpp:881-946: binary P4; a set bit is background (0), a clear bit foreground (255).  As in
// the reference the bits are taken as ONE continuous stream of width*height bits (no padding at the end of a row, which
// the PBM format would have when the width is not a multiple of 8) -- masks written by the reference's tool chain are laid
// out that way.
int Photo::readPbmMask(const string mname) {
    ifstream is(mname.c_str(), std::ios::binary);
    string magic;
    if (!is.is_open() || !(is >> magic) || magic != "P4") return -1;
    int w = 0, h = 0;
    while (is >> std::ws && is.peek() == '#') { string l; std::getline(is, l); }
    is >> w;
    while (is >> std::ws && is.peek() == '#') { string l; std::getline(is, l); }
    is >> h;
    is.get();
    if (!is || w != m_width || h != m_height) return -1;
    const size_t n = (size_t)w * h;
    vector<unsigned char> bytes((n + 7) / 8);
    is.read((char*)bytes.data(), (std::streamsize)bytes.size());
    if (!is) return -1;
    m_mask.resize(n);
    for (size_t i = 0; i < n; ++i) m_mask[i] = ((bytes[i >> 3] >> (7 - (i & 7))) & 1) ? 0 : 255;
    return 0;
}

void PhotoSet::init(const vector<int>& images, const string prefix, const int nimages, const int nillums, const int, const int, const int) {
    m_status = 0;
    m_images = images; m_nimages = nimages; m_nillums = nillums; m_prefix = prefix;  // photoSet.cpp:20-61
    for (int i = 0; i < m_nimages; ++i) m_dict[images[i]] = i;
    m_photos.assign(m_nimages, Photo());
    for (int i = 0; i < m_nimages; ++i) {
        char iname[1024], cname[1024], mname[1024];
        snprintf(iname, sizeof iname, "%simage/%04d%04d.ppm", prefix.c_str(), i, 0);
        snprintf(mname, sizeof mname, "%smask/%08d.pgm", prefix.c_str(), i);
        snprintf(cname, sizeof cname, "%stxt/%08d.txt", prefix.c_str(), i);
        if (m_photos[i].initCamera(cname) != 0) { m_status = -1; return; }
        if (m_photos[i].readPpm(iname) != 0) {  // Image::completeName (image.cpp:51-74): <name>.ppm if it exists, else <name>.jpg
            snprintf(iname, sizeof iname, "%simage/%04d%04d.jpg", prefix.c_str(), i, 0);
            if (m_photos[i].readJpeg(iname) != 0) { cerr << "Unsupported image format found (binary PPM or JPEG): " << iname << endl; m_status = -1; return; }
        }
        if (m_photos[i].readPgmMask(mname) != 0) {  // Image::alloc tries .pgm, then .pbm (image.cpp:143-147)
            snprintf(mname, sizeof mname, "%smask/%08d.pbm", prefix.c_str(), i);
            (void)m_photos[i].readPbmMask(mname);
        }
    }
}
void PhotoSet::setPhoto(int index, int width, int height, const float P[12], const unsigned char* rgb, const unsigned char* mask) {
    if ((int)m_photos.size() <= index) m_photos.resize(index + 1);
    Photo& ph = m_photos[index];
    ph.m_width = width; ph.m_height = height;
    memcpy(ph.m_projection, P, sizeof ph.m_projection);
    ph.m_image.assign(rgb, rgb + (size_t)width * height * 3);
    if (mask) ph.m_mask.assign(mask, mask + (size_t)width * height); else ph.m_mask.clear();
    m_nimages = (int)m_photos.size();
}
void PhotoSet::project(const int index, const Vector4f& coord, const int level, float icoord[3]) const {  // camera.cpp:310-326
    const float* P = m_photos[index].m_projection;
    const float s = 1.0f / (float)(1 << level);  // Camera::updateProjection (camera.cpp:91-100): rows 0 and 1 halved per level
    float v[3];
    for (int r = 0; r < 3; ++r) {
        float a = 0.0f;
        for (int k = 0; k < 4; ++k) a += (r < 2 ? P[4 * r + k] * s : P[4 * r + k]) * coord[k];
        v[r] = a;
    }
    if (v[2] <= 0.0f) { icoord[0] = icoord[1] = -65535.0f; icoord[2] = -1.0f; return; }
    icoord[0] = v[0] / v[2]; icoord[1] = v[1] / v[2]; icoord[2] = 1.0f;
}
int PhotoSet::getMask(const int index, const int ix, const int iy, const int level) const {  // image.cpp:765-781
    const Photo& ph = m_photos[index];
    if (level != 0 || ph.m_mask.empty()) return -1;
    if (ix < 0 || ph.m_width <= ix || iy < 0 || ph.m_height <= iy) return -1;
    return ph.m_mask[(size_t)iy * ph.m_width + ix];
}
int PhotoSet::image2index(const int image) const {  // photoSet.cpp:251-259
    auto pos = m_dict.find(image);
    return pos == m_dict.end() ? -1 : pos->second;
}

// ------------------------------------------------------------------ Patch (pmmvps/patch.cpp)
Patch::Patch() {
    m_coord = {0, 0, 0, 1}; m_normal = {0, 0, 0, 0};
    m_ncc = -1.0f; m_nimages = 0; m_iter = 0; m_collected = 0; m_flag = 0; m_dflag = 0; m_fix = 0; m_id = -1;
    m_dscale = 0.0f; m_ascale = 0.0f; m_tmp = 0.0f;
}
float Patch::score2(const float threshold) const { return std::max(0.0f, m_ncc - threshold) * (int)m_images.size(); }

std::istream& operator>>(std::istream& istr, Patch& rhs) {
    string header;
    int itmp = 0;
    istr >> header;
    for (int k = 0; k < 4; ++k) istr >> rhs.m_coord[k];
    for (int k = 0; k < 4; ++k) istr >> rhs.m_normal[k];
    istr >> rhs.m_ncc >> rhs.m_dscale >> rhs.m_ascale;
    if (header == "PATCHA") { int type; float dir[4]; istr >> type >> dir[0] >> dir[1] >> dir[2] >> dir[3]; }
    istr >> itmp;
    rhs.m_images.resize(std::max(itmp, 0));
    for (int i = 0; i < itmp; ++i) istr >> rhs.m_images[i];
    istr >> itmp;
    rhs.m_vimages.resize(std::max(itmp, 0));
    for (int i = 0; i < itmp; ++i) istr >> rhs.m_vimages[i];
    return istr;
}
std::ostream& operator<<(std::ostream& ostr, const Patch& rhs) {
    ostr << "PATCHES" << endl
         << rhs.m_coord[0] << " " << rhs.m_coord[1] << " " << rhs.m_coord[2] << " " << rhs.m_coord[3] << endl
         << rhs.m_normal[0] << " " << rhs.m_normal[1] << " " << rhs.m_normal[2] << " " << rhs.m_normal[3] << endl
         << rhs.m_ncc << ' ' << rhs.m_dscale << ' ' << rhs.m_ascale << endl
         << (int)rhs.m_images.size() << endl;
    for (int v : rhs.m_images) ostr << v << ' ';
    ostr << endl << (int)rhs.m_vimages.size() << endl;
    for (int v : rhs.m_vimages) ostr << v << ' ';
    ostr << endl;
    return ostr;
}

static mvs_patch to_record(const Patch& p) {
    mvs_patch r;
    memset(&r, 0, sizeof r);
    for (int k = 0; k < 4; ++k) { r.coord[k] = p.m_coord[k]; r.normal[k] = p.m_normal[k]; }
    r.ncc = p.m_ncc; r.dscale = p.m_dscale; r.ascale = p.m_ascale; r.tmp = p.m_tmp;
    r.nimages = std::min<int>((int)p.m_images.size(), MVS_MAX_IMAGES);   // what a record stores; the engine cuts to its own list cap
    r.nvimages = std::min<int>((int)p.m_vimages.size(), MVS_MAX_IMAGES);
    for (int i = 0; i < r.nimages; ++i) r.images[i] = (uint8_t)p.m_images[i];
    for (int i = 0; i < r.nvimages; ++i) r.vimages[i] = (uint8_t)p.m_vimages[i];
    r.flags = 1;
    return r;
}
static Ppatch from_record(const mvs_patch& r) {
    Ppatch pp(new Patch());
    for (int k = 0; k < 4; ++k) { pp->m_coord[k] = r.coord[k]; pp->m_normal[k] = r.normal[k]; }
    pp->m_ncc = r.ncc; pp->m_dscale = r.dscale; pp->m_ascale = r.ascale; pp->m_tmp = r.tmp;
    pp->m_images.assign(r.images, r.images + r.nimages);
    pp->m_vimages.assign(r.vimages, r.vimages + r.nvimages);
    pp->m_nimages = r.nimages; pp->m_id = r.id;
    return pp;
}

// ------------------------------------------------------------------ PatchManager
void PatchManager::init() {  // patch_manager.cpp:24-51 (the grids themselves are device-side)
    m_gwidths.assign(m_pmmvps.m_nimages, 0);
    m_gheights.assign(m_pmmvps.m_nimages, 0);
    for (int i = 0; i < m_pmmvps.m_nimages; ++i) mvs_engine_grid_dims(m_pmmvps.m_engine, i, &m_gwidths[i], &m_gheights[i]);
}
void PatchManager::image2index(Patch& patch) {  // patch_manager.cpp:53-63
    vector<int> out;
    for (int im : patch.m_images) { const int idx = m_pmmvps.m_photoSet.image2index(im); if (idx != -1) out.push_back(idx); }
    patch.m_images.swap(out);
}
void PatchManager::index2image(Patch& patch) {  // patch_manager.cpp:65-73
    for (int& v : patch.m_images) v = m_pmmvps.m_photoSet.m_images[v];
    for (int& v : patch.m_vimages) v = m_pmmvps.m_photoSet.m_images[v];
}
int PatchManager::upload(const vector<Ppatch>& pp) {
    vector<mvs_patch> recs;
    recs.reserve(pp.size());
    for (const Ppatch& p : pp) recs.push_back(to_record(*p));
    return mvs_engine_upload_patches(m_pmmvps.m_engine, (int64_t)recs.size(), recs.data());
}
void PatchManager::addPatches(const vector<Ppatch>& seeds) { (void)upload(seeds); }
int PatchManager::readPatches() { return readPatches(0); }
int PatchManager::readPatches(const int iter) {  // patch_manager.cpp:435-497
    char buffer[1024];
    snprintf(buffer, sizeof buffer, "%sply/%08d.patch", m_pmmvps.m_prefix.c_str(), iter);
    ifstream ifstr(buffer);
    if (!ifstr.is_open()) return -1;
    string header;
    int pnum = 0;
    ifstr >> header >> pnum;
    vector<Ppatch> pp;
    for (int p = 0; p < pnum; ++p) {
        Ppatch ppatch(new Patch());
        ifstr >> *ppatch;
        ppatch->m_fix = 0;
        ppatch->m_tmp = ppatch->score2(m_pmmvps.m_nccThreshold);
        ppatch->m_vimages.clear();
        image2index(*ppatch);
        if (ppatch->m_images.empty()) break;
        pp.push_back(ppatch);
    }
    return upload(pp);
}
void PatchManager::collectPatches(const int) {
    m_ppatches.clear();
    int64_t n = 0;
    if (mvs_engine_download_patches(m_pmmvps.m_engine, 0, nullptr, &n) != 0 || n == 0) return;
    vector<mvs_patch> recs((size_t)n);
    if (mvs_engine_download_patches(m_pmmvps.m_engine, n, recs.data(), &n) != 0) return;
    m_ppatches.reserve((size_t)n);
    for (const mvs_patch& r : recs) m_ppatches.push_back(from_record(r));
}
void PatchManager::writePatches(const string prefix, bool bExportPLY, bool bExportPatch, bool) {
    collectPatches(1);
    if (bExportPLY) writePly(m_ppatches, prefix + ".ply");
    if (bExportPatch) {
        ofstream ofstr((prefix + ".patch").c_str());
        ofstr << "PATCHES" << endl << (int)m_ppatches.size() << endl;
        for (const Ppatch& pp : m_ppatches) { Patch patch = *pp; index2image(patch); ofstr << patch << "\n"; }
    }
}
void PatchManager::writePly(const vector<Ppatch>& patches, const string filename) {
    // colour = mean over m_images of PhotoSet::getColor(image, coord, m_level) (patch_manager.cpp:565-583): the level-m_level
    // projection (camera.cpp:91-99: rows 0,1 halved per level) and a bilinear sample (image.cpp:447-472) of the
    // pyramid level the engine built
    const int level = m_pmmvps.m_level, nv = m_pmmvps.m_nimages;
    vector<vector<unsigned char> > img(nv);
    vector<int> W(nv, 0), H(nv, 0);
    vector<std::array<float, 12> > P(nv);
    for (int v = 0; v < nv; ++v) {
        const Photo& ph = m_pmmvps.m_photoSet.m_photos[v];
        for (int i = 0; i < 12; ++i) P[v][i] = ph.m_projection[i];
        for (int l = 0; l < level; ++l) for (int i = 0; i < 8; ++i) P[v][i] /= 2.0f;
        if (level == 0) { img[v] = ph.m_image; W[v] = ph.m_width; H[v] = ph.m_height; continue; }
        if (!m_pmmvps.m_engine || mvs_engine_get_pyramid(m_pmmvps.m_engine, v, level, nullptr, &W[v], &H[v]) != 0) continue;
        img[v].resize((size_t)3 * W[v] * H[v]);
        if (mvs_engine_get_pyramid(m_pmmvps.m_engine, v, level, img[v].data(), &W[v], &H[v]) != 0) img[v].clear();
    }
    ofstream ofstr(filename.c_str());
    ofstr << "ply\nformat ascii 1.0\nelement vertex " << (int)patches.size()
          << "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
             "property uchar diffuse_red\nproperty uchar diffuse_green\nproperty uchar diffuse_blue\nend_header\n";
    for (const Ppatch& p : patches) {
        float colorf[3] = {0.0f, 0.0f, 0.0f};
        int denom = 0;
        for (int image : p->m_images) {
            if (image < 0 || nv <= image) continue;
            denom++;  // the reference counts every listed view (patch_manager.cpp:581)
            if (img[image].empty()) continue;
            const float* q = P[image].data();
            const float* X = p->m_coord.data();
            const float z = q[8] * X[0] + q[9] * X[1] + q[10] * X[2] + q[11] * X[3];
            if (z <= 0.0f) continue;  // Camera::project returns (-65535, -65535): nothing to sample
            const float x = (q[0] * X[0] + q[1] * X[1] + q[2] * X[2] + q[3] * X[3]) / z;
            const float y = (q[4] * X[0] + q[5] * X[1] + q[6] * X[2] + q[7] * X[3]) / z;
            if (!(x >= 0.0f && y >= 0.0f && x < (float)(W[image] - 1) && y < (float)(H[image] - 1))) continue;  // the reference reads out of bounds here
            const int lx = (int)x, ly = (int)y;
            const float dx1 = x - lx, dx0 = 1.0f - dx1, dy1 = y - ly, dy0 = 1.0f - dy1;
            const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
            const unsigned char* r0 = &img[image][(size_t)3 * ((size_t)ly * W[image] + lx)];
            const unsigned char* r1 = r0 + (size_t)3 * W[image];
            for (int c = 0; c < 3; ++c) colorf[c] += (r0[c] * f00 + r1[c] * f01) + (r0[3 + c] * f10 + r1[3 + c] * f11);
        }
        int color[3] = {128, 128, 128};
        if (denom > 0)
            for (int c = 0; c < 3; ++c) color[c] = std::min(255, (int)floorf(colorf[c] / (float)denom + 0.5f));
        ofstr << p->m_coord[0] << ' ' << p->m_coord[1] << ' ' << p->m_coord[2] << ' ' << p->m_normal[0] << ' ' << p->m_normal[1] << ' '
              << p->m_normal[2] << ' ' << color[0] << ' ' << color[1] << ' ' << color[2] << '\n';
    }
}

// ------------------------------------------------------------------ Propagate
void Propagate::init() {  // propagate.cpp:23-26
    MAX_NUM_OF_PROPAG = 2;
    MAX_NUM_OF_PATCHES = MAX_NUM_OF_PROPAG * m_pmmvps.m_csize * m_pmmvps.m_csize;
}
void Propagate::run(const int iter) {  // propagate.cpp:28-64: the drop-in boundary
    m_ecount = m_fcount0 = m_fcount1 = m_pcount = 0;
    if (m_pmmvps.m_status != 0) return;
    int r = mvs_engine_set_thresholds(m_pmmvps.m_engine, m_pmmvps.m_nccThreshold, m_pmmvps.m_nccThresholdBefore, m_pmmvps.m_depth);
    if (r == 0) r = mvs_engine_propagate(m_pmmvps.m_engine, iter, &m_counters);
    if (r != 0) { cerr << "Propagate::run: " << mvs_last_error() << endl; m_pmmvps.m_status = r; return; }
    m_ecount = m_counters.patches; m_fcount0 = m_counters.fail0; m_fcount1 = m_counters.fail1;
    m_pcount = m_counters.inserted + m_counters.replaced;
    cerr << "total pass fail0 fail1 refinepatch: " << m_ecount << " " << m_pcount << " " << m_fcount0 << " " << m_fcount1 << " "
         << m_pcount + m_fcount1 << endl;
}

// ------------------------------------------------------------------ Optim / DepthNormInit
int Optim::probe(int op, Patch& patch, float* value) {
    if (m_pmmvps.m_status != 0) return -1;
    (void)mvs_engine_set_thresholds(m_pmmvps.m_engine, m_pmmvps.m_nccThreshold, m_pmmvps.m_nccThresholdBefore, m_pmmvps.m_depth);
    const mvs_patch in = to_record(patch);
    mvs_patch out = in;
    float f = 0.0f;
    int32_t flag = 0;
    const int r = mvs_engine_probe(m_pmmvps.m_engine, op, 1, &in, nullptr, &out, &f, &flag);
    if (r != 0) { cerr << "Optim: " << mvs_last_error() << endl; m_pmmvps.m_status = r; return -1; }
    if (op != MVS_PROBE_NCC) {
        const Ppatch p = from_record(out);
        const int id = patch.m_id;
        patch = *p;
        patch.m_id = id;
    }
    if (value) *value = f;
    return flag;
}
int Optim::preProcess(Patch& patch) { return probe(MVS_PROBE_PREPROCESS, patch, nullptr); }
void Optim::refinePatch(Patch& patch, const int) { (void)probe(MVS_PROBE_REFINE, patch, nullptr); }
int Optim::postProcess(Patch& patch) { return probe(MVS_PROBE_POSTPROCESS, patch, nullptr); }
float Optim::computeNcc(const Patch& patch) {
    Patch p = patch;
    float v = -1.0f;
    (void)probe(MVS_PROBE_NCC, p, &v);
    return v;
}
// ---- DepthNormInit
namespace {
struct HostCamera {  // Camera::updateCamera (camera.cpp:65-89, getCameraCenter 295-308) + Optim::setAxesScales (optim.cpp:43-65)
    float center[4], ipscale;
};
HostCamera host_camera(const Photo& ph) {
    const float* P = ph.m_projection;
    HostCamera c;
    // centre = -M^-1 p4 (Eigen's 3x3 inverse is the cofactor formula)
    const double m[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                           (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                           (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
    const double q[3] = {P[3], P[7], P[11]};
    for (int r = 0; r < 3; ++r) c.center[r] = (float)-(inv[3 * r] * q[0] + inv[3 * r + 1] * q[1] + inv[3 * r + 2] * q[2]);
    c.center[3] = 1.0f;
    const float on = std::sqrt(P[8] * P[8] + P[9] * P[9] + P[10] * P[10]);
    const float z[3] = {P[8] / on, P[9] / on, P[10] / on};
    float x[3] = {P[0], P[1], P[2]};
    float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    const float yn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    for (int k = 0; k < 3; ++k) y[k] /= yn;
    x[0] = y[1] * z[2] - y[2] * z[1]; x[1] = y[2] * z[0] - y[0] * z[2]; x[2] = y[0] * z[1] - y[1] * z[0];
    c.ipscale = (P[0] * x[0] + P[1] * x[1] + P[2] * x[2]) + (P[4] * y[0] + P[5] * y[1] + P[6] * y[2]);
    return c;
}
}  // namespace

void DepthNormInit::sortImages(Patch& patch, const int isFixed) const {  // optim.cpp:221-258
    const float threshold = 1.0f - std::cos(10.0f * (float)M_PI / 180.0f);
    vector<int> indexes0, indexes1;
    vector<float> units0, units1;
    vector<std::array<float, 4> > rays0, rays1;
    for (int image : patch.m_images) {  // computeUnits, optim.cpp:86-107
        const HostCamera cam = host_camera(m_pmmvps.m_photoSet.m_photos[image]);
        std::array<float, 4> ray;
        for (int k = 0; k < 4; ++k) ray[k] = cam.center[k] - patch.m_coord[k];
        const float fz = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2] + ray[3] * ray[3]);
        for (int k = 0; k < 4; ++k) ray[k] /= fz;
        const float dot = ray[0] * patch.m_normal[0] + ray[1] * patch.m_normal[1] + ray[2] * patch.m_normal[2] + ray[3] * patch.m_normal[3];
        if (dot <= 0.0f) continue;
        const float scale = cam.ipscale == 0.0f ? 1.0f : (float)(2.0 * fz * (0x0001 << m_pmmvps.m_level) / cam.ipscale);  // getUnit, optim.cpp:34-41
        indexes0.push_back(image); units0.push_back(scale / dot); rays0.push_back(ray);
    }
    patch.m_images.clear();
    if (indexes0.size() < 2) return;
    if (isFixed) units0[0] = 0.0f;
    while (!indexes0.empty()) {
        const int index = (int)(std::min_element(units0.begin(), units0.end()) - units0.begin());
        patch.m_images.push_back(indexes0[index]);
        indexes1.clear(); units1.clear(); rays1.clear();
        for (int i = 0; i < (int)rays0.size(); ++i) {
            if (i == index) continue;
            indexes1.push_back(indexes0[i]);
            rays1.push_back(rays0[i]);
            float d = 0.0f;
            for (int k = 0; k < 4; ++k) d += rays0[index][k] * rays0[i][k];
            const float ftmp = std::min(threshold, std::max(threshold / 2.0f, 1.0f - d));
            units1.push_back(units0[i] * threshold / ftmp);
        }
        indexes1.swap(indexes0); units1.swap(units0); rays1.swap(rays0);
    }
}

int DepthNormInit::readDepths(vector<std::array<float, 3> >& coords) {  // depth_normal_init.cpp:94-112
    char dname[1024];
    snprintf(dname, sizeof dname, "%sply/%08d.ply", m_prefix.c_str(), 0);
    vector<double> points;
    string err;
    if (readPlyVertices(dname, points, nullptr, &err) != 0) { cerr << "DepthNormInit::readDepths: " << err << endl; return -1; }
    coords.resize(points.size() / 3);
    for (size_t i = 0; i < coords.size(); ++i) coords[i] = {(float)points[3 * i], (float)points[3 * i + 1], (float)points[3 * i + 2]};
    return 0;
}
int DepthNormInit::readNormals(vector<vector<std::array<float, 3> > >& normals) {  // depth_normal_init.cpp:114-144
    for (int i = 0; i < m_nplys - 1; ++i) {
        char nname[1024];
        snprintf(nname, sizeof nname, "%sply/%08d.ply", m_prefix.c_str(), i + 1);
        vector<double> points, norms;
        string err;
        if (readPlyVertices(nname, points, &norms, &err) != 0) { cerr << "DepthNormInit::readNormals: " << err << endl; return -1; }
        if (norms.empty()) { cerr << "DepthNormInit::readNormals: no nx ny nz in " << nname << endl; return -1; }
        const Photo& ph = m_pmmvps.m_photoSet.m_photos[i];
        if (ph.m_txtType != 2) { cerr << "Not supported: " << ph.m_txtType << endl; return -1; }  // Camera::setR, camera.cpp:180-183
        const int width = ph.m_width, height = ph.m_height;
        normals[i].assign((size_t)width * height, std::array<float, 3>{0.0f, 0.0f, 0.0f});
        for (size_t n = 0; n < points.size() / 3; ++n) {
            const int x = (int)points[3 * n], y = (int)points[3 * n + 1];
            if (x < 0 || width <= x || y < 0 || height <= y) continue;  // the reference writes out of bounds here
            const float v[3] = {(float)norms[3 * n], (float)norms[3 * n + 1], (float)norms[3 * n + 2]};
            std::array<float, 3> r;
            for (int k = 0; k < 3; ++k) r[k] = ph.m_R[3 * k] * v[0] + ph.m_R[3 * k + 1] * v[1] + ph.m_R[3 * k + 2] * v[2];
            normals[i][(size_t)y * width + x] = r;
        }
    }
    return 0;
}
int DepthNormInit::buildPatches(vector<Ppatch>& ppatches) {  // depth_normal_init.cpp:34-91
    if (m_nplys <= 0) init(m_pmmvps.m_prefix, m_pmmvps.m_nimages + 1);  // pmmvps.cpp:45
    vector<std::array<float, 3> > coords;
    if (readDepths(coords) != 0) return -1;
    vector<vector<std::array<float, 3> > > normals((size_t)std::max(0, m_nplys - 1));
    if (readNormals(normals) != 0) return -1;
    const PhotoSet& ps = m_pmmvps.m_photoSet;
    for (size_t i = 0; i < coords.size(); ++i) {
        Ppatch ppatch(new Patch());
        ppatch->m_coord = {coords[i][0], coords[i][1], coords[i][2], 1.0f};
        vector<int> images;
        float n3[3] = {0.0f, 0.0f, 0.0f};
        for (int image = 0; image < m_pmmvps.m_nimages && image < (int)normals.size(); ++image) {
            float icoord[3];
            ps.project(image, ppatch->m_coord, 0, icoord);
            const int x = (int)floorf(icoord[0] + 0.5f), y = (int)floorf(icoord[1] + 0.5f);
            if (ps.getMask(image, x, y, 0) <= 0) continue;  // also skips every view without a mask (getMask = -1), as the reference does
            const std::array<float, 3>& nv = normals[image][(size_t)y * ps.getWidth(image, 0) + x];
            for (int k = 0; k < 3; ++k) n3[k] += nv[k];
            images.push_back(image);
        }
        float norm = std::sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
        if (images.size() < 2 || norm == 0.0f) continue;
        ppatch->m_images = images;
        for (int k = 0; k < 3; ++k) n3[k] /= (float)images.size();
        norm = std::sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
        for (int k = 0; k < 3; ++k) n3[k] /= norm;
        ppatch->m_normal = {n3[0], n3[1], n3[2], -(coords[i][0] * n3[0] + coords[i][1] * n3[1] + coords[i][2] * n3[2])};
        sortImages(*ppatch, 0);
        // the reference adds a patch even when sortImages left fewer than two views (m_images empty: it then sits in no
        // grid cell); the engine's pool holds no such patch
        if (ppatch->m_images.empty()) continue;
        ppatch->m_nimages = (int)ppatch->m_images.size();
        ppatches.push_back(ppatch);  // setGrids / addPatch: the engine computes cells and depth maps on upload
    }
    return 0;
}
void DepthNormInit::createPatches() {  // depth_normal_init.cpp:29-91
    if (m_isTest) {
        if (m_pmmvps.m_patchManager.readPatches() != 0) cerr << "DepthNormInit::createPatches: no ply/00000000.patch under " << m_pmmvps.m_prefix << endl;
        return;
    }
    vector<Ppatch> pp;
    if (buildPatches(pp) != 0) { m_pmmvps.m_status = MVS_ERR_ARG; return; }
    m_pmmvps.m_patchManager.addPatches(pp);
}

// ------------------------------------------------------------------ Filter
void Filter::run() {  // filter.cpp:25-49
    int64_t r4[4] = {0, 0, 0, 0};
    if (m_pmmvps.m_status != 0) return;
    int r = mvs_engine_set_thresholds(m_pmmvps.m_engine, m_pmmvps.m_nccThreshold, m_pmmvps.m_nccThresholdBefore, m_pmmvps.m_depth);
    if (r == 0) r = mvs_engine_filter(m_pmmvps.m_engine, r4);
    if (r != 0) { cerr << "Filter::run: " << mvs_last_error() << endl; m_pmmvps.m_status = r; return; }
    for (int k = 0; k < 4; ++k) m_removed[k] = r4[k];
    cerr << "FilterOutside/Exact/Neighbor/Groups removed: " << r4[0] << " " << r4[1] << " " << r4[2] << " " << r4[3] << endl;
}

// ------------------------------------------------------------------ PmMvps
PmMvps::PmMvps() : m_dnInit(*this), m_patchManager(*this), m_propagate(*this), m_optim(*this), m_filter(*this) {}
PmMvps::~PmMvps() { if (m_engine) mvs_engine_destroy(m_engine); }

int PmMvps::createEngine(float maxAngle, float quad) {
    mvs_config cfg;
    mvs_default_config(&cfg);
    cfg.nviews = m_nimages; cfg.level = m_level; cfg.csize = m_csize; cfg.wsize = m_wsize;
    cfg.minImageNum = m_minImageNumThreshold; cfg.nccThreshold = m_nccThreshold;
    cfg.maxAngleThreshold = maxAngle; cfg.quadThreshold = quad;
    cfg.depth = 0; cfg.seed = m_seed; cfg.refine_steps = m_refineSteps; cfg.view_propagation = m_viewPropagation; cfg.literal_groups = m_literalGroups;
    cfg.enable_check = 1;  // Optim::check from m_depth >= 2 (optim.cpp:292)
    cfg.device = m_device;
    if (m_world > 1 || !m_commIdFile.empty()) { cfg.shard_index = m_rank; cfg.shard_count = m_world; }
    // the reference's m_images / m_vimages are unbounded (optim.cpp:165-205); this library is linked against ONE build of the
    // engine, whose lists hold mvs_list_cap() views: a data set with more views belongs to the next build of the pair
    // (libmvskit_host_cap32.so / _cap64.so link libmvskit_engine_cap32.so / _cap64.so), else its lists are cut short
    if (m_nimages > mvs_list_cap()) {
        cerr << "PmMvps::init: " << m_nimages << " views, but this build keeps " << mvs_list_cap()
             << " views per patch list: use the host library built for the larger engine (cap32: up to 32 views, cap64: up to 64)" << endl;
        if (m_strictListCap) return MVS_ERR_ARG;
    }
    // three engine libraries export the same symbols with two record widths: a host library that resolved against the wrong one
    // (LD_PRELOAD, load order, rpath) would corrupt records silently on upload and download
    if (mvs_patch_bytes() != (int)sizeof(mvs_patch)) {
        cerr << "PmMvps::init: the loaded engine library has " << mvs_patch_bytes() << "-byte patch records, this host library was built for "
             << sizeof(mvs_patch) << " (MVS_MAX_IMAGES " << MVS_MAX_IMAGES << "): wrong libmvskit_engine*.so resolved" << endl;
        return MVS_ERR_ARG;
    }
    int r = mvs_engine_create(&cfg, &m_engine);
    if (r != 0) { cerr << "PmMvps::init: " << mvs_last_error() << endl; return r; }
    vector<mvs_view_desc> views(m_nimages);
    for (int i = 0; i < m_nimages; ++i) {
        const Photo& ph = m_photoSet.m_photos[i];
        views[i].width = ph.m_width; views[i].height = ph.m_height;
        memcpy(views[i].P, ph.m_projection, sizeof views[i].P);
        views[i].rgb = ph.m_image.data();
        views[i].mask = ph.m_mask.empty() ? nullptr : ph.m_mask.data();
    }
    r = mvs_engine_set_views(m_engine, m_nimages, views.data());
    if (r != 0) { cerr << "PmMvps::init: " << mvs_last_error() << endl; return r; }
    return joinRanks();
}

void PmMvps::setRanks(int rank, int world, const string& idFile, int device) {
    m_rank = rank; m_world = world; m_commIdFile = idFile;
    m_device = device >= 0 ? device : rank;
}
// The communicator id travels through a file: rank 0 writes it (temporary name + rename, so a reader never sees half of
// it), the other ranks wait for the file.  ncclCommInitRank inside mvs_engine_comm_init is the rendezvous itself.
// A file left behind by an earlier job must not be taken for this job's: the file carries, ahead of the id, a magic word
// and the job nonce (PmMvps::m_jobNonce -- MVS_JOB_NONCE in the environment, the same on every rank of a job, e.g. the
// launcher's job id; without one, the file's age decides: anything older than the reader's own start is stale), and rank 0
// removes the file once its communicator stands -- by then every rank has read it, or ncclCommInitRank would not have returned.
namespace {
struct IdFile { char magic[8]; unsigned long long nonce; long long written_ns; unsigned char id[MVS_COMM_ID_BYTES]; };
long long wall_ns() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); }
}
int PmMvps::joinRanks() {
    if (m_commIdFile.empty()) return 0;
    if (m_jobNonce == 0) { if (const char* n = getenv("MVS_JOB_NONCE")) m_jobNonce = std::strtoull(n, nullptr, 0); }
    // no explicit nonce: whatever the launcher gives every rank of ONE job and changes from job to job -- torchrun's run id, a
    // scheduler's job id, the rendezvous port (hashed; the age test below stays the last resort for bare launches)
    if (m_jobNonce == 0 && m_world > 1) {
        const char* keys[] = {"TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID", "MASTER_PORT"};
        unsigned long long h = 1469598103934665603ull;  // FNV-1a over "key=value" of the ones present
        bool any = false;
        for (const char* k : keys) {
            const char* v = getenv(k);
            if (!v || !*v) continue;
            any = true;
            for (const char* c = k; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
            for (const char* c = v; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
        }
        if (any) m_jobNonce = h | 1ull;
    }
    IdFile f;
    memset(&f, 0, sizeof f);
    if (m_rank == 0) {
        std::remove(m_commIdFile.c_str());  // whatever an earlier job left there
        if (int r = mvs_comm_unique_id(f.id)) { cerr << "PmMvps::init: " << mvs_last_error() << endl; return r; }
        memcpy(f.magic, "MVSKCID1", 8); f.nonce = m_jobNonce; f.written_ns = wall_ns();
        const string tmp = m_commIdFile + ".tmp";
        std::ofstream o(tmp.c_str(), std::ios::binary);
        o.write(reinterpret_cast<const char*>(&f), sizeof f);
        o.close();
        if (!o || std::rename(tmp.c_str(), m_commIdFile.c_str()) != 0) { cerr << "PmMvps::init: cannot write " << m_commIdFile << endl; return MVS_ERR_ARG; }
    } else {
        const long long started = m_startedNs ? m_startedNs : wall_ns() - 60ll * 1000000000ll;  // ranks of one job start within a minute of each other
        bool got = false;
        for (int tries = 0; tries < 1200 && !got; ++tries) {  // up to two minutes
            std::ifstream i(m_commIdFile.c_str(), std::ios::binary);
            IdFile g;
            if (i.is_open() && i.read(reinterpret_cast<char*>(&g), sizeof g) && i.gcount() == (std::streamsize)sizeof g && memcmp(g.magic, "MVSKCID1", 8) == 0 &&
                (m_jobNonce != 0 ? g.nonce == m_jobNonce : g.written_ns >= started)) { f = g; got = true; }
            else std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) { cerr << "PmMvps::init: no communicator id of this job in " << m_commIdFile << endl; return MVS_ERR_STATE; }
    }
    if (int r = mvs_engine_comm_init(m_engine, f.id, m_rank, m_world)) { cerr << "PmMvps::init: " << mvs_last_error() << endl; return r; }
    if (m_rank == 0) std::remove(m_commIdFile.c_str());
    return 0;
}

void PmMvps::init(const Option& option, const PhotoSet& photos) {  // pmmvps.cpp:18-68
    m_status = 0;
    m_images = option.m_images; m_nimages = option.m_nimages; m_nillums = option.m_nillums;
    m_prefix = option.m_prefix; m_level = option.m_level; m_csize = option.m_csize;
    m_nccThreshold = option.m_nccThreshold; m_wsize = option.m_wsize; m_minImageNumThreshold = option.m_minImageNum;
    m_visdata = option.m_visdata; m_visdata2 = option.m_visdata2;
    m_tau = std::min(option.m_minImageNum * 2, m_nimages);
    m_depth = 0;
    m_photoSet = photos;
    m_angleThreshold0 = (float)(60.0f * M_PI / 180.0f);
    m_angleThreshold1 = (float)(60.0f * M_PI / 180.0f);
    m_countThreshold1 = 4;
    m_neighborThreshold = 0.5f; m_neighborThreshold1 = 1.0f; m_neighborThreshold2 = 1.0f;
    m_nccThresholdBefore = m_nccThreshold - 0.3f;
    m_maxAngleThreshold = option.m_maxAngleThreshold;
    m_quadThreshold = option.m_quadThreshold;
    if (int r = createEngine(option.m_maxAngleThreshold, option.m_quadThreshold)) { m_status = r; return; }
    m_patchManager.init();
    m_dnInit.init(m_prefix, m_nimages + 1);  // pmmvps.cpp:45
    m_propagate.init();
    m_optim.init();
    m_filter.init();
}
void PmMvps::init(const Option& option) {
    PhotoSet ps;
    if (option.m_status != 0) { m_status = MVS_ERR_ARG; return; }
    ps.init(option.m_images, option.m_prefix, option.m_nimages, option.m_nillums, option.m_level + 3, option.m_wsize, 1);
    if (ps.m_status != 0) { m_status = MVS_ERR_ARG; return; }
    init(option, ps);
}
void PmMvps::updateThreshold() { m_nccThreshold -= 0.05f; m_nccThresholdBefore -= 0.05f; m_countThreshold1 = 2; }

void PmMvps::run() {  // pmmvps.cpp:76-114
    if (m_status != 0) return;
    if (m_writeFiles) m_dnInit.createPatches();  // pmmvps.cpp:83
    ++m_depth;
    for (int iter = 0; iter < ITER; ++iter) {
        cerr << "\n---------------------\nIteration: " << iter << "\n---------------------" << endl;
        m_propagate.run(iter);
        if (m_status != 0) return;
        if (m_writeFiles) m_patchManager.writePatches(m_prefix + "ply/refined_patches_before_refine_" + std::to_string(iter), true, false, false);
        m_filter.run();  // pmmvps.cpp:101
        if (m_status != 0) return;
        updateThreshold();
        ++m_depth;
        if (m_writeFiles) m_patchManager.writePatches(m_prefix + "ply/refined_patches_" + std::to_string(iter), true, false, false);
    }
}

}  // namespace mvshost

// ------------------------------------------------------------------ C entry used by the tests: run the mirror on in-memory inputs
static std::string g_ply_out;
// the next mvshost_run also writes its final patches as a PLY file (PatchManager::writePly) to `path`; "" switches it off
extern "C" void mvshost_set_ply_output(const char* path) { g_ply_out = path ? path : ""; }
// the next mvshost_run is rank `rank` of `world` (PmMvps::setRanks); world = 0 switches it off again
static struct { int rank = 0, world = 0, device = 0; std::string id_file; } g_ranks;
static bool g_run_filter = true;
extern "C" void mvshost_set_ranks(int rank, int world, const char* id_file, int device) {
    g_ranks.rank = rank; g_ranks.world = world; g_ranks.device = device; g_ranks.id_file = id_file ? id_file : "";
}
extern "C" void mvshost_set_filter(int on) { g_run_filter = on != 0; }
extern "C" int mvshost_run(int nviews, int width, int height, const float* P /*[n][12]*/, const unsigned char* rgb /*[n][H][W][3]*/,
                           int level, int csize, int wsize, int minImageNum, float nccThreshold, unsigned seed, int iters,
                           long long nseeds, const mvs_patch* seeds, long long cap, mvs_patch* out, long long* nout, long long* patches_total) {
    using namespace mvshost;
    Option option;
    option.m_nimages = nviews; option.m_nillums = 1; option.m_level = level; option.m_csize = csize; option.m_wsize = wsize;
    option.m_minImageNum = minImageNum; option.m_nccThreshold = nccThreshold; option.m_flag = -1;
    for (int i = 0; i < nviews; ++i) option.m_images.push_back(i);
    PhotoSet ps;
    ps.m_images = option.m_images;
    for (int i = 0; i < nviews; ++i) {
        ps.m_dict[i] = i;
        ps.setPhoto(i, width, height, P + 12 * i, rgb + (size_t)i * width * height * 3, nullptr);
    }
    PmMvps pmmvps;
    pmmvps.m_seed = seed; pmmvps.ITER = iters; pmmvps.m_writeFiles = false;
    if (g_ranks.world > 0) pmmvps.setRanks(g_ranks.rank, g_ranks.world, g_ranks.id_file, g_ranks.device);
    pmmvps.init(option, ps);
    if (pmmvps.m_status) return pmmvps.m_status;
    if (int r = mvs_engine_upload_patches(pmmvps.m_engine, nseeds, seeds)) return r;
    long long total = 0;
    ++pmmvps.m_depth;
    for (int iter = 0; iter < iters; ++iter) {
        pmmvps.m_propagate.run(iter);
        if (pmmvps.m_status) return pmmvps.m_status;
        total += pmmvps.m_propagate.m_ecount;
        if (g_run_filter) pmmvps.m_filter.run();
        if (pmmvps.m_status) return pmmvps.m_status;
        pmmvps.updateThreshold();
        ++pmmvps.m_depth;
    }
    pmmvps.m_patchManager.collectPatches();
    const auto& pp = pmmvps.m_patchManager.m_ppatches;
    *nout = (long long)pp.size();
    for (long long i = 0; i < std::min<long long>(cap, (long long)pp.size()); ++i) out[i] = to_record(*pp[i]);
    if (!g_ply_out.empty()) pmmvps.m_patchManager.writePly(pp, g_ply_out);
    if (patches_total) *patches_total = total;
    return 0;
}

// The reference's own driver sequence on a dataset directory (test/test.cpp:155-161): Option::init(prefix, "option"),
// PmMvps::init(option) (cameras txt/%08d.txt, images image/%04d%04d.ppm, masks mask/%08d.pgm), PmMvps::run (seeds from
// ply/00000000.patch; ply/refined_patches_<iter>.ply written after every iteration).  Returns the final patches.
// the next mvshost_run_dataset takes its seeds from ply/00000000.ply + the per-view normal maps (DepthNormInit::m_isTest = 0)
static bool g_seed_plys = false;
extern "C" void mvshost_set_seed_plys(int on) { g_seed_plys = on != 0; }
extern "C" int mvshost_run_dataset(const char* prefix, int iters, unsigned seed, long long cap, mvs_patch* out, long long* nout) {
    using namespace mvshost;
    Option option;
    option.init(prefix, "option");
    if (option.m_status != 0) return MVS_ERR_ARG;
    PmMvps pmmvps;
    pmmvps.m_seed = seed; pmmvps.ITER = iters; pmmvps.m_writeFiles = true;
    pmmvps.init(option);
    pmmvps.m_dnInit.m_isTest = g_seed_plys ? 0 : 1;
    pmmvps.run();
    if (pmmvps.m_status) return pmmvps.m_status;
    pmmvps.m_patchManager.collectPatches();
    const auto& pp = pmmvps.m_patchManager.m_ppatches;
    *nout = (long long)pp.size();
    for (long long i = 0; i < std::min<long long>(cap, (long long)pp.size()); ++i) out[i] = to_record(*pp[i]);
    pmmvps.m_patchManager.writePatches(std::string(prefix) + "ply/final", false, true, false);  // the .patch text format
    return 0;
}

// Option::init on a file: out_i = {nimages, level, csize, wsize, minImageNum, flag, #images}, out_f = {threshold, maxAngle, quad}
extern "C" int mvshost_option_probe(const char* prefix, const char* option, int* out_i, float* out_f) {
    mvshost::Option o;
    o.init(prefix, option);
    const int r = o.m_status;
    out_i[0] = o.m_nimages; out_i[1] = o.m_level; out_i[2] = o.m_csize; out_i[3] = o.m_wsize; out_i[4] = o.m_minImageNum;
    out_i[5] = o.m_flag; out_i[6] = (int)o.m_images.size();
    out_f[0] = o.m_nccThreshold; out_f[1] = o.m_maxAngleThreshold; out_f[2] = o.m_quadThreshold;
    return r;
}
// Patch text format (patch.cpp:31-79): parse `text`, write it back into `out` (cap bytes); returns the length
extern "C" int mvshost_patch_roundtrip(const char* text, char* out, int cap, mvs_patch* rec) {
    std::istringstream is(text);
    mvshost::Patch p;
    is >> p;
    if (rec) *rec = mvshost::to_record(p);
    std::ostringstream os;
    os << p;
    const std::string s = os.str();
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
// Camera text (camera.cpp:27-63): returns the 3x4 projection
// Optim::preProcess -> refinePatch -> postProcess on each of n seeds through the mirror's Optim class (one patch per call);
// flags[i] = -1 where a stage rejected the patch, ncc[i] = Optim::computeNcc of the seed
extern "C" int mvshost_optim_chain(int nviews, int width, int height, const float* P, const unsigned char* rgb, int level, int csize, int wsize,
                                   int minImageNum, float nccThreshold, unsigned seed, long long n, const mvs_patch* seeds, mvs_patch* out, int* flags, float* ncc) {
    using namespace mvshost;
    Option option;
    option.m_nimages = nviews; option.m_nillums = 1; option.m_level = level; option.m_csize = csize; option.m_wsize = wsize;
    option.m_minImageNum = minImageNum; option.m_nccThreshold = nccThreshold; option.m_flag = -1;
    for (int i = 0; i < nviews; ++i) option.m_images.push_back(i);
    PhotoSet ps;
    ps.m_images = option.m_images;
    for (int i = 0; i < nviews; ++i) { ps.m_dict[i] = i; ps.setPhoto(i, width, height, P + 12 * i, rgb + (size_t)i * width * height * 3, nullptr); }
    PmMvps pmmvps;
    pmmvps.m_seed = seed; pmmvps.m_writeFiles = false;
    pmmvps.init(option, ps);
    if (pmmvps.m_status) return pmmvps.m_status;
    for (long long i = 0; i < n; ++i) {
        Ppatch pp = from_record(seeds[i]);
        ncc[i] = pmmvps.m_optim.computeNcc(*pp);
        int f = pmmvps.m_optim.preProcess(*pp);
        if (f == 0) { pmmvps.m_optim.refinePatch(*pp, 100); f = pmmvps.m_optim.postProcess(*pp); }
        flags[i] = f;
        out[i] = to_record(*pp);
        if (pmmvps.m_status) return pmmvps.m_status;
    }
    return 0;
}
extern "C" int mvshost_pbm_probe(const char* mname, int width, int height, unsigned char* out) {
    mvshost::Photo ph;
    ph.m_width = width; ph.m_height = height;
    if (ph.readPbmMask(mname) != 0) return -1;
    memcpy(out, ph.m_mask.data(), ph.m_mask.size());
    return 0;
}
extern "C" int mvshost_camera_probe(const char* cname, float* P12) {
    mvshost::Photo ph;
    const int r = ph.initCamera(cname);
    memcpy(P12, ph.m_projection, sizeof ph.m_projection);
    return r;
}
// Photo::readJpeg on a file: interleaved RGB into out (capacity bytes); returns 0, -1 (unreadable) or -2 (capacity)
extern "C" int mvshost_jpeg_probe(const char* iname, int* width, int* height, unsigned char* out, long long capacity) {
    mvshost::Photo ph;
    if (ph.readJpeg(iname) != 0) return -1;
    *width = ph.m_width; *height = ph.m_height;
    if (out) { if ((long long)ph.m_image.size() > capacity) return -2; memcpy(out, ph.m_image.data(), ph.m_image.size()); }
    return 0;
}
// DepthNormInit's PLY branch (m_isTest = 0) on a dataset directory, host part only (no engine): Option::init, PhotoSet::init,
// DepthNormInit::buildPatches.  Returns the number of seed patches (the first `cap` in out) or a negative status.
extern "C" long long mvshost_seeds_from_plys(const char* prefix, long long cap, mvs_patch* out) {
    using namespace mvshost;
    Option option;
    option.init(prefix, "option");
    if (option.m_status != 0) return -1;
    PmMvps pmmvps;
    pmmvps.m_images = option.m_images; pmmvps.m_nimages = option.m_nimages; pmmvps.m_prefix = option.m_prefix;
    pmmvps.m_level = option.m_level; pmmvps.m_csize = option.m_csize;
    pmmvps.m_photoSet.init(option.m_images, option.m_prefix, option.m_nimages, option.m_nillums, option.m_level + 3, option.m_wsize, 1);
    if (pmmvps.m_photoSet.m_status != 0) return -2;
    pmmvps.m_dnInit.init(option.m_prefix, option.m_nimages + 1);
    pmmvps.m_dnInit.m_isTest = 0;
    vector<Ppatch> pp;
    if (pmmvps.m_dnInit.buildPatches(pp) != 0) return -3;
    for (long long i = 0; i < std::min<long long>(cap, (long long)pp.size()); ++i) out[i] = to_record(*pp[i]);
    return (long long)pp.size();
}
