/*
 * pmmvs_oracle.cpp -- CPU ORACLE (test infrastructure, see pmmvs_oracle.h).
 *
 * Dependency-free C++17 restatement of the reference hot path
 *   Propagate::run -> propagatePmImage -> propagatePatch -> generatePatch /
 *   Optim::{preProcess, refinePatch(cost_func), postProcess} + PatchManager grid ops.
 * Every function cites the reference file:line it follows (paths under /root/reference).
 *
 * PARITY UNPINNED (no golden vectors exist in the reference; it cannot be built here).
 *
 * Two schedules:
 *   FAITHFUL  sequential raster sweep over views and cells, live vector<> cell lists, per-call
 *             reseeded minstd_rand0 -- the reference's execution model (defects D1/D2 patched).
 *   ENGINE    the parallel schedule the MI355X engine implements: views are Jacobi, each
 *             iteration is two red/black colour passes, one destination cell is processed
 *             sequentially exactly like propagatePatch, cross-cell effects are committed
 *             between passes, counter-based RNG.
 *
 * Arithmetic conventions shared with the HIP engine (so TREE64 results can be compared
 * bit for bit): fp32, no implicit contraction (-ffp-contract=off), every dot product is a
 * left-to-right fmaf chain, own polynomial sin/cos/asin/acos/atan, the NLopt BOBYQA call
 * (optim.cpp:511-524, third party, absent) replaced by a fixed-budget halving random search
 * in the same (depth, angle1, angle2) parametrisation and bounds.
 */
#include "pmmvs_oracle.h"

#include <algorithm>
#include <iterator>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;

/* Storage of Patch::m_images / m_vimages inside the restatement.  The default build stores what a record holds (32); the
 * "wide" build (make wide: -DORC_WIDE_LISTS) stores 64, enough for the reference's unbounded lists on a 48-view scene, and
 * exists to measure what the engine's 16-view truncation costs (tests/test_oracle_kat.py::test_list_cap_48_views). */
#ifdef ORC_WIDE_LISTS
constexpr int MAXI = 64;
#else
constexpr int MAXI = ORC_MAX_IMAGES;
#endif
constexpr int LISTCAP_DEFAULT = 16; /* engine limit: m_images / m_vimages are truncated to 16 views (orc_config.list_cap overrides) */
#define LISTCAP (s.list_cap)
constexpr int NEWBASE = 0x40000000; /* provisional ids of patches staged by a destination cell */

/* ------------------------------------------------------------------ small vectors */
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
inline float dot4(const V4& a, const V4& b) { return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x))); }
inline float dot3(const V3& a, const V3& b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
inline float norm4(const V4& a) { return sqrtf(dot4(a, a)); }
inline float norm3(const V3& a) { return sqrtf(dot3(a, a)); }
inline V4 sub4(const V4& a, const V4& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 add4(const V4& a, const V4& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 mul4(const V4& a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 div4(const V4& a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline V3 sub3(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add3(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul3(const V3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 div3(const V3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
/* v / |v| as v * (1/|v|): Eigen 3.2 evaluates vector / scalar as a multiplication by the inverse */
inline V4 nrm4(const V4& a) { const float inv = 1.0f / norm4(a); return {a.x * inv, a.y * inv, a.z * inv, a.w * inv}; }
inline V3 nrm3(const V3& a) { const float inv = 1.0f / norm3(a); return {a.x * inv, a.y * inv, a.z * inv}; }
inline V4 scl4(const V4& a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V3 scl3(const V3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 cross3(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

/* ------------------------------------------------------------------ deterministic libm subset
 * Single-precision Cephes-style kernels, fixed evaluation order, only + - * / sqrt, so the HIP
 * engine reproduces them bit for bit.  Domain of pm_sinf/pm_cosf: |x| <= pi (decode() only needs
 * |x| < pi/2, optim.cpp:496-497). */
constexpr float PIO2_HI = 1.57079625129699707031f; /* float(pi/2) */
constexpr float PIO2_LO = 7.54978941586159635335e-08f;
constexpr float PIO4_F = 0.78539816339744830962f;
constexpr float PI_F = 3.14159265358979323846f;

inline float k_sinf(float x) { /* |x| <= pi/4 */
    float z = x * x;
    float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
    p = p * z - 1.6666654611e-1f;
    return x + x * z * p;
}
inline float k_cosf(float x) { /* |x| <= pi/4 */
    float z = x * x;
    float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + z * z * p;
}
float pm_sinf(float x) {
    float a = fabsf(x);
    float r;
    if (a <= PIO4_F) r = k_sinf(a);
    else if (a <= 3.0f * PIO4_F) r = k_cosf((PIO2_HI - a) + PIO2_LO);
    else r = k_sinf((PI_F - a));
    return x < 0.0f ? -r : r;
}
float pm_cosf(float x) {
    float a = fabsf(x);
    if (a <= PIO4_F) return k_cosf(a);
    if (a <= 3.0f * PIO4_F) return k_sinf((PIO2_HI - a) + PIO2_LO);
    return -k_cosf(PI_F - a);
}
float pm_asinf(float x) {
    float a = fabsf(x);
    if (a > 1.0f) a = 1.0f;
    float z, xx;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); xx = sqrtf(z); }
    else { z = a * a; xx = a; }
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    float r = p * z * xx + xx;
    if (flag) { r = r + r; r = (PIO2_HI - r) + PIO2_LO; }
    return x < 0.0f ? -r : r;
}
float pm_acosf(float x) {
    if (x < -0.5f) return PI_F - 2.0f * pm_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * pm_asinf(sqrtf(0.5f * (1.0f - x)));
    return PIO2_HI - pm_asinf(x);
}
float pm_atanf(float v) {
    float x = fabsf(v), y;
    if (x > 2.414213562373095f) { y = PIO2_HI; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PIO4_F; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    y = y + (p * z * x + x);
    return v < 0.0f ? -y : y;
}

/* ------------------------------------------------------------------ counter-based RNG (engine) */
inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
inline uint32_t rng_hash(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
    uint32_t h = mix32(seed ^ 0x9e3779b9u);
    h = mix32(h ^ a) + 0x85ebca6bu;
    h = mix32(h ^ b) + 0xc2b2ae35u;
    h = mix32(h ^ c) + 0x27d4eb2fu;
    h = mix32(h ^ d) + 0x165667b1u;
    h = mix32(h ^ e);
    return h;
}
/* uniform in [-0.5, 0.5), exact in fp32 */
inline float rng_uniform(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
    return (float)(rng_hash(seed, a, b, c, d, e) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

/* ------------------------------------------------------------------ data */
struct View {
    std::vector<int> W, H;                       /* image.cpp:162-165 */
    std::vector<std::array<float, 12>> P;        /* camera.cpp:91-100 */
    std::vector<std::array<float, 9>> Minv;      /* inverse of the 3x3 block, per level */
    V4 center, oaxis;                            /* camera.cpp:65-72 */
    V3 xaxis, yaxis, zaxis;                      /* optim.cpp:43-55 */
    float ipscale;                               /* optim.cpp:57-64 */
    std::vector<std::vector<uint8_t>> img;       /* image.hpp:76, RGB interleaved */
    std::vector<std::vector<uint8_t>> mask;      /* image.hpp:78 */
    int gw = 0, gh = 0;                          /* patch_manager.cpp:36-37 */
    bool set = false;
};

struct Patch { /* patch.hpp:23-67 */
    V4 coord{0, 0, 0, 1}, normal{0, 0, 0, 0};
    float ncc = -1.0f, dscale = 0.0f, ascale = 0.0f, tmp = 0.0f;
    int nimg = 0, img[MAXI], gx[MAXI], gy[MAXI];
    int nvimg = 0, vimg[MAXI], vgx[MAXI], vgy[MAXI];
    bool alive = true;
    int sweep_view = -1, dest_cell = -1; /* engine: where it was created */
};

struct Tex { /* one grabbed 7x7 texture: channel-major, 64 lanes, lanes >= n are zero */
    float c[3][64];
    float inv; /* 1 / msd after normalize_tex; the samples stay centred, the scale is applied in dot_tex */
    float ave[3]; /* channel means found by normalize_tex (the pivots of the class-lane evaluations, see tex_stats_class16) */
    bool ok;
};

struct Scene;

/* Context of one destination cell in the engine schedule: the live list of (view, cell) and the
 * patches created there in this pass, which are not in the pool yet. */
struct DestCtx {
    int v = -1, cell = -1;
    std::vector<int> list;      /* ids, sorted desc (ncc, id asc); ids >= NEWBASE index `staged` */
    std::vector<Patch> staged;  /* creation order */
    std::vector<int> kills;     /* pool ids evicted */
};

struct Scene {
    orc_config cfg;
    int tau = 0;                 /* pmmvps.cpp:32 */
    int maxLevel = 0;            /* pmmvps.cpp:36 */
    int cap = 0;                 /* MAX_NUM_OF_PATCHES, propagate.cpp:25 */
    float nccThreshold, nccThresholdBefore;
    float angleThreshold0, angleThreshold1;      /* pmmvps.cpp:54-55 */
    float neighborThreshold, neighborThreshold1, neighborThreshold2; /* pmmvps.cpp:59-61 */
    float cosAngle0, cosAngle1;                  /* cosf() of the above */
    float cosMinAngle, cosMaxAngle;              /* checkAngles window as cosines */
    float cosNeighborTypo, cosNeighbor120;       /* pmmvps.cpp:124 (typo kept, D6) and :150 */
    float sortThreshold;                         /* optim.cpp:222 */
    float ascaleConst;                           /* optim.cpp:487 */
    float inv_sz, inv_3sz;                       /* 1/(wsize^2), 1/(3 wsize^2): the means of optim.cpp:924,932,608 as multiplications */
    int depth = 0;
    std::vector<View> views;
    std::vector<Patch> pool;
    /* faithful: live lists */
    std::vector<std::vector<std::vector<int>>> pgrids, vpgrids; /* patch_manager.hpp:96-98 */
    std::vector<std::vector<int>> dpgrids;                      /* patch_manager.hpp:100, -1 = m_MAXDEPTH */
    /* engine: CSR snapshot */
    std::vector<std::vector<int>> csr_start, csr_ids, vcsr_start, vcsr_ids;
    std::vector<DestCtx> staged_cells; /* results of the last engine pass, order (view, cell) */
    int list_cap = LISTCAP_DEFAULT;              /* m_images / m_vimages are truncated to this many views */
    mutable std::atomic<int64_t> list_truncations{0}; /* addImages / setVImagesVGrids / filterExact calls that wanted a longer list */
    int64_t cell_budget = 0;
    double last_sweep_seconds = 0.0; /* engine schedule: wall time of the last colour pass's parallel loop */
    double time_budget = 0.0; /* stop the sweep (faithful) / the colour pass (engine) after this many seconds (0 = off) */
    bool finalized = false;
    orc_counters cnt{};
};

/* ------------------------------------------------------------------ camera (image/camera.cpp) */
void invert3(const float* P, float* Minv) { /* Eigen Matrix3f::inverse, camera.cpp:304,335; done in double */
    double a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double inv[9] = {A, -(b * i - c * h), b * f - c * e, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * e - b * d};
    for (int k = 0; k < 9; ++k) Minv[k] = (float)(inv[k] / det);
}

/* Camera::project, camera.cpp:310-326 */
inline V3 project(const View& vw, const V4& X, int level) {
    const float* P = vw.P[level].data();
    float r0 = fma_(P[3], X.w, fma_(P[2], X.z, fma_(P[1], X.y, P[0] * X.x)));
    float r1 = fma_(P[7], X.w, fma_(P[6], X.z, fma_(P[5], X.y, P[4] * X.x)));
    float r2 = fma_(P[11], X.w, fma_(P[10], X.z, fma_(P[9], X.y, P[8] * X.x)));
    if (r2 <= 0.0f) return {-65535.0f, -65535.0f, -1.0f};
    const float inv = 1.0f / r2; /* icoord / icoord(2), camera.cpp:318, as a multiplication by the inverse */
    V3 ic{r0 * inv, r1 * inv, 1.0f};
    const float lo = (float)(INT_MIN + 3.0f), hi = (float)(INT_MAX - 3.0f);
    ic.x = std::max(lo, std::min(hi, ic.x));
    ic.y = std::max(lo, std::min(hi, ic.y));
    return ic;
}

/* Camera::unproject, camera.cpp:329-337 (icoord.z carries the depth) */
inline V4 unproject(const View& vw, const V3& ic, int level) {
    const float* P = vw.P[level].data();
    const float* M = vw.Minv[level].data();
    V3 b{ic.x - P[3], ic.y - P[7], ic.z - P[11]};
    return {fma_(M[2], b.z, fma_(M[1], b.y, M[0] * b.x)), fma_(M[5], b.z, fma_(M[4], b.y, M[3] * b.x)),
            fma_(M[8], b.z, fma_(M[7], b.y, M[6] * b.x)), 1.0f};
}

void setup_camera(Scene& s, View& vw, const float* P0) {
    vw.P.resize(s.maxLevel);
    vw.Minv.resize(s.maxLevel);
    for (int k = 0; k < 12; ++k) vw.P[0][k] = P0[k];
    for (int l = 1; l < s.maxLevel; ++l) { /* camera.cpp:95-99 */
        vw.P[l] = vw.P[l - 1];
        for (int k = 0; k < 8; ++k) vw.P[l][k] /= 2.0f;
    }
    for (int l = 0; l < s.maxLevel; ++l) invert3(vw.P[l].data(), vw.Minv[l].data());
    /* m_oaxis = row(2) / |row(2).head(3)|, camera.cpp:68-69 */
    const float* r2 = &vw.P[0][8];
    float n = norm3({r2[0], r2[1], r2[2]});
    vw.oaxis = {r2[0] / n, r2[1] / n, r2[2] / n, r2[3] / n};
    /* centre = -M^-1 q, camera.cpp:295-308 (double, rounded once) */
    {
        const float* P = vw.P[0].data();
        double a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
        double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
        double det = a * A + b * B + c * C;
        double inv[9] = {A, -(b * i - c * h), b * f - c * e, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * e - b * d};
        double q[3] = {P[3], P[7], P[11]};
        double cx = -(inv[0] * q[0] + inv[1] * q[1] + inv[2] * q[2]) / det;
        double cy = -(inv[3] * q[0] + inv[4] * q[1] + inv[5] * q[2]) / det;
        double cz = -(inv[6] * q[0] + inv[7] * q[1] + inv[8] * q[2]) / det;
        vw.center = {(float)cx, (float)cy, (float)cz, 1.0f};
    }
    /* Optim::setAxesScales, optim.cpp:43-65 */
    vw.zaxis = {vw.oaxis.x, vw.oaxis.y, vw.oaxis.z};
    V3 xa{vw.P[0][0], vw.P[0][1], vw.P[0][2]};
    vw.yaxis = cross3(vw.zaxis, xa);
    vw.yaxis = div3(vw.yaxis, norm3(vw.yaxis));
    vw.xaxis = cross3(vw.yaxis, vw.zaxis);
    V4 x4{vw.xaxis.x, vw.xaxis.y, vw.xaxis.z, 0.0f}, y4{vw.yaxis.x, vw.yaxis.y, vw.yaxis.z, 0.0f};
    V4 row0{vw.P[0][0], vw.P[0][1], vw.P[0][2], vw.P[0][3]}, row1{vw.P[0][4], vw.P[0][5], vw.P[0][6], vw.P[0][7]};
    vw.ipscale = dot4(row0, x4) + dot4(row1, y4);
}

/* ------------------------------------------------------------------ image (image/image.cpp) */
/* Image::buildImagePyramid, image.cpp:245-315, filter == 0.  The 4x4 mask is normalised and the
 * sum divided out once more (D8: border taps are dropped without renormalising). */
void build_image_pyramid(Scene& s, View& vw) {
    float mask[4][4];
    const float base[4] = {1.0f, 3.0f, 3.0f, 1.0f};
    float sum = 0.0f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { mask[i][j] = base[i] * base[j]; sum += mask[i][j]; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) mask[i][j] /= sum; /* exact: /64 */
    float msum = 0.0f; /* mask.sum() after normalisation: exactly 1 */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) msum += mask[i][j];
    for (int level = 1; level < s.maxLevel; ++level) {
        const int w = vw.W[level], h = vw.H[level], pw = vw.W[level - 1], ph = vw.H[level - 1];
        vw.img[level].assign((size_t)w * h * 3, 0);
        const std::vector<uint8_t>& src = vw.img[level - 1];
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            float col[3] = {0.0f, 0.0f, 0.0f};
            for (int i = -1; i < 3; ++i) {
                const int yt = 2 * y + i;
                if (yt < 0 || ph - 1 < yt) continue;
                for (int j = -1; j < 3; ++j) {
                    const int xt = 2 * x + j;
                    if (xt < 0 || pw - 1 < xt) continue;
                    const size_t idx = ((size_t)yt * pw + xt) * 3;
                    for (int c = 0; c < 3; ++c) col[c] += mask[i + 1][j + 1] * (float)src[idx + c];
                }
            }
            const size_t o = ((size_t)y * w + x) * 3;
            for (int c = 0; c < 3; ++c) vw.img[level][o + c] = (uint8_t)((int)floorf(col[c] / msum + 0.5f));
        }
    }
}

/* Image::buildMaskPyramid, image.cpp:717-747.  ys[1]/xs[1] = min(prev_dim, 2y+1) can index one row
 * or column past the end in the reference for odd sizes; clamped to prev_dim-1 here. */
void build_mask_pyramid(Scene& s, View& vw) {
    for (int level = 1; level < s.maxLevel; ++level) {
        const int w = vw.W[level], h = vw.H[level], pw = vw.W[level - 1], ph = vw.H[level - 1];
        vw.mask[level].assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y) {
            const int ys[2] = {2 * y, std::min(ph - 1, 2 * y + 1)};
            for (int x = 0; x < w; ++x) {
                const int xs[2] = {2 * x, std::min(pw - 1, 2 * x + 1)};
                int inside = 0;
                for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i)
                    if (vw.mask[level - 1][(size_t)ys[j] * pw + xs[i]]) ++inside;
                vw.mask[level][(size_t)y * w + x] = inside > 0 ? 255 : 0;
            }
        }
    }
}

/* Image::getColor(float,float,level), bilinear, image.cpp:447-472 */
inline void get_color(const View& vw, float x, float y, int level, float* rgb) {
    const int lx = (int)x, ly = (int)y;
    const int w = vw.W[level];
    const size_t i0 = 3 * ((size_t)ly * w + lx), i1 = i0 + 3 * (size_t)w;
    const float dx1 = x - lx, dx0 = 1.0f - dx1, dy1 = y - ly, dy0 = 1.0f - dy1;
    const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
    const uint8_t* p0 = &vw.img[level][i0];
    const uint8_t* p1 = &vw.img[level][i1];
    for (int c = 0; c < 3; ++c)
        rgb[c] = fma_((float)p1[3 + c], f11, fma_((float)p0[3 + c], f10, fma_((float)p1[c], f01, (float)p0[c] * f00)));
}

/* Image::getMask(float,float,level), image.cpp:749-781 */
inline int get_mask(const View& vw, float fx, float fy, int level) {
    if (vw.mask.empty() || vw.mask[level].empty()) return -1;
    const int ix = (int)floorf(fx + 0.5f), iy = (int)floorf(fy + 0.5f);
    if (ix < 0 || vw.W[level] <= ix || iy < 0 || vw.H[level] <= iy) return -1;
    return vw.mask[level][(size_t)iy * vw.W[level] + ix];
}

/* ------------------------------------------------------------------ reductions */
inline float reduce_seq(const float* a, int n) { /* reference order: optim.cpp:605-607, 921-923, 928-931 */
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += a[i];
    return s;
}
inline float reduce_tree64(const float* a) { /* wave64 butterfly, offsets 1,2,4,8,16,32 */
    float b[64], t[64];
    for (int i = 0; i < 64; ++i) b[i] = a[i];
    for (int off = 1; off < 64; off <<= 1) {
        for (int i = 0; i < 64; ++i) t[i] = b[i] + b[i ^ off];
        for (int i = 0; i < 64; ++i) b[i] = t[i];
    }
    return b[0];
}
/* The engine's sample layout ("class lanes"): lane c of a 16-lane row owns samples c, c + 16, c + 32 of a window (njx slots,
 * the last one partly filled); a window of 16 k + 1 samples (7x7) leaves one "extra" sample, taken by the lane that
 * owns the view's sampling frame and added last.  Every sum over the samples of a window runs in this order in the
 * engine arithmetic: inside a lane slot-ascending, then one DPP row tree (partners 1, 2, 4, 8 apart), then the extra. */
struct ClsLayout { int nj, rem, rx, njx, nl, lim; };
inline ClsLayout cls_layout_of(int wsize) {
    ClsLayout L;
    const int wsz = wsize * wsize;
    L.nj = wsz >> 4; L.rem = wsz & 15;
    L.rx = L.rem == 1 ? 1 : 0;
    L.njx = L.nj + (L.rem > 1 ? 1 : 0);
    L.nl = std::max(L.njx, L.rx ? 1 : 0);
    L.lim = L.rem > 1 ? wsz : 16 * L.nj;
    return L;
}
inline ClsLayout cls_layout(const Scene& s) { return cls_layout_of(s.cfg.wsize); }
inline float reduce_tree16(const float* a) { /* one DPP row: partners 1, 2, 4, 8 apart */
    float b[16], t[16];
    for (int i = 0; i < 16; ++i) b[i] = a[i];
    for (int off = 1; off < 16; off <<= 1) {
        for (int i = 0; i < 16; ++i) t[i] = b[i] + b[i ^ off];
        for (int i = 0; i < 16; ++i) b[i] = t[i];
    }
    return b[0];
}
inline float reduce_cls16(const Scene& s, const float* a) {
    const ClsLayout L = cls_layout(s);
    float lane[16];
    for (int c = 0; c < 16; ++c) {
        float t = 0.0f;
        for (int j = 0; j < L.njx; ++j) { const int q = c + 16 * j; if (q < L.lim) t += a[q]; }
        lane[c] = t;
    }
    float r = reduce_tree16(lane);
    for (int e = 0; e < L.rx; ++e) r += a[16 * L.nj + e];
    return r;
}
inline float reduce(const Scene& s, const float* a64, int n) {
    return s.cfg.sum_mode == ORC_SUM_TREE64 ? reduce_cls16(s, a64) : reduce_seq(a64, n);
}

/* ------------------------------------------------------------------ Optim (pmmvps/optim.cpp) */
/* Optim::getUnit, optim.cpp:34-41.  The reference evaluates 2.0 * fz * 2^level / ipscale in double and rounds to float;
 * 2 * fz * 2^level is exact, so one fp32 division gives the same value except for double-rounding ties. */
inline float get_unit(const Scene& s, int v, const V4& coord) {
    const View& vw = s.views[v];
    const float fz = norm4(sub4(coord, vw.center));
    if (vw.ipscale == 0.0f) return 1.0f;
    return (2.0f * fz * (float)(0x0001 << s.cfg.level)) / vw.ipscale;
}

/* Optim::getPAxes, optim.cpp:67-84 */
void get_paxes(const Scene& s, int v, const V4& coord, const V4& normal, V4& px, V4& py) {
    const View& vw = s.views[v];
    const float pscale = get_unit(s, v, coord);
    V3 n3{normal.x, normal.y, normal.z};
    V3 y3 = cross3(n3, vw.xaxis);
    y3 = nrm3(y3);
    V3 x3 = cross3(y3, n3);
    px = {x3.x * pscale, x3.y * pscale, x3.z * pscale, 0.0f};
    py = {y3.x * pscale, y3.y * pscale, y3.z * pscale, 0.0f};
    const V3 c0 = project(vw, coord, s.cfg.level);
    const float xdis = norm3(sub3(project(vw, add4(coord, px), s.cfg.level), c0));
    const float ydis = norm3(sub3(project(vw, add4(coord, py), s.cfg.level), c0));
    px = scl4(px, 1.0f / xdis);
    py = scl4(py, 1.0f / ydis);
}

inline float robustincc(float incc) { return incc / (1 + 3 * incc); }      /* optim.cpp:622-624 */
inline float unrobustincc(float rincc) { return rincc / (1 - 3 * rincc); } /* optim.cpp:626-628 */

/* levelDiff = floor(log2(ratio) + 0.5) clamped to [-level, 2], optim.cpp:807-809, as threshold
 * compares (identical except within 1 ulp of sqrt(2)*2^k). */
inline int level_diff(const Scene& s, float ratio) {
    /* thresholds 2^(k-0.5), k = -3..2 */
    static const float T[6] = {0.088388347648318f, 0.176776695296637f, 0.353553390593274f,
                               0.707106781186548f, 1.414213562373095f, 2.828427124746190f};
    int ld = -4;
    for (int k = 0; k < 6; ++k) if (ratio >= T[k]) ld = k - 3;
    return std::max(-s.cfg.level, std::min(2, ld));
}
inline float pow2_level(int ld) { /* Optim::myPow2, optim.cpp:785-788 */
    static const float sc[] = {0.0625f, 0.125f, 0.25f, 0.5f, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024};
    return sc[ld + 4];
}

/* Optim::getTexSafe, optim.cpp:895-915 */
inline int get_tex_safe(const Scene& s, int v, int size, const V3& c, const V3& dx, const V3& dy, int level) {
    const int margin = size / 2;
    const float m = (float)margin;
    const float tlx = (c.x - dx.x * m) - dy.x * m, trx = (c.x + dx.x * m) - dy.x * m;
    const float blx = (c.x - dx.x * m) + dy.x * m, brx = (c.x + dx.x * m) + dy.x * m;
    const float tly = (c.y - dx.y * m) - dy.y * m, try_ = (c.y + dx.y * m) - dy.y * m;
    const float bly = (c.y - dx.y * m) + dy.y * m, bry = (c.y + dx.y * m) + dy.y * m;
    const float minx = std::min(tlx, std::min(trx, std::min(blx, brx)));
    const float maxx = std::max(tlx, std::max(trx, std::max(blx, brx)));
    const float miny = std::min(tly, std::min(try_, std::min(bly, bry)));
    const float maxy = std::max(tly, std::max(try_, std::max(bly, bry)));
    const int margin2 = 2;
    const View& vw = s.views[v];
    if (minx < margin2 || vw.W[level] - 1 - margin2 <= maxx || miny < margin2 || vw.H[level] - 1 - margin2 <= maxy) return -1;
    return 0;
}

/* Optim::getTex (single illumination), optim.cpp:790-844 */
int get_tex(const Scene& s, const V4& coord, const V4& px, const V4& py, const V4& pz, int v, Tex& tex, orc_counters* cnt) {
    tex.ok = false;
    const View& vw = s.views[v];
    const int size = s.cfg.wsize;
    V4 ray = sub4(vw.center, coord);
    ray = nrm4(ray);
    const float weight = std::max(0.0f, dot4(ray, pz));
    if (weight < s.cosAngle1) return -1;
    const int margin = size / 2;
    V3 center = project(vw, coord, s.cfg.level);
    V3 dx = sub3(project(vw, add4(coord, px), s.cfg.level), center);
    V3 dy = sub3(project(vw, add4(coord, py), s.cfg.level), center);
    const float ratio = (norm3(dx) + norm3(dy)) / 2.0f;
    const int ld = level_diff(s, ratio);
    const float scale = pow2_level(ld);
    const int newLevel = s.cfg.level + ld;
    const float iscale = 1.0f / scale; /* power of two: exact */
    center = scl3(center, iscale);
    dx = scl3(dx, iscale);
    dy = scl3(dy, iscale);
    if (get_tex_safe(s, v, size, center, dx, dy, newLevel) == -1) return -1;
    const float m = (float)margin;
    const V3 tl{(center.x - dx.x * m) - dy.x * m, (center.y - dx.y * m) - dy.y * m, 0.0f};
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 64; ++i) tex.c[c][i] = 0.0f;
    for (int y = 0; y < size; ++y) for (int x = 0; x < size; ++x) { /* optim.cpp:835-842 */
        const float sx = fma_(dy.x, (float)y, fma_(dx.x, (float)x, tl.x));
        const float sy = fma_(dy.y, (float)y, fma_(dx.y, (float)x, tl.y));
        float rgb[3];
        get_color(vw, sx, sy, newLevel, rgb);
        const int ind = y * size + x;
        tex.c[0][ind] = rgb[0]; tex.c[1][ind] = rgb[1]; tex.c[2][ind] = rgb[2];
    }
    tex.ok = true;
    if (cnt) cnt->view_evals++;
    return 0;
}

/* Optim::normalize, optim.cpp:917-940 */
void normalize_tex(const Scene& s, Tex& tex) {
    const int sz = s.cfg.wsize * s.cfg.wsize;
    float ave[3];
    for (int c = 0; c < 3; ++c) { ave[c] = reduce(s, tex.c[c], sz) * s.inv_sz; tex.ave[c] = ave[c]; }
    float sq[64];
    for (int i = 0; i < 64; ++i) sq[i] = 0.0f;
    for (int i = 0; i < sz; ++i) {
        const float d0 = tex.c[0][i] - ave[0], d1 = tex.c[1][i] - ave[1], d2 = tex.c[2][i] - ave[2];
        tex.c[0][i] = d0; tex.c[1][i] = d1; tex.c[2][i] = d2;
        sq[i] = fma_(d2, d2, fma_(d1, d1, d0 * d0));
    }
    const float ssd = reduce(s, sq, sz);
    float msd = sqrtf(ssd * s.inv_3sz);
    if (msd == 0.0f) msd = 1.0f;
    /* (tex - ave) / msd, optim.cpp:937-939: the division is kept as the factor 1/msd and applied to the dot
     * product of the centred textures (dot_tex) -- the same value up to fp reassociation, and it lets the
     * engine reduce sum(d0*d1) without waiting for the square root. */
    tex.inv = 1.0f / msd;
}

/* Optim::dot, optim.cpp:601-609 */
float dot_tex(const Scene& s, const Tex& a, const Tex& b) {
    const int sz = s.cfg.wsize * s.cfg.wsize;
    float p[64];
    for (int i = 0; i < 64; ++i) p[i] = 0.0f;
    for (int i = 0; i < sz; ++i) p[i] = fma_(a.c[2][i], b.c[2][i], fma_(a.c[1][i], b.c[1][i], a.c[0][i] * b.c[0][i]));
    return (reduce(s, p, sz) * (a.inv * b.inv)) * s.inv_3sz;
}

/* Optim::computeUnits(patch, units), optim.cpp:109-132, then computeWeights, optim.cpp:942-948 */
void compute_weights(const Scene& s, const V4& coord, const V4& normal, const int* img, int n, float* w) {
    for (int i = 0; i < n; ++i) {
        float unit = get_unit(s, img[i], coord);
        V4 ray = sub4(s.views[img[i]].center, coord);
        ray = nrm4(ray);
        const float d = dot4(ray, normal);
        if (0.0f < d) unit /= d; else unit = (float)(INT_MAX / 2);
        w[i] = unit;
    }
    for (int i = 1; i < n; ++i) w[i] = std::min(1.0f, w[0] / w[i]);
    if (n > 0) w[0] = 1.0f;
}

/* Optim::computeINCC, optim.cpp:630-706 (non-PAIRNCC branch) */
float compute_incc(const Scene& s, const V4& coord, const V4& normal, const int* img, int n, const float* weights,
                   int robust, orc_counters* cnt) {
    if (n < 2) return 2.0f;
    V4 px, py;
    get_paxes(s, img[0], coord, normal, px, py);
    const int sz = std::min(s.tau, n);
    if (cnt) cnt->evals++;
    Tex t0, ti;
    if (get_tex(s, coord, px, py, normal, img[0], t0, cnt) == 0) normalize_tex(s, t0);
    if (!t0.ok) return 2.0f;
    float score = 0.0f, total = 0.0f;
    for (int i = 1; i < sz; ++i) {
        if (get_tex(s, coord, px, py, normal, img[i], ti, cnt) == 0) normalize_tex(s, ti);
        if (!ti.ok) continue;
        total += weights[i];
        const float incc = (float)(1.0 - dot_tex(s, t0, ti));
        score += (robust ? robustincc(incc) : incc) * weights[i];
    }
    if (total == 0.0f) return 2.0f;
    return score / total;
}

/* PatchManager::computeNcc, patch_manager.cpp:401-404 */
float compute_ncc(const Scene& s, const Patch& p, orc_counters* cnt) {
    float w[MAXI];
    compute_weights(s, p.coord, p.normal, p.img, p.nimg, w);
    return 1.0f - unrobustincc(compute_incc(s, p.coord, p.normal, p.img, p.nimg, w, 1, cnt));
}

/* Optim::setINCCs (vector), optim.cpp:708-746: reference view against ALL listed views (D9) */
void set_inccs(const Scene& s, const Patch& p, const int* idx, int n, int robust, float* inccs, orc_counters* cnt, uint64_t* okmask = nullptr) {
    V4 px, py;
    get_paxes(s, idx[0], p.coord, p.normal, px, py);
    if (cnt) cnt->evals++;
    Tex t0, ti;
    if (okmask) *okmask = 0u;
    if (get_tex(s, p.coord, px, py, p.normal, idx[0], t0, cnt) == 0) normalize_tex(s, t0);
    if (!t0.ok) { for (int i = 0; i < n; ++i) inccs[i] = 2.0f; return; }
    inccs[0] = 0.0f;
    if (okmask) *okmask = 1u;
    for (int i = 1; i < n; ++i) {
        if (get_tex(s, p.coord, px, py, p.normal, idx[i], ti, cnt) == 0) normalize_tex(s, ti);
        if (!ti.ok) { inccs[i] = 2.0f; continue; }
        if (okmask) *okmask |= (uint64_t)1 << i;  /* one bit per view of the list: up to 64 in the wide build */
        const float d = 1.0f - dot_tex(s, t0, ti);
        inccs[i] = robust ? robustincc(d) : d;
    }
}

/* Optim::setINCCs (matrix), optim.cpp:748-783 */
void set_inccs_matrix(const Scene& s, const Patch& p, const int* idx, int n, int robust, float* inccs /* n*n */, orc_counters* cnt) {
    V4 px, py;
    get_paxes(s, idx[0], p.coord, p.normal, px, py);
    if (cnt) cnt->evals++;
    std::vector<Tex> t(n);
    for (int i = 0; i < n; ++i)
        if (get_tex(s, p.coord, px, py, p.normal, idx[i], t[i], cnt) == 0) normalize_tex(s, t[i]);
    for (int i = 0; i < n; ++i) {
        inccs[i * n + i] = 0.0f;
        for (int j = i + 1; j < n; ++j) {
            float val = 2.0f;
            if (t[i].ok && t[j].ok) {
                /* the V x V pair products run over the samples in the reference's sequential order (optim.cpp:605-607) in both
                 * modes.  SEQ: the dot product of a sample's three channels, then added (Eigen's Vector3f::dot).  TREE64 (engine
                 * arithmetic): ONE chain acc = fma(a[k], b[k], acc) over k = 3 sample + channel -- what a f32 MFMA computes (the
                 * 32- and 64-view engine builds take the Gram matrix of the textures from v_mfma_f32_32x32x2_f32, which is bit
                 * for bit a k-ordered fmaf chain; the 16-view build runs the same chain with one lane per pair) */
                const int sz = s.cfg.wsize * s.cfg.wsize;
                float acc = 0.0f;
                if (s.cfg.sum_mode == ORC_SUM_TREE64) {
                    for (int q = 0; q < sz; ++q) {
                        acc = fma_(t[i].c[0][q], t[j].c[0][q], acc); acc = fma_(t[i].c[1][q], t[j].c[1][q], acc); acc = fma_(t[i].c[2][q], t[j].c[2][q], acc);
                    }
                } else
                for (int q = 0; q < sz; ++q)
                    acc += fma_(t[i].c[2][q], t[j].c[2][q], fma_(t[i].c[1][q], t[j].c[1][q], t[i].c[0][q] * t[j].c[0][q]));
                const float d = 1.0f - (acc * (t[i].inv * t[j].inv)) * s.inv_3sz;
                val = robust ? robustincc(d) : d;
            }
            inccs[i * n + j] = inccs[j * n + i] = val;
        }
    }
}

/* PatchManager::setGrids cell rule, patch_manager.cpp:241-250 */
inline void cell_of(const Scene& s, int v, const V4& coord, int& ix, int& iy) {
    const V3 ic = project(s.views[v], coord, s.cfg.level);
    ix = ((int)floorf(ic.x + 0.5f)) / s.cfg.csize;
    iy = ((int)floorf(ic.y + 0.5f)) / s.cfg.csize;
}
void set_grids(const Scene& s, Patch& p) {
    for (int i = 0; i < p.nimg; ++i) cell_of(s, p.img[i], p.coord, p.gx[i], p.gy[i]);
}
void set_vgrids(const Scene& s, Patch& p) { /* patch_manager.cpp:252-261 */
    for (int i = 0; i < p.nvimg; ++i) cell_of(s, p.vimg[i], p.coord, p.vgx[i], p.vgy[i]);
}
/* PatchManager::setGridsImages, patch_manager.cpp:223-239 */
void set_grids_images(const Scene& s, Patch& p, const int* images, int n) {
    p.nimg = 0;
    for (int i = 0; i < n; ++i) {
        int ix, iy;
        cell_of(s, images[i], p.coord, ix, iy);
        const View& vw = s.views[images[i]];
        if (0 <= ix && ix < vw.gw && 0 <= iy && iy < vw.gh) {
            p.img[p.nimg] = images[i]; p.gx[p.nimg] = ix; p.gy[p.nimg] = iy; ++p.nimg;
        }
    }
}

/* Optim::addImages, optim.cpp:165-205 (m_visdata2[ref] = every other view ascending, option.cpp:151-170) */
void add_images(const Scene& s, Patch& p) {
    bool visib[256] = {false};
    for (int i = 0; i < p.nimg; ++i) visib[p.img[i]] = true;
    const int ref = p.img[0];
    for (int v = 0; v < s.cfg.nviews; ++v) {
        if (v == ref || visib[v]) continue;
        const View& vw = s.views[v];
        const V3 ic = project(vw, p.coord, s.cfg.level);
        if (ic.x < 0.0f || vw.W[s.cfg.level] - 1 <= ic.x || ic.y < 0.0f || vw.H[s.cfg.level] - 1 <= ic.y) continue;
        V4 ray = sub4(vw.center, p.coord);
        ray = nrm4(ray);
        if (s.cosAngle0 <= dot4(ray, p.normal)) {
            if (p.nimg < LISTCAP) p.img[p.nimg++] = v;
            else s.list_truncations.fetch_add(1, std::memory_order_relaxed);
        }
    }
}

/* Optim::constraintImages, optim.cpp:207-219 */
/* w_keep / n_keep (engine schedule, first constraintImages of postProcess): refinePatch left the final m_ncc to this
 * evaluation -- computeINCC at the refined patch samples the first min(tau, n_keep) of these very textures -- so it is
 * taken here as the tail of computeINCC (optim.cpp:690-705) over the robust INCCs of those views */
void constraint_images(const Scene& s, Patch& p, float nccThreshold, orc_counters* cnt, const float* w_keep = nullptr, int n_keep = 0) {
    float inccs[MAXI];
    uint64_t ok = 0;
    set_inccs(s, p, p.img, p.nimg, 0, inccs, cnt, &ok);
    if (w_keep) {
        float incc = 2.0f;
        if (n_keep >= 2 && (ok & 1u)) {
            const int sz = std::min(s.tau, n_keep);
            float score = 0.0f, total = 0.0f;
            for (int i = 1; i < sz; ++i) {
                if (!((ok >> i) & 1u)) continue;
                total += w_keep[i];
                score += robustincc(inccs[i]) * w_keep[i];
            }
            if (total != 0.0f) incc = score / total;
        }
        p.ncc = 1.0f - unrobustincc(incc);
    }
    int n = 1;
    for (int i = 1; i < p.nimg; ++i) if (inccs[i] < 1.0f - nccThreshold) p.img[n++] = p.img[i];
    p.nimg = n;
}

/* Optim::sortImages (isFixed = 1), optim.cpp:221-258 */
void sort_images(const Scene& s, Patch& p) {
    const float threshold = s.sortThreshold;
    int idx0[MAXI], n0 = 0;
    float units0[MAXI];
    V4 rays0[MAXI];
    for (int i = 0; i < p.nimg; ++i) { /* computeUnits(patch, indexes, units, rays), optim.cpp:86-107 */
        V4 ray = sub4(s.views[p.img[i]].center, p.coord);
        ray = nrm4(ray);
        const float d = dot4(ray, p.normal);
        if (d <= 0.0f) continue;
        idx0[n0] = p.img[i]; units0[n0] = get_unit(s, p.img[i], p.coord) / d; rays0[n0] = ray; ++n0;
    }
    p.nimg = 0;
    if (n0 < 2) return;
    units0[0] = 0.0f;
    while (n0 > 0) {
        int index = 0;
        for (int i = 1; i < n0; ++i) if (units0[i] < units0[index]) index = i; /* min_element: first minimum */
        p.img[p.nimg++] = idx0[index];
        int n1 = 0;
        const V4 rsel = rays0[index];
        for (int i = 0; i < n0; ++i) {
            if (i == index) continue;
            const float ftmp = std::min(threshold, std::max(threshold / 2.0f, 1.0f - dot4(rsel, rays0[i])));
            idx0[n1] = idx0[i]; rays0[n1] = rays0[i]; units0[n1] = units0[i] * threshold / ftmp; ++n1;
        }
        n0 = n1;
    }
}

/* PatchManager::setScales, patch_manager.cpp:378-399 */
void set_scales(const Scene& s, Patch& p) {
    const float unit = get_unit(s, p.img[0], p.coord);
    const float unit2 = 2.0f * unit;
    V4 ray = sub4(p.coord, s.views[p.img[0]].center);
    ray = nrm4(ray);
    const int num = std::min(s.tau, p.nimg);
    for (int i = 1; i < num; ++i) {
        const View& vw = s.views[p.img[i]];
        const V3 d = sub3(project(vw, p.coord, s.cfg.level), project(vw, sub4(p.coord, mul4(ray, unit2)), s.cfg.level));
        p.dscale += norm3(d);
    }
    p.dscale /= num - 1;
    p.dscale = unit2 / p.dscale;
    p.ascale = pm_atanf(p.dscale / (unit * s.cfg.wsize / 2.0f));
}

/* PhotoSet::checkAngles, photoSet.cpp:77-103; minAngle < acos(dot) < maxAngle as cosine compares */
int check_angles(const Scene& s, const V4& coord, const int* idx, int n) {
    V4 rays[MAXI];
    for (int i = 0; i < n; ++i) { rays[i] = sub4(s.views[idx[i]].center, coord); rays[i] = nrm4(rays[i]); }
    int count = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) {
        const float d = std::max(-1.0f, std::min(1.0f, dot4(rays[i], rays[j])));
        if (d < s.cosMinAngle && s.cosMaxAngle < d) ++count;
    }
    return count < 1 ? -1 : 0;
}

/* Optim::preProcess, optim.cpp:137-163 */
int pre_process(const Scene& s, Patch& p, orc_counters* cnt) {
    add_images(s, p);
    constraint_images(s, p, s.nccThresholdBefore, cnt);
    sort_images(s, p);
    if (p.nimg > 0) set_scales(s, p);
    if (p.nimg < s.cfg.minImageNum) return -1;
    if (check_angles(s, p.coord, p.img, p.nimg) == -1) { p.nimg = 0; return -1; }
    return 0;
}

/* Optim::filterImagesByAngle, optim.cpp:325-346 */
void filter_images_by_angle(const Scene& s, Patch& p) {
    int n = 0;
    for (int i = 0; i < p.nimg; ++i) {
        V4 ray = sub4(s.views[p.img[i]].center, p.coord);
        ray = nrm4(ray);
        if (dot4(ray, p.normal) < s.cosAngle1) {
            if (i == 0) { p.nimg = 0; return; }
        } else p.img[n++] = p.img[i];
    }
    p.nimg = n;
}

/* Optim::setRefImage, optim.cpp:348-383 */
void set_ref_image(const Scene& s, Patch& p, orc_counters* cnt) {
    if (p.nimg == 0) return;
    const int n = p.nimg;
    std::vector<float> inccs((size_t)n * n);
    set_inccs_matrix(s, p, p.img, n, 1, inccs.data(), cnt);
    int refindex = -1;
    float refncc = (float)(INT_MAX / 2);
    for (int i = 0; i < n; ++i) {
        float sum = 0.0f;
        for (int j = 0; j < n; ++j) sum += inccs[i * n + j];
        if (sum < refncc) { refncc = sum; refindex = i; }
    }
    if (refindex < 0) return; /* all NaN; the reference would index [-1] */
    std::swap(p.img[0], p.img[refindex]);
}

/* ---- encode / decode, optim.cpp:549-599 (x kept in float: products of two floats round once) */
struct RefineCtx {
    V4 center, ray;
    float dscale, ascale;
    int ref;
};
void encode(const Scene& s, const RefineCtx& rc, const V4& coord, const V4& normal, float* x) {
    x[0] = dot4(sub4(coord, rc.center), rc.ray) / rc.dscale;
    const View& vw = s.views[rc.ref];
    const V3 n3{normal.x, normal.y, normal.z};
    const float fx = dot3(vw.xaxis, n3), fy = dot3(vw.yaxis, n3), fz = dot3(vw.zaxis, n3);
    float a2 = pm_asinf(std::max(-1.0f, std::min(1.0f, fy)));
    const float cosb = pm_cosf(a2);
    float a1;
    if (cosb == 0.0f) a1 = 0.0f;
    else {
        const float sina = fx / cosb, cosa = -fz / cosb;
        a1 = pm_acosf(std::max(-1.0f, std::min(1.0f, cosa)));
        if (sina < 0.0f) a1 = -a1;
    }
    x[1] = a1 / rc.ascale;
    x[2] = a2 / rc.ascale;
}
void decode(const Scene& s, const RefineCtx& rc, const float* x, V4& coord, V4& normal) {
    const float t = rc.dscale * x[0];
    coord = {fma_(t, rc.ray.x, rc.center.x), fma_(t, rc.ray.y, rc.center.y), fma_(t, rc.ray.z, rc.center.z), fma_(t, rc.ray.w, rc.center.w)};
    const float angle1 = x[1] * rc.ascale, angle2 = x[2] * rc.ascale;
    const float s1 = pm_sinf(angle1), c1 = pm_cosf(angle1), s2 = pm_sinf(angle2), c2 = pm_cosf(angle2);
    const float fx = s1 * c2, fy = s2, fz = -c1 * c2;
    const View& vw = s.views[rc.ref];
    normal = {fma_(vw.zaxis.x, fz, fma_(vw.yaxis.x, fy, vw.xaxis.x * fx)),
              fma_(vw.zaxis.y, fz, fma_(vw.yaxis.y, fy, vw.xaxis.y * fx)),
              fma_(vw.zaxis.z, fz, fma_(vw.yaxis.z, fy, vw.xaxis.z * fx)), 0.0f};
}

/* Optim::cost_func, optim.cpp:401-468 (pairwise == 0 branch) */
/* piv != nullptr (engine arithmetic, the first evaluation of refinePatch): piv[i] receives the mean colour of view i at this
 * point (128 for a view that was not sampled) -- the pivots of the class-lane evaluations of the refinement steps.  The
 * engine samples every view of the list even when the reference view fails, so the means are taken the same way here. */
double cost_func(const Scene& s, const RefineCtx& rc, const int* idx, int n, const float* x, orc_counters* cnt, float (*piv)[3] = nullptr) {
    V4 coord, normal, px, py;
    decode(s, rc, x, coord, normal);
    get_paxes(s, idx[0], coord, normal, px, py);
    const int sz = std::min(s.tau, n);
    const int minimum = std::min(s.cfg.minImageNum, sz);
    if (cnt) cnt->evals++;
    Tex t0, ti;
    if (get_tex(s, coord, px, py, normal, idx[0], t0, cnt) == 0) normalize_tex(s, t0);
    if (piv) {
        for (int c = 0; c < 3; ++c) piv[0][c] = t0.ok ? t0.ave[c] : 128.0f;
        for (int i = 1; i < sz; ++i) {
            Tex tp;
            const bool okp = get_tex(s, coord, px, py, normal, idx[i], tp, nullptr) == 0;
            if (okp) normalize_tex(s, tp);
            for (int c = 0; c < 3; ++c) piv[i][c] = okp ? tp.ave[c] : 128.0f;
        }
    }
    if (!t0.ok) return 2.0;
    double ans = 0.0;
    int denom = 0;
    for (int i = 1; i < sz; ++i) {
        if (get_tex(s, coord, px, py, normal, idx[i], ti, cnt) == 0) normalize_tex(s, ti);
        if (!ti.ok) continue;
        ans += robustincc((float)(1.0 - dot_tex(s, t0, ti)));
        denom++;
    }
    if (denom < minimum - 1) return 2.0;
    return ans / denom;
}

/* ---- class-lane arithmetic of the engine's refinement steps (mvs_device.cuh, eval_steps3) --------------------------------
 * The 7x7 samples of one (proposal, view) are dealt over 16 lanes: lane c walks samples c, c + 16, c + 32 (as far as they
 * exist) and the few samples beyond the last full 16 ("extras": sample 48 of a 7x7 window) sit in a lane of their own,
 * whose sums are added to lane e.  Colours are taken relative to a pivot p per view, c' = colour - p, and normalize /
 * dot (optim.cpp:917-940, 601-609) are computed from running sums instead of a centred copy of the texture:
 *     S1 = sum c',  S2 = sum |c'|^2,  S01 = sum c' . c0'   (c0' = the reference view's colour of the same sample)
 *     mean m = S1 / n,  ssd = max(S2 - S1 . m, 0),  dot = S01 - S1 . m0,  INCC = 1 - dot (inv0 inv) / 3n
 * -- the same quantities (sum (c - mean)^2 = sum c^2 - n mean^2), the pivot keeping the squares small enough for fp32. */
struct ClsSums { float s1[3], s2, s01; };
/* tex: raw colours of one view; piv: its pivot; c0p: the reference view's c' per sample (nullptr for the reference view
 * itself); cp_out: this view's c' per sample */
void tex_stats_class16(const Scene& s, const Tex& tex, const float* piv, const float (*c0p)[64], float (*cp_out)[64], ClsSums& out) {
    const ClsLayout L = cls_layout(s);
    const int wsz = s.cfg.wsize * s.cfg.wsize;
    float cp[3][64];
    for (int i = 0; i < wsz; ++i) for (int c = 0; c < 3; ++c) cp[c][i] = fma_(-piv[c], 1.0f, tex.c[c][i]);
    float l1[3][16], l2[16], l01[16];
    const int lim = L.lim;
    for (int c = 0; c < 16; ++c) {
        float s1r = 0.0f, s1g = 0.0f, s1b = 0.0f, s2 = 0.0f, s01 = 0.0f;
        for (int j = 0; j < L.njx; ++j) {
            const int q = c + 16 * j;
            if (q >= lim) continue;
            const float r = cp[0][q], g = cp[1][q], b = cp[2][q];
            s1r += r; s1g += g; s1b += b;
            s2 = fma_(r, r, s2); s2 = fma_(g, g, s2); s2 = fma_(b, b, s2);
            if (c0p) { s01 = fma_(r, c0p[0][q], s01); s01 = fma_(g, c0p[1][q], s01); s01 = fma_(b, c0p[2][q], s01); }
        }
        l1[0][c] = s1r; l1[1][c] = s1g; l1[2][c] = s1b; l2[c] = s2; l01[c] = s01;
    }
    for (int c = 0; c < 3; ++c) out.s1[c] = reduce_tree16(l1[c]);
    out.s2 = reduce_tree16(l2);
    out.s01 = reduce_tree16(l01);
    for (int e = 0; e < L.rx; ++e) { /* the extras: added to the finished row sums, in sample order (the engine's frame lanes) */
        const int q = 16 * L.nj + e;
        const float r = cp[0][q], g = cp[1][q], b = cp[2][q];
        out.s1[0] += r; out.s1[1] += g; out.s1[2] += b;
        out.s2 = fma_(r, r, out.s2); out.s2 = fma_(g, g, out.s2); out.s2 = fma_(b, b, out.s2);
        if (c0p) { out.s01 = fma_(r, c0p[0][q], out.s01); out.s01 = fma_(g, c0p[1][q], out.s01); out.s01 = fma_(b, c0p[2][q], out.s01); }
    }
    if (cp_out) for (int c = 0; c < 3; ++c) for (int i = 0; i < 64; ++i) cp_out[c][i] = i < wsz ? cp[c][i] : 0.0f;
}
inline float cls_inv_msd(const Scene& s, const ClsSums& q, float* mean3) {
    for (int c = 0; c < 3; ++c) mean3[c] = q.s1[c] * s.inv_sz;
    const float ssd = std::max(q.s2 - fma_(q.s1[2], mean3[2], fma_(q.s1[1], mean3[1], q.s1[0] * mean3[0])), 0.0f);
    float msd = sqrtf(ssd * s.inv_3sz);
    if (msd == 0.0f) msd = 1.0f;
    return 1.0f / msd;
}
/* Optim::cost_func (optim.cpp:401-468) in the class-lane arithmetic */
double cost_func_cls(const Scene& s, const RefineCtx& rc, const int* idx, int n, const float* x, const float (*piv)[3], orc_counters* cnt) {
    V4 coord, normal, px, py;
    decode(s, rc, x, coord, normal);
    get_paxes(s, idx[0], coord, normal, px, py);
    const int sz = std::min(s.tau, n);
    const int minimum = std::min(s.cfg.minImageNum, sz);
    if (cnt) cnt->evals++;
    Tex t0, ti;
    if (get_tex(s, coord, px, py, normal, idx[0], t0, cnt) != 0) return 2.0;
    float c0p[3][64];
    ClsSums q0, qi;
    tex_stats_class16(s, t0, piv[0], nullptr, c0p, q0);
    float m0[3], mi[3];
    const float inv0 = cls_inv_msd(s, q0, m0);
    double ans = 0.0;
    int denom = 0;
    for (int i = 1; i < sz; ++i) {
        if (get_tex(s, coord, px, py, normal, idx[i], ti, cnt) != 0) continue;
        tex_stats_class16(s, ti, piv[i], c0p, nullptr, qi);
        const float inv = cls_inv_msd(s, qi, mi);
        const float dot = qi.s01 - fma_(qi.s1[2], m0[2], fma_(qi.s1[1], m0[1], qi.s1[0] * m0[0]));
        const float incc = 1.0f - (dot * (inv0 * inv)) * s.inv_3sz;
        ans += robustincc(incc);
        denom++;
    }
    if (denom < minimum - 1) return 2.0;
    return ans / denom;
}

/* Optim::refinePatch, optim.cpp:480-547.  The NLopt LN_BOBYQA call (511-524) is replaced by a
 * halving random search: K steps, per step 4 proposals around the step's start point
 * (depth only / angles only / both / both mirrored), the step's best is kept if it improves; ranges halve.
 * Same variables, same bounds (angles +-23.99999 units of pi/48, depth unbounded). */
/* w_keep != nullptr (engine schedule inside propagatePatch): the weights go out and the final m_ncc is left to the first
 * constraintImages of postProcess, which samples the same textures anyway */
int refine_patch(const Scene& s, Patch& p, const uint32_t key[4], orc_counters* cnt, float* w_keep = nullptr) {
    RefineCtx rc;
    rc.center = p.coord;
    rc.ref = p.img[0];
    rc.ray = sub4(p.coord, s.views[rc.ref].center);
    rc.ray = nrm4(rc.ray);
    rc.dscale = p.dscale;
    rc.ascale = s.ascaleConst;
    float w[MAXI];
    compute_weights(s, p.coord, p.normal, p.img, p.nimg, w);
    float x[3];
    encode(s, rc, p.coord, p.normal, x);
    const float amin = -23.99999f, amax = 23.99999f;
    x[1] = std::max(std::min(x[1], amax), amin);
    x[2] = std::max(std::min(x[2], amax), amin);
    /* engine arithmetic: the three proposals of a step are evaluated in the class-lane layout, relative to the view means
     * of this first evaluation; the reference's summation order (ORC_SUM_SEQ) keeps normalize / dot as they are */
    const ClsLayout L = cls_layout(s);
    const bool cls = s.cfg.sum_mode == ORC_SUM_TREE64 && L.nl >= 1 && L.njx <= 3;
    float piv[MAXI][3];
    for (int i = 0; i < MAXI; ++i) piv[i][0] = piv[i][1] = piv[i][2] = 128.0f;
    double fbest = cost_func(s, rc, p.img, p.nimg, x, cnt, cls ? piv : nullptr);
    float rd = s.cfg.refine_rd0, ra = s.cfg.refine_ra0;
    for (int k = 0; k < s.cfg.refine_steps; ++k) {
        float cand[4][3];
        double f[4];
        for (int j = 0; j < 4; ++j) {
            if (j < 3) {
                const uint32_t draw = 16u + (uint32_t)(k * 3 + j) * 3u;
                const float u0 = 2.0f * rng_uniform(s.cfg.seed, key[0], key[1], key[2], key[3], draw + 0);
                const float u1 = 2.0f * rng_uniform(s.cfg.seed, key[0], key[1], key[2], key[3], draw + 1);
                const float u2 = 2.0f * rng_uniform(s.cfg.seed, key[0], key[1], key[2], key[3], draw + 2);
                cand[j][0] = (j == 1) ? x[0] : fma_(u0, rd, x[0]);
                cand[j][1] = (j == 0) ? x[1] : std::max(std::min(fma_(u1, ra, x[1]), amax), amin);
                cand[j][2] = (j == 0) ? x[2] : std::max(std::min(fma_(u2, ra, x[2]), amax), amin);
            } else { /* the "both" proposal mirrored about the step's start */
                cand[3][0] = x[0] - (cand[2][0] - x[0]);
                cand[3][1] = std::max(std::min(x[1] - (cand[2][1] - x[1]), amax), amin);
                cand[3][2] = std::max(std::min(x[2] - (cand[2][2] - x[2]), amax), amin);
            }
            f[j] = cls ? cost_func_cls(s, rc, p.img, p.nimg, cand[j], piv, cnt) : cost_func(s, rc, p.img, p.nimg, cand[j], cnt);
        }
        int jb = 0;
        for (int j = 1; j < 4; ++j) if (f[j] < f[jb]) jb = j;
        if (f[jb] < fbest) { fbest = f[jb]; x[0] = cand[jb][0]; x[1] = cand[jb][1]; x[2] = cand[jb][2]; }
        rd *= 0.5f; ra *= 0.5f;
    }
    decode(s, rc, x, p.coord, p.normal); /* optim.cpp:535-539 */
    p.normal.w = 0.0f;
    if (w_keep) { for (int i = 0; i < MAXI; ++i) w_keep[i] = i < p.nimg ? w[i] : 0.0f; }
    else p.ncc = 1.0f - unrobustincc(compute_incc(s, p.coord, p.normal, p.img, p.nimg, w, 1, cnt));
    return 0;
}

/* ------------------------------------------------------------------ list access shared by both schedules */
struct Span { const int* p; int n; };
inline const Patch& get_patch(const Scene& s, int id, const DestCtx* ctx) {
    if (id >= NEWBASE) return ctx->staged[id - NEWBASE];
    return s.pool[id];
}
inline Span cell_list(const Scene& s, int kind, int v, int cell, const DestCtx* ctx) {
    if (s.cfg.schedule == ORC_SCHEDULE_FAITHFUL) {
        const std::vector<int>& l = kind == 0 ? s.pgrids[v][cell] : s.vpgrids[v][cell];
        return {l.data(), (int)l.size()};
    }
    if (kind == 0 && ctx && ctx->v == v && ctx->cell == cell) return {ctx->list.data(), (int)ctx->list.size()};
    const std::vector<int>& st = kind == 0 ? s.csr_start[v] : s.vcsr_start[v];
    const std::vector<int>& ids = kind == 0 ? s.csr_ids[v] : s.vcsr_ids[v];
    if (st.empty()) return {nullptr, 0};
    return {ids.data() + st[cell], st[cell + 1] - st[cell]};
}

/* PatchManager::isVisible, patch_manager.cpp:335-376 */
int is_visible(const Scene& s, const Patch& p, int image, int ix, int iy, float strict, const DestCtx* ctx) {
    const View& vw = s.views[image];
    if (ix < 0 || vw.gw <= ix || iy < 0 || vw.gh <= iy) return 0;
    if (s.depth == 0) return 1;
    const int dp = s.dpgrids[image][iy * vw.gw + ix];
    if (dp < 0) return 1;
    const Patch& dpp = get_patch(s, dp, ctx);
    V4 ray = sub4(p.coord, vw.center);
    ray = nrm4(ray);
    const float diff = dot4(ray, sub4(p.coord, dpp.coord));
    /* patch_manager.cpp:366: `const float factor = std::min(2.0, 2.0 + ray.dot(normal))` -- the double minimum is narrowed to float, and
     * the product and the comparison of :369 run in float (2 + dot is exact in double, so the narrowing equals the float sum) */
    const float factor = (float)std::min(2.0, 2.0 + (double)dot4(ray, p.normal));
    return diff < get_unit(s, image, p.coord) * (float)s.cfg.csize * strict * factor ? 1 : 0;
}

/* PatchManager::setVImagesVGrids, patch_manager.cpp:267-301 */
void set_vimages_vgrids(const Scene& s, Patch& p, const DestCtx* ctx) {
    bool visib[256] = {false};
    for (int i = 0; i < p.nimg; ++i) visib[p.img[i]] = true;
    for (int i = 0; i < p.nvimg; ++i) visib[p.vimg[i]] = true;
    for (int image = 0; image < s.cfg.nviews; ++image) {
        if (visib[image]) continue;
        int ix, iy;
        cell_of(s, image, p.coord, ix, iy);
        if (is_visible(s, p, image, ix, iy, s.neighborThreshold, ctx) == 0) continue;
        if (p.nvimg < LISTCAP) { p.vimg[p.nvimg] = image; p.vgx[p.nvimg] = ix; p.vgy[p.nvimg] = iy; ++p.nvimg; }
        else s.list_truncations.fetch_add(1, std::memory_order_relaxed);
    }
}

inline float score2(const Patch& p, float thr) { return std::max(0.0f, p.ncc - thr) * p.nimg; } /* patch.cpp:27-29 */

/* PmMvps::isNeighbor, pmmvps.cpp:117-147 (deg/rad typo at :124 kept, D6) */
int is_neighbor(const Scene& s, const Patch& lhs, const Patch& rhs, float hunit, float thr) {
    if (dot4(lhs.normal, rhs.normal) < s.cosNeighborTypo) return 0;
    const V4 diff = sub4(lhs.coord, rhs.coord);
    const float vunit = lhs.dscale + rhs.dscale;
    const float f0 = dot4(lhs.normal, diff), f1 = dot4(rhs.normal, diff);
    float ftmp = (fabsf(f0) + fabsf(f1)) / 2.0f;
    ftmp /= vunit;
    const V4 h = add4(sub4(diff, mul4(lhs.normal, f0)), sub4(diff, mul4(rhs.normal, f1)));
    const float hsize = norm4(h) / 2.0f / hunit;
    if (1.0f < hsize) ftmp /= std::min(2.0f, hsize);
    return ftmp < thr ? 1 : 0;
}
int is_neighbor(const Scene& s, const Patch& lhs, const Patch& rhs, float thr) {
    const float hunit = (get_unit(s, lhs.img[0], lhs.coord) + get_unit(s, rhs.img[0], rhs.coord)) / 2.0f * s.cfg.csize;
    return is_neighbor(s, lhs, rhs, hunit, thr);
}
/* PmMvps::isNeighborRadius, pmmvps.cpp:149-180 */
int is_neighbor_radius(const Scene& s, const Patch& lhs, const Patch& rhs, float hunit, float thr, float radius) {
    if (dot4(lhs.normal, rhs.normal) < s.cosNeighbor120) return 0;
    const V4 diff = sub4(rhs.coord, lhs.coord);
    const float vunit = lhs.dscale + rhs.dscale;
    const float f0 = dot4(lhs.normal, diff), f1 = dot4(rhs.normal, diff);
    float ftmp = (fabsf(f0) + fabsf(f1)) / 2.0f;
    ftmp /= vunit;
    const V4 h = sub4(sub4(mul4(diff, 2.0f), mul4(lhs.normal, f0)), mul4(rhs.normal, f1));
    const float hsize = norm4(h) / 2.0f / hunit;
    if (radius / hunit < hsize) return 0;
    if (1.0f < hsize) ftmp /= std::min(2.0f, hsize);
    return ftmp < thr ? 1 : 0;
}

/* Filter::computeGain, filter.cpp:108-146 */
float compute_gain(const Scene& s, const Patch& p, const DestCtx* ctx) {
    float gain = score2(p, s.nccThreshold);
    for (int i = 0; i < p.nimg; ++i) {
        const int v = p.img[i];
        const int cell = p.gy[i] * s.views[v].gw + p.gx[i];
        float maxpressure = 0.0f;
        const Span l = cell_list(s, 0, v, cell, ctx);
        for (int j = 0; j < l.n; ++j) {
            const Patch& q = get_patch(s, l.p[j], ctx);
            if (!is_neighbor(s, p, q, s.neighborThreshold1)) maxpressure = std::max(maxpressure, q.ncc - s.nccThreshold);
        }
        gain -= maxpressure;
    }
    for (int i = 0; i < p.nvimg; ++i) {
        const int v = p.vimg[i];
        const float pdepth = dot4(s.views[v].oaxis, p.coord); /* Camera::computeDepth, camera.cpp:339-346 */
        const int cell = p.vgy[i] * s.views[v].gw + p.vgx[i];
        float maxpressure = 0.0f;
        const Span l = cell_list(s, 0, v, cell, ctx);
        for (int j = 0; j < l.n; ++j) {
            const Patch& q = get_patch(s, l.p[j], ctx);
            const float bdepth = dot4(s.views[v].oaxis, q.coord);
            if (pdepth < bdepth && !is_neighbor(s, p, q, s.neighborThreshold1)) maxpressure = std::max(maxpressure, q.ncc - s.nccThreshold);
        }
        gain -= maxpressure;
    }
    return gain;
}

/* Propagate::computeRadius, propagate.cpp:474-481 */
float compute_radius(const Scene& s, const Patch& p) {
    float units[MAXI];
    for (int i = 0; i < p.nimg; ++i) {
        float unit = get_unit(s, p.img[i], p.coord);
        V4 ray = sub4(s.views[p.img[i]].center, p.coord);
        ray = nrm4(ray);
        const float d = dot4(ray, p.normal);
        if (0.0f < d) unit /= d; else unit = (float)(INT_MAX / 2);
        units[i] = unit;
    }
    std::sort(units, units + p.nimg); /* nth_element(begin+1): the second smallest */
    return units[std::min(1, p.nimg - 1)] * s.cfg.csize;
}

/* PatchManager::findNeighbors, patch_manager.cpp:671-728 */
/* Engine arithmetic (ORC_SUM_TREE64): the order in which filterQuad's sums run over the neighbours.  The GPU finds
 * the neighbours through a hash set of every id it meets in the scanned cell lists (`visited`), an ordered
 * linear-probing table (each key sits behind larger keys only), whose layout does not depend on the order of the
 * insertions: it is the layout of inserting the keys in descending order with plain linear probing.  The accepted ids
 * are taken in slot order.  Table size: first_cap slots while at most 7/8 of them are visited and at most first_rows ids
 * are accepted (mvs_check.cuh: 2048 / 576 -- MVS_HASH_CAP, MVS_ROW_CAP; 4096 / 1152 in Optim::check of the 64-view build; 2048 / 448 in
 * Filter::filterNeighbor, MVS_FILTER_ROW_CAP),
 * else 16384 (Filter::filterNeighbor's second launch). */
void engine_neighbor_order(std::vector<int> visited, std::vector<int>& nb /* sorted unique in, slot order out */, size_t first_cap, size_t first_rows) {
    std::sort(visited.begin(), visited.end());
    visited.erase(std::unique(visited.begin(), visited.end()), visited.end());
    size_t cap = (visited.size() <= first_cap / 8 * 7 && nb.size() <= first_rows) ? first_cap : 16384;
    while (visited.size() * 8 > cap * 7) cap *= 2; /* beyond the engine's limits (it reports an error there) */
    std::vector<int> table(cap, -1);
    for (size_t k = visited.size(); k-- > 0;) {
        size_t p = ((uint32_t)visited[k] * 0x9E3779B1u) >> (32 - __builtin_ctzll((unsigned long long)cap)); /* set_home: Fibonacci hashing */
        while (table[p] != -1) p = (p + 1) & (cap - 1);
        table[p] = visited[k];
    }
    std::vector<int> out;
    out.reserve(nb.size());
    for (size_t p = 0; p < cap; ++p)
        if (table[p] >= 0 && std::binary_search(nb.begin(), nb.end(), table[p])) out.push_back(table[p]);
    nb.swap(out);
}

void find_neighbors(const Scene& s, const Patch& p, std::vector<int>& nb, float scale, int margin, const DestCtx* ctx, bool in_check = false) {
    std::vector<int> visited;
    const float radius = (float)(1.5 * margin * compute_radius(s, p));
    float unit = 0.0f;
    for (int i = 0; i < p.nimg; ++i) unit += get_unit(s, p.img[i], p.coord);
    unit /= p.nimg;
    unit *= s.cfg.csize;
    for (int i = 0; i < p.nimg; ++i) {
        const int v = p.img[i];
        const View& vw = s.views[v];
        for (int dy = -margin; dy <= margin; ++dy) {
            const int yt = p.gy[i] + dy;
            if (yt < 0 || vw.gh <= yt) continue;
            for (int dx = -margin; dx <= margin; ++dx) {
                const int xt = p.gx[i] + dx;
                if (xt < 0 || vw.gw <= xt) continue;
                const int cell = yt * vw.gw + xt;
                for (int kind = 0; kind < 2; ++kind) {
                    const Span l = cell_list(s, kind, v, cell, ctx);
                    for (int j = 0; j < l.n; ++j) {
                        visited.push_back(l.p[j]);
                        if (is_neighbor_radius(s, p, get_patch(s, l.p[j], ctx), unit, s.neighborThreshold * scale, radius)) nb.push_back(l.p[j]);
                    }
                }
            }
        }
    }
    std::sort(nb.begin(), nb.end());
    nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
    if (s.cfg.sum_mode == ORC_SUM_TREE64) {
        /* every engine build: a 2048-slot first tier (rounds 3-4 gave the 64-view build's Optim::check 4096 slots, which its 22 KB of LDS
         * per wave had room for; with 16-view chunks of textures it runs in the 12 KB of the other builds), then 16384 */
        engine_neighbor_order(visited, nb, 2048, in_check ? 576 : 448);  /* Filter::filterNeighbor: MVS_FILTER_ROW_CAP */
    }
}

/* Filter::ortho, filter.cpp:394-409 */
void ortho(const V4& z, V4& x, V4& y) {
    if (fabsf(z.x) > 0.5f) x = {z.y, -z.x, 0, 0};
    else if (fabsf(z.y) > 0.5f) x = {0, z.z, -z.y, 0};
    else x = {-z.z, 0, z.x, 0};
    x = div4(x, norm4(x));
    y = {z.y * x.z - z.z * x.y, z.z * x.x - z.x * x.z, z.x * x.y - z.y * x.x, 0};
}

/* Sums over the n neighbours of Filter::filterQuad.  SEQ: the reference's order.  TREE64 (engine): lane l adds the
 * elements l, l + 64, l + 128, ... in that order, then the 64 partial sums go through the wave butterfly. */
float reduce_n_f32(const Scene& s, const std::vector<float>& a) {
    const int n = (int)a.size();
    if (s.cfg.sum_mode != ORC_SUM_TREE64) { float t = 0.0f; for (int i = 0; i < n; ++i) t += a[i]; return t; }
    float part[64];
    for (int l = 0; l < 64; ++l) { part[l] = 0.0f; for (int t = l; t < n; t += 64) part[l] += a[t]; }
    return reduce_tree64(part);
}
double reduce_tree64_f64(const double* a) {
    double b[64], t[64];
    for (int i = 0; i < 64; ++i) b[i] = a[i];
    for (int off = 1; off < 64; off <<= 1) {
        for (int i = 0; i < 64; ++i) t[i] = b[i] + b[i ^ off];
        for (int i = 0; i < 64; ++i) b[i] = t[i];
    }
    return b[0];
}

/* Filter::lls, filter.cpp:411-430: least squares n x 5 (Eigen jacobiSvd there; normal equations with partial-pivot
 * Gaussian elimination in double here).  build_normal_equations forms M = [A^T A | A^T b]; solve5 solves it. */
void build_normal_equations(const Scene& s, const std::vector<std::array<float, 5>>& A, const std::vector<float>& b, double M[5][6]) {
    const int n = (int)A.size();
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 6; ++j) {
        if (s.cfg.sum_mode != ORC_SUM_TREE64) {
            double acc = 0.0;
            for (int r = 0; r < n; ++r) acc += (double)A[r][i] * (double)(j < 5 ? A[r][j] : b[r]);
            M[i][j] = acc;
        } else {
            double part[64];
            for (int l = 0; l < 64; ++l) {
                part[l] = 0.0;
                for (int r = l; r < n; r += 64) part[l] += (double)A[r][i] * (double)(j < 5 ? A[r][j] : b[r]);
            }
            M[i][j] = reduce_tree64_f64(part);
        }
    }
}
bool solve5(double M[5][6], double* x) {
    for (int c = 0; c < 5; ++c) {
        int piv = c;
        for (int r = c + 1; r < 5; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (fabs(M[piv][c]) < 1e-30) return false;
        if (piv != c) for (int k = 0; k < 6; ++k) std::swap(M[c][k], M[piv][k]);
        for (int r = c + 1; r < 5; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int k = c; k < 6; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 4; r >= 0; --r) {
        double acc = M[r][5];
        for (int k = r + 1; k < 5; ++k) acc -= M[r][k] * x[k];
        x[r] = acc / M[r][r];
    }
    return true;
}

/* Filter::filterQuad, filter.cpp:329-392: the residual of the quadric fitted to the neighbours' coordinates */
float quad_residual(const Scene& s, const Patch& p, const std::vector<V4>& nc) {
    V4 xdir, ydir;
    ortho(p.normal, xdir, ydir);
    const int n = (int)nc.size();
    std::vector<float> dist(n);
    for (int i = 0; i < n; ++i) dist[i] = norm4(sub4(nc[i], p.coord));
    float h = reduce_n_f32(s, dist);
    h /= n;
    std::vector<std::array<float, 5>> A(n);
    std::vector<float> b(n), fxs(n), fys(n), fzs(n);
    for (int i = 0; i < n; ++i) {
        const V4 diff = sub4(nc[i], p.coord);
        fxs[i] = dot4(diff, xdir) / h;
        fys[i] = dot4(diff, ydir) / h;
        fzs[i] = dot4(diff, p.normal);
        A[i] = {fxs[i] * fxs[i], fys[i] * fys[i], fxs[i] * fys[i], fxs[i], fys[i]};
        b[i] = fzs[i];
    }
    double xd[5] = {0, 0, 0, 0, 0};
    double M[5][6];
    build_normal_equations(s, A, b, M);
    if (!solve5(M, xd)) for (double& v : xd) v = 0.0;
    float x[5];
    for (int i = 0; i < 5; ++i) x[i] = (float)xd[i];
    const int inum = std::min(s.tau, p.nimg);
    float unit = 0.0f;
    for (int i = 0; i < inum; ++i) unit += get_unit(s, p.img[i], p.coord);
    unit /= inum;
    std::vector<float> rs(n);
    for (int i = 0; i < n; ++i) {
        const float res = x[0] * (fxs[i] * fxs[i]) + x[1] * (fys[i] * fys[i]) + x[2] * (fxs[i] * fys[i]) + x[3] * fxs[i] + x[4] * fys[i] - fzs[i];
        rs[i] = fabsf(res) / unit;
    }
    float residual = reduce_n_f32(s, rs);
    residual /= (n - 5);
    return residual;
}
int filter_quad(const Scene& s, const Patch& p, const std::vector<int>& nb, const DestCtx* ctx) {
    std::vector<V4> nc(nb.size());
    for (size_t i = 0; i < nb.size(); ++i) nc[i] = get_patch(s, nb[i], ctx).coord;
    return quad_residual(s, p, nc) < s.cfg.quadThreshold ? 0 : 1;
}

/* Optim::check, optim.cpp:300-323 */
int check_patch(const Scene& s, Patch& p, const DestCtx* ctx) {
    const float gain = compute_gain(s, p, ctx);
    p.tmp = gain;
    if (gain < 0.0f) { p.nimg = 0; return 1; }
    std::vector<int> nb;
    find_neighbors(s, p, nb, 4.0f, 2, ctx, true);
    if (6 < (int)nb.size() && filter_quad(s, p, nb, ctx)) { p.nimg = 0; return 1; }
    return 0;
}

/* ------------------------------------------------------------------ Filter::run (pmmvps/filter.cpp:25-49) */
void set_ref_image(const Scene& s, Patch& p, orc_counters* cnt);
void build_csr(Scene& s, bool vgrid);
void build_depth_maps(Scene& s);
void rebuild_live_from_pool(Scene& s);

/* Filter::setDepthMapsVGridsVPGridsAddPatchV, filter.cpp:628-655: depth maps from the alive patches, m_vimages
 * recomputed (additive == 0) or extended (additive == 1), both cell indices rebuilt.  No MAX_NUM_OF_PATCHES trim here. */
void filter_rebuild(Scene& s, int additive) {
    for (Patch& p : s.pool) if (p.alive) set_grids(s, p);
    build_csr(s, false);
    build_depth_maps(s);
    for (Patch& p : s.pool) {
        if (!p.alive) continue;
        if (additive == 0) p.nvimg = 0;
        set_vgrids(s, p);
        set_vimages_vgrids(s, p, nullptr);
    }
    build_csr(s, true);
    if (s.cfg.schedule == ORC_SCHEDULE_FAITHFUL) rebuild_live_from_pool(s);
}

/* Filter::filterOutside, filter.cpp:51-106: all gains on the same snapshot, then the removals */
int filter_outside(Scene& s) {
    std::vector<int> dead;
    for (size_t id = 0; id < s.pool.size(); ++id)
        if (s.pool[id].alive && compute_gain(s, s.pool[id], nullptr) < 0.0f) dead.push_back((int)id);
    for (int id : dead) s.pool[id].alive = false;
    return (int)dead.size();
}

/* Filter::filterExact, filter.cpp:148-263: a view stays in m_images if the patch passes the depth-map test in its
 * cell or one of the 4 neighbouring cells (filterExactSub 211-263); the surviving views are listed in ascending
 * view order (the image-major loop), the reference view is re-picked (setRefImage), too few views = removal. */
int filter_exact(Scene& s) {
    int removed = 0;
    std::vector<Patch> updated;
    std::vector<int> ids;
    for (size_t id = 0; id < s.pool.size(); ++id) {
        Patch p = s.pool[id];
        if (!p.alive) continue;
        int keepv[256], nk = 0;
        for (int image = 0; image < s.cfg.nviews; ++image) {
            int k = -1;
            for (int i = 0; i < p.nimg; ++i) if (p.img[i] == image) { k = i; break; }
            if (k < 0) continue;
            const View& vw = s.views[image];
            const int x = p.gx[k], y = p.gy[k], w = vw.gw, h = vw.gh;
            if (x < 0 || w <= x || y < 0 || h <= y) continue; /* not in any m_pgrids list */
            int safe = 0;
            if (is_visible(s, p, image, x, y, s.neighborThreshold1, nullptr)) safe = 1;
            else if (0 < x && is_visible(s, p, image, x - 1, y, s.neighborThreshold1, nullptr)) safe = 1;
            else if (x < w - 1 && is_visible(s, p, image, x + 1, y, s.neighborThreshold1, nullptr)) safe = 1;
            else if (0 < y && is_visible(s, p, image, x, y - 1, s.neighborThreshold1, nullptr)) safe = 1;
            else if (y < h - 1 && is_visible(s, p, image, x, y + 1, s.neighborThreshold1, nullptr)) safe = 1;
            if (safe) keepv[nk++] = image;
        }
        p.nimg = std::min(nk, LISTCAP);
        for (int i = 0; i < p.nimg; ++i) p.img[i] = keepv[i];
        if (s.cfg.minImageNum <= p.nimg) { set_ref_image(s, p, &s.cnt); set_grids(s, p); }
        else { p.alive = false; ++removed; }
        updated.push_back(p);
        ids.push_back((int)id);
    }
    for (size_t k = 0; k < ids.size(); ++k) s.pool[ids[k]] = updated[k];
    return removed;
}

/* Filter::filterNeighbor(1), filter.cpp:265-327 */
int filter_neighbor(Scene& s) {
    std::vector<int> dead;
    for (size_t id = 0; id < s.pool.size(); ++id) {
        if (!s.pool[id].alive) continue;
        std::vector<int> nb;
        find_neighbors(s, s.pool[id], nb, 4.0f, 2, nullptr);
        if ((int)nb.size() < 6 || filter_quad(s, s.pool[id], nb, nullptr)) dead.push_back((int)id);
    }
    for (int id : dead) s.pool[id].alive = false;
    return (int)dead.size();
}

/* Filter::filterSmallGroups, filter.cpp:432-578.  Patch q hangs on patch p when q is listed (m_pgrids or m_vpgrids) in
 * one of the 3x3 cells around p in p's reference view and isNeighbor(p, q, m_neighborThreshold2).
 * FAITHFUL: the reference's breadth-first labelling in patch order (the relation is directed, so the grouping depends
 * on that order).  ENGINE: connected components of the symmetrised relation (union-find).  Groups smaller than
 * max(20, N / 10000) are removed. */
void small_group_edges(const Scene& s, int pid, std::vector<int>& out) {
    const Patch& p = s.pool[pid];
    const int v = p.img[0];
    const View& vw = s.views[v];
    for (int y = -1; y <= 1; ++y) {
        const int yt = p.gy[0] + y;
        if (yt < 0 || vw.gh <= yt) continue;
        for (int x = -1; x <= 1; ++x) {
            const int xt = p.gx[0] + x;
            if (xt < 0 || vw.gw <= xt) continue;
            for (int kind = 0; kind < 2; ++kind) {
                const Span l = cell_list(s, kind, v, yt * vw.gw + xt, nullptr);
                for (int j = 0; j < l.n; ++j)
                    if (is_neighbor(s, p, s.pool[l.p[j]], s.neighborThreshold2)) out.push_back(l.p[j]);
            }
        }
    }
}
/* mode: 0 = the schedule's own labelling, 1 = the literal breadth-first labelling, 2 = components; dead_out: the patches that
 * would be removed are listed there and the pool is left alone (orc_small_groups_compare), else they are removed */
int filter_small_groups(Scene& s, int mode = 0, std::vector<int>* dead_out = nullptr) {
    std::vector<int> alive;
    for (size_t id = 0; id < s.pool.size(); ++id) if (s.pool[id].alive) alive.push_back((int)id);
    const int psize = (int)alive.size();
    if (psize == 0) return 0;
    std::vector<int> label(s.pool.size(), -1);
    int ngroups = 0;
    const bool literal = mode == 1 || (mode == 0 && (s.cfg.schedule == ORC_SCHEDULE_FAITHFUL || s.cfg.literal_groups));
    if (literal) {
        for (int root : alive) {
            if (label[root] != -1) continue;
            const int id = ngroups++;
            label[root] = id;
            std::vector<int> queue{root};
            for (size_t qh = 0; qh < queue.size(); ++qh) {
                std::vector<int> e;
                small_group_edges(s, queue[qh], e);
                for (int q : e) if (label[q] == -1) { label[q] = id; queue.push_back(q); }
            }
        }
    } else {
        std::vector<int> parent(s.pool.size());
        for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
        auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
        for (int pid : alive) {
            std::vector<int> e;
            small_group_edges(s, pid, e);
            for (int q : e) {
                int a = find(pid), b = find(q);
                if (a == b) continue;
                if (a > b) std::swap(a, b);
                parent[b] = a; /* smaller id is the root */
            }
        }
        for (int pid : alive) label[pid] = find(pid);
        ngroups = (int)s.pool.size();
    }
    std::vector<int> size(std::max(ngroups, 1), 0);
    for (int pid : alive) ++size[label[pid]];
    const int threshold = std::max(20, psize / 10000);
    int removed = 0;
    for (int pid : alive) if (size[label[pid]] < threshold) {
        if (dead_out) dead_out->push_back(pid); else s.pool[pid].alive = false;
        ++removed;
    }
    return removed;
}

void filter_run(Scene& s, int64_t* removed4) {
    filter_rebuild(s, 0);
    removed4[0] = filter_outside(s);
    filter_rebuild(s, 1);
    removed4[1] = filter_exact(s);
    filter_rebuild(s, 1);
    removed4[2] = filter_neighbor(s);
    filter_rebuild(s, 1);
    removed4[3] = filter_small_groups(s);
    filter_rebuild(s, 1);
}

/* PhotoSet::getMask(coord, level), photoSet.cpp:223-233 + Photo::getMask, photo.cpp:48-55 */
int get_mask_all(const Scene& s, const V4& coord) {
    for (int v = 0; v < s.cfg.nviews; ++v) {
        const View& vw = s.views[v];
        if (vw.mask.empty() || vw.mask[s.cfg.level].empty()) continue;
        const V3 ic = project(vw, coord, s.cfg.level);
        if (get_mask(vw, ic.x, ic.y, s.cfg.level) == 0) return 0;
    }
    return -1;
}

/* Optim::postProcess, optim.cpp:260-298 */
int post_process(const Scene& s, Patch& p, const DestCtx* ctx, orc_counters* cnt, const float* w_keep = nullptr) {
    if (p.nimg < s.cfg.minImageNum) return -1;
    if (get_mask_all(s, p.coord) == 0) return -1;
    const int n_keep = p.nimg;
    add_images(s, p);
    constraint_images(s, p, s.nccThreshold, cnt, w_keep, n_keep);
    filter_images_by_angle(s, p);
    if (p.nimg < s.cfg.minImageNum) return -1;
    set_grids(s, p);
    const int ref_before = p.img[0];
    /* engine schedule: setRefImage works on the textures the constraintImages above has just sampled (same patch, same
     * reference view, a subset of its views) -- the engine keeps them in LDS and does not sample or count them again */
    set_ref_image(s, p, (s.cfg.schedule == ORC_SCHEDULE_ENGINE && !s.cfg.literal_evals) ? nullptr : cnt);
    /* engine schedule: with the reference view unchanged the second constraintImages would sample the very textures of
     * the first one for the views that passed it, under the same threshold, and remove nothing -- it is not run */
    if (s.cfg.schedule != ORC_SCHEDULE_ENGINE || s.cfg.literal_evals || p.img[0] != ref_before) constraint_images(s, p, s.nccThreshold, cnt);
    if (p.nimg < s.cfg.minImageNum) return -1;
    set_grids(s, p);
    p.tmp = score2(p, s.nccThreshold);
    if (s.depth) set_vimages_vgrids(s, p, ctx);
    if (2 <= s.depth && s.cfg.enable_check && check_patch(s, p, ctx)) return -1;
    return 0;
}

/* Propagate::generatePatch, propagate.cpp:220-237 */
/* as_image >= 0 (view propagation, trash/propagate_view_propagation_simiar_to_original.cpp:125-147): the patch is
 * re-anchored on the ray of view `as_image` instead of its reference view -- depth along that view's optical axis,
 * back-projection through that view -- and Optim::swapImage (optim.cpp:385-395) makes it the reference view. */
/* need_ncc = false (engine schedule, destination cell not full): the initial m_ncc is read by the replace-worst pre-filter
 * only (propagate.cpp:170) and overwritten by refinePatch, so the engine does not compute it when nothing reads it. */
bool generate_patch(const Scene& s, const Patch& src, const V3& icoord, Patch& out, orc_counters* cnt, int as_image = -1, bool need_ncc = true) {
    out = Patch();
    int images[MAXI];
    for (int i = 0; i < src.nimg; ++i) images[i] = src.img[i];
    if (as_image >= 0 && images[0] != as_image) {
        int k = -1;
        for (int i = 1; i < src.nimg; ++i) if (images[i] == as_image) { k = i; break; }
        if (k < 0) return false;
        std::swap(images[0], images[k]);
    }
    const int image = images[0];
    const View& vw = s.views[image];
    const float depth = dot4(vw.oaxis, src.coord);
    const V3 nic{depth * icoord.x, depth * icoord.y, depth * icoord.z};
    out.coord = unproject(vw, nic, s.cfg.level);
    out.normal = src.normal;
    set_grids_images(s, out, images, src.nimg);
    if (out.nimg == 0) return false;
    if (need_ncc) out.ncc = compute_ncc(s, out, cnt);
    return true;
}

/* ------------------------------------------------------------------ faithful PatchManager (live lists) */
/* PatchManager::updateDepthMaps, patch_manager.cpp:191-221 */
void update_depth_maps(Scene& s, int id) {
    const Patch& p = s.pool[id];
    for (int image = 0; image < s.cfg.nviews; ++image) {
        const View& vw = s.views[image];
        const V3 ic = project(vw, p.coord, s.cfg.level);
        const float fx = ic.x / s.cfg.csize, fy = ic.y / s.cfg.csize;
        const int xs[2] = {(int)floorf(fx), (int)ceilf(fx)}, ys[2] = {(int)floorf(fy), (int)ceilf(fy)};
        const float depth = dot4(vw.oaxis, p.coord);
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) {
            if (xs[i] < 0 || vw.gw <= xs[i] || ys[j] < 0 || vw.gh <= ys[j]) continue;
            int& slot = s.dpgrids[image][ys[j] * vw.gw + xs[i]];
            if (slot < 0) slot = id;
            else if (depth < dot4(vw.oaxis, s.pool[slot].coord)) slot = id;
        }
    }
}
/* PatchManager::addPatch, patch_manager.cpp:158-189 */
void add_patch_live(Scene& s, int id) {
    Patch& p = s.pool[id];
    for (int i = 0; i < p.nimg; ++i) s.pgrids[p.img[i]][p.gy[i] * s.views[p.img[i]].gw + p.gx[i]].push_back(id);
    if (s.depth == 0) return;
    for (int i = 0; i < p.nvimg; ++i) s.vpgrids[p.vimg[i]][p.vgy[i] * s.views[p.vimg[i]].gw + p.vgx[i]].push_back(id);
    update_depth_maps(s, id);
}
/* PatchManager::removePatch, patch_manager.cpp:303-325 */
void remove_patch_live(Scene& s, int id) {
    Patch& p = s.pool[id];
    for (int i = 0; i < p.nimg; ++i) {
        std::vector<int>& l = s.pgrids[p.img[i]][p.gy[i] * s.views[p.img[i]].gw + p.gx[i]];
        l.erase(std::remove(l.begin(), l.end(), id), l.end());
    }
    for (int i = 0; i < p.nvimg; ++i) {
        std::vector<int>& l = s.vpgrids[p.vimg[i]][p.vgy[i] * s.views[p.vimg[i]].gw + p.vgx[i]];
        l.erase(std::remove(l.begin(), l.end(), id), l.end());
    }
    p.alive = false;
}
/* PatchManager::sortPatches (descending), patch_manager.cpp:406-433: the O(n^2) exchange sort */
void sort_patches_live(Scene& s, std::vector<int>& l) {
    const int n = (int)l.size();
    for (int i = 0; i < n; ++i) if (s.pool[l[i]].ncc < 0.0f) s.pool[l[i]].ncc = compute_ncc(s, s.pool[l[i]], &s.cnt);
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j)
        if (s.pool[l[i]].ncc < s.pool[l[j]].ncc) std::swap(l[i], l[j]);
}

uint64_t g_faithful_counter = 0;

/* Propagate::propagatePatch(ppatch, image, index), propagate.cpp:126-218 */
void propagate_patch_faithful(Scene& s, int src, int image, int index) {
    {
        std::vector<int>& l = s.pgrids[image][index];
        sort_patches_live(s, l);
        int np = (int)l.size();
        if (np > s.cap) {
            std::vector<int> extra(l.begin() + s.cap, l.end());
            for (int k = (int)extra.size() - 1; k >= 0; --k) { remove_patch_live(s, extra[k]); s.cnt.trimmed++; }
        }
    }
    std::minstd_rand0 generator; /* std::default_random_engine on libstdc++ (D4) */
    std::uniform_real_distribution<float> distribution(-0.5, 0.5);
    const int gw = s.views[image].gw;
    const int cx = index % gw, cy = index / gw;
    const V3 icoord{(s.cfg.csize * (2 * cx + 1) - 1) / 2.0f, (s.cfg.csize * (2 * cy + 1) - 1) / 2.0f, 1.0f};
    for (int it = 0; it < s.cfg.max_propag; ++it) {
        std::vector<int>& l = s.pgrids[image][index];
        const int np = (int)l.size();
        Patch cand;
        if (np < s.cap) {
            const float a = distribution(generator) * s.cfg.csize;
            const float b = distribution(generator) * s.cfg.csize;
            const V3 nic{icoord.x + a, icoord.y + b, icoord.z + 0.0f};
            if (!generate_patch(s, s.pool[src], nic, cand, &s.cnt)) continue;
            s.cnt.candidates++;
        } else {
            sort_patches_live(s, l);
            const int worst = l[s.cap - 1];
            const V3 ic = project(s.views[image], s.pool[worst].coord, s.cfg.level);
            if (!generate_patch(s, s.pool[src], ic, cand, &s.cnt)) continue;
            s.cnt.candidates++;
            if (cand.ncc < s.pool[worst].ncc) { s.cnt.prefiltered++; continue; }
        }
        s.cnt.patches++;
        if (pre_process(s, cand, &s.cnt) == -1) { s.cnt.fail0++; continue; }
        const uint64_t c = g_faithful_counter++;
        const uint32_t key[4] = {0xFA17u, (uint32_t)(c & 0xffffffffu), (uint32_t)(c >> 32), 0u};
        refine_patch(s, cand, key, &s.cnt);
        if (post_process(s, cand, nullptr, &s.cnt) == -1) { s.cnt.fail1++; continue; }
        if (np == s.cap) {
            std::vector<int>& l2 = s.pgrids[image][index];
            remove_patch_live(s, l2[s.cap - 1]);
            s.cnt.replaced++;
        } else s.cnt.inserted++;
        cand.alive = true;
        s.pool.push_back(cand);
        add_patch_live(s, (int)s.pool.size() - 1);
    }
}

/* Propagate::propagatePmImage, propagate.cpp:72-124 (D1: queue dropped; D2: out-of-grid and
 * row-wrapping targets skipped; D3: the `end` cell is never a source, kept) */
void propagate_faithful(Scene& s, int iter) {
    int64_t visited = 0;
    const auto t_start = std::chrono::steady_clock::now();
    for (int image = 0; image < s.cfg.nviews; ++image) {
        const int gw = s.views[image].gw, gh = s.views[image].gh;
        int start = 0, end = gw * gh - 1, inc = 1;
        if (iter % 2 == 1) { start = gw * gh - 1; end = 0; inc = -1; }
        for (int index = start; index != end; index += inc) {
            std::vector<int> l = s.pgrids[image][index]; /* copy, propagate.cpp:88 */
            sort_patches_live(s, l);
            int np = (int)l.size();
            if (np == 0) continue;
            if (s.cell_budget > 0 && visited >= s.cell_budget) return;
            if (s.time_budget > 0.0 && (visited & 15) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > s.time_budget) return;
            ++visited;
            if (np > s.cap) {
                for (int i = np - 1; i >= s.cap; --i) { remove_patch_live(s, l[i]); s.cnt.trimmed++; }
                np = s.cap;
            }
            const int cx = index % gw, cy = index / gw;
            for (int n = 0; n < np; ++n) {
                if (!s.pool[l[n]].alive) continue; /* evicted while this cell was being processed */
                if (s.pool[l[n]].img[0] != image) continue;
                const int tx = cx + inc, ty = cy + inc;
                if (0 <= tx && tx < gw) propagate_patch_faithful(s, l[n], image, index + inc);
                if (!s.pool[l[n]].alive) continue;
                if (0 <= ty && ty < gh) propagate_patch_faithful(s, l[n], image, index + inc * gw);
            }
        }
    }
}

/* ------------------------------------------------------------------ engine schedule */
inline bool rank_before(const Scene& s, int a, int b, const DestCtx* ctx) { /* (ncc desc, id asc) */
    const float na = get_patch(s, a, ctx).ncc, nb = get_patch(s, b, ctx).ncc;
    if (na != nb) return na > nb;
    return a < b;
}

/* Rebuild the per-(view, cell) CSR lists from the alive pool (PatchManager::addPatch semantics,
 * patch_manager.cpp:158-186), each list sorted (ncc desc, id asc) as sortPatches leaves it. */
void build_csr(Scene& s, bool vgrid) {
    auto& starts = vgrid ? s.vcsr_start : s.csr_start;
    auto& idsv = vgrid ? s.vcsr_ids : s.csr_ids;
    starts.assign(s.cfg.nviews, {});
    idsv.assign(s.cfg.nviews, {});
    for (int v = 0; v < s.cfg.nviews; ++v) starts[v].assign((size_t)s.views[v].gw * s.views[v].gh + 1, 0);
    for (size_t id = 0; id < s.pool.size(); ++id) {
        const Patch& p = s.pool[id];
        if (!p.alive) continue;
        const int n = vgrid ? p.nvimg : p.nimg;
        for (int i = 0; i < n; ++i) {
            const int v = vgrid ? p.vimg[i] : p.img[i];
            const int gx = vgrid ? p.vgx[i] : p.gx[i], gy = vgrid ? p.vgy[i] : p.gy[i];
            const View& vw = s.views[v];
            if (gx < 0 || vw.gw <= gx || gy < 0 || vw.gh <= gy) continue;
            starts[v][(size_t)gy * vw.gw + gx + 1]++;
        }
    }
    for (int v = 0; v < s.cfg.nviews; ++v) {
        for (size_t c = 1; c < starts[v].size(); ++c) starts[v][c] += starts[v][c - 1];
        idsv[v].assign(starts[v].back(), -1);
    }
    std::vector<std::vector<int>> fill(s.cfg.nviews);
    for (int v = 0; v < s.cfg.nviews; ++v) fill[v].assign(starts[v].begin(), starts[v].end() - 1);
    for (size_t id = 0; id < s.pool.size(); ++id) {
        const Patch& p = s.pool[id];
        if (!p.alive) continue;
        const int n = vgrid ? p.nvimg : p.nimg;
        for (int i = 0; i < n; ++i) {
            const int v = vgrid ? p.vimg[i] : p.img[i];
            const int gx = vgrid ? p.vgx[i] : p.gx[i], gy = vgrid ? p.vgy[i] : p.gy[i];
            const View& vw = s.views[v];
            if (gx < 0 || vw.gw <= gx || gy < 0 || vw.gh <= gy) continue;
            idsv[v][fill[v][(size_t)gy * vw.gw + gx]++] = (int)id;
        }
    }
    for (int v = 0; v < s.cfg.nviews; ++v)
        for (size_t c = 0; c + 1 < starts[v].size(); ++c)
            std::sort(idsv[v].begin() + starts[v][c], idsv[v].begin() + starts[v][c + 1],
                      [&](int a, int b) { return rank_before(s, a, b, nullptr); });
}

/* Rebuild m_dpgrids from the alive pool (Filter::setDepthMaps, filter.cpp:580-626, same rule as
 * updateDepthMaps: nearest depth wins, first come (= lowest id) on ties). */
void build_depth_maps(Scene& s) {
    for (int v = 0; v < s.cfg.nviews; ++v) s.dpgrids[v].assign((size_t)s.views[v].gw * s.views[v].gh, -1);
    for (size_t id = 0; id < s.pool.size(); ++id) if (s.pool[id].alive) update_depth_maps(s, (int)id);
}

void engine_prepare(Scene& s) {
    /* sortPatches: patches with ncc < 0 get their score first (patch_manager.cpp:411-415) */
    for (Patch& p : s.pool) if (p.alive && p.ncc < 0.0f) p.ncc = compute_ncc(s, p, &s.cnt);
    for (Patch& p : s.pool) if (p.alive) { set_grids(s, p); set_vgrids(s, p); }
    build_csr(s, false);
    /* trim every list to MAX_NUM_OF_PATCHES (propagate.cpp:94-99,130-135), all cells decide on the
     * same snapshot; a trimmed patch is removed from every view (removePatch). */
    bool any = false;
    for (int v = 0; v < s.cfg.nviews; ++v)
        for (size_t c = 0; c + 1 < s.csr_start[v].size(); ++c)
            for (int k = s.csr_start[v][c] + s.cap; k < s.csr_start[v][c + 1]; ++k) {
                Patch& p = s.pool[s.csr_ids[v][k]];
                if (p.alive) { p.alive = false; s.cnt.trimmed++; any = true; }
            }
    if (any) build_csr(s, false);
    build_csr(s, true);
    build_depth_maps(s);
}

void insert_sorted(const Scene& s, DestCtx& ctx, int id) {
    size_t pos = 0;
    while (pos < ctx.list.size() && rank_before(s, ctx.list[pos], id, &ctx)) ++pos;
    ctx.list.insert(ctx.list.begin() + pos, id);
}

/* Propagate::propagatePatch on the live list of one destination cell, propagate.cpp:126-218 */
void propagate_patch_engine(const Scene& s, DestCtx& ctx, int src, int image, int index, int iter, int srcslot, orc_counters& cnt,
                            int as_image = -1) {
    const int gw = s.views[image].gw;
    const int cx = index % gw, cy = index / gw;
    const V3 icoord{(s.cfg.csize * (2 * cx + 1) - 1) / 2.0f, (s.cfg.csize * (2 * cy + 1) - 1) / 2.0f, 1.0f};
    const Patch srcp = s.pool[src];
    for (int it = 0; it < s.cfg.max_propag; ++it) {
        const int np = (int)ctx.list.size();
        const uint32_t key[4] = {(uint32_t)iter, (uint32_t)image, (uint32_t)index, (uint32_t)(srcslot * 16 + it)};
        Patch cand;
        int worst = -1;
        if (np < s.cap) {
            const float a = rng_uniform(s.cfg.seed, key[0], key[1], key[2], key[3], 0) * s.cfg.csize;
            const float b = rng_uniform(s.cfg.seed, key[0], key[1], key[2], key[3], 1) * s.cfg.csize;
            const V3 nic{icoord.x + a, icoord.y + b, 1.0f};
            if (!generate_patch(s, srcp, nic, cand, &cnt, as_image, s.cfg.literal_evals != 0)) continue;
            cnt.candidates++;
        } else {
            worst = ctx.list[s.cap - 1];
            const Patch& wp = get_patch(s, worst, &ctx);
            const V3 ic = project(s.views[image], wp.coord, s.cfg.level);
            if (!generate_patch(s, srcp, ic, cand, &cnt, as_image)) continue;
            cnt.candidates++;
            if (cand.ncc < wp.ncc) { cnt.prefiltered++; continue; }
        }
        cnt.patches++;
        if (pre_process(s, cand, &cnt) == -1) { cnt.fail0++; continue; }
        float w_keep[MAXI];
        float* const wk = s.cfg.literal_evals ? nullptr : w_keep;  /* literal: refinePatch evaluates its own final m_ncc */
        refine_patch(s, cand, key, &cnt, wk);
        if (post_process(s, cand, &ctx, &cnt, wk) == -1) { cnt.fail1++; continue; }
        if (np == s.cap) { /* removePatch(worst), propagate.cpp:198-201 */
            ctx.list.erase(ctx.list.begin() + (s.cap - 1));
            if (worst >= NEWBASE) ctx.staged[worst - NEWBASE].alive = false;
            else ctx.kills.push_back(worst);
            cnt.replaced++;
        } else cnt.inserted++;
        cand.alive = true;
        cand.sweep_view = image;
        cand.dest_cell = index;
        ctx.staged.push_back(cand);
        const int nid = NEWBASE + (int)ctx.staged.size() - 1;
        for (int i = 0; i < cand.nimg; ++i) /* addPatch into this cell's live list if it lands here */
            if (cand.img[i] == image && cand.gy[i] * gw + cand.gx[i] == index) { insert_sorted(s, ctx, nid); break; }
    }
}

void dest_cell_engine(const Scene& s, DestCtx& ctx, int image, int index, int iter, int inc, orc_counters& cnt) {
    const View& vw = s.views[image];
    const int gw = vw.gw, gh = vw.gh;
    const int cx = index % gw, cy = index / gw;
    ctx.v = image; ctx.cell = index;
    {
        const int b = s.csr_start[image][index], e = s.csr_start[image][index + 1];
        ctx.list.assign(s.csr_ids[image].begin() + b, s.csr_ids[image].begin() + e);
    }
    /* sources in the order the raster sweep reaches them: the cell above/below first (index - inc*gw),
     * then the cell beside (index - inc); propagate.cpp:106-108 */
    const int sx[2] = {cx, cx - inc}, sy[2] = {cy - inc, cy};
    for (int sidx = 0; sidx < 2; ++sidx) {
        if (sx[sidx] < 0 || gw <= sx[sidx] || sy[sidx] < 0 || gh <= sy[sidx]) continue;
        const int scell = sy[sidx] * gw + sx[sidx];
        const int b = s.csr_start[image][scell], e = s.csr_start[image][scell + 1];
        for (int n = 0; n < e - b; ++n) {
            const int sid = s.csr_ids[image][b + n];
            if (s.pool[sid].img[0] != image) continue;
            propagate_patch_engine(s, ctx, sid, image, index, iter, sidx * s.cap + n, cnt);
        }
    }
    /* View propagation -- the branch the reference keeps commented out (propagate.cpp:110-120), after the design in
     * trash/propagate_view_propagation_simiar_to_original.cpp:95-147: a patch of another reference view that is listed
     * in this cell (it is visible in `image` and projects here) proposes itself with `image` as the reference view.
     * Same trial logic as the spatial sources (fill / replace-worst, pre/refine/post); sources are this cell's list
     * as it stood at the start of the pass. */
    if (s.cfg.view_propagation) {
        const int b = s.csr_start[image][index], e = s.csr_start[image][index + 1];
        for (int n = 0; n < e - b; ++n) {
            const int sid = s.csr_ids[image][b + n];
            if (s.pool[sid].img[0] == image) continue;
            propagate_patch_engine(s, ctx, sid, image, index, iter, 2 * s.cap + n, cnt, image);
        }
    }
}

void add_counters(orc_counters& a, const orc_counters& b) {
    a.candidates += b.candidates; a.prefiltered += b.prefiltered; a.patches += b.patches; a.fail0 += b.fail0;
    a.fail1 += b.fail1; a.inserted += b.inserted; a.replaced += b.replaced; a.evals += b.evals;
    a.view_evals += b.view_evals; a.trimmed += b.trimmed;
}

void engine_pass(Scene& s, int iter, int pass) {
    engine_prepare(s);
    const int inc = (iter % 2 == 0) ? 1 : -1;
    const int colour = pass & 1;
    s.staged_cells.clear();
    struct Job { int v, cell; };
    std::vector<Job> jobs;
    const bool ranged = s.cfg.shard_count > 1;
    const int vb = ranged ? 0 : s.cfg.view_begin, vs = ranged ? 1 : std::max(1, s.cfg.view_stride);
    for (int v = vb; v < s.cfg.nviews; v += vs) {
        const View& vw = s.views[v];
        for (int cy = 0; cy < vw.gh; ++cy) for (int cx = 0; cx < vw.gw; ++cx) {
            if (((cx + cy) & 1) != colour) continue;
            /* skip cells with no possible source */
            const int sx[2] = {cx, cx - inc}, sy[2] = {cy - inc, cy};
            bool has = false;
            for (int k = 0; k < 2 && !has; ++k) {
                if (sx[k] < 0 || vw.gw <= sx[k] || sy[k] < 0 || vw.gh <= sy[k]) continue;
                const int sc = sy[k] * vw.gw + sx[k];
                has = s.csr_start[v][sc + 1] > s.csr_start[v][sc];
            }
            if (s.cfg.view_propagation) { const int sc = cy * vw.gw + cx; has = has || s.csr_start[v][sc + 1] > s.csr_start[v][sc]; }
            if (has) jobs.push_back({v, cy * vw.gw + cx});
        }
    }
    if (ranged) { /* this shard's contiguous range of the (view, cell) sequence */
        const size_t lo = jobs.size() * (size_t)s.cfg.shard_index / (size_t)s.cfg.shard_count;
        const size_t hi = jobs.size() * (size_t)(s.cfg.shard_index + 1) / (size_t)s.cfg.shard_count;
        jobs = std::vector<Job>(jobs.begin() + lo, jobs.begin() + hi);
    }
    std::vector<DestCtx> out(jobs.size());
    const int nthreads = std::max(1, s.cfg.nthreads);
    std::vector<orc_counters> tc(nthreads);
    for (auto& c : tc) memset(&c, 0, sizeof c);
    /* time budget (bench.py's all-core CPU figure): destination cells not started when it runs out are skipped */
    const auto t_start = std::chrono::steady_clock::now();
    int stop = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (long j = 0; j < (long)jobs.size(); ++j) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        if (s.time_budget > 0.0) {
            int st;
#pragma omp atomic read
            st = stop;
            if (st) continue;
            if ((j & 15) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > s.time_budget) {
#pragma omp atomic write
                stop = 1;
                continue;
            }
        }
        dest_cell_engine(s, out[j], jobs[j].v, jobs[j].cell, iter, inc, tc[tid]);
    }
    s.last_sweep_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    for (auto& c : tc) add_counters(s.cnt, c);
    for (auto& ctx : out) if (!ctx.staged.empty() || !ctx.kills.empty()) s.staged_cells.push_back(std::move(ctx));
}

void to_rec(const Patch& p, int id, orc_patch& r) {
    memset(&r, 0, sizeof r);
    r.coord[0] = p.coord.x; r.coord[1] = p.coord.y; r.coord[2] = p.coord.z; r.coord[3] = p.coord.w;
    r.normal[0] = p.normal.x; r.normal[1] = p.normal.y; r.normal[2] = p.normal.z; r.normal[3] = p.normal.w;
    r.ncc = p.ncc; r.dscale = p.dscale; r.ascale = p.ascale; r.tmp = p.tmp;
    r.nimages = p.nimg; r.nvimages = p.nvimg; r.flags = p.alive ? 1 : 0; r.id = id;
    for (int i = 0; i < std::min(p.nimg, (int)ORC_MAX_IMAGES); ++i) r.images[i] = (uint8_t)p.img[i];   /* a record holds 32 (wide build: the tail is cut) */
    for (int i = 0; i < std::min(p.nvimg, (int)ORC_MAX_IMAGES); ++i) r.vimages[i] = (uint8_t)p.vimg[i];
}
void from_rec(const Scene& s, const orc_patch& r, Patch& p) {
    p = Patch();
    p.coord = {r.coord[0], r.coord[1], r.coord[2], r.coord[3]};
    p.normal = {r.normal[0], r.normal[1], r.normal[2], r.normal[3]};
    p.ncc = r.ncc; p.dscale = r.dscale; p.ascale = r.ascale; p.tmp = r.tmp;
    p.nimg = std::min(std::min(r.nimages, LISTCAP), (int)ORC_MAX_IMAGES); p.nvimg = std::min(std::min(r.nvimages, LISTCAP), (int)ORC_MAX_IMAGES);
    for (int i = 0; i < p.nimg; ++i) p.img[i] = r.images[i];
    for (int i = 0; i < p.nvimg; ++i) p.vimg[i] = r.vimages[i];
    p.alive = true;
}

void engine_commit_local(Scene& s) {
    for (const DestCtx& ctx : s.staged_cells) for (int id : ctx.kills) s.pool[id].alive = false;
    for (const DestCtx& ctx : s.staged_cells) for (const Patch& p : ctx.staged) if (p.alive) s.pool.push_back(p);
    s.staged_cells.clear();
}

void rebuild_live_from_pool(Scene& s) {
    for (int v = 0; v < s.cfg.nviews; ++v) {
        const size_t n = (size_t)s.views[v].gw * s.views[v].gh;
        s.pgrids[v].assign(n, {}); s.vpgrids[v].assign(n, {}); s.dpgrids[v].assign(n, -1);
    }
    for (size_t id = 0; id < s.pool.size(); ++id) {
        Patch& p = s.pool[id];
        if (!p.alive) continue;
        set_grids(s, p); set_vgrids(s, p);
        add_patch_live(s, (int)id);
    }
}

void derive_thresholds(Scene& s) { /* PmMvps::init, pmmvps.cpp:32-36,54-67 */
    const orc_config& c = s.cfg;
    s.tau = std::min(c.minImageNum * 2, c.nviews);
    s.maxLevel = c.level + 3;
    s.cap = c.max_propag * c.csize * c.csize;
    s.list_cap = c.list_cap > 0 ? std::min(c.list_cap, MAXI) : LISTCAP_DEFAULT;
    s.nccThreshold = c.nccThreshold;
    s.nccThresholdBefore = c.nccThreshold - 0.3f;
    s.angleThreshold0 = (float)(60.0f * M_PI / 180.0f);
    s.angleThreshold1 = (float)(60.0f * M_PI / 180.0f);
    s.neighborThreshold = 0.5f; s.neighborThreshold1 = 1.0f; s.neighborThreshold2 = 1.0f;
    /* volatile: evaluated by libm at run time, never folded by the compiler */
    volatile float a0 = s.angleThreshold0, a1 = s.angleThreshold1;
    volatile double amin = (double)c.maxAngleThreshold, amax = (double)s.angleThreshold1;
    volatile float typo = (float)(120.0f / M_PI * 180.0f);
    volatile double d120 = 120.0f * M_PI / 180.0f, d10 = 10.0f * M_PI / 180.0f;
    s.cosAngle0 = cosf(a0);
    s.cosAngle1 = cosf(a1);
    s.cosMinAngle = (float)cos(amin); /* angle > minAngle  <=>  dot < cos(minAngle) */
    s.cosMaxAngle = (float)cos(amax); /* angle < maxAngle  <=>  dot > cos(maxAngle) */
    s.cosNeighborTypo = cosf(typo);
    s.cosNeighbor120 = (float)cos(d120);
    s.sortThreshold = (float)(1.0f - cos(d10));
    s.ascaleConst = (float)(M_PI / 48.0f);
    s.inv_sz = 1.0f / (float)(c.wsize * c.wsize);
    s.inv_3sz = 1.0f / (float)(3 * c.wsize * c.wsize);
    s.depth = c.depth;
}

} // namespace

/* =================================================================== C API */
struct orc_scene { Scene s; };

extern "C" {

void orc_default_config(orc_config* c) { /* Option::Option, option.cpp:19-33 */
    memset(c, 0, sizeof *c);
    c->nviews = 0; c->level = 1; c->csize = 2; c->wsize = 7; c->minImageNum = 3; c->max_propag = 2;
    c->nccThreshold = 0.7f; c->maxAngleThreshold = (float)(10.0f * M_PI / 180.0f); c->quadThreshold = 2.5f;
    c->depth = 1; c->seed = 1; c->schedule = ORC_SCHEDULE_ENGINE; c->sum_mode = ORC_SUM_TREE64;
    c->refine_steps = 6; c->refine_rd0 = 4.0f; c->refine_ra0 = 4.0f; c->enable_check = 1;
    c->view_begin = 0; c->view_stride = 1; c->nthreads = 1;
}

const char* orc_last_error(void) { return g_err.c_str(); }

orc_scene* orc_create(const orc_config* cfg) {
    if (!cfg || cfg->nviews < 1 || cfg->nviews > 255 || cfg->wsize < 1 || cfg->wsize > 8 || cfg->csize < 1 ||
        cfg->level < 0 || cfg->level > 4 || cfg->max_propag < 1) { g_err = "orc_create: bad config"; return nullptr; }
    orc_scene* h = new orc_scene();
    h->s.cfg = *cfg;
    derive_thresholds(h->s);
    h->s.views.resize(cfg->nviews);
    memset(&h->s.cnt, 0, sizeof h->s.cnt);
    return h;
}
void orc_destroy(orc_scene* h) { delete h; }

int orc_set_view(orc_scene* h, int v, int W, int H, const float* P, const uint8_t* rgb, const uint8_t* mask) {
    Scene& s = h->s;
    if (v < 0 || v >= s.cfg.nviews || W < 8 || H < 8 || !P || !rgb) { g_err = "orc_set_view: bad argument"; return -1; }
    View& vw = s.views[v];
    vw.W.assign(s.maxLevel, 0); vw.H.assign(s.maxLevel, 0);
    vw.W[0] = W; vw.H[0] = H;
    for (int l = 1; l < s.maxLevel; ++l) { vw.W[l] = vw.W[l - 1] / 2; vw.H[l] = vw.H[l - 1] / 2; }
    vw.img.assign(s.maxLevel, {});
    vw.img[0].assign(rgb, rgb + (size_t)W * H * 3);
    vw.mask.assign(s.maxLevel, {});
    if (mask) {
        vw.mask[0].assign(mask, mask + (size_t)W * H);
        for (auto& m : vw.mask[0]) m = m > 127 ? 255 : 0; /* image.cpp:170-177 */
    }
    setup_camera(s, vw, P);
    build_image_pyramid(s, vw);
    if (mask) build_mask_pyramid(s, vw);
    vw.gh = (vw.H[s.cfg.level] + s.cfg.csize - 1) / s.cfg.csize;
    vw.gw = (vw.W[s.cfg.level] + s.cfg.csize - 1) / s.cfg.csize;
    vw.set = true;
    return 0;
}

int orc_finalize_views(orc_scene* h) {
    Scene& s = h->s;
    for (const View& vw : s.views) if (!vw.set) { g_err = "orc_finalize_views: view not set"; return -1; }
    s.pgrids.assign(s.cfg.nviews, {}); s.vpgrids.assign(s.cfg.nviews, {}); s.dpgrids.assign(s.cfg.nviews, {});
    for (int v = 0; v < s.cfg.nviews; ++v) {
        const size_t n = (size_t)s.views[v].gw * s.views[v].gh;
        s.pgrids[v].assign(n, {}); s.vpgrids[v].assign(n, {}); s.dpgrids[v].assign(n, -1);
    }
    s.finalized = true;
    return 0;
}

int orc_get_pyramid(orc_scene* h, int v, int level, uint8_t* out, int* W, int* H) {
    Scene& s = h->s;
    if (v < 0 || v >= s.cfg.nviews || level < 0 || level >= s.maxLevel) return -1;
    if (W) *W = s.views[v].W[level];
    if (H) *H = s.views[v].H[level];
    if (out) memcpy(out, s.views[v].img[level].data(), s.views[v].img[level].size());
    return 0;
}
int orc_get_camera(orc_scene* h, int v, float* c, float* o, float* x, float* y, float* z, float* ip) {
    const View& vw = h->s.views[v];
    if (c) { c[0] = vw.center.x; c[1] = vw.center.y; c[2] = vw.center.z; c[3] = vw.center.w; }
    if (o) { o[0] = vw.oaxis.x; o[1] = vw.oaxis.y; o[2] = vw.oaxis.z; o[3] = vw.oaxis.w; }
    if (x) { x[0] = vw.xaxis.x; x[1] = vw.xaxis.y; x[2] = vw.xaxis.z; }
    if (y) { y[0] = vw.yaxis.x; y[1] = vw.yaxis.y; y[2] = vw.yaxis.z; }
    if (z) { z[0] = vw.zaxis.x; z[1] = vw.zaxis.y; z[2] = vw.zaxis.z; }
    if (ip) *ip = vw.ipscale;
    return 0;
}
int orc_grid_dims(orc_scene* h, int v, int* gw, int* gh) { *gw = h->s.views[v].gw; *gh = h->s.views[v].gh; return 0; }

int orc_set_thresholds(orc_scene* h, float ncc, float before, int depth) {
    h->s.nccThreshold = ncc; h->s.nccThresholdBefore = before; h->s.depth = depth; return 0;
}
int orc_get_thresholds(orc_scene* h, float* ncc, float* before, int* depth) {
    *ncc = h->s.nccThreshold; *before = h->s.nccThresholdBefore; *depth = h->s.depth; return 0;
}
int orc_update_threshold(orc_scene* h) { /* pmmvps.cpp:70-74 and ++m_depth at :105 */
    h->s.nccThreshold -= 0.05f; h->s.nccThresholdBefore -= 0.05f; ++h->s.depth; return 0;
}

int orc_add_patches(orc_scene* h, int n, const orc_patch* recs) { /* readPatches, patch_manager.cpp:450-463 */
    Scene& s = h->s;
    if (!s.finalized) { g_err = "orc_add_patches: views not finalized"; return -1; }
    for (int k = 0; k < n; ++k) {
        Patch p;
        from_rec(s, recs[k], p);
        if (p.nimg == 0) continue;
        p.tmp = score2(p, s.nccThreshold);
        p.nvimg = 0;
        set_grids(s, p);
        s.pool.push_back(p);
        if (s.cfg.schedule == ORC_SCHEDULE_FAITHFUL) {
            const int saved = s.depth;
            s.depth = 0; /* seeds are added while m_depth == 0 (pmmvps.cpp:84-85) */
            add_patch_live(s, (int)s.pool.size() - 1);
            s.depth = saved;
        }
    }
    return 0;
}
int orc_num_patches(orc_scene* h) { int n = 0; for (const Patch& p : h->s.pool) n += p.alive; return n; }
int orc_get_patches(orc_scene* h, int cap, orc_patch* out) {
    int n = 0;
    for (size_t id = 0; id < h->s.pool.size() && n < cap; ++id) if (h->s.pool[id].alive) to_rec(h->s.pool[id], (int)id, out[n++]);
    return n;
}
int orc_clear_patches(orc_scene* h) {
    h->s.pool.clear(); h->s.staged_cells.clear();
    if (h->s.finalized) orc_finalize_views(h);
    return 0;
}
int64_t orc_list_truncations(orc_scene* h) { return h->s.list_truncations.load(); }
int orc_list_storage(void) { return MAXI; }
/* Filter::run with both labellings of its last stage compared on the pool that stage meets: out[0] = alive patches there, out[1] =
 * removed by the connected components of the symmetrised relation (ENGINE schedule, the GPU), out[2] = removed by the reference's
 * breadth-first labelling in patch order (filter.cpp:432-524), out[3] = removed by both; the schedule's own labelling is applied */
int orc_small_groups_compare(orc_scene* h, int64_t* out) {
    Scene& s = h->s;
    /* Filter::run up to its fourth stage (filter.cpp:25-45), then both labellings side by side, then the schedule's own */
    int64_t removed4[4];
    filter_rebuild(s, 0);
    removed4[0] = filter_outside(s);
    filter_rebuild(s, 1);
    removed4[1] = filter_exact(s);
    filter_rebuild(s, 1);
    removed4[2] = filter_neighbor(s);
    filter_rebuild(s, 1);
    std::vector<int> comp, lit;
    filter_small_groups(s, 2, &comp);
    filter_small_groups(s, 1, &lit);
    std::sort(comp.begin(), comp.end()); std::sort(lit.begin(), lit.end());
    std::vector<int> both;
    std::set_intersection(comp.begin(), comp.end(), lit.begin(), lit.end(), std::back_inserter(both));
    int64_t alive = 0;
    for (const Patch& p : s.pool) alive += p.alive ? 1 : 0;
    out[0] = alive; out[1] = (int64_t)comp.size(); out[2] = (int64_t)lit.size(); out[3] = (int64_t)both.size();
    removed4[3] = filter_small_groups(s);
    (void)removed4;
    filter_rebuild(s, 1);
    return 0;
}
/* Filter::filterSmallGroups' relation on the current pool (after a filter_rebuild): out[0] = alive patches, out[1] = directed edges
 * p -> q (q != p), out[2] = those without the reverse edge q -> p */
int orc_group_edge_stats(orc_scene* h, int64_t* out) {
    Scene& s = h->s;
    filter_rebuild(s, 1);
    std::vector<std::vector<int>> adj(s.pool.size());
    int64_t alive = 0, edges = 0, oneway = 0;
    for (size_t id = 0; id < s.pool.size(); ++id) {
        if (!s.pool[id].alive) continue;
        ++alive;
        small_group_edges(s, (int)id, adj[id]);
        std::sort(adj[id].begin(), adj[id].end());
        adj[id].erase(std::unique(adj[id].begin(), adj[id].end()), adj[id].end());
    }
    for (size_t id = 0; id < s.pool.size(); ++id)
        for (int q : adj[id]) {
            if (q == (int)id) continue;
            ++edges;
            if (!std::binary_search(adj[q].begin(), adj[q].end(), (int)id)) ++oneway;
        }
    out[0] = alive; out[1] = edges; out[2] = oneway;
    return 0;
}
int orc_patch_bytes(void) { return (int)sizeof(orc_patch); }
int orc_set_cell_budget(orc_scene* h, int64_t n) { h->s.cell_budget = n; return 0; }
int orc_set_time_budget(orc_scene* h, double seconds) { h->s.time_budget = seconds; return 0; }
double orc_last_sweep_seconds(orc_scene* h) { return h->s.last_sweep_seconds; }

int orc_engine_pass(orc_scene* h, int iter, int pass, orc_counters* out) {
    Scene& s = h->s;
    memset(&s.cnt, 0, sizeof s.cnt);
    engine_pass(s, iter, pass);
    if (out) *out = s.cnt;
    return 0;
}
int orc_export_new(orc_scene* h, int cap, orc_patch* out, int32_t* per_view) {
    Scene& s = h->s;
    if (per_view) for (int v = 0; v < s.cfg.nviews; ++v) per_view[v] = 0;
    int n = 0;
    for (const DestCtx& ctx : s.staged_cells) for (const Patch& p : ctx.staged) {
        if (!p.alive) continue;
        if (out && n < cap) { to_rec(p, p.dest_cell, out[n]); out[n].flags = 1 | (p.sweep_view << 8); }
        if (per_view) per_view[p.sweep_view]++;
        ++n;
    }
    return n;
}
int orc_export_kills(orc_scene* h, int cap, int32_t* ids) {
    int n = 0;
    for (const DestCtx& ctx : h->s.staged_cells) for (int id : ctx.kills) { if (ids && n < cap) ids[n] = id; ++n; }
    return n;
}
int orc_commit(orc_scene* h, int n_new, const orc_patch* recs, int n_kill, const int32_t* kill_ids) {
    Scene& s = h->s;
    for (int k = 0; k < n_kill; ++k) if (kill_ids[k] >= 0 && kill_ids[k] < (int)s.pool.size()) s.pool[kill_ids[k]].alive = false;
    for (int k = 0; k < n_new; ++k) {
        Patch p;
        from_rec(s, recs[k], p);
        set_grids(s, p); set_vgrids(s, p);
        s.pool.push_back(p);
    }
    s.staged_cells.clear();
    return 0;
}

int orc_propagate(orc_scene* h, int iter, orc_counters* out) { /* Propagate::run, propagate.cpp:28-64 */
    Scene& s = h->s;
    if (!s.finalized) { g_err = "orc_propagate: views not finalized"; return -1; }
    orc_counters total;
    memset(&total, 0, sizeof total);
    if (s.cfg.schedule == ORC_SCHEDULE_FAITHFUL) {
        memset(&s.cnt, 0, sizeof s.cnt);
        propagate_faithful(s, iter);
        total = s.cnt;
    } else {
        for (int pass = 0; pass < 2; ++pass) {
            memset(&s.cnt, 0, sizeof s.cnt);
            engine_pass(s, iter, pass);
            add_counters(total, s.cnt);
            engine_commit_local(s);
        }
        for (Patch& p : s.pool) if (p.alive) { set_grids(s, p); set_vgrids(s, p); }
    }
    if (out) *out = total;
    return 0;
}

int orc_filter(orc_scene* h, int64_t* removed4) { /* Filter::run, filter.cpp:25-49 */
    int64_t r[4] = {0, 0, 0, 0};
    filter_run(h->s, r);
    if (removed4) for (int k = 0; k < 4; ++k) removed4[k] = r[k];
    return 0;
}

int orc_depth_normal_map(orc_scene* h, int view, int kind, float* depth, float* normal, int32_t* ids) {
    Scene& s = h->s;
    if (view < 0 || view >= s.cfg.nviews) return -1;
    const View& vw = s.views[view];
    const size_t n = (size_t)vw.gw * vw.gh;
    std::vector<int> sel(n, -1);
    if (kind == 0) {
        if (s.cfg.schedule == ORC_SCHEDULE_ENGINE) build_depth_maps(s);
        else { /* rebuild as Filter::setDepthMaps does, so removed patches drop out */
            build_depth_maps(s);
        }
        sel = s.dpgrids[view];
    } else {
        for (size_t id = 0; id < s.pool.size(); ++id) {
            const Patch& p = s.pool[id];
            if (!p.alive || p.nimg == 0 || p.img[0] != view) continue;
            int ix, iy;
            cell_of(s, view, p.coord, ix, iy);
            if (ix < 0 || vw.gw <= ix || iy < 0 || vw.gh <= iy) continue;
            int& cur = sel[(size_t)iy * vw.gw + ix];
            if (cur < 0 || p.ncc > s.pool[cur].ncc) cur = (int)id;
        }
    }
    for (size_t c = 0; c < n; ++c) {
        if (sel[c] < 0) {
            if (depth) depth[c] = NAN;
            if (normal) normal[3 * c] = normal[3 * c + 1] = normal[3 * c + 2] = NAN;
        } else {
            const Patch& p = s.pool[sel[c]];
            if (depth) depth[c] = dot4(vw.oaxis, p.coord);
            if (normal) { normal[3 * c] = p.normal.x; normal[3 * c + 1] = p.normal.y; normal[3 * c + 2] = p.normal.z; }
        }
        if (ids) ids[c] = sel[c];
    }
    return 0;
}

/* ---------------------------------------------------------------- probes */
static V4 v4(const float* a) { return {a[0], a[1], a[2], a[3]}; }
static void prep_probe(Scene& s, const orc_patch* r, Patch& p) { from_rec(s, *r, p); set_grids(s, p); set_vgrids(s, p); }

int orc_project(orc_scene* h, int v, const float* c, int level, float* ic) {
    const V3 r = project(h->s.views[v], v4(c), level); ic[0] = r.x; ic[1] = r.y; ic[2] = r.z; return 0;
}
int orc_unproject(orc_scene* h, int v, const float* ic, int level, float* c) {
    const V4 r = unproject(h->s.views[v], {ic[0], ic[1], ic[2]}, level); c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w; return 0;
}
float orc_get_unit(orc_scene* h, int v, const float* c) { return get_unit(h->s, v, v4(c)); }
int orc_get_paxes(orc_scene* h, int v, const float* c, const float* n, float* px, float* py) {
    V4 a, b; get_paxes(h->s, v, v4(c), v4(n), a, b);
    px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w; py[0] = b.x; py[1] = b.y; py[2] = b.z; py[3] = b.w; return 0;
}
int orc_get_color(orc_scene* h, int v, float x, float y, int level, float* rgb) { get_color(h->s.views[v], x, y, level, rgb); return 0; }
int orc_get_tex(orc_scene* h, const float* c, const float* px, const float* py, const float* n, int v, float* out, int normalize) {
    Tex t;
    const int flag = get_tex(h->s, v4(c), v4(px), v4(py), v4(n), v, t, nullptr);
    if (flag != 0) return flag;
    if (normalize) normalize_tex(h->s, t);
    const int sz = h->s.cfg.wsize * h->s.cfg.wsize;
    const float sc = normalize ? t.inv : 1.0f;
    for (int i = 0; i < sz; ++i) for (int ch = 0; ch < 3; ++ch) out[3 * i + ch] = t.c[ch][i] * sc;
    return 0;
}
float orc_compute_incc(orc_scene* h, const orc_patch* r, int robust) {
    Patch p; prep_probe(h->s, r, p);
    float w[MAXI]; compute_weights(h->s, p.coord, p.normal, p.img, p.nimg, w);
    return compute_incc(h->s, p.coord, p.normal, p.img, p.nimg, w, robust, nullptr);
}
float orc_compute_ncc(orc_scene* h, const orc_patch* r) { Patch p; prep_probe(h->s, r, p); return compute_ncc(h->s, p, nullptr); }
int orc_set_inccs(orc_scene* h, const orc_patch* r, int robust, float* inccs) {
    Patch p; prep_probe(h->s, r, p); set_inccs(h->s, p, p.img, p.nimg, robust, inccs, nullptr); return p.nimg;
}
int orc_set_inccs_matrix(orc_scene* h, const orc_patch* r, int robust, float* inccs) {
    Patch p; prep_probe(h->s, r, p); set_inccs_matrix(h->s, p, p.img, p.nimg, robust, inccs, nullptr); return p.nimg;
}
int orc_preprocess(orc_scene* h, orc_patch* r) {
    Patch p; prep_probe(h->s, r, p);
    const int f = pre_process(h->s, p, nullptr); to_rec(p, r->id, *r); return f;
}
int orc_refine(orc_scene* h, orc_patch* r, const uint32_t* key4) {
    Patch p; prep_probe(h->s, r, p);
    const int f = refine_patch(h->s, p, key4, nullptr); to_rec(p, r->id, *r); return f;
}
int orc_postprocess(orc_scene* h, orc_patch* r) {
    Scene& s = h->s;
    Patch p; prep_probe(s, r, p);
    if (s.cfg.schedule == ORC_SCHEDULE_ENGINE) engine_prepare(s);
    const int f = post_process(s, p, nullptr, nullptr); to_rec(p, r->id, *r); return f;
}
static void probe_rc(Scene& s, const Patch& p, RefineCtx& rc) {
    rc.center = p.coord; rc.ref = p.img[0];
    rc.ray = sub4(p.coord, s.views[rc.ref].center); rc.ray = nrm4(rc.ray);
    rc.dscale = p.dscale; rc.ascale = s.ascaleConst;
}
double orc_cost(orc_scene* h, const orc_patch* r, const float* x3) {
    Patch p; prep_probe(h->s, r, p); RefineCtx rc; probe_rc(h->s, p, rc);
    return cost_func(h->s, rc, p.img, p.nimg, x3, nullptr);
}
int orc_encode(orc_scene* h, const orc_patch* r, float* x3) {
    Patch p; prep_probe(h->s, r, p); RefineCtx rc; probe_rc(h->s, p, rc); encode(h->s, rc, p.coord, p.normal, x3); return 0;
}
int orc_decode(orc_scene* h, const orc_patch* r, const float* x3, float* c, float* n) {
    Patch p; prep_probe(h->s, r, p); RefineCtx rc; probe_rc(h->s, p, rc);
    V4 cc, nn; decode(h->s, rc, x3, cc, nn);
    c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w; n[0] = nn.x; n[1] = nn.y; n[2] = nn.z; n[3] = nn.w; return 0;
}
int orc_generate_patch(orc_scene* h, const orc_patch* src, const float* ic, orc_patch* out) {
    Patch p, q; prep_probe(h->s, src, p);
    if (!generate_patch(h->s, p, {ic[0], ic[1], ic[2]}, q, nullptr)) return -1;
    to_rec(q, -1, *out); return 0;
}
float orc_quad_residual(orc_scene* h, const orc_patch* r, const float* coords4, int n) {
    Patch p; prep_probe(h->s, r, p);
    std::vector<V4> nc((size_t)n);
    for (int i = 0; i < n; ++i) nc[(size_t)i] = {coords4[4 * i], coords4[4 * i + 1], coords4[4 * i + 2], coords4[4 * i + 3]};
    return quad_residual(h->s, p, nc);
}
float orc_robustincc(float v) { return robustincc(v); }
float orc_unrobustincc(float v) { return unrobustincc(v); }
void orc_minstd_draws(int n, float* out) {
    std::minstd_rand0 g; std::uniform_real_distribution<float> d(-0.5, 0.5);
    for (int i = 0; i < n; ++i) out[i] = d(g);
}
float orc_rng_uniform(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) { return rng_uniform(seed, a, b, c, d, e); }
float orc_sinf(float x) { return pm_sinf(x); }
float orc_cosf(float x) { return pm_cosf(x); }
float orc_asinf(float x) { return pm_asinf(x); }
float orc_acosf(float x) { return pm_acosf(x); }
float orc_atanf(float x) { return pm_atanf(x); }
int orc_is_neighbor(orc_scene* h, const orc_patch* a, const orc_patch* b, float thr) {
    Patch p, q; prep_probe(h->s, a, p); prep_probe(h->s, b, q); return is_neighbor(h->s, p, q, thr);
}
float orc_compute_gain(orc_scene* h, const orc_patch* r) {
    Scene& s = h->s; Patch p; prep_probe(s, r, p);
    if (s.cfg.schedule == ORC_SCHEDULE_ENGINE) engine_prepare(s);
    return compute_gain(s, p, nullptr);
}
int orc_check(orc_scene* h, orc_patch* r) {
    Scene& s = h->s; Patch p; prep_probe(s, r, p);
    if (s.cfg.schedule == ORC_SCHEDULE_ENGINE) engine_prepare(s);
    const int f = check_patch(s, p, nullptr); to_rec(p, r->id, *r); return f;
}

} // extern "C"
