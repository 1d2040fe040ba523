// mvs_device.cuh -- device functions of the PatchMatch-MVS engine for gfx950 (wave64).
//
// Execution model: ONE 64-lane wavefront works on one patch / one destination cell.  Control flow is
// wave-uniform.  Three lane layouts coexist in registers:
//   * class lanes: lane 16 r + c holds samples c, c + 16, c + 32 of the 7x7 texture window (optim.cpp:835-842) its 16-lane
//     row works on -- a proposal in a refinement step, a view in a single evaluation; sums over a window are lane sums, a
//     DPP row tree (pairing 1, 2, 4, 8) and the window's 49th sample, taken by the lane that owns the sampling frame;
//   * view lanes: lane j holds element j of a per-view array (Patch::m_images[j], its ray, unit, INCC ...),
//     read back with v_readlane when a loop needs element j uniformly;
//   * frame lanes: lane 16*g + i holds the sampling frame of view i for proposal g (g < 4): the four
//     proposals of one refinement step share one pass of decode / getPAxes / projection arithmetic.
// Per-view scalars that need a square root or a division (msd, 1/msd, robust INCC) are gathered into view
// lanes first, so one vector instruction sequence serves all views of an evaluation.
// Arithmetic follows DESIGN.md "engine arithmetic": fp32, no implicit contraction (-ffp-contract=off), dot
// products as left-to-right fmaf chains, own polynomial sin/cos/asin/acos/atan -- the same operation order
// the CPU oracle's TREE64 mode uses, so results can be compared bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include "mvs_types.h"

namespace mvsdev {

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

#define DEV __device__ __forceinline__
// The stages around the refinement loop can be compiled as real calls (MVS_OUTLINE=1): their register live ranges
// then do not interfere with the hot loop.
#ifndef MVS_OUTLINE
#define MVS_OUTLINE 0
#endif
// how far the texel loads run ahead of their use (1 = the next round / view only)
#ifndef MVS_EV_DEPTH
#define MVS_EV_DEPTH 1  // rounds of a single evaluation whose loads are in flight beyond the current one
#endif
#ifndef MVS_ST_DEPTH
#define MVS_ST_DEPTH 1  // views of a refinement step whose loads are in flight beyond the current one
#endif
#if MVS_OUTLINE
#define STAGE __device__ __noinline__
#else
#define STAGE __device__ __forceinline__
#endif

DEV int lane_id() { return (int)(threadIdx.x & 63u); }
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV int rli(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
DEV float rlf(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
DEV int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
DEV unsigned long long ballot(bool p) { return __ballot(p); }
// one bit per entry of a view list: 32 bits serve the 16- and 32-view builds, the 64-view build needs all 64 lanes
#if MVS_LISTCAP > 32
typedef unsigned long long vmask_t;
DEV int vpop(vmask_t m) { return __popcll(m); }
#define MVS_VBITS 64
#else
typedef unsigned vmask_t;
DEV int vpop(vmask_t m) { return __popc(m); }
#define MVS_VBITS 32
#endif
DEV vmask_t vballot(bool p) { return (vmask_t)__ballot(p); }

DEV float dot4(F4 a, F4 b) { return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x))); }
DEV float dot3(F3 a, F3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// Correctly rounded square root (== the oracle's sqrtf) for zero, normal and infinite arguments: v_sqrt_f32 is within one
// ulp, and the residuals of its two neighbours (one fma each) decide which of the three is the rounded root.  This is
// the compiler's own IEEE expansion without the rescaling of arguments below 2^-96, which this path never produces
// (squared lengths and texture variances), and at half its instruction count.
DEV float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rd = __builtin_fmaf(-dn, s, x), ru = __builtin_fmaf(-up, s, x);
    s = rd <= 0.0f ? dn : s;
    s = ru > 0.0f ? up : s;
    return s;
}
// Correctly rounded reciprocal and quotient in 3 and 6 instructions where the compiler's IEEE expansion takes 11 (v_div_scale x 2,
// v_rcp, four fma, v_div_fmas, v_div_fixup): v_rcp_f32 is within one ulp, one Newton step in fma lands on RN(1 / x); the quotient is
// Markstein's q' = RN(q + (a - b q) y) with y = RN(1 / b).  tools/microbench/div_exact.hip runs rcp_rn against `1.0f / x` for EVERY
// float with 2^-120 <= |x| < 2^121 (4.04 * 10^9 inputs: no difference) and div_rn against `a / b` on 1.18 * 10^10 pairs (none) --
// profiles/r04_div_exact.txt.  Outside that range (zero, subnormal, infinite or NaN divisors, quotients that over- or underflow) they
// differ from IEEE division; they are used only where the divisor is a length, a depth, a focal scale or 1 + 3 incc of this path.
#ifndef MVS_FASTDIV
#define MVS_FASTDIV 1
#endif
DEV float rcp_rn(float x) {
#if MVS_FASTDIV
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#else
    return 1.0f / x;
#endif
}
DEV float div_rn(float a, float b) {
#if MVS_FASTDIV
    const float y = rcp_rn(b), q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
#else
    return a / b;
#endif
}
DEV float norm4(F4 a) { return sqrt_rn(dot4(a, a)); }
DEV float norm3(F3 a) { return sqrt_rn(dot3(a, a)); }
DEV F4 sub4(F4 a, F4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
DEV F4 add4(F4 a, F4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DEV F4 mul4(F4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
DEV F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// v / |v| as v * (1/|v|) (Eigen 3.2 vector / scalar semantics; one IEEE division instead of three or four)
DEV F4 nrm4(F4 a) { const float inv = rcp_rn(norm4(a)); return {a.x * inv, a.y * inv, a.z * inv, a.w * inv}; }
DEV F3 nrm3(F3 a) { const float inv = rcp_rn(norm3(a)); return {a.x * inv, a.y * inv, a.z * inv}; }
DEV F4 scl4(F4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
DEV F3 scl3(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DEV F3 cross3(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV F4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
DEV F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// ------------------------------------------------------------------ wave butterflies
// sum over the 64 lanes, pairing order 1,2,4,8,16,32; every lane ends with the same bits.
template <int CTRL> DEV float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
DEV float wave_sum(float x) {
    x = x + dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]   : i ^ 1
    x = x + dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]   : i ^ 2
    x = x + dpp_f<0x141>(x);  // row_half_mirror       : other quad of the 8 (== i ^ 4 once quads are uniform)
    x = x + dpp_f<0x140>(x);  // row_mirror            : other half of the row of 16 (== i ^ 8)
    // every row of 16 now holds its row sum r0..r3 in all of its lanes.  i ^ 16 pairs rows (0,1) and (2,3), i ^ 32 the
    // halves: (r0 + r1) + (r2 + r3).  row_bcast15 into rows 1 and 3 gives r1 + r0 and r3 + r2, row_bcast31 into rows 2,3
    // gives (r3 + r2) + (r1 + r0) in row 3 -- the same sums (fp addition commutes), 3 instructions, no LDS pipeline.
    // Only lane 63 is read, so the broadcasts may run on every row (rows without a source lane add 0: bound_ctrl), which
    // lets each step be a single v_add_f32_dpp.
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xf, 0xf, true));  // row_bcast:15
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x143, 0xf, 0xf, true));  // row_bcast:31
    return rlf(x, 63);
}
// (bound_ctrl form of a DPP read: lanes without a source read 0)
template <int CTRL> DEV float dpp0_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
DEV float wave_min(float x) {
    x = fminf(x, dpp_f<0xB1>(x));
    x = fminf(x, dpp_f<0x4E>(x));
    x = fminf(x, dpp_f<0x141>(x));
    x = fminf(x, dpp_f<0x140>(x));
    return fminf(fminf(rlf(x, 0), rlf(x, 16)), fminf(rlf(x, 32), rlf(x, 48)));
}

// ------------------------------------------------------------------ deterministic libm subset
// (same kernels, constants and evaluation order as the oracle's pm_* functions)
#define MVS_PIO2_HI 1.57079625129699707031f
#define MVS_PIO2_LO 7.54978941586159635335e-08f
#define MVS_PIO4 0.78539816339744830962f
#define MVS_PI 3.14159265358979323846f
DEV float k_sinf(float x) {
    float z = x * x;
    float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
    p = p * z - 1.6666654611e-1f;
    return x + x * z * p;
}
DEV float k_cosf(float x) {
    float z = x * x;
    float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + z * z * p;
}
DEV float pm_sinf(float x) {
    float a = fabsf(x), r;
    if (a <= MVS_PIO4) r = k_sinf(a);
    else if (a <= 3.0f * MVS_PIO4) r = k_cosf((MVS_PIO2_HI - a) + MVS_PIO2_LO);
    else r = k_sinf(MVS_PI - a);
    return x < 0.0f ? -r : r;
}
DEV float pm_cosf(float x) {
    float a = fabsf(x);
    if (a <= MVS_PIO4) return k_cosf(a);
    if (a <= 3.0f * MVS_PIO4) return k_sinf((MVS_PIO2_HI - a) + MVS_PIO2_LO);
    return -k_cosf(MVS_PI - a);
}
// sin and cos of one angle: the same three ranges, kernels and reduced arguments as pm_sinf / pm_cosf, evaluated once
// and selected (lanes of one wave usually fall in different ranges, so the branches of the two calls would all run)
DEV void pm_sincosf(float x, float& s, float& c) {
    const float a = fabsf(x);
    const bool r0 = a <= MVS_PIO4, r1 = a <= 3.0f * MVS_PIO4;
    const float t = r0 ? a : (r1 ? (MVS_PIO2_HI - a) + MVS_PIO2_LO : MVS_PI - a);
    const float ks = k_sinf(t), kc = k_cosf(t);
    const float sr = (r0 || !r1) ? ks : kc;
    s = x < 0.0f ? -sr : sr;
    c = r0 ? kc : (r1 ? ks : -kc);
}
DEV float pm_asinf(float x) {
    float a = fabsf(x);
    if (a > 1.0f) a = 1.0f;
    float z, xx;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); xx = sqrt_rn(z); }
    else { z = a * a; xx = a; }
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    float r = p * z * xx + xx;
    if (flag) { r = r + r; r = (MVS_PIO2_HI - r) + MVS_PIO2_LO; }
    return x < 0.0f ? -r : r;
}
DEV float pm_acosf(float x) {
    if (x < -0.5f) return MVS_PI - 2.0f * pm_asinf(sqrt_rn(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * pm_asinf(sqrt_rn(0.5f * (1.0f - x)));
    return MVS_PIO2_HI - pm_asinf(x);
}
DEV float pm_atanf(float v) {
    float x = fabsf(v), y;
    if (x > 2.414213562373095f) { y = MVS_PIO2_HI; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = MVS_PIO4; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    y = y + (p * z * x + x);
    return v < 0.0f ? -y : y;
}

// ------------------------------------------------------------------ counter-based RNG
DEV uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
DEV float rng_uniform(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
    uint32_t h = mix32(seed ^ 0x9e3779b9u);
    h = mix32(h ^ a) + 0x85ebca6bu;
    h = mix32(h ^ b) + 0xc2b2ae35u;
    h = mix32(h ^ c) + 0x27d4eb2fu;
    h = mix32(h ^ d) + 0x165667b1u;
    h = mix32(h ^ e);
    return (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

// ------------------------------------------------------------------ camera (image/camera.cpp)
// Camera::project, camera.cpp:310-326
DEV F3 project(const DView* vw, F4 X, int level) {
    const float* P = vw->P[level];
    float r0 = fma_(P[3], X.w, fma_(P[2], X.z, fma_(P[1], X.y, P[0] * X.x)));
    float r1 = fma_(P[7], X.w, fma_(P[6], X.z, fma_(P[5], X.y, P[4] * X.x)));
    float r2 = fma_(P[11], X.w, fma_(P[10], X.z, fma_(P[9], X.y, P[8] * X.x)));
    if (r2 <= 0.0f) return {-65535.0f, -65535.0f, -1.0f};
    const float lo = (float)(INT_MIN + 3.0f), hi = (float)(INT_MAX - 3.0f);
    const float inv = rcp_rn(r2);
    F3 ic{r0 * inv, r1 * inv, 1.0f};
    ic.x = fmaxf(lo, fminf(hi, ic.x));
    ic.y = fmaxf(lo, fminf(hi, ic.y));
    return ic;
}
// Camera::project on a projection matrix held in registers, without a branch (same values as project())
DEV F3 project_regs(const float (&P)[12], F4 X) {
    const float r0 = fma_(P[3], X.w, fma_(P[2], X.z, fma_(P[1], X.y, P[0] * X.x)));
    const float r1 = fma_(P[7], X.w, fma_(P[6], X.z, fma_(P[5], X.y, P[4] * X.x)));
    const float r2 = fma_(P[11], X.w, fma_(P[10], X.z, fma_(P[9], X.y, P[8] * X.x)));
    const float lo = (float)(INT_MIN + 3.0f), hi = (float)(INT_MAX - 3.0f);
    const float inv = rcp_rn(r2);  // r2 <= 0 is replaced below whatever this gave
    F3 ic{fmaxf(lo, fminf(hi, r0 * inv)), fmaxf(lo, fminf(hi, r1 * inv)), 1.0f};
    if (r2 <= 0.0f) ic = {-65535.0f, -65535.0f, -1.0f};
    return ic;
}
DEV void load_P(const DView* vw, int level, float (&P)[12]) {
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = vw->P[level][k];
}
// Camera::unproject at m_level, camera.cpp:329-337
DEV F4 unproject(const DView* vw, F3 ic, int level) {
    const float* P = vw->P[level];
    const float* M = vw->Minv;
    F3 b{ic.x - P[3], ic.y - P[7], ic.z - P[11]};
    return {fma_(M[2], b.z, fma_(M[1], b.y, M[0] * b.x)), fma_(M[5], b.z, fma_(M[4], b.y, M[3] * b.x)),
            fma_(M[8], b.z, fma_(M[7], b.y, M[6] * b.x)), 1.0f};
}
// Optim::getUnit, optim.cpp:34-41 (2 * fz * 2^level is exact, so one fp32 division; see the oracle)
DEV float get_unit(const DParams& prm, const DView* vw, F4 coord) {
    const float fz = norm4(sub4(coord, ld4(vw->center)));
    const float ips = vw->ipscale;
    if (ips == 0.0f) return 1.0f;
    return div_rn(2.0f * fz * (float)(1 << prm.level), ips);
}
// PatchManager::setGrids cell rule, patch_manager.cpp:241-250
DEV void cell_of(const DParams& prm, const DView* vw, F4 coord, int& ix, int& iy) {
    float P[12];
    load_P(vw, prm.level, P);
    const F3 ic = project_regs(P, coord);
    ix = ((int)floorf(ic.x + 0.5f)) / prm.csize;
    iy = ((int)floorf(ic.y + 0.5f)) / prm.csize;
}
// Optim::getPAxes, optim.cpp:67-84 (the view's constants are loaded once, up front).
// Every caller hands in a patch that is the same in all 16 lanes of a row (one patch per wave in the single evaluations, a proposal per
// row in the refinement steps, a patch per 16 / 32 / 64 lanes in Filter::filterExact).  The three projections of the reference view --
// of the patch's centre and of the centre moved along either axis -- are independent, so lanes 0, 1, 2 of every row take one each
// (the other lanes repeat the centre's), the two distances and their reciprocals come out of ONE norm / division sequence in lanes 1
// and 2, and the row reads them back with row broadcasts (DPP row_newbcast, no LDS): one projection, one square root and one division
// per step where each lane used to run three, two and two.  The point a lane projects is coord + a px + b py with (a, b) = (0, 0),
// (1, 0) or (0, 1): fma(b, py, fma(a, px, coord)) is coord, RN(coord + px) or RN(coord + py) exactly -- the reference's coord + pxaxis.
template <int LANE> DEV float row_bcast_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + LANE, 0xf, 0xf, false));  // row_newbcast:LANE
}
DEV void get_paxes(const DParams& prm, const DView* vw, F4 coord, F4 normal, F4& px, F4& py) {
    const F4 ctr = ld4(vw->center);
    const float ips = vw->ipscale;
    const F3 xax = ld3(vw->xaxis);
    float P[12];
    load_P(vw, prm.level, P);
    float pscale = 1.0f;  // get_unit
    if (ips != 0.0f) pscale = div_rn(2.0f * norm4(sub4(coord, ctr)) * (float)(1 << prm.level), ips);
    F3 n3{normal.x, normal.y, normal.z};
    F3 y3 = cross3(n3, xax);
    y3 = nrm3(y3);
    F3 x3 = cross3(y3, n3);
    px = {x3.x * pscale, x3.y * pscale, x3.z * pscale, 0.0f};
    py = {y3.x * pscale, y3.y * pscale, y3.z * pscale, 0.0f};
#ifndef MVS_PAXES_SPREAD
#define MVS_PAXES_SPREAD 1
#endif
#if MVS_PAXES_SPREAD
    const int j = lane_id() & 15;
    const float a = j == 1 ? 1.0f : 0.0f, b = j == 2 ? 1.0f : 0.0f;
    const F4 X{fma_(b, py.x, fma_(a, px.x, coord.x)), fma_(b, py.y, fma_(a, px.y, coord.y)), fma_(b, py.z, fma_(a, px.z, coord.z)), coord.w};
    const F3 ic = project_regs(P, X);
    const F3 c0{row_bcast_f<0>(ic.x), row_bcast_f<0>(ic.y), row_bcast_f<0>(ic.z)};
    const float inv = rcp_rn(norm3(sub3(ic, c0)));  // lane 1: 1 / xdis, lane 2: 1 / ydis (lane 0 and the rest: 1 / 0, never read)
    px = scl4(px, row_bcast_f<1>(inv));
    py = scl4(py, row_bcast_f<2>(inv));
#else
    const F3 c0 = project_regs(P, coord);
    const float xdis = norm3(sub3(project_regs(P, add4(coord, px)), c0));
    const float ydis = norm3(sub3(project_regs(P, add4(coord, py)), c0));
    px = scl4(px, rcp_rn(xdis));
    py = scl4(py, rcp_rn(ydis));
#endif
}
DEV float robustincc(float incc) { return div_rn(incc, 1 + 3 * incc); }
DEV float unrobustincc(float r) { return div_rn(r, 1 - 3 * r); }

// ------------------------------------------------------------------ texture frames (frame lanes)
// Head of Optim::getTex, optim.cpp:790-818: per (proposal, view) the sampling frame (top-left, dx, dy), the
// pyramid level, and the base pointer / width of that level so that the sampling loop needs no further loads.
// A rejected view keeps ok = 0 and a harmless frame (the 2x2 texels at the origin of the view's own image), so
// that sampling can stay branch-free; its results are masked out by the caller.
struct Frame {
    float tlx, tly, dxx, dxy, dyx, dyy;
    int w;               // width of the chosen level
    int ok;              // 1 = the view passed the angle gate and getTexSafe
    unsigned img_lo, img_hi;  // RGBA8 pyramid level base address
};
// levelDiff = clamp(floor(log2(ratio) + .5)) (optim.cpp:808) as comparisons against 2^(k - .5), k = -3 .. 2 ("level
// pick").  Those six thresholds are ONE float mantissa (that of sqrt 2, 0x3504F3) under six exponents, so for ratio = 2^e * 1.m the
// largest k with ratio >= 2^(k - .5) is e + (m >= 0x3504F3): adding 0x800000 - 0x3504F3 to the bits carries into the exponent exactly
// then -- three integer instructions for six compares and selects; the clamp to [-4, 2] takes care of zero, tiny and huge ratios as
// the chain of comparisons did (a NaN ratio, which only a degenerate frame that is rejected anyway can produce, read -4 there and
// reads 2 here).
#ifndef MVS_FS1
#define MVS_FS1 1
#endif
DEV int level_diff(const DParams& prm, float ratio) {
#if MVS_FS1
    static_assert(0x3504F3 == (0x3FB504F3 & 0x7FFFFF), "mantissa of 1.414213562373095f");
    const int k = ((__float_as_int(ratio) + (0x800000 - 0x3504F3)) >> 23) - 127;
    const int ld = max(-4, min(2, k));
    return max(-prm.level, min(2, ld));
#else
    int ld = -4;
    if (ratio >= 0.088388347648318f) ld = -3;
    if (ratio >= 0.176776695296637f) ld = -2;
    if (ratio >= 0.353553390593274f) ld = -1;
    if (ratio >= 0.707106781186548f) ld = 0;
    if (ratio >= 1.414213562373095f) ld = 1;
    if (ratio >= 2.828427124746190f) ld = 2;
    return max(-prm.level, min(2, ld));
#endif
}
DEV float pow2_level(int ld) { return __int_as_float((127 + ld) << 23); }  // Optim::myPow2, exact powers of two
// Straight-line: every load of the view's constants is issued at the top (one wait instead of one per early exit --
// a lane that leaves early saves nothing while its neighbours go on), the gates only select the result.
DEV Frame make_frame(const DParams& prm, F4 coord, F4 px, F4 py, F4 pz, int v, bool active) {
    const DView* vw = prm.views + (active ? v : 0);
    const F4 ctr = ld4(vw->center);
    float P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = vw->P[prm.level][k];
    const int W0 = vw->W[0], H0 = vw->H[0];
    const unsigned long long base_level = (unsigned long long)vw->img[prm.level];
    const F4 ray = nrm4(sub4(ctr, coord));
    const float weight = fmaxf(0.0f, dot4(ray, pz));
    F3 center = project_regs(P, coord);
    F3 dx = sub3(project_regs(P, add4(coord, px)), center);
    F3 dy = sub3(project_regs(P, add4(coord, py)), center);
    const float ratio = (norm3(dx) + norm3(dy)) / 2.0f;
    const int ld = level_diff(prm, ratio);
    const float iscale = pow2_level(-ld);  // exact reciprocal of a power of two
    const int newLevel = prm.level + ld;
    const unsigned long long base_new = (unsigned long long)vw->img[newLevel];  // the one load that depends on the arithmetic
    center = scl3(center, iscale);
    dx = scl3(dx, iscale);
    dy = scl3(dy, iscale);
    // Optim::getTexSafe, optim.cpp:895-915
    // The four corners are (c -+ a) -+ b with a = dx * m, b = dy * m (each product rounded once, each sum rounded): all four sign
    // combinations.  Rounding is monotone, so the smallest of the four is (c - |a|) - |b| and the largest (c + |a|) + |b| -- the very
    // values the reference's min / max over the corners pick (optim.cpp:902-912), without forming the other three corners.
    const float m = (float)(prm.wsize / 2);
#if MVS_FS1
    const float ax = dx.x * m, bx = dy.x * m, ay = dx.y * m, by = dy.y * m;
    const float tlx = (center.x - ax) - bx, tly = (center.y - ay) - by;
    const float minx = (center.x - fabsf(ax)) - fabsf(bx), maxx = (center.x + fabsf(ax)) + fabsf(bx);
    const float miny = (center.y - fabsf(ay)) - fabsf(by), maxy = (center.y + fabsf(ay)) + fabsf(by);
#else
    const float tlx = (center.x - dx.x * m) - dy.x * m, trx = (center.x + dx.x * m) - dy.x * m;
    const float blx = (center.x - dx.x * m) + dy.x * m, brx = (center.x + dx.x * m) + dy.x * m;
    const float tly = (center.y - dx.y * m) - dy.y * m, try_ = (center.y + dx.y * m) - dy.y * m;
    const float bly = (center.y - dx.y * m) + dy.y * m, bry = (center.y + dx.y * m) + dy.y * m;
    const float minx = fminf(tlx, fminf(trx, fminf(blx, brx))), maxx = fmaxf(tlx, fmaxf(trx, fmaxf(blx, brx)));
    const float miny = fminf(tly, fminf(try_, fminf(bly, bry))), maxy = fmaxf(tly, fmaxf(try_, fmaxf(bly, bry)));
#endif
    const int margin2 = 2;
    const int W = W0 >> newLevel, H = H0 >> newLevel;  // the pyramid halves (rounding down) at every level
    const bool inside = !(minx < margin2 || W - 1 - margin2 <= maxx || miny < margin2 || H - 1 - margin2 <= maxy);
    const bool ok = active && !(weight < prm.cosAngle1) && inside;
    Frame f;
    f.tlx = ok ? tlx : 0.0f; f.tly = ok ? tly : 0.0f;
    f.dxx = ok ? dx.x : 0.0f; f.dxy = ok ? dx.y : 0.0f; f.dyx = ok ? dy.x : 0.0f; f.dyy = ok ? dy.y : 0.0f;
    f.w = ok ? W : (W0 >> prm.level);
    f.ok = ok ? 1 : 0;
    const unsigned long long a = ok ? base_new : base_level;
    f.img_lo = (unsigned)(a & 0xffffffffull); f.img_hi = (unsigned)(a >> 32);
    return f;
}

struct __attribute__((packed, aligned(4))) Texel2 { uint32_t a, b; };

// Per-lane constants of the class-lane sample layout: lane 16 r + c owns samples c, c + 16, c + 32 of the window its row
// works on (a proposal in a refinement step, a view in a single evaluation).  A window of 16 k + 1 samples (7x7) has one
// sample more, the "extra" xbase, which the lane that owns the sampling frame takes.
struct ClsConst {
    unsigned cs[3];  // per slot j: fx | fy << 8 | valid << 16
    int fb;          // first lane of this lane's row (16 r)
    int xbase, nx;   // the extra: sample xbase if nx == 1 (wave-uniform)
};
// Per-wave working state.
struct WaveCtx {
    int lane;
    ClsConst cc;
    unsigned evals, view_evals;
#ifdef MVS_STAGE_TIMING
    unsigned long long st_acc[8];  // diagnostic build: phase times, kept in registers and flushed once by the kernel
    unsigned long long st_t;
#endif
};
#ifdef MVS_STAGE_TIMING
#define WC_T0(wc) { (wc).st_t = (unsigned long long)__builtin_amdgcn_s_memtime(); }
#define WC_ADD(wc, k) { const unsigned long long t1_ = (unsigned long long)__builtin_amdgcn_s_memtime(); (wc).st_acc[k] += t1_ - (wc).st_t; (wc).st_t = t1_; }
#else
#define WC_T0(wc)
#define WC_ADD(wc, k)
#endif

// The sampling frames of an evaluation are published in LDS (frames_publish) and read back by the sampling lanes: uniform
// or per-row ds_read_b128 instead of ten v_readlane_b32 per sample on the vector ALU, the unit this kernel saturates.
#define MVS_PIVOT_LDS4 192                         // float4 index of the per-view pivot colours (refinePatch, class lanes)
#define MVS_FRAME_LDS_BYTES (64 * 48 + 16 * 16)    // frame lanes 0..63, 12 dwords each, + 16 pivots; the start of the kernel's dynamic LDS
#define MVS_FRAME1_LDS_BYTES (48 * (MVS_LISTCAP > 16 ? MVS_LISTCAP : 16))  // what a single-proposal evaluation publishes there
extern __shared__ float4 mvs_dyn_lds4[];
DEV void frames_publish(const WaveCtx& wc, const Frame& f, int nlanes) {
    __syncthreads();  // whatever used the region before (setRefImage textures, Optim::check rows) is done
    if (wc.lane < nlanes) {
        float4* d = mvs_dyn_lds4 + 3 * wc.lane;
        d[0] = make_float4(f.tlx, f.tly, f.dxx, f.dxy);
        d[1] = make_float4(f.dyx, f.dyy, __int_as_float(f.w), __int_as_float(f.ok));
        d[2] = make_float4(__int_as_float((int)f.img_lo), __int_as_float((int)f.img_hi), 0.0f, 0.0f);
    }
    __syncthreads();
}
// second half of Optim::normalize, optim.cpp:932-939, on whatever lanes hold an ssd: 1 / msd
DEV float inv_msd(const DParams& prm, float ssd) {
    float msd = sqrt_rn(ssd * prm.inv_3sz);
    if (msd == 0.0f) msd = 1.0f;
    return rcp_rn(msd);
}

// ------------------------------------------------------------------ class-lane evaluation (the four proposals of a refinement step)
// Optim::getTex sampling + normalize + dot for proposals g = 0..3 against the views k < n, with a lane per (proposal,
// sample class) instead of a lane per sample: lane 16 g + c walks its samples c, c + 16, c + 32 of EVERY view and keeps
// running sums -- colour, colour^2 and colour x reference colour (the reference view's colours of its own samples stay in
// registers) -- so no sum crosses lanes per sample.  Per view the five sums are finished by the four DPP steps of a
// 16-lane row, once for all four proposals, and dropped into view lane 16 g + k; the per-view scalars (1/msd, INCC) then
// come out of one vector sequence.  A wave instruction of the view loop covers 64 samples.
// The few samples beyond the last full 16 ("extras": sample 48 of a 7x7 window) are taken afterwards by the FRAME lanes:
// lane 16 g + k samples them in its own frame (proposal g, view k) and adds them to the sums it has just received.
// Arithmetic (mirrored by the oracle, tex_stats_class16): colours are taken relative to a per-view pivot p (the view's
// mean colour in refinePatch's first evaluation), c' = blend - p;  S1 = sum c', S2 = sum |c'|^2, S01 = sum c' . c0';
// mean m = S1 / n;  ssd = max(S2 - S1 . m, 0);  dot = S01 - S1 . m0;  INCC = 1 - dot (inv0 inv) / 3n.  The sums run
// j-ascending inside a lane, the row tree pairs lanes 1, 2, 4, 8 apart, the extras are added last, in sample order.
struct ClsPend { Texel2 q0, q1; float dx1, dy1; };
DEV ClsConst make_cls(const DParams& prm, int lane) {
    ClsConst cc;
    const int wsz = prm.wsz, nj = wsz >> 4, rem = wsz & 15;
    const int njx = nj + (rem > 1 ? 1 : 0), lim = rem > 1 ? wsz : 16 * nj;
    const int row = lane >> 4, c = lane & 15;
    cc.fb = 16 * row;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = c + 16 * j;
        const bool valid = j < njx && q < lim;
        cc.cs[j] = valid ? (unsigned)(q % prm.wsize) | ((unsigned)(q / prm.wsize) << 8) | (1u << 16) : 0u;
    }
    cc.xbase = 16 * nj; cc.nx = rem == 1 ? 1 : 0;
    return cc;
}
DEV float cvt_ub0(unsigned x) { return (float)(x & 255u); }
DEV float cvt_ub1(unsigned x) { return (float)((x >> 8) & 255u); }
DEV float cvt_ub2(unsigned x) { return (float)((x >> 16) & 255u); }
// The sample constants are unpacked where they are used (three v_cvt_f32_ubyte): hoisted out of the refinement loop as nine
// floats they only turn into scratch traffic.  The empty asm hides the loop invariance from the optimiser.
DEV unsigned cls_opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
struct ClsFrame { float tlx, tly, dxx, dxy, dyx, dyy; int w; unsigned long long base; };
DEV ClsFrame cls_frame(int fidx) {
    const float4* s = mvs_dyn_lds4 + 3 * fidx;
    const float4 A = s[0], B = s[1];
    const float2 Cc = *reinterpret_cast<const float2*>(s + 2);
    ClsFrame f;
    f.tlx = A.x; f.tly = A.y; f.dxx = A.z; f.dxy = A.w; f.dyx = B.x; f.dyy = B.y; f.w = __float_as_int(B.z);
    f.base = ((unsigned long long)(unsigned)__float_as_int(Cc.y) << 32) | (unsigned long long)(unsigned)__float_as_int(Cc.x);
    return f;
}
DEV ClsPend cls_issue(const ClsFrame& f, unsigned cs) {
    typedef const __attribute__((address_space(1))) uint32_t* GlobalTexels;
    // a slot without a sample has cs = 0: it reads the window's first texels (a valid address) and is weighted out in
    // cls_colour, so no select is needed here
    const float fx = cvt_ub0(cs), fy = cvt_ub1(cs);
    const float sx = fma_(f.dyx, fy, fma_(f.dxx, fx, f.tlx));
    const float sy = fma_(f.dyy, fy, fma_(f.dxy, fx, f.tly));
    const int lx = (int)sx, ly = (int)sy;
    const unsigned long long a0 = f.base + 4ull * (unsigned long long)(unsigned)(ly * f.w + lx);
    const GlobalTexels t0 = (GlobalTexels)a0, t1 = (GlobalTexels)(a0 + 4ull * (unsigned long long)(unsigned)f.w);
    ClsPend p;
    p.q0.a = t0[0]; p.q0.b = t0[1];
    p.q1.a = t1[0]; p.q1.b = t1[1];
    // sx - (float)(int)sx for the non-negative positions getTexSafe lets through: one v_fract_f32 (exact)
    p.dx1 = __builtin_amdgcn_fractf(sx); p.dy1 = __builtin_amdgcn_fractf(sy);
    return p;
}
// bilinear blend (Image::getColor, image.cpp:447-472) minus the pivot; 0 on lanes whose sample does not exist
DEV void cls_colour(const ClsPend& p, unsigned cs, float pr, float pg, float pb, float& r, float& g, float& b) {
    const Texel2 q0 = p.q0, q1 = p.q1;
    const float fm = cvt_ub2(cs);  // 1, or 0 for a slot without a sample: both column weights vanish, and with them all four
    const float dx1 = p.dx1 * fm, dx0 = fm - dx1, dy1 = p.dy1, dy0 = 1.0f - dy1;
    const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
    r = fma_((float)(q1.b & 255u), f11, fma_((float)(q0.b & 255u), f10, fma_((float)(q1.a & 255u), f01, (float)(q0.a & 255u) * f00)));
    g = fma_((float)((q1.b >> 8) & 255u), f11, fma_((float)((q0.b >> 8) & 255u), f10, fma_((float)((q1.a >> 8) & 255u), f01, (float)((q0.a >> 8) & 255u) * f00)));
    b = fma_((float)((q1.b >> 16) & 255u), f11, fma_((float)((q0.b >> 16) & 255u), f10, fma_((float)((q1.a >> 16) & 255u), f01, (float)((q0.a >> 16) & 255u) * f00)));
    r = fma_(-pr, fm, r); g = fma_(-pg, fm, g); b = fma_(-pb, fm, b);
}
DEV float bperm_f(int addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))); }
#define MVS_ROW_STEP(x, C) x = x + dpp_f<C>(x);
// (bound_ctrl form: every lane has a source within its row, so no "old" value has to be provided)
#define MVS_ROW_STEP5(C) { const float t0 = dpp0_f<C>(s1r), t1 = dpp0_f<C>(s1g), t2 = dpp0_f<C>(s1b), t3 = dpp0_f<C>(s2), t4 = dpp0_f<C>(s01); \
                           s1r = s1r + t0; s1g = s1g + t1; s1b = s1b + t2; s2 = s2 + t3; s01 = s01 + t4; }
#define MVS_ROW_STEP4(C) { const float t0 = dpp0_f<C>(s1r), t1 = dpp0_f<C>(s1g), t2 = dpp0_f<C>(s1b), t3 = dpp0_f<C>(s2); \
                           s1r = s1r + t0; s1g = s1g + t1; s1b = s1b + t2; s2 = s2 + t3; }
// frames: published in LDS for frame lanes 16 g + k (frames_publish); okm[g] = views of proposal g that sample.
// Leaves in frame lane 16 g + k (k >= 1) the INCC of view k against the reference view of proposal g.
// Straight-line per view (always three sample slots per lane; a slot without a sample contributes exact zeros), the
// reference view peeled off, and the loads of view k + 1 issued as the slots of view k are consumed.
DEV void eval_steps4(const DParams& prm, WaveCtx& wc, const ClsConst& cc, const Frame& f, int n, unsigned (&okm)[4], float& incc_l) {
    frames_publish(wc, f, 64);
    const unsigned long long okb = ballot(f.ok != 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) okm[g] = (unsigned)((okb >> (16 * g)) & 0xffffull);
#pragma unroll
    for (int g = 0; g < 4; ++g) wc.view_evals += (okm[g] & 1u) ? (unsigned)__popc(okm[g]) : 0u;
    const int fb = cc.fb;
    const int lc = wc.lane & 15;
    float c0[3][3];
    ClsPend pend[3];
#if MVS_ST_DEPTH > 1
    ClsPend ahead[3];  // the loads of the view after next (many-view builds: one or two waves per SIMD hide no latency)
#endif
    float P1r, P1g, P1b, P2, P01 = 0.0f;
    const int vlast = max(n - 1, 0);
    {
        const ClsFrame fr = cls_frame(fb);
#pragma unroll
        for (int j = 0; j < 3; ++j) pend[j] = cls_issue(fr, cls_opaque(cc.cs[j]));
#if MVS_ST_DEPTH > 1
        const ClsFrame f1 = cls_frame(fb + min(1, vlast));
#pragma unroll
        for (int j = 0; j < 3; ++j) ahead[j] = cls_issue(f1, cls_opaque(cc.cs[j]));
#endif
    }
    {   // the reference view
        const float4 pv = mvs_dyn_lds4[MVS_PIVOT_LDS4];
        const ClsFrame fn = cls_frame(fb + min(MVS_ST_DEPTH, vlast));
        float s1r = 0.0f, s1g = 0.0f, s1b = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned cs = cls_opaque(cc.cs[j]);
            float r, g, b;
            cls_colour(pend[j], cs, pv.x, pv.y, pv.z, r, g, b);
#if MVS_ST_DEPTH > 1
            pend[j] = ahead[j]; ahead[j] = cls_issue(fn, cs);
#else
            pend[j] = cls_issue(fn, cs);
#endif
            c0[j][0] = r; c0[j][1] = g; c0[j][2] = b;
            s1r += r; s1g += g; s1b += b;
            s2 = fma_(r, r, s2); s2 = fma_(g, g, s2); s2 = fma_(b, b, s2);
        }
        MVS_ROW_STEP4(0xB1) MVS_ROW_STEP4(0x4E) MVS_ROW_STEP4(0x141) MVS_ROW_STEP4(0x140)
        P1r = s1r; P1g = s1g; P1b = s1b; P2 = s2;  // lanes lc != 0 are overwritten below or unused
    }
    for (int k = 1; k < n; ++k) {
        const float4 pv = mvs_dyn_lds4[MVS_PIVOT_LDS4 + k];
        const ClsFrame fn = cls_frame(fb + min(k + MVS_ST_DEPTH, vlast));
        float s1r = 0.0f, s1g = 0.0f, s1b = 0.0f, s2 = 0.0f, s01 = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned cs = cls_opaque(cc.cs[j]);
            float r, g, b;
            cls_colour(pend[j], cs, pv.x, pv.y, pv.z, r, g, b);
#if MVS_ST_DEPTH > 1
            pend[j] = ahead[j]; ahead[j] = cls_issue(fn, cs);
#else
            pend[j] = cls_issue(fn, cs);
#endif
            s1r += r; s1g += g; s1b += b;
            s2 = fma_(r, r, s2); s2 = fma_(g, g, s2); s2 = fma_(b, b, s2);
            s01 = fma_(r, c0[j][0], s01); s01 = fma_(g, c0[j][1], s01); s01 = fma_(b, c0[j][2], s01);
        }
        MVS_ROW_STEP5(0xB1) MVS_ROW_STEP5(0x4E) MVS_ROW_STEP5(0x141) MVS_ROW_STEP5(0x140)  // the row tree (lanes 1, 2, 4, 8 apart)
        if (lc == k) { P1r = s1r; P1g = s1g; P1b = s1b; P2 = s2; P01 = s01; }
    }
    const int a0 = (wc.lane & 48) << 2;  // ds_bpermute address of lane 16 g: the reference view of this lane's proposal
    // the extras: frame lane 16 g + k samples them in its own frame (a frame that does not sample holds a harmless address
    // and its sums are never looked at) and adds them to the sums of (proposal g, view k) it holds
    if (cc.nx > 0) {
        const float4 pv = mvs_dyn_lds4[MVS_PIVOT_LDS4 + lc];
        ClsFrame fr;
        fr.tlx = f.tlx; fr.tly = f.tly; fr.dxx = f.dxx; fr.dxy = f.dxy; fr.dyx = f.dyx; fr.dyy = f.dyy; fr.w = f.w;
        fr.base = ((unsigned long long)f.img_hi << 32) | (unsigned long long)f.img_lo;
        {
            const int q = cc.xbase;
            const unsigned cs = (unsigned)(q % prm.wsize) | ((unsigned)(q / prm.wsize) << 8) | (1u << 16);
            const ClsPend pe = cls_issue(fr, cs);
            float r, g, b;
            cls_colour(pe, cs, pv.x, pv.y, pv.z, r, g, b);
            const float r0 = bperm_f(a0, r), g0 = bperm_f(a0, g), b0 = bperm_f(a0, b);
            P1r += r; P1g += g; P1b += b;
            P2 = fma_(r, r, P2); P2 = fma_(g, g, P2); P2 = fma_(b, b, P2);
            P01 = fma_(r, r0, P01); P01 = fma_(g, g0, P01); P01 = fma_(b, b0, P01);
        }
    }
    // view lane 16 g + k: mean, ssd, centred product of view k; the reference view's mean and 1/msd come from lane 16 g
    const float m_r = P1r * prm.inv_sz, m_g = P1g * prm.inv_sz, m_b = P1b * prm.inv_sz;
    const float ssd = fmaxf(P2 - fma_(P1b, m_b, fma_(P1g, m_g, P1r * m_r)), 0.0f);
    const float inv_l = inv_msd(prm, ssd);
    const float m0r = bperm_f(a0, m_r), m0g = bperm_f(a0, m_g), m0b = bperm_f(a0, m_b), inv0 = bperm_f(a0, inv_l);
    const float dot = P01 - fma_(P1b, m0b, fma_(P1g, m0g, P1r * m0r));
    incc_l = 1.0f - (dot * (inv0 * inv_l)) * prm.inv_3sz;
}

// ------------------------------------------------------------------ class-lane evaluation of ONE patch against n views
// Every single evaluation (computeINCC, setINCCs, constraintImages, the first cost_func of refinePatch, setRefImage's
// textures): frame lane k < n holds the sampling frame of view k.  The four 16-lane rows take four views at a time --
// row r of round t samples view 4 t + r, lane 16 r + c its samples c, c + 16, c + 32 -- and the frame lanes take the
// extra sample of their own view beforehand.  Optim::normalize and Optim::dot in two passes as the reference has them:
// channel means (lane sums, row tree, extra), centring, then sum of squares and product with the centred reference
// texture (same order).  The reference view is view 0 = row 0 of round 0; its centred colours reach the other rows
// through ds_bpermute once, after which every row holds them for the samples it owns.
// The per-view scalars of view v are formed in lane 16 (v & 3) + (v >> 2) and moved to view lane v at the end.
// Leaves in view lane k >= 1 the INCC of view k against the reference view, okm[0] = views that sampled.
// PIV: piv[0..2] receive, in view lane k, the channel means of view k (128 for a view that was not sampled) -- the
// pivots of the class-lane steps that follow in refinePatch.
// texs != nullptr: the centred texture of view k goes to LDS, texs[3 k tstride + 3 sample + channel], and *ssd_out
// receives, in view lane k, its sum of squares -- what Optim::setRefImage needs.
DEV void cls_raw(const ClsPend& p, unsigned cs, float& r, float& g, float& b) {  // bilinear blend; 0 on a slot without a sample
    const Texel2 q0 = p.q0, q1 = p.q1;
    const float fm = cvt_ub2(cs);
    const float dx1 = p.dx1 * fm, dx0 = fm - dx1, dy1 = p.dy1, dy0 = 1.0f - dy1;
    const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
    r = fma_((float)(q1.b & 255u), f11, fma_((float)(q0.b & 255u), f10, fma_((float)(q1.a & 255u), f01, (float)(q0.a & 255u) * f00)));
    g = fma_((float)((q1.b >> 8) & 255u), f11, fma_((float)((q0.b >> 8) & 255u), f10, fma_((float)((q1.a >> 8) & 255u), f01, (float)((q0.a >> 8) & 255u) * f00)));
    b = fma_((float)((q1.b >> 16) & 255u), f11, fma_((float)((q0.b >> 16) & 255u), f10, fma_((float)((q1.a >> 16) & 255u), f01, (float)((q0.a >> 16) & 255u) * f00)));
}
#define MVS_ROW_STEP3(C) { const float t0 = dpp0_f<C>(s1r), t1 = dpp0_f<C>(s1g), t2 = dpp0_f<C>(s1b); s1r = s1r + t0; s1g = s1g + t1; s1b = s1b + t2; }
#define MVS_ROW_STEP2(C) { const float t0 = dpp0_f<C>(sq), t1 = dpp0_f<C>(dt); sq = sq + t0; dt = dt + t1; }
#ifndef MVS_EV_PREFETCH
#define MVS_EV_PREFETCH 1
#endif

#ifndef MVS_PAIR_MFMA
#define MVS_PAIR_MFMA (MVS_LISTCAP > 16)  // setRefImage's pair sums on the matrix cores (the 32- and 64-view builds)
#endif
#if MVS_PAIR_MFMA
// The Gram matrix G[a][b] = sum(k) t_a[k] t_b[k] of the kept textures is taken WHILE the views are sampled, in chunks of 16 views, so
// that only one chunk of textures lies in LDS at a time (the textures of a whole 32- or 64-view list were what held these builds at one
// or two waves per SIMD).  Every 16 x 16 tile runs down the k-ordered chain acc = fma(t_a[k], t_b[k], acc), which is what
// v_mfma_f32_16x16x4_f32 computes (37 of them for a 7 x 7 window).  G then replaces the textures in LDS (prm.gram_ld floats per row).
// The MFMA operands of a finished chunk go to PRIVATE memory -- a dynamically indexed per-lane array, i.e. the compiler's scratch
// segment: L1 / L2 resident, no slot bookkeeping, each lane reads back exactly what it wrote -- and come back eight k-steps at a time:
//   MVS_GRAM_SCRATCH 1 (the 32-view build): the tiles of a chunk with the earlier chunks are taken as soon as the chunk is sampled,
//     the accumulators of all tiles (3 x 4 registers) stay in registers until the list is through;
//   MVS_GRAM_SCRATCH 2 (the 64-view build): nothing of the matrix lives in registers while the views are sampled -- all chunks'
//     operands go out, and when the list is through the (up to 10) tiles are taken one after the other from there (4 accumulator
//     registers): 44 spilled registers at three waves per SIMD where the first form spills 146.
// Rounds 3-4 kept chunk A (half the list) in 37 / 74 registers while chunk B was sampled: the 64-view build then needed the 256 VGPRs
// and the 22 KB of LDS of two waves per SIMD and ran at 11.5 M patches/s on 48 x 540p; 13.6 M with form 1, 14.5 M with form 2
// (gpurun_out/r04l, r04m); the 32-view build 17.4 M either way, 16.6 M with form 2.
#ifndef MVS_GRAM_SCRATCH
#define MVS_GRAM_SCRATCH (MVS_LISTCAP > 32 ? 2 : 1)
#endif
#define MVS_GRAM_CH 16
#define MVS_GRAM_NCH (MVS_LISTCAP / 16)                          // chunks at most: 2 or 4 (1 in an experimental 16-view build)
#define MVS_GRAM_NT (MVS_GRAM_NCH * (MVS_GRAM_NCH + 1) / 2)     // tiles (p <= c): 3 or 10
#define MVS_GRAM_KSP 40                                         // MVS_GRAM_KS rounded up to groups of 8 k-steps
#define MVS_GRAM_T(p, c) ((c) * ((c) + 1) / 2 + (p))
#define MVS_GRAM_KK 4                                           // k values per MFMA (16x16x4)
#define MVS_GRAM_KS ((147 + MVS_GRAM_KK - 1) / MVS_GRAM_KK)      // MFMAs per tile for a 7x7 window (3 * 49 elements): 37
#define MVS_GRAM_NACC 4                                         // accumulator registers of a tile
typedef float gram_acc_t __attribute__((ext_vector_type(MVS_GRAM_NACC)));
DEV gram_acc_t gram_mfma(float a, float b, gram_acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// accumulator register r of lane l holds tile element (row, col)
DEV int gram_col(int lane) { return lane & 15; }
DEV int gram_row(int lane, int r) { return (lane >> 4) * 4 + r; }
template <int N> struct GramIC { static constexpr int value = N; };
#endif
template <bool PIV = false>
DEV void eval_views(const DParams& prm, WaveCtx& wc, const Frame& f, int n, vmask_t (&okm)[1], float& incc_l, float* piv = nullptr,
                    float* texs = nullptr, int tstride = 0, float* ssd_out = nullptr) {
#if MVS_PAIR_MFMA
    // texs != nullptr: the Gram matrix of the centred textures is left at texs[a * LD + b] (see MVS_GRAM_CH above)
#if MVS_GRAM_SCRATCH == 2
    float gops[MVS_GRAM_NCH * MVS_GRAM_KSP];           // private memory: the MFMA operands of ALL chunks (lane l: t_{l % 16}[4 s + l / 16])
#elif MVS_GRAM_SCRATCH
    gram_acc_t gacc[MVS_GRAM_NT];                      // tile (p, c), p <= c, at MVS_GRAM_T(p, c)
    float gops[(MVS_GRAM_NCH > 1 ? MVS_GRAM_NCH - 1 : 1) * MVS_GRAM_KSP];  // private memory: the MFMA operands of the finished chunks (lane l: t_{l % 16}[4 s + l / 16]); unused with one chunk
#pragma unroll
    for (int q = 0; q < MVS_GRAM_NT; ++q)
#pragma unroll
        for (int r = 0; r < MVS_GRAM_NACC; ++r) gacc[q][r] = 0.0f;
#endif
    const int gK = 3 * prm.wsz, gtp = 3 * tstride;
    const int gk0 = wc.lane / MVS_GRAM_CH;                       // this lane's k within an MFMA
    const float* const grow = texs + (wc.lane & (MVS_GRAM_CH - 1)) * gtp;  // this lane's view of the chunk in LDS
#endif
    const ClsConst& cc = wc.cc;
    frames_publish(wc, f, MVS_LISTCAP > 16 ? MVS_LISTCAP : 16);
    okm[0] = vballot(f.ok != 0);  // a frame is only ever valid on a lane < n <= MVS_LISTCAP
    wc.view_evals += (okm[0] & 1u) ? (unsigned)vpop(okm[0]) : 0u;
    const int row = wc.lane >> 4, lc = wc.lane & 15;
    // the extra sample of view k in frame lane k (raw colours): its loads go out first, the first round's behind them
    float fxr = 0.0f, fxg = 0.0f, fxb = 0.0f;
    unsigned xcs = 0u;
    ClsPend pe;
    if (cc.nx > 0) {
        ClsFrame fr;
        fr.tlx = f.tlx; fr.tly = f.tly; fr.dxx = f.dxx; fr.dxy = f.dxy; fr.dyx = f.dyx; fr.dyy = f.dyy; fr.w = f.w;
        fr.base = ((unsigned long long)f.img_hi << 32) | (unsigned long long)f.img_lo;
        xcs = (unsigned)(cc.xbase % prm.wsize) | ((unsigned)(cc.xbase / prm.wsize) << 8) | (1u << 16);
        pe = cls_issue(fr, xcs);
    }
    float d0[3][3], d0x[3] = {0.0f, 0.0f, 0.0f};
    float ssd_l = 1.0f, dot_l = 0.0f, mr_l = 128.0f, mg_l = 128.0f, mb_l = 128.0f;
    ClsPend pend[3];
#if MVS_EV_DEPTH > 1
    ClsPend ahead[MVS_EV_DEPTH - 1][3];  // the loads of the rounds t + 1 .. t + MVS_EV_DEPTH - 1 (many-view builds: one or two waves per SIMD hide no latency)
#endif
    {
        const int last = max(n - 1, 0);  // an empty list never reaches this point; the clamps below stay inside the frames all the same
        const ClsFrame fr = cls_frame(min(row, last));
#pragma unroll
        for (int j = 0; j < 3; ++j) pend[j] = cls_issue(fr, cls_opaque(cc.cs[j]));
#if MVS_EV_DEPTH > 1
#pragma unroll
        for (int d = 0; d < MVS_EV_DEPTH - 1; ++d) {
            const ClsFrame fd = cls_frame(min(row + 4 * (d + 1), last));
#pragma unroll
            for (int j = 0; j < 3; ++j) ahead[d][j] = cls_issue(fd, cls_opaque(cc.cs[j]));
        }
#endif
    }
    if (cc.nx > 0) cls_raw(pe, xcs, fxr, fxg, fxb);
    const int rounds = (n + 3) >> 2;
    auto round_body = [&](const int t) {
        const int v = 4 * t + row, vq = min(v, max(n - 1, 0));
#if MVS_EV_PREFETCH
        const ClsFrame fn = cls_frame(min(v + 4 * MVS_EV_DEPTH, max(n - 1, 0)));
#endif
        float cr[3], cg[3], cb[3];
        float s1r = 0.0f, s1g = 0.0f, s1b = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned cs = cls_opaque(cc.cs[j]);
            cls_raw(pend[j], cs, cr[j], cg[j], cb[j]);
#if MVS_EV_PREFETCH
#if MVS_EV_DEPTH > 1
            pend[j] = ahead[0][j];
#pragma unroll
            for (int d = 0; d + 1 < MVS_EV_DEPTH - 1; ++d) ahead[d][j] = ahead[d + 1][j];
            ahead[MVS_EV_DEPTH - 2][j] = cls_issue(fn, cs);
#else
            pend[j] = cls_issue(fn, cs);
#endif
#endif
            s1r += cr[j]; s1g += cg[j]; s1b += cb[j];
        }
        MVS_ROW_STEP3(0xB1) MVS_ROW_STEP3(0x4E) MVS_ROW_STEP3(0x141) MVS_ROW_STEP3(0x140)
        float xr = 0.0f, xg = 0.0f, xb = 0.0f;
        if (cc.nx > 0) {
            xr = bperm_f(4 * vq, fxr); xg = bperm_f(4 * vq, fxg); xb = bperm_f(4 * vq, fxb);
            s1r += xr; s1g += xg; s1b += xb;
        }
        const float mr = s1r * prm.inv_sz, mg = s1g * prm.inv_sz, mb = s1b * prm.inv_sz;
        float er[3], eg[3], eb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float fm = cvt_ub2(cls_opaque(cc.cs[j]));
            er[j] = fma_(-mr, fm, cr[j]); eg[j] = fma_(-mg, fm, cg[j]); eb[j] = fma_(-mb, fm, cb[j]);
        }
        const float exr = xr - mr, exg = xg - mg, exb = xb - mb;
        if (t == 0) {  // the centred reference texture: from row 0 to every row
            const int a0 = 4 * lc;
#pragma unroll
            for (int j = 0; j < 3; ++j) { d0[j][0] = bperm_f(a0, er[j]); d0[j][1] = bperm_f(a0, eg[j]); d0[j][2] = bperm_f(a0, eb[j]); }
            if (cc.nx > 0) { d0x[0] = bperm_f(0, exr); d0x[1] = bperm_f(0, exg); d0x[2] = bperm_f(0, exb); }
        }
        float sq = 0.0f, dt = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            sq += fma_(eb[j], eb[j], fma_(eg[j], eg[j], er[j] * er[j]));
            dt += fma_(d0[j][2], eb[j], fma_(d0[j][1], eg[j], d0[j][0] * er[j]));
        }
        MVS_ROW_STEP2(0xB1) MVS_ROW_STEP2(0x4E) MVS_ROW_STEP2(0x141) MVS_ROW_STEP2(0x140)
        if (cc.nx > 0) {
            sq += fma_(exb, exb, fma_(exg, exg, exr * exr));
            dt += fma_(d0x[2], exb, fma_(d0x[1], exg, d0x[0] * exr));
        }
        if (lc == t) { ssd_l = sq; dot_l = dt; if (PIV) { mr_l = mr; mg_l = mg; mb_l = mb; } }
#if !MVS_EV_PREFETCH
        if (t + 1 < rounds) {
            const ClsFrame fn = cls_frame(min(v + 4, max(n - 1, 0)));
#pragma unroll
            for (int j = 0; j < 3; ++j) pend[j] = cls_issue(fn, cls_opaque(cc.cs[j]));
        }
#endif
#if MVS_PAIR_MFMA
        if (texs) {  // sample-major: element 3 q + channel of the view's row of 3 * tstride floats (the k order of the pair sums); slot = view % CH
            float* tv = texs + (3 * (v & (MVS_GRAM_CH - 1))) * tstride;
            const bool have = v < n;  // rows without a view hold zeros: their products vanish and are never looked at
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cc.cs[j] >> 16) { const int q = 3 * (lc + 16 * j); tv[q] = have ? er[j] : 0.0f; tv[q + 1] = have ? eg[j] : 0.0f; tv[q + 2] = have ? eb[j] : 0.0f; }
            if (cc.nx > 0 && lc == 0) { tv[3 * cc.xbase] = have ? exr : 0.0f; tv[3 * cc.xbase + 1] = have ? exg : 0.0f; tv[3 * cc.xbase + 2] = have ? exb : 0.0f; }
        }
#else
        if (texs && v < n) {  // sample-major: element 3 q + channel of view v's row of 3 * tstride floats (the k order of the pair sums)
            float* tv = texs + (3 * v) * tstride;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cc.cs[j] >> 16) { const int q = 3 * (lc + 16 * j); tv[q] = er[j]; tv[q + 1] = eg[j]; tv[q + 2] = eb[j]; }
            if (cc.nx > 0 && lc == 0) { tv[3 * cc.xbase] = exr; tv[3 * cc.xbase + 1] = exg; tv[3 * cc.xbase + 2] = exb; }
        }
#endif
    };
#if MVS_PAIR_MFMA && MVS_GRAM_SCRATCH == 2
    // Variant: NOTHING of the Gram matrix lives in registers while the views are sampled.  A finished chunk's operands go to private
    // memory; when the list is through, the tiles are taken one after the other from there (four accumulator registers) and written
    // to LDS over the textures.
    if (texs) {
        int ch = 0;
        for (int t = 0; t < rounds; ++t) {
            round_body(t);
            if ((t & 3) == 3 || t + 1 == rounds) {
                const int rc = (t & 3) + 1;
                __syncthreads();
                for (int vz = 4 * rc + (wc.lane >> 4); vz < MVS_GRAM_CH; vz += 4)
                    for (int q = lc; q < gtp; q += 16) texs[vz * gtp + q] = 0.0f;
                __syncthreads();
#pragma unroll 1
                for (int s0 = 0; s0 < MVS_GRAM_KSP; s0 += 8) {
                    float tb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = MVS_GRAM_KK * (s0 + j) + gk0;
                        tb[j] = (s0 + j < MVS_GRAM_KS && k < gK) ? grow[k] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) gops[ch * MVS_GRAM_KSP + s0 + j] = tb[j];
                }
                __syncthreads();
                ++ch;
            }
        }
        const int LD = prm.gram_ld, col = gram_col(wc.lane);
#pragma unroll 1
        for (int c2 = 0; c2 < ch; ++c2) {
#pragma unroll 1
            for (int pq = 0; pq <= c2; ++pq) {
                gram_acc_t acc;
#pragma unroll
                for (int r = 0; r < MVS_GRAM_NACC; ++r) acc[r] = 0.0f;
#pragma unroll 1
                for (int s0 = 0; s0 < MVS_GRAM_KSP; s0 += 8) {
                    float a8[8], b8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a8[j] = gops[pq * MVS_GRAM_KSP + s0 + j]; b8[j] = gops[c2 * MVS_GRAM_KSP + s0 + j]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (s0 + j < MVS_GRAM_KS) acc = gram_mfma(a8[j], b8[j], acc);
                }
#pragma unroll
                for (int r = 0; r < MVS_GRAM_NACC; ++r) {
                    const int rw = gram_row(wc.lane, r);
                    texs[(MVS_GRAM_CH * pq + rw) * LD + MVS_GRAM_CH * c2 + col] = acc[r];
                    if (pq != c2) texs[(MVS_GRAM_CH * c2 + col) * LD + MVS_GRAM_CH * pq + rw] = acc[r];
                }
            }
        }
        // rows / columns of chunks the list never reached are never read (set_ref_image reads indices < n_eval)
        __syncthreads();
    } else
        for (int t = 0; t < rounds; ++t) round_body(t);
#elif MVS_PAIR_MFMA && MVS_GRAM_SCRATCH
    // chunk C is complete after `rc` rounds of its own (or the list ends inside it): its diagonal tile from LDS, its tiles with the
    // earlier chunks p < C (their operands back from private memory, eight k-steps in flight), and its own operands out
    auto chunk_done = [&](auto Cc, const int rc) {
        constexpr int C = decltype(Cc)::value;
        __syncthreads();
        // rows of the chunk that no round wrote (the list ends inside it) must read as zeros
        for (int vz = 4 * rc + (wc.lane >> 4); vz < MVS_GRAM_CH; vz += 4)
            for (int q = lc; q < gtp; q += 16) texs[vz * gtp + q] = 0.0f;
        __syncthreads();
#pragma unroll 1
        for (int s0 = 0; s0 < MVS_GRAM_KSP; s0 += 8) {
            float tb[8], ap[C > 0 ? C : 1][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = MVS_GRAM_KK * (s0 + j) + gk0;
                tb[j] = (s0 + j < MVS_GRAM_KS && k < gK) ? grow[k] : 0.0f;
#pragma unroll
                for (int pq = 0; pq < C; ++pq) ap[pq][j] = gops[pq * MVS_GRAM_KSP + s0 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (s0 + j < MVS_GRAM_KS) {  // wave-uniform
                    gacc[MVS_GRAM_T(C, C)] = gram_mfma(tb[j], tb[j], gacc[MVS_GRAM_T(C, C)]);
#pragma unroll
                    for (int pq = 0; pq < C; ++pq) gacc[MVS_GRAM_T(pq, C)] = gram_mfma(ap[pq][j], tb[j], gacc[MVS_GRAM_T(pq, C)]);
                }
            }
            if (C < MVS_GRAM_NCH - 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gops[(C < MVS_GRAM_NCH - 1 ? C : 0) * MVS_GRAM_KSP + s0 + j] = tb[j];
            }
        }
        __syncthreads();
    };
    if (texs) {
        int ch = 0;
        for (int t = 0; t < rounds; ++t) {
            round_body(t);
            if ((t & 3) == 3 || t + 1 == rounds) {
                const int rc = (t & 3) + 1;
                switch (ch) {
                    case 0: chunk_done(GramIC<0>{}, rc); break;
#if MVS_GRAM_NCH > 1
                    case 1: chunk_done(GramIC<1>{}, rc); break;
#endif
#if MVS_GRAM_NCH > 2
                    case 2: chunk_done(GramIC<2>{}, rc); break;
                    default: chunk_done(GramIC<3>{}, rc); break;
#else
                    default: break;
#endif
                }
                ++ch;
            }
        }
        // the tiles to LDS over the textures: G[a][b], a and b in the order of this evaluation's list (LD = prm.gram_ld columns)
        __syncthreads();
        const int LD = prm.gram_ld, col = gram_col(wc.lane);
#pragma unroll
        for (int c2 = 0; c2 < MVS_GRAM_NCH; ++c2) {
            if (MVS_GRAM_CH * c2 >= LD) continue;  // chunks the data set's list length never reaches (wave-uniform)
#pragma unroll
            for (int pq = 0; pq <= c2; ++pq) {
#pragma unroll
                for (int r = 0; r < MVS_GRAM_NACC; ++r) {
                    const int rw = gram_row(wc.lane, r);
                    const float gv = gacc[MVS_GRAM_T(pq, c2)][r];
                    texs[(MVS_GRAM_CH * pq + rw) * LD + MVS_GRAM_CH * c2 + col] = gv;
                    if (pq != c2) texs[(MVS_GRAM_CH * c2 + col) * LD + MVS_GRAM_CH * pq + rw] = gv;
                }
            }
        }
        __syncthreads();
    } else
        for (int t = 0; t < rounds; ++t) round_body(t);
#else
    for (int t = 0; t < rounds; ++t) round_body(t);
#endif
    // lane 16 (v & 3) + (v >> 2): 1 / msd and the INCC of view v; then to view lane v
    const float inv_l = inv_msd(prm, ssd_l);
    const float inv0 = rlf(inv_l, 0);
    const float incc_v = 1.0f - (dot_l * (inv0 * inv_l)) * prm.inv_3sz;
    const int src = 4 * (16 * (wc.lane & 3) + ((wc.lane >> 2) & 15));
    incc_l = bperm_f(src, incc_v);
    if (ssd_out) *ssd_out = bperm_f(src, ssd_l);
    if (PIV) {
        const bool okk = (okm[0] >> (wc.lane & (MVS_VBITS - 1))) & 1u;
        const float a0 = bperm_f(src, mr_l), a1 = bperm_f(src, mg_l), a2 = bperm_f(src, mb_l);
        piv[0] = (wc.lane < MVS_VBITS && okk) ? a0 : 128.0f; piv[1] = (wc.lane < MVS_VBITS && okk) ? a1 : 128.0f; piv[2] = (wc.lane < MVS_VBITS && okk) ? a2 : 128.0f;
    }
}

// ------------------------------------------------------------------ candidate patch (registers)
// Uniform scalars plus view-lane arrays: lane j holds m_images[j], m_grids[j], m_vimages[j], m_vgrids[j].
struct Cand {
    F4 coord, normal;
    float ncc, dscale, ascale, tmp;
    int nimg, nvimg;
    int img, gx, gy;     // view lanes
    int vimg, vgx, vgy;  // view lanes
};

// Optim::computeUnits + computeWeights, optim.cpp:109-132, 942-948: returns the view-lane weight array
DEV float compute_weights(const DParams& prm, const WaveCtx& wc, F4 coord, F4 normal, int img, int n) {
    float unit = 1.0f;
    {
        const DView* vw = prm.views + (wc.lane < n ? img : 0);
        const F4 ctr = ld4(vw->center);
        const float ips = vw->ipscale;
        const F4 dc = sub4(coord, ctr);
        float u = 1.0f;  // get_unit
        if (ips != 0.0f) u = div_rn(2.0f * norm4(dc) * (float)(1 << prm.level), ips);
        const F4 ray = nrm4(sub4(ctr, coord));
        const float d = dot4(ray, normal);
        u = (0.0f < d) ? div_rn(u, d) : (float)(INT_MAX / 2);
        if (wc.lane < n) unit = u;
    }
    const float w0 = rlf(unit, 0);
    float w = fminf(1.0f, div_rn(w0, unit));
    if (wc.lane == 0) w = 1.0f;
    return w;
}

// Optim::computeINCC, optim.cpp:630-706
DEV float compute_incc(const DParams& prm, WaveCtx& wc, F4 coord, F4 normal, int img, int n, float weights, int robust) {
    if (n < 2) return 2.0f;
    const int ref = rli(img, 0);
    F4 px, py;
    get_paxes(prm, prm.views + ref, coord, normal, px, py);
    const int sz = min(prm.tau, n);
    wc.evals++;
    const Frame f = make_frame(prm, coord, px, py, normal, img, wc.lane < sz);
    vmask_t okm[1];
    float incc_l;
    eval_views(prm, wc, f, sz, okm, incc_l);
    if (!(okm[0] & 1u)) return 2.0f;
    const float val_l = robust ? robustincc(incc_l) : incc_l;
    float score = 0.0f, total = 0.0f;
    for (int i = 1; i < sz; ++i) {
        if (!((okm[0] >> i) & 1u)) continue;
        const float w = rlf(weights, i);
        total += w;
        score += rlf(val_l, i) * w;
    }
    if (total == 0.0f) return 2.0f;
    return div_rn(score, total);
}
// tail of Optim::computeINCC (optim.cpp:690-705) on per-view robust INCCs that are already known
DEV float weighted_incc(const DParams& prm, vmask_t okm, float val_l, float weights, int n) {
    if (n < 2 || !(okm & 1u)) return 2.0f;
    const int sz = min(prm.tau, n);
    float score = 0.0f, total = 0.0f;
    for (int i = 1; i < sz; ++i) {
        if (!((okm >> i) & 1u)) continue;
        const float w = rlf(weights, i);
        total += w;
        score += rlf(val_l, i) * w;
    }
    if (total == 0.0f) return 2.0f;
    return div_rn(score, total);
}
// PatchManager::computeNcc, patch_manager.cpp:401-404
DEV float compute_ncc(const DParams& prm, WaveCtx& wc, F4 coord, F4 normal, int img, int n) {
    const float w = compute_weights(prm, wc, coord, normal, img, n);
    return 1.0f - unrobustincc(compute_incc(prm, wc, coord, normal, img, n, w, 1));
}

// Optim::setINCCs (vector), optim.cpp:708-746: returns the view-lane INCC array (reference vs every view)
DEV float set_inccs(const DParams& prm, WaveCtx& wc, F4 coord, F4 normal, int img, int n, int robust, vmask_t* okm_out = nullptr,
                    float* texs = nullptr, int tstride = 0, float* ssd_out = nullptr) {
    const int ref = rli(img, 0);
    F4 px, py;
    get_paxes(prm, prm.views + ref, coord, normal, px, py);
    wc.evals++;
    const Frame f = make_frame(prm, coord, px, py, normal, img, wc.lane < n);
    vmask_t okm[1];
    float incc_l;
    eval_views(prm, wc, f, n, okm, incc_l, nullptr, texs, tstride, ssd_out);
    if (okm_out) *okm_out = okm[0];
    if (!(okm[0] & 1u)) return 2.0f;
    float incc = robust ? robustincc(incc_l) : incc_l;
    if (wc.lane >= MVS_LISTCAP || !((okm[0] >> (wc.lane & (MVS_VBITS - 1))) & 1u)) incc = 2.0f;
    if (wc.lane == 0) incc = 0.0f;
    return incc;
}

// compaction of view-lane arrays through LDS scratch: keeps lanes with `keep`, order preserved
DEV int compact1(int* scratch, const WaveCtx& wc, bool keep, int& a) {
    const unsigned long long m = ballot(keep);
    const int pos = __popcll(m & ((1ull << wc.lane) - 1ull));
    __syncthreads();
    if (keep) scratch[pos] = a;
    __syncthreads();
    a = scratch[wc.lane];
    return __popcll(m);
}

// Optim::addImages, optim.cpp:165-205
DEV void add_images(const DParams& prm, const WaveCtx& wc, int* scratch, Cand& c) {
    const int ref = rli(c.img, 0);
    bool visib = false;
    for (int j = 0; j < c.nimg; ++j) visib |= (rli(c.img, j) == wc.lane);
    bool q = false;
    if (wc.lane < prm.nviews && wc.lane != ref && !visib) {
        const DView* vw = prm.views + wc.lane;
        float P[12];
        load_P(vw, prm.level, P);
        const int W = vw->W[prm.level], H = vw->H[prm.level];
        const F4 ctr = ld4(vw->center);
        const F3 ic = project_regs(P, c.coord);
        const F4 ray = nrm4(sub4(ctr, c.coord));
        q = !(ic.x < 0.0f || W - 1 <= ic.x || ic.y < 0.0f || H - 1 <= ic.y) && prm.cosAngle0 <= dot4(ray, c.normal);
    }
    const unsigned long long m = ballot(q);
    const int pos = c.nimg + __popcll(m & ((1ull << wc.lane) - 1ull));
    __syncthreads();
    if (wc.lane < c.nimg) scratch[wc.lane] = c.img;
    if (q && pos < MVS_LISTCAP) scratch[pos] = wc.lane;
    __syncthreads();
    c.nimg = min(MVS_LISTCAP, c.nimg + (int)__popcll(m));
    c.img = scratch[wc.lane];
}

// Optim::constraintImages, optim.cpp:207-219
// keep_n > 0 (first constraintImages of postProcess inside the sweep): refinePatch left the final m_ncc to this
// evaluation -- computeINCC at the refined patch samples the first min(tau, keep_n) of these very textures -- so it is
// taken here, as the tail of computeINCC over the robust INCCs of those views with the weights refinePatch computed.
// Kept textures of an evaluation (postProcess: constraintImages hands them to setRefImage): where they lie in LDS, the sum
// of squares of each (view lanes, indexed like the list at the time of the evaluation), which views sampled, and for every
// entry of the list as it is NOW the index it had then (the list is compacted in between).
struct KeptTex {
    float* texs; int tstride;
    float ssd;       // view lanes (old index)
    vmask_t okm;     // bit k: old view k sampled
    int orig;        // view lanes (current index) -> old index
    int n_eval;      // length of the list at the time of the evaluation
};
DEV void constraint_images(const DParams& prm, WaveCtx& wc, int* scratch, Cand& c, float nccThreshold, float keep_w = 0.0f, int keep_n = 0,
                           KeptTex* kt = nullptr) {
    vmask_t okm = 0u;
    float ssd = 1.0f;
    const float inccs = set_inccs(prm, wc, c.coord, c.normal, c.img, c.nimg, 0, &okm, kt ? kt->texs : nullptr, kt ? kt->tstride : 0, kt ? &ssd : nullptr);
    if (keep_n > 0) c.ncc = 1.0f - unrobustincc(weighted_incc(prm, okm, robustincc(inccs), keep_w, keep_n));
    const bool keep = wc.lane == 0 || (wc.lane < c.nimg && inccs < 1.0f - nccThreshold);
    if (kt) {
        kt->ssd = ssd; kt->okm = okm; kt->orig = wc.lane; kt->n_eval = c.nimg;
        (void)compact1(scratch, wc, keep, kt->orig);
    }
    c.nimg = compact1(scratch, wc, keep, c.img);
}

// Optim::sortImages (isFixed = 1), optim.cpp:221-258
DEV void sort_images(const DParams& prm, const WaveCtx& wc, Cand& c) {
    F4 ray{0, 0, 0, 0};
    float unit = 0.0f;
    bool valid = false;
    {  // computeUnits(patch, indexes, units, rays), optim.cpp:86-107
        const bool in = wc.lane < c.nimg;
        const DView* vw = prm.views + (in ? c.img : 0);
        const F4 ctr = ld4(vw->center);
        const float ips = vw->ipscale;
        const F4 r = nrm4(sub4(ctr, c.coord));
        const float d = dot4(r, c.normal);
        float u = 1.0f;  // get_unit
        if (ips != 0.0f) u = div_rn(2.0f * norm4(sub4(c.coord, ctr)) * (float)(1 << prm.level), ips);
        valid = in && !(d <= 0.0f);
        if (in) ray = r;
        if (valid) unit = div_rn(u, d);
    }
    const unsigned long long vm = ballot(valid);
    const int n0 = __popcll(vm);
    if (n0 < 2) { c.nimg = 0; return; }
    const int first = __ffsll((long long)vm) - 1;
    if (wc.lane == first) unit = 0.0f;
    unsigned long long active = vm;
    int out = 0, k = 0;
    const float thr = prm.sortThreshold;
    while (active) {
        const bool act = (active >> wc.lane) & 1ull;
        const float m = wave_min(act ? unit : __int_as_float(0x7f800000));
        const unsigned long long eq = ballot(act && unit == m);
        const int sel = eq ? __ffsll((long long)eq) - 1 : __ffsll((long long)active) - 1;  // NaN guard: first remaining
        const int vsel = rli(c.img, sel);
        if (wc.lane == k) out = vsel;
        const F4 rsel{rlf(ray.x, sel), rlf(ray.y, sel), rlf(ray.z, sel), rlf(ray.w, sel)};
        active &= ~(1ull << sel);
        if (act && wc.lane != sel) {
            const float ftmp = fminf(thr, fmaxf(thr / 2.0f, 1.0f - dot4(rsel, ray)));
            unit = div_rn(unit * thr, ftmp);
        }
        ++k;
    }
    c.img = out;
    c.nimg = k;
}

// PatchManager::setScales, patch_manager.cpp:378-399
DEV void set_scales(const DParams& prm, const WaveCtx& wc, Cand& c) {
    const int ref = rli(c.img, 0);
    const DView* rv = prm.views + ref;
    const float unit = get_unit(prm, rv, c.coord);
    const float unit2 = 2.0f * unit;
    const F4 ray = nrm4(sub4(c.coord, ld4(rv->center)));
    const int num = min(prm.tau, c.nimg);
    float dn = 0.0f;
    {
        const bool in = wc.lane >= 1 && wc.lane < num;
        float P[12];
        load_P(prm.views + (in ? c.img : 0), prm.level, P);
        const float d = norm3(sub3(project_regs(P, c.coord), project_regs(P, sub4(c.coord, mul4(ray, unit2)))));
        if (in) dn = d;
    }
    float ds = c.dscale;
    for (int i = 1; i < num; ++i) ds += rlf(dn, i);
    ds = div_rn(ds, (float)(num - 1));
    ds = unit2 / ds;
    c.dscale = ds;
    c.ascale = pm_atanf(ds / (unit * (float)prm.wsize / 2.0f));
}

// PhotoSet::checkAngles, photoSet.cpp:77-103 (window test on cosines)
DEV int check_angles(const DParams& prm, const WaveCtx& wc, const Cand& c) {
    F4 ray{0, 0, 0, 0};
    if (wc.lane < c.nimg) ray = nrm4(sub4(ld4((prm.views + c.img)->center), c.coord));
    int count = 0;
    for (int j = 1; j < c.nimg; ++j) {
        const F4 rj{rlf(ray.x, j), rlf(ray.y, j), rlf(ray.z, j), rlf(ray.w, j)};
        const float d = fmaxf(-1.0f, fminf(1.0f, dot4(ray, rj)));
        count += __popcll(ballot(wc.lane < j && d < prm.cosMinAngle && prm.cosMaxAngle < d));
    }
    return count < 1 ? -1 : 0;
}

// Optim::preProcess, optim.cpp:137-163
STAGE int pre_process(const DParams& prm, WaveCtx& wc, int* scratch, Cand& c) {
    add_images(prm, wc, scratch, c);
    constraint_images(prm, wc, scratch, c, prm.nccThresholdBefore);
    sort_images(prm, wc, c);
    if (c.nimg > 0) set_scales(prm, wc, c);
    if (c.nimg < prm.minImageNum) return -1;
    if (check_angles(prm, wc, c) == -1) { c.nimg = 0; return -1; }
    return 0;
}

// ------------------------------------------------------------------ refinement
struct RefineCtx {
    F4 center, ray;
    float dscale, ascale;
    int ref;
};
// Optim::encode, optim.cpp:549-580
DEV void encode(const DParams& prm, const RefineCtx& rc, F4 coord, F4 normal, float* x) {
    x[0] = dot4(sub4(coord, rc.center), rc.ray) / rc.dscale;
    const DView* vw = prm.views + rc.ref;
    const F3 n3{normal.x, normal.y, normal.z};
    const float fx = dot3(ld3(vw->xaxis), n3), fy = dot3(ld3(vw->yaxis), n3), fz = dot3(ld3(vw->zaxis), n3);
    const float a2 = pm_asinf(fmaxf(-1.0f, fminf(1.0f, fy)));
    const float cosb = pm_cosf(a2);
    float a1;
    if (cosb == 0.0f) a1 = 0.0f;
    else {
        const float sina = fx / cosb, cosa = -fz / cosb;
        a1 = pm_acosf(fmaxf(-1.0f, fminf(1.0f, cosa)));
        if (sina < 0.0f) a1 = -a1;
    }
    x[1] = a1 / rc.ascale;
    x[2] = a2 / rc.ascale;
}
// Optim::decode, optim.cpp:582-599 (x0..x2 may differ per lane: frame lanes decode their own proposal)
DEV void decode(const DParams& prm, const RefineCtx& rc, float x0, float x1, float x2, F4& coord, F4& normal) {
    const float t = rc.dscale * x0;
    coord = {fma_(t, rc.ray.x, rc.center.x), fma_(t, rc.ray.y, rc.center.y), fma_(t, rc.ray.z, rc.center.z), fma_(t, rc.ray.w, rc.center.w)};
    const float angle1 = x1 * rc.ascale, angle2 = x2 * rc.ascale;
    float s1, c1, s2, c2;
    pm_sincosf(angle1, s1, c1);
    pm_sincosf(angle2, s2, c2);
    const float fx = s1 * c2, fy = s2, fz = -c1 * c2;
    const DView* vw = prm.views + rc.ref;
    normal = {fma_(vw->zaxis[0], fz, fma_(vw->yaxis[0], fy, vw->xaxis[0] * fx)),
              fma_(vw->zaxis[1], fz, fma_(vw->yaxis[1], fy, vw->xaxis[1] * fx)),
              fma_(vw->zaxis[2], fz, fma_(vw->yaxis[2], fy, vw->xaxis[2] * fx)), 0.0f};
}
// Optim::cost_func (optim.cpp:401-468) for up to four proposals at once: lane 16*g + i decodes proposal g
// (x0..x2 hold that lane's proposal), builds the patch axes and the frame of view i; then each proposal is
// evaluated in turn.  imgx = m_images replicated into every group of 16 lanes.
DEV void sum_of_group(const WaveCtx& wc, unsigned okm, float val_l, int g, int sz, double& ans, int& denom) {
    ans = 0.0;
    denom = 0;
    for (int i = 1; i < sz; ++i) {
        if (!((okm >> i) & 1u)) continue;
        ans += (double)rlf(val_l, 16 * g + i);
        denom++;
    }
}
DEV double rld(double x, int l) {
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)rli((int)(b & 0xffffffffll), l), hi = (unsigned)rli((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
DEV double cost_of_group(const DParams& prm, const WaveCtx& wc, unsigned okm, float val_l, int g, int sz, int minimum) {
    if (!(okm & 1u)) return 2.0;
    double ans;
    int denom;
    sum_of_group(wc, okm, val_l, g, sz, ans, denom);
    if (denom < minimum - 1) return 2.0;
    return ans / (double)denom;
}
DEV void cost_func4(const DParams& prm, WaveCtx& wc, const RefineCtx& rc, int imgx, int n, bool four, float x0, float x1, float x2,
                    double (&fv)[4], float* piv = nullptr, const ClsConst* cc = nullptr) {
    F4 coord, normal, px, py;
    decode(prm, rc, x0, x1, x2, coord, normal);
    get_paxes(prm, prm.views + rc.ref, coord, normal, px, py);
    const int sz = min(prm.tau, n);
    const int minimum = min(prm.minImageNum, sz);
    const int g = wc.lane >> 4, i = wc.lane & 15;
    const Frame f = make_frame(prm, coord, px, py, normal, imgx, g < (four ? 4 : 1) && i < sz);
    float incc_l;
    fv[1] = fv[2] = fv[3] = 2.0;
    if (four) {
        wc.evals += 4;
        unsigned okm[4];
        eval_steps4(prm, wc, *cc, f, sz, okm, incc_l);
        const float val_l = robustincc(incc_l);
        // the four means, lane j < 4 = proposal j: its views' robust INCCs come over from the frame lanes 16 j + i one by one
        // (ds_bpermute) and are added in the order of cost_func's loop (optim.cpp:451-465, i ascending, in double); one fp64
        // division for all four
        const int gl = wc.lane & 3;
        const unsigned okl = gl == 1 ? okm[1] : (gl == 2 ? okm[2] : (gl == 3 ? okm[3] : okm[0]));
        double num = 0.0;
        int den = 0;
        for (int i = 1; i < sz; ++i) {
            const float v = bperm_f(4 * (16 * gl + i), val_l);
            const bool ok = (okl >> i) & 1u;
            num = ok ? num + (double)v : num;
            den += ok ? 1 : 0;
        }
        const double q = num / (double)den;
#pragma unroll
        for (int j = 0; j < 4; ++j) fv[j] = ((okm[j] & 1u) && rli(den, j) >= minimum - 1) ? rld(q, j) : 2.0;
    } else {
        wc.evals += 1;
        vmask_t okm[1];
        if (piv) eval_views<true>(prm, wc, f, sz, okm, incc_l, piv);  // refinePatch's first evaluation: the view means become the pivots
        else eval_views(prm, wc, f, sz, okm, incc_l);
        const float val_l = robustincc(incc_l);
        fv[0] = cost_of_group(prm, wc, (unsigned)okm[0], val_l, 0, sz, minimum);  // sz <= tau <= 16 views
    }
}
// Optim::refinePatch, optim.cpp:480-547, BOBYQA replaced by the halving random search
// w_out != nullptr (inside the sweep): the weights go out and the final m_ncc is left to postProcess
STAGE void refine_patch(const DParams& prm, WaveCtx& wc, Cand& c, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, float* w_out = nullptr) {
    RefineCtx rc;
    rc.center = c.coord;
    rc.ref = rli(c.img, 0);
    rc.ray = nrm4(sub4(c.coord, ld4((prm.views + rc.ref)->center)));
    rc.dscale = c.dscale;
    rc.ascale = prm.ascaleConst;
    const float w = compute_weights(prm, wc, c.coord, c.normal, c.img, c.nimg);
    const int imgx = __shfl(c.img, wc.lane & 15);
    float x[3];
    encode(prm, rc, c.coord, c.normal, x);
    const float amin = -23.99999f, amax = 23.99999f;
    x[1] = fmaxf(fminf(x[1], amax), amin);
    x[2] = fmaxf(fminf(x[2], amax), amin);
    float bx0 = x[0], bx1 = x[1], bx2 = x[2];
    double fv[4];
    float piv[3] = {128.0f, 128.0f, 128.0f};  // view lanes: the mean colour of view k at the starting point
    cost_func4(prm, wc, rc, imgx, c.nimg, false, bx0, bx1, bx2, fv, piv);
    __syncthreads();
    if (wc.lane < 16) mvs_dyn_lds4[MVS_PIVOT_LDS4 + wc.lane] = make_float4(piv[0], piv[1], piv[2], 0.0f);
    __syncthreads();
    double fbest = fv[0];
    const ClsConst cc = wc.cc;
    float rd = prm.rd0, ra = prm.ra0;
    const int g = wc.lane >> 4;  // this lane's proposal: 0 depth only, 1 angles only, 2 both, 3 both mirrored about the step's start
    const uint32_t gj = (uint32_t)min(g, 2);
    for (int k = 0; k < prm.refine_steps; ++k) {
        const uint32_t draw = 16u + ((uint32_t)(k * 3) + gj) * 3u;
        const float u0 = 2.0f * rng_uniform(prm.seed, k0, k1, k2, k3, draw + 0);
        const float u1 = 2.0f * rng_uniform(prm.seed, k0, k1, k2, k3, draw + 1);
        const float u2 = 2.0f * rng_uniform(prm.seed, k0, k1, k2, k3, draw + 2);
        float cx0 = (gj == 1u) ? bx0 : fma_(u0, rd, bx0);
        float cx1 = (gj == 0u) ? bx1 : fmaxf(fminf(fma_(u1, ra, bx1), amax), amin);
        float cx2 = (gj == 0u) ? bx2 : fmaxf(fminf(fma_(u2, ra, bx2), amax), amin);
        if (g == 3) {
            cx0 = bx0 - (cx0 - bx0);
            cx1 = fmaxf(fminf(bx1 - (cx1 - bx1), amax), amin);
            cx2 = fmaxf(fminf(bx2 - (cx2 - bx2), amax), amin);
        }
        cost_func4(prm, wc, rc, imgx, c.nimg, true, cx0, cx1, cx2, fv, nullptr, &cc);
        int jb = 0;
        double fstep = fv[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) if (fv[j] < fstep) { fstep = fv[j]; jb = j; }
        if (fstep < fbest) { fbest = fstep; bx0 = rlf(cx0, 16 * jb); bx1 = rlf(cx1, 16 * jb); bx2 = rlf(cx2, 16 * jb); }
        rd *= 0.5f; ra *= 0.5f;
    }
    x[0] = bx0; x[1] = bx1; x[2] = bx2;
    decode(prm, rc, x[0], x[1], x[2], c.coord, c.normal);
    c.normal.w = 0.0f;
    if (w_out) *w_out = w;
    else c.ncc = 1.0f - unrobustincc(compute_incc(prm, wc, c.coord, c.normal, c.img, c.nimg, w, 1));
}

// ------------------------------------------------------------------ post-processing
// Optim::filterImagesByAngle, optim.cpp:325-346
DEV void filter_images_by_angle(const DParams& prm, const WaveCtx& wc, int* scratch, Cand& c, KeptTex* kt = nullptr) {
    bool bad = false;
    if (wc.lane < c.nimg) {
        const F4 ray = nrm4(sub4(ld4((prm.views + c.img)->center), c.coord));
        bad = dot4(ray, c.normal) < prm.cosAngle1;
    }
    const unsigned long long bm = ballot(bad);
    if (bm & 1ull) { c.nimg = 0; return; }
    if (kt) (void)compact1(scratch, wc, wc.lane < c.nimg && !bad, kt->orig);
    c.nimg = compact1(scratch, wc, wc.lane < c.nimg && !bad, c.img);
}
DEV void set_grids(const DParams& prm, const WaveCtx& wc, Cand& c) {
    c.gx = 0; c.gy = 0;
    if (wc.lane < c.nimg) cell_of(prm, prm.views + c.img, c.coord, c.gx, c.gy);
}
// Optim::setRefImage, optim.cpp:348-383 with Optim::setINCCs (matrix), optim.cpp:748-783.
// texs: LDS [LISTCAP][3][tstride] centred textures.  The V(V-1)/2 pair products get one lane each and are summed
// over the samples in the reference's sequential order (optim.cpp:605-607).
DEV int pair_index(int a, int b, int n) { return a * (2 * n - a - 1) / 2 + (b - a - 1); }  // a < b < n
// kt != nullptr (postProcess): the centred textures of these views are in LDS already -- the constraintImages just before
// sampled them at this very patch with this very reference view (engine schedule: they are not sampled a second time, the
// work counters do not count a second evaluation); entry i of the list is entry kt->orig of that evaluation.
// `pre` (Filter::filterExact, which has the frames of four patches made side by side): lane i < nimg holds the finished frame of view c.img
// -- made by make_frame from getPAxes of the FIRST view of the list, as below.
DEV void set_ref_image(const DParams& prm, WaveCtx& wc, float* texs, int tstride, Cand& c, const KeptTex* kt = nullptr, const Frame* pre = nullptr) {
    if (c.nimg == 0) return;
    const int n = c.nimg;
    const int ref = rli(c.img, 0);
    WC_T0(wc)
    vmask_t okmask = 0;
    float ssd_l = 1.0f;
    int orig = wc.lane;  // where view i's texture lies
    if (kt) {
        texs = kt->texs;
        orig = wc.lane < n ? kt->orig : 0;
        ssd_l = __shfl(kt->ssd, orig);
        okmask = vballot(wc.lane < n && ((kt->okm >> orig) & 1u));
        __syncthreads();
    } else {
    Frame f;
    if (pre) f = *pre;
    else {
        F4 px, py;
        get_paxes(prm, prm.views + ref, c.coord, c.normal, px, py);
        f = make_frame(prm, c.coord, px, py, c.normal, c.img, wc.lane < n);
    }
    wc.evals++;
    WC_ADD(wc, 1)
    // centred textures to LDS behind the frames this evaluation publishes, their ssd to view lanes
    texs += MVS_FRAME1_LDS_BYTES / 4;
    vmask_t okm1[1];
    float incc_unused;
    eval_views(prm, wc, f, n, okm1, incc_unused, nullptr, texs, tstride, &ssd_l);
    okmask = okm1[0];
    if (!(okmask & 1u)) wc.view_evals += (unsigned)vpop(okmask);  // Optim::setINCCs (matrix) samples every view, whatever the first one did
    }
    const float inv_l = inv_msd(prm, ssd_l);
    WC_ADD(wc, 2)
    __syncthreads();
    // The pair products sum(k) t_a[k] t_b[k] over the 3 wsz elements of two centred textures, k = 3 sample + channel, as ONE
    // k-ordered chain acc = fma(t_a[k], t_b[k], acc) (the reference sums sample by sample, optim.cpp:601-609) -- the order of a
    // f32 MFMA, which is bit for bit such a chain (one rounding per product, no wider accumulator).  The robust INCC of pair
    // q = pair_index(a, b) goes to LDS behind the textures.
#if !MVS_PAIR_MFMA
    const int npairs = n * (n - 1) / 2;
    float* pairv = texs + prm.list_n * 3 * tstride;  // behind the textures (list_n = min(MVS_LISTCAP, nviews): no list is longer)
#endif
#if MVS_PAIR_MFMA
    // The evaluation left the Gram matrix of its textures at texs[a * MVS_GRAM_LD + b] (eval_views).  Row by row, lane b > a turns
    // G[a][b] into the robust INCC of the pair (both triangles); `ne` = the list length of that evaluation, its views' 1 / msd and
    // sampled bits by evaluation-time index.
    {
        const int MVS_GRAM_LD = prm.gram_ld;  // the data set's list length rounded up to a chunk
        const int ne = kt ? kt->n_eval : n;
        const float inv_e = kt ? inv_msd(prm, kt->ssd) : inv_l;
        const vmask_t ok_e = kt ? kt->okm : okmask;
        const bool okb = wc.lane < ne && ((ok_e >> (wc.lane & (MVS_VBITS - 1))) & 1u);
        for (int a = 0; a + 1 < ne; ++a) {
            const float inva = rlf(inv_e, a);
            const bool oka = (ok_e >> a) & 1u;
            if (wc.lane > a && wc.lane < ne) {
                float val = robustincc(1.0f - (texs[a * MVS_GRAM_LD + wc.lane] * (inva * inv_e)) * prm.inv_3sz);
                if (!(oka && okb)) val = 2.0f;
                texs[a * MVS_GRAM_LD + wc.lane] = val;
                texs[wc.lane * MVS_GRAM_LD + a] = val;
            }
        }
    }
#else
    // one lane per pair (a, b), a < b < n: 120 pairs = 2 rounds of 64 lanes for 16 views
    for (int r = 0; r * 64 < npairs; ++r) {
        const int q0 = wc.lane + 64 * r;
        int q = q0;
        const bool act = q < npairs;
        int a = 0;
        if (act) { while (q >= n - 1 - a) { q -= n - 1 - a; ++a; } }
        const int b = act ? a + 1 + q : 1;
        const float* ta = texs + (__shfl(orig, a) * 3) * tstride;
        const float* tb = texs + (__shfl(orig, b) * 3) * tstride;
        float acc = 0.0f;
        for (int i = 0; i < 3 * prm.wsz; i += 3) {
            acc = fma_(ta[i], tb[i], acc); acc = fma_(ta[i + 1], tb[i + 1], acc); acc = fma_(ta[i + 2], tb[i + 2], acc);
        }
        const float inva = __shfl(inv_l, a), invb = __shfl(inv_l, b);
        float val = robustincc(1.0f - (acc * (inva * invb)) * prm.inv_3sz);
        if (!(act && ((okmask >> a) & 1u) && ((okmask >> b) & 1u))) val = 2.0f;
        if (act) pairv[q0] = val;
    }
#endif
    __syncthreads();
    WC_ADD(wc, 3)
    // view lane i: sum over j of inccs[i][j], j ascending (std::accumulate, optim.cpp:368)
    float acc = 0.0f;
#if MVS_PAIR_MFMA
    {
        const int MVS_GRAM_LD = prm.gram_ld;
        const int oi = wc.lane < n ? orig : 0;  // where view i sat in the evaluation whose matrix this is
        for (int j = 0; j < n; ++j) {
            const float v = texs[oi * MVS_GRAM_LD + rli(orig, j)];
            if (wc.lane < n && wc.lane != j) acc += v;
        }
    }
#else
    for (int j = 0; j < n; ++j) {
        const int i = min(wc.lane, n - 1);
        const int a = min(i, j), b = max(i, j);
        const int q = a < b ? pair_index(a, b, n) : 0;
        const float v = pairv[q];
        if (wc.lane < n && wc.lane != j) acc += v;
    }
#endif
    const float big = (float)(INT_MAX / 2);
    const bool cand = wc.lane < n && acc < big;
    const float m = wave_min(cand ? acc : __int_as_float(0x7f800000));
    const unsigned long long eq = ballot(cand && acc == m);
    WC_ADD(wc, 4)
    if (!eq) return;
    const int refindex = __ffsll((long long)eq) - 1;
    const int vref = rli(c.img, refindex), v0 = rli(c.img, 0);
    if (wc.lane == 0) c.img = vref;
    else if (wc.lane == refindex) c.img = v0;
}

// PatchManager::isVisible, patch_manager.cpp:335-376 (per view lane)
DEV int is_visible(const DParams& prm, const Cand& c, int image, int ix, int iy, float strict) {
    const DView* vw = prm.views + image;
    if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) return 0;
    if (prm.depth == 0) return 1;
    const unsigned long long dp = prm.dpgrid[vw->cell_base + iy * vw->gw + ix];
    if (dp == ~0ull) return 1;
    const DPatch* q = prm.pool + (uint32_t)(dp & 0xffffffffull);
    const F4 qc = ld4(q->coord);
    const F4 ray = nrm4(sub4(c.coord, ld4(vw->center)));
    const float diff = dot4(ray, sub4(c.coord, qc));
    // patch_manager.cpp:366-369: the factor is a float there (the double minimum narrowed; 2 + dot is exact in double, so this float sum is
    // the same value), and the product and the comparison run in float
    const float factor = fminf(2.0f, 2.0f + dot4(ray, c.normal));
    const float lhs = get_unit(prm, vw, c.coord) * (float)prm.csize * strict;
    return diff < lhs * factor ? 1 : 0;
}
// PatchManager::setVImagesVGrids, patch_manager.cpp:267-301
DEV void set_vimages_vgrids(const DParams& prm, const WaveCtx& wc, int* scratch, Cand& c) {
    bool visib = false;
    for (int j = 0; j < c.nimg; ++j) visib |= (rli(c.img, j) == wc.lane);
    for (int j = 0; j < c.nvimg; ++j) visib |= (rli(c.vimg, j) == wc.lane);
    bool q = false;
    int ix = 0, iy = 0;
    if (wc.lane < prm.nviews && !visib) {
        cell_of(prm, prm.views + wc.lane, c.coord, ix, iy);
        q = is_visible(prm, c, wc.lane, ix, iy, prm.neighborThreshold) != 0;
    }
    const unsigned long long m = ballot(q);
    const int pos = c.nvimg + __popcll(m & ((1ull << wc.lane) - 1ull));
    __syncthreads();
    if (wc.lane < c.nvimg) { scratch[wc.lane] = c.vimg; scratch[64 + wc.lane] = c.vgx; scratch[128 + wc.lane] = c.vgy; }
    if (q && pos < MVS_LISTCAP) { scratch[pos] = wc.lane; scratch[64 + pos] = ix; scratch[128 + pos] = iy; }
    __syncthreads();
    c.nvimg = min(MVS_LISTCAP, c.nvimg + (int)__popcll(m));
    c.vimg = scratch[wc.lane]; c.vgx = scratch[64 + wc.lane]; c.vgy = scratch[128 + wc.lane];
}
// PhotoSet::getMask(coord, level), photoSet.cpp:223-233
DEV int get_mask_all(const DParams& prm, const WaveCtx& wc, const Cand& c) {
    bool zero = false;
    if (wc.lane < prm.nviews) {
        const DView* vw = prm.views + wc.lane;
        if (vw->mask) {
            const F3 ic = project(vw, c.coord, prm.level);
            const int ix = (int)floorf(ic.x + 0.5f), iy = (int)floorf(ic.y + 0.5f);
            if (!(ix < 0 || vw->W[prm.level] <= ix || iy < 0 || vw->H[prm.level] <= iy))
                zero = vw->mask[(size_t)iy * vw->W[prm.level] + ix] == 0;
        }
    }
    return ballot(zero) ? 0 : -1;
}
DEV float score2(const Cand& c, float thr) { return fmaxf(0.0f, c.ncc - thr) * (float)c.nimg; }

// Optim::postProcess, optim.cpp:260-298 (Optim::check is applied by the caller, which owns the cell lists)
STAGE int post_process(const DParams& prm, WaveCtx& wc, int* scratch, float* texs, int tstride, Cand& c, float keep_w = 0.0f, bool keep = false) {
    if (c.nimg < prm.minImageNum) return -1;
    if (get_mask_all(prm, wc, c) == 0) return -1;
    const int keep_n = keep ? c.nimg : 0;
    add_images(prm, wc, scratch, c);
    // the textures of this evaluation stay in LDS (behind the frame region, which the evaluation itself uses) for setRefImage
    KeptTex kt;
    kt.texs = texs + MVS_FRAME1_LDS_BYTES / 4; kt.tstride = tstride; kt.ssd = 1.0f; kt.okm = 0u; kt.orig = wc.lane; kt.n_eval = 0;
    constraint_images(prm, wc, scratch, c, prm.nccThreshold, keep_w, keep_n, &kt);
    filter_images_by_angle(prm, wc, scratch, c, &kt);
    if (c.nimg < prm.minImageNum) return -1;
    set_grids(prm, wc, c);
    const int ref_before = rli(c.img, 0);
    set_ref_image(prm, wc, texs, tstride, c, &kt);
    // Same reference view as before: the second constraintImages would sample the very textures of the first one for
    // the views that passed it, under the same threshold, and remove nothing.  It runs when the reference changed.
    if (rli(c.img, 0) != ref_before) constraint_images(prm, wc, scratch, c, prm.nccThreshold);
    if (c.nimg < prm.minImageNum) return -1;
    set_grids(prm, wc, c);
    c.tmp = score2(c, prm.nccThreshold);
    if (prm.depth) set_vimages_vgrids(prm, wc, scratch, c);
    return 0;
}

// ------------------------------------------------------------------ record <-> registers
DEV void load_cand(const DPatch* p, const WaveCtx& wc, Cand& c) {
    c.coord = ld4(p->coord);
    c.normal = ld4(p->normal);
    c.ncc = p->ncc; c.dscale = p->dscale; c.ascale = p->ascale; c.tmp = p->tmp;
    c.nimg = min(p->nimages, MVS_LISTCAP);
    c.nvimg = min(p->nvimages, MVS_LISTCAP);
    c.img = (wc.lane < MVS_MAXI) ? (int)p->images[wc.lane] : 0;
    c.vimg = (wc.lane < MVS_MAXI) ? (int)p->vimages[wc.lane] : 0;
    c.gx = c.gy = c.vgx = c.vgy = 0;
}
DEV void store_cand(DPatch* p, const WaveCtx& wc, const Cand& c, int flags, int id) {
    if (wc.lane == 0) {
        p->coord[0] = c.coord.x; p->coord[1] = c.coord.y; p->coord[2] = c.coord.z; p->coord[3] = c.coord.w;
        p->normal[0] = c.normal.x; p->normal[1] = c.normal.y; p->normal[2] = c.normal.z; p->normal[3] = c.normal.w;
        p->ncc = c.ncc; p->dscale = c.dscale; p->ascale = c.ascale; p->tmp = c.tmp;
        p->nimages = c.nimg; p->nvimages = c.nvimg; p->flags = flags; p->id = id;
    }
    if (wc.lane < MVS_MAXI) {
        p->images[wc.lane] = (uint8_t)(wc.lane < c.nimg ? c.img : 0);
        p->vimages[wc.lane] = (uint8_t)(wc.lane < c.nvimg ? c.vimg : 0);
    }
}
DEV WaveCtx make_wave_ctx(const DParams& prm) {
    WaveCtx wc;
    wc.lane = lane_id();
    wc.cc = make_cls(prm, wc.lane);
    wc.evals = 0; wc.view_evals = 0;
#ifdef MVS_STAGE_TIMING
    for (int k = 0; k < 8; ++k) wc.st_acc[k] = 0;
    wc.st_t = 0;
#endif
    return wc;
}

// Propagate::generatePatch, propagate.cpp:220-237.  `src` is in registers (view lanes hold m_images).
// as_view >= 0 (view propagation): the patch is re-anchored on the ray of view `as_view` and Optim::swapImage
// (optim.cpp:385-395) makes that view the reference.
// need_ncc = false (engine schedule, destination cell not full): the initial score is only ever read by the
// replace-worst pre-filter (propagate.cpp:170); refinePatch overwrites it, so it is not computed when nothing reads it.
STAGE bool generate_patch(const DParams& prm, WaveCtx& wc, int* scratch, const Cand& src, F3 icoord, Cand& out, int as_view = -1, bool need_ncc = true) {
    int simg = src.img;
    if (as_view >= 0 && rli(src.img, 0) != as_view) {
        const unsigned long long hit = ballot(0 < wc.lane && wc.lane < src.nimg && src.img == as_view);
        if (hit == 0ull) return false;
        const int k = __ffsll((long long)hit) - 1, first = rli(src.img, 0);
        simg = wc.lane == 0 ? as_view : (wc.lane == k ? first : src.img);
    }
    const int image = rli(simg, 0);
    const DView* vw = prm.views + image;
    const float depth = dot4(ld4(vw->oaxis), src.coord);
    const F3 nic{depth * icoord.x, depth * icoord.y, depth * icoord.z};
    out.coord = unproject(vw, nic, prm.level);
    out.normal = src.normal;
    out.ncc = -1.0f; out.dscale = 0.0f; out.ascale = 0.0f; out.tmp = 0.0f;
    out.nvimg = 0; out.vimg = 0; out.vgx = 0; out.vgy = 0;
    // PatchManager::setGridsImages, patch_manager.cpp:223-239
    int ix = 0, iy = 0;
    bool keep = false;
    if (wc.lane < src.nimg) {
        const DView* v2 = prm.views + simg;
        cell_of(prm, v2, out.coord, ix, iy);
        keep = 0 <= ix && ix < v2->gw && 0 <= iy && iy < v2->gh;
    }
    out.img = simg; out.gx = ix; out.gy = iy;
    const unsigned long long m = ballot(keep);
    const int pos = __popcll(m & ((1ull << wc.lane) - 1ull));
    __syncthreads();
    if (keep) { scratch[pos] = out.img; scratch[64 + pos] = ix; scratch[128 + pos] = iy; }
    __syncthreads();
    out.nimg = __popcll(m);
    out.img = scratch[wc.lane]; out.gx = scratch[64 + wc.lane]; out.gy = scratch[128 + wc.lane];
    if (out.nimg == 0) return false;
    if (need_ncc) out.ncc = compute_ncc(prm, wc, out.coord, out.normal, out.img, out.nimg);
    return true;
}

}  // namespace mvsdev
