// mvs_engine.cpp -- host side of the C ABI in include/mvskit_engine.h: device memory, camera set-up,
// the per-pass schedule (index build -> sweep -> commit) on one HIP stream, HIP-event timing.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: librccl is opened at run time (see Rccl below)

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mvskit_engine.h"
#include "mvs_kernels.h"
#include "mvs_types.h"

static_assert(sizeof(mvs_patch) == sizeof(DPatch), "mvs_patch and DPatch must be the same bytes");
static_assert((MVS_LISTCAP == 16 || MVS_LISTCAP == 32 || MVS_LISTCAP == 64) && MVS_LISTCAP <= MVS_MAXI && MVS_MAX_IMAGES == MVS_MAXI, "limits out of sync");

namespace {
thread_local std::string g_err;

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            g_err = std::string(#expr) + ": " + hipGetErrorString(_e);                                     \
            return MVS_ERR_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    int64_t cap = 0;
    bool headroom = false;  // the cell indexes only: a buffer that has to grow a second time takes half as much again
    int ensure(int64_t n) {
        if (n <= cap) return MVS_OK;
        // (headroom) The cell indexes of a pool that grows from iteration to iteration: freeing and allocating 10 GB costs ~100 ms, and an
        // exact fit did it at every index build of a 48 x 4K run.  (A caller that knows how large its pool will get sizes the indexes
        // up front: mvs_engine_reserve.)  Every other buffer gets exactly what it asks for.
        const int64_t exact = std::max<int64_t>(n, 16);
        int64_t want = (p && headroom) ? exact + exact / 2 : exact;
        // The contents are never needed across a growth.  The new buffer is allocated BEFORE the old one is freed where both fit, so
        // that a failure leaves the old buffer in place (a failed mvs_engine_reserve must not take the indexes away); if they do not
        // fit side by side the old one goes first.
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, (size_t)want * sizeof(T));
        if (e != hipSuccess && want > exact) { (void)hipGetLastError(); want = exact; e = hipMalloc((void**)&q, (size_t)want * sizeof(T)); }  // no room for the headroom
        if (e != hipSuccess && p) {
            (void)hipGetLastError();
            (void)hipFree(p); p = nullptr; cap = 0;
            e = hipMalloc((void**)&q, (size_t)want * sizeof(T));
        }
        if (e != hipSuccess) { (void)hipGetLastError(); g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return MVS_ERR_HIP; }
        if (p) (void)hipFree(p);
        p = q; cap = want;
        return MVS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// RCCL, opened at run time.  A process that has loaded torch already holds torch's own librccl: that copy is reused
// (one RCCL per process); otherwise the ROCm installation's is opened.  Nothing here is needed on one GPU.
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // optional (mvs_engine_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;  // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // MVS_CCL_LIBRARY: another library with the same eight entry points (a site's own RCCL build; the shared-memory
    // loopback of tests/loopback_ccl, which lets several ranks share the one GPU of a test box)
    if (const char* over = getenv("MVS_CCL_LIBRARY")) {
        if (*over) { r.h = dlopen(over, RTLD_NOW | RTLD_LOCAL); if (!r.h) return r; }
    }
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char* paths[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* n : paths) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.h) return r;
#define MVS_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name))
    MVS_SYM(GetUniqueId, "ncclGetUniqueId"); MVS_SYM(CommInitRank, "ncclCommInitRank"); MVS_SYM(CommDestroy, "ncclCommDestroy");
    MVS_SYM(AllGather, "ncclAllGather"); MVS_SYM(Broadcast, "ncclBroadcast"); MVS_SYM(GroupStart, "ncclGroupStart");
    MVS_SYM(GroupEnd, "ncclGroupEnd"); MVS_SYM(GetErrorString, "ncclGetErrorString");
    MVS_SYM(CommCount, "ncclCommCount"); MVS_SYM(CommUserRank, "ncclCommUserRank");
#undef MVS_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd && r.GetErrorString;
    return r;
}
#define NCCLCHK(expr)                                                                                      \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            g_err = std::string(#expr) + ": " + rccl().GetErrorString(_r);                                 \
            return MVS_ERR_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

// roctx ranges (tracing) around upload / index build / colour-pass sweep / exchange / commit /
// Filter::run stages; libroctx64 is opened at run time and the ranges cost nothing when no profiler listens.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx& roctx() {
    static Roctx r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void* h = nullptr;
    const char* names[] = {"libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so.4"};
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return r;
    r.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    r.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!r.push || !r.pop) { r.push = nullptr; r.pop = nullptr; }
    return r;
}
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};
}  // namespace

struct mvs_engine {
    mvs_config cfg{};
    DParams prm{};
    hipStream_t stream = nullptr;
    bool have_views = false;
    std::vector<DView> hviews;
    DevBuf<DView> dviews;
    std::vector<uint32_t*> img_bufs;
    std::vector<uint8_t*> mask_bufs;
    int total_cells = 0;
    // pool
    DevBuf<DPatch> pool, pool_alt;  // pool_alt: target of the stable compaction done at every commit
    DevBuf<uint8_t> kill;
    int64_t pool_n = 0;
    bool ncc_dirty = false;
    // index
    DevBuf<int32_t> cnt, cursor, vcnt, vcursor;
    DevBuf<csr_off_t> start, vstart;       // list offsets (64-bit)
    DevBuf<csr_off_t> start_raw;           // m_pgrids offsets of a build with the trim, before the lists are packed end to end
    DevBuf<int32_t> id32_raw;              // ... and its ids, each list compacted to the front of its own range
    DevBuf<int64_t> scan_tmp;              // block sums of the scans (used as int32 or as csr_off_t)
    DevBuf<unsigned long long> ids;        // the (descending ncc, id) sort keys of an index build: transient, one buffer serves both grids
    DevBuf<ListKey> key;                   // (m_ncc, reference view) per m_pgrids entry
    DevBuf<int32_t> id32, vid32;           // the ids alone
    DevBuf<int32_t> uf_parent, uf_size;  // Filter::filterSmallGroups union-find
    DevBuf<int32_t> group_edges;         // its literal labelling: (root, root) pairs of the one-way edges between sets
    DevBuf<int32_t> cnt_alive, vcnt_alive;
    DevBuf<unsigned long long> dpgrid, best;
    bool index_valid = false;
    DevBuf<uint32_t> dirty;      // Filter::run: one bit per depth-map cell whose nearest patch the last stage removed
    bool dirty_marked = false;
    DevBuf<float4> geo;            // Filter::run: the packed geometry of the pool (DParams::geo), 2 words per patch
    DevBuf<uint8_t> geo_ref;       //              and the reference views
    bool geo_valid = false;        // between pack_geometry and the end of that Filter::run
    bool lists_dense[2] = {false, false};  // m_pgrids / m_vpgrids index built without the trim: the lists of neighbouring cells lie end to end
    // sweep / staging
    DevBuf<DPatch> staging;
    DevBuf<int32_t> job_stage, job_nstage, job_cnt, job_base_scan, kill_cnt, kill_base, per_view;
    DevBuf<unsigned long long> misc;  // [0] stage_counter, [1..2] fill_ncc evals, [4..7] small counters of the stages, [8..264) the trim's count in 256 parts
    DevBuf<DCounters> counters;
    DevBuf<int32_t> error_flag;
    DevBuf<int32_t> big_tables, retry_jobs;  // Optim::check's second tier (k_sweep_retry): 256 id sets of 16384 ints, the cells to run again
    int64_t retried_cells = 0;               // destination cells that went to the second tier since the engine was created
    int64_t pass_retried = 0;                // ... in the last pass
    SweepArgs sa{};
    bool staged = false;      // a pass has run and was not committed yet
    bool counted = false;     // commit_count + scans done for the staged pass
    int64_t n_new = 0, n_kill = 0;
    std::vector<int32_t> h_per_view;
    // probes / downloads
    DevBuf<DPatch> tmp_rec_in, tmp_rec_out;
    DevBuf<float> tmp_f_in, tmp_f_out;
    DevBuf<int32_t> tmp_i;
    DevBuf<uint8_t> tmp_bytes;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    mvs_timing timing{};
    mvs_filter_stats fstats{};
    int64_t fstats_exchange_bytes = 0;  // bytes this rank received in the last Filter::run's exchanges
    DevBuf<unsigned long long> fstat_buf;  // [1024][4] partial sums of Filter::filterNeighbor's work counts
    hipEvent_t fev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // multi-GPU (mvs_engine_comm_*): one RCCL communicator over the engines of the job
    ncclComm_t comm = nullptr;
    bool comm_owned = false;
    int comm_rank = 0, comm_world = 1;
    DevBuf<int64_t> comm_counts;   // [MVS_XCHG_WORDS] mine + [MVS_XCHG_WORDS * world] gathered
    DevBuf<int32_t> comm_kill_ids; // all ranks' kill ids
    int pass_status = MVS_OK;      // what the last mvs_engine_pass returned: the exchange carries it to the other ranks
    std::string pass_error;
    bool fault_fired = false;      // MVS_FAULT_PASS (fault injection, see pass_impl) fires once per engine
};

namespace {

int64_t total_cells_of(const mvs_engine* e) { return e->total_cells; }

void derive_params(mvs_engine* e) {  // PmMvps::init, pmmvps.cpp:32-36,54-67
    const mvs_config& c = e->cfg;
    DParams& p = e->prm;
    p.nviews = c.nviews; p.level = c.level; p.csize = c.csize; p.wsize = c.wsize; p.wsz = c.wsize * c.wsize;
    p.minImageNum = c.minImageNum;
    p.tau = std::min(c.minImageNum * 2, c.nviews);
    p.max_propag = c.max_propag;
    p.cap = c.max_propag * c.csize * c.csize;
    p.depth = c.depth; p.enable_check = c.enable_check; p.view_propagation = c.view_propagation ? 1 : 0;
    p.seed = c.seed; p.refine_steps = c.refine_steps; p.rd0 = c.refine_rd0; p.ra0 = c.refine_ra0;
    p.nccThreshold = c.nccThreshold;
    p.nccThresholdBefore = c.nccThreshold - 0.3f;
    // volatile: the libm calls below run at run time (the same glibc the oracle calls), not folded by the compiler
    volatile float a0 = (float)(60.0f * M_PI / 180.0f), a1 = (float)(60.0f * M_PI / 180.0f);
    volatile double amin = (double)c.maxAngleThreshold, amax = (double)a1;
    volatile float typo = (float)(120.0f / M_PI * 180.0f);
    volatile double d120 = 120.0f * M_PI / 180.0f, d10 = 10.0f * M_PI / 180.0f;
    p.cosAngle0 = cosf(a0);
    p.cosAngle1 = cosf(a1);
    p.cosMinAngle = (float)cos(amin);
    p.cosMaxAngle = (float)cos(amax);
    p.cosNeighborTypo = cosf(typo);  // pmmvps.cpp:124 (deg/rad typo kept)
    p.cosNeighbor120 = (float)cos(d120);
    p.sortThreshold = (float)(1.0f - cos(d10));
    p.ascaleConst = (float)(M_PI / 48.0f);
    p.inv_sz = 1.0f / (float)(c.wsize * c.wsize);
    p.inv_3sz = 1.0f / (float)(3 * c.wsize * c.wsize);
    p.neighborThreshold = 0.5f; p.neighborThreshold1 = 1.0f;
    p.quadThreshold = c.quadThreshold;
    p.list_n = std::min<int>(MVS_LISTCAP, c.nviews);
    p.gram_ld = (p.list_n + 15) / 16 * 16;
}

void invert3(const float* P, float* Minv) {  // Matrix3f::inverse (camera.cpp:304,335), in double
    double a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double inv[9] = {A, -(b * i - c * h), b * f - c * e, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * e - b * d};
    for (int k = 0; k < 9; ++k) Minv[k] = (float)(inv[k] / det);
}
inline float hfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
inline float hdot3(const float* a, const float* b) { return hfma(a[2], b[2], hfma(a[1], b[1], a[0] * b[0])); }
inline float hdot4(const float* a, const float* b) { return hfma(a[3], b[3], hfma(a[2], b[2], hfma(a[1], b[1], a[0] * b[0]))); }
inline void hcross(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// Camera::updateCamera (camera.cpp:65-100) + Optim::setAxesScales (optim.cpp:43-65)
void setup_camera(const mvs_engine* e, DView& vw, const float* P0) {
    const int maxLevel = e->cfg.level + 3;
    for (int k = 0; k < 12; ++k) vw.P[0][k] = P0[k];
    for (int l = 1; l < maxLevel; ++l) {
        for (int k = 0; k < 12; ++k) vw.P[l][k] = vw.P[l - 1][k];
        for (int k = 0; k < 8; ++k) vw.P[l][k] /= 2.0f;
    }
    invert3(vw.P[e->cfg.level], vw.Minv);
    const float* r2 = &vw.P[0][8];
    const float n = sqrtf(hdot3(r2, r2));
    for (int k = 0; k < 4; ++k) vw.oaxis[k] = r2[k] / n;
    {
        const float* P = vw.P[0];
        double a = P[0], b = P[1], c = P[2], d = P[4], ee = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
        double A = ee * i - f * h, B = -(d * i - f * g), C = d * h - ee * g;
        double det = a * A + b * B + c * C;
        double inv[9] = {A, -(b * i - c * h), b * f - c * ee, B, a * i - c * g, -(a * f - c * d), C, -(a * h - b * g), a * ee - b * d};
        double q[3] = {P[3], P[7], P[11]};
        vw.center[0] = (float)(-(inv[0] * q[0] + inv[1] * q[1] + inv[2] * q[2]) / det);
        vw.center[1] = (float)(-(inv[3] * q[0] + inv[4] * q[1] + inv[5] * q[2]) / det);
        vw.center[2] = (float)(-(inv[6] * q[0] + inv[7] * q[1] + inv[8] * q[2]) / det);
        vw.center[3] = 1.0f;
    }
    for (int k = 0; k < 3; ++k) vw.zaxis[k] = vw.oaxis[k];
    float xa[3] = {vw.P[0][0], vw.P[0][1], vw.P[0][2]};
    hcross(vw.zaxis, xa, vw.yaxis);
    const float yn = sqrtf(hdot3(vw.yaxis, vw.yaxis));
    for (int k = 0; k < 3; ++k) vw.yaxis[k] /= yn;
    hcross(vw.yaxis, vw.zaxis, vw.xaxis);
    const float x4[4] = {vw.xaxis[0], vw.xaxis[1], vw.xaxis[2], 0.0f}, y4[4] = {vw.yaxis[0], vw.yaxis[1], vw.yaxis[2], 0.0f};
    vw.ipscale = hdot4(&vw.P[0][0], x4) + hdot4(&vw.P[0][4], y4);
}

void free_views(mvs_engine* e) {
    for (uint32_t* p : e->img_bufs) if (p) (void)hipFree(p);
    for (uint8_t* p : e->mask_bufs) if (p) (void)hipFree(p);
    e->img_bufs.clear(); e->mask_bufs.clear();
    e->have_views = false;
}

DParams current_params(mvs_engine* e) {
    DParams p = e->prm;
    p.views = e->dviews.p;
    p.pool = e->pool.p;
    p.pool_n = e->pool_n;
    p.total_cells = e->total_cells;
    p.csr_start = e->start.p; p.csr_cnt = e->cnt_alive.p; p.csr_key = e->key.p; p.csr_id32 = e->id32.p;
    p.vcsr_start = e->vstart.p; p.vcsr_cnt = e->vcnt_alive.p; p.vcsr_id32 = e->vid32.p;
    p.dpgrid = e->dpgrid.p;
    p.geo = e->geo_valid ? e->geo.p : nullptr;
    p.geo_ref = e->geo_valid ? e->geo_ref.p : nullptr;
    return p;
}

bool want_vgrid(const mvs_engine* e) { return e->prm.depth >= 2 && e->prm.enable_check; }

// One cell index (m_pgrids or m_vpgrids): count -> scan -> fill -> per-cell sort (ncc desc, id asc) [-> trim to
// MAX_NUM_OF_PATCHES] -> compaction to alive entries, written out as id (and ListKey) streams.
// `unordered` (the rebuilds inside Filter::run, never with the trim): the entries of a list in no particular order, see k_index_fill_direct
int build_list(mvs_engine* e, bool vgrid, bool trim, bool unordered = false) {
    hipStream_t st = e->stream;
    const int64_t nc = e->total_cells;
    DParams p = current_params(e);
    DevBuf<int32_t>& cnt = vgrid ? e->vcnt : e->cnt;
    DevBuf<csr_off_t>& start = vgrid ? e->vstart : e->start;
    DevBuf<int32_t>& cursor = vgrid ? e->vcursor : e->cursor;
    DevBuf<unsigned long long>& ids = e->ids;
    DevBuf<int32_t>& id32 = vgrid ? e->vid32 : e->id32;
    DevBuf<int32_t>& cnt_alive = vgrid ? e->vcnt_alive : e->cnt_alive;
    HIPCHK(hipMemsetAsync(cnt.p, 0, (size_t)(nc + 1) * sizeof(int32_t), st));
    mvsk_index_count(p, vgrid ? nullptr : cnt.p, vgrid ? cnt.p : nullptr, nullptr, st);
    const bool direct = unordered && !trim;
    DevBuf<csr_off_t>& raw = e->start_raw;  // the offsets before the trim (ordered lists): the scan writes them where the sort wants them
    csr_off_t* const first_scan = direct ? start.p : raw.p;
    mvsk_exclusive_scan_off(cnt.p, first_scan, nc, reinterpret_cast<csr_off_t*>(e->scan_tmp.p), st);
    csr_off_t tot64 = 0;  // the number of memberships = the scan's last element (a per-wave atomicAdd to one counter cost the count a third of its time)
    HIPCHK(hipMemcpyAsync(&tot64, first_scan + nc, sizeof tot64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t tot = (int64_t)tot64;
    if (!unordered) { if (int r = ids.ensure(tot + 16)) return r; }
    if (!vgrid && !unordered) { if (int r = e->key.ensure(tot + 16)) return r; }
    if (int r = id32.ensure(tot + 16)) return r;
    p = current_params(e);
    HIPCHK(hipMemsetAsync(cursor.p, 0, (size_t)(nc + 1) * sizeof(int32_t), st));
    if (direct) {
        mvsk_index_fill_direct(p, vgrid ? 1 : 0, start.p, cursor.p, id32.p, st);
        std::swap(cnt.p, cnt_alive.p);  // every entry is alive: the counts ARE the alive counts (two buffers of one size; the next build clears its own)
        std::swap(cnt.cap, cnt_alive.cap);
        e->lists_dense[vgrid ? 1 : 0] = true;
        return MVS_OK;
    }
    // sorted lists [and the trim]: keys -> per-cell sort -> [trim] -> each list's alive entries to the front of its range -> the lists
    // packed end to end (a second scan, over the alive counts), so that every index the engine builds is dense
    if (int r = e->id32_raw.ensure(tot + 16)) return r;
    mvsk_index_fill(p, vgrid ? 1 : 0, raw.p, cursor.p, ids.p, st);
    mvsk_index_sort_trim(p, raw.p, ids.p, trim ? 1 : 0, e->misc.p + 8, st);  // [8 .. 264): the trim's count as 256 partial sums
    mvsk_index_finalize(p, vgrid ? 1 : 0, raw.p, ids.p, e->id32_raw.p, cnt_alive.p, st);
    mvsk_exclusive_scan_off(cnt_alive.p, start.p, nc, reinterpret_cast<csr_off_t*>(e->scan_tmp.p), st);
    mvsk_index_pack(p, raw.p, start.p, cnt_alive.p, ids.p, e->id32_raw.p, vgrid ? nullptr : e->key.p, id32.p, st);
    e->lists_dense[vgrid ? 1 : 0] = true;
    return MVS_OK;
}
int build_depth(mvs_engine* e) {  // m_dpgrids from the alive pool
    const DParams p = current_params(e);
    HIPCHK(hipMemsetAsync(e->dpgrid.p, 0xff, (size_t)e->total_cells * sizeof(unsigned long long), e->stream));
    mvsk_depth_maps(p, e->dpgrid.p, nullptr, e->stream);
    return MVS_OK;
}
// Index build of a propagation pass: seed scores, m_pgrids lists with the MAX_NUM_OF_PATCHES trim, m_vpgrids lists
// (needed by Optim::check), depth maps.
int build_index(mvs_engine* e, unsigned long long* trimmed_out) {
    hipStream_t st = e->stream;
    const DParams p = current_params(e);
    HIPCHK(hipMemsetAsync(e->misc.p + 1, 0, 3 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(e->misc.p + 8, 0, 256 * sizeof(unsigned long long), st));
    // PatchManager::sortPatches re-scores every patch whose m_ncc < 0 each time it meets it
    // (patch_manager.cpp:411-415); a wave that finds m_ncc >= 0 exits at once.
    mvsk_fill_ncc(p, e->misc.p + 1, st);
    e->ncc_dirty = false;
    if (int r = build_list(e, false, true)) return r;
    // m_vpgrids: no reader depends on the order inside its lists (findNeighbors' set, filterSmallGroups' unions): written without keys and sort
    if (want_vgrid(e)) if (int r = build_list(e, true, false, true)) return r;
    if (int r = build_depth(e)) return r;
    if (trimmed_out) {
        unsigned long long part[256];
        HIPCHK(hipMemcpyAsync(part, e->misc.p + 8, sizeof part, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        *trimmed_out = 0;
        for (unsigned long long v : part) *trimmed_out += v;
    }
    HIPCHK(hipGetLastError());
    e->index_valid = true;
    return MVS_OK;
}
// Filter::setDepthMapsVGridsVPGridsAddPatchV, filter.cpp:628-655
// setDepthMapsVGridsVPGridsAddPatchV between the stages of Filter::run.  The depth maps and m_vimages come from the pool; the two
// grid indexes are built only for a stage that walks them: filterOutside (computeGain) reads m_pgrids, filterExact neither,
// filterNeighbor and filterSmallGroups both -- and after the last stage the pool is compacted, which invalidates them anyway.
// Filter::run over the ranks of a multi-GPU job: every stage is a per-patch kernel on a snapshot, so rank r runs it on the patches
// [pool_n r / N, pool_n (r + 1) / N) and the ranks then hand each other what the stage wrote -- its kill bytes, and the records
// themselves where it rewrote lists (filterExact: m_images; setVImagesVGrids: m_vimages) -- as in-place broadcasts of the ranges
// (an all-gather-v without staging; ranges are contiguous and every rank computes the same bounds).  What works on the whole pool
// at once stays replicated: the grid indexes, the depth maps, filterSmallGroups' union-find.  One rank: the whole pool, no exchange.
void filter_range(const mvs_engine* e, int64_t& first, int64_t& last) {
    first = 0; last = e->pool_n;
    if (e->comm && e->comm_world > 1) {
        first = e->pool_n * e->comm_rank / e->comm_world;
        last = e->pool_n * (e->comm_rank + 1) / e->comm_world;
    }
}
// A failure on ONE rank inside the collective Filter::run (an allocation that does not fit there, a HIP error) must not leave the
// others waiting in the next broadcast.  Every step of the call goes through FilterRun: after a local failure the rank does no more
// work of its own and only keeps the appointments -- an agreement (one int64 per rank, all-gathered) stands in front of EVERY
// collective of the call, so the next collective any rank reaches is an agreement, in which all ranks see the first failed rank's
// status, give the call up together and return that status (the scheme of mvs_engine_exchange's status word).
struct FilterRun {
    mvs_engine* e;
    int local = MVS_OK;   // first failure on this rank
    std::string err;
    int agreed = MVS_OK;  // != MVS_OK: the ranks have agreed to give the call up with this status
    bool multi() const { return e->comm && e->comm_world > 1; }
    bool live() const { return local == MVS_OK && agreed == MVS_OK; }
    void note(int r) { if (r != MVS_OK && local == MVS_OK) { local = r; err = g_err; } }
};
// runs a step (an expression returning an mvs_status) unless a failure is pending
#define FR(expr) do { if (fr.live()) fr.note(expr); } while (0)
int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return MVS_OK;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return MVS_ERR_HIP;
}
#define FRHIP(expr) FR(hip_status((expr), #expr))
#ifdef MVS_FAULT_INJECTION
// test builds only (libmvskit_engine_faultinj.so): MVS_FAULT_FILTER=<rank>:<point> makes that rank fail at that point of Filter::run
void filter_fault_point(FilterRun& fr, int point) {
    const char* f = getenv("MVS_FAULT_FILTER");
    int r = -1, p = -1;
    if (f && fr.live() && sscanf(f, "%d:%d", &r, &p) == 2 && r == fr.e->comm_rank && p == point) {
        g_err = "mvs_engine_filter: failure injected by MVS_FAULT_FILTER";
        fr.note(MVS_ERR_CAPACITY);
    }
}
#else
inline void filter_fault_point(FilterRun&, int) {}
#endif
// the agreement: returns MVS_OK when every rank is fine, else the status all ranks give the call up with
int filter_agree(FilterRun& fr) {
    mvs_engine* e = fr.e;
    if (fr.agreed != MVS_OK) return fr.agreed;
    if (!fr.multi()) {
        if (fr.local != MVS_OK) { fr.agreed = fr.local; g_err = fr.err; }
        return fr.agreed;
    }
    // comm_counts was sized before the first collective of the call; without it not even a status can be sent
    const int64_t mine = fr.local;
    std::vector<int64_t> all((size_t)e->comm_world, 0);
    hipStream_t st = e->stream;
    if (hipMemcpyAsync(e->comm_counts.p, &mine, sizeof mine, hipMemcpyHostToDevice, st) != hipSuccess ||
        rccl().AllGather(e->comm_counts.p, e->comm_counts.p + 1, 1, ncclInt64, e->comm, st) != ncclSuccess ||
        hipMemcpyAsync(all.data(), e->comm_counts.p + 1, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        g_err = "mvs_engine_filter: the status all-gather itself failed (the communicator is unusable; the job must be torn down)";
        fr.agreed = MVS_ERR_HIP;
        return fr.agreed;
    }
    for (int r = 0; r < e->comm_world; ++r) {
        if (all[r] == MVS_OK) continue;
        fr.agreed = (int)all[r];
        if (r == e->comm_rank) g_err = fr.err.empty() ? std::string("mvs_engine_filter: this rank failed") : fr.err;
        else g_err = "mvs_engine_filter: rank " + std::to_string(r) + " reported status " + std::to_string((int)all[r]) + "; all ranks give the call up";
        break;
    }
    return fr.agreed;
}
// in-place broadcasts of every rank's share of the kill bytes and / or records (an all-gather-v without staging); returns != 0 when
// the call is given up
int filter_exchange(FilterRun& fr, bool kills, bool records) {
    mvs_engine* e = fr.e;
    if (!fr.multi()) return filter_agree(fr);
    if (int a = filter_agree(fr)) return a;
    if (e->pool_n == 0) return MVS_OK;
    const Rccl& R = rccl();
    hipStream_t st = e->stream;
    const int N = e->comm_world;
    ncclResult_t first_err = R.GroupStart();
    for (int r = 0; r < N && first_err == ncclSuccess; ++r) {
        const int64_t lo = e->pool_n * r / N, hi = e->pool_n * (r + 1) / N;
        if (hi <= lo) continue;
        if (kills) first_err = R.Broadcast(e->kill.p + lo, e->kill.p + lo, (size_t)(hi - lo), ncclUint8, r, e->comm, st);
        if (records && first_err == ncclSuccess)
            first_err = R.Broadcast(e->pool.p + lo, e->pool.p + lo, (size_t)(hi - lo) * sizeof(DPatch), ncclUint8, r, e->comm, st);
        e->fstats_exchange_bytes += (r == e->comm_rank) ? 0 : (kills ? (hi - lo) : 0) + (records ? (hi - lo) * (int64_t)sizeof(DPatch) : 0);
    }
    const ncclResult_t end = R.GroupEnd();
    if (first_err == ncclSuccess) first_err = end;
    if (first_err != ncclSuccess) {  // the transport itself: nothing can be agreed on any more
        g_err = std::string("Filter::run exchange: ") + R.GetErrorString(first_err);
        fr.note(MVS_ERR_HIP); fr.agreed = MVS_ERR_HIP;
        return fr.agreed;
    }
    return MVS_OK;
}
// Filter::filterSmallGroups with the reference's own labelling (filter.cpp:432-524; mvs_config.literal_groups): a breadth-first search
// in patch order over the DIRECTED relation "q is listed in the 3x3 cells around p in p's reference view and isNeighbor(p, q)" -- the
// first unlabelled patch claims everything it reaches.  On the GPU: (1) the sets joined by edges that run BOTH ways (union-find;
// inside such a set everything reaches everything, so the search claims a set whole or not at all, and the first unlabelled patch
// is always the smallest index of its set = its root); (2) the edges that are left between different sets, as (root, root) pairs --
// a few thousand; (3) on the host, the search over that condensed graph, sets in the order of their roots; (4) the size of its
// literal group written over every set's size, and the removal as usual.
int literal_small_groups(mvs_engine* e, int threshold) {
    hipStream_t st = e->stream;
    const DParams p = current_params(e);
    const int cap = 8 << 20;  // (root, root) pairs
    if (int r = e->group_edges.ensure(2 * (int64_t)cap)) return r;
    int32_t* nedges = reinterpret_cast<int32_t*>(e->misc.p + 7);
    HIPCHK(hipMemsetAsync(nedges, 0, sizeof(unsigned long long), st));
    mvsk_groups_literal_edges(p, e->uf_parent.p, e->uf_size.p, e->group_edges.p, nedges, cap, st);
    int32_t ne = 0;
    HIPCHK(hipMemcpyAsync(&ne, nedges, sizeof ne, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (ne > cap) { g_err = "Filter::filterSmallGroups (literal labelling): more than 8 M one-way edges between sets"; return MVS_ERR_CAPACITY; }
    if (ne > 0) {
        std::vector<int32_t> pairs(2 * (size_t)ne);
        HIPCHK(hipMemcpy(pairs.data(), e->group_edges.p, pairs.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        // the sets that have such an edge, in the order of their roots
        std::vector<int32_t> nodes(pairs);
        std::sort(nodes.begin(), nodes.end());
        nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
        const size_t nn = nodes.size();
        auto idx = [&](int32_t root) { return (size_t)(std::lower_bound(nodes.begin(), nodes.end(), root) - nodes.begin()); };
        std::vector<int32_t> sizes(nn);
        if (int r = e->tmp_i.ensure((int64_t)2 * nn + 16)) return r;
        HIPCHK(hipMemcpy(e->tmp_i.p, nodes.data(), nn * sizeof(int32_t), hipMemcpyHostToDevice));
        mvsk_gather_i32(e->uf_size.p, e->tmp_i.p, e->tmp_i.p + nn, (int64_t)nn, st);
        HIPCHK(hipMemcpyAsync(sizes.data(), e->tmp_i.p + nn, nn * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<std::vector<uint32_t>> adj(nn);
        for (int32_t k = 0; k < ne; ++k) adj[idx(pairs[2 * k])].push_back((uint32_t)idx(pairs[2 * k + 1]));
        std::vector<int32_t> label(nn, -1), gsize;
        std::vector<uint32_t> queue;
        for (size_t r0 = 0; r0 < nn; ++r0) {  // ascending roots = ascending first patches
            if (label[r0] != -1) continue;
            const int32_t gid = (int32_t)gsize.size();
            int64_t total = 0;
            queue.assign(1, (uint32_t)r0);
            label[r0] = gid;
            for (size_t qh = 0; qh < queue.size(); ++qh) {
                const uint32_t u = queue[qh];
                total += sizes[u];
                for (uint32_t v : adj[u]) if (label[v] == -1) { label[v] = gid; queue.push_back(v); }
            }
            gsize.push_back((int32_t)std::min<int64_t>(total, INT32_MAX));
        }
        std::vector<int32_t> newsize(nn);
        for (size_t u = 0; u < nn; ++u) newsize[u] = gsize[label[u]];
        HIPCHK(hipMemcpy(e->tmp_i.p + nn, newsize.data(), nn * sizeof(int32_t), hipMemcpyHostToDevice));
        mvsk_scatter_i32(e->uf_size.p, e->tmp_i.p, e->tmp_i.p + nn, (int64_t)nn, st);
    }
    mvsk_groups_kill(p, e->uf_parent.p, e->uf_size.p, threshold, e->kill.p, st);
    return MVS_OK;
}
// `incr` (after a stage's removals, marked by apply_kills): 1 = the depth maps are brought up to date in the marked cells only,
// 2 = and so is m_vimages -- allowed when the stage removed patches and left the lists of the others alone.
int rebuild_head(mvs_engine* e, int additive, bool need_pgrid, int incr) {
    if (need_pgrid) { if (int r = build_list(e, false, false, true)) return r; }
    if (incr && e->dirty_marked) mvsk_depth_maps(current_params(e), e->dpgrid.p, e->dirty.p, e->stream);
    else { incr = 0; if (int r = build_depth(e)) return r; }
    e->dirty_marked = false;
    int64_t first, last;
    filter_range(e, first, last);
    mvsk_filter_vimages(current_params(e), additive, first, last, incr == 2 && additive ? e->dirty.p : nullptr, e->stream);
    return MVS_OK;
}
int rebuild_tail(mvs_engine* e, bool need_vpgrid) {
    if (need_vpgrid) { if (int r = build_list(e, true, false, true)) return r; }
    HIPCHK(hipGetLastError());
    return MVS_OK;
}
// Filter::run never moves a patch, so what its stages read of the patches they MEET -- position, normal, depth scale, score, and (in
// filterOutside, before filterExact picks reference views anew) the reference view -- is packed once per call: 32 bytes + 1 per patch
// instead of a 96-byte record (192 in the 64-view build) that straddles cache lines.  filterNeighbor met ~40 KB of records per patch and ran at
// the HBM rate (5.7 TB/s, profiles/r04_pmc_traffic_per_kernel_filter_run.csv).  A pool with a record whose coord.w / normal.w are not
// 1 / 0 (only a caller's own seeds can be), or no memory for the copy: the stages read the records, as they did.
int pack_geometry(mvs_engine* e) {
    e->geo_valid = false;
    if (e->pool_n <= 0) return MVS_OK;
#ifdef MVS_FAULT_INJECTION  // test builds only: MVS_FAULT_NOPACK=1 keeps Filter::run on the records, so a test can compare the two paths
    if (const char* f = getenv("MVS_FAULT_NOPACK")) { if (atoi(f)) return MVS_OK; }
#endif
    if (e->geo.ensure(2 * e->pool_n) != MVS_OK || e->geo_ref.ensure(e->pool_n) != MVS_OK) { (void)hipGetLastError(); return MVS_OK; }
    hipStream_t st = e->stream;
    int32_t* bad = reinterpret_cast<int32_t*>(e->misc.p + 5);
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(unsigned long long), st));
    mvsk_geo_pack(e->pool.p, e->pool_n, e->geo.p, e->geo_ref.p, bad, st);
    int32_t hbad = 1;
    HIPCHK(hipMemcpyAsync(&hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    e->geo_valid = hbad == 0;
    return MVS_OK;
}
// returns != 0 when the call is given up (all ranks alike)
int filter_rebuild(FilterRun& fr, int additive, bool need_pgrid, bool need_vpgrid, int incr = 0) {
    FR(rebuild_head(fr.e, additive, need_pgrid, incr));
    filter_fault_point(fr, 2);
    if (int a = filter_exchange(fr, false, true)) return a;  // m_vimages of the other ranks' patches
    FR(rebuild_tail(fr.e, need_vpgrid));
    return MVS_OK;
}
// counts and applies the kill flags a filter stage has set
// `mark` (inside Filter::run, the depth maps being those of the pool as it stands): the cells that name a patch about to be
// removed are emptied and marked, for filter_rebuild's incremental passes
int apply_kills(mvs_engine* e, int64_t* removed, bool mark = false) {
    hipStream_t st = e->stream;
    *removed = 0;
    e->dirty_marked = false;
    if (e->pool_n == 0) return MVS_OK;
    if (mark) {
        const int64_t words = (e->total_cells + 31) / 32;
        if (int r = e->dirty.ensure(words + 1)) return r;
        HIPCHK(hipMemsetAsync(e->dirty.p, 0, (size_t)words * sizeof(uint32_t), st));
        mvsk_depth_mark_dirty(current_params(e), e->kill.p, e->dpgrid.p, e->dirty.p, st);
        e->dirty_marked = true;
    }
    mvsk_kill_count(e->kill.p, e->pool_n, e->kill_cnt.p, st);
    mvsk_exclusive_scan(e->kill_cnt.p, e->kill_base.p, e->pool_n, reinterpret_cast<int32_t*>(e->scan_tmp.p), st);
    int32_t nk = 0;
    HIPCHK(hipMemcpyAsync(&nk, e->kill_base.p + e->pool_n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    mvsk_apply_kill_flags(e->pool.p, e->kill.p, e->pool_n, st);
    HIPCHK(hipStreamSynchronize(st));
    *removed = nk;
    return MVS_OK;
}

int ensure_counts(mvs_engine* e) {  // commit_count + scans for the staged pass
    if (e->counted) return MVS_OK;
    hipStream_t st = e->stream;
    const int64_t nj = e->sa.njobs;
    e->n_new = 0; e->n_kill = 0;
    e->h_per_view.assign(e->cfg.nviews, 0);
    if (nj > 0) {
        mvsk_commit_count(e->sa, e->job_cnt.p, st);
        mvsk_exclusive_scan(e->job_cnt.p, e->job_base_scan.p, nj, reinterpret_cast<int32_t*>(e->scan_tmp.p), st);
        std::vector<int32_t> bounds(e->sa.nsweep_views + 1);
        for (int s = 0; s < e->sa.nsweep_views; ++s)
            HIPCHK(hipMemcpyAsync(&bounds[s], e->job_base_scan.p + e->sa.job_base[s], sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&bounds[e->sa.nsweep_views], e->job_base_scan.p + nj, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        e->n_new = bounds[e->sa.nsweep_views];
        for (int s = 0; s < e->sa.nsweep_views; ++s) e->h_per_view[e->sa.sweep_views[s]] = bounds[s + 1] - bounds[s];
    }
    if (e->pool_n > 0) {
        mvsk_kill_count(e->kill.p, e->pool_n, e->kill_cnt.p, st);
        mvsk_exclusive_scan(e->kill_cnt.p, e->kill_base.p, e->pool_n, reinterpret_cast<int32_t*>(e->scan_tmp.p), st);
        int32_t nk = 0;
        HIPCHK(hipMemcpyAsync(&nk, e->kill_base.p + e->pool_n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        e->n_kill = nk;
    }
    e->counted = true;
    return MVS_OK;
}

// Stable compaction of the pool: dead records (evicted, trimmed) are dropped, the relative order of the
// survivors -- the only thing ids are used for -- is kept.  Runs at every commit, on every rank alike.
int compact_pool(mvs_engine* e) {
    hipStream_t st = e->stream;
    if (e->pool_n == 0) return MVS_OK;
    mvsk_alive_count(e->pool.p, e->pool_n, e->kill_cnt.p, st);
    mvsk_exclusive_scan(e->kill_cnt.p, e->kill_base.p, e->pool_n, reinterpret_cast<int32_t*>(e->scan_tmp.p), st);
    int32_t alive = 0;
    HIPCHK(hipMemcpyAsync(&alive, e->kill_base.p + e->pool_n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (alive == e->pool_n) return MVS_OK;
    mvsk_alive_gather(e->pool.p, e->pool_n, e->kill_base.p, e->pool_alt.p, e->pool_alt.cap, st);
    std::swap(e->pool.p, e->pool_alt.p);
    std::swap(e->pool.cap, e->pool_alt.cap);
    e->pool_n = alive;
    return MVS_OK;
}

void add_counters(mvs_counters& a, const mvs_counters& b) {
    a.candidates += b.candidates; a.prefiltered += b.prefiltered; a.patches += b.patches; a.fail0 += b.fail0; a.fail1 += b.fail1;
    a.inserted += b.inserted; a.replaced += b.replaced; a.evals += b.evals; a.view_evals += b.view_evals; a.trimmed += b.trimmed;
}

}  // namespace

extern "C" {

const char* mvs_last_error(void) { return g_err.c_str(); }

int mvs_list_cap(void) { return MVS_LISTCAP; }
int mvs_patch_bytes(void) { return (int)sizeof(mvs_patch); }

int mvs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mvs_default_config(mvs_config* c) {  // Option::Option, option.cpp:19-33
    memset(c, 0, sizeof *c);
    c->level = 1; c->csize = 2; c->wsize = 7; c->minImageNum = 3; c->max_propag = 2;
    c->nccThreshold = 0.7f; c->maxAngleThreshold = (float)(10.0f * M_PI / 180.0f); c->quadThreshold = 2.5f;
    c->depth = 1; c->seed = 1; c->refine_steps = 6; c->refine_rd0 = 4.0f; c->refine_ra0 = 4.0f; c->enable_check = 1;
    c->view_begin = 0; c->view_stride = 1; c->device = 0; c->max_patches = 0;
}

int mvs_engine_create(const mvs_config* cfg, mvs_engine** out) {
    if (!cfg || !out) { g_err = "mvs_engine_create: null argument"; return MVS_ERR_ARG; }
    if (cfg->shard_count > 1 && (cfg->shard_index < 0 || cfg->shard_index >= cfg->shard_count)) { g_err = "mvs_engine_create: bad shard_index"; return MVS_ERR_ARG; }
    if (cfg->nviews < 1 || cfg->nviews > MVS_MAXVIEWS || cfg->wsize < 1 || cfg->wsize > 7 || cfg->csize < 1 || cfg->level < 0 ||
        cfg->level > 4 || cfg->max_propag < 1 || cfg->max_propag > 16 || cfg->max_propag * cfg->csize * cfg->csize > MVS_CAPMAX ||
        cfg->view_stride < 1 || cfg->view_begin < 0 || cfg->minImageNum < 1 ||
        std::min(cfg->minImageNum * 2, cfg->nviews) > 16 /* tau (pmmvps.cpp:32) views of a proposal sit in 16 frame lanes */) {
        g_err = "mvs_engine_create: configuration out of range (nviews <= 64, wsize <= 7, min(2 minImageNum, nviews) <= 16, max_propag*csize^2 <= 32)";
        return MVS_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_err = "mvs_engine_create: no HIP device (the engine has no CPU path)"; return MVS_ERR_NO_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_err = "mvs_engine_create: bad device ordinal"; return MVS_ERR_ARG; }
    HIPCHK(hipSetDevice(cfg->device));
    mvs_engine* e = new mvs_engine();
    e->cfg = *cfg;
    derive_params(e);
    hipError_t he = hipStreamCreate(&e->stream);
    if (he != hipSuccess) { g_err = std::string("hipStreamCreate: ") + hipGetErrorString(he); delete e; return MVS_ERR_HIP; }
    for (auto& ev : e->ev) (void)hipEventCreate(&ev);
    for (auto& ev : e->fev) (void)hipEventCreate(&ev);
    if (e->misc.ensure(8 + 256) || e->counters.ensure(MVS_COUNTER_SLOTS) || e->error_flag.ensure(1)) { delete e; return MVS_ERR_HIP; }
    (void)hipMemset(e->misc.p, 0, (8 + 256) * sizeof(unsigned long long));
    (void)hipMemset(e->error_flag.p, 0, sizeof(int32_t));
    *out = e;
    return MVS_OK;
}

int mvs_engine_destroy(mvs_engine* e) {
    if (!e) return MVS_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipStreamSynchronize(e->stream);
    (void)mvs_engine_comm_release(e);
    e->comm_counts.release(); e->comm_kill_ids.release();
    e->geo.release(); e->geo_ref.release();
    free_views(e);
    e->dviews.release(); e->pool.release(); e->pool_alt.release(); e->kill.release();
    e->cnt.release(); e->start.release(); e->cursor.release(); e->ids.release(); e->vcnt.release(); e->vstart.release();
    e->uf_parent.release(); e->uf_size.release(); e->group_edges.release(); e->dirty.release();
    e->vcursor.release(); e->key.release(); e->start_raw.release(); e->id32_raw.release(); e->id32.release(); e->vid32.release(); e->cnt_alive.release(); e->vcnt_alive.release(); e->scan_tmp.release(); e->dpgrid.release(); e->best.release();
    e->staging.release(); e->job_stage.release(); e->job_nstage.release(); e->job_cnt.release(); e->job_base_scan.release();
    e->kill_cnt.release(); e->kill_base.release(); e->per_view.release(); e->misc.release(); e->counters.release(); e->error_flag.release();
    e->big_tables.release(); e->retry_jobs.release();
    e->tmp_rec_in.release(); e->tmp_rec_out.release(); e->tmp_f_in.release(); e->tmp_f_out.release(); e->tmp_i.release(); e->tmp_bytes.release();
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->fev) if (ev) (void)hipEventDestroy(ev);
    e->fstat_buf.release();
    (void)hipStreamDestroy(e->stream);
    delete e;
    return MVS_OK;
}

int mvs_engine_set_views(mvs_engine* e, int nviews, const mvs_view_desc* views) {
    if (!e || !views || nviews != e->cfg.nviews) { g_err = "mvs_engine_set_views: nviews must equal the configured number of views"; return MVS_ERR_ARG; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    Range rg("mvs:set_views (upload + pyramids)");
    free_views(e);
    const int maxLevel = e->cfg.level + 3;
    e->hviews.assign(nviews, DView{});
    int cell_base = 0;
    size_t max_bytes = 0;
    for (int v = 0; v < nviews; ++v) {
        if (views[v].width < 8 || views[v].height < 8 || !views[v].rgb) { g_err = "mvs_engine_set_views: bad view"; return MVS_ERR_ARG; }
        max_bytes = std::max(max_bytes, (size_t)views[v].width * views[v].height * 3);
    }
    if (int r = e->tmp_bytes.ensure((int64_t)max_bytes)) return r;
    for (int v = 0; v < nviews; ++v) {
        DView& vw = e->hviews[v];
        vw.W[0] = views[v].width; vw.H[0] = views[v].height;
        for (int l = 1; l < maxLevel; ++l) { vw.W[l] = vw.W[l - 1] / 2; vw.H[l] = vw.H[l - 1] / 2; }
        setup_camera(e, vw, views[v].P);
        const size_t n0 = (size_t)vw.W[0] * vw.H[0];
        HIPCHK(hipMemcpyAsync(e->tmp_bytes.p, views[v].rgb, n0 * 3, hipMemcpyHostToDevice, st));
        for (int l = 0; l < maxLevel; ++l) {
            uint32_t* buf = nullptr;
            HIPCHK(hipMalloc((void**)&buf, std::max<size_t>((size_t)vw.W[l] * vw.H[l], 1) * sizeof(uint32_t)));
            e->img_bufs.push_back(buf);
            vw.img[l] = buf;
            if (l == 0) mvsk_rgb_to_rgba(e->tmp_bytes.p, buf, (int64_t)n0, st);
            else mvsk_pyr_down(vw.img[l - 1], vw.W[l - 1], vw.H[l - 1], buf, vw.W[l], vw.H[l], st);
        }
        vw.mask = nullptr;
        if (views[v].mask) {  // Image::alloc mask handling (image.cpp:166-183) + buildMaskPyramid (image.cpp:717-747)
            HIPCHK(hipStreamSynchronize(st));
            uint8_t* prev = nullptr;
            for (int l = 0; l <= e->cfg.level; ++l) {
                uint8_t* m = nullptr;
                HIPCHK(hipMalloc((void**)&m, std::max<size_t>((size_t)vw.W[l] * vw.H[l], 1)));
                e->mask_bufs.push_back(m);
                if (l == 0) {
                    HIPCHK(hipMemcpyAsync(m, views[v].mask, n0, hipMemcpyHostToDevice, st));
                    mvsk_mask_binarise(m, (int64_t)n0, st);
                } else mvsk_mask_down(prev, vw.W[l - 1], vw.H[l - 1], m, vw.W[l], vw.H[l], st);
                prev = m;
            }
            vw.mask = prev;
        }
        HIPCHK(hipStreamSynchronize(st));  // tmp_bytes is reused by the next view
        vw.gh = (vw.H[e->cfg.level] + e->cfg.csize - 1) / e->cfg.csize;  // patch_manager.cpp:36-37
        vw.gw = (vw.W[e->cfg.level] + e->cfg.csize - 1) / e->cfg.csize;
        vw.cell_base = cell_base;
        cell_base += vw.gw * vw.gh;
    }
    e->total_cells = cell_base;
    if (int r = e->dviews.ensure(nviews)) return r;
    HIPCHK(hipMemcpyAsync(e->dviews.p, e->hviews.data(), sizeof(DView) * nviews, hipMemcpyHostToDevice, st));
    const int64_t nc = e->total_cells;
    if (e->cnt.ensure(nc + 2) || e->start.ensure(nc + 2) || e->cursor.ensure(nc + 2) || e->vcnt.ensure(nc + 2) || e->vstart.ensure(nc + 2) ||
        e->vcursor.ensure(nc + 2) || e->start_raw.ensure(nc + 2) || e->dpgrid.ensure(nc + 2) || e->best.ensure(nc + 2) || e->cnt_alive.ensure(nc + 2) || e->vcnt_alive.ensure(nc + 2))
        return MVS_ERR_HIP;
    // Pool capacity: mvs_config.max_patches, else 4 patches per cell (the 1080p runs settle near 1 per cell) -- but never
    // more than a sixth of the device's memory for each of the two pool buffers: the index, staging and scans grow with it.
    // The bound is taken from the TOTAL memory, not from what happens to be free: the ranks of a job must arrive at the same
    // capacity (what still differs -- an explicit max_patches per rank -- is settled collectively in mvs_engine_exchange).
    int64_t pool_cap = e->cfg.max_patches > 0 ? e->cfg.max_patches : 4 * nc;
    if (e->cfg.max_patches <= 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
            pool_cap = std::max<int64_t>(std::min<int64_t>(pool_cap, (int64_t)(total_b / 6 / sizeof(DPatch))), std::min<int64_t>(pool_cap, nc));
            // ... and, on a card that is not empty (several ranks on one GPU in a rehearsal, another process, a second engine), no more
            // than fits what is free now: the pool-sized buffers below take 2.5 records + 17 bytes per patch, and the index comes on
            // top.  Ranks that arrive at different capacities this way are settled in mvs_engine_exchange (minimum headroom).
            const double per_patch = 2.5 * (double)sizeof(DPatch) + 17.0;
            const int64_t fit = (int64_t)(0.6 * (double)free_b / per_patch);
            if (fit < pool_cap) pool_cap = std::max<int64_t>(fit, std::min<int64_t>(pool_cap, nc / 4 + 1024));
        }
    }
    e->ids.headroom = e->key.headroom = e->id32.headroom = e->id32_raw.headroom = e->vid32.headroom = true;
    e->geo.headroom = e->geo_ref.headroom = true;
    if (e->pool.ensure(pool_cap) || e->pool_alt.ensure(pool_cap) || e->kill.ensure(pool_cap) || e->kill_cnt.ensure(pool_cap + 2) || e->kill_base.ensure(pool_cap + 2)) return MVS_ERR_HIP;
    HIPCHK(hipMemsetAsync(e->kill.p, 0, (size_t)pool_cap, st));
    // jobs of one colour pass over every view (upper bound, used to size the staging bookkeeping)
    int64_t njobs_max = 0;
    for (int v = 0; v < nviews; ++v) njobs_max += (int64_t)((e->hviews[v].gw + 1) / 2) * e->hviews[v].gh;
    const int maxstage = (e->prm.view_propagation ? 3 : 2) * e->prm.cap * e->prm.max_propag;
    if (e->job_stage.ensure(njobs_max * maxstage) || e->job_nstage.ensure(njobs_max + 2) || e->job_cnt.ensure(njobs_max + 2) ||
        e->job_base_scan.ensure(njobs_max + 2) || e->per_view.ensure(MVS_MAXVIEWS))
        return MVS_ERR_HIP;
    const int64_t scan_n = std::max<int64_t>(std::max<int64_t>(nc, pool_cap), njobs_max);
    if (e->scan_tmp.ensure(scan_n / 256 + 4096)) return MVS_ERR_HIP;
    // Optim::check's second tier: its global-memory id sets and the list of cells to run again (16 MB + 4 B per job) -- here, not in the
    // first pass with check, so that no pass allocates
    if (e->big_tables.ensure((int64_t)256 * 16384) || e->retry_jobs.ensure(std::max<int64_t>(njobs_max, 16))) return MVS_ERR_HIP;
    // likewise what Filter::run needs per patch and per cell (union-find, retry list, the bit per depth-map cell)
    if (e->uf_parent.ensure(pool_cap) || e->uf_size.ensure(pool_cap) || e->dirty.ensure((nc + 31) / 32 + 1) || e->fstat_buf.ensure(4096)) return MVS_ERR_HIP;
    if (e->staging.ensure(std::max<int64_t>(pool_cap / 2, 1024))) return MVS_ERR_HIP;
    if (e->tmp_rec_out.ensure(16)) return MVS_ERR_HIP;
    HIPCHK(hipStreamSynchronize(st));
    e->pool_n = 0;
    e->have_views = true;
    e->index_valid = false;
    e->staged = false;
    return MVS_OK;
}

int mvs_engine_grid_dims(mvs_engine* e, int view, int* gw, int* gh) {
    if (!e || !e->have_views || view < 0 || view >= e->cfg.nviews) { g_err = "mvs_engine_grid_dims: bad view / views not set"; return MVS_ERR_STATE; }
    if (gw) *gw = e->hviews[view].gw;
    if (gh) *gh = e->hviews[view].gh;
    return MVS_OK;
}

int mvs_engine_get_pyramid(mvs_engine* e, int view, int level, uint8_t* rgb_out, int* W, int* H) {
    if (!e || !e->have_views || view < 0 || view >= e->cfg.nviews || level < 0 || level >= e->cfg.level + 3) { g_err = "mvs_engine_get_pyramid: bad argument"; return MVS_ERR_ARG; }
    HIPCHK(hipSetDevice(e->cfg.device));
    const DView& vw = e->hviews[view];
    if (W) *W = vw.W[level];
    if (H) *H = vw.H[level];
    if (!rgb_out) return MVS_OK;
    const int64_t n = (int64_t)vw.W[level] * vw.H[level];
    if (int r = e->tmp_bytes.ensure(n * 3)) return r;
    mvsk_rgba_to_rgb(vw.img[level], e->tmp_bytes.p, n, e->stream);
    HIPCHK(hipMemcpyAsync(rgb_out, e->tmp_bytes.p, (size_t)n * 3, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MVS_OK;
}

int mvs_engine_set_thresholds(mvs_engine* e, float ncc, float before, int depth) {
    if (!e) return MVS_ERR_ARG;
    e->prm.nccThreshold = ncc; e->prm.nccThresholdBefore = before; e->prm.depth = depth;
    return MVS_OK;
}
int mvs_engine_get_thresholds(mvs_engine* e, float* ncc, float* before, int* depth) {
    if (!e) return MVS_ERR_ARG;
    if (ncc) *ncc = e->prm.nccThreshold;
    if (before) *before = e->prm.nccThresholdBefore;
    if (depth) *depth = e->prm.depth;
    return MVS_OK;
}
int mvs_engine_update_threshold(mvs_engine* e) {  // pmmvps.cpp:70-74 and :105
    if (!e) return MVS_ERR_ARG;
    e->prm.nccThreshold -= 0.05f; e->prm.nccThresholdBefore -= 0.05f; ++e->prm.depth;
    return MVS_OK;
}

int mvs_engine_upload_patches(mvs_engine* e, int64_t n, const mvs_patch* patches) {
    if (!e || !e->have_views) { g_err = "mvs_engine_upload_patches: views not set"; return MVS_ERR_STATE; }
    if (n < 0 || (n > 0 && !patches)) return MVS_ERR_ARG;
    if (e->staged) { g_err = "mvs_engine_upload_patches: a pass is waiting for its commit"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    Range rg("mvs:upload_patches");
    // readPatches (patch_manager.cpp:450-463): m_fix = 0, m_tmp = score2, m_vimages cleared, empty m_images dropped
    std::vector<mvs_patch> recs;
    recs.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        mvs_patch p = patches[i];
        p.nimages = std::min(p.nimages, MVS_LISTCAP);
        if (p.nimages <= 0) continue;
        memset(p.images + p.nimages, 0, sizeof p.images - (size_t)p.nimages);  // entries past the (truncated) list are not data
        p.nvimages = 0;
        memset(p.vimages, 0, sizeof p.vimages);
        p.tmp = std::max(0.0f, p.ncc - e->prm.nccThreshold) * p.nimages;
        p.flags = 1;
        p.id = 0;
        recs.push_back(p);
    }
    if (e->pool_n + (int64_t)recs.size() > e->pool.cap) { g_err = "mvs_engine_upload_patches: patch pool capacity exceeded (raise mvs_config.max_patches)"; return MVS_ERR_CAPACITY; }
    if (!recs.empty()) HIPCHK(hipMemcpyAsync(e->pool.p + e->pool_n, recs.data(), recs.size() * sizeof(mvs_patch), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->pool_n += (int64_t)recs.size();
    e->ncc_dirty = true;
    e->index_valid = false;
    return MVS_OK;
}

// Sizes the buffers of both cell indexes for `list_entries` memberships each (0: MAX_NUM_OF_PATCHES per cell of every view, what
// m_pgrids holds after the trim), so that Propagate::run / Filter::run allocate nothing while the lists stay below that: the first
// iterations of a run otherwise grow them inside the call (free + allocate, gigabytes at a time).
int mvs_engine_reserve(mvs_engine* e, int64_t list_entries) {
    if (!e || !e->have_views || list_entries < 0) { g_err = "mvs_engine_reserve: views not set, or a negative size"; return MVS_ERR_ARG; }
    HIPCHK(hipSetDevice(e->cfg.device));
    int64_t n = list_entries > 0 ? list_entries : e->total_cells * (int64_t)(e->cfg.max_propag * e->cfg.csize * e->cfg.csize);
    // a buffer that is allocated here is also written once: the first use of fresh device memory is slow in a process that has just
    // come up (the first m_vpgrids build of the first process on a box took 95 ms instead of 21), and this call is the set-up
    bool fresh = false;
    auto take = [&](auto& buf, int64_t want) -> int {
        const int64_t before = buf.cap;
        if (int r = buf.ensure(want)) return r;
        if (buf.cap != before) {
            fresh = true;
            if (hipMemsetAsync(buf.p, 0, (size_t)buf.cap * sizeof(*buf.p), e->stream) != hipSuccess) { (void)hipGetLastError(); return MVS_ERR_HIP; }
        }
        return MVS_OK;
    };
    if (take(e->ids, n + 16) || take(e->key, n + 16) || take(e->id32, n + 16) || take(e->id32_raw, n + 16) || take(e->vid32, n + 16)) return MVS_ERR_HIP;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (fresh) e->index_valid = false;  // the lists lived in the buffers that were replaced
    return MVS_OK;
}

int mvs_engine_clear_patches(mvs_engine* e) {
    if (!e) return MVS_ERR_ARG;
    HIPCHK(hipSetDevice(e->cfg.device));
    if (e->kill.p) HIPCHK(hipMemsetAsync(e->kill.p, 0, (size_t)e->kill.cap, e->stream));
    e->pool_n = 0; e->staged = false; e->counted = false; e->index_valid = false;
    return MVS_OK;
}

int mvs_engine_num_patches(mvs_engine* e, int64_t* n_alive) {
    if (!e || !n_alive) return MVS_ERR_ARG;
    HIPCHK(hipSetDevice(e->cfg.device));
    *n_alive = 0;
    if (e->pool_n == 0) return MVS_OK;
    mvsk_alive_count(e->pool.p, e->pool_n, e->kill_cnt.p, e->stream);
    mvsk_exclusive_scan(e->kill_cnt.p, e->kill_base.p, e->pool_n, reinterpret_cast<int32_t*>(e->scan_tmp.p), e->stream);
    int32_t tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, e->kill_base.p + e->pool_n, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *n_alive = tot;
    return MVS_OK;
}

int mvs_engine_download_patches(mvs_engine* e, int64_t cap, mvs_patch* out, int64_t* n) {
    if (!e || !n) return MVS_ERR_ARG;
    int64_t alive = 0;
    if (int r = mvs_engine_num_patches(e, &alive)) return r;  // leaves kill_base = exclusive scan of the alive flags
    *n = alive;
    if (!out || alive == 0) return MVS_OK;
    const int64_t m = std::min(cap, alive);
    if (int r = e->tmp_rec_out.ensure(alive)) return r;
    mvsk_alive_gather(e->pool.p, e->pool_n, e->kill_base.p, e->tmp_rec_out.p, alive, e->stream);
    HIPCHK(hipMemcpyAsync(out, e->tmp_rec_out.p, (size_t)m * sizeof(mvs_patch), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MVS_OK;
}

// A pass that failed (staging or Optim::check capacity in this rank's shard, a HIP error) leaves its status in the engine:
// with a communicator attached the next mvs_engine_exchange hands it to every rank, so that all of them give the pass up
// together instead of waiting in a collective for a rank that has already returned.
static int pass_impl(mvs_engine* e, int iter, int pass, mvs_counters* out);
int mvs_engine_pass(mvs_engine* e, int iter, int pass, mvs_counters* out) {
    const int r = pass_impl(e, iter, pass, out);
    if (e) { e->pass_status = r; e->pass_error = r ? g_err : std::string(); }
    return r;
}
// forgets a staged pass (after an error): its records are dropped, the eviction flags it set are cleared, the pool is as it
// was when the pass began (minus the patches the MAX_NUM_OF_PATCHES trim removed, which every rank removes alike)
static void discard_pass(mvs_engine* e) {
    if (e->kill.p && e->pool_n > 0) (void)hipMemsetAsync(e->kill.p, 0, (size_t)e->pool_n, e->stream);
    (void)hipStreamSynchronize(e->stream);
    e->staged = false; e->counted = false; e->index_valid = false;
}
static int pass_impl(mvs_engine* e, int iter, int pass, mvs_counters* out) {
    if (!e || !e->have_views) { g_err = "mvs_engine_pass: views not set"; return MVS_ERR_STATE; }
    if (e->staged) { g_err = "mvs_engine_pass: the previous pass was not committed"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    HIPCHK(hipEventRecord(e->ev[0], st));
    unsigned long long trimmed = 0;
    {
        Range rg("mvs:index");
        if (int r = build_index(e, &trimmed)) return r;
    }
    HIPCHK(hipEventRecord(e->ev[1], st));
    Range rg_sweep(pass & 1 ? "mvs:sweep colour 1" : "mvs:sweep colour 0");
    // jobs: one per (swept view, row, half column) of the pass colour
    SweepArgs& a = e->sa;
    memset(&a, 0, sizeof a);
    a.iter = iter; a.inc = (iter % 2 == 0) ? 1 : -1; a.colour = pass & 1;
    int64_t nj = 0;
    const bool ranged = e->cfg.shard_count > 1;  // shard by job range over all views, else by whole views
    for (int v = ranged ? 0 : e->cfg.view_begin; v < e->cfg.nviews; v += ranged ? 1 : e->cfg.view_stride) {
        a.sweep_views[a.nsweep_views] = v;
        a.job_base[a.nsweep_views] = (int32_t)nj;
        ++a.nsweep_views;
        nj += (int64_t)((e->hviews[v].gw + 1) / 2) * e->hviews[v].gh;
    }
    a.njobs = nj;
    a.job_lo = 0; a.job_hi = nj;
    if (ranged) {
        // Contiguous ranges of equal WORK, not of equal job count: every rank holds the same index, so every rank computes the
        // same per-job work proxy (k_job_work), the same prefix sums and the same cuts; rank order stays commit order.  Equal
        // counts gave each of 8 ranks a band of one and a half views, and such bands differ in work by tens of per cent.
        // MVS_SPLIT_PROXY=0|1|2 (development): 0 = equal job counts, 1 = source entries, 2 = expected trials by kind (default).
        const int split_mode = getenv("MVS_SPLIT_PROXY") ? atoi(getenv("MVS_SPLIT_PROXY")) : 2;
        const int N = e->cfg.shard_count, R = e->cfg.shard_index;
        std::vector<int32_t> cuts(N + 1, -1);
        int64_t total_work = 0;
        if (split_mode > 0 && nj > 0) {
            const DParams p0 = current_params(e);
            int shift = 0;  // the scan is 32-bit: scale the proxy down if its worst case would not fit
            // (every job's work is rounded UP after the shift, so the 32-bit prefix sum may exceed the shifted bound by up to nj)
            while ((((int64_t)3 * e->prm.cap * e->prm.max_propag * 32 * nj) >> shift) + nj >= (int64_t)INT32_MAX) ++shift;
            mvsk_job_work(p0, a, split_mode, shift, e->job_cnt.p, st);
            mvsk_exclusive_scan(e->job_cnt.p, e->job_base_scan.p, nj, reinterpret_cast<int32_t*>(e->scan_tmp.p), st);
            if (int r = e->tmp_i.ensure(N + 1)) return r;
            HIPCHK(hipMemsetAsync(e->tmp_i.p, 0xff, (size_t)(N + 1) * sizeof(int32_t), st));
            mvsk_job_cuts(e->job_base_scan.p, nj, N, e->tmp_i.p, st);
            int32_t tw = 0;
            HIPCHK(hipMemcpyAsync(cuts.data(), e->tmp_i.p, (size_t)(N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&tw, e->job_base_scan.p + nj, sizeof tw, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            total_work = tw;
        }
        cuts[0] = 0; cuts[N] = (int32_t)nj;
        for (int r = 1; r < N; ++r) {
            if (total_work <= 0 || cuts[r] < 0) cuts[r] = (int32_t)(nj * r / N);  // no work anywhere: equal counts
            cuts[r] = std::max(cuts[r], cuts[r - 1]);
        }
        a.job_lo = cuts[R];
        a.job_hi = cuts[R + 1];
        HIPCHK(hipMemsetAsync(e->job_nstage.p, 0, (size_t)nj * sizeof(int32_t), st));  // jobs of other shards stage nothing
    }
    a.staging = e->staging.p; a.staging_cap = e->staging.cap;
    a.stage_counter = e->misc.p;
    a.job_stage = e->job_stage.p; a.job_nstage = e->job_nstage.p;
    a.maxstage = (e->prm.view_propagation ? 3 : 2) * e->prm.cap * e->prm.max_propag;
    a.kill = e->kill.p;
    a.counters = e->counters.p;
    a.error_flag = e->error_flag.p;
    if (want_vgrid(e)) {  // Optim::check runs in this pass: its second tier needs its global-memory tables and the list of cells
        if (int r = e->big_tables.ensure((int64_t)256 * 16384)) return r;
        if (int r = e->retry_jobs.ensure(std::max<int64_t>(nj, 16))) return r;
    }
    a.big_tables = e->big_tables.p; a.retry_jobs = e->retry_jobs.p; a.nretry = reinterpret_cast<int32_t*>(e->misc.p + 5);
    HIPCHK(hipMemsetAsync(e->misc.p + 5, 0, sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(e->misc.p, 0, sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(e->counters.p, 0, MVS_COUNTER_SLOTS * sizeof(DCounters), st));
    HIPCHK(hipMemsetAsync(e->error_flag.p, 0, sizeof(int32_t), st));
    const DParams p = current_params(e);
    mvsk_sweep(p, a, st);
    e->pass_retried = 0;
    if (want_vgrid(e)) {  // cells whose Optim::check outgrew the wave's LDS run again on the second tier (normally none)
        int32_t nretry = 0;
        HIPCHK(hipMemcpyAsync(&nretry, a.nretry, sizeof nretry, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        e->pass_retried = nretry;
        if (nretry > 0) { mvsk_sweep_retry(p, a, nretry, st); e->retried_cells += nretry; }
    }
    HIPCHK(hipEventRecord(e->ev[2], st));
    DCounters hc;
    std::vector<DCounters> hcs(MVS_COUNTER_SLOTS);
    int32_t herr = 0;
    unsigned long long fill[2] = {0, 0};  // evaluations spent on seeds whose m_ncc was < 0 (sortPatches)
    HIPCHK(hipMemcpyAsync(fill, e->misc.p + 1, sizeof fill, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hcs.data(), e->counters.p, hcs.size() * sizeof(DCounters), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&herr, e->error_flag.p, sizeof herr, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    memset(&hc, 0, sizeof hc);
    for (const DCounters& c : hcs) {
        hc.candidates += c.candidates; hc.prefiltered += c.prefiltered; hc.patches += c.patches; hc.fail0 += c.fail0; hc.fail1 += c.fail1;
        hc.inserted += c.inserted; hc.replaced += c.replaced; hc.evals += c.evals; hc.view_evals += c.view_evals; hc.trimmed += c.trimmed;
        for (int k = 0; k < 16; ++k) hc.stage[k] = (k == 12 || k == 13) ? std::max(hc.stage[k], c.stage[k]) : hc.stage[k] + c.stage[k];
    }
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]); e->timing.index_ms = ms;
    (void)hipEventElapsedTime(&ms, e->ev[1], e->ev[2]); e->timing.sweep_ms = ms;
    e->timing.commit_ms = 0.0f; e->timing.sweep_launches = 1; e->timing.exchange_ms = 0.0f; e->timing.exchange_bytes = 0;
    e->timing.check_retried_cells = e->pass_retried;
    e->staged = true; e->counted = false;
    if (out) {
        out->candidates = (int64_t)hc.candidates; out->prefiltered = (int64_t)hc.prefiltered; out->patches = (int64_t)hc.patches;
        out->fail0 = (int64_t)hc.fail0; out->fail1 = (int64_t)hc.fail1; out->inserted = (int64_t)hc.inserted; out->replaced = (int64_t)hc.replaced;
        out->evals = (int64_t)(hc.evals + fill[0]); out->view_evals = (int64_t)(hc.view_evals + fill[1]); out->trimmed = (int64_t)trimmed;
    }
#ifdef MVS_STAGE_TIMING
    {
        static const char* nm[8] = {"wave", "generate", "pre", "refine", "post", "check", "stage", "prologue"};
        fprintf(stderr, "[stage cycles]");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)hc.stage[k] / (double)(hc.stage[0] ? hc.stage[0] : 1));
        static const char* nm2[4] = {"gain", "search", "sort", "quad"};
        for (int k = 0; k < 4; ++k) fprintf(stderr, " %s %.1f%%", nm2[k], 100.0 * (double)hc.stage[8 + k] / (double)(hc.stage[0] ? hc.stage[0] : 1));
        fprintf(stderr, " setRefImage pairs %.1f%% choice %.1f%%", 100.0 * (double)hc.stage[14] / (double)(hc.stage[0] ? hc.stage[0] : 1), 100.0 * (double)hc.stage[15] / (double)(hc.stage[0] ? hc.stage[0] : 1));
        fprintf(stderr, "  (wave cycles %.3e; max visited %llu, max neighbours %llu)\n", (double)hc.stage[0], hc.stage[12], hc.stage[13]);
    }
#endif
    // Fault injection: MVS_FAULT_PASS=<shard>:<iter>:<pass> makes that one pass of that shard report a
    // capacity failure after its sweep, once -- how the tests drive a rank-local failure through the collective status word.
#ifdef MVS_FAULT_INJECTION  // test builds only (libmvskit_engine_faultinj.so, mvskit_amd/build.py): the product library never reads the variable
    if (const char* f = getenv("MVS_FAULT_PASS")) {
        int fr = -1, fi = -1, fp = -1;
        if (!e->fault_fired && sscanf(f, "%d:%d:%d", &fr, &fi, &fp) == 3 && fr == (e->cfg.shard_count > 1 ? e->cfg.shard_index : 0) && fi == iter && fp == pass) {
            e->fault_fired = true;
            g_err = "mvs_engine_pass: capacity failure injected by MVS_FAULT_PASS";
            return MVS_ERR_CAPACITY;
        }
    }
#endif
    if (herr & 3) { g_err = "mvs_engine_pass: staging capacity exceeded (raise mvs_config.max_patches)"; return MVS_ERR_CAPACITY; }
    if (herr & 4) {
        g_err = "mvs_engine_pass: Optim::check met more than 14336 patches around one patch, or more than 4064 neighbours (engine limit)";
        return MVS_ERR_CAPACITY;
    }
    return MVS_OK;
}

int mvs_engine_export_counts(mvs_engine* e, int64_t* n_new, int64_t* n_kill, int32_t* per_view_new) {
    if (!e || !e->staged) { g_err = "mvs_engine_export_counts: no pass to export"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    if (int r = ensure_counts(e)) return r;
    if (n_new) *n_new = e->n_new;
    if (n_kill) *n_kill = e->n_kill;
    if (per_view_new) for (int v = 0; v < e->cfg.nviews; ++v) per_view_new[v] = e->h_per_view[v];
    return MVS_OK;
}

int mvs_engine_export_device(mvs_engine* e, void* d_new, int64_t cap_new, void* d_kill, int64_t cap_kill) {
    if (!e || !e->staged) { g_err = "mvs_engine_export_device: no pass to export"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    if (int r = ensure_counts(e)) return r;
    if (e->n_new > cap_new || e->n_kill > cap_kill) { g_err = "mvs_engine_export_device: buffers too small"; return MVS_ERR_CAPACITY; }
    if (e->n_new > 0) mvsk_commit_copy(e->sa, e->job_base_scan.p, (DPatch*)d_new, cap_new, nullptr, 1, e->stream);
    if (e->n_kill > 0) mvsk_kill_export(e->kill.p, e->pool_n, e->kill_base.p, (int32_t*)d_kill, cap_kill, e->stream);
    HIPCHK(hipStreamSynchronize(e->stream));
    return MVS_OK;
}

int mvs_engine_commit_device(mvs_engine* e, const void* d_new, int64_t n_new, const void* d_kill, int64_t n_kill) {
    if (!e || !e->have_views) return MVS_ERR_STATE;
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    if (e->pool_n + n_new > e->pool.cap) { g_err = "mvs_engine_commit_device: patch pool capacity exceeded (raise mvs_config.max_patches)"; return MVS_ERR_CAPACITY; }
    HIPCHK(hipEventRecord(e->ev[2], st));
    mvsk_apply_kill_ids(e->pool.p, (const int32_t*)d_kill, n_kill, e->pool_n, st);
    mvsk_append_records(e->pool.p, e->pool_n, (const DPatch*)d_new, n_new, st);
    if (e->pool_n > 0) HIPCHK(hipMemsetAsync(e->kill.p, 0, (size_t)e->pool_n, st));
    e->pool_n += n_new;
    if (int r = compact_pool(e)) return r;
    HIPCHK(hipEventRecord(e->ev[3], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->timing.commit_ms = ms;
    e->staged = false; e->counted = false; e->index_valid = false;
    return MVS_OK;
}

int mvs_engine_commit_local(mvs_engine* e) {
    if (!e || !e->staged) { g_err = "mvs_engine_commit_local: no pass to commit"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    Range rg("mvs:commit");
    HIPCHK(hipEventRecord(e->ev[2], st));
    if (int r = ensure_counts(e)) return r;
    if (e->pool_n + e->n_new > e->pool.cap) { g_err = "mvs_engine_commit_local: patch pool capacity exceeded (raise mvs_config.max_patches)"; return MVS_ERR_CAPACITY; }
    if (e->n_new > 0) mvsk_commit_copy(e->sa, e->job_base_scan.p, e->pool.p + e->pool_n, e->pool.cap - e->pool_n, nullptr, 0, st);
    mvsk_apply_kill_flags(e->pool.p, e->kill.p, e->pool_n, st);
    e->pool_n += e->n_new;
    if (int r = compact_pool(e)) return r;
    HIPCHK(hipEventRecord(e->ev[3], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->timing.commit_ms = ms;
    e->staged = false; e->counted = false; e->index_valid = false;
    return MVS_OK;
}

// ---- multi-GPU: RCCL communicator + the per-pass exchange (include/mvskit_engine.h, "multi-GPU through the C ABI")
int mvs_comm_unique_id(void* id_out) {
    if (!id_out) return MVS_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == MVS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!rccl().ok) { g_err = "mvs_comm_unique_id: librccl could not be opened"; return MVS_ERR_STATE; }
    ncclUniqueId id;
    NCCLCHK(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return MVS_OK;
}
static int comm_check(mvs_engine* e, int rank, int world, const char* who) {
    if (!e || world < 1 || rank < 0 || rank >= world) { g_err = std::string(who) + ": bad rank / world"; return MVS_ERR_ARG; }
    if (e->comm) { g_err = std::string(who) + ": a communicator is already attached"; return MVS_ERR_STATE; }
    const int sc = e->cfg.shard_count > 1 ? e->cfg.shard_count : 1, si = e->cfg.shard_count > 1 ? e->cfg.shard_index : 0;
    if (sc != world || si != rank || e->cfg.view_begin != 0 || e->cfg.view_stride != 1) {
        g_err = std::string(who) + ": the engine must be created with shard_index = rank and shard_count = world (contiguous job ranges: rank order is the commit order)";
        return MVS_ERR_ARG;
    }
    if (!rccl().ok) { g_err = std::string(who) + ": librccl could not be opened"; return MVS_ERR_STATE; }
    return MVS_OK;
}
int mvs_engine_comm_init(mvs_engine* e, const void* id, int rank, int world) {
    if (!id) return MVS_ERR_ARG;
    if (int r = comm_check(e, rank, world, "mvs_engine_comm_init")) return r;
    HIPCHK(hipSetDevice(e->cfg.device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    NCCLCHK(rccl().CommInitRank(&c, world, uid, rank));
    e->comm = c; e->comm_owned = true; e->comm_rank = rank; e->comm_world = world;
    return MVS_OK;
}
int mvs_engine_comm_attach(mvs_engine* e, void* nccl_comm, int rank, int world) {
    if (!nccl_comm) return MVS_ERR_ARG;
    if (int r = comm_check(e, rank, world, "mvs_engine_comm_attach")) return r;
    e->comm = (ncclComm_t)nccl_comm; e->comm_owned = false; e->comm_rank = rank; e->comm_world = world;
    return MVS_OK;
}
int mvs_engine_comm_release(mvs_engine* e) {
    if (!e) return MVS_ERR_ARG;
    if (e->comm && e->comm_owned && rccl().ok) {
        (void)hipSetDevice(e->cfg.device);
        (void)hipStreamSynchronize(e->stream);
        (void)rccl().CommDestroy(e->comm);
    }
    e->comm = nullptr; e->comm_owned = false; e->comm_rank = 0; e->comm_world = 1;
    return MVS_OK;
}

int mvs_engine_comm_info(mvs_engine* e, int* rank, int* world, int* comm_count, int* comm_rank) {
    if (!e) return MVS_ERR_ARG;
    if (rank) *rank = e->comm_rank;
    if (world) *world = e->comm ? e->comm_world : 0;  // 0: no communicator attached
    int cc = -1, cr = -1;  // what the communicator itself says (ncclCommCount / ncclCommUserRank), -1 where it cannot be asked
    if (e->comm && rccl().ok) {
        if (rccl().CommCount && rccl().CommCount(e->comm, &cc) != ncclSuccess) cc = -1;
        if (rccl().CommUserRank && rccl().CommUserRank(e->comm, &cr) != ncclSuccess) cr = -1;
    }
    if (comm_count) *comm_count = cc;
    if (comm_rank) *comm_rank = cr;
    return MVS_OK;
}

#define MVS_XCHG_WORDS 5  // what a rank tells the others before anything is moved: {new records, evicted ids, status, pool headroom, kill-id capacity}
int mvs_engine_exchange(mvs_engine* e) {
    if (!e) { g_err = "mvs_engine_exchange: null engine"; return MVS_ERR_ARG; }
    if (!e->comm) { g_err = "mvs_engine_exchange: no communicator (mvs_engine_comm_init / _attach)"; return MVS_ERR_STATE; }
    if (!e->staged && e->pass_status == MVS_OK) { g_err = "mvs_engine_exchange: no pass to exchange"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    const Rccl& R = rccl();
    const int world = e->comm_world, rank = e->comm_rank;
    Range rg("mvs:exchange");
    // (0) everything that can fail on this rank alone happens BEFORE the first collective and ends up in a status word
    int local = e->pass_status;
    std::string local_err = e->pass_error;
    auto fail_local = [&](int r) { if (local == MVS_OK) { local = r; local_err = g_err; } };
    if (hipEventRecord(e->ev[4], st) != hipSuccess) { g_err = "mvs_engine_exchange: hipEventRecord failed"; fail_local(MVS_ERR_HIP); }
    if (local == MVS_OK) { if (int r = ensure_counts(e)) fail_local(r); }
    if (int r = e->comm_counts.ensure(MVS_XCHG_WORDS * (1 + (int64_t)world))) fail_local(r);
    // the evicted ids of all ranks (a rank evicts pool patches from its own cells only, but a patch sits in the cells of several
    // views, so the union may name an id twice): one allocation of pool-capacity ids, the limit agreed as the minimum over the ranks
    if (int r = e->comm_kill_ids.ensure(std::max<int64_t>(e->pool.cap, 1024))) fail_local(r);
    // The ONE way out before the first collective: no device word to put the status in (a 40-byte allocation failed).  This rank
    // can then tell nobody anything; the other ranks wait in their all-gather until the launcher tears the job down (bench.py's
    // spawn_ranks and torch.distributed.run both end the job when a rank exits non-zero) -- INTEGRATION.md, "failures".
    if (!e->comm_counts.p) { g_err = "mvs_engine_exchange: no device memory for the status word; the job must be torn down by its launcher"; return MVS_ERR_HIP; }
    if (local != MVS_OK) { e->n_new = 0; e->n_kill = 0; }
    // (1) one all-gather of MVS_XCHG_WORDS int64 per rank.  A failure of the host-to-device copy of this rank's words does not skip
    // the collective: the status word is then written by a kernel-free fallback (hipMemsetD32Async of the status alone), so the
    // others still see a failure and nobody waits; a failure of the all-gather or of its read-back is a failure of the transport
    // itself -- past that nothing can be agreed on.
    int64_t mine[MVS_XCHG_WORDS] = {e->n_new, e->n_kill, (int64_t)local, e->pool.cap - e->pool_n, e->comm_kill_ids.cap};
    std::vector<int64_t> all(MVS_XCHG_WORDS * (size_t)world);
    if (hipMemcpyAsync(e->comm_counts.p, mine, sizeof mine, hipMemcpyHostToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        g_err = "mvs_engine_exchange: host-to-device copy of the count words failed"; fail_local(MVS_ERR_HIP);
        // {0, 0, MVS_ERR_HIP (all ones in the low word, sign-extended by the second), 0, 0}: zero the words, then the status
        (void)hipMemsetAsync(e->comm_counts.p, 0, sizeof mine, st);
        (void)hipMemsetD32Async((hipDeviceptr_t)(e->comm_counts.p + 2), (int)(uint32_t)MVS_ERR_HIP, 2, st);
    }
    NCCLCHK(R.AllGather(e->comm_counts.p, e->comm_counts.p + MVS_XCHG_WORDS, MVS_XCHG_WORDS, ncclInt64, e->comm, st));
    HIPCHK(hipMemcpyAsync(all.data(), e->comm_counts.p + MVS_XCHG_WORDS, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int64_t> off_new(world + 1, 0), off_kill(world + 1, 0);
    int64_t headroom = INT64_MAX, kill_cap = INT64_MAX;
    int bad_rank = -1, bad_status = MVS_OK;
    for (int r = 0; r < world; ++r) {
        const int64_t* w = &all[MVS_XCHG_WORDS * (size_t)r];
        off_new[r + 1] = off_new[r] + w[0]; off_kill[r + 1] = off_kill[r] + w[1];
        if (w[2] != MVS_OK && bad_rank < 0) { bad_rank = r; bad_status = (int)w[2]; }
        headroom = std::min(headroom, w[3]); kill_cap = std::min(kill_cap, w[4]);
    }
    const int64_t tot_new = off_new[world], tot_kill = off_kill[world];
    // every rank sees the same words and takes the same way out: the pass is given up everywhere, nobody waits
    if (bad_rank >= 0) {
        discard_pass(e);
        e->pass_status = MVS_OK;
        if (bad_rank == rank) g_err = local_err.empty() ? std::string("mvs_engine_exchange: this rank's pass failed") : local_err;
        else g_err = "mvs_engine_exchange: rank " + std::to_string(bad_rank) + " reported status " + std::to_string(bad_status) + " for this pass; all ranks give it up";
        return bad_status;
    }
    if (all[MVS_XCHG_WORDS * (size_t)rank] != e->n_new || all[MVS_XCHG_WORDS * (size_t)rank + 1] != e->n_kill) {
        g_err = "mvs_engine_exchange: count all-gather returned other values for this rank"; return MVS_ERR_STATE;
    }
    if (tot_new > headroom || tot_kill > kill_cap) {  // decided on the minimum over the ranks: all fail alike, whatever their own capacity
        discard_pass(e);
        g_err = "mvs_engine_exchange: patch pool capacity exceeded on at least one rank (raise mvs_config.max_patches)";
        return MVS_ERR_CAPACITY;
    }
    // (2) this rank's block goes straight to its final place behind the pool, (3) every block is broadcast in place.
    // Job ranges are contiguous and ascending in rank, so the concatenation in rank order is the global
    // (view, cell, creation) order of the 1-GPU commit.
    DPatch* tail = e->pool.p + e->pool_n;
    if (e->n_new > 0) mvsk_commit_copy(e->sa, e->job_base_scan.p, tail + off_new[rank], e->n_new, nullptr, 0, st);
    if (e->n_kill > 0) mvsk_kill_export(e->kill.p, e->pool_n, e->kill_base.p, e->comm_kill_ids.p + off_kill[rank], e->n_kill, st);
    {
        // an error inside the group still closes it (a group left open would swallow every later call of this thread)
        ncclResult_t first = R.GroupStart();
        for (int r = 0; r < world && first == ncclSuccess; ++r) {
            const int64_t nn = all[MVS_XCHG_WORDS * (size_t)r], nk = all[MVS_XCHG_WORDS * (size_t)r + 1];
            if (nn > 0) first = R.Broadcast(tail + off_new[r], tail + off_new[r], (size_t)nn * sizeof(DPatch), ncclUint8, r, e->comm, st);
            if (nk > 0 && first == ncclSuccess) first = R.Broadcast(e->comm_kill_ids.p + off_kill[r], e->comm_kill_ids.p + off_kill[r], (size_t)nk, ncclInt32, r, e->comm, st);
        }
        const ncclResult_t end = R.GroupEnd();
        if (first == ncclSuccess) first = end;
        if (first != ncclSuccess) { g_err = std::string("mvs_engine_exchange: record / kill-id broadcast: ") + R.GetErrorString(first); return MVS_ERR_HIP; }
    }
    HIPCHK(hipEventRecord(e->ev[5], st));
    // commit of the union (the same on every rank)
    {
        Range rc("mvs:commit");
        HIPCHK(hipEventRecord(e->ev[2], st));
        mvsk_apply_kill_ids(e->pool.p, e->comm_kill_ids.p, tot_kill, e->pool_n, st);
        if (e->pool_n > 0) HIPCHK(hipMemsetAsync(e->kill.p, 0, (size_t)e->pool_n, st));
        e->pool_n += tot_new;
        if (int r = compact_pool(e)) return r;
        HIPCHK(hipEventRecord(e->ev[3], st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
    }
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e->ev[4], e->ev[5]); e->timing.exchange_ms = ms;
    (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->timing.commit_ms = ms;
    e->timing.exchange_bytes = (tot_new - e->n_new) * (int64_t)sizeof(DPatch) + (tot_kill - e->n_kill) * 4 + 8 * MVS_XCHG_WORDS * (int64_t)(world - 1);
    e->staged = false; e->counted = false; e->index_valid = false;
    return MVS_OK;
}

int mvs_engine_propagate(mvs_engine* e, int iter, mvs_counters* out) {  // Propagate::run, propagate.cpp:28-64
    mvs_counters total;
    memset(&total, 0, sizeof total);
    mvs_timing tt{};
    for (int pass = 0; pass < 2; ++pass) {
        mvs_counters c;
        memset(&c, 0, sizeof c);
        const int rp = mvs_engine_pass(e, iter, pass, &c);
        if (e && e->comm) {  // the exchange is reached whatever the pass returned: it is where the ranks agree on a failure
            if (int r = mvs_engine_exchange(e)) return r;
        } else {
            if (rp) { if (e && e->staged) { const std::string keep = g_err; discard_pass(e); g_err = keep; } return rp; }
            if (int r = mvs_engine_commit_local(e)) return r;
        }
        add_counters(total, c);
        tt.index_ms += e->timing.index_ms; tt.sweep_ms += e->timing.sweep_ms; tt.commit_ms += e->timing.commit_ms; tt.sweep_launches += 1;
        tt.exchange_ms += e->timing.exchange_ms; tt.exchange_bytes += e->timing.exchange_bytes; tt.check_retried_cells += e->timing.check_retried_cells;
    }
    e->timing = tt;
    if (out) *out = total;
    return MVS_OK;
}

int mvs_engine_filter(mvs_engine* e, int64_t* removed4) {  // Filter::run, filter.cpp:25-49
    if (!e || !e->have_views) { g_err = "mvs_engine_filter: views not set"; return MVS_ERR_STATE; }
    if (e->staged) { g_err = "mvs_engine_filter: a pass is waiting for its commit"; return MVS_ERR_STATE; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    int64_t rem[4] = {0, 0, 0, 0};
    Range rg("mvs:Filter::run");
    FilterRun fr{e};
    // the one thing without which a rank cannot even report a failure: the word(s) of the agreement
    if (fr.multi()) { if (int r = e->comm_counts.ensure(MVS_XCHG_WORDS * (1 + (int64_t)e->comm_world))) return r; }
    FRHIP(hipEventRecord(e->ev[0], st));
    FRHIP(hipMemsetAsync(e->error_flag.p, 0, sizeof(int32_t), st));
    if (e->pool_n > 0) FRHIP(hipMemsetAsync(e->kill.p, 0, (size_t)e->pool_n, st));
    e->fstats = mvs_filter_stats{};
    FR(e->fstat_buf.ensure(4096 + 16 + 2048));  // [4096..]: k_filter_neighbor's stage cycles (-DMVS_STAGE_TIMING); [4112..]: k_filter_exact's [1024][2] work counts
    FRHIP(hipMemsetAsync(e->fstat_buf.p, 0, (4096 + 16 + 2048) * sizeof(unsigned long long), st));
    FR(mvs_engine_num_patches(e, &e->fstats.patches_in));
    e->fstats_exchange_bytes = 0;
    int64_t first = 0, last = 0;  // this rank's share of the pool (everything on one GPU)
    int32_t herr = 0;
    // every `break` below: the ranks have agreed to give the call up (or, on one GPU, a step failed)
    do {
        if (filter_rebuild(fr, 0, true, false)) break;
        FR(pack_geometry(e));
        FRHIP(hipEventRecord(e->fev[0], st));
        filter_range(e, first, last);
        if (fr.live()) mvsk_filter_outside(current_params(e), e->kill.p, first, last, st);          // filterOutside
        FRHIP(hipEventRecord(e->fev[1], st));
        filter_fault_point(fr, 0);
        if (filter_exchange(fr, true, false)) break;
        FR(apply_kills(e, &rem[0], true));
        // a stage that removed nothing leaves the depth maps, hence m_vimages (additive pass) and both grids, as they are
        if (rem[0] > 0) { if (filter_rebuild(fr, 1, false, false, 2)) break; }
        e->fstats.exact_patches = e->fstats.patches_in - rem[0];
        FRHIP(hipMemsetAsync(e->misc.p + 1, 0, 2 * sizeof(unsigned long long), st));
        FRHIP(hipEventRecord(e->fev[2], st));
        filter_range(e, first, last);
#ifdef MVS_STAGE_TIMING
        FRHIP(hipMemsetAsync(e->counters.p, 0, sizeof(DCounters), st));
        if (fr.live()) {
            mvsk_filter_exact(current_params(e), e->kill.p, e->fstat_buf.p + 4112, e->counters.p->stage, first, last, st);
            DCounters hc;
            (void)hipMemcpyAsync(&hc, e->counters.p, sizeof hc, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            static const char* nm[8] = {"wave", "frames", "sampling", "pair sums", "choice", "load", "visibility", "lists"};
            fprintf(stderr, "[filterExact cycles]");
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)hc.stage[k] / (double)(hc.stage[0] ? hc.stage[0] : 1));
            fprintf(stderr, " (wave cycles %.3e)\n", (double)hc.stage[0]);
        }
#else
        if (fr.live()) mvsk_filter_exact(current_params(e), e->kill.p, e->fstat_buf.p + 4112, nullptr, first, last, st);  // filterExact
#endif
        FRHIP(hipEventRecord(e->fev[3], st));
        {
            std::vector<unsigned long long> ev(2048, 0ull);
            FRHIP(hipMemcpyAsync(ev.data(), e->fstat_buf.p + 4112, ev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            FRHIP(hipStreamSynchronize(st));
            unsigned long long ev2[2] = {0, 0};
            for (size_t k = 0; k < ev.size(); ++k) ev2[k & 1] += ev[k];
            e->fstats.exact_view_evals = (int64_t)ev2[1];
        }
        filter_fault_point(fr, 1);
        if (filter_exchange(fr, true, true)) break;  // kill bytes + the rewritten m_images
        FR(apply_kills(e, &rem[1], true));
        if (filter_rebuild(fr, 1, true, true, 1)) break;  // filterExact rewrote m_images: every patch's m_vimages is tested anew
        e->fstats.neighbor_patches = e->fstats.exact_patches - rem[1];
        {                                                                              // filterNeighbor(1)
            FR(e->uf_parent.ensure(e->pool.cap));                                      // reused as the retry list
            FRHIP(hipMemsetAsync(e->misc.p + 4, 0, sizeof(unsigned long long), st));
            int32_t* nretry = reinterpret_cast<int32_t*>(e->misc.p + 4);
            FRHIP(hipEventRecord(e->fev[4], st));
            if (fr.live() && (!e->lists_dense[0] || !e->lists_dense[1])) { g_err = "Filter::filterNeighbor: the grid indexes must come from a rebuild without the trim"; fr.note(MVS_ERR_ARG); }
            filter_range(e, first, last);
            int32_t nr = 0;
            if (fr.live()) mvsk_filter_neighbor(current_params(e), e->kill.p, e->uf_parent.p, nretry, e->error_flag.p, e->fstat_buf.p, first, last, st);
            FRHIP(hipMemcpyAsync(&nr, nretry, sizeof nr, hipMemcpyDeviceToHost, st));
            FRHIP(hipStreamSynchronize(st));
            FRHIP(hipGetLastError());
            e->fstats.neighbor_retried = nr;
            if (fr.live()) mvsk_filter_neighbor_retry(current_params(e), e->kill.p, e->uf_parent.p, nr, e->error_flag.p, e->fstat_buf.p, st);
            FRHIP(hipGetLastError());  // a refused launch (LDS request) must not pass for "nothing to filter"
            FRHIP(hipEventRecord(e->fev[5], st));
#ifdef MVS_STAGE_TIMING
            if (fr.live()) {
                unsigned long long hc[16] = {0};
                (void)hipMemcpyAsync(hc, e->fstat_buf.p + 4096, sizeof hc, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
                const double w = (double)(hc[0] ? hc[0] : 1);
                fprintf(stderr, "[filterNeighbor cycles] load + grids %.1f%% set build %.1f%% gather + predicate %.1f%% filterQuad %.1f%% (wave cycles %.3e)\n",
                        100.0 * hc[1] / w, 100.0 * hc[9] / w, 100.0 * hc[10] / w, 100.0 * hc[11] / w, (double)hc[0]);
            }
#endif
        }
        filter_fault_point(fr, 3);
        if (filter_exchange(fr, true, false)) break;
        FR(apply_kills(e, &rem[2], true));
        // filterNeighbor removes a handful of patches (46 of 6 M in a second call at 1080p): m_pgrids is not rebuilt for them -- its one
        // reader left, filterSmallGroups, skips a listed patch that is dead (k_groups_edges) -- while m_vpgrids is, because the removals
        // can ADD memberships (a patch becomes visible where its occluder went)
        if (rem[2] > 0) { if (filter_rebuild(fr, 1, false, true, 2)) break; }
        {                                                                              // filterSmallGroups (replicated: every rank on the whole pool)
            int64_t alive = 0;
            FR(mvs_engine_num_patches(e, &alive));
            FR(e->uf_parent.ensure(e->pool.cap));
            FR(e->uf_size.ensure(e->pool.cap));
            const int threshold = (int)std::max<int64_t>(20, alive / 10000);
            FRHIP(hipEventRecord(e->fev[6], st));
            if (fr.live()) {
                if (!e->cfg.literal_groups) mvsk_groups(current_params(e), e->uf_parent.p, e->uf_size.p, threshold, e->kill.p, st);
                else fr.note(literal_small_groups(e, threshold));
            }
            FRHIP(hipEventRecord(e->fev[7], st));
            FR(apply_kills(e, &rem[3], true));
        }
        if (rem[3] > 0) { if (filter_rebuild(fr, 1, false, false, 2)) break; }
        FR(compact_pool(e));
        FRHIP(hipMemcpyAsync(&herr, e->error_flag.p, sizeof herr, hipMemcpyDeviceToHost, st));
        FRHIP(hipEventRecord(e->ev[1], st));
        FRHIP(hipStreamSynchronize(st));
        FRHIP(hipGetLastError());
        if (fr.live() && (herr & 4)) {  // this rank's share met the engine's limit: every rank returns that status
            g_err = "mvs_engine_filter: more than 14336 patches around one patch, or more than 4064 neighbours (engine limit)";
            fr.note(MVS_ERR_CAPACITY);
        }
    } while (0);
    const int status = filter_agree(fr);  // the closing agreement (one GPU: this rank's own status)
    e->index_valid = false;
    e->geo_valid = false;  // compact_pool renumbers the patches
    if (status != MVS_OK) {
        // given up: what the unfinished stage marked is forgotten.  The stages that completed stand (on every rank alike); a stage that
        // was under way may have rewritten the lists of this rank's share only -- after an error the caller re-uploads or stops.
        const std::string keep = g_err;
        if (e->kill.p && e->pool_n > 0) (void)hipMemsetAsync(e->kill.p, 0, (size_t)e->pool_n, st);
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        e->dirty_marked = false;
        g_err = keep;
        return status;
    }
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
    e->timing = mvs_timing{};
    e->timing.index_ms = ms;  // whole Filter::run
    {
        mvs_filter_stats& f = e->fstats;
        f.total_ms = ms;
        (void)hipEventElapsedTime(&f.outside_ms, e->fev[0], e->fev[1]);
        (void)hipEventElapsedTime(&f.exact_ms, e->fev[2], e->fev[3]);
        (void)hipEventElapsedTime(&f.neighbor_ms, e->fev[4], e->fev[5]);
        (void)hipEventElapsedTime(&f.groups_ms, e->fev[6], e->fev[7]);
        f.rebuild_ms = f.total_ms - f.outside_ms - f.exact_ms - f.neighbor_ms - f.groups_ms;  // rebuilds + the scans between the stages
        std::vector<unsigned long long> part(4096);
        HIPCHK(hipMemcpy(part.data(), e->fstat_buf.p, part.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long sum4[4] = {0, 0, 0, 0};
        for (int b = 0; b < 1024; ++b) for (int k = 0; k < 4; ++k) sum4[k] += part[4 * b + k];
        f.neighbor_tasks = (int64_t)sum4[0]; f.neighbor_entries = (int64_t)sum4[1]; f.neighbor_visited = (int64_t)sum4[2]; f.neighbor_accepted = (int64_t)sum4[3];
    }
    e->fstats.exchange_bytes = e->fstats_exchange_bytes;
    if (removed4) for (int k = 0; k < 4; ++k) removed4[k] = rem[k];
    return MVS_OK;
}

int mvs_engine_filter_stats(mvs_engine* e, mvs_filter_stats* out) {
    if (!e || !out) return MVS_ERR_ARG;
    *out = e->fstats;
    return MVS_OK;
}

int mvs_engine_last_timing(mvs_engine* e, mvs_timing* t) {
    if (!e || !t) return MVS_ERR_ARG;
    *t = e->timing;
    return MVS_OK;
}

int mvs_engine_depth_normal_map(mvs_engine* e, int view, int kind, float* depth, float* normal, int32_t* ids) {
    if (!e || !e->have_views || view < 0 || view >= e->cfg.nviews || kind < 0 || kind > 1) { g_err = "mvs_engine_depth_normal_map: bad argument"; return MVS_ERR_ARG; }
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    const DView& vw = e->hviews[view];
    const int ncells = vw.gw * vw.gh;
    const DParams p = current_params(e);
    const unsigned long long* sel = nullptr;
    if (kind == 0) {
        HIPCHK(hipMemsetAsync(e->dpgrid.p, 0xff, (size_t)e->total_cells * sizeof(unsigned long long), st));
        mvsk_depth_maps(p, e->dpgrid.p, nullptr, st);
        sel = e->dpgrid.p + vw.cell_base;
        e->index_valid = false;
    } else {
        HIPCHK(hipMemsetAsync(e->best.p, 0, (size_t)ncells * sizeof(unsigned long long), st));
        mvsk_best_ncc_map(p, view, e->best.p, st);
        sel = e->best.p;
    }
    if (e->tmp_f_out.ensure((int64_t)ncells * 4) || e->tmp_i.ensure(ncells)) return MVS_ERR_HIP;
    mvsk_map_extract(p, view, kind, sel, e->tmp_f_out.p, e->tmp_f_out.p + ncells, e->tmp_i.p, ncells, st);
    if (depth) HIPCHK(hipMemcpyAsync(depth, e->tmp_f_out.p, (size_t)ncells * sizeof(float), hipMemcpyDeviceToHost, st));
    if (normal) HIPCHK(hipMemcpyAsync(normal, e->tmp_f_out.p + ncells, (size_t)ncells * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (ids) HIPCHK(hipMemcpyAsync(ids, e->tmp_i.p, (size_t)ncells * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MVS_OK;
}

int mvs_engine_probe(mvs_engine* e, int op, int64_t n, const mvs_patch* in_rec, const float* in_f, mvs_patch* out_rec, float* out_f, int32_t* out_i) {
    if (!e || !e->have_views || n < 0 || op < 0 || op > 5) { g_err = "mvs_engine_probe: bad argument / views not set"; return MVS_ERR_ARG; }
    if (n == 0) return MVS_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    if (op == MVS_PROBE_POSTPROCESS && e->prm.depth > 0) if (int r = build_index(e, nullptr)) return r;  // setVImagesVGrids reads m_dpgrids
    const int64_t nf_out = op == MVS_PROBE_MATH ? 5 * n : n;
    if (e->tmp_rec_in.ensure(n) || e->tmp_rec_out.ensure(n) || e->tmp_f_in.ensure(n) || e->tmp_f_out.ensure(nf_out) || e->tmp_i.ensure(n + 1)) return MVS_ERR_HIP;
    if (op == MVS_PROBE_MATH) {
        if (!in_f || !out_f) return MVS_ERR_ARG;
        HIPCHK(hipMemcpyAsync(e->tmp_f_in.p, in_f, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    } else {
        if (!in_rec) return MVS_ERR_ARG;
        HIPCHK(hipMemcpyAsync(e->tmp_rec_in.p, in_rec, (size_t)n * sizeof(mvs_patch), hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemsetAsync(e->tmp_i.p, 0, (size_t)(n + 1) * sizeof(int32_t), st));
    const DParams p = current_params(e);
    mvsk_probe(p, op, n, e->tmp_rec_in.p, e->tmp_f_in.p, e->tmp_rec_out.p, e->tmp_f_out.p, e->tmp_i.p, st);
    if (out_rec && (op == MVS_PROBE_PREPROCESS || op == MVS_PROBE_REFINE || op == MVS_PROBE_POSTPROCESS))
        HIPCHK(hipMemcpyAsync(out_rec, e->tmp_rec_out.p, (size_t)n * sizeof(mvs_patch), hipMemcpyDeviceToHost, st));
    if (out_f && (op == MVS_PROBE_NCC || op == MVS_PROBE_COST || op == MVS_PROBE_MATH))
        HIPCHK(hipMemcpyAsync(out_f, e->tmp_f_out.p, (size_t)nf_out * sizeof(float), hipMemcpyDeviceToHost, st));
    if (out_i) HIPCHK(hipMemcpyAsync(out_i, e->tmp_i.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return MVS_OK;
}

}  // extern "C"
