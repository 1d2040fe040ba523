// mvs_check.cuh -- Optim::check (pmmvps/optim.cpp:300-323) for one candidate, one wavefront.
//   Filter::computeGain      filter.cpp:108-146      view lanes, each walking the list of its cell
//   PatchManager::findNeighbors patch_manager.cpp:671-728  lanes over (image, cell) pairs, ids into an LDS hash set,
//                                                     then sorted ascending (the oracle's std::sort + unique order)
//   Filter::filterQuad       filter.cpp:329-392      rows lane-parallel, sums in the reference's sequential order
//   Filter::lls              filter.cpp:411-430      normal equations + pivoted elimination in double (as the oracle)
// Lists of other cells/views are read from the pass snapshot (CSR, dead entries skipped); the list of the
// destination cell being processed is the live one, published by the caller in LDS.
#pragma once
#include "mvs_device.cuh"

namespace mvsdev {

// LDS of the neighbour search: a hash set / sorted id list of HCAP ints and 3 floats per neighbour kept for filterQuad.
// Optim::check (inside the sweep) uses the small configuration; Filter::filterNeighbor, which sees the untrimmed lists
// of every patch, the large one.
// One LDS region of MVS_SET_LDS_FLOATS dwords serves the whole neighbour search: first the id set (HCAP slots), then --
// the accepted ids compacted to its front -- the 3 floats per neighbour filterQuad keeps, stored behind the ids.
// 2048 slots (at most 7/8 visited) and 576 neighbours fit 9472 B, which with the 768 B of static LDS is 10 KB per wave:
// 16 waves per CU.  The oracle picks the table size by the same rule (engine_neighbor_order).
// all three builds: a 2048-slot first tier (the many-view builds, which meet more patches, lean on the second tier more often; the
// oracle's rule: "the small set if it fits, else 16384").  Rounds 3-4 gave the 64-view build 4096 slots, which its 22 KB of LDS per
// wave had room for; with 16-view chunks of textures it runs in the 12 KB of the other builds.
#define MVS_HASH_CAP 2048
#define MVS_ROW_CAP 576
#define MVS_CHECK_LDS_FLOATS 2368
// the id set (HCAP slots), or -- the accepted ids compacted to its front -- RCAP ids rounded up to 64, 3 floats per neighbour behind them
// and the 20 doubles of filterQuad's normal equations (+ alignment): whichever is larger
#define MVS_SET_ROWS_FLOATS(RCAP) ((((RCAP) + 63) & ~63) + 3 * (RCAP) + 64)
#define MVS_SET_LDS_FLOATS(HCAP, RCAP) ((HCAP) > MVS_SET_ROWS_FLOATS(RCAP) ? (HCAP) : MVS_SET_ROWS_FLOATS(RCAP))
static_assert(MVS_SET_LDS_FLOATS(MVS_HASH_CAP, MVS_ROW_CAP) <= MVS_CHECK_LDS_FLOATS, "neighbour search LDS");
#define MVS_FILTER_HASH_CAP 2048           // Filter::filterNeighbor, first launch over all patches: the small configuration in both builds
#define MVS_FILTER_ROW_CAP 448            // 448 + 3 * 448 + 64 floats of rows fit the 8 KB of the set: 20 waves per CU (at 576: 16)
#define MVS_FILTER2_HASH_CAP 16384         // second launch over the patches the first could not hold: exactly 64 KB of dynamic LDS (the
#define MVS_FILTER2_ROW_CAP 4064           // kernel has no static LDS), which also holds 4096 + 3 * 4064 + 64 dwords of ids, rows and sums
static_assert(MVS_SET_LDS_FLOATS(MVS_FILTER2_HASH_CAP, MVS_FILTER2_ROW_CAP) * 4 <= 65536, "the retry launch of Filter::filterNeighbor asks for at most 64 KB");
DEV int rows_offset(int n) { return (n + 63) & ~63; }  // the rows start behind the ids, on a 256-byte boundary

struct CheckCtx {
    const DPatch* staging;  // records created by this pass (ids >= MVS_NEWBASE)
    int live_view, live_cell, live_n;
    const int* live_ids;    // LDS: the destination cell's current list
    unsigned long long* st; // diagnostic build (-DMVS_STAGE_TIMING): DCounters::stage, else unused
};
#define MVS_BIG_SLOTS 256   // global-memory id sets of Optim::check's second tier: one per block of k_sweep_retry
#ifdef MVS_STAGE_TIMING
#define CK_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define CK_ADD(k) { const unsigned long long t1_ = CK_NOW(); if (cx.st) cx.st[k] += t1_ - ck_t; ck_t = t1_; }
#define CK_BEGIN() unsigned long long ck_t = CK_NOW();
#else
#define CK_ADD(k)
#define CK_BEGIN()
#endif

DEV const DPatch* patch_ptr(const DParams& prm, const CheckCtx& cx, int id) {
    return id >= MVS_NEWBASE ? cx.staging + (id - MVS_NEWBASE) : prm.pool + id;
}

// geometry of a patch needed by the neighbour predicates
struct PGeo { F4 coord, normal; float dscale, ncc; int ref; };
DEV PGeo load_geo(const DPatch* p) { return {ld4(p->coord), ld4(p->normal), p->dscale, p->ncc, (int)p->images[0]}; }
// The same of pool patch `id`.  PK (Filter::run's stages, DParams::geo set): from the packed copy -- one 32-byte sector, and a byte for
// the reference view where the caller uses it -- instead of the 96- to 192-byte record; what is not stored is what k_geo_pack verified
// (coord.w == 1, normal.w == 0), so the values are the record's bit for bit.
template <bool PK> DEV PGeo pool_geo(const DParams& prm, int id) {
    if (PK) {
        const float4 a = prm.geo[2 * (size_t)id], b = prm.geo[2 * (size_t)id + 1];
        return {{a.x, a.y, a.z, 1.0f}, {b.x, b.y, b.z, 0.0f}, a.w, b.w, (int)prm.geo_ref[id]};
    }
    return load_geo(prm.pool + id);
}

// PmMvps::isNeighbor, pmmvps.cpp:117-147 (deg/rad typo at :124 kept)
DEV int is_neighbor_h(const DParams& prm, const PGeo& l, const PGeo& r, float hunit, float thr) {
    if (dot4(l.normal, r.normal) < prm.cosNeighborTypo) return 0;
    const F4 diff = sub4(l.coord, r.coord);
    const float vunit = l.dscale + r.dscale;
    const float f0 = dot4(l.normal, diff), f1 = dot4(r.normal, diff);
    float ftmp = (fabsf(f0) + fabsf(f1)) / 2.0f;
    ftmp = div_rn(ftmp, vunit);  // two seeds without a depth scale (vunit 0): infinity or NaN either way, and neither is < thr
    const F4 h = add4(sub4(diff, mul4(l.normal, f0)), sub4(diff, mul4(r.normal, f1)));
    const float hsize = div_rn(norm4(h) / 2.0f, hunit);
    if (1.0f < hsize) ftmp = div_rn(ftmp, fminf(2.0f, hsize));
    return ftmp < thr ? 1 : 0;
}
DEV int is_neighbor(const DParams& prm, const PGeo& l, const PGeo& r, float thr) {
    const float hunit = (get_unit(prm, prm.views + l.ref, l.coord) + get_unit(prm, prm.views + r.ref, r.coord)) / 2.0f * (float)prm.csize;
    return is_neighbor_h(prm, l, r, hunit, thr);
}
// PmMvps::isNeighborRadius, pmmvps.cpp:149-180
DEV int is_neighbor_radius(const DParams& prm, const PGeo& l, const PGeo& r, float hunit, float thr, float radius) {
    if (dot4(l.normal, r.normal) < prm.cosNeighbor120) return 0;
    const F4 diff = sub4(r.coord, l.coord);
    const float vunit = l.dscale + r.dscale;
    const float f0 = dot4(l.normal, diff), f1 = dot4(r.normal, diff);
    float ftmp = (fabsf(f0) + fabsf(f1)) / 2.0f;
    ftmp = div_rn(ftmp, vunit);
    const F4 h = sub4(sub4(mul4(diff, 2.0f), mul4(l.normal, f0)), mul4(r.normal, f1));
    const float hsize = div_rn(norm4(h) / 2.0f, hunit);
    if (div_rn(radius, hunit) < hsize) return 0;
    if (1.0f < hsize) ftmp = div_rn(ftmp, fminf(2.0f, hsize));
    return ftmp < thr ? 1 : 0;
}

// List of (kind, view, cell): kind 0 = m_pgrids, 1 = m_vpgrids.  Snapshot lists are contiguous runs of ids (alive entries only), the
// geometry of an entry coming from its pool record; the destination cell being processed is read through its live id list instead.
struct ListRef { const int32_t* ids; int n; bool live; };
DEV ListRef cell_span(const DParams& prm, const CheckCtx& cx, int kind, int view, int cell) {
    if (kind == 0 && view == cx.live_view && cell == cx.live_cell) return {nullptr, cx.live_n, true};
    const int g = (prm.views + view)->cell_base + cell;
    if (kind == 0) return {prm.csr_id32 + prm.csr_start[g], prm.csr_cnt[g], false};
    return {prm.vcsr_id32 + prm.vcsr_start[g], prm.vcsr_cnt[g], false};
}
template <bool PK = false> DEV PGeo entry_geo(const DParams& prm, const CheckCtx& cx, const ListRef& l, int j, int& id) {
    if (l.live) { id = cx.live_ids[j]; return load_geo(patch_ptr(prm, cx, id)); }
    id = l.ids[j];
    return pool_geo<PK>(prm, id);  // a snapshot list names pool patches only
}

// Filter::computeGain, filter.cpp:108-146.  One lane per (view, list entry) pair -- a list holds at most
// MAX_NUM_OF_PATCHES entries after the trim, so one or two rounds of 64 pairs cover all views -- and the per-view maxima
// (a maximum does not depend on the order) are collected with LDS atomics.  tmp: MVS_LISTCAP ints of LDS.
template <bool PK = false> DEV void gain_part(const DParams& prm, const WaveCtx& wc, const CheckCtx& cx, const Cand& c, const PGeo& me, bool visible_part, int* tmp, float& gain) {
    const int nv = visible_part ? c.nvimg : c.nimg;
    if (nv == 0) return;
    int ln = 0, lcell = 0, lview = 0;  // view lane i: its cell list
    float pdepth = 0.0f;
    if (wc.lane < nv) {
        lview = visible_part ? c.vimg : c.img;
        const DView* vw = prm.views + lview;
        lcell = (visible_part ? c.vgy : c.gy) * vw->gw + (visible_part ? c.vgx : c.gx);
        ln = cell_span(prm, cx, 0, lview, lcell).n;
        if (visible_part) pdepth = dot4(ld4(vw->oaxis), c.coord);
    }
    int off = ln;  // inclusive prefix over the view lanes, then exclusive
#pragma unroll
    for (int d = 1; d < MVS_LISTCAP; d <<= 1) { const int t = __shfl_up(off, d); if (wc.lane >= d) off += t; }
    const int total = rli(off, nv - 1);
    off -= ln;
    __syncthreads();
    if (wc.lane < MVS_LISTCAP) tmp[wc.lane] = 0;  // +0.0f
    __syncthreads();
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + wc.lane;
        int owner = 0;  // the last view lane whose first pair is <= t
        for (int i = 1; i < nv; ++i) owner += (rli(off, i) <= t) ? 1 : 0;
        const int j = t - __shfl(off, owner);
        const int v = __shfl(lview, owner), cell = __shfl(lcell, owner);
        const float pd = __shfl(pdepth, owner);
        if (t < total) {
            const ListRef l = cell_span(prm, cx, 0, v, cell);
            int id;
            const PGeo g = entry_geo<PK>(prm, cx, l, j, id);
            bool counts = true;
            if (visible_part) counts = pd < dot4(ld4((prm.views + v)->oaxis), g.coord);
            if (counts && !is_neighbor(prm, me, g, prm.neighborThreshold1)) {
                const float val = g.ncc - prm.nccThreshold;
                if (val > 0.0f) atomicMax(&tmp[owner], __float_as_int(val));  // positive floats order like their bit patterns
            }
        }
    }
    __syncthreads();
    const float vmax = wc.lane < MVS_LISTCAP ? __int_as_float(tmp[wc.lane]) : 0.0f;
    for (int i = 0; i < nv; ++i) gain -= rlf(vmax, i);
}
template <bool PK = false> DEV float compute_gain(const DParams& prm, const WaveCtx& wc, const CheckCtx& cx, const Cand& c, int* tmp) {
    const PGeo me{c.coord, c.normal, c.dscale, c.ncc, rli(c.img, 0)};
    float gain = score2(c, prm.nccThreshold);
    gain_part<PK>(prm, wc, cx, c, me, false, tmp, gain);
    gain_part<PK>(prm, wc, cx, c, me, true, tmp, gain);
    return gain;
}

// PatchManager::findNeighbors, patch_manager.cpp:671-728 (scale 4, margin 2 as Optim::check calls it).
// Leaves the neighbour ids in `table[0..count)` (LDS, HCAP ints) and returns count, or -1 when the set does not fit.
//
// A patch shows up in the cells of several views, so the ids met in the nimg x (2 margin + 1)^2 cell lists go through
// a hash set first and the predicate runs once per id.  The set is an ORDERED linear-probing table (Amble & Knuth):
// a probe that meets a smaller key takes its slot and carries the smaller key on, which with atomicMax needs no lock
// and leaves -- whatever the order of the insertions -- the one layout in which every key sits behind larger keys only:
// the layout of inserting the keys in descending order with plain linear probing.  The slot order of that layout is the
// order in which filterQuad sums over the neighbours; the oracle rebuilds it the simple way, so no sort is needed here.
//   phase A: every lane takes one (view, cell) pair, loads the ids of its two lists four at a time and inserts them;
//   phase B: the table is streamed 64 slots at a time: record gather (next chunk's loads in flight), predicate,
//            ballot compaction of the accepted ids to the front of the table.
#define MVS_SET_EMPTY (-1)
// The id set and the rows normally live in LDS (G = false: plain accesses).  G = true is the second tier of Optim::check inside
// the sweep (k_sweep_retry): a neighbourhood that does not fit the wave's LDS gets a 16384-slot table in GLOBAL memory
// (SweepArgs::big_tables, one per block).  There every access is an agent-scope atomic load / store, i.e. served by L2: the
// table's atomicMax updates happen in L2, and a plain load could still find a line in this CU's vector cache from an earlier use.
template <bool G> DEV int tb_ld(const int* t, int i) {
    if (G) return __hip_atomic_load(t + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t[i];
}
template <bool G> DEV void tb_st(int* t, int i, int v) {
    if (G) __hip_atomic_store(t + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else t[i] = v;
}
template <bool G> DEV float tb_ldf(const float* t, int i) { return __int_as_float(tb_ld<G>(reinterpret_cast<const int*>(t), i)); }
template <bool G> DEV void tb_stf(float* t, int i, float v) { tb_st<G>(reinterpret_cast<int*>(t), i, __float_as_int(v)); }
// home slot of an id: multiplicative (Fibonacci) hashing, the top log2(capacity) bits of id * 2^32 / phi -- one multiply and
// a shift per list entry instead of the avalanche of mix32 (ids are dense small integers; the oracle's set_home is the same)
template <int HCAP> DEV unsigned set_home(int k) {
    static_assert((HCAP & (HCAP - 1)) == 0 && HCAP >= 64, "power of two");
    return ((uint32_t)k * 0x9E3779B1u) >> (32 - __builtin_ctz((unsigned)HCAP));
}
// the probe sequence of set_insert from slot p on (the key has already lost or given up its first slot)
DEV bool set_insert_from(int* table, unsigned mask, int k, unsigned p) {
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const int old = atomicMax(&table[p], k);
        if (old == k || old == MVS_SET_EMPTY) return true;
        if (old < k) k = old;
        p = (p + 1) & mask;
    }
    return false;
}
#define MVS_FN_INFLIGHT 4  // id loads in flight in the row walk (8, 16, 32 measured the same or slower)
// inclusive running maximum over the lanes (values >= -1): four row_shr steps inside the rows of 16, then row_bcast:15 / :31 carry the rows' ends on
DEV int wave_max_scan(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xc, 0xf, false));
    return x;
}
#ifndef MVS_FN_MARKS
#define MVS_FN_MARKS 1
#endif
#ifndef MVS_FN_JOINT_PROBE
#define MVS_FN_JOINT_PROBE 1
#endif
// `marks` (MK): 64 ints of LDS outside the table -- the run of an id by marks and a running maximum instead of a binary search, see below
// JP: the probe sequences of a lane's four keys side by side (see below) -- fewer waits, more instructions: the sweep, three waves per
// SIMD and waiting, gains 0.4 % from it; Filter::filterNeighbor, whose vector ALU is 89 % busy, loses 6 %
template <int HCAP, bool G = false, bool PK = false, bool MK = false, bool JP = false>
DEV int find_neighbors(const DParams& prm, const WaveCtx& wc, const CheckCtx& cx, const Cand& c, int* table, float scale, int margin,
                       unsigned* stats = nullptr, int* marks = nullptr) {
    const PGeo me{c.coord, c.normal, c.dscale, c.ncc, rli(c.img, 0)};
    // Propagate::computeRadius, propagate.cpp:474-481: the second smallest unit
    float u = __int_as_float(0x7f800000);
    float gu = 0.0f;
    if (wc.lane < c.nimg) {
        const DView* vw = prm.views + c.img;
        gu = get_unit(prm, vw, c.coord);
        u = gu;
        const F4 ray = nrm4(sub4(ld4(vw->center), c.coord));
        const float d = dot4(ray, c.normal);
        if (0.0f < d) u = div_rn(u, d); else u = (float)(INT_MAX / 2);
    }
    const float m1 = wave_min(u);
    const unsigned long long eq = ballot(wc.lane < c.nimg && u == m1);
    const int first = __ffsll((long long)eq) - 1;
    const float m2 = wave_min(wc.lane == first ? __int_as_float(0x7f800000) : u);
    const float second = c.nimg > 1 ? m2 : m1;
    const float radius = (float)(1.5 * (double)margin * (double)(second * (float)prm.csize));
    float unit = 0.0f;
    for (int i = 0; i < c.nimg; ++i) unit += rlf(gu, i);
    unit = div_rn(unit, (float)c.nimg);
    unit *= (float)prm.csize;
    const float thr = prm.neighborThreshold * scale;
    CK_BEGIN()
    __syncthreads();
    for (int t = wc.lane; t < HCAP; t += 64) tb_st<G>(table, t, MVS_SET_EMPTY);
    __syncthreads();
    // ---- phase A.  Every index the engine builds is dense -- the lists of neighbouring cells lie end to end -- so the (2 margin + 1)
    // cells of a grid row are ONE run of ids.  A lane per row (view, kind, dy) fetches its run, the runs are laid end to end and the
    // lanes take consecutive ids (the run of an id: a binary search over the running totals, which sit in the lanes): 64 useful ids
    // per load and per round of LDS atomics, where a lane per list ran as long as the longest list of 64.  (Filter::filterNeighbor has
    // walked its lists this way since round 3; inside the sweep the m_pgrids index had gaps where the trim had removed entries, and a
    // lane per (view, cell) list walked both lists of its cell: since round 4 the trimmed index is packed, k_index_pack.)
    // The destination cell being processed is read through its LIVE list: its row is cut in two around it -- the left part stays in
    // the row's own lane, the right part and the live list take two extra lanes behind the regular rows.
    // The final layout does not depend on the order of the insertions.
    const int side = 2 * margin + 1;
    const int nimg = __builtin_amdgcn_readfirstlane(c.nimg);
    const int nrow = 2 * side * nimg;
    // is the live cell inside the window of its view?
    int live_i = -1, lcx = 0, lcy = 0;
    if (cx.live_view >= 0) {
        const unsigned long long lm = ballot(wc.lane < nimg && c.img == cx.live_view);
        if (lm) {
            const int i = __ffsll((long long)lm) - 1;
            const int gw = (prm.views + cx.live_view)->gw;
            lcx = cx.live_cell % gw; lcy = cx.live_cell / gw;
            const int gx = rli(c.gx, i), gy = rli(c.gy, i);
            if (abs(lcx - gx) <= margin && abs(lcy - gy) <= margin) live_i = i;
        }
    }
    const int nrow_all = nrow + (live_i >= 0 ? 2 : 0);
    bool full = false;
    unsigned n_entries = 0;
    for (int r0 = 0; r0 < nrow_all; r0 += 64) {
        const int r = r0 + wc.lane, rc = min(r, nrow - 1);
        const bool right = live_i >= 0 && r == nrow, livel = live_i >= 0 && r == nrow + 1;
        const int i = right ? live_i : rc / (2 * side), dy = rc % side - margin;
        const bool vk = !right && ((rc / side) & 1) != 0;
        const int v = __shfl(c.img, i), gx = __shfl(c.gx, i), gy = __shfl(c.gy, i);
        csr_off_t b = 0;
        int len = 0, src = vk ? 1 : 0;  // 0: m_pgrids ids, 1: m_vpgrids ids, 2: the live list (LDS)
        if (livel) { len = cx.live_n; src = 2; }
        else if (r < nrow_all) {
            const DView* vw = prm.views + v;
            const int yt = right ? lcy : gy + dy;
            int x0 = max(gx - margin, 0), x1 = min(gx + margin, vw->gw - 1);
            if (right) x0 = lcx + 1;
            else if (live_i == i && !vk && yt == lcy) x1 = min(x1, lcx - 1);  // the live cell's row: its part left of the cell
            if (0 <= yt && yt < vw->gh && x0 <= x1) {
                const csr_off_t* st = vk ? prm.vcsr_start : prm.csr_start;
                const int g0 = vw->cell_base + yt * vw->gw + x0;
                b = st[g0];
                len = (int)(st[g0 + (x1 - x0) + 1] - b);
            }
        }
        int P = len;  // the running total over the lanes: inclusive, then exclusive
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(P, d); if (wc.lane >= d) P += o; }
        const int total = __builtin_amdgcn_readlane(P, 63);
        P -= len;
        if (wc.lane == 0) n_entries += (unsigned)total;
        // (MK) The run of id k without a search per id: every run marks the position it begins at -- its lane number, a byte, in a
        // 256-position window laid out so that lane l reads positions l, 64 + l, 128 + l, 192 + l as ONE dword -- and the run of a
        // position is the running maximum of the marks up to it (runs begin in lane order): four DPP scans and three carries for 256
        // ids, where the binary search over the running totals took six ds_bpermute round trips per 64.  What an id needs of its run
        // is then one 64-bit value, the address its k is an index from (the live list: a tag and the LDS index).
        unsigned blo = 0u, bhi = 0u;
        if (MK) {
            const unsigned long long base = src == 2 ? ((0x7fffffffull << 32) | (unsigned)(-P))
                                                     : (unsigned long long)((long long)(uintptr_t)(src == 1 ? prm.vcsr_id32 : prm.csr_id32) + 4ll * (long long)(b - (csr_off_t)P));
            blo = (unsigned)base; bhi = (unsigned)(base >> 32);
        }
        int carry = 0;
        for (int k0 = 0; k0 < total; k0 += 64 * MVS_FN_INFLIGHT) {
            int id[MVS_FN_INFLIGHT];
            if (MK) {
                static_assert(MVS_FN_INFLIGHT == 4, "one byte per 64 positions in a dword of marks");
                marks[wc.lane] = -1;
                const int pm = P - k0;
                if (len > 0 && (unsigned)pm < 256u) reinterpret_cast<signed char*>(marks)[(pm & 63) * 4 + (pm >> 6)] = (signed char)wc.lane;
                __syncthreads();
                const int m4 = marks[wc.lane];
                int run[4] = {(m4 << 24) >> 24, (m4 << 16) >> 24, (m4 << 8) >> 24, m4 >> 24};
#pragma unroll
                for (int q = 0; q < 4; ++q) run[q] = wave_max_scan(run[q]);
                run[0] = max(run[0], carry);
#pragma unroll
                for (int q = 1; q < 4; ++q) run[q] = max(run[q], __builtin_amdgcn_readlane(run[q - 1], 63));
                carry = __builtin_amdgcn_readlane(run[3], 63);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = k0 + 64 * q + wc.lane;
                    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(run[q] << 2, (int)blo), hi = (unsigned)__builtin_amdgcn_ds_bpermute(run[q] << 2, (int)bhi);
                    int v_id = MVS_SET_EMPTY;
#if MVS_FN_MARKS == 2  // debugging: the address by the search as well; loads through that one and reports a difference
                    {
                        int lo2 = 0;
                        for (int step = 32; step >= 1; step >>= 1) { const int pc = __shfl(P, lo2 + step); if (pc <= k) lo2 += step; }
                        const csr_off_t bb = __shfl(b, lo2);
                        const int pl = __shfl(P, lo2), sr = __shfl(src, lo2);
                        if (k < total) {
                            const int32_t* want = sr == 2 ? nullptr : (sr == 1 ? prm.vcsr_id32 : prm.csr_id32) + (bb + (k - pl));
                            const int32_t* got = reinterpret_cast<const int32_t*>(((unsigned long long)hi << 32) | lo) + k;
                            const bool livegot = live_i >= 0 && hi == 0x7fffffffu;
                            if ((sr == 2) != livegot || (sr == 2 ? ((int)lo + k != k - pl) : (want != got)))
                                printf("[marks] block %d lane %d k %d total %d run %d search %d sr %d hi %08x lo %08x want %p got %p live_i %d nrow_all %d r0 %d k0 %d\n", (int)blockIdx.x, wc.lane, k, total, run[q], lo2, sr, hi, lo, want, got, live_i, nrow_all, r0, k0);
                            v_id = sr == 2 ? cx.live_ids[k - pl] : *want;
                        }
                    }
#else
                    // Two loads in their own address spaces.  As ONE generic pointer the compiler folds the "+ 64 q" of k into the flat
                    // instruction's offset field, and the address without it lies below the LDS aperture for the first ids of the live list
                    // (k - 64 q < P): the hardware picks the aperture before it adds the offset -- a memory aperture violation.
                    if (k < total) {
                        typedef const int32_t __attribute__((address_space(1)))* gptr_i32;
                        typedef const int __attribute__((address_space(3)))* lptr_i32;
                        if (live_i >= 0 && hi == 0x7fffffffu) v_id = ((lptr_i32)cx.live_ids)[(int)lo + k];
                        else v_id = ((gptr_i32)(((unsigned long long)hi << 32) | lo))[k];
                    }
#endif
                    id[q] = v_id;
                }
            } else
#pragma unroll
            for (int q = 0; q < MVS_FN_INFLIGHT; ++q) {
                const int k = k0 + 64 * q + wc.lane;
                int lo = 0;  // the run id k falls in: the last lane whose run begins at or before k
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) { const int pc = __shfl(P, lo + step); if (pc <= k) lo += step; }
                const csr_off_t bb = __shfl(b, lo);
                const int pl = __shfl(P, lo), sr = __shfl(src, lo);
                int v_id = MVS_SET_EMPTY;
                if (k < total) v_id = sr == 2 ? cx.live_ids[k - pl] : (sr == 1 ? prm.vcsr_id32 : prm.csr_id32)[bb + (k - pl)];
                id[q] = v_id;
            }
#pragma unroll
            for (int u0 = 0; u0 < MVS_FN_INFLIGHT; u0 += 4) {
                if (k0 + 64 * u0 >= total) break;
                unsigned slot[4];
                int old[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    slot[q] = set_home<HCAP>(id[u0 + q]);
                    old[q] = id[u0 + q] != MVS_SET_EMPTY ? atomicMax(&table[slot[q]], id[u0 + q]) : id[u0 + q];
                }
                if (JP) {
                // the keys that lost or took an occupied slot carry on from the next slot -- the four of a lane side by side, one
                // round of atomics for all of them per step (four probe loops one after the other waited four times as often)
                int key[4];
                bool pend[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pend[q] = !(old[q] == id[u0 + q] || old[q] == MVS_SET_EMPTY);
                    key[q] = old[q] < id[u0 + q] ? old[q] : id[u0 + q];
                    slot[q] = (slot[q] + 1) & (HCAP - 1);
                }
                for (unsigned probe = 0; ballot(pend[0] | pend[1] | pend[2] | pend[3]) != 0ull; ++probe) {
                    if (probe > (unsigned)(HCAP - 1)) { full = true; break; }
                    int o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = pend[q] ? atomicMax(&table[slot[q]], key[q]) : key[q];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (o[q] == key[q] || o[q] == MVS_SET_EMPTY) pend[q] = false;
                        else if (o[q] < key[q]) key[q] = o[q];
                        slot[q] = (slot[q] + 1) & (HCAP - 1);
                    }
                }
                } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (old[q] == id[u0 + q] || old[q] == MVS_SET_EMPTY) continue;
                    if (!set_insert_from(table, HCAP - 1, old[q] < id[u0 + q] ? old[q] : id[u0 + q], (slot[q] + 1) & (HCAP - 1))) full = true;
                }
                }
            }
        }
    }
    __syncthreads();
    CK_ADD(9)
    if (ballot(full)) return -1;
    // ---- phase B: the visited ids to the front of the table (slot order kept), then record gather + predicate over
    // that dense list, the next 64 records in flight while the current ones are tested
    int visited = 0;
    for (int k = 0; k < HCAP / 64; k += 4) {  // four chunks of 64 slots per step: their reads are in flight together
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = tb_ld<G>(table, (k + u) * 64 + wc.lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long m = ballot(v[u] >= 0);
            if (v[u] >= 0) tb_st<G>(table, visited + __popcll(m & ((1ull << wc.lane) - 1ull)), v[u]);  // visited + rank <= 64 (k + u) + lane: already read
            visited += (int)__popcll(m);
        }
    }
    __syncthreads();
    int count = 0;
    int v_next = wc.lane < visited ? tb_ld<G>(table, wc.lane) : -1;
    auto geo_of = [&](int id) { return PK ? pool_geo<true>(prm, id) : load_geo(patch_ptr(prm, cx, id)); };  // PK: no live list, no staged records
    PGeo g_next = geo_of(max(v_next, 0));  // lanes past the end read record 0: harmless, unused
    for (int b0 = 0; b0 < visited; b0 += 64) {
        const int v_cur = v_next;
        const PGeo g_cur = g_next;
        if (b0 + 64 < visited) {
            v_next = b0 + 64 + wc.lane < visited ? tb_ld<G>(table, b0 + 64 + wc.lane) : -1;
            g_next = geo_of(max(v_next, 0));
        }
        const bool acc = v_cur >= 0 && is_neighbor_radius(prm, me, g_cur, unit, thr, radius);
        const unsigned long long m = ballot(acc);
        if (acc) tb_st<G>(table, count + __popcll(m & ((1ull << wc.lane) - 1ull)), v_cur);  // count + rank <= b0 + lane: already read
        count += (int)__popcll(m);
    }
    __syncthreads();
    CK_ADD(10)
#ifdef MVS_STAGE_TIMING
    if (cx.st) { cx.st[12] = cx.st[12] > (unsigned long long)visited ? cx.st[12] : (unsigned long long)visited; cx.st[13] = cx.st[13] > (unsigned long long)count ? cx.st[13] : (unsigned long long)count; }
#endif
    if (visited > HCAP - HCAP / 8) return -1;  // beyond 7/8 full the oracle's table-size rule picks the next size
    if (stats) {
        for (int d = 32; d >= 1; d >>= 1) n_entries += (unsigned)__shfl_xor((int)n_entries, d);
        stats[0] = 2u * (unsigned)(nimg * side * side); stats[1] = n_entries; stats[2] = (unsigned)visited; stats[3] = (unsigned)count;
    }
    return count;
}

// Filter::ortho, filter.cpp:394-409
DEV void ortho(F4 z, F4& x, F4& y) {
    if (fabsf(z.x) > 0.5f) x = {z.y, -z.x, 0.0f, 0.0f};
    else if (fabsf(z.y) > 0.5f) x = {0.0f, z.z, -z.y, 0.0f};
    else x = {-z.z, 0.0f, z.x, 0.0f};
    const float n = norm4(x);
    x = {div_rn(x.x, n), div_rn(x.y, n), div_rn(x.z, n), div_rn(x.w, n)};
    y = {z.y * x.z - z.z * x.y, z.z * x.x - z.x * x.z, z.x * x.y - z.y * x.x, 0.0f};
}

// double-precision wave butterfly, same pairing order as wave_sum
template <int CTRL, int RM> DEV double dpp_d(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, RM, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RM, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
DEV double wave_sum_f64(double x) {
    x = x + dpp_d<0xB1, 0xf>(x);
    x = x + dpp_d<0x4E, 0xf>(x);
    x = x + dpp_d<0x141, 0xf>(x);
    x = x + dpp_d<0x140, 0xf>(x);
    x = x + dpp_d<0x142, 0xa>(x);
    x = x + dpp_d<0x143, 0xc>(x);
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// the same butterfly, the sum left in lane 63 only (no v_readlane: the value stays in vector registers)
DEV double wave_sum_f64_l63(double x) {
    x = x + dpp_d<0xB1, 0xf>(x);
    x = x + dpp_d<0x4E, 0xf>(x);
    x = x + dpp_d<0x141, 0xf>(x);
    x = x + dpp_d<0x140, 0xf>(x);
    x = x + dpp_d<0x142, 0xa>(x);
    x = x + dpp_d<0x143, 0xc>(x);
    return x;
}
// N sums per lane -> one lane per sum, through the same tree of additions as wave_sum_f64 (neighbouring lanes, pairs of pairs, ...,
// rows, pairs of rows, halves): at the step over distance D the lower lane of a pair keeps the first (N + 1) / 2 of the sums and hands
// over the rest, the upper lane the other way round, and each adds what it receives to what it keeps -- the two operands of every
// addition are those of the butterfly (addition commutes), but N + N/2 + N/4 + ... of them are made instead of 6 N.
template <int D> DEV double xchg_f64(double x) {
    if (D == 1) return dpp_d<0xB1, 0xf>(x);  // quad_perm [1, 0, 3, 2]
    if (D == 2) return dpp_d<0x4E, 0xf>(x);  // quad_perm [2, 3, 0, 1]
    const long long b = __double_as_longlong(x);
    const int lo = __shfl_xor((int)(b & 0xffffffffll), D), hi = __shfl_xor((int)(b >> 32), D);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
template <int N, int D> DEV void scatter_step_f64(double* v, int lane) {
    constexpr int K = (N + 1) / 2;
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double first = v[j], second = (K + j < N) ? v[K + j] : 0.0;
        const double keep = up ? second : first, send = up ? first : second;
        v[j] = keep + xchg_f64<D>(send);
    }
}
// v[0..20) in every lane -> the wave's sum k in v[0] of lane owner(k); returns the k this lane owns, or -1
DEV int wave_sums20_f64(double* v, int lane) {
    scatter_step_f64<20, 1>(v, lane);
    scatter_step_f64<10, 2>(v, lane);
    scatter_step_f64<5, 4>(v, lane);
    scatter_step_f64<3, 8>(v, lane);
    scatter_step_f64<2, 16>(v, lane);
    scatter_step_f64<1, 32>(v, lane);
    // which sum ended up in this lane's v[0]: every step works on the same array length N in all lanes (20, 10, 5, 3, 2, 1), of which
    // a lane's first `cnt` entries are sums (the upper lane of a step may get fewer than the lower one), v[0] being sum `off`
    int off = 0, cnt = 20, len = 20;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int k = (len + 1) / 2;
        if (lane & d) { off += k; cnt = max(cnt - k, 0); } else cnt = min(cnt, k);
        len = k;
    }
    return cnt >= 1 ? off : -1;
}
DEV double shfl_f64(double x, int src) {
    const long long b = __double_as_longlong(x);
    const int lo = __shfl((int)(b & 0xffffffffll), src), hi = __shfl((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// Filter::filterQuad, filter.cpp:329-392.  nb = sorted ids in LDS (n of them); rows = LDS scratch of 3*n floats
// (fx, fy, fz per neighbour).  The three sums over the neighbours (mean distance, normal equations, residual) are
// lane-strided partial sums (lane l takes neighbours l, l+64, ...) followed by a wave butterfly.
template <bool G = false, bool PK = false>
DEV int filter_quad(const DParams& prm, const WaveCtx& wc, const CheckCtx& cx, const Cand& c, const int* nb, int n, float* rows, double* sums_lds = nullptr) {
    F4 xdir, ydir;
    ortho(c.normal, xdir, ydir);
    // one pass over the neighbours' records: the distance for the mean h, and the three projections of the offset, which wait
    // in the rows for the division by h
    float hp = 0.0f;
    for (int t = wc.lane; t < n; t += 64) {
        const int qid = tb_ld<G>(nb, t);
        F4 qc;
        if (PK) { const float4 a = prm.geo[2 * (size_t)qid]; qc = {a.x, a.y, a.z, 1.0f}; }
        else qc = ld4(patch_ptr(prm, cx, qid)->coord);
        const F4 diff = sub4(qc, c.coord);
        hp += norm4(diff);
        tb_stf<G>(rows, 3 * t + 0, dot4(diff, xdir)); tb_stf<G>(rows, 3 * t + 1, dot4(diff, ydir)); tb_stf<G>(rows, 3 * t + 2, dot4(diff, c.normal));
    }
    float h = wave_sum(hp);
    h = div_rn(h, (float)n);
    // Filter::lls: M = [A^T A | A^T b] with A = (fx^2, fy^2, fx fy, fx, fy), b = fz; 15 + 5 distinct sums, in double
    double acc[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = 0.0;
    for (int t = wc.lane; t < n; t += 64) {  // each lane reads back what it wrote
        const float fx = div_rn(tb_ldf<G>(rows, 3 * t + 0), h), fy = div_rn(tb_ldf<G>(rows, 3 * t + 1), h), fz = tb_ldf<G>(rows, 3 * t + 2);
        tb_stf<G>(rows, 3 * t + 0, fx); tb_stf<G>(rows, 3 * t + 1, fy);
        const float a[6] = {fx * fx, fy * fy, fx * fy, fx, fy, fz};
        int k = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {  // the product of two floats is exact in double: fma IS the reference's multiply-then-add, bit for bit
                if (j == 5) acc[15 + i] = __builtin_fma((double)a[i], (double)a[5], acc[15 + i]);
                else { acc[k] = __builtin_fma((double)a[i], (double)a[j], acc[k]); ++k; }
            }
    }
    // The 20 sums go through LDS into lanes: lane 6 r + c holds M[r][c] (A^T A | A^T b, symmetric part mirrored), and the
    // pivoted elimination of Filter::lls runs with one matrix element per lane -- the same operations in the same order as
    // the oracle's solve5 (every element update is f = M[r][col] / M[col][col]; M[r][k] -= f * M[col][k]), but 2 VGPRs of
    // matrix instead of 60 uniform registers.
    double* sums = sums_lds ? sums_lds : reinterpret_cast<double*>(rows + 3 * ((n + 1) & ~1));  // behind the rows, 8-byte aligned (always LDS)
    __syncthreads();
    {
        const int mine = wave_sums20_f64(acc, wc.lane);
        if (mine >= 0) sums[mine] = acc[0];
    }
    __syncthreads();
    const int lr = min(wc.lane / 6, 4), lc = wc.lane % 6;
    double m;
    {
        const int i = min(lr, lc), j = max(lr, lc);
        const int idx = lc == 5 ? 15 + lr : i * 5 - (i * (i - 1)) / 2 + (j - i);
        m = sums[idx];
    }
    // Rows are not moved when a pivot is chosen: perm[r] names the lane row that holds logical row r (the row exchange of the
    // oracle's solve5 becomes an exchange of two entries of perm), and the pivot's value comes out of the search itself.  Two
    // ds_bpermute round trips per column (the column for the search; the multiplier and the pivot row for the update) and one
    // for the whole back substitution.
    double x[5] = {0, 0, 0, 0, 0};
    bool solved = true;
    int perm[5] = {0, 1, 2, 3, 4};
    unsigned done = 0u;  // lane rows already used as pivot rows
#pragma unroll
    for (int col = 0; col < 5; ++col) {
        double cv[5];
#pragma unroll
        for (int r = col; r < 5; ++r) cv[r] = shfl_f64(m, perm[r] * 6 + col);
        int piv = col;
        double pv = cv[col], best = fabs(cv[col]);
#pragma unroll
        for (int r = col + 1; r < 5; ++r) {
            const double av = fabs(cv[r]);
            if (av > best) { best = av; piv = r; pv = cv[r]; }
        }
#pragma unroll
        for (int r = col + 1; r < 5; ++r) {
            if (piv == r) { const int t = perm[col]; perm[col] = perm[r]; perm[r] = t; }
        }
        done |= 1u << perm[col];
        if (fabs(pv) < 1e-30) solved = false;
        if (solved) {
            const double f = shfl_f64(m, lr * 6 + col) / pv;
            const double pc = shfl_f64(m, perm[col] * 6 + lc);
            if (!((done >> lr) & 1u) && lc >= col) m -= f * pc;
        }
    }
    if (solved) {
        double mr[5][6];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int k = r; k < 6; ++k) mr[r][k] = shfl_f64(m, perm[r] * 6 + k);
#pragma unroll
        for (int r = 4; r >= 0; --r) {
            double a2 = mr[r][5];
#pragma unroll
            for (int k = r + 1; k < 5; ++k) a2 -= mr[r][k] * x[k];
            x[r] = a2 / mr[r][r];
        }
    }
    const float x0 = solved ? (float)x[0] : 0.0f, x1 = solved ? (float)x[1] : 0.0f, x2 = solved ? (float)x[2] : 0.0f,
                x3 = solved ? (float)x[3] : 0.0f, x4 = solved ? (float)x[4] : 0.0f;
    const int inum = min(prm.tau, c.nimg);
    float unit = 0.0f;
    {
        float gu = 0.0f;
        if (wc.lane < inum) gu = get_unit(prm, prm.views + c.img, c.coord);
        for (int i = 0; i < inum; ++i) unit += rlf(gu, i);
        unit = div_rn(unit, (float)inum);
    }
    float rp = 0.0f;
    for (int t = wc.lane; t < n; t += 64) {
        const float fx = tb_ldf<G>(rows, 3 * t), fy = tb_ldf<G>(rows, 3 * t + 1), fz = tb_ldf<G>(rows, 3 * t + 2);
        const float res = x0 * (fx * fx) + x1 * (fy * fy) + x2 * (fx * fy) + x3 * fx + x4 * fy - fz;
        rp += div_rn(fabsf(res), unit);
    }
    float residual = wave_sum(rp);
    residual = div_rn(residual, (float)(n - 5));
    return residual < prm.quadThreshold ? 0 : 1;
}

#ifndef MVS_CHECK_OUTLINE
#define MVS_CHECK_OUTLINE 0
#endif
#if MVS_CHECK_OUTLINE
#define MVS_CHECK_FN __device__ __noinline__
#else
#define MVS_CHECK_FN DEV
#endif
// Optim::check, optim.cpp:300-323.  lds: MVS_CHECK_LDS_FLOATS floats (the kernel's dynamic LDS region).
// Returns 1 when the patch is rejected.  Neighbours beyond MVS_ROW_CAP are ignored (and flagged in *overflow).
// Optim::check, optim.cpp:300-323.  lds: MVS_CHECK_LDS_FLOATS floats (the kernel's dynamic LDS region).
// Returns 1 when the patch is rejected, 0 when it passes, and -1 when its 5x5-cell neighbourhood does not fit the wave's LDS id
// set / row buffer (more than 7/8 of MVS_HASH_CAP distinct patches met, or more than MVS_ROW_CAP neighbours) and BIG is false:
// the reference's findNeighbors (patch_manager.cpp:671-728) is unbounded, so the caller hands the whole destination cell to
// the second tier.  BIG = true IS that second tier (k_sweep_retry; rare, exact, slow): the search runs again on a
// MVS_FILTER2_HASH_CAP-slot table in global memory (`big_table`: the size Filter::filterNeighbor's second launch uses -- the
// oracle's table-size rule is "the small set if it fits, else 16384"), filterQuad keeps its rows there and its 20 sums in LDS.
template <bool BIG = false>
MVS_CHECK_FN int check_patch(const DParams& prm, const WaveCtx& wc, const CheckCtx& cx, Cand& c, float* lds, int* overflow, int* big_table = nullptr) {
#ifndef MVS_CHECK_STAGES
#define MVS_CHECK_STAGES 3  // timing experiments only: 1 = gain, 2 = + neighbours, 3 = everything
#endif
    CK_BEGIN()
    const float gain = compute_gain(prm, wc, cx, c, reinterpret_cast<int*>(lds));  // the set's LDS is free until findNeighbors
    CK_ADD(8)
    c.tmp = gain;
    if (gain < 0.0f) { c.nimg = 0; return 1; }
    if (MVS_CHECK_STAGES < 2) return 0;
    int* table = reinterpret_cast<int*>(lds);
    static_assert(MVS_HASH_CAP + 64 <= MVS_CHECK_LDS_FLOATS, "the marks of the row walk sit behind the id set");
    int n = find_neighbors<MVS_HASH_CAP, false, false, MVS_FN_MARKS != 0, MVS_FN_JOINT_PROBE != 0>(prm, wc, cx, c, table, 4.0f, 2, nullptr, table + MVS_HASH_CAP);
    if (n < 0 || n > MVS_ROW_CAP) {
        if (!BIG) return -1;
        n = find_neighbors<MVS_FILTER2_HASH_CAP, true>(prm, wc, cx, c, big_table, 4.0f, 2);
        if (n < 0 || n > MVS_FILTER2_ROW_CAP) { if (wc.lane == 0) atomicOr(overflow, 4); return 0; }  // the engine's limit
        if (6 < n && filter_quad<true>(prm, wc, cx, c, big_table, n, reinterpret_cast<float*>(big_table) + rows_offset(n), reinterpret_cast<double*>(lds))) { c.nimg = 0; return 1; }
        return 0;
    }
    if (MVS_CHECK_STAGES < 3) return 0;
    if (6 < n) {
        CK_BEGIN()
        const int fq = filter_quad(prm, wc, cx, c, table, n, lds + rows_offset(n));
        CK_ADD(11)
        if (fq) { c.nimg = 0; return 1; }
    }
    return 0;
}

}  // namespace mvsdev
