// mvs_types.h -- device-side data layout of the MI355X PatchMatch-MVS engine (internal).
#pragma once
#include <stdint.h>

#define MVS_WAVE 64
#define MVS_MAXLEV 7       // level (<=4) + 3 pyramid levels (pmmvps.cpp:36)
#ifndef MVS_LISTCAP
#define MVS_LISTCAP 16     // m_images / m_vimages are truncated to 16 views
#endif
#if MVS_LISTCAP > 32
#define MVS_MAXI 64        // the 64-view build (libmvskit_engine_cap64.so): records of 192 bytes, no list is ever cut (nviews <= 64)
#else
#define MVS_MAXI 32        // storage in the record
#endif
#define MVS_MAXVIEWS 64    // one lane per view in the per-view phases
#define MVS_CAPMAX 32      // MAX_NUM_OF_PATCHES = max_propag * csize^2 <= 32
#define MVS_NEWBASE 0x40000000  // ids >= NEWBASE: staged record NEWBASE + slot

// Patch record (pmmvps/patch.hpp:33-66); same bytes as mvs_patch in include/mvskit_engine.h.
#define MVS_FLAG_ALIVE 1
#define MVS_FLAG_SETTLED 2  // m_images is what Filter::filterExact's setRefImage made of its current set of views (k_filter_exact)
struct DPatch {
    float coord[4];
    float normal[4];
    float ncc, dscale, ascale, tmp;
    int32_t nimages, nvimages, flags, id;
    uint8_t images[MVS_MAXI];
    uint8_t vimages[MVS_MAXI];
};
static_assert(sizeof(DPatch) == 64 + 2 * MVS_MAXI, "record is 128 bytes (192 in the 64-view build)");
#define MVS_REC_U4 (sizeof(DPatch) / 16)  // a record as 16-byte words

// The cell index (PatchManager::m_pgrids / m_vpgrids as lists, patch_manager.hpp:90-104), rebuilt from the pool before every pass:
// per membership -- a patch sits in the list of one cell in each of its views -- the 4-byte patch id, for m_pgrids also an 8-byte
// ListKey (m_ncc and the reference view: what the sweep reads of its own and its source cells), 64-bit offsets per cell.  The geometry
// of a listed patch comes from its pool record (one cache line).  With the transient sort keys of a build (one buffer serves both
// grids) that is 12 bytes per membership and no 2^31 limit on patches x views per patch -- rounds 1-3 kept a 48-byte entry with the
// geometry per membership behind 32-bit offsets (60 bytes, 55 M patches at 48 views); measured on the 16-view build the id lists
// are as fast in the sweep and faster in Filter::run's rebuilds.
typedef int64_t csr_off_t;
struct ListKey {
    float ncc;
    int32_t ref;
};
static_assert(sizeof(ListKey) == 8, "list key is 8 bytes");

// One view resident in HBM: camera (image/camera.cpp:65-100), Optim axes (optim.cpp:43-65), pyramid
// (image/image.cpp:245-315) as RGBA8 texels (one 32-bit load per texel), mask at m_level, grid size.
struct DView {
    float P[MVS_MAXLEV][12];
    float Minv[9];  // inverse of the 3x3 block of P[level]
    float center[4];
    float oaxis[4];
    float xaxis[3], yaxis[3], zaxis[3];
    float ipscale;
    int32_t W[MVS_MAXLEV], H[MVS_MAXLEV];
    const uint32_t* img[MVS_MAXLEV];
    const uint8_t* mask;  // level m_level, or null
    int32_t gw, gh;
    int32_t cell_base;  // offset of this view's cells in the concatenated per-cell arrays
    int32_t pad;
};

// Kernel-constant parameters (Option + PmMvps thresholds, pmmvps.cpp:18-68).
struct DParams {
    int32_t nviews, level, csize, wsize, wsz, minImageNum, tau, cap, max_propag, depth, enable_check, view_propagation;
    uint32_t seed;
    int32_t refine_steps;
    float rd0, ra0;
    float nccThreshold, nccThresholdBefore;
    float cosAngle0, cosAngle1, cosMinAngle, cosMaxAngle, cosNeighborTypo, cosNeighbor120;
    float sortThreshold, ascaleConst, neighborThreshold, neighborThreshold1, quadThreshold;
    float inv_sz, inv_3sz;  // 1/wsize^2 and 1/(3 wsize^2)
    int32_t total_cells;
    int32_t list_n;         // min(MVS_LISTCAP, nviews): no list is longer; sizes the kept textures of setRefImage
    int32_t gram_ld;        // many-view builds: row pitch of setRefImage's Gram matrix in LDS = list_n rounded up to a chunk of 16 views
    const DView* views;
    DPatch* pool;
    int64_t pool_n;
    // index (rebuilt every pass): cell g holds entries [csr_start[g], csr_start[g] + csr_cnt[g]) -- alive patches only, m_pgrids sorted
    // (ncc desc, id asc) in Propagate::run
    const csr_off_t* csr_start;
    const int32_t* csr_cnt;
    const ListKey* csr_key;    // (m_ncc, reference view), parallel to csr_id32; valid after a Propagate::run index build
    const int32_t* csr_id32;   // the ids: findNeighbors' first phase walks nothing else (4 B per patch met)
    const csr_off_t* vcsr_start;
    const int32_t* vcsr_cnt;
    const int32_t* vcsr_id32;
    // Inside Filter::run only (else null): the geometry the neighbour predicates read of a listed patch, packed to 32 bytes -- two
    // 16-byte words per patch, (coord.xyz, m_dscale) and (normal.xyz, m_ncc) -- and its reference view as a byte: one 32-byte sector per
    // patch met instead of two or three cache lines of its record (k_geo_pack; only when every coord.w is 1 and every normal.w is 0).
    const float4* geo;
    const uint8_t* geo_ref;
    const unsigned long long* dpgrid;  // (sortable depth << 32 | id), ~0ull = m_MAXDEPTH
};

#ifndef MVS_COUNTER_SLOTS
#define MVS_COUNTER_SLOTS 256  // SweepArgs::counters: that many DCounters, job j adds into slot j % MVS_COUNTER_SLOTS
#endif
struct DCounters {
    unsigned long long candidates, prefiltered, patches, fail0, fail1, inserted, replaced, evals, view_evals, trimmed;
    // diagnostic build (-DMVS_STAGE_TIMING): wave cycles (s_memtime) per stage of the sweep, summed over waves:
    // 0 whole wave, 1 generatePatch, 2 preProcess, 3 refinePatch, 4 postProcess, 5 check, 6 staging/insert, 7 prologue
    // 8 computeGain, 9 findNeighbors search, 10 compaction + sort, 11 filterQuad
    unsigned long long stage[16];
};

// Arguments of one sweep launch (one colour pass over the owned views).
struct SweepArgs {
    int32_t iter, inc, colour;
    int32_t nsweep_views;
    int32_t sweep_views[MVS_MAXVIEWS];
    int32_t job_base[MVS_MAXVIEWS];  // first job id of each swept view
    int32_t halfw_max, gh_max;
    int64_t njobs;
    int64_t job_lo, job_hi;  // the jobs this launch runs: [job_lo, job_hi) of [0, njobs) (sharding by job range)
    DPatch* staging;
    int64_t staging_cap;
    unsigned long long* stage_counter;
    int32_t* job_stage;   // [njobs][maxstage] staging slots in creation order
    int32_t* job_nstage;  // [njobs]
    int32_t maxstage;
    uint8_t* kill;        // [pool_cap]
    DCounters* counters;
    int32_t* error_flag;
    int32_t* big_tables;  // Optim::check's second tier (k_sweep_retry): 16384-slot id sets in global memory, one per block
    int32_t* retry_jobs;  // the destination cells (jobs) the first launch handed to the second tier, and
    int32_t* nretry;      //   how many
};
