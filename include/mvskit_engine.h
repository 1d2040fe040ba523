/*
 * mvskit_engine.h -- C ABI of the MI355X PatchMatch-MVS propagation/optimisation engine.
 *
 * This is the drop-in boundary for the hot path of imkaywu/MVSKit:
 *     PmMvps::run  ->  Propagate::run(iter)                       pmmvps/pmmvps.cpp:95
 *         -> propagatePmImage / propagatePatch / generatePatch    pmmvps/propagate.cpp:72-237
 *         -> Optim::preProcess / refinePatch / postProcess        pmmvps/optim.cpp:137-547
 *         -> PatchManager grid operations                         pmmvps/patch_manager.cpp:158-433
 * The reference has no FFI: its boundary is the C++ object graph PmMvps owns by value
 * (pmmvps/pmmvps.hpp:93-103).  A maintainer replaces the body of Propagate::run with calls to the
 * entry points below (INTEGRATION.md shows the patch); mvskit_amd/host/ holds a host-side mirror
 * of Option / PhotoSet / PatchManager / Propagate / PmMvps that does exactly that.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every call returns 0 on success
 * and a negative mvs_status otherwise (never exit(); the reference exit(1)s, e.g.
 * pmmvps/propagate.cpp:39-42).  Host buffers stay owned by the caller; the engine owns all device
 * memory.  One host thread per handle.  All device work runs on one HIP stream per handle.
 */
#ifndef MVSKIT_ENGINE_H
#define MVSKIT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MVS_MAX_IMAGES
#define MVS_MAX_IMAGES 32 /* storage of Patch::m_images / m_vimages in a record; 64 for libmvskit_engine_cap64.so (compile callers with -DMVS_MAX_IMAGES=64) */
#endif
#define MVS_LIST_CAP 16   /* list length of the DEFAULT build (libmvskit_engine.so); the cap32 / cap64 builds hold 32 / 64 views per list.  A library
                           * that holds the data set's view count never cuts a list; ask the loaded one with mvs_list_cap() */

typedef enum mvs_status {
    MVS_OK = 0,
    MVS_ERR_ARG = -1,      /* bad argument / configuration */
    MVS_ERR_STATE = -2,    /* call out of order (views not set, ...) */
    MVS_ERR_HIP = -3,      /* HIP runtime error, see mvs_last_error() */
    MVS_ERR_CAPACITY = -4, /* patch pool or staging capacity exceeded */
    MVS_ERR_NO_DEVICE = -5
} mvs_status;

/* Patch record: pmmvps/patch.hpp:33-66.  coord.w = 1, normal.w = 0.  128 bytes (192 with MVS_MAX_IMAGES 64: mvs_patch_bytes()). */
typedef struct mvs_patch {
    float coord[4];   /* Patch::m_coord */
    float normal[4];  /* Patch::m_normal */
    float ncc;        /* Patch::m_ncc; < 0 = not computed yet (patch.cpp:12) */
    float dscale;     /* Patch::m_dscale */
    float ascale;     /* Patch::m_ascale */
    float tmp;        /* Patch::m_tmp */
    int32_t nimages;  /* m_images.size() */
    int32_t nvimages; /* m_vimages.size() */
    int32_t flags;    /* bit 0: alive; bit 1: the engine's own (the list is settled: Filter::filterExact need not recompute the reference
                       * view while the patch keeps its views; cleared on upload).  In exported "new" records bits 8.. hold the swept view */
    int32_t id;       /* pool index (download) / destination cell (exported "new" records) */
    uint8_t images[MVS_MAX_IMAGES];  /* Patch::m_images, [0] is the reference view */
    uint8_t vimages[MVS_MAX_IMAGES]; /* Patch::m_vimages */
} mvs_patch;

/* Option (pmmvps/option.hpp:20-73) + the engine's own knobs. */
typedef struct mvs_config {
    int32_t nviews;          /* Option::m_nimages */
    int32_t level;           /* Option::m_level */
    int32_t csize;           /* Option::m_csize */
    int32_t wsize;           /* Option::m_wsize (<= 7: the window's samples are dealt over the lanes of one wavefront) */
    int32_t minImageNum;     /* Option::m_minImageNum; tau = min(2 * minImageNum, nviews) must not exceed 16 */
    int32_t max_propag;      /* Propagate::MAX_NUM_OF_PROPAG, propagate.cpp:24 */
    float nccThreshold;      /* Option::m_nccThreshold */
    float maxAngleThreshold; /* Option::m_maxAngleThreshold, radians */
    float quadThreshold;     /* Option::m_quadThreshold */
    int32_t depth;           /* PmMvps::m_depth when Propagate::run is entered */
    uint32_t seed;           /* counter-based RNG seed */
    int32_t refine_steps;    /* halving steps of the refiner, four proposals each; 1 + 4*steps cost evaluations (default 6: 25) */
    float refine_rd0;        /* initial depth range in units of Patch::m_dscale */
    float refine_ra0;        /* initial angle range in units of pi/48 (optim.cpp:487) */
    int32_t enable_check;    /* Optim::check when depth >= 2 (optim.cpp:292) */
    int32_t view_begin;      /* views view_begin, view_begin+view_stride, ... are swept by this engine */
    int32_t view_stride;
    int32_t device;          /* HIP device ordinal */
    int32_t view_propagation;/* 1 = also run the view-propagation branch the reference keeps commented out (propagate.cpp:110-120) */
    int32_t shard_index;     /* shard_count > 1: this engine sweeps the shard_index-th of shard_count equal, contiguous */
    int32_t shard_count;     /*   ranges of the (view, cell) sequence instead of whole views (view_begin/view_stride ignored) */
    int32_t literal_groups;  /* Filter::filterSmallGroups: 0 = groups are the connected components of the symmetrised neighbour relation
                              * (one pass of a union-find; default), 1 = the reference's breadth-first labelling in patch order over the
                              * directed relation (filter.cpp:432-524), exactly: two passes and a small search on the host.
                              * (Sits in what was padding before max_patches: sizeof(mvs_config) is unchanged.) */
    int64_t max_patches;     /* patch pool capacity (0 = 4 * total cells, at most a sixth of the device's memory per pool buffer) */
} mvs_config;

/* One view: PhotoSet::m_photos[i] (image/photoSet.hpp:62).  P is the row-major 3x4 level-0 projection
 * (`CONTOUR` camera text, image/camera.cpp:110-116); rgb is the level-0 image as Image::m_images[0]
 * holds it, interleaved uint8 RGB, H rows of W pixels (image/image.hpp:76); mask is H*W uint8 or NULL. */
typedef struct mvs_view_desc {
    int32_t width, height;
    float P[12];
    const uint8_t* rgb;
    const uint8_t* mask;
} mvs_view_desc;

/* Propagate's counters (pmmvps/propagate.hpp:63-69) plus work counts for the metric. */
typedef struct mvs_counters {
    int64_t candidates;  /* generatePatch returned a patch */
    int64_t prefiltered; /* cand.ncc < worst.ncc, propagate.cpp:170 */
    int64_t patches;     /* reached Optim::preProcess, propagate.cpp:182 -- the unit of "patches/s" */
    int64_t fail0;       /* m_fcount0 */
    int64_t fail1;       /* m_fcount1 */
    int64_t inserted;    /* pcount */
    int64_t replaced;    /* rcount */
    int64_t evals;       /* texture evaluations (one getPAxes + up to V getTex) */
    int64_t view_evals;  /* getTex calls that sampled: 588 algorithmic bytes each */
    int64_t trimmed;     /* removed by the MAX_NUM_OF_PATCHES trim */
} mvs_counters;

typedef struct mvs_engine mvs_engine;

const char* mvs_last_error(void);
int mvs_device_count(void);
/* Patch::m_images / m_vimages are unbounded in the reference (optim.cpp:165-205 pushes every qualifying view); the engine keeps
 * them in wavefront lanes, one view per lane.  Three builds of the same sources: libmvskit_engine.so holds 16 views per list,
 * libmvskit_engine_cap32.so 32, libmvskit_engine_cap64.so 64 (= the engine's view limit; 192-byte records).  Load the smallest one
 * whose mvs_list_cap() >= the data set's view count: then NO list is ever cut.  A smaller library on a larger data set keeps the
 * first mvs_list_cap() entries of a list (mvskit_amd.engine.Engine and the host mirror pick the library by view count). */
int mvs_list_cap(void);
int mvs_patch_bytes(void); /* sizeof(mvs_patch) in this build of the library: 128, or 192 in libmvskit_engine_cap64.so -- a caller checks it against its own */
void mvs_default_config(mvs_config* cfg); /* Option::Option, option.cpp:19-33 */

/* PmMvps::init (pmmvps.cpp:18-68): thresholds, tau = min(2*minImageNum, nviews), maxLevel = level+3.
 * LIMITS -- Option::init (pmmvps/option.cpp:53-116) takes any value; the engine returns MVS_ERR_ARG outside these:
 *     nviews            1 .. 64      one wavefront lane per view in the per-view stages (64-view lists need libmvskit_engine_cap64.so)
 *     wsize             1 .. 7       the wsize^2 samples of a window are dealt over 3 slots x 16 lanes + one extra sample (49)
 *     tau               <= 16        tau = min(2 * minImageNum, nviews) views of a proposal sit in 16 frame lanes: any minImageNum
 *                                    on <= 16 views, minImageNum <= 8 beyond
 *     max_propag*csize^2 <= 32       MAX_NUM_OF_PATCHES (propagate.cpp:24-25): a cell's live list is held in 32 lanes; max_propag <= 16
 *     level             0 .. 4       level + 3 pyramid levels, 7 at most
 *     images            >= 8 x 8 pixels at level 0 (mvs_engine_set_views)
 * and MVS_ERR_CAPACITY at run time beyond: max_patches records in the pool (default 4 per cell), 2^30 patches (ids of staged
 * records start at 0x40000000), and 14336 distinct patches / 4064 neighbours around one patch in Optim::check / filterNeighbor. */
int mvs_engine_create(const mvs_config* cfg, mvs_engine** out);
int mvs_engine_destroy(mvs_engine* e);

/* PhotoSet::init + Image::buildImagePyramid + Camera::updateCamera + Optim::setAxesScales +
 * PatchManager::init: uploads level 0, builds pyramids, cameras and grids on the device. */
int mvs_engine_set_views(mvs_engine* e, int nviews, const mvs_view_desc* views);
int mvs_engine_grid_dims(mvs_engine* e, int view, int* gw, int* gh); /* patch_manager.cpp:36-37 */
int mvs_engine_get_pyramid(mvs_engine* e, int view, int level, uint8_t* rgb_out, int* W, int* H);

/* thresholds: PmMvps::m_nccThreshold, m_nccThresholdBefore, m_depth */
int mvs_engine_set_thresholds(mvs_engine* e, float nccThreshold, float nccThresholdBefore, int depth);
int mvs_engine_get_thresholds(mvs_engine* e, float* nccThreshold, float* nccThresholdBefore, int* depth);
int mvs_engine_update_threshold(mvs_engine* e); /* PmMvps::updateThreshold + ++m_depth, pmmvps.cpp:70-74,105 */

/* PatchManager::readPatches tail (patch_manager.cpp:450-463): seeds -> pool */
int mvs_engine_upload_patches(mvs_engine* e, int64_t n, const mvs_patch* patches);
int mvs_engine_clear_patches(mvs_engine* e);
/* Optional: sizes the two cell indexes (PatchManager::m_pgrids / m_vpgrids as lists, patch_manager.hpp) for `list_entries` memberships
 * each up front -- 0 = MAX_NUM_OF_PATCHES per cell of every view -- so that the calls below allocate nothing while the lists stay
 * below that.  Without it the buffers grow inside the first iterations of a run.  A buffer that the call allocates is written once
 * (the first use of fresh device memory is slow); a request the buffers already hold changes nothing. */
int mvs_engine_reserve(mvs_engine* e, int64_t list_entries);
int mvs_engine_num_patches(mvs_engine* e, int64_t* n_alive);
int mvs_engine_download_patches(mvs_engine* e, int64_t cap, mvs_patch* out, int64_t* n); /* collectPatches */

/* Propagate::run(iter), propagate.cpp:28-64: two colour passes, each = index build + sweep + commit */
int mvs_engine_propagate(mvs_engine* e, int iter, mvs_counters* out);

/* Filter::run (pmmvps/filter.cpp:25-49): filterOutside, filterExact, filterNeighbor(1), filterSmallGroups with the
 * depth-map / m_vimages rebuilds in between; removed4 = patches removed by each of the four.  filterSmallGroups
 * groups by connected components of the symmetrised neighbour relation.
 * With a communicator attached (mvs_engine_comm_init / _attach) the call is collective: every rank runs the per-patch stages
 * on its share of the pool and the ranks exchange what the stages wrote; all ranks end with the same pool and the same counts. */
int mvs_engine_filter(mvs_engine* e, int64_t* removed4);
/* what the last mvs_engine_filter did: HIP-event time of each stage's kernel(s) and the work counts behind the
 * algorithmic-bytes model of DESIGN.md ("Filter::run"). */
typedef struct mvs_filter_stats {
    float outside_ms, exact_ms, neighbor_ms, groups_ms; /* the four filters */
    float rebuild_ms;                                   /* the five setDepthMapsVGridsVPGridsAddPatchV rebuilds together */
    float total_ms;
    int64_t patches_in;          /* alive patches when Filter::run started */
    int64_t exact_patches;       /* alive patches filterExact looked at */
    int64_t exact_view_evals;    /* getTex calls of its setRefImage (588 algorithmic bytes each) */
    int64_t neighbor_patches;    /* alive patches filterNeighbor looked at */
    int64_t neighbor_tasks;      /* (view, cell) lists opened by findNeighbors: 25 cells x m_images, two grids each */
    int64_t neighbor_entries;    /* list entries walked (4-byte ids) */
    int64_t neighbor_visited;    /* distinct patches met: one 48-byte geometry gather each */
    int64_t neighbor_accepted;   /* neighbours handed to filterQuad */
    int64_t neighbor_retried;    /* patches whose neighbourhood did not fit the first launch's id set (second launch, 16384 slots) */
    int64_t exchange_bytes;      /* multi-GPU: bytes this rank received from the others (kill bytes, rewritten records); the work counts
                                  * above then cover this rank's share of the pool only */
} mvs_filter_stats;
int mvs_engine_filter_stats(mvs_engine* e, mvs_filter_stats* out);

/* The same split for view-sharded runs: every rank holds the whole pool, sweeps its own views
 * (view_begin/view_stride) and exchanges what it created before every rank commits the union. */
int mvs_engine_pass(mvs_engine* e, int iter, int pass, mvs_counters* out); /* index build + sweep */
int mvs_engine_export_counts(mvs_engine* e, int64_t* n_new, int64_t* n_kill, int32_t* per_view_new /* [nviews] */);
/* device buffers owned by the caller (e.g. torch tensors): records ordered (view, cell, sequence) */
int mvs_engine_export_device(mvs_engine* e, void* d_new_records, int64_t cap_new, void* d_kill_ids, int64_t cap_kill);
int mvs_engine_commit_device(mvs_engine* e, const void* d_new_records, int64_t n_new, const void* d_kill_ids, int64_t n_kill);
int mvs_engine_commit_local(mvs_engine* e);

/* ---- multi-GPU through the C ABI (no reference counterpart: the reference is one thread on one CPU).
 * One engine per process and GPU; every engine is created with shard_index = rank, shard_count = world, holds the whole
 * pool and all pyramids, and sweeps its contiguous range of the (view, cell) job sequence.  The engines of a job share an
 * RCCL communicator; after each colour pass mvs_engine_exchange all-gathers, on the engine's own stream, (1) five int64 per
 * rank in ONE ncclAllGather -- {new records, evicted ids, this rank's status of the pass, pool headroom, kill-id capacity}: a rank
 * whose pass failed makes every rank give the pass up and return the same status from the same call -- (2) the new records
 * (sizeof(mvs_patch) bytes each), each rank's block broadcast straight into its
 * final place behind the pool (ranges are contiguous, so rank order IS the global (view, cell, creation) order), and
 * (3) the ids of evicted patches -- then commits the union, so all pools stay identical and equal to the 1-GPU result.
 * With a communicator attached, mvs_engine_propagate does pass + exchange itself: PmMvps::run needs no other change.
 * librccl is opened at run time (dlopen); without it these calls return MVS_ERR_STATE and everything else still works.
 * Environment: MVS_CCL_LIBRARY=<path> opens that library instead -- anything exporting ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclAllGather, ncclBroadcast, ncclGroupStart, ncclGroupEnd, ncclGetErrorString (a site's own RCCL build; the
 * shared-memory loopback of tests/loopback_ccl, with which several ranks can share the one GPU of a test box). */
#define MVS_COMM_ID_BYTES 128
int mvs_comm_unique_id(void* id_out /* MVS_COMM_ID_BYTES */); /* ncclGetUniqueId: rank 0 calls it and hands the bytes to the other ranks (file, socket, MPI ...) */
int mvs_engine_comm_init(mvs_engine* e, const void* id /* MVS_COMM_ID_BYTES */, int rank, int world); /* ncclCommInitRank on the engine's device; collective */
int mvs_engine_comm_attach(mvs_engine* e, void* nccl_comm /* ncclComm_t the host owns */, int rank, int world);
int mvs_engine_comm_release(mvs_engine* e); /* destroys the communicator comm_init made / forgets an attached one */
int mvs_engine_exchange(mvs_engine* e);     /* after mvs_engine_pass: all-gather + commit of the union; collective */
/* rank / world the engine was given (world 0: no communicator) and what the communicator itself reports (ncclCommCount,
 * ncclCommUserRank; -1 where the collective library does not export them) -- for a launcher that wants to see that N ranks really
 * share one communicator (bench.py --gpus N prints it as `rccl_world`) */
int mvs_engine_comm_info(mvs_engine* e, int* rank, int* world, int* comm_count, int* comm_rank);
/* FAILURES in a multi-rank job.  A failure on one rank (its pass overflowed, an allocation did not fit, a HIP error) reaches every
 * rank: mvs_engine_exchange's first all-gather and the agreement in front of every collective of mvs_engine_filter carry a status
 * word, all ranks give the call up together and return the same status; nobody waits.  Two things cannot be agreed on: no device
 * memory for the status word itself, and a failure of the collective library -- the rank returns MVS_ERR_HIP alone and the job
 * must be torn down by its launcher (torch.distributed.run and bench.py --gpus N both end the job when a rank exits non-zero).
 * After an error from mvs_engine_filter the stages that completed stand on every rank alike; the stage that was under way may
 * have rewritten lists of this rank's share only: re-upload the patches or stop. */

/* parity artefact: kind 0 = m_dpgrids patch, kind 1 = best-NCC patch of
 * m_pgrids[view][cell] whose reference view is `view`.  depth[gw*gh] = oaxis.coord, normal[gw*gh*3],
 * ids[gw*gh]; empty cells are NaN / -1.  Host buffers. */
int mvs_engine_depth_normal_map(mvs_engine* e, int view, int kind, float* depth, float* normal, int32_t* ids);

/* Batched single functions of the path, for parity tests and kernel benchmarks.
 * op: see mvs_probe_op.  in/out are host arrays of n records / values. */
typedef enum mvs_probe_op {
    MVS_PROBE_NCC = 0,        /* PatchManager::computeNcc      -> out_f[n] */
    MVS_PROBE_PREPROCESS = 1, /* Optim::preProcess             -> out_rec[n], out_i[n] = flag */
    MVS_PROBE_REFINE = 2,     /* Optim::refinePatch            -> out_rec[n]; key = (0,0,i,0) */
    MVS_PROBE_POSTPROCESS = 3,/* Optim::postProcess            -> out_rec[n], out_i[n] = flag */
    MVS_PROBE_COST = 4,       /* Optim::cost_func at encode(p) -> out_f[n] */
    MVS_PROBE_MATH = 5        /* in_f[n] -> out_f[5n]: sin, cos, asin, acos, atan of each input */
} mvs_probe_op;
int mvs_engine_probe(mvs_engine* e, int op, int64_t n, const mvs_patch* in_rec, const float* in_f,
                     mvs_patch* out_rec, float* out_f, int32_t* out_i);

/* timing of the last mvs_engine_pass / mvs_engine_propagate, measured with HIP events on the engine's stream */
typedef struct mvs_timing {
    float index_ms;  /* index build (CSR, trim, depth maps) */
    float sweep_ms;  /* the sweep kernel(s) */
    float commit_ms; /* commit */
    int32_t sweep_launches;
    float exchange_ms;      /* mvs_engine_exchange: count all-gather + record / kill-id broadcasts (HIP events) */
    int64_t exchange_bytes; /* bytes this rank received in them */
    int64_t check_retried_cells; /* destination cells whose Optim::check met more patches than the wave's LDS id set holds and that ran
                                  * again on the second tier (a 16384-slot set in global memory); normally 0 */
} mvs_timing;
int mvs_engine_last_timing(mvs_engine* e, mvs_timing* t);

#ifdef __cplusplus
}
#endif
#endif /* MVSKIT_ENGINE_H */
