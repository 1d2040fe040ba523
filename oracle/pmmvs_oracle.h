/*
 * pmmvs_oracle.h -- C API of the CPU ORACLE for the PatchMatch-MVS propagate+optim path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the reference
 * algorithm (imkaywu/MVSKit, pmmvps/{propagate,optim,patch_manager,filter}.cpp and
 * image/{camera,image,photoSet}.cpp).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.  The product
 * (mvskit_amd/, include/mvskit_engine.h) never includes, links or calls anything here.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, fixtures or tests for this path
 * and cannot be built in this image (Eigen, CImg and NLopt are un-vendored and absent), so
 * the restatement is pinned only by closed-form known-answer tests (tests/golden/) and by
 * the libstdc++ RNG draws recorded in SURVEY.md section 0.4.
 */
#ifndef PMMVS_ORACLE_H
#define PMMVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_WIDE_LISTS
#define ORC_MAX_IMAGES 64 /* the wide build (make wide): 64 views per list, records of 192 bytes -- what the 64-view engine build is checked against */
#else
#define ORC_MAX_IMAGES 32 /* capacity of Patch::m_images / m_vimages in the restatement */
#endif

/* Patch record, follows pmmvps/patch.hpp:33-66 (coord w=1, normal w=0). */
typedef struct orc_patch {
    float coord[4];
    float normal[4];
    float ncc;    /* m_ncc, <0 = not computed (patch.cpp:12) */
    float dscale; /* m_dscale */
    float ascale; /* m_ascale */
    float tmp;    /* m_tmp (score2 / gain) */
    int32_t nimages;  /* m_images.size() */
    int32_t nvimages; /* m_vimages.size() */
    int32_t flags;    /* bit0 = alive */
    int32_t id;       /* pool index */
    uint8_t images[ORC_MAX_IMAGES];  /* m_images, [0] = reference view */
    uint8_t vimages[ORC_MAX_IMAGES]; /* m_vimages */
} orc_patch; /* 128 bytes (192 in the wide build: orc_patch_bytes()) */

enum { ORC_SCHEDULE_FAITHFUL = 0, ORC_SCHEDULE_ENGINE = 1 };
enum { ORC_SUM_SEQ = 0, ORC_SUM_TREE64 = 1 };

/* Option (pmmvps/option.hpp:20-73) + PmMvps thresholds (pmmvps.cpp:54-67) + engine knobs. */
typedef struct orc_config {
    int32_t nviews;
    int32_t level;        /* Option::m_level */
    int32_t csize;        /* Option::m_csize */
    int32_t wsize;        /* Option::m_wsize (<= 8) */
    int32_t minImageNum;  /* Option::m_minImageNum */
    int32_t max_propag;   /* Propagate::MAX_NUM_OF_PROPAG (propagate.cpp:24) */
    float nccThreshold;   /* Option::m_nccThreshold */
    float maxAngleThreshold; /* radians, Option::m_maxAngleThreshold */
    float quadThreshold;  /* Option::m_quadThreshold */
    int32_t depth;        /* PmMvps::m_depth at the time of Propagate::run */
    /* engine knobs (no reference counterpart) */
    uint32_t seed;
    int32_t schedule;     /* ORC_SCHEDULE_* */
    int32_t sum_mode;     /* ORC_SUM_* */
    int32_t refine_steps; /* K halving steps, 4 proposals each: R = 1 + 4K evaluations (default 6: 25) */
    float refine_rd0;     /* initial depth range, units of m_dscale */
    float refine_ra0;     /* initial angle range, units of pi/48 */
    int32_t enable_check; /* run Optim::check when depth >= 2 */
    int32_t view_begin;   /* engine schedule: views [view_begin, view_end) with stride are swept */
    int32_t view_stride;
    int32_t nthreads;     /* engine schedule only: OpenMP threads over destination cells */
    int32_t view_propagation; /* engine schedule only: the disabled branch of propagate.cpp:110-120 (see dest_cell_engine) */
    int32_t shard_index;  /* engine schedule, shard_count > 1: sweep the shard_index-th of shard_count contiguous ranges */
    int32_t shard_count;  /*   of the destination cells of all views (view_begin / view_stride ignored) */
    int32_t list_cap;     /* m_images / m_vimages are truncated to this many views; 0 = 16, the engine's limit (MVS_LIST_CAP).
                           * Values above 32 need the wide build (make -C oracle wide), see orc_list_storage() */
    int32_t literal_evals; /* engine schedule only, 1 = no evaluation shortcuts: the initial m_ncc of every candidate
                            * (propagate.cpp:235), refinePatch's own final computeINCC (optim.cpp:541) and the second
                            * constraintImages of postProcess (optim.cpp:286) are always evaluated, as the reference does.
                            * The patches must come out identical; only the work counters (evals, view_evals) differ. */
    int32_t literal_groups; /* engine schedule only, 1 = Filter::filterSmallGroups labels as the FAITHFUL schedule does (breadth-first
                             * in patch order, filter.cpp:432-524) instead of by connected components: mvs_config.literal_groups */
} orc_config;

typedef struct orc_counters {
    int64_t candidates;  /* generatePatch calls that returned a patch */
    int64_t prefiltered; /* rejected by cand.ncc < worst.ncc (propagate.cpp:170) */
    int64_t patches;     /* entered Optim::preProcess (propagate.cpp:182): the metric's unit */
    int64_t fail0;       /* m_fcount0 */
    int64_t fail1;       /* m_fcount1 */
    int64_t inserted;    /* pcount */
    int64_t replaced;    /* rcount */
    int64_t evals;       /* texture evaluations (one getPAxes + <=V getTex) */
    int64_t view_evals;  /* getTex calls that sampled (flag == 0) */
    int64_t trimmed;     /* patches removed by the MAX_NUM_OF_PATCHES trim */
} orc_counters;

typedef struct orc_scene orc_scene;

void orc_default_config(orc_config* cfg);
orc_scene* orc_create(const orc_config* cfg);
void orc_destroy(orc_scene* s);
const char* orc_last_error(void);

/* P = row-major 3x4 level-0 projection (CONTOUR, camera.cpp:110-116); rgb = HxWx3 uint8
 * (image.hpp:76); mask = HxW uint8 or NULL. */
int orc_set_view(orc_scene* s, int v, int W, int H, const float* P, const uint8_t* rgb, const uint8_t* mask);
int orc_finalize_views(orc_scene* s);
int orc_get_pyramid(orc_scene* s, int v, int level, uint8_t* rgb_out, int* W, int* H);
int orc_get_camera(orc_scene* s, int v, float* center4, float* oaxis4, float* xaxis3, float* yaxis3, float* zaxis3, float* ipscale);
int orc_grid_dims(orc_scene* s, int v, int* gw, int* gh);

int orc_set_thresholds(orc_scene* s, float nccThreshold, float nccThresholdBefore, int depth);
int orc_get_thresholds(orc_scene* s, float* nccThreshold, float* nccThresholdBefore, int* depth);
int orc_update_threshold(orc_scene* s); /* PmMvps::updateThreshold + ++m_depth (pmmvps.cpp:70-74,105) */

int orc_add_patches(orc_scene* s, int n, const orc_patch* p); /* readPatches tail: setGrids + addPatch */
int orc_num_patches(orc_scene* s);                            /* alive patches */
int orc_get_patches(orc_scene* s, int cap, orc_patch* out);    /* alive patches in id order */
int orc_clear_patches(orc_scene* s);

int orc_propagate(orc_scene* s, int iter, orc_counters* out);  /* Propagate::run(iter) */
/* Filter::run (filter.cpp:25-49): removed4 = patches removed by filterOutside / Exact / Neighbor / SmallGroups */
int orc_filter(orc_scene* s, int64_t* removed4);
/* faithful schedule only: bound the work (for timing a sample). <=0 means unlimited. */
int64_t orc_list_truncations(orc_scene* s); /* times a view list wanted to grow past list_cap since orc_create */
int orc_list_storage(void);                 /* views a list can hold in this build: 32, or 64 in the wide build */
int orc_patch_bytes(void);                  /* sizeof(orc_patch) in this build */
/* orc_filter with its last stage, Filter::filterSmallGroups (filter.cpp:432-578), looked at twice: how the connected components the
 * ENGINE schedule labels by compare with the reference's breadth-first labelling on the pool that stage meets (out[4]: alive,
 * removed by components, by the literal labelling, by both); and the relation itself on the current pool (out[3]: alive,
 * directed edges, edges without their reverse) */
int orc_small_groups_compare(orc_scene* s, int64_t* out4);
int orc_group_edge_stats(orc_scene* s, int64_t* out3);
int orc_set_cell_budget(orc_scene* s, int64_t max_source_cells);
int orc_set_time_budget(orc_scene* s, double seconds);
double orc_last_sweep_seconds(orc_scene* h); /* engine schedule: wall time of the last colour pass's parallel loop */

/* view-sharded exchange (engine schedule): records created by the last pass / ids killed */
int orc_engine_pass(orc_scene* s, int iter, int pass, orc_counters* out); /* sweep without commit */
int orc_export_new(orc_scene* s, int cap, orc_patch* out, int32_t* per_view_counts);
int orc_export_kills(orc_scene* s, int cap, int32_t* ids);
int orc_commit(orc_scene* s, int n_new, const orc_patch* recs, int n_kill, const int32_t* kill_ids);

/* kind 0: m_dpgrids patch; kind 1: best-NCC patch of m_pgrids[view][cell] with m_images[0]==view.
 * depth[gw*gh], normal[gw*gh*3], ids[gw*gh]; empty = NaN / -1. */
int orc_depth_normal_map(orc_scene* s, int view, int kind, float* depth, float* normal, int32_t* ids);

/* ---- probes of single reference functions, used by the known-answer tests ---- */
int orc_project(orc_scene* s, int v, const float* coord4, int level, float* icoord3);          /* camera.cpp:310-326 */
int orc_unproject(orc_scene* s, int v, const float* icoord3, int level, float* coord4);        /* camera.cpp:329-337 */
float orc_get_unit(orc_scene* s, int v, const float* coord4);                                   /* optim.cpp:34-41 */
int orc_get_paxes(orc_scene* s, int v, const float* coord4, const float* normal4, float* px4, float* py4); /* optim.cpp:67-84 */
int orc_get_color(orc_scene* s, int v, float x, float y, int level, float* rgb3);               /* image.cpp:447-472 */
/* returns flag (0 ok, -1 rejected); tex = wsize*wsize*3 floats, normalized if normalize != 0 */
int orc_get_tex(orc_scene* s, const float* coord4, const float* px4, const float* py4, const float* normal4,
                int v, float* tex, int normalize);                                               /* optim.cpp:790-844,917-940 */
float orc_compute_incc(orc_scene* s, const orc_patch* p, int robust);                           /* optim.cpp:630-706 */
float orc_compute_ncc(orc_scene* s, const orc_patch* p);                                        /* patch_manager.cpp:401-404 */
int orc_set_inccs(orc_scene* s, const orc_patch* p, int robust, float* inccs);                   /* optim.cpp:708-746 */
int orc_set_inccs_matrix(orc_scene* s, const orc_patch* p, int robust, float* inccs);            /* optim.cpp:748-783 */
int orc_preprocess(orc_scene* s, orc_patch* p);                                                  /* optim.cpp:137-163 */
int orc_refine(orc_scene* s, orc_patch* p, const uint32_t* key4);                                /* optim.cpp:480-547 */
int orc_postprocess(orc_scene* s, orc_patch* p);                                                 /* optim.cpp:260-298 */
double orc_cost(orc_scene* s, const orc_patch* p, const float* x3);                              /* optim.cpp:401-468 */
int orc_encode(orc_scene* s, const orc_patch* p, float* x3);                                     /* optim.cpp:549-580 */
int orc_decode(orc_scene* s, const orc_patch* p, const float* x3, float* coord4, float* normal4);/* optim.cpp:582-599 */
int orc_generate_patch(orc_scene* s, const orc_patch* src, const float* icoord3, orc_patch* out);/* propagate.cpp:220-237 */
float orc_quad_residual(orc_scene* s, const orc_patch* p, const float* coords4, int n);            /* filter.cpp:329-392: the residual filterQuad compares with m_quadThreshold */
float orc_robustincc(float incc);
float orc_unrobustincc(float rincc);
void orc_minstd_draws(int n, float* out);  /* propagate.cpp:139-140 on this libstdc++ */
float orc_rng_uniform(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e); /* engine counter RNG */
float orc_sinf(float x);
float orc_cosf(float x);
float orc_asinf(float x);
float orc_acosf(float x);
float orc_atanf(float x);
int orc_is_neighbor(orc_scene* s, const orc_patch* a, const orc_patch* b, float thr);            /* pmmvps.cpp:117-147 */
float orc_compute_gain(orc_scene* s, const orc_patch* p);                                        /* filter.cpp:108-146 */
int orc_check(orc_scene* s, orc_patch* p);                                                        /* optim.cpp:300-323 */

#ifdef __cplusplus
}
#endif
#endif
