// mvs_kernels.h -- launchers of the HIP kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "mvs_types.h"

void mvsk_rgb_to_rgba(const uint8_t* rgb, uint32_t* out, int64_t n, hipStream_t st);
void mvsk_rgba_to_rgb(const uint32_t* in, uint8_t* rgb, int64_t n, hipStream_t st);
void mvsk_pyr_down(const uint32_t* src, int pw, int ph, uint32_t* dst, int w, int h, hipStream_t st);
void mvsk_mask_down(const uint8_t* src, int pw, int ph, uint8_t* dst, int w, int h, hipStream_t st);
void mvsk_mask_binarise(uint8_t* m, int64_t n, hipStream_t st);
void mvsk_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* tmp, hipStream_t st);
void mvsk_index_count(const DParams& prm, int32_t* cnt, int32_t* vcnt, unsigned long long* total, hipStream_t st);
void mvsk_exclusive_scan_off(const int32_t* in, csr_off_t* out, int64_t n, csr_off_t* tmp, hipStream_t st);
void mvsk_index_fill(const DParams& prm, int vgrid, const csr_off_t* start, int32_t* cursor, unsigned long long* ids, hipStream_t st);
void mvsk_index_fill_direct(const DParams& prm, int vgrid, const csr_off_t* start, int32_t* cursor, int32_t* id32, hipStream_t st);
void mvsk_index_sort_trim(const DParams& prm, const csr_off_t* start, unsigned long long* ids, int do_trim, unsigned long long* trimmed, hipStream_t st);
void mvsk_index_finalize(const DParams& prm, int vgrid, const csr_off_t* start, unsigned long long* ids, int32_t* id32, int32_t* cnt_alive, hipStream_t st);
void mvsk_index_pack(const DParams& prm, const csr_off_t* start, const csr_off_t* start2, const int32_t* cnt_alive, const unsigned long long* ids, const int32_t* id32_in,
                     ListKey* key, int32_t* id32_out, hipStream_t st);
void mvsk_depth_maps(const DParams& prm, unsigned long long* dp, const uint32_t* dirty, hipStream_t st);
void mvsk_depth_mark_dirty(const DParams& prm, const uint8_t* kill, unsigned long long* dp, uint32_t* dirty, hipStream_t st);
void mvsk_best_ncc_map(const DParams& prm, int view, unsigned long long* best, hipStream_t st);
void mvsk_map_extract(const DParams& prm, int view, int kind, const unsigned long long* sel, float* depth, float* normal, int32_t* ids, int ncells, hipStream_t st);
void mvsk_fill_ncc(const DParams& prm, unsigned long long* evals, hipStream_t st);
size_t mvsk_sweep_lds_bytes(const DParams& prm);
void mvsk_sweep(const DParams& prm, const SweepArgs& a, hipStream_t st);
void mvsk_sweep_retry(const DParams& prm, const SweepArgs& a, int nretry, hipStream_t st);
void mvsk_job_work(const DParams& prm, const SweepArgs& a, int mode, int shift, int32_t* work, hipStream_t st);
void mvsk_job_cuts(const int32_t* scan, int64_t njobs, int n, int32_t* cuts, hipStream_t st);
void mvsk_commit_count(const SweepArgs& a, int32_t* cnt, hipStream_t st);
void mvsk_commit_copy(const SweepArgs& a, const int32_t* base, DPatch* dst, int64_t dst_cap, int32_t* per_view, int keep_key, hipStream_t st);
void mvsk_kill_count(const uint8_t* kill, int64_t n, int32_t* cnt, hipStream_t st);
void mvsk_kill_export(const uint8_t* kill, int64_t n, const int32_t* base, int32_t* ids, int64_t cap, hipStream_t st);
void mvsk_apply_kill_flags(DPatch* pool, uint8_t* kill, int64_t n, hipStream_t st);
void mvsk_apply_kill_ids(DPatch* pool, const int32_t* ids, int64_t n, int64_t pool_n, hipStream_t st);
void mvsk_append_records(DPatch* pool, int64_t pool_n, const DPatch* recs, int64_t n, hipStream_t st);
void mvsk_alive_count(const DPatch* pool, int64_t n, int32_t* cnt, hipStream_t st);
void mvsk_alive_gather(const DPatch* pool, int64_t n, const int32_t* base, DPatch* out, int64_t cap, hipStream_t st);
void mvsk_filter_vimages(const DParams& prm, int additive, int64_t first, int64_t last, const uint32_t* dirty, hipStream_t st);
void mvsk_geo_pack(const DPatch* pool, int64_t n, float4* geo, uint8_t* ref, int* bad, hipStream_t st);
void mvsk_filter_outside(const DParams& prm, uint8_t* kill, int64_t first, int64_t last, hipStream_t st);
void mvsk_filter_exact(const DParams& prm, uint8_t* kill, unsigned long long* evals, unsigned long long* stage, int64_t first, int64_t last, hipStream_t st);
void mvsk_filter_neighbor(const DParams& prm, uint8_t* kill, int32_t* retry, int32_t* nretry, int32_t* overflow, unsigned long long* stats, int64_t first, int64_t last, hipStream_t st);
void mvsk_filter_neighbor_retry(const DParams& prm, uint8_t* kill, const int32_t* todo, int32_t ntodo, int32_t* overflow, unsigned long long* stats, hipStream_t st);
void mvsk_groups(const DParams& prm, int* parent, int* size, int threshold, uint8_t* kill, hipStream_t st);
void mvsk_groups_literal_edges(const DParams& prm, int* parent, int* size, int* edges2, int* nedges, int cap, hipStream_t st);
void mvsk_gather_i32(const int32_t* src, const int32_t* idx, int32_t* out, int64_t n, hipStream_t st);
void mvsk_scatter_i32(int32_t* dst, const int32_t* idx, const int32_t* val, int64_t n, hipStream_t st);
void mvsk_groups_kill(const DParams& prm, const int* parent, const int* size, int threshold, uint8_t* kill, hipStream_t st);
void mvsk_probe(const DParams& prm, int op, int64_t n, const DPatch* in, const float* in_f, DPatch* out, float* out_f, int32_t* out_i, hipStream_t st);
