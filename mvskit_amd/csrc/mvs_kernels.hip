// mvs_kernels.hip -- HIP kernels of the MI355X PatchMatch-MVS engine (gfx950, wave64).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "mvs_device.cuh"
#include "mvs_check.cuh"
#include "mvs_kernels.h"

using namespace mvsdev;

// =================================================================== K0: images
// interleaved RGB (image/image.hpp:76) -> RGBA8 texels, one 32-bit load per texel
__global__ void k_rgb_to_rgba(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = rgb + 3 * i;
    out[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
__global__ void k_rgba_to_rgb(const uint32_t* __restrict__ in, uint8_t* __restrict__ rgb, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = in[i];
    rgb[3 * i] = t & 255u; rgb[3 * i + 1] = (t >> 8) & 255u; rgb[3 * i + 2] = (t >> 16) & 255u;
}
// Image::buildImagePyramid, image.cpp:245-315 (filter 0): 4x4 [1 3 3 1]x[1 3 3 1]/64, stride 2, taps outside
// the image dropped without renormalising (D8), round half up.
__global__ void k_pyr_down(const uint32_t* __restrict__ src, int pw, int ph, uint32_t* __restrict__ dst, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float base[4] = {1.0f, 3.0f, 3.0f, 1.0f};
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int i = -1; i < 3; ++i) {
        const int yt = 2 * y + i;
        if (yt < 0 || ph - 1 < yt) continue;
        for (int j = -1; j < 3; ++j) {
            const int xt = 2 * x + j;
            if (xt < 0 || pw - 1 < xt) continue;
            const float m = (base[i + 1] * base[j + 1]) / 64.0f;
            const uint32_t t = src[(size_t)yt * pw + xt];
            c0 += m * (float)(t & 255u); c1 += m * (float)((t >> 8) & 255u); c2 += m * (float)((t >> 16) & 255u);
        }
    }
    const uint32_t r = (uint8_t)((int)floorf(c0 + 0.5f)), g = (uint8_t)((int)floorf(c1 + 0.5f)), b = (uint8_t)((int)floorf(c2 + 0.5f));
    dst[(size_t)y * w + x] = r | (g << 8) | (b << 16);
}
// Image::buildMaskPyramid, image.cpp:717-747 (indices clamped to the previous level)
__global__ void k_mask_down(const uint8_t* __restrict__ src, int pw, int ph, uint8_t* __restrict__ dst, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const int ys[2] = {2 * y, min(ph - 1, 2 * y + 1)}, xs[2] = {2 * x, min(pw - 1, 2 * x + 1)};
    int inside = 0;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) inside += src[(size_t)ys[j] * pw + xs[i]] ? 1 : 0;
    dst[(size_t)y * w + x] = inside > 0 ? 255 : 0;
}
__global__ void k_mask_binarise(uint8_t* m, int64_t n) {  // image.cpp:170-177
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m[i] = m[i] > 127 ? 255 : 0;
}

// =================================================================== scan (exclusive; int32 counts in, int32 or int64 offsets out)
#define SCAN_BLOCK 256
#define SCAN_ITEMS 4
template <typename TI, typename TO>
__global__ void k_scan_block(const TI* __restrict__ in, TO* __restrict__ out, TO* __restrict__ block_sums, int64_t n_in) {
    __shared__ TO s[SCAN_BLOCK];
    const int64_t base = ((int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    TO v[SCAN_ITEMS], sum = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n_in) ? (TO)in[base + k] : (TO)0; sum += v[k]; }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        TO t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : (TO)0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    TO excl = s[threadIdx.x] - sum;
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k <= n_in) out[base + k] = excl; excl += v[k]; }
    if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = s[threadIdx.x];
}
template <typename TO>
__global__ void k_scan_add(TO* __restrict__ out, const TO* __restrict__ block_offsets, int64_t n_out) {
    const int64_t base = ((int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    const TO o = block_offsets[blockIdx.x];
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n_out) out[base + k] += o;
}
// out[i] = sum of in[0..i) for i in [0, n]; out[n] is the total (out has n+1 slots, in has n; in-place allowed when the types agree).
// tmp: block sums of every recursion level, at least n/512 + 64 elements of TO.
template <typename TI, typename TO>
void launch_exclusive_scan(const TI* in, TO* out, int64_t n, TO* tmp, hipStream_t st) {
    const int64_t per = (int64_t)SCAN_BLOCK * SCAN_ITEMS;
    const int64_t nb = (n + 1 + per - 1) / per;
    hipLaunchKernelGGL((k_scan_block<TI, TO>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, tmp, n);
    if (nb > 1) {
        TO* tmp2 = tmp + nb + 1;
        launch_exclusive_scan<TO, TO>(tmp, tmp, nb, tmp2, st);
        hipLaunchKernelGGL((k_scan_add<TO>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, out, tmp, n + 1);
    }
}

DEV uint32_t sortable_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// list entry of the index build: ascending key order = descending m_ncc, then ascending id (PatchManager::sortPatches)
__device__ __forceinline__ unsigned long long list_key(float ncc, int64_t id) {
    return ((unsigned long long)(~sortable_f32(ncc + 0.0f)) << 32) | (unsigned long long)(uint32_t)id;  // + 0.0f: -0 sorts as +0
}
// m_pgrids entry `pos` (an offset into the index): the fields the sweep reads of its own and its source cells
DEV int pgrid_id(const DParams& prm, csr_off_t pos) {
    return prm.csr_id32[pos];
}
DEV float pgrid_ncc(const DParams& prm, csr_off_t pos) {
    return prm.csr_key[pos].ncc;
}
DEV int pgrid_ref(const DParams& prm, csr_off_t pos) {
    return prm.csr_key[pos].ref;
}
// =================================================================== index build
// cnt[gcell] += 1 for every (patch, view) membership: PatchManager::addPatch, patch_manager.cpp:158-186
__global__ void k_index_count(DParams prm, int32_t* __restrict__ cnt, int32_t* __restrict__ vcnt, unsigned long long* __restrict__ total) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int mine = 0;  // list entries this patch adds (their sum sizes the id / key buffers)
    if (id < prm.pool_n && (prm.pool[id].flags & 1)) {
        const DPatch* p = prm.pool + id;
        const F4 coord = ld4(p->coord);
        const int n = cnt ? min(p->nimages, MVS_LISTCAP) : 0;
        for (int i = 0; i < n; ++i) {
            const DView* vw = prm.views + p->images[i];
            int ix, iy;
            cell_of(prm, vw, coord, ix, iy);
            if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) continue;
            atomicAdd(&cnt[vw->cell_base + iy * vw->gw + ix], 1);
            ++mine;
        }
        if (vcnt) {
            const int nv = min(p->nvimages, MVS_LISTCAP);
            for (int i = 0; i < nv; ++i) {
                const DView* vw = prm.views + p->vimages[i];
                int ix, iy;
                cell_of(prm, vw, coord, ix, iy);
                if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) continue;
                atomicAdd(&vcnt[vw->cell_base + iy * vw->gw + ix], 1);
                ++mine;
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if (total && (threadIdx.x & 63) == 0 && mine) atomicAdd(total, (unsigned long long)mine);  // (the index build reads the sum off its scan instead)
}
// one grid per launch (vgrid = 0: m_pgrids from m_images, 1: m_vpgrids from m_vimages): the sort key travels with the id, the per-cell
// sort reads no patch
__global__ void k_index_fill(DParams prm, int vgrid, const csr_off_t* __restrict__ start, int32_t* __restrict__ cursor, unsigned long long* __restrict__ ids) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    const F4 coord = ld4(p->coord);
    const unsigned long long key = list_key(p->ncc, id);
    const int n = vgrid ? min(p->nvimages, MVS_LISTCAP) : min(p->nimages, MVS_LISTCAP);
    for (int i = 0; i < n; ++i) {
        const DView* vw = prm.views + (vgrid ? p->vimages[i] : p->images[i]);
        int ix, iy;
        cell_of(prm, vw, coord, ix, iy);
        if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) continue;
        const int g = vw->cell_base + iy * vw->gw + ix;
        ids[start[g] + atomicAdd(&cursor[g], 1)] = key;
    }
}
// The index between the stages of Filter::run: no trim, and no stage depends on the order inside a list (computeGain takes maxima,
// findNeighbors builds a set whose layout is the same for every insertion order, filterSmallGroups joins sets), so the thread that
// holds the record writes the finished entry straight into its slot: no keys, no per-cell sort, no gather of the records.
// (The entry is the id; no reader inside Filter::run looks at csr_key.)
__global__ void k_index_fill_direct(DParams prm, int vgrid, const csr_off_t* __restrict__ start, int32_t* __restrict__ cursor, int32_t* __restrict__ id32) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    const F4 coord = ld4(p->coord);
    const int n = vgrid ? min(p->nvimages, MVS_LISTCAP) : min(p->nimages, MVS_LISTCAP);
    for (int i = 0; i < n; ++i) {
        const DView* vw = prm.views + (vgrid ? p->vimages[i] : p->images[i]);
        int ix, iy;
        cell_of(prm, vw, coord, ix, iy);
        if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) continue;
        const int g = vw->cell_base + iy * vw->gw + ix;
        const csr_off_t slot = start[g] + atomicAdd(&cursor[g], 1);
        id32[slot] = (int32_t)id;
    }
}
// PatchManager::sortPatches (descending NCC; ties by id) per cell, then the MAX_NUM_OF_PATCHES trim
// (propagate.cpp:94-99,130-135): every cell decides on the same snapshot; a trimmed patch dies everywhere.
__global__ void k_index_sort_trim(DParams prm, const csr_off_t* __restrict__ start, unsigned long long* __restrict__ ids, int do_trim,
                                  unsigned long long* __restrict__ trimmed) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = g < prm.total_cells;
    const csr_off_t b = in ? start[g] : 0, e = in ? start[g + 1] : 0;
    for (csr_off_t i = b + 1; i < e; ++i) {  // insertion sort, ascending keys
        const unsigned long long k = ids[i];
        csr_off_t j = i - 1;
        while (j >= b) {
            const unsigned long long o = ids[j];
            if (o < k) break;
            ids[j + 1] = o;
            --j;
        }
        ids[j + 1] = k;
    }
    if (do_trim) {
        int mine = 0;
        for (csr_off_t k = b + prm.cap; k < e; ++k) {
            const int old = atomicAnd(&prm.pool[(uint32_t)ids[k]].flags, ~1);
            mine += old & 1;
        }
        // one counter update per wave, not per trimmed patch, and into one of 256 partial sums (waves that end on an atomic to ONE
        // address leave in single file: the host adds the partial sums)
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(trimmed + (blockIdx.x & 255u), (unsigned long long)mine);
    }
}
// After the trim: every list is compacted to its alive entries (order kept) at the FRONT of its own range -- the id to id32, the
// (m_ncc, reference view) key of an m_pgrids entry over the sort key it replaces (a cell's thread reads position k >= b + n before it
// writes position b + n) -- and counted.  k_index_pack then closes the gaps the trim left between the lists.
__global__ void k_index_finalize(DParams prm, int vgrid, const csr_off_t* __restrict__ start, unsigned long long* __restrict__ ids,
                                 int32_t* __restrict__ id32, int32_t* __restrict__ cnt_alive) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= prm.total_cells) return;
    const csr_off_t b = start[g], e = start[g + 1];
    int n = 0;
    for (csr_off_t k = b; k < e; ++k) {
        const int id = (int)(uint32_t)ids[k];
        const DPatch* p = prm.pool + id;
        if (!(p->flags & 1)) continue;
        if (!vgrid) ids[b + n] = ((unsigned long long)(uint32_t)p->images[0] << 32) | (unsigned long long)__float_as_uint(p->ncc);  // m_vpgrids' readers want ids only
        id32[b + n] = id;
        ++n;
    }
    cnt_alive[g] = n;
}
// The lists of neighbouring cells end to end (start2 = the scan of the alive counts): Optim::check's findNeighbors reads the cells of a
// grid row as ONE run of ids, which the gaps the trim left would break.
__global__ void k_index_pack(DParams prm, const csr_off_t* __restrict__ start, const csr_off_t* __restrict__ start2, const int32_t* __restrict__ cnt_alive,
                             const unsigned long long* __restrict__ ids, const int32_t* __restrict__ id32_in, ListKey* __restrict__ key, int32_t* __restrict__ id32_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= prm.total_cells) return;
    const csr_off_t b = start[g], b2 = start2[g];
    const int n = cnt_alive[g];
    for (int k = 0; k < n; ++k) {
        id32_out[b2 + k] = id32_in[b + k];
        if (key) {
            const unsigned long long w = ids[b + k];
            ListKey lk;
            lk.ncc = __uint_as_float((uint32_t)w); lk.ref = (int32_t)(w >> 32);
            key[b2 + k] = lk;
        }
    }
}
// PatchManager::updateDepthMaps, patch_manager.cpp:191-221, over the alive pool (Filter::setDepthMaps, filter.cpp:580-626)
// A lane per patch, a wave per view (the four waves of a block share 64 consecutive patches and take the views in turn): patches that
// follow each other in the pool lie next to each other on the surface, so the 64 cells a wave touches in one view's map share
// cache lines -- with a lane per (patch, view) pair every lane of a wave wrote into another view's map.
// `dirty` (one bit per cell, or null): only the cells whose nearest patch a filter stage has just removed are recomputed
// (k_depth_mark_dirty emptied them); every other cell's nearest patch is still alive, so its entry stands.
__global__ __launch_bounds__(256) void k_depth_maps(DParams prm, unsigned long long* __restrict__ dp, const uint32_t* __restrict__ dirty) {
    const int64_t id = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63u);
    if (id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    const F4 coord = ld4(p->coord);
    for (int image = (int)(threadIdx.x >> 6); image < prm.nviews; image += 4) {
        const DView* vw = prm.views + image;
        const F3 ic = project(vw, coord, prm.level);
        const float fx = ic.x / (float)prm.csize, fy = ic.y / (float)prm.csize;
        const int xs[2] = {(int)floorf(fx), (int)ceilf(fx)}, ys[2] = {(int)floorf(fy), (int)ceilf(fy)};
        const float depth = dot4(ld4(vw->oaxis), coord);
        const unsigned long long key = ((unsigned long long)sortable_f32(depth) << 32) | (unsigned long long)(uint32_t)id;
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) {
            if (xs[i] < 0 || vw->gw <= xs[i] || ys[j] < 0 || vw->gh <= ys[j]) continue;
            if (i == 1 && xs[1] == xs[0]) continue;  // same cell twice: idempotent, skip
            if (j == 1 && ys[1] == ys[0]) continue;
            // the cell's value only ever decreases: a plain read that already shows a nearer patch saves the atomic (most do)
            const int cell = vw->cell_base + ys[j] * vw->gw + xs[i];
            if (dirty && !((dirty[cell >> 5] >> (cell & 31)) & 1u)) continue;
            unsigned long long* cellp = &dp[cell];
            if (key < __builtin_nontemporal_load(cellp)) atomicMin(cellp, key);
        }
    }
}
// Before a filter stage's removals are applied: the cells of the depth maps that name a patch about to go are emptied and marked.
// (A cell names one patch, so only that patch's thread writes it.)
__global__ __launch_bounds__(256) void k_depth_mark_dirty(DParams prm, const uint8_t* __restrict__ kill, unsigned long long* __restrict__ dp, uint32_t* __restrict__ dirty) {
    const int64_t id = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63u);
    if (id >= prm.pool_n || !kill[id]) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    const F4 coord = ld4(p->coord);
    for (int image = (int)(threadIdx.x >> 6); image < prm.nviews; image += 4) {
        const DView* vw = prm.views + image;
        const F3 ic = project(vw, coord, prm.level);
        const float fx = ic.x / (float)prm.csize, fy = ic.y / (float)prm.csize;
        const int xs[2] = {(int)floorf(fx), (int)ceilf(fx)}, ys[2] = {(int)floorf(fy), (int)ceilf(fy)};
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) {
            if (xs[i] < 0 || vw->gw <= xs[i] || ys[j] < 0 || vw->gh <= ys[j]) continue;
            if (i == 1 && xs[1] == xs[0]) continue;
            if (j == 1 && ys[1] == ys[0]) continue;
            const int cell = vw->cell_base + ys[j] * vw->gw + xs[i];
            if ((uint32_t)(dp[cell] & 0xffffffffull) != (uint32_t)id) continue;
            dp[cell] = ~0ull;
            atomicOr(&dirty[cell >> 5], 1u << (cell & 31));
        }
    }
}
// best-NCC patch per cell among those whose reference view is `view` (parity artefact)
__global__ void k_best_ncc_map(DParams prm, int view, unsigned long long* __restrict__ best) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1) || p->nimages == 0 || p->images[0] != view) return;
    const DView* vw = prm.views + view;
    int ix, iy;
    cell_of(prm, vw, ld4(p->coord), ix, iy);
    if (ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy) return;
    const unsigned long long key = ((unsigned long long)sortable_f32(p->ncc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)id);
    atomicMax(&best[iy * vw->gw + ix], key);
}
__global__ void k_map_extract(DParams prm, int view, int kind, const unsigned long long* __restrict__ sel, float* __restrict__ depth,
                              float* __restrict__ normal, int32_t* __restrict__ ids, int ncells) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const unsigned long long k = sel[c];
    const bool empty = kind == 0 ? (k == ~0ull) : (k == 0ull);
    if (empty) {
        depth[c] = __int_as_float(0x7fc00000);
        normal[3 * c] = normal[3 * c + 1] = normal[3 * c + 2] = __int_as_float(0x7fc00000);
        ids[c] = -1;
        return;
    }
    const uint32_t id = kind == 0 ? (uint32_t)(k & 0xffffffffull) : 0xffffffffu - (uint32_t)(k & 0xffffffffull);
    const DPatch* p = prm.pool + id;
    depth[c] = dot4(ld4((prm.views + view)->oaxis), ld4(p->coord));
    normal[3 * c] = p->normal[0]; normal[3 * c + 1] = p->normal[1]; normal[3 * c + 2] = p->normal[2];
    ids[c] = (int32_t)id;
}

// PatchManager::sortPatches head (patch_manager.cpp:411-415): patches with m_ncc < 0 get their score.
// One wave looks at 64 patches (a lane each) and then scores, one after the other, those that need it: in steady state
// none does, and the launch is 64 times smaller than one wave per patch.
__global__ __launch_bounds__(64) void k_fill_ncc(DParams prm, unsigned long long* evals) {
    WaveCtx wc = make_wave_ctx(prm);
    const int64_t mine = (int64_t)blockIdx.x * 64 + wc.lane;
    bool need = false;
    if (mine < prm.pool_n) need = (prm.pool[mine].flags & 1) && (prm.pool[mine].ncc < 0.0f);
    unsigned long long todo = ballot(need);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        DPatch* p = prm.pool + ((int64_t)blockIdx.x * 64 + l);
        Cand c;
        load_cand(p, wc, c);
        const float ncc = compute_ncc(prm, wc, c.coord, c.normal, c.img, c.nimg);
        if (wc.lane == 0) p->ncc = ncc;
    }
    if (wc.lane == 0 && wc.evals) { atomicAdd(evals, (unsigned long long)wc.evals); atomicAdd(evals + 1, (unsigned long long)wc.view_evals); }
}

// =================================================================== K4: the sweep
// One wavefront per destination cell of the pass colour.  Propagate::propagatePmImage (propagate.cpp:72-124)
// turned inside out: the cell gathers from the cell above/below and the cell beside it, in the order the
// raster sweep would reach them, and runs Propagate::propagatePatch (propagate.cpp:126-218) on its own list.
DEV bool rank_before(float na, int a, float nb, int b) { return (na != nb) ? (na > nb) : (a < b); }
// job -> (swept view, destination cell): one job per (view, row, half column) of the pass colour
DEV void job_cell(const DParams& prm, const SweepArgs& a, int64_t job, int& v, int& cx, int& cy) {
    int s = 0;
    while (s + 1 < a.nsweep_views && job >= a.job_base[s + 1]) ++s;
    v = a.sweep_views[s];
    const int gw = (prm.views + v)->gw, halfw = (gw + 1) / 2;
    const int local = (int)(job - a.job_base[s]);
    cy = local / halfw;
    cx = 2 * (local % halfw) + ((a.colour + cy) & 1);
}
// Work proxy of a job, for cutting the job sequence into ranges of equal WORK (multi-GPU: every rank holds the same index,
// computes the same proxy and finds the same cuts).  A job runs max_propag trials per source entry whose reference view is
// the swept view (propagate.cpp:102-108); a trial into a cell that still has room runs the whole pipeline (preProcess,
// refinePatch's 25 evaluations, postProcess, check), a trial into a full cell one evaluation and, mostly, the pre-filter
// (propagate.cpp:166-173); about four trials in five that run the pipeline add a patch to the cell.
// mode 1: source entries alone; mode 2: 32 per expected pipeline trial + 2 per expected pre-filter trial.
__global__ void k_job_work(DParams prm, SweepArgs a, int mode, int shift, int32_t* __restrict__ work) {
    const int64_t job = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= a.njobs) return;
    int v, cx, cy;
    job_cell(prm, a, job, v, cx, cy);
    const DView* vw = prm.views + v;
    const int gw = vw->gw, gh = vw->gh;
    int w = 0;
    if (cx < gw && cy < gh) {
        const int sxs[3] = {cx, cx - a.inc, cx}, sys[3] = {cy - a.inc, cy, cy};
        const int nsrc = prm.view_propagation ? 3 : 2;
        int n = 0;
        for (int k = 0; k < nsrc; ++k) {
            if (sxs[k] < 0 || gw <= sxs[k] || sys[k] < 0 || gh <= sys[k]) continue;
            const int g = vw->cell_base + sys[k] * gw + sxs[k];
            const csr_off_t sb = prm.csr_start[g];
            const int sn = prm.csr_cnt[g];
            for (int j = 0; j < sn; ++j) n += ((pgrid_ref(prm, sb + j) == v) != (k == 2)) ? 1 : 0;
        }
        if (mode == 1) w = n;
        else {
            const int trials = n * prm.max_propag;
            const int room = max(prm.cap - prm.csr_cnt[vw->cell_base + cy * gw + cx], 0);
            const int full = min(trials, (room * 5 + 3) / 4);
            w = 32 * full + 2 * (trials - full);
        }
    }
    work[job] = (w + (1 << shift) - 1) >> shift;
}
// cuts[r] = first job whose inclusive work prefix exceeds total * r / n (r = 1 .. n-1): exactly one job qualifies per r
__global__ void k_job_cuts(const int32_t* __restrict__ scan, int64_t njobs, int n, int32_t* __restrict__ cuts) {
    const int64_t job = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= njobs) return;
    const long long total = scan[njobs], p0 = scan[job], p1 = scan[job + 1];
    if (p0 == p1) return;
    for (int r = 1; r < n; ++r) {
        const long long target = total * r / n;
        if (p0 <= target && target < p1) cuts[r] = (int32_t)job;
    }
}

#ifndef MVS_SWEEP_WAVES
#define MVS_SWEEP_WAVES 3  // waves per SIMD the register allocator is asked to fit: 168 VGPRs (4 waves = 128 VGPRs spills ~110 of them; measured 19.4 vs 18.9 M patches/s)
#endif
// BIG = false: the sweep proper (k_sweep).  BIG = true: the second tier (k_sweep_retry) -- the few destination cells in which an
// Optim::check met a neighbourhood larger than the wave's LDS id set run again, from the start, with a global-memory table for such
// checks (mvs_check.cuh).  A cell of the first launch that meets such a check gives up: nothing it staged counts (job_nstage = 0)
// and its counters are dropped.  The cells of a colour pass are independent of each other and a cell's run is deterministic (RNG
// keyed by iteration, view, cell, source slot, trial), so the second run repeats the first up to the point where it gave up -- the
// pool patches the first run evicted until then are exactly the first evictions of the second: their kill flags may stand.
#ifdef MVS_STAGE_TIMING
#define ST_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define ST_ADD(k, t0) { const unsigned long long t1_ = ST_NOW(); st_acc[k] += t1_ - (t0); (t0) = t1_; }
#else
#define ST_ADD(k, t0)
#endif
#ifndef MVS_XCD_CHUNK
#define MVS_XCD_CHUNK 128
#endif
template <bool BIG>
DEV void sweep_cell(const DParams& prm, const SweepArgs& a, const int64_t job, int* s_scratch, float* s_texs, int* big_table) {
#ifdef MVS_STAGE_TIMING
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = ST_NOW();
    unsigned long long st_t = st_begin;
#endif
    int v, cx, cy;
    job_cell(prm, a, job, v, cx, cy);
    const DView* vw = prm.views + v;
    const int gw = vw->gw, gh = vw->gh;
    WaveCtx wc = make_wave_ctx(prm);
    if (wc.lane == 0) a.job_nstage[job] = 0;
    if (cx >= gw || cy >= gh) return;
    const int cell = cy * gw + cx;
    const int inc = a.inc;
    const int sxs[2] = {cx, cx - inc}, sys[2] = {cy - inc, cy};
    // any source at all?
    bool has = false;
    for (int k = 0; k < 2; ++k) {
        if (sxs[k] < 0 || gw <= sxs[k] || sys[k] < 0 || gh <= sys[k]) continue;
        const int g = vw->cell_base + sys[k] * gw + sxs[k];
        has |= prm.csr_cnt[g] > 0;
    }
    if (prm.view_propagation) has |= prm.csr_cnt[vw->cell_base + cy * gw + cx] > 0;
    if (!has) return;

    const int tstride = prm.wsz;  // odd for 7x7 / 5x5 windows: the lane-per-pair reads of setRefImage fall in distinct banks
    unsigned n_cand = 0, n_pref = 0, n_patch = 0, n_f0 = 0, n_f1 = 0, n_ins = 0, n_rep = 0;
    // live list of this cell as view-lane arrays: lane k holds entry k (sorted: ncc desc, id asc)
    int L_id = -1;
    float L_ncc = 0.0f;
    int L_n = 0;
    {
        const int g = vw->cell_base + cell;
        const csr_off_t fb = prm.csr_start[g];
        L_n = min(prm.csr_cnt[g], MVS_CAPMAX);
        if (wc.lane < L_n) { L_id = pgrid_id(prm, fb + wc.lane); L_ncc = pgrid_ncc(prm, fb + wc.lane); }
    }
    int ns = 0;  // staged records of this job
    bool gave_up = false;
    const float icx = (float)(prm.csize * (2 * cx + 1) - 1) / 2.0f, icy = (float)(prm.csize * (2 * cy + 1) - 1) / 2.0f;

    // sources: the cell above/below, the cell beside (propagate.cpp:104-108), and -- view propagation, the branch the
    // reference keeps commented out (propagate.cpp:110-120) -- this cell's own list: a patch of another reference view
    // that is listed here proposes itself with view v as the reference
    const int nsrc = prm.view_propagation ? 3 : 2;
    for (int sidx = 0; sidx < nsrc; ++sidx) {
        const int scx = sidx < 2 ? sxs[sidx] : cx, scy = sidx < 2 ? sys[sidx] : cy;
        if (scx < 0 || gw <= scx || scy < 0 || gh <= scy) continue;
        const int g = vw->cell_base + scy * gw + scx;
        const csr_off_t sb = prm.csr_start[g];
        const int sn = min(prm.csr_cnt[g], MVS_CAPMAX);  // a trimmed list: at most MAX_NUM_OF_PATCHES <= 32 entries
        const int as_view = sidx == 2 ? v : -1;
        // the entries of the source list, a lane each, in ONE round trip (the walk used to read entry after entry: a dependent load
        // per entry, six in seven of them only to find another reference view); then the sources in list order
        int E_id = 0;
        bool E_take = false;
        if (wc.lane < sn) { E_id = pgrid_id(prm, sb + wc.lane); E_take = (pgrid_ref(prm, sb + wc.lane) == v) != (sidx == 2); }
        unsigned todo = (unsigned)ballot(E_take);
        while (todo) {
            const int n = __ffs((int)todo) - 1;
            todo &= todo - 1u;
            const DPatch* sp = prm.pool + rli(E_id, n);
            const int srcslot = sidx * prm.cap + n;
            // ---- Propagate::propagatePatch, propagate.cpp:153-213
            for (int it = 0; it < prm.max_propag; ++it) {
                ST_ADD(7, st_t)
                const int np = L_n;
                const uint32_t k0 = (uint32_t)a.iter, k1 = (uint32_t)v, k2 = (uint32_t)cell, k3 = (uint32_t)(srcslot * 16 + it);
                Cand c;
                int worst = -1;
                float worst_ncc = 0.0f;
                F3 ic;
                if (np < prm.cap) {
                    const float ra = rng_uniform(prm.seed, k0, k1, k2, k3, 0) * (float)prm.csize;
                    const float rb = rng_uniform(prm.seed, k0, k1, k2, k3, 1) * (float)prm.csize;
                    ic = {icx + ra, icy + rb, 1.0f};
                } else {
                    worst = rli(L_id, prm.cap - 1);
                    worst_ncc = rlf(L_ncc, prm.cap - 1);
                    const DPatch* wp = worst >= MVS_NEWBASE ? a.staging + (worst - MVS_NEWBASE) : prm.pool + worst;
                    ic = project(vw, ld4(wp->coord), prm.level);
                }
                {
                    Cand src;  // loaded per trial: its registers are free again during the refinement
                    load_cand(sp, wc, src);
                    const bool gen_ok = generate_patch(prm, wc, s_scratch, src, ic, c, as_view, np >= prm.cap);
                    ST_ADD(1, st_t)
                    if (!gen_ok) continue;
                }
                ++n_cand;
                if (np >= prm.cap && c.ncc < worst_ncc) { ++n_pref; continue; }
                ++n_patch;
                const int pre_r = pre_process(prm, wc, s_scratch, c);
                ST_ADD(2, st_t)
                if (pre_r == -1) { ++n_f0; continue; }
                float keep_w;  // computeWeights of refinePatch, for the m_ncc postProcess takes from its first evaluation
                refine_patch(prm, wc, c, k0, k1, k2, k3, &keep_w);
                ST_ADD(3, st_t)
                const int post_r = post_process(prm, wc, s_scratch, s_texs, tstride, c, keep_w, true);
                ST_ADD(4, st_t)
                if (post_r == -1) { ++n_f1; continue; }
                if (prm.depth >= 2 && prm.enable_check) {  // Optim::check, optim.cpp:292
                    __syncthreads();
                    if (wc.lane < MVS_CAPMAX) s_scratch[wc.lane] = L_id;  // publish the live list of this cell
                    __syncthreads();
                    #ifdef MVS_STAGE_TIMING
                    const CheckCtx cx{a.staging, v, cell, L_n, s_scratch, st_acc};
#else
                    const CheckCtx cx{a.staging, v, cell, L_n, s_scratch, nullptr};
#endif
                    const int chk_r = check_patch<BIG>(prm, wc, cx, c, s_texs, a.error_flag, big_table);
                    ST_ADD(5, st_t)
                    if (!BIG && chk_r < 0) { gave_up = true; break; }  // the neighbourhood does not fit LDS: this cell goes to k_sweep_retry
                    if (chk_r) { ++n_f1; continue; }
                }
                // staging slot for the accepted patch
                unsigned long long slot64 = 0;
                if (wc.lane == 0) slot64 = atomicAdd(a.stage_counter, 1ull);
                const int64_t slot = ((int64_t)rfl((int)(slot64 >> 32)) << 32) | (uint32_t)rfl((int)(slot64 & 0xffffffffull));
                if (slot >= a.staging_cap || ns >= a.maxstage) {
                    if (wc.lane == 0) atomicOr(a.error_flag, 1);
                    continue;
                }
                if (np == prm.cap) {  // removePatch(worst), propagate.cpp:198-201
                    if (wc.lane == 0) {
                        if (worst >= MVS_NEWBASE) a.staging[worst - MVS_NEWBASE].flags &= ~1;
                        else a.kill[worst] = 1;
                    }
                    --L_n;
                    ++n_rep;
                } else ++n_ins;
                store_cand(a.staging + slot, wc, c, 1 | (v << 8), cell);
                if (wc.lane == 0) a.job_stage[job * a.maxstage + ns] = (int32_t)slot;
                ++ns;
                // PatchManager::addPatch into this cell's own list if the patch lands here
                const bool lands = ballot(wc.lane < c.nimg && c.img == v && c.gy * gw + c.gx == cell) != 0ull;
                if (lands) {
                    const int nid = MVS_NEWBASE + (int)slot;
                    const int pos = __popcll(ballot(wc.lane < L_n && rank_before(L_ncc, L_id, c.ncc, nid)));
                    const int up_id = __shfl_up(L_id, 1);
                    const float up_ncc = __shfl_up(L_ncc, 1);
                    if (wc.lane > pos) { L_id = up_id; L_ncc = up_ncc; }
                    if (wc.lane == pos) { L_id = nid; L_ncc = c.ncc; }
                    ++L_n;
                }
                ST_ADD(6, st_t)
            }
            if (gave_up) break;
        }
        if (gave_up) break;
    }
    if (!BIG && gave_up) {
        if (wc.lane == 0) { a.job_nstage[job] = 0; a.retry_jobs[atomicAdd(a.nretry, 1)] = (int32_t)job; }
        return;
    }
    if (wc.lane == 0) {
        a.job_nstage[job] = ns;
        DCounters* C = a.counters + (job & (MVS_COUNTER_SLOTS - 1));  // partial sums, added up by the host: a million waves on ONE cache line queue up
        if (n_cand) atomicAdd(&C->candidates, (unsigned long long)n_cand);
        if (n_pref) atomicAdd(&C->prefiltered, (unsigned long long)n_pref);
        if (n_patch) atomicAdd(&C->patches, (unsigned long long)n_patch);
        if (n_f0) atomicAdd(&C->fail0, (unsigned long long)n_f0);
        if (n_f1) atomicAdd(&C->fail1, (unsigned long long)n_f1);
        if (n_ins) atomicAdd(&C->inserted, (unsigned long long)n_ins);
        if (n_rep) atomicAdd(&C->replaced, (unsigned long long)n_rep);
        if (wc.evals) atomicAdd(&C->evals, (unsigned long long)wc.evals);
        if (wc.view_evals) atomicAdd(&C->view_evals, (unsigned long long)wc.view_evals);
#ifdef MVS_STAGE_TIMING
        st_acc[0] = ST_NOW() - st_begin;
        for (int k = 0; k < 12; ++k) if (st_acc[k]) atomicAdd(&C->stage[k], st_acc[k]);
        atomicMax(&C->stage[12], st_acc[12]); atomicMax(&C->stage[13], st_acc[13]);
        atomicAdd(&C->stage[14], wc.st_acc[3]); atomicAdd(&C->stage[15], wc.st_acc[4]);  // inside postProcess: setRefImage's pair sums and choice
#endif
    }
}

__global__ __launch_bounds__(64, MVS_SWEEP_WAVES) void k_sweep(DParams prm, SweepArgs a) {
    __shared__ int s_scratch[192];
    extern __shared__ float s_texs[];
    // XCD-aware job order: blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8).
#if MVS_XCD_CHUNK > 0
    // chunks of MVS_XCD_CHUNK consecutive jobs (a stretch of one grid row) go to one XCD, consecutive chunks to
    // consecutive XCDs: the destination and its source cells share an L2, and every XCD gets the same mix of cheap and
    // expensive regions.  (One contiguous band of cells per XCD left XCDs idle for a quarter of the launch: the bands
    // -- one and a half views each -- differ in work; measured 770 -> 603 ms per iteration.)
    const int64_t bi = blockIdx.x >> 3;
    const int64_t job = a.job_lo + (bi / MVS_XCD_CHUNK) * (8 * MVS_XCD_CHUNK) + (int64_t)(blockIdx.x & 7u) * MVS_XCD_CHUNK + (bi % MVS_XCD_CHUNK);
#else
    const int64_t chunk = (a.job_hi - a.job_lo + 7) / 8;
    const int64_t job = a.job_lo + (int64_t)(blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
#endif
    if (job >= a.job_hi) return;
    sweep_cell<false>(prm, a, job, s_scratch, s_texs, nullptr);
}
// the second tier: block b runs the cells retry_jobs[b], retry_jobs[b + gridDim.x], ... with big_tables slot b
__global__ __launch_bounds__(64, 1) void k_sweep_retry(DParams prm, SweepArgs a, int nretry) {
    __shared__ int s_scratch[192];
    extern __shared__ float s_texs[];
    int* big_table = a.big_tables + (size_t)blockIdx.x * MVS_FILTER2_HASH_CAP;
    for (int k = blockIdx.x; k < nretry; k += gridDim.x) {
        __syncthreads();
        sweep_cell<true>(prm, a, (int64_t)a.retry_jobs[k], s_scratch, s_texs, big_table);
    }
}

// =================================================================== commit
// alive staged records per job -> cnt[job]
__global__ void k_commit_count(SweepArgs a, int32_t* __restrict__ cnt) {
    const int64_t job = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= a.njobs) return;
    const int ns = a.job_nstage[job];
    int c = 0;
    for (int k = 0; k < ns; ++k) c += a.staging[a.job_stage[job * a.maxstage + k]].flags & 1;
    cnt[job] = c;
}
// copy alive staged records to dst[base[job] + r] in creation order: global order (view, cell, sequence)
__global__ void k_commit_copy(SweepArgs a, const int32_t* __restrict__ base, DPatch* __restrict__ dst, int64_t dst_cap, int32_t* __restrict__ per_view,
                              int keep_key) {
    const int64_t job = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= a.njobs) return;
    const int ns = a.job_nstage[job];
    int r = 0;
    for (int k = 0; k < ns; ++k) {
        const DPatch* sp = a.staging + a.job_stage[job * a.maxstage + k];
        if (!(sp->flags & 1)) continue;
        const int64_t o = (int64_t)base[job] + r++;
        if (o >= dst_cap) { atomicOr(a.error_flag, 2); return; }
        const uint4* s4 = reinterpret_cast<const uint4*>(sp);
        uint4* d4 = reinterpret_cast<uint4*>(dst + o);
        for (int w = 0; w < (int)MVS_REC_U4; ++w) d4[w] = s4[w];
        if (per_view) atomicAdd(&per_view[(sp->flags >> 8) & 0xff], 1);
        if (!keep_key) { dst[o].flags = 1; dst[o].id = 0; }
    }
}
__global__ void k_kill_count(const uint8_t* __restrict__ kill, int64_t n, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = kill[i] ? 1 : 0;
}
__global__ void k_kill_export(const uint8_t* __restrict__ kill, int64_t n, const int32_t* __restrict__ base, int32_t* __restrict__ ids, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && kill[i] && base[i] < cap) ids[base[i]] = (int32_t)i;
}
__global__ void k_apply_kill_flags(DPatch* pool, uint8_t* kill, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && kill[i]) { pool[i].flags &= ~1; kill[i] = 0; }
}
__global__ void k_apply_kill_ids(DPatch* pool, const int32_t* __restrict__ ids, int64_t n, int64_t pool_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0 && ids[i] < pool_n) pool[ids[i]].flags &= ~1;
}
__global__ void k_append_records(DPatch* pool, int64_t pool_n, const DPatch* __restrict__ recs, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* s4 = reinterpret_cast<const uint4*>(recs + i);
    uint4* d4 = reinterpret_cast<uint4*>(pool + pool_n + i);
    for (int w = 0; w < (int)MVS_REC_U4; ++w) d4[w] = s4[w];
    pool[pool_n + i].flags = 1;
    pool[pool_n + i].id = 0;
}
__global__ void k_alive_count(const DPatch* __restrict__ pool, int64_t n, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = pool[i].flags & 1;
}
__global__ void k_alive_gather(const DPatch* __restrict__ pool, int64_t n, const int32_t* __restrict__ base, DPatch* __restrict__ out, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(pool[i].flags & 1) || base[i] >= cap) return;
    const uint4* s4 = reinterpret_cast<const uint4*>(pool + i);
    uint4* d4 = reinterpret_cast<uint4*>(out + base[i]);
    for (int w = 0; w < (int)MVS_REC_U4; ++w) d4[w] = s4[w];
    out[base[i]].id = (int32_t)i;
}

// =================================================================== Filter::run (pmmvps/filter.cpp:25-49)
DEV void set_vgrids(const DParams& prm, const WaveCtx& wc, Cand& c) {  // PatchManager::setVGrids, patch_manager.cpp:252-261
    c.vgx = 0; c.vgy = 0;
    if (wc.lane < c.nvimg) cell_of(prm, prm.views + c.vimg, c.coord, c.vgx, c.vgy);
}
DEV void store_lists(DPatch* p, const WaveCtx& wc, const Cand& c) {
    if (wc.lane == 0) { p->nimages = c.nimg; p->nvimages = c.nvimg; }
    if (wc.lane < MVS_MAXI) {
        p->images[wc.lane] = (uint8_t)(wc.lane < c.nimg ? c.img : 0);
        p->vimages[wc.lane] = (uint8_t)(wc.lane < c.nvimg ? c.vimg : 0);
    }
}
// Filter::setVGridsVPGrids (filter.cpp:657-664): m_vimages cleared (additive == 0) or kept, then setVImagesVGrids
// (all Filter::run kernels take a patch range [first, last): a rank of a multi-GPU job filters its share of the pool)
// GL lanes per patch (a lane per view; GL >= the views and >= the list cap), 64 / GL patches per wave: with 12 views a wave per
// patch leaves 52 lanes idle through the two dependent gathers of isVisible.
template <int GL>
__global__ __launch_bounds__(64) void k_filter_vimages(DParams prm, int additive, int64_t first, int64_t last, const uint32_t* __restrict__ dirty) {
    __shared__ int s_new[64];
    const int lane = (int)threadIdx.x, g = lane / GL, i = lane % GL;
    const int64_t id = first + (int64_t)blockIdx.x * (64 / GL) + g;
    const bool have = id < last;
    DPatch* p = prm.pool + (have ? id : first);
    const bool act = have && (p->flags & 1);
    const int nimg = act ? min(p->nimages, MVS_LISTCAP) : 0;
    const int nv0 = act && additive ? min(p->nvimages, MVS_LISTCAP) : 0;
    const int my_img = i < MVS_MAXI ? (int)p->images[i] : 0, my_vimg = i < MVS_MAXI ? (int)p->vimages[i] : 0;
    // the views the patch is listed in, as a mask shared by the group
    unsigned long long listed = (i < nimg ? 1ull << my_img : 0ull) | (i < nv0 ? 1ull << my_vimg : 0ull);
#pragma unroll
    for (int d = 1; d < GL; d <<= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)listed, d), hi = (unsigned)__shfl_xor((int)(unsigned)(listed >> 32), d);
        listed |= ((unsigned long long)hi << 32) | lo;
    }
    bool q = false;
    if (act && i < prm.nviews && !((listed >> i) & 1ull)) {
        Cand c;
        c.coord = ld4(p->coord); c.normal = ld4(p->normal);
        int ix, iy;
        const DView* vw = prm.views + i;
        cell_of(prm, vw, c.coord, ix, iy);
        // additive pass after removals only (`dirty`): the view was tested and found hidden when the lists were last brought up to
        // date, and the answer can only change where the depth map did -- in a cell whose nearest patch has been removed
        bool test = true;
        if (dirty && !(ix < 0 || vw->gw <= ix || iy < 0 || vw->gh <= iy)) { const int cell = vw->cell_base + iy * vw->gw + ix; test = (dirty[cell >> 5] >> (cell & 31)) & 1u; }
        q = test && is_visible(prm, c, i, ix, iy, prm.neighborThreshold) != 0;
    }
    const unsigned long long all = __ballot(q);
    const unsigned long long gm = GL == 64 ? all : (all >> (GL * g)) & ((1ull << (GL & 63)) - 1ull);
    const int added = (int)__popcll(gm), pos = (int)__popcll(gm & ((1ull << i) - 1ull));
    if (q) s_new[GL * g + pos] = i;  // the r-th view added, in ascending order
    __syncthreads();
    const int nv = min(MVS_LISTCAP, nv0 + added);
    if (act) {
        if (i == 0) { p->nimages = nimg; p->nvimages = nv; }
        if (i < MVS_MAXI) {
            p->images[i] = (uint8_t)(i < nimg ? my_img : 0);
            p->vimages[i] = (uint8_t)(i < nv0 ? my_vimg : (i < nv ? s_new[GL * g + (i - nv0)] : 0));
        }
        for (int k = i + GL; k < MVS_MAXI; k += GL) { p->images[k] = 0; p->vimages[k] = 0; }  // storage beyond the list cap (GL >= MVS_LISTCAP)
    }
}
// Filter::filterOutside, filter.cpp:51-106: gain < 0 -> removed
__global__ __launch_bounds__(64) void k_filter_outside(DParams prm, uint8_t* kill, int64_t first) {
    __shared__ int s_dummy[1];
    __shared__ int s_gain[MVS_LISTCAP];
    const int64_t pid = first + (int64_t)blockIdx.x;
    const DPatch* p = prm.pool + pid;
    if (!(p->flags & 1)) return;
    WaveCtx wc = make_wave_ctx(prm);
    Cand c;
    load_cand(p, wc, c);
    set_grids(prm, wc, c);
    set_vgrids(prm, wc, c);
    const CheckCtx cx{prm.pool, -1, -1, 0, s_dummy, nullptr};
    const float gain = prm.geo ? compute_gain<true>(prm, wc, cx, c, s_gain) : compute_gain<false>(prm, wc, cx, c, s_gain);
    if (wc.lane == 0 && gain < 0.0f) kill[pid] = 1;
}
// The packed geometry of Filter::run's stages (DParams::geo, geo_ref): a lane per pool record.  *bad becomes 1 when an alive record's
// coord.w is not 1 or its normal.w not 0 -- the packed form leaves them out, so the stages then read the records as before.
__global__ void k_geo_pack(const DPatch* pool, int64_t n, float4* geo, uint8_t* ref, int* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DPatch* p = pool + i;
    const F4 c = ld4(p->coord), nr = ld4(p->normal);
    geo[2 * i] = make_float4(c.x, c.y, c.z, p->dscale);
    geo[2 * i + 1] = make_float4(nr.x, nr.y, nr.z, p->ncc);
    ref[i] = p->images[0];
    if ((p->flags & 1) && !(c.w == 1.0f && nr.w == 0.0f)) *bad = 1;
}
// Filter::filterExact, filter.cpp:148-263.
// Visibility phase (filterExactSub: PatchManager::isVisible in the patch's cell and its four neighbours, per view of
// m_images): FOUR patches per wave, lane 16 q + i = view i of patch 4 * block + q.  The five depth-map cells of a lane are
// loaded together, then the five patches they name, then the five tests run on one ray / unit / factor -- two dependent
// gathers per lane instead of ten, at four times the lanes per instruction of the one-patch-per-wave form.
// Then, patch by patch: the surviving views in ascending order and Optim::setRefImage with the whole wave.
#define MVS_FE_LANES (MVS_LISTCAP <= 16 ? 16 : (MVS_LISTCAP <= 32 ? 32 : 64))  // lanes per patch in the visibility phase
#define MVS_FE_PATCHES (64 / MVS_FE_LANES)
__global__ __launch_bounds__(64) void k_filter_exact(DParams prm, uint8_t* kill, unsigned long long* evals, unsigned long long* stage, int64_t first, int64_t last) {
    __shared__ int s_scratch[192];
    extern __shared__ float s_texs[];
    WaveCtx wc = make_wave_ctx(prm);
#ifdef MVS_STAGE_TIMING
    const unsigned long long fe_begin = (unsigned long long)__builtin_amdgcn_s_memtime();
    WC_T0(wc)
#endif
    const int q = wc.lane / MVS_FE_LANES, i = wc.lane % MVS_FE_LANES;
    const int64_t id = first + (int64_t)blockIdx.x * MVS_FE_PATCHES + q;
    const bool have = id < last;
    const DPatch* pl = prm.pool + (have ? id : 0);
    const bool alive_l = have && (pl->flags & 1);
    const int nimg_l = alive_l ? min(pl->nimages, MVS_LISTCAP) : 0;
    const bool act = i < nimg_l;
    const int image = act ? (int)pl->images[i] : 0;
    const F4 coord = ld4(pl->coord), normal = ld4(pl->normal);
    WC_ADD(wc, 5)
    bool safe = false;
    {
        const DView* vw = prm.views + image;
        const int w = vw->gw, h = vw->gh, cbase = vw->cell_base;
        const F4 ctr = ld4(vw->center);
        const float ips = vw->ipscale;
        int x, y;
        cell_of(prm, vw, coord, x, y);  // PatchManager::setGrids
        const bool in = act && !(x < 0 || w <= x || y < 0 || h <= y);
        // the five cells of filterExactSub, each behind its guard (filter.cpp:221-251)
        const int dxs[5] = {0, -1, 1, 0, 0}, dys[5] = {0, 0, 0, -1, 1};
        bool ok[5];
        unsigned long long dp[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int xx = x + dxs[k], yy = y + dys[k];
            ok[k] = in && 0 <= xx && xx < w && 0 <= yy && yy < h;  // the guards 0 < x, x < w-1, ... and isVisible's own range test
            dp[k] = prm.dpgrid[cbase + (ok[k] ? yy * w + xx : 0)];
        }
        F4 qc[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) qc[k] = ld4((prm.pool + (dp[k] == ~0ull ? 0u : (uint32_t)(dp[k] & 0xffffffffull)))->coord);
        // PatchManager::isVisible, patch_manager.cpp:335-376: everything but the depth difference is the same for the five
        const F4 ray = nrm4(sub4(coord, ctr));
        const float factor = fminf(2.0f, 2.0f + dot4(ray, normal));  // a float in the reference (patch_manager.cpp:366), see is_visible
        float unit = 1.0f;  // get_unit
        if (ips != 0.0f) unit = (2.0f * norm4(sub4(coord, ctr)) * (float)(1 << prm.level)) / ips;
        const float rhs = (unit * (float)prm.csize * prm.neighborThreshold1) * factor;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float diff = dot4(ray, sub4(coord, qc[k]));
            const bool vis = prm.depth == 0 || dp[k] == ~0ull || diff < rhs;
            safe |= ok[k] && vis;
        }
    }
    WC_ADD(wc, 6)
    const unsigned long long safe_b = ballot(safe);
    const unsigned long long alive_b = ballot(alive_l);
    const int tstride = prm.wsz;  // odd for 7x7 / 5x5 windows: the lane-per-pair reads of setRefImage fall in distinct banks
    // The sampling frames of setRefImage for all patches of the wave side by side, in the lanes of the visibility phase (lane
    // MVS_FE_LANES q + i = view i of patch q): getPAxes from the patch's first surviving view in ascending order, i.e. its smallest,
    // and make_frame per surviving view -- once per wave, where patch by patch it ran MVS_FE_PATCHES times with a handful of lanes
    // at work (a third of this kernel's wave time).  The loop below hands each patch's frames to its view lanes.
    Frame fpre = {};
    {
        const unsigned long long gbits = MVS_FE_LANES == 64 ? ~0ull : ((1ull << (MVS_FE_LANES & 63)) - 1ull) << ((MVS_FE_LANES * q) & 63);
        const bool skip_l = (pl->flags & MVS_FLAG_SETTLED) && (int)__popcll(safe_b & gbits) == nimg_l;  // settled and keeps every view
        if (ballot(alive_l && !skip_l)) {
            int vmin = safe ? image : INT_MAX;
            for (int d = 1; d < MVS_FE_LANES; d <<= 1) vmin = min(vmin, __shfl_xor(vmin, d));
            F4 px, py;
            get_paxes(prm, prm.views + (vmin == INT_MAX ? 0 : vmin), coord, normal, px, py);
            fpre = make_frame(prm, coord, px, py, normal, image, safe);
        }
    }
    WC_ADD(wc, 1)
    for (int g = 0; g < MVS_FE_PATCHES; ++g) {
        if (!((alive_b >> ((MVS_FE_LANES * g) & 63)) & 1ull)) continue;
        DPatch* p = prm.pool + (first + (int64_t)blockIdx.x * MVS_FE_PATCHES + g);
        const int pflags = p->flags;
        Cand c;
        load_cand(p, wc, c);
        const vmask_t sm = (vmask_t)((safe_b >> ((MVS_FE_LANES * g) & 63)) & (MVS_FE_LANES == 64 ? ~0ull : (1ull << (MVS_FE_LANES & 63)) - 1ull));  // bit i: view m_images[i] of patch g survives
        // A patch whose list IS the outcome of this stage's setRefImage for its set of views (MVS_FLAG_SETTLED, set below) and which
        // keeps every view: the outcome is a function of the position, the normal, the set of views and the pyramids alone (the
        // survivors are taken in ascending view order, the axes from the first of them), so it would come out as it stands.
        if ((pflags & MVS_FLAG_SETTLED) && (int)vpop(sm) == c.nimg) continue;
        // the survivors in ascending view order (the image-major loop of filterExactSub)
        __syncthreads();
        if (wc.lane < MVS_MAXVIEWS) s_scratch[wc.lane] = 0;
        __syncthreads();
        if (wc.lane < c.nimg && ((sm >> wc.lane) & 1u)) s_scratch[c.img] = 1 + MVS_FE_LANES * g + wc.lane;  // where this view's frame lies
        __syncthreads();
        const int src1 = s_scratch[wc.lane];
        const bool present = src1 != 0;
        const unsigned long long pm = ballot(present);
        const int pos = __popcll(pm & ((1ull << wc.lane) - 1ull));
        __syncthreads();
        if (present && pos < MVS_LISTCAP) { s_scratch[64 + pos] = wc.lane; s_scratch[128 + pos] = src1 - 1; }
        __syncthreads();
        c.nimg = min((int)__popcll(pm), MVS_LISTCAP);
        c.img = s_scratch[64 + wc.lane];
        if (prm.minImageNum <= c.nimg) {
            // view lane k < nimg takes the frame of its view; the other lanes a harmless one (the origin of some view's image, ok = 0)
            Frame f;
            {
                const bool mine = wc.lane < c.nimg;
                const int src = mine ? s_scratch[128 + wc.lane] : MVS_FE_LANES * g;
                f.tlx = __shfl(fpre.tlx, src); f.tly = __shfl(fpre.tly, src);
                f.dxx = __shfl(fpre.dxx, src); f.dxy = __shfl(fpre.dxy, src); f.dyx = __shfl(fpre.dyx, src); f.dyy = __shfl(fpre.dyy, src);
                f.w = __shfl(fpre.w, src); f.ok = __shfl(fpre.ok, src);
                f.img_lo = (unsigned)__shfl((int)fpre.img_lo, src); f.img_hi = (unsigned)__shfl((int)fpre.img_hi, src);
                if (!mine) { f.tlx = f.tly = f.dxx = f.dxy = f.dyx = f.dyy = 0.0f; f.ok = 0; }
            }
            WC_ADD(wc, 7)
            set_ref_image(prm, wc, s_texs, tstride, c, nullptr, &f);  // a patch that gets here made the wave compute fpre
            store_lists(p, wc, c);
            if (wc.lane == 0) p->flags = pflags | MVS_FLAG_SETTLED;
            WC_ADD(wc, 7)
        } else {
            if (wc.lane == 0) kill[first + (int64_t)blockIdx.x * MVS_FE_PATCHES + g] = 1;
        }
    }
    // the two work counts as 1024 partial sums (the host adds them): 1.5 M waves adding to one address queue up behind each other
    if (wc.lane == 0 && wc.evals) { atomicAdd(evals + 2 * (blockIdx.x & 1023u), (unsigned long long)wc.evals); atomicAdd(evals + 2 * (blockIdx.x & 1023u) + 1, (unsigned long long)wc.view_evals); }
#ifdef MVS_STAGE_TIMING
    if (stage && wc.lane == 0) {
        atomicAdd(stage, (unsigned long long)__builtin_amdgcn_s_memtime() - fe_begin);
        for (int k = 1; k < 8; ++k) atomicAdd(stage + k, wc.st_acc[k]);
    }
#endif
}
// Filter::filterNeighbor(1), filter.cpp:265-327: fewer than 6 neighbours, or a bad quadric fit.
// First launch (todo == nullptr): every patch, with a hash set / row buffer that fits 2 waves per SIMD; a patch that
// does not fit is appended to `retry`.  Second launch: only those patches, with the large configuration.
template <int HCAP, int RCAP>
__global__ __launch_bounds__(64) void k_filter_neighbor(DParams prm, uint8_t* kill, const int32_t* todo, int32_t ntodo, int32_t* retry, int32_t* nretry,
                                                        int32_t* overflow, unsigned long long* stats /* [1024][4], spread over blocks */, int64_t first) {
    extern __shared__ float s_lds[];
    int* const s_dummy = reinterpret_cast<int*>(s_lds);  // the live list of a destination cell: none here (live_view = -1), never read
    const int64_t id = todo ? (blockIdx.x < (unsigned)ntodo ? todo[blockIdx.x] : -1) : first + (int64_t)blockIdx.x;
    if (id < 0 || id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    WaveCtx wc = make_wave_ctx(prm);
#ifdef MVS_STAGE_TIMING  // wave cycles: stats[4096 + k], k = 0 whole wave, 1 load + grids, 9 / 10 findNeighbors' two phases, 11 filterQuad
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = CK_NOW();
    auto st_flush = [&]() { st_acc[0] = CK_NOW() - st_begin; if (wc.lane == 0) for (int k = 0; k < 12; ++k) if (st_acc[k]) atomicAdd(stats + 4096 + k, st_acc[k]); };
#endif
    Cand c;
    load_cand(p, wc, c);
    set_grids(prm, wc, c);
#ifdef MVS_STAGE_TIMING
    const CheckCtx cx{prm.pool, -1, -1, 0, s_dummy, st_acc};
    st_acc[1] = CK_NOW() - st_begin;
#else
    const CheckCtx cx{prm.pool, -1, -1, 0, s_dummy, nullptr};
#endif
    int* table = reinterpret_cast<int*>(s_lds);
    unsigned st4[4] = {0u, 0u, 0u, 0u};
    const bool pk = prm.geo != nullptr;
    // the first launch keeps the marks of the row walk behind its set (64 dwords more than the set needs); the second launch's table
    // fills the 64 KB a block may have, and it walks its rows by the binary search
    constexpr bool MK = MVS_FN_MARKS != 0 && HCAP == MVS_FILTER_HASH_CAP;
    int* const marks = reinterpret_cast<int*>(s_lds) + MVS_SET_LDS_FLOATS(HCAP, RCAP);
    const int n = pk ? find_neighbors<HCAP, false, true, MK>(prm, wc, cx, c, table, 4.0f, 2, st4, marks) : find_neighbors<HCAP, false, false, MK>(prm, wc, cx, c, table, 4.0f, 2, st4, marks);
    if (n < 0 || n > RCAP) {
        if (wc.lane == 0) {
            if (retry) retry[atomicAdd(nretry, 1)] = (int32_t)id;
            else atomicOr(overflow, 4);
        }
#ifdef MVS_STAGE_TIMING
        st_flush();
#endif
        return;
    }
    if (wc.lane < 4) atomicAdd(stats + 4 * (blockIdx.x & 1023u) + wc.lane, (unsigned long long)(wc.lane == 0 ? st4[0] : wc.lane == 1 ? st4[1] : wc.lane == 2 ? st4[2] : st4[3]));
    const bool reject = n < 6 || (pk ? filter_quad<false, true>(prm, wc, cx, c, table, n, s_lds + rows_offset(n)) : filter_quad<false, false>(prm, wc, cx, c, table, n, s_lds + rows_offset(n))) != 0;
    if (wc.lane == 0 && reject) kill[id] = 1;
#ifdef MVS_STAGE_TIMING
    st_acc[11] = CK_NOW() - st_begin - st_acc[1] - st_acc[9] - st_acc[10];
    st_flush();
#endif
}
// Filter::filterSmallGroups, filter.cpp:432-578, as connected components of the symmetrised relation: lock-free
// union-find, the smaller id becomes the root.
DEV int uf_find(int* parent, int a) {
    while (true) {
        const int pa = __atomic_load_n(&parent[a], __ATOMIC_RELAXED);
        if (pa == a) return a;
        const int gp = __atomic_load_n(&parent[pa], __ATOMIC_RELAXED);
        if (gp != pa) __atomic_store_n(&parent[a], gp, __ATOMIC_RELAXED);
        a = pa;
    }
}
DEV void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        if (atomicCAS(&parent[b], b, a) == b) return;
    }
}
__global__ void k_groups_init(int* parent, int* size, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = (int)i; size[i] = 0; }
}
// The relation of Filter::filterSmallGroups: q hangs on p when q is listed (m_pgrids or m_vpgrids) in the 3x3 cells around p in p's
// reference view and isNeighbor(p, q).  MODE 0: every edge joins its two ends (connected components of the symmetrised relation).
// MODE 1 (first pass of the literal labelling): only an edge whose reverse exists too joins them -- q -> p exists iff p is listed in
// q's reference view w (w in p's m_images or m_vimages, p's cell there inside the grid) within one cell of q's own cell; isNeighbor is
// symmetric.  MODE 2 (second pass): the edges that still run between two different sets, as (root of p, root of q) pairs.
// One wave per patch: the 18 lists (3x3 cells, m_pgrids and m_vpgrids) are laid end to end and the lanes take consecutive entries, so
// a wave reads its 48-byte entries as contiguous runs (a lane per patch reads one cache line per lane and load, and 2048 lanes per
// CU evict each other's lines between the three loads of an entry: 23.7 ms per call at 1080p against 13.5 for this form).
template <int MODE>
__global__ __launch_bounds__(256) void k_groups_edges(DParams prm, int* parent, int2* edges, int* nedges, int cap) {
    const int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    if (id >= prm.pool_n) return;
    const DPatch* p = prm.pool + id;
    if (!(p->flags & 1)) return;
    const PGeo me = load_geo(p);
    const DView* vw = prm.views + me.ref;
    unsigned long long listed = 0ull;  // the views p is listed in
    if (MODE == 1) {
        for (int i = 0; i < min(p->nimages, MVS_LISTCAP); ++i) listed |= 1ull << p->images[i];
        for (int i = 0; i < min(p->nvimages, MVS_LISTCAP); ++i) listed |= 1ull << p->vimages[i];
    }
    int gx, gy;
    cell_of(prm, vw, me.coord, gx, gy);
    // lane l < 18: list l = (kind, dy, dx); its start and length, and the running total before it
    csr_off_t lstart = 0;
    int ln = 0;
    if (lane < 18) {
        const int kind = lane / 9, yt = gy + (lane % 9) / 3 - 1, xt = gx + lane % 3 - 1;
        if (!(yt < 0 || vw->gh <= yt || xt < 0 || vw->gw <= xt)) {
            const int g = vw->cell_base + yt * vw->gw + xt;
            lstart = kind == 0 ? prm.csr_start[g] : prm.vcsr_start[g];
            ln = kind == 0 ? prm.csr_cnt[g] : prm.vcsr_cnt[g];
        }
    }
    int before = ln;  // inclusive running total over the lanes, then exclusive
    for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(before, d); if (lane >= d) before += o; }
    const int total = __builtin_amdgcn_readlane(before, 17);
    before -= ln;
    int myroot = -1;
    // entry k of the 18 lists laid end to end: the list it falls in is the last lane whose list begins at or before k (lanes 18.. hold the
    // total; an empty list never wins)
    auto entry_at = [&](int k, int& eid) -> bool {
        int lo = 0;
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1) { const int pc = __shfl(before, lo + step); if (pc <= k) lo += step; }
        const int l_first = __shfl(before, lo);
        const csr_off_t l_start = __shfl(lstart, lo);
        if (k >= total) return false;
        eid = (lo >= 9 ? prm.vcsr_id32 : prm.csr_id32)[l_start + (k - l_first)];  // the lists hold ids; edge() fetches the geometry from the record
        return true;
    };
    auto edge = [&](const int eid) {
        if (eid == (int)id) return;
        // two patches that hang on the same node are in one set already: one load instead of the predicate and the two root searches
        // of a union -- the common case once the large component has formed and its paths are short
        if (MODE != 2 && __atomic_load_n(&parent[eid], __ATOMIC_RELAXED) == __atomic_load_n(&parent[id], __ATOMIC_RELAXED)) return;
        // (only for the entries the same-set test lets through)  A listed patch may be dead: the m_pgrids lists are not rebuilt after
        // Filter::filterNeighbor's handful of removals (mvs_engine_filter) -- its flags lie in the line that holds its geometry
        if (!(prm.pool[eid].flags & 1)) return;
        const PGeo q = load_geo(prm.pool + eid);
        if (!is_neighbor(prm, me, q, 1.0f /* m_neighborThreshold2, pmmvps.cpp:61 */)) return;
        if (MODE == 0) uf_union(parent, (int)id, eid);
        else if (MODE == 1) {
            if (!((listed >> q.ref) & 1ull)) return;
            const DView* qw = prm.views + q.ref;
            int px, py, qx, qy;
            cell_of(prm, qw, me.coord, px, py);
            cell_of(prm, qw, q.coord, qx, qy);
            if (px < 0 || qw->gw <= px || py < 0 || qw->gh <= py) return;
            if (abs(px - qx) <= 1 && abs(py - qy) <= 1) uf_union(parent, (int)id, eid);
        } else {
            if (myroot < 0) myroot = uf_find(parent, (int)id);
            const int rq = uf_find(parent, eid);
            if (rq != myroot) {
                const int k2 = atomicAdd(nedges, 1);
                if (k2 < cap) edges[k2] = make_int2(myroot, rq);
            }
        }
    };
    for (int k0 = 0; k0 < total; k0 += 128) {  // two rounds of entries in flight (a patch meets ~120)
        int e0 = -1, e1 = -1;
        const bool h0 = entry_at(k0 + lane, e0), h1 = entry_at(k0 + 64 + lane, e1);
        if (h0) edge(e0);
        if (h1) edge(e1);
    }
}
// `sat`: a set's size is only ever compared with the removal threshold, so counting stops there (a plain load first) -- the one giant
// component drew 94 k adds to one address per call.  INT_MAX where the sizes themselves are used (the literal labelling).
__global__ void k_groups_count(DParams prm, int* parent, int* size, int sat) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= prm.pool_n || !(prm.pool[id].flags & 1)) return;
    const int r = uf_find(parent, (int)id);
    parent[id] = r;
    // one giant component holds almost every patch: lanes that share the first active lane's root add once
    const int r0 = __builtin_amdgcn_readfirstlane(r);
    const unsigned long long same = __ballot(r == r0);
    if (r == r0) { if ((int)(threadIdx.x & 63u) == __ffsll((long long)same) - 1 && __atomic_load_n(&size[r0], __ATOMIC_RELAXED) < sat) atomicAdd(&size[r0], (int)__popcll(same)); }
    else if (__atomic_load_n(&size[r], __ATOMIC_RELAXED) < sat) atomicAdd(&size[r], 1);
}
__global__ void k_groups_kill(DParams prm, const int* parent, const int* size, int threshold, uint8_t* kill) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= prm.pool_n || !(prm.pool[id].flags & 1)) return;
    if (size[parent[id]] < threshold) kill[id] = 1;
}

// =================================================================== probes (single functions, batched)
__global__ __launch_bounds__(64) void k_probe(DParams prm, int op, int64_t n, const DPatch* __restrict__ in, const float* __restrict__ in_f,
                                              DPatch* __restrict__ out, float* __restrict__ out_f, int32_t* __restrict__ out_i) {
    __shared__ int s_scratch[192];
    extern __shared__ float s_texs[];
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    WaveCtx wc = make_wave_ctx(prm);
    const int tstride = prm.wsz;  // odd for 7x7 / 5x5 windows: the lane-per-pair reads of setRefImage fall in distinct banks
    if (op == 5) {  // MVS_PROBE_MATH
        if (wc.lane == 0) {
            const float x = in_f[i];
            out_f[5 * i] = pm_sinf(x); out_f[5 * i + 1] = pm_cosf(x); out_f[5 * i + 2] = pm_asinf(x); out_f[5 * i + 3] = pm_acosf(x);
            out_f[5 * i + 4] = pm_atanf(x);
        }
        return;
    }
    Cand c;
    load_cand(in + i, wc, c);
    if (op == 0) {  // MVS_PROBE_NCC
        const float ncc = compute_ncc(prm, wc, c.coord, c.normal, c.img, c.nimg);
        if (wc.lane == 0) out_f[i] = ncc;
    } else if (op == 1) {
        const int f = pre_process(prm, wc, s_scratch, c);
        store_cand(out + i, wc, c, 1, (int)i);
        if (wc.lane == 0) out_i[i] = f;
    } else if (op == 2) {
        refine_patch(prm, wc, c, 0u, 0u, (uint32_t)i, 0u);
        store_cand(out + i, wc, c, 1, (int)i);
    } else if (op == 3) {
        int f = post_process(prm, wc, s_scratch, s_texs, tstride, c);
        if (f == 0 && prm.depth >= 2 && prm.enable_check) {
            const CheckCtx cx{in, -1, -1, 0, s_scratch, nullptr};  // no staged ids in a probe (a null pointer here crashes clang 22)
            const int chk = check_patch(prm, wc, cx, c, s_texs, out_i + n);  // out_i[n]: overflow flag word
            if (chk < 0) { if (wc.lane == 0) atomicOr(out_i + n, 4); }       // no second tier in a probe
            else if (chk) f = -1;
        }
        store_cand(out + i, wc, c, 1, (int)i);
        if (wc.lane == 0) out_i[i] = f;
    } else if (op == 4) {
        RefineCtx rc;
        rc.center = c.coord; rc.ref = rli(c.img, 0);
        rc.ray = nrm4(sub4(c.coord, ld4((prm.views + rc.ref)->center)));
        rc.dscale = c.dscale; rc.ascale = prm.ascaleConst;
        float x[3];
        encode(prm, rc, c.coord, c.normal, x);
        double fv[4];
        cost_func4(prm, wc, rc, __shfl(c.img, wc.lane & 15), c.nimg, false, x[0], x[1], x[2], fv);
        if (wc.lane == 0) out_f[i] = (float)fv[0];
    }
}

// refinePatch alone (occupancy experiment / kernel benchmark): same arithmetic as probe op 2
__global__ __launch_bounds__(64, MVS_SWEEP_WAVES) void k_probe_refine(DParams prm, int64_t n, const DPatch* __restrict__ in, DPatch* __restrict__ out) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    WaveCtx wc = make_wave_ctx(prm);
    Cand c;
    load_cand(in + i, wc, c);
    refine_patch(prm, wc, c, 0u, 0u, (uint32_t)i, 0u);
    store_cand(out + i, wc, c, 1, (int)i);
}

// =================================================================== host-callable launchers
static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

void mvsk_rgb_to_rgba(const uint8_t* rgb, uint32_t* out, int64_t n, hipStream_t st) { hipLaunchKernelGGL(k_rgb_to_rgba, dim3(nblk(n, 256)), dim3(256), 0, st, rgb, out, n); }
void mvsk_rgba_to_rgb(const uint32_t* in, uint8_t* rgb, int64_t n, hipStream_t st) { hipLaunchKernelGGL(k_rgba_to_rgb, dim3(nblk(n, 256)), dim3(256), 0, st, in, rgb, n); }
void mvsk_pyr_down(const uint32_t* src, int pw, int ph, uint32_t* dst, int w, int h, hipStream_t st) {
    hipLaunchKernelGGL(k_pyr_down, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, st, src, pw, ph, dst, w, h);
}
void mvsk_mask_down(const uint8_t* src, int pw, int ph, uint8_t* dst, int w, int h, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_down, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, st, src, pw, ph, dst, w, h);
}
void mvsk_mask_binarise(uint8_t* m, int64_t n, hipStream_t st) { hipLaunchKernelGGL(k_mask_binarise, dim3(nblk(n, 256)), dim3(256), 0, st, m, n); }
void mvsk_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* tmp, hipStream_t st) { launch_exclusive_scan<int32_t, int32_t>(in, out, n, tmp, st); }
// the per-cell counts -> 64-bit list offsets; tmp as above, in elements of csr_off_t
void mvsk_exclusive_scan_off(const int32_t* in, csr_off_t* out, int64_t n, csr_off_t* tmp, hipStream_t st) { launch_exclusive_scan<int32_t, csr_off_t>(in, out, n, tmp, st); }
void mvsk_index_count(const DParams& prm, int32_t* cnt, int32_t* vcnt, unsigned long long* total, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_index_count, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, cnt, vcnt, total);
}
void mvsk_index_fill(const DParams& prm, int vgrid, const csr_off_t* start, int32_t* cursor, unsigned long long* ids, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_index_fill, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, vgrid, start, cursor, ids);
}
void mvsk_index_fill_direct(const DParams& prm, int vgrid, const csr_off_t* start, int32_t* cursor, int32_t* id32, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_index_fill_direct, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, vgrid, start, cursor, id32);
}
void mvsk_index_sort_trim(const DParams& prm, const csr_off_t* start, unsigned long long* ids, int do_trim, unsigned long long* trimmed, hipStream_t st) {
    hipLaunchKernelGGL(k_index_sort_trim, dim3(nblk(prm.total_cells, 256)), dim3(256), 0, st, prm, start, ids, do_trim, trimmed);
}
void mvsk_index_finalize(const DParams& prm, int vgrid, const csr_off_t* start, unsigned long long* ids, int32_t* id32, int32_t* cnt_alive, hipStream_t st) {
    hipLaunchKernelGGL(k_index_finalize, dim3(nblk(prm.total_cells, 256)), dim3(256), 0, st, prm, vgrid, start, ids, id32, cnt_alive);
}
void mvsk_index_pack(const DParams& prm, const csr_off_t* start, const csr_off_t* start2, const int32_t* cnt_alive, const unsigned long long* ids, const int32_t* id32_in,
                     ListKey* key, int32_t* id32_out, hipStream_t st) {
    hipLaunchKernelGGL(k_index_pack, dim3(nblk(prm.total_cells, 256)), dim3(256), 0, st, prm, start, start2, cnt_alive, ids, id32_in, key, id32_out);
}
void mvsk_depth_maps(const DParams& prm, unsigned long long* dp, const uint32_t* dirty, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_depth_maps, dim3(nblk(prm.pool_n, 64)), dim3(256), 0, st, prm, dp, dirty);
}
void mvsk_depth_mark_dirty(const DParams& prm, const uint8_t* kill, unsigned long long* dp, uint32_t* dirty, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_depth_mark_dirty, dim3(nblk(prm.pool_n, 64)), dim3(256), 0, st, prm, kill, dp, dirty);
}
void mvsk_best_ncc_map(const DParams& prm, int view, unsigned long long* best, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_best_ncc_map, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, view, best);
}
void mvsk_map_extract(const DParams& prm, int view, int kind, const unsigned long long* sel, float* depth, float* normal, int32_t* ids, int ncells, hipStream_t st) {
    hipLaunchKernelGGL(k_map_extract, dim3(nblk(ncells, 256)), dim3(256), 0, st, prm, view, kind, sel, depth, normal, ids, ncells);
}
void mvsk_fill_ncc(const DParams& prm, unsigned long long* evals, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_fill_ncc, dim3((unsigned)((prm.pool_n + 63) / 64)), dim3(64), MVS_FRAME_LDS_BYTES, st, prm, evals);
}
size_t mvsk_sweep_lds_bytes(const DParams& prm) {
    // setRefImage: the centred textures of MVS_LISTCAP views (9408 B at wsize 7 and 16 views) + one value per view pair;
    // in postProcess they lie behind the frame region (the evaluation that produces them publishes its frames there).
    // The 32- and 64-view builds keep one chunk of MVS_GRAM_CH views at a time and overlay the Gram matrix on it (mvs_device.cuh).
#if MVS_PAIR_MFMA
    const size_t tex_f = (size_t)MVS_GRAM_CH * 3 * prm.wsz, gram_f = (size_t)prm.gram_ld * prm.gram_ld;
    const size_t texs = MVS_FRAME1_LDS_BYTES + (tex_f > gram_f ? tex_f : gram_f) * sizeof(float);
#else
    const size_t ln = (size_t)prm.list_n;  // min(MVS_LISTCAP, nviews): a 12-view data set keeps 12 textures, not 16
    const size_t texs = MVS_FRAME1_LDS_BYTES + (ln * 3 * prm.wsz + ln * (ln - 1) / 2) * sizeof(float);
#endif
    const size_t chk = (size_t)MVS_CHECK_LDS_FLOATS * sizeof(float);                      // Optim::check hash set + rows
    const size_t need = texs > chk ? texs : chk;
    return need > (size_t)MVS_FRAME_LDS_BYTES ? need : (size_t)MVS_FRAME_LDS_BYTES;  // the frames + pivots of a refinement step
}
void mvsk_sweep(const DParams& prm, const SweepArgs& a, hipStream_t st) {
    const int64_t nloc = a.job_hi - a.job_lo;
    if (nloc <= 0) return;
    const int64_t chunk = (nloc + 7) / 8;
    // development knob: MVS_SWEEP_LDS_PAD=<bytes> raises the block's LDS allocation, i.e. lowers the waves per SIMD
    static const size_t pad = getenv("MVS_SWEEP_LDS_PAD") ? (size_t)atol(getenv("MVS_SWEEP_LDS_PAD")) : 0;
#if MVS_XCD_CHUNK > 0
    const int64_t nblocks = (nloc + 8 * MVS_XCD_CHUNK - 1) / (8 * MVS_XCD_CHUNK) * (8 * MVS_XCD_CHUNK);
#else
    const int64_t nblocks = chunk * 8;
#endif
    hipLaunchKernelGGL(k_sweep, dim3((unsigned)nblocks), dim3(64), mvsk_sweep_lds_bytes(prm) + pad, st, prm, a);
}
void mvsk_job_work(const DParams& prm, const SweepArgs& a, int mode, int shift, int32_t* work, hipStream_t st) {
    if (a.njobs > 0) hipLaunchKernelGGL(k_job_work, dim3(nblk(a.njobs, 256)), dim3(256), 0, st, prm, a, mode, shift, work);
}
void mvsk_job_cuts(const int32_t* scan, int64_t njobs, int n, int32_t* cuts, hipStream_t st) {
    if (njobs > 0) hipLaunchKernelGGL(k_job_cuts, dim3(nblk(njobs, 256)), dim3(256), 0, st, scan, njobs, n, cuts);
}
void mvsk_sweep_retry(const DParams& prm, const SweepArgs& a, int nretry, hipStream_t st) {
    if (nretry <= 0) return;
    hipLaunchKernelGGL(k_sweep_retry, dim3((unsigned)std::min(nretry, MVS_BIG_SLOTS)), dim3(64), mvsk_sweep_lds_bytes(prm), st, prm, a, nretry);
}
void mvsk_commit_count(const SweepArgs& a, int32_t* cnt, hipStream_t st) { hipLaunchKernelGGL(k_commit_count, dim3(nblk(a.njobs, 256)), dim3(256), 0, st, a, cnt); }
void mvsk_commit_copy(const SweepArgs& a, const int32_t* base, DPatch* dst, int64_t dst_cap, int32_t* per_view, int keep_key, hipStream_t st) {
    hipLaunchKernelGGL(k_commit_copy, dim3(nblk(a.njobs, 256)), dim3(256), 0, st, a, base, dst, dst_cap, per_view, keep_key);
}
void mvsk_kill_count(const uint8_t* kill, int64_t n, int32_t* cnt, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_kill_count, dim3(nblk(n, 256)), dim3(256), 0, st, kill, n, cnt); }
void mvsk_kill_export(const uint8_t* kill, int64_t n, const int32_t* base, int32_t* ids, int64_t cap, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_kill_export, dim3(nblk(n, 256)), dim3(256), 0, st, kill, n, base, ids, cap);
}
void mvsk_apply_kill_flags(DPatch* pool, uint8_t* kill, int64_t n, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_apply_kill_flags, dim3(nblk(n, 256)), dim3(256), 0, st, pool, kill, n); }
void mvsk_apply_kill_ids(DPatch* pool, const int32_t* ids, int64_t n, int64_t pool_n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_apply_kill_ids, dim3(nblk(n, 256)), dim3(256), 0, st, pool, ids, n, pool_n);
}
void mvsk_append_records(DPatch* pool, int64_t pool_n, const DPatch* recs, int64_t n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_append_records, dim3(nblk(n, 256)), dim3(256), 0, st, pool, pool_n, recs, n);
}
void mvsk_alive_count(const DPatch* pool, int64_t n, int32_t* cnt, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_alive_count, dim3(nblk(n, 256)), dim3(256), 0, st, pool, n, cnt); }
void mvsk_alive_gather(const DPatch* pool, int64_t n, const int32_t* base, DPatch* out, int64_t cap, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_alive_gather, dim3(nblk(n, 256)), dim3(256), 0, st, pool, n, base, out, cap);
}
void mvsk_filter_vimages(const DParams& prm, int additive, int64_t first, int64_t last, const uint32_t* dirty, hipStream_t st) {
    if (last <= first) return;
    const int gl = std::max(prm.nviews <= 16 ? 16 : (prm.nviews <= 32 ? 32 : 64), (int)MVS_LISTCAP);
    const unsigned nb = (unsigned)((last - first + 64 / gl - 1) / (64 / gl));
    if (gl == 16) hipLaunchKernelGGL(k_filter_vimages<16>, dim3(nb), dim3(64), 0, st, prm, additive, first, last, dirty);
    else if (gl == 32) hipLaunchKernelGGL(k_filter_vimages<32>, dim3(nb), dim3(64), 0, st, prm, additive, first, last, dirty);
    else hipLaunchKernelGGL(k_filter_vimages<64>, dim3(nb), dim3(64), 0, st, prm, additive, first, last, dirty);
}
void mvsk_geo_pack(const DPatch* pool, int64_t n, float4* geo, uint8_t* ref, int* bad, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_geo_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pool, n, geo, ref, bad);
}
void mvsk_filter_outside(const DParams& prm, uint8_t* kill, int64_t first, int64_t last, hipStream_t st) {
    if (last > first) hipLaunchKernelGGL(k_filter_outside, dim3((unsigned)(last - first)), dim3(64), 0, st, prm, kill, first);
}
// Filter::filterExact needs the LDS of setRefImage only (frames + the textures of a list), not Optim::check's id set: 7.9 KB instead of
// 9.5 KB per wave at 12 views -- the kernel waits on dependent gathers (vector ALU busy 37 %), so resident waves are what it lacks
size_t mvsk_texs_lds_bytes(const DParams& prm) {
#if MVS_PAIR_MFMA
    const size_t tex_f = (size_t)MVS_GRAM_CH * 3 * prm.wsz, gram_f = (size_t)prm.gram_ld * prm.gram_ld;
    const size_t texs = MVS_FRAME1_LDS_BYTES + (tex_f > gram_f ? tex_f : gram_f) * sizeof(float);
#else
    const size_t ln = (size_t)prm.list_n;
    const size_t texs = MVS_FRAME1_LDS_BYTES + (ln * 3 * prm.wsz + ln * (ln - 1) / 2) * sizeof(float);
#endif
    return texs > (size_t)MVS_FRAME_LDS_BYTES ? texs : (size_t)MVS_FRAME_LDS_BYTES;
}
void mvsk_filter_exact(const DParams& prm, uint8_t* kill, unsigned long long* evals, unsigned long long* stage, int64_t first, int64_t last, hipStream_t st) {
    if (last > first) hipLaunchKernelGGL(k_filter_exact, dim3((unsigned)((last - first + MVS_FE_PATCHES - 1) / MVS_FE_PATCHES)), dim3(64), mvsk_texs_lds_bytes(prm), st, prm, kill, evals, stage, first, last);
}
void mvsk_filter_neighbor(const DParams& prm, uint8_t* kill, int32_t* retry, int32_t* nretry, int32_t* overflow, unsigned long long* stats, int64_t first, int64_t last, hipStream_t st) {
    if (last <= first) return;
    hipLaunchKernelGGL((k_filter_neighbor<MVS_FILTER_HASH_CAP, MVS_FILTER_ROW_CAP>), dim3((unsigned)(last - first)), dim3(64),
                       (size_t)(MVS_SET_LDS_FLOATS(MVS_FILTER_HASH_CAP, MVS_FILTER_ROW_CAP) + 64) * sizeof(float), st, prm, kill, (const int32_t*)nullptr, 0, retry, nretry, overflow, stats, first);
}
void mvsk_filter_neighbor_retry(const DParams& prm, uint8_t* kill, const int32_t* todo, int32_t ntodo, int32_t* overflow, unsigned long long* stats, hipStream_t st) {
    if (ntodo <= 0) return;
    hipLaunchKernelGGL((k_filter_neighbor<MVS_FILTER2_HASH_CAP, MVS_FILTER2_ROW_CAP>), dim3((unsigned)ntodo), dim3(64),
                       (size_t)MVS_SET_LDS_FLOATS(MVS_FILTER2_HASH_CAP, MVS_FILTER2_ROW_CAP) * sizeof(float), st, prm, kill, todo, ntodo, (int32_t*)nullptr, (int32_t*)nullptr, overflow, stats, (int64_t)0);
}
void mvsk_groups(const DParams& prm, int* parent, int* size, int threshold, uint8_t* kill, hipStream_t st) {
    if (prm.pool_n <= 0) return;
    hipLaunchKernelGGL(k_groups_init, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, parent, size, prm.pool_n);
    hipLaunchKernelGGL(k_groups_edges<0>, dim3(nblk(prm.pool_n, 4)), dim3(256), 0, st, prm, parent, (int2*)nullptr, (int*)nullptr, 0);
    hipLaunchKernelGGL(k_groups_count, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, parent, size, threshold);
    hipLaunchKernelGGL(k_groups_kill, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, parent, size, threshold, kill);
}
// the literal labelling in three steps (mvs_engine.cpp does the middle one on the host): sets joined by edges that run both ways and
// the edges left between different sets; then, with the sizes of the literal groups written over the sets' sizes, the removal
void mvsk_groups_literal_edges(const DParams& prm, int* parent, int* size, int* edges2, int* nedges, int cap, hipStream_t st) {
    if (prm.pool_n <= 0) return;
    hipLaunchKernelGGL(k_groups_init, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, parent, size, prm.pool_n);
    hipLaunchKernelGGL(k_groups_edges<1>, dim3(nblk(prm.pool_n, 4)), dim3(256), 0, st, prm, parent, (int2*)nullptr, (int*)nullptr, 0);
    hipLaunchKernelGGL(k_groups_count, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, parent, size, INT_MAX);  // also flattens parent[] to the roots
    hipLaunchKernelGGL(k_groups_edges<2>, dim3(nblk(prm.pool_n, 4)), dim3(256), 0, st, prm, parent, reinterpret_cast<int2*>(edges2), nedges, cap);
}
__global__ void k_gather_i32(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ void k_scatter_i32(int32_t* __restrict__ dst, const int32_t* __restrict__ idx, const int32_t* __restrict__ val, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}
void mvsk_gather_i32(const int32_t* src, const int32_t* idx, int32_t* out, int64_t n, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_gather_i32, dim3(nblk(n, 256)), dim3(256), 0, st, src, idx, out, n); }
void mvsk_scatter_i32(int32_t* dst, const int32_t* idx, const int32_t* val, int64_t n, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_scatter_i32, dim3(nblk(n, 256)), dim3(256), 0, st, dst, idx, val, n); }
void mvsk_groups_kill(const DParams& prm, const int* parent, const int* size, int threshold, uint8_t* kill, hipStream_t st) {
    if (prm.pool_n > 0) hipLaunchKernelGGL(k_groups_kill, dim3(nblk(prm.pool_n, 256)), dim3(256), 0, st, prm, parent, size, threshold, kill);
}
void mvsk_probe(const DParams& prm, int op, int64_t n, const DPatch* in, const float* in_f, DPatch* out, float* out_f, int32_t* out_i, hipStream_t st) {
    if (n > 0 && op == 2) { hipLaunchKernelGGL(k_probe_refine, dim3((unsigned)n), dim3(64), MVS_FRAME_LDS_BYTES, st, prm, n, in, out); return; }
    if (n > 0) hipLaunchKernelGGL(k_probe, dim3((unsigned)n), dim3(64), mvsk_sweep_lds_bytes(prm), st, prm, op, n, in, in_f, out, out_f, out_i);
}
