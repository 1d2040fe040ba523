// tools/microbench/run_lookup.hip: the run lookup of findNeighbors' row walk (mvs_check.cuh, MK) on its own -- 64 runs of random lengths
// laid end to end, the run of every position by marks + running maximum against a plain search.  No address is dereferenced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ int wave_max_scan(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xc, 0xf, false));
    return x;
}
__global__ __launch_bounds__(64) void k(const int* lens, unsigned long long* bad, int* first_bad) {
    __shared__ int marks[64];
    const int lane = threadIdx.x;
    const int len = lens[blockIdx.x * 64 + lane];
    int P = len;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(P, d); if (lane >= d) P += o; }
    const int total = __builtin_amdgcn_readlane(P, 63);
    P -= len;
    int carry = 0;
    unsigned long long nbad = 0;
    for (int k0 = 0; k0 < total; k0 += 256) {
        marks[lane] = -1;
        const int pm = P - k0;
        if (len > 0 && (unsigned)pm < 256u) reinterpret_cast<signed char*>(marks)[(pm & 63) * 4 + (pm >> 6)] = (signed char)lane;
        __syncthreads();
        const int m4 = marks[lane];
        int run[4] = {(m4 << 24) >> 24, (m4 << 16) >> 24, (m4 << 8) >> 24, m4 >> 24};
#pragma unroll
        for (int q = 0; q < 4; ++q) run[q] = wave_max_scan(run[q]);
        run[0] = max(run[0], carry);
#pragma unroll
        for (int q = 1; q < 4; ++q) run[q] = max(run[q], __builtin_amdgcn_readlane(run[q - 1], 63));
        carry = __builtin_amdgcn_readlane(run[3], 63);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = k0 + 64 * q + lane;
            const int pl = __builtin_amdgcn_ds_bpermute(run[q] << 2, P);
            int lo = 0;
            for (int step = 32; step >= 1; step >>= 1) { const int pc = __shfl(P, lo + step); if (pc <= kk) lo += step; }
            const int pref = __shfl(P, lo);
            if (kk < total && pl != pref) { ++nbad; atomicMin(first_bad, blockIdx.x); }
        }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const int nb = 1 << 16;
    std::vector<int> h(nb * 64);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int b = 0; b < nb; ++b)
        for (int l = 0; l < 64; ++l) {
            int v = (rnd() % 10 < 3) ? 0 : (int)(rnd() % 40);
            if (b % 7 == 0 && l == (b / 7) % 64) v = 300 + (int)(rnd() % 400);
            if (b % 11 == 0 && l >= 50) v = 0;
            h[b * 64 + l] = v;
        }
    int* d; unsigned long long* bad; int* fb;
    hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 8); int big = 1 << 30; hipMemcpy(fb, &big, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, bad, fb);
    unsigned long long hb = 1; int hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
    printf("run lookup: %d waves of 64 runs, %llu positions differ from the search (first wave %d)\n%s\n", nb, hb, hb ? hf : -1, hb ? "FAIL" : "PASS");
    return hb ? 1 : 0;
}
