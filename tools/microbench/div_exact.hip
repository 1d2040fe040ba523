// tools/microbench/div_exact.hip -- does the short reciprocal / division of mvs_device.cuh (rcp_rn, div_rn) return the IEEE-754
// round-to-nearest result?  EXHAUSTIVE for the reciprocal: every float bit pattern with a biased exponent in [7, 247] (|x| in
// 2^-120 .. 2^120, both signs: 2 x 241 x 2^23 inputs) against the compiler's IEEE division 1.0f / x.  The division: Markstein's
// theorem (q' = RN(q + (a - b q) y) is a / b correctly rounded when y = RN(1 / b) and q is a faithful quotient, barring over- and
// underflow) makes it follow from the reciprocal; checked here on 2^33 pseudo-random pairs with exponents in [-40, 40] and on
// mantissa patterns that are hard for division (all ones, one, alternating).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o div_exact div_exact.hip && ./div_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_rn(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_rn(float a, float b) {
    const float y = rcp_rn(b);
    const float q = a * y;
    const float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, y, q);
}
__global__ void k_rcp(unsigned long long* bad, unsigned* first_bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 2 signs x 241 exponents x 2^23 mantissas
    const uint32_t mant = (uint32_t)(i & 0x7fffffu);
    const uint32_t ex = 7u + (uint32_t)((i >> 23) % 241u);
    const uint32_t sg = (uint32_t)((i >> 23) / 241u) & 1u;
    const uint32_t bits = (sg << 31) | (ex << 23) | mant;
    const float x = __uint_as_float(bits);
    volatile float one = 1.0f;
    const float want = one / x;
    const float got = rcp_rn(x);
    if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first_bad, bits); }
}
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
__global__ void k_div(unsigned long long* bad, unsigned* ex_a, unsigned* ex_b, int mode, uint64_t first) {
    const uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ha = mix((uint32_t)i * 2654435761u + 12345u + (uint32_t)(i >> 32)), hb = mix(ha ^ 0x9e3779b9u ^ (uint32_t)(i >> 7));
    uint32_t ma = ha & 0x7fffffu, mb = hb & 0x7fffffu;
    if (mode == 1) mb = 0x7fffffu - (hb & 0xffu);   // divisors just below a power of two
    if (mode == 2) { ma = 0x7fffffu - (ha & 0xfu); mb = hb & 0xffu; }
    if (mode == 3) { ma = (ha & 1u) ? 0x555555u : 0x2aaaaau; }
    const uint32_t ea = 127u - 40u + (mix(ha) % 81u), eb = 127u - 40u + (mix(hb) % 81u);
    const float a = __uint_as_float(((ha >> 31) << 31) | (ea << 23) | ma), b = __uint_as_float(((hb >> 31) << 31) | (eb << 23) | mb);
    const float want = a / b, got = div_rn(a, b);
    if (__float_as_uint(want) != __float_as_uint(got)) { if (atomicAdd(bad, 1ull) == 0ull) { *ex_a = __float_as_uint(a); *ex_b = __float_as_uint(b); } }
}
int main() {
    unsigned long long* bad; unsigned* fb;
    hipMalloc(&bad, 16); hipMalloc(&fb, 16);
    unsigned long long hbad = 0; unsigned hfb[2] = {0xffffffffu, 0};
    hipMemcpy(bad, &hbad, 8, hipMemcpyHostToDevice); hipMemcpy(fb, hfb, 8, hipMemcpyHostToDevice);
    const uint64_t n = 2ull * 241ull * (1ull << 23);
    hipLaunchKernelGGL(k_rcp, dim3((unsigned)(n / 256)), dim3(256), 0, 0, bad, fb);
    hipDeviceSynchronize();
    hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hfb, fb, 8, hipMemcpyDeviceToHost);
    printf("rcp_rn: %llu inputs (all floats with 2^-120 <= |x| < 2^121), %llu differ from 1.0f / x; smallest differing pattern 0x%08x\n", (unsigned long long)n, hbad, hfb[0]);
    unsigned long long total_bad = hbad;
    for (int mode = 0; mode < 4; ++mode) {
        hbad = 0; hipMemcpy(bad, &hbad, 8, hipMemcpyHostToDevice);
        const uint64_t nd = mode == 0 ? (1ull << 33) : (1ull << 30);
        for (uint64_t done = 0; done < nd; done += (1ull << 30))
            hipLaunchKernelGGL(k_div, dim3((unsigned)((1ull << 30) / 256)), dim3(256), 0, 0, bad, fb, fb + 1, mode, done);
        hipDeviceSynchronize();
        hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hfb, fb, 8, hipMemcpyDeviceToHost);
        printf("div_rn mode %d: %llu pairs, %llu differ from a / b (first: a 0x%08x b 0x%08x)\n", mode, (unsigned long long)nd, hbad, hfb[0], hfb[1]);
        total_bad += hbad;
    }
    printf(total_bad ? "FAIL\n" : "PASS\n");
    return total_bad ? 1 : 0;
}
