// Issue cost of the vector instructions k_sweep is made of, on gfx950: each kernel runs ITER x 32 independent copies of one
// instruction (8 accumulators, 4 rounds) per wave, 1 or 3 waves per SIMD on every CU; the cost is reported relative to
// v_fma_f32 (4 cycles per wave64 instruction).  Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(name, decl, init, body, sink)                                         \
    __global__ void name(float* out) {                                               \
        decl;                                                                        \
        init;                                                                        \
        for (int it = 0; it < ITER; ++it) {                                          \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                          \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) { body; }              \
            }                                                                        \
        }                                                                            \
        float s = 0.f;                                                               \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) s += sink;                     \
        if (s == 123.456f) out[threadIdx.x] = s;                                     \
    }

#define FINIT float a[8]; float b = threadIdx.x * 1e-9f + 1.0f, c = 1e-9f
#define FSET  _Pragma("unroll") for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k
KERNEL(k_fma, FINIT, FSET, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)), a[k])
KERNEL(k_add, FINIT, FSET, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)), a[k])
KERNEL(k_mul, FINIT, FSET, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)), a[k])
KERNEL(k_fract, FINIT, FSET, asm volatile("v_fract_f32 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_floor, FINIT, FSET, asm volatile("v_floor_f32 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_cvt_ub0, FINIT, FSET, asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_cvt_ub3, FINIT, FSET, asm volatile("v_cvt_f32_ubyte3 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_cvt_i32, FINIT, FSET, asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_rcp, FINIT, FSET, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_sqrt, FINIT, FSET, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_mul_lo, FINIT, FSET, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)), a[k])
KERNEL(k_mad_u32_u24, FINIT, FSET, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)), a[k])
KERNEL(k_lshl_add, FINIT, FSET, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[k]) : "v"(b)), a[k])
KERNEL(k_add_dpp, FINIT, FSET, asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k])), a[k])
KERNEL(k_mov_dpp, FINIT, FSET, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k])), a[k])
KERNEL(k_readlane, FINIT; int sg = 0, FSET, asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[k])), a[k] + sg)
KERNEL(k_bperm, FINIT; float idx = (threadIdx.x ^ 1) * 4, FSET, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[k]) : "v"(idx)), a[k])
KERNEL(k_fmamix, FINIT, FSET, asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(a[k]) : "v"(b), "v"(c)), a[k])

#define PINIT f2 a[8]; f2 b = {threadIdx.x * 1e-9f + 1.0f, 1.0f}, c = {1e-9f, 2e-9f}
#define PSET  _Pragma("unroll") for (int k = 0; k < 8; ++k) a[k] = f2{(float)threadIdx.x, (float)k}
KERNEL(k_pk_fma, PINIT, PSET, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)), a[k].x + a[k].y)
KERNEL(k_pk_mul, PINIT, PSET, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)), a[k].x + a[k].y)
KERNEL(k_pk_add, PINIT, PSET, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)), a[k].x + a[k].y)

#define DINIT double a[8]; double b = threadIdx.x * 1e-9 + 1.0, c = 1e-9
#define DSET  _Pragma("unroll") for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k
KERNEL(k_fma_f64, DINIT, DSET, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)), (float)a[k])
KERNEL(k_add_f64, DINIT, DSET, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c)), (float)a[k])
KERNEL(k_mad_u64_u32, DINIT; unsigned m = threadIdx.x, DSET, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[k]) : "v"(m) : "vcc"), (float)a[k])
KERNEL(k_lshl_add_u64, DINIT, DSET, asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[k]) : "v"(b)), (float)a[k])

struct K { const char* name; void (*fn)(float*); };
#define E(n) {#n, n}
static K ks[] = {E(k_fma), E(k_add), E(k_mul), E(k_pk_fma), E(k_pk_mul), E(k_pk_add), E(k_fract), E(k_floor), E(k_cvt_ub0), E(k_cvt_ub3),
                 E(k_cvt_i32), E(k_rcp), E(k_sqrt), E(k_mul_lo), E(k_mad_u32_u24), E(k_lshl_add), E(k_add_dpp), E(k_mov_dpp), E(k_readlane),
                 E(k_bperm), E(k_fmamix), E(k_fma_f64), E(k_add_f64), E(k_mad_u64_u32), E(k_lshl_add_u64)};

int main() {
    float* out;
    if (hipMalloc(&out, 4096) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base[2] = {0, 0};
    for (int wi = 0; wi < 2; ++wi) {
        const int waves = wi == 0 ? 1 : 3;
        for (auto& k : ks) {
            const int blocks = cus * 4 * waves;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double per = best * 1e-3 / ((double)ITER * 32 * waves);  // seconds per instruction per SIMD
            if (k.fn == k_fma) base[wi] = per;
            printf("%d wave(s)/SIMD %-16s %8.3f ms  %6.2f cycles (v_fma_f32 = 4)  %6.2f at %d kHz\n", waves, k.name, best, 4.0 * per / base[wi], per * prop.clockRate * 1e3, prop.clockRate);
        }
    }
    return 0;
}
