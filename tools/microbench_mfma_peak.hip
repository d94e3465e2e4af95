// microbench_mfma_peak.hip — what does v_mfma_f32_32x32x16_f16 sustain on this chip, and at what shader clock?
// The codec decoder's roofline is priced against the 2.5 PFLOP/s dense 16-bit peak (= 32 cycles per instruction per SIMD at 2.4 GHz).
// This measures the ceiling a kernel can actually reach over tens of milliseconds: every CU, W waves per SIMD, register operands only,
// 4 independent accumulator blocks per wave; s_memtime (shader clock) against s_memrealtime (100 MHz) gives the clock under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(float* out, unsigned long long* stamps, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* st;
    CK(hipMalloc((void**)&out, 4096)); CK(hipMalloc((void**)&st, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    for (int wps = 1; wps <= 2; ++wps) {
        for (int iters : {20000, 200000, 1000000}) {
            const int grid = cus * wps;   // one 256-thread workgroup = one wave per SIMD of a CU
            hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, st, 1000);   // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, st, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
            const double flop = (double)grid * 4 /*waves*/ * iters * 4.0 * 32 * 32 * 16 * 2;
            const double mhz = (double)h[0] / ((double)h[1] / 100.0);   // shader ticks per microsecond of the 100 MHz clock
            printf("%d CUs, %d wave(s) per SIMD, %8d x 4 MFMAs per wave: %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)  shader clock %.0f MHz  cycles per MFMA per SIMD %.1f\n",
                   cus, wps, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0, mhz, (double)h[0] / ((double)iters * 4 * wps));
        }
    }
    return 0;
}
