// microbench_clock.hip — what shader clock do tiny dependent kernels actually run at?
// Chain: 30 x [streaming GEMV-like kernel] + 1 single-wave ALU kernel that stamps s_memtime (shader
// clock) and s_memrealtime (100 MHz), replayed from a hipGraph like the decode step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ w, float* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * 256) { u32x4 v = w[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = 1.f;
}
// ITER dependent fmas per lane, 1 wave; stamps[0..3] = memtime start/end, realtime start/end
__global__ __launch_bounds__(64) void k_alu(float* out, unsigned long long* stamps, int iters, const float* in) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = in[threadIdx.x];   // one global load
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0000001f, 0.5f);
    unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; stamps[2] = t2; stamps[3] = r0; stamps[4] = r2; }
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t bytes = 8u << 20;
    u32x4* w; CK(hipMalloc((void**)&w, bytes * 32)); CK(hipMemset(w, 1, bytes * 32));
    float *out, *in; CK(hipMalloc((void**)&out, 4096)); CK(hipMalloc((void**)&in, 4096)); CK(hipMemset(in, 0, 4096));
    unsigned long long* st; CK(hipMalloc((void**)&st, 64));
    for (int nstream : { 0, 30 }) for (int iters : { 1000, 20000 }) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int rep = 0; rep < 8; ++rep) {
            for (int i = 0; i < nstream; ++i) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, s, w + (size_t)((rep * 30 + i) % 32) * (bytes / 16), out + 64, bytes / 16);
            hipLaunchKernelGGL(k_alu, dim3(1), dim3(64), 0, s, out, st, iters, in);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[5]; CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
        double cyc = (double)(h[2] - h[0]), real_us = (double)(h[4] - h[3]) / 100.0;
        printf("streams/alu=%2d iters=%5d: graph %.2f us per node | alu kernel: %.0f cycles in %.2f us -> %.0f MHz; first load %.0f cycles; %.2f cycles/fma\n",
               nstream, iters, ms * 1e3 / 100 / (8 * (nstream + 1)), cyc, real_us, cyc / real_us, (double)(h[1] - h[0]), (double)(h[2] - h[1]) / iters);
    }
    return 0;
}
