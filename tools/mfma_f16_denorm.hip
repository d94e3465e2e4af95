// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (decides whether an fp16 hi/lo split can carry 22 bits for small values)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row][k]: lane = row (+32 for k>=8); put a_val at A[row=lane][k=0] for lanes<32, b_val at B[k=0][col=lane]
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f16v c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[][2] = { { 1.0f, 1.0f }, { 9.5367431640625e-07f /*2^-20 subnormal*/, 1024.f }, { 3.0517578125e-05f /*2^-15 subnormal*/, 2.f },
                              { 5.9604644775390625e-08f /*2^-24 smallest subnormal*/, 1024.f }, { 6.103515625e-05f /*2^-14 min normal*/, 1.f } };
    for (auto& v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v[0], v[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%.10g b=%g -> mfma %.10g (exact %.10g)\n", v[0], v[1], h, v[0] * v[1]);
    }
    return 0;
}
