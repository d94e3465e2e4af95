// minimal PMC probe: one streaming-read kernel of a known byte count (256 MiB), default stream, no graphs
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const uint4* p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567u) out[0] = acc;
}
int main() {
    const size_t bytes = 256u << 20;
    uint4* p; unsigned* o;
    hipMalloc((void**)&p, bytes); hipMalloc((void**)&o, 4); hipMemset(p, 1, bytes);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, p, bytes / 16, o);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
