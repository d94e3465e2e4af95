// microbench_gemv.hip — floors for the batch-1 decode GEMV chain on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_gemv.hip -o /tmp/mb && /tmp/mb
// Times chains of DEPENDENT launches (hipGraph replay), one decoder layer = [qkv 4096x1024,
// o 1024x2048, gate/up 2x3072x1024, down 1024x3072] bf16, cycling over NL distinct layers
// (NL=28: 880 MB, HBM-streamed; NL=5: 157 MB, Infinity-Cache resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__global__ void k_empty(float* x) { if (threadIdx.x == 9999) x[0] = 1.f; }

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// MODE 0: full GEMV (x loads + fma + reduce + store); 1: weight loads only (xor-reduced so they are not dead)
template <int NCH, int RW, int MODE, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void k_gemv_t(const bf16_t* __restrict__ W, const float* __restrict__ x, float* __restrict__ out, int N) {
    constexpr int K = NCH * 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * (blockDim.x >> 6) + wave) * RW;
    u32x4 w[RW][NCH];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)n * K + c * 512 + lane * 8);
            w[r][c] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    }
    if (MODE == 1) {
        unsigned acc = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc ^= w[r][c].x ^ w[r][c].y ^ w[r][c].z ^ w[r][c].w;
        if (acc == 0x12345678u) out[n0] = 1.f;
        return;
    }
    float xv[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(x + c * 512 + lane * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + c * 512 + lane * 8 + 4);
        xv[c][0] = a.x; xv[c][1] = a.y; xv[c][2] = a.z; xv[c][3] = a.w; xv[c][4] = b.x; xv[c][5] = b.y; xv[c][6] = b.z; xv[c][7] = b.w;
    }
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned wu[4] = { w[r][c].x, w[r][c].y, w[r][c].z, w[r][c].w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = fmaf(xv[c][2 * j], __uint_as_float(wu[j] << 16), s);
                s = fmaf(xv[c][2 * j + 1], __uint_as_float(wu[j] & 0xFFFF0000u), s);
            }
        }
        s = wave_sum(s);
        if (lane == r) mine = s;
    }
    if (lane < RW && n0 + lane < N) out[n0 + lane] = mine * 1e-3f;
}

struct Layer { bf16_t *qkv, *o, *gu, *down; };

// MODE 2 ("k-split"): one 512-wide K chunk per wave; the KS=NCH waves of a row group combine through LDS.
// Workgroup = NCH k-waves x RG row groups, each wave RW rows.  Wide and shallow: ~16-32 waves per CU.
template <int NCH, int RW, int RG, bool NT>
__global__ __launch_bounds__(64 * NCH * RG) void k_gemv_ks(const bf16_t* __restrict__ W, const float* __restrict__ x, float* __restrict__ out, int N) {
    constexpr int K = NCH * 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ks = wave % NCH, rg = wave / NCH;
    const int n0 = (blockIdx.x * RG + rg) * RW;
    __shared__ float part[RG][NCH][RW];
    u32x4 w[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)n * K + ks * 512 + lane * 8);
        w[r] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    const float4 a = *reinterpret_cast<const float4*>(x + ks * 512 + lane * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + ks * 512 + lane * 8 + 4);
    const float xv[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const unsigned wu[4] = { w[r].x, w[r].y, w[r].z, w[r].w };
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s = fmaf(xv[2 * j], __uint_as_float(wu[j] << 16), s);
            s = fmaf(xv[2 * j + 1], __uint_as_float(wu[j] & 0xFFFF0000u), s);
        }
        s = wave_sum(s);
        if (lane == 0) part[rg][ks][r] = s;
    }
    __syncthreads();
    if (ks == 0 && lane < RW && n0 + lane < N) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) s += part[rg][c][lane];
        out[n0 + lane] = s * 1e-3f;
    }
}
template <bool NT, int RWQ, int RGQ, int RGO, int RWG, int RGG, int RGD>
static void record_layer_ks(const Layer& L, float* x, float* t1, float* t2, hipStream_t s) {
    hipLaunchKernelGGL((k_gemv_ks<2, RWQ, RGQ, NT>), dim3(4096 / (RWQ * RGQ)), dim3(64 * 2 * RGQ), 0, s, L.qkv, x, t1, 4096);
    hipLaunchKernelGGL((k_gemv_ks<4, 1, RGO, NT>), dim3(1024 / RGO), dim3(64 * 4 * RGO), 0, s, L.o, t1, t2, 1024);
    hipLaunchKernelGGL((k_gemv_ks<2, RWG, RGG, NT>), dim3(6144 / (RWG * RGG)), dim3(64 * 2 * RGG), 0, s, L.gu, t2, t1, 6144);
    hipLaunchKernelGGL((k_gemv_ks<6, 1, RGD, NT>), dim3(1024 / RGD), dim3(64 * 6 * RGD), 0, s, L.down, t1, x, 1024);
}


template <int MODE, bool NT, int TPB>
static void record_layer(const Layer& L, float* x, float* t1, float* t2, hipStream_t s) {
    constexpr int WPB = TPB / 64;
    // qkv: N=4096 K=1024 RW=4 ; o: N=1024 K=2048 RW=1 ; gate/up: N=6144 K=1024 RW=6?? use RW=3 twice-size N ; down: N=1024 K=3072 RW=1
    hipLaunchKernelGGL((k_gemv_t<2, 4, MODE, NT, TPB>), dim3(4096 / (WPB * 4)), dim3(TPB), 0, s, L.qkv, x, t1, 4096);
    hipLaunchKernelGGL((k_gemv_t<4, 1, MODE, NT, TPB>), dim3(1024 / WPB), dim3(TPB), 0, s, L.o, t1, t2, 1024);
    hipLaunchKernelGGL((k_gemv_t<2, 6, MODE, NT, TPB>), dim3(6144 / (WPB * 6)), dim3(TPB), 0, s, L.gu, t2, t1, 6144);
    hipLaunchKernelGGL((k_gemv_t<6, 1, MODE, NT, TPB>), dim3(1024 / WPB), dim3(TPB), 0, s, L.down, t1, x, 1024);
}

template <typename F>
static float time_graph(F rec, hipStream_t s, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rec();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NLMAX = 28;
    std::vector<Layer> L(NLMAX);
    for (auto& l : L) {
        CK(hipMalloc((void**)&l.qkv, (size_t)4096 * 1024 * 2)); CK(hipMalloc((void**)&l.o, (size_t)1024 * 2048 * 2));
        CK(hipMalloc((void**)&l.gu, (size_t)6144 * 1024 * 2)); CK(hipMalloc((void**)&l.down, (size_t)1024 * 3072 * 2));
        CK(hipMemset(l.qkv, 0x3c, (size_t)4096 * 1024 * 2)); CK(hipMemset(l.o, 0x3c, (size_t)1024 * 2048 * 2));
        CK(hipMemset(l.gu, 0x3c, (size_t)6144 * 1024 * 2)); CK(hipMemset(l.down, 0x3c, (size_t)1024 * 3072 * 2));
    }
    float *x, *t1, *t2;
    CK(hipMalloc((void**)&x, 65536)); CK(hipMalloc((void**)&t1, 65536)); CK(hipMalloc((void**)&t2, 65536));
    CK(hipMemset(x, 0, 65536)); CK(hipMemset(t1, 0, 65536)); CK(hipMemset(t2, 0, 65536));
    const double layer_mb = (4096.0 * 1024 + 1024 * 2048 + 6144 * 1024 + 1024 * 3072) * 2 / 1e6;

    // floor: chains of empty kernels
    for (int wg : { 1, 256, 1024 }) {
        float ms = time_graph([&] { for (int i = 0; i < 112; ++i) hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, s, x); }, s, 50);
        printf("empty chain, %4d WGs x256: %.2f us / launch\n", wg, ms * 1e3 / 112);
    }
    for (int nl : { 28, 5 }) {
        auto run = [&](const char* name, auto fn) {
            float ms = time_graph([&] { for (int r = 0; r < 28 / nl + (28 % nl ? 1 : 0); ++r) for (int i = 0; i < nl; ++i) fn(L[i]); }, s, 40);
            const int layers = (28 / nl + (28 % nl ? 1 : 0)) * nl;
            printf("NL=%2d %-34s %.2f us/layer  %.2f us/kernel  %.0f GB/s\n", nl, name, ms * 1e3 / layers, ms * 1e3 / layers / 4, layer_mb / (ms / layers) );
        };
        run("full gemv, 256thr, default loads", [&](const Layer& l) { record_layer<0, false, 256>(l, x, t1, t2, s); });
        run("full gemv, 256thr, nontemporal", [&](const Layer& l) { record_layer<0, true, 256>(l, x, t1, t2, s); });
        run("loads only, 256thr, default", [&](const Layer& l) { record_layer<1, false, 256>(l, x, t1, t2, s); });
        run("loads only, 256thr, nontemporal", [&](const Layer& l) { record_layer<1, true, 256>(l, x, t1, t2, s); });
        run("ksplit RW1 (qkv rg2,o rg1,gu rg2,d rg1)", [&](const Layer& l) { record_layer_ks<true, 1, 2, 1, 1, 2, 1>(l, x, t1, t2, s); });
        run("ksplit RW2 (qkv rg2,o rg1,gu rg2,d rg1)", [&](const Layer& l) { record_layer_ks<true, 2, 2, 1, 2, 2, 1>(l, x, t1, t2, s); });
        run("ksplit RW2 bigger WGs (rg4,2,4,1)", [&](const Layer& l) { record_layer_ks<true, 2, 4, 2, 2, 4, 1>(l, x, t1, t2, s); });
        run("ksplit RW4 (rg2,o rg2,gu rg2,d rg1)", [&](const Layer& l) { record_layer_ks<true, 4, 2, 2, 4, 2, 1>(l, x, t1, t2, s); });
        run("ksplit RW1 default loads", [&](const Layer& l) { record_layer_ks<false, 1, 2, 1, 1, 2, 1>(l, x, t1, t2, s); });
    }
    return 0;
}
