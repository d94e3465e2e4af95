// pk_probe.hip — a 70-line kernel, outside the product, beside the product's vocoder: does packed fp32 return wrong sums?
// Built as a shared library (tools/pk_probe_beside_vocoder.py drives it through ctypes while a q3tts engine runs batched vocoder jobs
// from another host thread).  The kernel is the loop structure of the vocoder's last conv (8 lanes per input row, 12 channels x 8 taps
// of weights in registers, 8 passes of 32 rows, DPP reduction, LDS exchange), once as hipcc emits it (384 v_pk_fma_f32) and once with
// every sum kept scalar.  Each launch's 689 535 outputs are compared bit for bit with a reference taken on an idle chip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/pk_probe.hip -o tools/exp/libpkprobe.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

struct Args { const float* in; int T_in; int C_in; float* out; int T_out; const float* W; const float* bias; int taps; int dil; int clamp; };

static __device__ __forceinline__ float opaque(float v) { asm volatile("" : "+v"(v)); return v; }

#define ROWS 256
template <bool PACKED>
__global__ __launch_bounds__(256) void k_lastconv(Args a) {
    constexpr int CPT = 12, MAXT = 8, NP = ROWS / 32;
    __shared__ float ds[ROWS][MAXT + 1];
    const int tid = threadIdx.x, s8 = tid & 7, rl = tid >> 3;
    const int halo = (a.taps - 1) * a.dil, TO = ROWS - halo, t0 = blockIdx.x * TO;
    float x[NP][CPT];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int src = t0 - halo + p * 32 + rl;
        const int sc = src < 0 ? 0 : (src < a.T_in ? src : a.T_in - 1);
        const float* xr = a.in + (size_t)sc * a.C_in + s8 * CPT;
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            x[p][c] = v.x; x[p][c + 1] = v.y; x[p][c + 2] = v.z; x[p][c + 3] = v.w;
        }
    }
    float w[MAXT][CPT];
#pragma unroll
    for (int tap = 0; tap < MAXT; ++tap) {
        const float* wr = a.W + (size_t)(tap < a.taps ? tap : 0) * a.C_in + s8 * CPT;
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wr + c);
            const bool on = tap < a.taps;
            w[tap][c] = on ? v.x : 0.f; w[tap][c + 1] = on ? v.y : 0.f; w[tap][c + 2] = on ? v.z : 0.f; w[tap][c + 3] = on ? v.w : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int src = t0 - halo + p * 32 + rl;
        const bool inr = src >= 0 && src < a.T_in;
        float d[MAXT];
#pragma unroll
        for (int tap = 0; tap < MAXT; ++tap) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPT; ++c) acc = fmaf(w[tap][c], x[p][c], acc);
            acc = inr ? acc : 0.f;
            if constexpr (!PACKED) acc = opaque(acc);
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, true));
            if constexpr (!PACKED) acc = opaque(acc);
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xF, 0xF, true));
            if constexpr (!PACKED) acc = opaque(acc);
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x141, 0xF, 0xF, true));
            d[tap] = PACKED ? acc : opaque(acc);
        }
        if (s8 == 0) {
#pragma unroll
            for (int tap = 0; tap < MAXT; ++tap) ds[p * 32 + rl][tap] = d[tap];
        }
    }
    __syncthreads();
    const int t = t0 + tid;
    if (tid < TO && t < a.T_out) {
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < MAXT; ++tap)
            if (tap < a.taps) acc += ds[tid + tap * a.dil][tap];
        float v = acc + (a.bias ? a.bias[0] : 0.f);
        if (a.clamp) v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        a.out[t] = v;
    }
}


static float *g_in, *g_W, *g_bias, *g_out;
static std::vector<float> g_ref, g_got;
static hipStream_t g_st;
static int g_T = 3 * 229845, g_tiles;

extern "C" int pk_probe_init() {
    const int C = 96, taps = 7, T = g_T;
    std::vector<float> hin((size_t)T * C), hw((size_t)taps * C), hb(1, 0.01f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hin) v = rnd() * 0.2f;
    for (auto& v : hw) v = rnd() * 0.3f;
    CK(hipMalloc((void**)&g_in, hin.size() * 4)); CK(hipMalloc((void**)&g_W, hw.size() * 4)); CK(hipMalloc((void**)&g_bias, 4)); CK(hipMalloc((void**)&g_out, (size_t)T * 4));
    CK(hipMemcpy(g_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g_W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_bias, hb.data(), 4, hipMemcpyHostToDevice));
    CK(hipStreamCreate(&g_st));
    g_tiles = (T + (ROWS - 6) - 1) / (ROWS - 6);
    g_ref.resize((size_t)T); g_got.resize((size_t)T);
    Args a{ g_in, T, C, g_out, T, g_W, g_bias, taps, 1, 1 };
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_lastconv<false>, dim3(g_tiles), dim3(256), 0, g_st, a); CK(hipStreamSynchronize(g_st));
    CK(hipMemcpy(g_ref.data(), g_out, (size_t)T * 4, hipMemcpyDeviceToHost));
    return g_tiles;
}
// reps launches of the packed (1) or scalar (0) build; returns launches with a wrong output, *bad_samples = wrong samples in all, *worst = largest |error|
extern "C" int pk_probe_run(int packed, int reps, long* bad_samples, float* worst) {
    const int T = g_T;
    Args a{ g_in, T, 96, g_out, T, g_W, g_bias, 7, 1, 1 };
    int bad_launch = 0; *bad_samples = 0; *worst = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(g_out, 0xFF, (size_t)T * 4, g_st));
        if (packed) hipLaunchKernelGGL(k_lastconv<true>, dim3(g_tiles), dim3(256), 0, g_st, a);
        else hipLaunchKernelGGL(k_lastconv<false>, dim3(g_tiles), dim3(256), 0, g_st, a);
        CK(hipStreamSynchronize(g_st));
        CK(hipMemcpy(g_got.data(), g_out, (size_t)T * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t i = 0; i < (size_t)T; ++i)
            if (memcmp(&g_got[i], &g_ref[i], 4) != 0) { ++nb; float d = g_got[i] - g_ref[i]; d = d < 0 ? -d : d; if (d > *worst) *worst = d; }
        *bad_samples += nb; bad_launch += nb != 0;
    }
    return bad_launch;
}
