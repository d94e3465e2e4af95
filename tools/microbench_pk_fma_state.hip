// microbench_pk_fma_state.hip — does a dense v_pk_fma_f32 kernel return wrong sums when it starts shortly after sustained heavy load?
// Round 5 (profiles/r05_hunt/README.txt): the vocoder's last conv, compiled to packed fp32 FMAs, returned wrong partial sums in the
// product — only within a few hundred microseconds of the preceding heavy kernels, never after >= 1 ms of idleness, never with scalar
// FMAs.  This is the same experiment WITHOUT the product: the conv below is that kernel's loop structure (8 lanes per input row, 12
// channels x 8 taps of weights in registers, 8 passes of 32 rows, DPP reduction, LDS exchange) in a packed and a scalar build, a
// matrix-core loop plays the heavy phase, one sleeping wave the pause.  Each repetition: [heavy for --heavy-ms] -> [pause] -> conv ->
// compare every output with a reference taken on an idle chip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_pk_fma_state.hip -o tools/mb_pk_state && tools/mb_pk_state
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

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

// the heavy phase: register-only matrix-core work on every CU until `us` microseconds of the 100 MHz clock have passed
__global__ __launch_bounds__(256) void k_heavy(float* sink, int us) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)us * 100ull) {
        for (int i = 0; i < 64; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;
}
// the same with a streaming read of `n` floats per pass (an HBM-bound heavy phase)
__global__ __launch_bounds__(256) void k_heavy_mem(const float4* src, size_t n4, float* sink, int us) {
    float4 s = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)us * 100ull) {
        for (int k = 0; k < 16; ++k) { const float4 v = src[i % n4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; i += (size_t)gridDim.x * 256; }
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}
// matrix cores, HBM stream and LDS traffic at once (what a fused conv kernel of the vocoder does): waves 0-1 of a workgroup multiply, waves 2-3
// stream 16-byte reads through LDS
__global__ __launch_bounds__(256) void k_heavy_mix(const float4* src, size_t n4, float* sink, int us) {
    __shared__ float4 lds[2][128];
    const int wave = threadIdx.x >> 6;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    if (wave < 2) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
        float16v c0 = {}, c1 = {};
        while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)us * 100ull)
            for (int i = 0; i < 64; ++i) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); }
        for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    } else {
        const int l = threadIdx.x - 128;
        size_t i = (size_t)blockIdx.x * 128 + l;
        while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)us * 100ull)
            for (int k = 0; k < 16; ++k) {
                lds[k & 1][l] = src[i % n4];
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the write has landed before the neighbour's slot is read
                const float4 v = lds[k & 1][l ^ 1];
                acc += v.x + v.y + v.z + v.w; i += (size_t)gridDim.x * 128;
            }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_idle_us(int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
    int reps = 300, heavy_us = 3000, T = 3 * 229845;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--heavy-ms")) heavy_us = (int)(atof(argv[++i]) * 1000);
        else if (!strcmp(argv[i], "--rows")) T = atoi(argv[++i]);
    }
    const int C = 96, taps = 7;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<float> hin((size_t)T * C), hw((size_t)taps * C), hb(1, 0.01f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hin) v = rnd() * 0.2f;
    for (auto& v : hw) v = rnd() * 0.3f;
    float *in, *W, *bias, *out, *sink; float4* big;
    const size_t big_n4 = (size_t)64 << 20;   // 1 GiB of float4 for the memory-bound heavy phase
    CK(hipMalloc((void**)&in, hin.size() * 4)); CK(hipMalloc((void**)&W, hw.size() * 4)); CK(hipMalloc((void**)&bias, 4));
    CK(hipMalloc((void**)&out, (size_t)T * 4)); CK(hipMalloc((void**)&sink, 4096)); CK(hipMalloc((void**)&big, big_n4 * 16));
    CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), 4, hipMemcpyHostToDevice)); CK(hipMemset(big, 0, big_n4 * 16));
    Args a{ in, T, C, out, T, W, bias, taps, 1, 1 };
    const int TO = ROWS - (taps - 1), tiles = (T + TO - 1) / TO;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<float> ref((size_t)T), got((size_t)T), ref2((size_t)T);
    // references on an idle chip: scalar and packed must agree bit for bit (fma is fma)
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_lastconv<false>, dim3(tiles), dim3(256), 0, st, a); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref.data(), out, (size_t)T * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0xFF, (size_t)T * 4));
    hipLaunchKernelGGL(k_lastconv<true>, dim3(tiles), dim3(256), 0, st, a); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref2.data(), out, (size_t)T * 4, hipMemcpyDeviceToHost));
    size_t dif = 0; for (size_t i = 0; i < (size_t)T; ++i) dif += memcmp(&ref[i], &ref2[i], 4) != 0;
    printf("%d CUs; conv of %d rows x %d channels, %d taps = %d workgroups; idle chip: packed vs scalar outputs differ at %zu of %d samples\n", cus, T, C, taps, tiles, dif, T);
    struct Phase { const char* name; int heavy; int gap_us; bool packed; };
    const Phase phases[] = {
        { "packed, no heavy phase", 0, 0, true }, { "packed, mfma heavy, no pause", 1, 0, true }, { "packed, mfma heavy, 10 us pause", 1, 10, true },
        { "packed, mfma heavy, 100 us pause", 1, 100, true }, { "packed, mfma heavy, 1 ms pause", 1, 1000, true },
        { "packed, hbm heavy, no pause", 2, 0, true }, { "packed, hbm heavy, 10 us pause", 2, 10, true }, { "packed, hbm heavy, 1 ms pause", 2, 1000, true },
        { "packed, mixed heavy, no pause", 3, 0, true }, { "packed, mixed heavy, 10 us pause", 3, 10, true }, { "packed, mixed heavy, 100 us pause", 3, 100, true },
        { "packed, mixed heavy, 1 ms pause", 3, 1000, true },
        { "scalar, mfma heavy, 10 us pause", 1, 10, false }, { "scalar, hbm heavy, 10 us pause", 2, 10, false },
    };
    for (const Phase& ph : phases) {
        long bad_launch = 0, bad_samples = 0; size_t first_bad = 0; float worst = 0.f;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(out, 0xFF, (size_t)T * 4, st));
            if (ph.heavy == 1) hipLaunchKernelGGL(k_heavy, dim3(cus * 2), dim3(256), 0, st, sink, heavy_us);
            if (ph.heavy == 2) hipLaunchKernelGGL(k_heavy_mem, dim3(cus * 8), dim3(256), 0, st, big, big_n4, sink, heavy_us);
            if (ph.heavy == 3) hipLaunchKernelGGL(k_heavy_mix, dim3(cus * 2), dim3(256), 0, st, big, big_n4, sink, heavy_us);
            if (ph.gap_us) hipLaunchKernelGGL(k_idle_us, dim3(1), dim3(64), 0, st, ph.gap_us);
            if (ph.packed) hipLaunchKernelGGL(k_lastconv<true>, dim3(tiles), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_lastconv<false>, dim3(tiles), dim3(256), 0, st, a);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(got.data(), out, (size_t)T * 4, hipMemcpyDeviceToHost));
            long nb = 0;
            for (size_t i = 0; i < (size_t)T; ++i)
                if (memcmp(&got[i], &ref[i], 4) != 0) { if (!nb && !bad_launch) first_bad = i; ++nb; const float d = got[i] - ref[i]; worst = d < 0 ? (-d > worst ? -d : worst) : (d > worst ? d : worst); }
            bad_samples += nb; bad_launch += nb != 0;
        }
        printf("%-36s %4d launches: %4ld with a wrong output, %6ld wrong samples, worst |error| %.3g%s\n", ph.name, reps, bad_launch, bad_samples, worst,
               bad_launch ? "" : "");
        if (bad_launch) printf("    first wrong sample %zu = tile %zu, output %zu of its 250\n", first_bad, first_bad / TO, first_bad % TO);
        fflush(stdout);
    }
    return 0;
}
