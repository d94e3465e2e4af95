// microbench_stage_chain.hip — b=1 decode is a chain of ~470 dependent GEMV launches at ~4.6 us each.  Round 2 priced a persistent kernel
// with an atomic-counter grid barrier (5.5 us at 256 workgroups) and did not build it.  The split-K seam of round 3 showed that a hand-off
// built from plain sc1 stores and ONE polled line costs ~1 us per memory round trip, no returning atomic anywhere.  This measures the
// decode chain's stage built that way:
//   L  launches: a hipGraph of STAGES kernels; each: 256 workgroups x (32 KB of bf16-like weights from a rotating 512 MB buffer, the 4 KB
//      input vector the previous kernel wrote, 4 dot products, 4 floats out)            [today's structure, k_gemv1-shaped]
//   P  persistent: ONE launch of 256 workgroups loops over the stages; a stage = {weights of this stage (requested BEFORE the wait: they
//      do not depend on the previous stage), wait until all 256 generation flags show the previous stage, sc1-load the 4 KB vector, 4 dot
//      products, sc1-store 4 floats, drain, sc1-store own flag = stage + 1}.  The wait is one wave polling the 1 KB flag array with one
//      16-byte load per lane.  Bounded spins; a timeout sets a word and every workgroup leaves.
//   P2 the same with the OUTPUT vector three times as long (12 KB in, 12 rows out per workgroup: the gate/up -> down edge)
// Results must agree between L and P (the chain is deterministic: small integers).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_stage_chain tools/microbench_stage_chain.hip ; ./tools/mb_stage_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define AUX_SC1 16
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NWG 256

static __device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one stage's arithmetic for one workgroup: 4 waves = 4 output rows (RPW rows each when the vector is longer), K = 1024 * KX
template <int KX>
static __device__ __forceinline__ float stage_dot(const u32x4 (&w)[2 * KX], const f32x4 (&x)[4 * KX]) {
    // weights: 2 x 16 B per lane per 1024 columns (16 bf16), x: 16 floats per lane per 1024 columns.  Integers only: exact.
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 2 * KX; ++c) {
        const unsigned wu[4] = { w[c].x, w[c].y, w[c].z, w[c].w };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 xv = x[c * 2 + (j >> 1)];
            const float a = (j & 1) ? xv.z : xv.x, b = (j & 1) ? xv.w : xv.y;
            acc = fmaf(a, (float)(wu[j] & 1u), acc);
            acc = fmaf(b, (float)((wu[j] >> 16) & 1u), acc);
        }
    }
    return acc;
}

template <int KX>
__global__ __launch_bounds__(256) void k_stage(const unsigned* wbuf, size_t wwords, const float* xin, float* xout, int stage) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const unsigned* w = wbuf + (((size_t)stage * NWG + wg) * (8192 * KX)) % (wwords - 8192 * KX) + (size_t)wave * 2048 * KX;
    u32x4 wr[2 * KX];
    f32x4 xr[4 * KX];
#pragma unroll
    for (int c = 0; c < 2 * KX; ++c) wr[c] = *reinterpret_cast<const u32x4*>(w + (c * 64 + lane) * 4);
#pragma unroll
    for (int c = 0; c < 4 * KX; ++c) xr[c] = *reinterpret_cast<const f32x4*>(xin + (c * 64 + lane) * 4);
    float s = wave_sum(stage_dot<KX>(wr, xr));
    // next vector: element n = 4 wg + wave (+ 1024 j for the longer vector) = (sum mod 7) + 1, small integers forever
    if (lane == 0)
        for (int j = 0; j < KX; ++j) xout[j * 1024 + wg * 4 + wave] = (float)((int)(s + (float)(stage + j)) % 7 + 1);
}

template <int KX>
__global__ __launch_bounds__(256) void k_persistent(const unsigned* wbuf, size_t wwords, float* xbuf /* [2][1024 KX] */, unsigned* flags /* [NWG] */,
                                                     unsigned* tmo, int stages, int spin_max) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    __shared__ unsigned go;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, 2 * 1024 * KX * 4, 0x00020000);
    for (int stage = 0; stage < stages; ++stage) {
        // 1. this stage's weights: independent of the previous stage, in flight during the wait
        const unsigned* w = wbuf + (((size_t)stage * NWG + wg) * (8192 * KX)) % (wwords - 8192 * KX) + (size_t)wave * 2048 * KX;
        u32x4 wr[2 * KX];
#pragma unroll
        for (int c = 0; c < 2 * KX; ++c) wr[c] = *reinterpret_cast<const u32x4*>(w + (c * 64 + lane) * 4);
        // 2. wait for every workgroup's flag of the previous stage (generation values: never reset)
        if (stage > 0) {
            if (wave == 0) {
                unsigned ok = 0;
                for (int i = 0; i < spin_max; ++i) {
                    asm volatile("" ::: "memory");   // a fresh load every poll: without it the (non-atomic) buffer load is loop-invariant to the compiler
                    const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(flags, 0, NWG * 4, 0x00020000), lane * 16, 0, AUX_SC1);
                    const bool all = f.x >= (unsigned)stage && f.y >= (unsigned)stage && f.z >= (unsigned)stage && f.w >= (unsigned)stage;
                    if (__ballot(all) == ~0ull) { ok = 1; break; }
                }
                if (!ok && lane == 0) __hip_atomic_store(tmo, 1u, RLX, AGENT);
                if (lane == 0) go = ok;
            }
            __syncthreads();
            if (!go) return;
        }
        // 3. the vector the previous stage produced (sc1: L1 never holds it)
        f32x4 xr[4 * KX];
        const unsigned xoff = (unsigned)((stage & 1) * 1024 * KX * 4);
#pragma unroll
        for (int c = 0; c < 4 * KX; ++c) xr[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (c * 64 + lane) * 16, 0, AUX_SC1));
        float s = wave_sum(stage_dot<KX>(wr, xr));
        // 4. publish: values write-through, every storing wave drains, one flag per workgroup
        const unsigned ooff = (unsigned)(((stage + 1) & 1) * 1024 * KX * 4);
        if (lane == 0)
            for (int j = 0; j < KX; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)((int)(s + (float)(stage + j)) % 7 + 1)), rx, ooff + (j * 1024 + wg * 4 + wave) * 4, 0, AUX_SC1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + wg, (unsigned)(stage + 1), RLX, AGENT);
    }
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 400, replays = 10;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t wwords = (size_t)128 << 20;   // 512 MB
    unsigned* wbuf; CK(hipMalloc(&wbuf, wwords * 4));
    {   // weights: low bit of every half-word random-ish but fixed
        std::vector<unsigned> h((size_t)1 << 20);
        unsigned v = 12345;
        for (auto& e : h) { v = v * 1664525u + 1013904223u; e = (v >> 8) & 0x00010001u; }
        for (size_t o = 0; o < wwords; o += h.size()) CK(hipMemcpy(wbuf + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    float* xb; CK(hipMalloc(&xb, 2 * 3072 * 4));
    unsigned *flags, *tmo; CK(hipMalloc(&flags, NWG * 4)); CK(hipMalloc(&tmo, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kx : { 1, 3 }) {
        std::vector<float> x0(3072, 1.0f), ref(1024 * kx), got(1024 * kx);
        // ---- L: launch chain ----
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int s = 0; s < stages; ++s) {
            if (kx == 1) hipLaunchKernelGGL(k_stage<1>, dim3(NWG), dim3(256), 0, st, wbuf, wwords, xb + (s & 1) * 1024, xb + ((s + 1) & 1) * 1024, s);
            else hipLaunchKernelGGL(k_stage<3>, dim3(NWG), dim3(256), 0, st, wbuf, wwords, xb + (s & 1) * 3072, xb + ((s + 1) & 1) * 3072, s);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipMemcpy(xb, x0.data(), 1024 * kx * 4, hipMemcpyHostToDevice));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ref.data(), xb + (stages & 1) * 1024 * kx, 1024 * kx * 4, hipMemcpyDeviceToHost));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float msL; CK(hipEventElapsedTime(&msL, e0, e1));
        // ---- P: persistent ----
        float msP = 0;
        size_t bad = 0; unsigned timeouts = 0;
        for (int rep = 0; rep <= replays; ++rep) {
            CK(hipMemcpyAsync(xb, x0.data(), 1024 * kx * 4, hipMemcpyHostToDevice, st));
            CK(hipMemsetAsync(flags, 0, NWG * 4, st)); CK(hipMemsetAsync(tmo, 0, 64, st));
            CK(hipEventRecord(e0, st));
            if (kx == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(NWG), dim3(256), 0, st, wbuf, wwords, xb, flags, tmo, stages, 1 << 16);
            else hipLaunchKernelGGL(k_persistent<3>, dim3(NWG), dim3(256), 0, st, wbuf, wwords, xb, flags, tmo, stages, 1 << 16);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) msP += ms;
            if (rep == 0) {
                CK(hipMemcpy(got.data(), xb + (stages & 1) * 1024 * kx, 1024 * kx * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
            }
            unsigned t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost)); timeouts += t;
        }
        printf("vector %4d floats, %d stages:  launch chain %.2f us/stage   persistent + flag wait %.2f us/stage   mismatches %zu  timeouts %u\n",
               1024 * kx, stages, msL * 1e3 / replays / stages, msP * 1e3 / replays / stages, bad, timeouts);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
