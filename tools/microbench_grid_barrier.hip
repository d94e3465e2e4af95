// microbench_grid_barrier.hip — what does a grid-wide barrier cost on THIS box, next to the 1.6 us kernel boundary the decode chain pays?
// (DESIGN.md rejected a persistent kernel per predictor pass on the guide's barrier price, 4.1-4.7 us at 256
// workgroups, without measuring it here.)
//
// One launch of NWG workgroups (256 threads, one per CU at most: all resident) runs ITERS barriers back to back; variants:
//   flat      one monotonic counter: release fence -> atomic add -> poll (relaxed, s_sleep) -> acquire fence
//   nofence   the same without the two fences: the pure arrive/poll cost (a lower bound; NOT a valid barrier for data)
//   xcd       hierarchical: 8 group counters (blockIdx & 7 = the blocks that share an XCD under round-robin placement: speed only, any
//             partition is correct), group leader -> top counter -> generation word; every workgroup acquires
//   xcd+data  xcd, and between barriers every workgroup publishes 4 KB (plain stores under the release fence) and reads the 4 KB another workgroup
//             published in the previous phase: the all-to-all hand-off of an activation vector that a fused decode stage needs
// Every spin is bounded (2^20 polls, then a timeout word is set and the workgroup leaves): a mis-sized grid ends, it never hangs.
// Times: in-kernel wall clock (s_memrealtime, 100 MHz) of workgroup 0 over the ITERS barriers, and the host-paired event time of the
// whole launch minus an empty launch of the same grid.  Also: a chain of ITERS trivial dependent kernels in a hipGraph (the boundary).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

struct BarState {            // every word on its own 128-byte line
    unsigned top[32];        // [0] counter
    unsigned gen[32];        // [0] generation
    unsigned grp[8][32];     // per-group counters
    unsigned tmo[32];        // [0] timeout flag
};

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned want, unsigned* tmo) {
    for (unsigned spins = 0; spins < (1u << 20); ++spins) {
        if (__hip_atomic_load(p, RLX, AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(tmo, 1u, RLX, AGENT);
    return false;
}

template <int MODE>   // 0 flat, 1 nofence, 2 xcd, 3 xcd + data
__global__ __launch_bounds__(256) void k_barriers(BarState* st, int iters, u64* stamps, float* buf, float* sink) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const int g = wg & 7, gsize = (nwg + 7 - g) / 8;    // members of my group
    __shared__ int ok;
    if (tid == 0) ok = 1;
    __syncthreads();
    u64 t0 = 0;
    if (wg == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {   // publish this phase's 4 KB: 256 threads x 16 B, plain stores; the barrier's release fence (L2 write-back) covers them
            *reinterpret_cast<float4*>(&buf[((size_t)(it & 1) * nwg + wg) * 1024 + tid * 4]) = make_float4((float)(it + wg), 1.f, 2.f, 3.f);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup's barrier
        }
        __syncthreads();
        if (tid == 0 && ok) {
            const unsigned want = (unsigned)(it + 1);
            if (MODE == 0 || MODE == 1) {
                if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_fetch_add(&st->top[0], 1u, RLX, AGENT);
                if (!wait_ge(&st->top[0], want * (unsigned)nwg, &st->tmo[0])) ok = 0;
                if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned old = __hip_atomic_fetch_add(&st->grp[g][0], 1u, RLX, AGENT);
                if (old + 1 == want * (unsigned)gsize) {   // last of my group this generation
                    const unsigned ot = __hip_atomic_fetch_add(&st->top[0], 1u, RLX, AGENT);
                    const int ngroups = nwg < 8 ? nwg : 8;
                    if (ot + 1 == want * (unsigned)ngroups) __hip_atomic_store(&st->gen[0], want, RLX, AGENT);
                }
                if (!wait_ge(&st->gen[0], want, &st->tmo[0])) ok = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (MODE == 3) {   // read what a DIFFERENT workgroup published this phase (behind the acquire)
            const int src = (wg + 1 + it) % nwg;
            const float4 v = *reinterpret_cast<const float4*>(&buf[((size_t)(it & 1) * nwg + src) * 1024 + tid * 4]);
            acc += v.x + v.y;
        }
        if (!ok) break;
    }
    if (wg == 0 && tid == 0) { stamps[0] = t0; stamps[1] = __builtin_amdgcn_s_memrealtime(); }
    if (MODE == 3 && acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_empty(float* sink) { if (sink == nullptr && threadIdx.x == 999) sink[0] = 0.f; }
__global__ __launch_bounds__(256) void k_chain(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }

template <int MODE>
static void run(const char* name, int nwg, int iters, BarState* st, u64* stamps, float* buf, float* sink, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best_in = 1e30, best_host = 1e30;
    int timeouts = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(st, 0, sizeof(BarState), s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((k_barriers<MODE>), dim3(nwg), dim3(256), 0, s, st, iters, stamps, buf, sink);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        u64 h[2]; CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        BarState hs; CK(hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost));
        timeouts += hs.tmo[0];
        const double in_us = (double)(h[1] - h[0]) / 100.0 / iters, host_us = ms * 1e3 / iters;
        if (rep > 0) { best_in = in_us < best_in ? in_us : best_in; best_host = host_us < best_host ? host_us : best_host; }
    }
    printf("%-9s %3d workgroups x %d barriers: %.2f us per barrier in-kernel (workgroup 0), %.2f us host-paired%s\n", name, nwg, iters, best_in, best_host,
           timeouts ? "  ** TIMEOUT: grid not co-resident? **" : "");
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    BarState* st; CK(hipMalloc((void**)&st, sizeof(BarState)));
    u64* stamps; CK(hipMalloc((void**)&stamps, 64));
    float *buf, *sink; CK(hipMalloc((void**)&buf, (size_t)2 * 256 * 1024 * sizeof(float))); CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(buf, 0, (size_t)2 * 256 * 1024 * sizeof(float)));
    const int iters = 200;
    for (int nwg : { 64, 128, 256 }) {
        run<1>("nofence", nwg, iters, st, stamps, buf, sink, s);
        run<0>("flat", nwg, iters, st, stamps, buf, sink, s);
        run<2>("xcd", nwg, iters, st, stamps, buf, sink, s);
        run<3>("xcd+data", nwg, iters, st, stamps, buf, sink, s);
    }
    // the alternative: ITERS dependent trivial kernels of the same grid replayed from a hipGraph (kernel boundary)
    for (int nwg : { 64, 256 }) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_chain, dim3(nwg), dim3(256), 0, s, sink);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("boundary  %3d workgroups: %.2f us per dependent trivial kernel (hipGraph replay)\n", nwg, ms * 1e3 / 10 / iters);
    }
    return 0;
}
