// unit test of csrc/q3_wave_sort.h on the GPU: hipcc --offload-arch=gfx950 -O2 tools/wave_sort_test.hip -o tools/exp/wave_sort_test
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../leaxer-qwen3-tts_amd/csrc/q3_wave_sort.h"

__global__ void k(const float* in, float* sorted, float* sk, int* st, float* scan_f, int* scan_i, int* xors, float* merged) {
    const int lane = threadIdx.x;
    const float v = in[blockIdx.x * 64 + lane];
    {   // wave_merge4x16_desc: four runs of 16 sorted desc / asc / desc / asc (what k_sample's waves leave in LDS) -> 64 descending
        __shared__ float run[64];
        float r = v;   // sort each run of 16 descending with the network's first 10 stages, odd runs stored reversed
        r = q3::wave_sort_step<2, 1>(r, lane);
        r = q3::wave_sort_step<4, 2>(r, lane); r = q3::wave_sort_step<4, 1>(r, lane);
        r = q3::wave_sort_step<8, 4>(r, lane); r = q3::wave_sort_step<8, 2>(r, lane); r = q3::wave_sort_step<8, 1>(r, lane);
        r = q3::wave_sort_step<16, 8>(r, lane); r = q3::wave_sort_step<16, 4>(r, lane); r = q3::wave_sort_step<16, 2>(r, lane); r = q3::wave_sort_step<16, 1>(r, lane);
        // the network leaves runs alternately descending / ascending already (K = 16 blocks): exactly the layout the merge expects
        run[lane] = r;
        __syncthreads();
        merged[blockIdx.x * 64 + lane] = q3::wave_merge4x16_desc(run[lane], lane);
    }
    sorted[blockIdx.x * 64 + lane] = q3::wave_sort_desc(v, lane);
    float kk = v; int t = lane;
    q3::wave_sort_desc_kv(kk, t, lane);
    sk[blockIdx.x * 64 + lane] = kk; st[blockIdx.x * 64 + lane] = t;
    scan_f[blockIdx.x * 64 + lane] = q3::wave_scan_incl_f(v);
    scan_i[blockIdx.x * 64 + lane] = q3::wave_scan_incl_i((int)(v * 8));
    if (blockIdx.x == 0) {
        xors[0 * 64 + lane] = q3::wave_xor_lane_i<1>(lane, lane); xors[1 * 64 + lane] = q3::wave_xor_lane_i<2>(lane, lane);
        xors[2 * 64 + lane] = q3::wave_xor_lane_i<4>(lane, lane); xors[3 * 64 + lane] = q3::wave_xor_lane_i<8>(lane, lane);
        xors[4 * 64 + lane] = q3::wave_xor_lane_i<16>(lane, lane); xors[5 * 64 + lane] = q3::wave_xor_lane_i<32>(lane, lane);
    }
}
int main() {
    const int NB = 2000;
    std::vector<float> h(NB * 64);
    srand(3);
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 64; ++i) {
            float x = (float)(rand() % (b % 5 == 0 ? 7 : 100000)) / 8.0f - 3.0f;          // every 5th block is full of ties
            if (b % 7 == 0 && i % 3 == 0) x = -INFINITY;
            h[b * 64 + i] = x;
        }
    float *din, *ds, *dk, *dsf, *dm; int *dt, *dsi, *dx;
    hipMalloc(&dm, NB * 256);
    hipMalloc(&din, NB * 256); hipMalloc(&ds, NB * 256); hipMalloc(&dk, NB * 256); hipMalloc(&dt, NB * 256); hipMalloc(&dsf, NB * 256); hipMalloc(&dsi, NB * 256); hipMalloc(&dx, 6 * 256);
    hipMemcpy(din, h.data(), NB * 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, din, ds, dk, dt, dsf, dsi, dx, dm);
    std::vector<float> s(NB * 64), sk(NB * 64), sf(NB * 64); std::vector<int> st(NB * 64), si(NB * 64), x(6 * 64);
    hipMemcpy(s.data(), ds, NB * 256, hipMemcpyDeviceToHost); hipMemcpy(sk.data(), dk, NB * 256, hipMemcpyDeviceToHost);
    hipMemcpy(st.data(), dt, NB * 256, hipMemcpyDeviceToHost); hipMemcpy(sf.data(), dsf, NB * 256, hipMemcpyDeviceToHost);
    hipMemcpy(si.data(), dsi, NB * 256, hipMemcpyDeviceToHost); hipMemcpy(x.data(), dx, 6 * 256, hipMemcpyDeviceToHost);
    std::vector<float> mg(NB * 64);
    hipMemcpy(mg.data(), dm, NB * 256, hipMemcpyDeviceToHost);
    int bad = 0;
    const int js[6] = {1, 2, 4, 8, 16, 32};
    for (int q = 0; q < 6; ++q) for (int i = 0; i < 64; ++i) if (x[q * 64 + i] != (i ^ js[q])) { if (bad < 10) printf("xor%d lane %d -> %d\n", js[q], i, x[q * 64 + i]); ++bad; }
    for (int b = 0; b < NB; ++b) {
        std::vector<std::pair<float, int>> ref(64);
        for (int i = 0; i < 64; ++i) ref[i] = { h[b * 64 + i], i };
        std::stable_sort(ref.begin(), ref.end(), [](auto& a, auto& c) { return a.first > c.first; });
        double cf = 0; long ci = 0;
        for (int i = 0; i < 64; ++i) {
            if (mg[b * 64 + i] != ref[i].first) { if (bad < 10) printf("merge block %d rank %d: %g want %g\n", b, i, mg[b*64+i], ref[i].first); ++bad; }
            if (s[b * 64 + i] != ref[i].first || sk[b * 64 + i] != ref[i].first || st[b * 64 + i] != ref[i].second) { if (bad < 10) printf("block %d rank %d: %g/%g tag %d, want %g tag %d\n", b, i, s[b*64+i], sk[b*64+i], st[b*64+i], ref[i].first, ref[i].second); ++bad; }
            cf += h[b * 64 + i]; ci += (int)(h[b * 64 + i] * 8);
            if (si[b * 64 + i] != ci && std::isfinite(h[b * 64 + i]) && b % 7 != 0) { if (bad < 10) printf("iscan block %d lane %d: %d want %ld\n", b, i, si[b*64+i], ci); ++bad; }
            if (b % 7 != 0 && std::fabs(sf[b * 64 + i] - cf) > 1e-3 * (1 + std::fabs(cf))) { if (bad < 10) printf("fscan block %d lane %d: %g want %g\n", b, i, sf[b*64+i], cf); ++bad; }
        }
    }
    printf(bad ? "FAILED: %d mismatches\n" : "wave sort / scan / xor-lane: all %d checks passed\n", bad ? bad : NB * 64 * 4 + 384);
    return bad != 0;
}
