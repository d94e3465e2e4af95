// q3_decode_kernels.hip — gfx950 kernels of the autoregressive hot path:
//   talker_prefill / talker_decode / code_predictor sessions of the reference
//   (src/tts_onnx.cpp:615-757) and its host sampler (:878-950), rebuilt as HIP.
//
// Design notes:
//  * Decode is weight-streaming bound (M = batch rows <= 8 per launch): k_gemv streams bf16 weight
//    rows straight global->VGPR in 16 B/lane (1 KiB per wave instruction), fp32 activations from L2,
//    fp32 FMA on the VALU, with RMSNorm / residual / SwiGLU / bias fused so a decoder layer is
//    five launches.  (LDS staging is pure overhead for an operand streamed once and not shared
//    between waves — cdna_hip_programming.md "GEMV / M <= 16 decode weights".)
//  * k_attn fuses per-head q/k RMSNorm + RoPE + KV append + softmax(QK^T)V over a paged fp32 cache.
//  * k_sample keeps temperature/top-k/top-p sampling on device and gathers the sampled token's
//    embedding in its epilogue, so a generated frame needs no host round trip.
// Built with -ffp-contract=off: elementwise math rounds like the fp32 oracle; dot products use
// explicit fmaf.
#include "q3_common.h"
#include "q3_wave_sort.h"
#include <algorithm>

namespace q3 {

// -DQ3_SAMPLE_PROF: phase timestamps (100 MHz wall clock) of the last workgroup-0 run of an instrumented kernel; tools/ only
#ifdef Q3_SAMPLE_PROF
__device__ long long g_kernel_prof[32];
void sample_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kernel_prof), sizeof(long long) * 32); }
#define KP_MARK(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_kernel_prof[k] = wall_clock64(); } while (0)
#else
#define KP_MARK(k) do { } while (0)
#endif

// Cross-lane reductions on the DPP path (no LDS round trip; __shfl_xor lowers to ds_bpermute + a full
// lgkmcnt wait per step).  quad_perm / row_ror stay inside a 16-lane row; row_bcast15/31 carry the
// row totals up to lane 63, which is broadcast through an SGPR.
#define Q3_DPP_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define Q3_DPP_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define Q3_DPP_ROR4 0x124  // row_ror:4
#define Q3_DPP_ROR8 0x128  // row_ror:8
#define Q3_DPP_BCAST15 0x142
#define Q3_DPP_BCAST31 0x143
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dpp_f(float oldv, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int dpp_i(int oldv, int v) {
    return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROW_MASK, 0xF, false);
}
// sum over the 16 lanes of a DPP row (every lane of the row gets the total)
static __device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<Q3_DPP_XOR1, 0xF>(0.f, v);
    v += dpp_f<Q3_DPP_XOR2, 0xF>(0.f, v);
    v += dpp_f<Q3_DPP_ROR4, 0xF>(0.f, v);
    v += dpp_f<Q3_DPP_ROR8, 0xF>(0.f, v);
    return v;
}
static __device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    v += dpp_f<Q3_DPP_BCAST15, 0xA>(0.f, v);
    v += dpp_f<Q3_DPP_BCAST31, 0xC>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
static __device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<Q3_DPP_XOR1, 0xF>(v, v));
    v = fmaxf(v, dpp_f<Q3_DPP_XOR2, 0xF>(v, v));
    v = fmaxf(v, dpp_f<Q3_DPP_ROR4, 0xF>(v, v));
    v = fmaxf(v, dpp_f<Q3_DPP_ROR8, 0xF>(v, v));
    v = fmaxf(v, dpp_f<Q3_DPP_BCAST15, 0xA>(v, v));
    v = fmaxf(v, dpp_f<Q3_DPP_BCAST31, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
static __device__ __forceinline__ float lane_bcast(float v, int src_lane /* wave-uniform */) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
static __device__ __forceinline__ int lane_bcast_i(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
static __device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
static __device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }
static __device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ================================================================================================
// k_gemv
// ================================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // one 16-byte access (HIP's float4 STRUCT is split into scalars by SROA and re-merged only where the vectorizer can prove it)
template <bool NT>
static __device__ __forceinline__ uint4 ldw(const bf16_t* p) {
    u32x4 v;
    if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else v = *reinterpret_cast<const u32x4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int MT, int R, int EPI, bool NORM, bool NT>
__global__ __launch_bounds__(256) void k_gemv(GemvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int K = a.K, N = a.N, M = a.M;

    float inv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) inv[m] = 1.f;
    if (NORM) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < M) {
                const float* xr = a.x + (size_t)m * a.ldx;
                float ss = 0.f;
                for (int k = lane * 4; k < K; k += 256) {
                    float4 v = *reinterpret_cast<const float4*>(xr + k);
                    ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
                }
                ss = wave_sum(ss);
                inv[m] = 1.0f / sqrtf(ss / (float)K + a.eps);
            }
        }
        if (a.xn_out != nullptr && blockIdx.x == 0 && wave == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < M) {
                    const float* xr = a.x + (size_t)m * a.ldx;
                    float* o = a.xn_out + (size_t)m * a.ld_xn;
                    for (int k = lane * 4; k < K; k += 256) {
                        float4 v = *reinterpret_cast<const float4*>(xr + k);
                        float4 g = *reinterpret_cast<const float4*>(a.gamma + k);
                        float4 r;
                        r.x = g.x * (v.x * inv[m]); r.y = g.y * (v.y * inv[m]);
                        r.z = g.z * (v.z * inv[m]); r.w = g.w * (v.w * inv[m]);
                        *reinterpret_cast<float4*>(o + k) = r;
                    }
                }
            }
        }
    }

    for (int n0 = gw * R; n0 < N; n0 += nw * R) {
        float acc[MT][R];
        float acc2[MT][R];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) { acc[m][r] = 0.f; acc2[m][r] = 0.f; }

#pragma unroll 2
        for (int k = lane * 8; k < K; k += 512) {
            uint4 w[R], w2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = n0 + r < N ? n0 + r : N - 1;
                w[r] = ldw<NT>(a.W + (size_t)n * K + k);
                if (EPI == EPI_SWIGLU) w2[r] = ldw<NT>(a.W2 + (size_t)n * K + k);
            }
            float g[8];
            if (NORM) {
                float4 g0 = *reinterpret_cast<const float4*>(a.gamma + k);
                float4 g1 = *reinterpret_cast<const float4*>(a.gamma + k + 4);
                g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < M) {
                    const float* xr = a.x + (size_t)m * a.ldx + k;
                    float4 x0 = *reinterpret_cast<const float4*>(xr);
                    float4 x1 = *reinterpret_cast<const float4*>(xr + 4);
                    float xv[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
                    if (NORM) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = g[j] * (xv[j] * inv[m]);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t wu[4] = { w[r].x, w[r].y, w[r].z, w[r].w };
                        float s = acc[m][r];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s = fmaf(xv[2 * j], bf_lo(wu[j]), s); s = fmaf(xv[2 * j + 1], bf_hi(wu[j]), s); }
                        acc[m][r] = s;
                        if (EPI == EPI_SWIGLU) {
                            const uint32_t vu[4] = { w2[r].x, w2[r].y, w2[r].z, w2[r].w };
                            float s2 = acc2[m][r];
#pragma unroll
                            for (int j = 0; j < 4; ++j) { s2 = fmaf(xv[2 * j], bf_lo(vu[j]), s2); s2 = fmaf(xv[2 * j + 1], bf_hi(vu[j]), s2); }
                            acc2[m][r] = s2;
                        }
                    }
                }
            }
        }
        // cross-lane reduction, then lane (m*R + r) owns output (m, n0 + r)
        float mine = 0.f, mine2 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v = wave_sum(acc[m][r]);
                if (lane == m * R + r) mine = v;
                if (EPI == EPI_SWIGLU) {
                    float v2 = wave_sum(acc2[m][r]);
                    if (lane == m * R + r) mine2 = v2;
                }
            }
        if (lane < MT * R) {
            const int m = lane / R, r = lane % R, n = n0 + r;
            if (m < M && n < N) {
                float o;
                if (EPI == EPI_STORE) o = mine;
                else if (EPI == EPI_RESIDUAL) o = a.res[(size_t)m * a.ldres + n] + mine;
                else if (EPI == EPI_SWIGLU) o = silu_f(mine) * mine2;
                else if (EPI == EPI_BIAS) o = mine + a.bias[n];
                else o = silu_f(mine + a.bias[n]);
                a.out[(size_t)m * a.ldo + n] = o;
            }
        }
    }
}

template <int MT, int R, int EPI, bool NORM>
static void gemv_go(const GemvArgs& a, int grid, hipStream_t s) {
    if (a.nt) hipLaunchKernelGGL((k_gemv<MT, R, EPI, NORM, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemv<MT, R, EPI, NORM, false>), dim3(grid), dim3(256), 0, s, a);
}
template <int MT, int R>
static void gemv_epi(const GemvArgs& a, int grid, hipStream_t s) {
    const bool norm = a.gamma != nullptr;
    switch (a.epi) {
    case EPI_STORE: norm ? gemv_go<MT, R, EPI_STORE, true>(a, grid, s) : gemv_go<MT, R, EPI_STORE, false>(a, grid, s); break;
    case EPI_SWIGLU: norm ? gemv_go<MT, R, EPI_SWIGLU, true>(a, grid, s) : gemv_go<MT, R, EPI_SWIGLU, false>(a, grid, s); break;
    case EPI_RESIDUAL: gemv_go<MT, R, EPI_RESIDUAL, false>(a, grid, s); break;
    case EPI_BIAS: gemv_go<MT, R, EPI_BIAS, false>(a, grid, s); break;
    case EPI_BIAS_SILU: gemv_go<MT, R, EPI_BIAS_SILU, false>(a, grid, s); break;
    default: throw Error("gemv: bad epilogue");
    }
}
template <int MT>
static void gemv_r(const GemvArgs& a, hipStream_t s) {
    // rows per wave: 2 when there are plenty of rows, else 1 so that >= 256 workgroups exist
    const int R = (a.N >= 4096 && a.epi != EPI_SWIGLU) ? 2 : 1;
    int grid = (a.N + 4 * R - 1) / (4 * R);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    if (R == 2) gemv_epi<MT, 2>(a, grid, s); else gemv_epi<MT, 1>(a, grid, s);
}


// ================================================================================================
// k_gemv1 — single-pass decode GEMV for M <= 2 rows and K = NCH*512: every weight load of the wave
// (RW rows x NCH x 16 B/lane) is issued before anything else, the activation row is read once into
// registers (its RMSNorm statistics come from those same registers), so the kernel is one memory
// latency deep.  COMB: the activation rows are the combination of split-T attention partials.
// ================================================================================================
static __device__ __forceinline__ uint4 ldw_rt(const bf16_t* p, bool nt) {
    u32x4 v;
    if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else v = *reinterpret_cast<const u32x4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The leading scalar parameters repeat the fields the first loads need: the file is built with -amdgpu-kernarg-preload-count=16, so the
// command processor hands them to the wave in SGPRs at launch instead of the wave fetching its kernarg segment first (~0.35 us per
// launch on a chain of dependent GEMVs; a by-value struct is not preloadable).
template <int MT, int NCH, int RW, int EPI, bool NORM, bool COMB>
__global__ __launch_bounds__(256) void k_gemv1(const bf16_t* pW, const bf16_t* pW2, const float* px, const float* pgamma, const float* pepi,
                                                int pN, int pM, int pldx, int pldepi, int pnt, GemvArgs a) {
    constexpr int K = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // provably wave-uniform: row addresses stay scalar
    const int n0 = (blockIdx.x * 4 + wave) * RW;
    const int N = pN, M = pM;
    __shared__ float xs[COMB ? MT * K : 1];

    KP_MARK(30); KP_MARK(31);   // back to back: the first one absorbs the kernarg fetch, their distance is the cost of a mark
    KP_MARK(8);
    // 1. weights: everything this wave will ever read, in flight at once
    uint4 w[RW][NCH], w2[RW][NCH];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[r][c] = ldw_rt(pW + (size_t)n * K + c * 512 + lane * 8, pnt != 0);
            if (EPI == EPI_SWIGLU) w2[r][c] = ldw_rt(pW2 + (size_t)n * K + c * 512 + lane * 8, pnt != 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0); // keep the weight loads ahead of everything below (hipcc sinks them otherwise)
    // residual / bias operands of the epilogue: fetched now, not after the reduction
    float epi_in = 0.f;
    if (EPI == EPI_RESIDUAL || EPI == EPI_BIAS || EPI == EPI_BIAS_SILU) {
        if (lane < MT * RW) {
            const int m = lane / RW, n = n0 + lane % RW;
            if (m < M && n < N) epi_in = EPI == EPI_RESIDUAL ? pepi[(size_t)m * pldepi + n] : pepi[n];
        }
    }
    // 2. activations
    float xv[MT][NCH][8];
    if (COMB) {
        for (int k8 = threadIdx.x * 8; k8 < K; k8 += 2048) {
            const int head = k8 / a.pd, e0 = k8 % a.pd;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < M) {
                    const int bi = m / a.pn_new, inew = m % a.pn_new;
                    const int pos = (a.ppos_dev ? a.ppos_dev[a.pslot_offset + bi] : a.ppos_scalar) + inew;
                    int nact = pos / a.pchunk + 1;
                    if (nact > a.pS) nact = a.pS;
                    const size_t idx = ((size_t)m * a.pheads + head) * a.pS;
                    // online merge in branch-free batches of CQ splits (clamped addresses, zero weight past nact).  A batch is one memory
                    // round (the partials come from other XCDs: ~0.4 us each); 12 covers a context of 1536 tokens in one round where
                    // batches of 4 took three (loads past nact repeat the last split's addresses: same cache lines, no extra traffic)
                    constexpr int CQ = 12;
                    float mx = -INFINITY, L = 0.f, O[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    for (int sp0 = 0; sp0 < nact; sp0 += CQ) {
                        float pmv[CQ], plv[CQ];
                        float4 o0[CQ], o1[CQ];
#pragma unroll
                        for (int q = 0; q < CQ; ++q) {
                            const int sp = sp0 + q < nact ? sp0 + q : nact - 1;
                            pmv[q] = a.pm[idx + sp]; plv[q] = a.pl[idx + sp];
                            o0[q] = *reinterpret_cast<const float4*>(a.po + (idx + sp) * a.pd + e0);
                            o1[q] = *reinterpret_cast<const float4*>(a.po + (idx + sp) * a.pd + e0 + 4);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float mn = mx;
#pragma unroll
                        for (int q = 0; q < CQ; ++q) mn = fmaxf(mn, pmv[q]);
                        const float corr = __expf(mx - mn);
                        L *= corr;
#pragma unroll
                        for (int j = 0; j < 8; ++j) O[j] *= corr;
#pragma unroll
                        for (int q = 0; q < CQ; ++q) {
                            const float wgt = sp0 + q < nact ? __expf(pmv[q] - mn) : 0.f;
                            L += wgt * plv[q];
                            O[0] += wgt * o0[q].x; O[1] += wgt * o0[q].y; O[2] += wgt * o0[q].z; O[3] += wgt * o0[q].w;
                            O[4] += wgt * o1[q].x; O[5] += wgt * o1[q].y; O[6] += wgt * o1[q].z; O[7] += wgt * o1[q].w;
                        }
                        mx = mn;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[m * K + k8 + j] = O[j] / L;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float4 x0 = *reinterpret_cast<const float4*>(&xs[m * K + c * 512 + lane * 8]);
                const float4 x1 = *reinterpret_cast<const float4*>(&xs[m * K + c * 512 + lane * 8 + 4]);
                xv[m][c][0] = x0.x; xv[m][c][1] = x0.y; xv[m][c][2] = x0.z; xv[m][c][3] = x0.w;
                xv[m][c][4] = x1.x; xv[m][c][5] = x1.y; xv[m][c][6] = x1.z; xv[m][c][7] = x1.w;
            }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float* xr = px + (size_t)(m < M ? m : 0) * pldx + c * 512 + lane * 8;
                const float4 x0 = *reinterpret_cast<const float4*>(xr);
                const float4 x1 = *reinterpret_cast<const float4*>(xr + 4);
                xv[m][c][0] = x0.x; xv[m][c][1] = x0.y; xv[m][c][2] = x0.z; xv[m][c][3] = x0.w;
                xv[m][c][4] = x1.x; xv[m][c][5] = x1.y; xv[m][c][6] = x1.z; xv[m][c][7] = x1.w;
            }
    }
    float g[NORM ? NCH : 1][8];
    if (NORM) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 g0 = *reinterpret_cast<const float4*>(pgamma + c * 512 + lane * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(pgamma + c * 512 + lane * 8 + 4);
            g[c][0] = g0.x; g[c][1] = g0.y; g[c][2] = g0.z; g[c][3] = g0.w; g[c][4] = g1.x; g[c][5] = g1.y; g[c][6] = g1.z; g[c][7] = g1.w;
        }
    }
    KP_MARK(9);
    __builtin_amdgcn_sched_barrier(0); // all loads of the kernel are in flight past this point
    if (NORM) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) ss = fmaf(xv[m][c][j], xv[m][c][j], ss);
            ss = wave_sum(ss);
            const float inv = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[m][c][j] = g[c][j] * (xv[m][c][j] * inv);
            if (a.xn_out != nullptr && blockIdx.x == 0 && wave == 0 && m < M) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float* o = a.xn_out + (size_t)m * a.ld_xn + c * 512 + lane * 8;
                    *reinterpret_cast<float4*>(o) = make_float4(xv[m][c][0], xv[m][c][1], xv[m][c][2], xv[m][c][3]);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(xv[m][c][4], xv[m][c][5], xv[m][c][6], xv[m][c][7]);
                }
            }
        }
    }
    KP_MARK(10);
    // 3. dot products
    float mine = 0.f, mine2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t wu[4] = { w[r][c].x, w[r][c].y, w[r][c].z, w[r][c].w };
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1 = fmaf(xv[m][c][2 * j], bf_lo(wu[j]), s1); s1 = fmaf(xv[m][c][2 * j + 1], bf_hi(wu[j]), s1); }
                if (EPI == EPI_SWIGLU) {
                    const uint32_t vu[4] = { w2[r][c].x, w2[r][c].y, w2[r][c].z, w2[r][c].w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s2 = fmaf(xv[m][c][2 * j], bf_lo(vu[j]), s2); s2 = fmaf(xv[m][c][2 * j + 1], bf_hi(vu[j]), s2); }
                }
            }
            s1 = wave_sum(s1);
            if (lane == m * RW + r) mine = s1;
            if (EPI == EPI_SWIGLU) { s2 = wave_sum(s2); if (lane == m * RW + r) mine2 = s2; }
        }
    KP_MARK(11);
    if (lane < MT * RW) {
        const int m = lane / RW, r = lane % RW, n = n0 + r;
        if (m < M && n < N) {
            float o;
            if (EPI == EPI_STORE) o = mine;
            else if (EPI == EPI_RESIDUAL) o = epi_in + mine;
            else if (EPI == EPI_SWIGLU) o = silu_f(mine) * mine2;
            else if (EPI == EPI_BIAS) o = mine + epi_in;
            else o = silu_f(mine + epi_in);
            a.out[(size_t)m * a.ldo + n] = o;
        }
    }
}

static int gemv1_rw(const GemvArgs& a) {
    int rw = a.N / 1024;
    if (rw < 1) rw = 1;
    if (rw > 4) rw = 4;
    if (a.epi == EPI_SWIGLU && rw > 3) rw = 3;
    if (a.K > 3072 && rw > 2) rw = 2;   // 12 chunks per row: two rows of weights + the activation row fill the register file
    return rw;
}
bool gemv_fast_path(const GemvArgs& a) {
    if (a.M < 1 || a.M > 2) return false;
    if (a.K != 1024 && a.K != 2048 && a.K != 3072 && !(a.K == 6144 && a.M == 1)) return false;   // 6144: the 1.7B talker's down projection
    if (a.ldx % 4 != 0 && !a.po) return false;
    const bool norm = a.gamma != nullptr, comb = a.po != nullptr;
    if (comb && (norm || a.epi != EPI_RESIDUAL || a.pd % 8 != 0 || a.pheads * a.pd != a.K)) return false;
    if (norm && a.epi != EPI_STORE && a.epi != EPI_SWIGLU) return false;
    return true;
}

template <int MT, int NCH, int RW>
static void gemv1_launch(const GemvArgs& a, int grid, hipStream_t s) {
    const bool norm = a.gamma != nullptr, comb = a.po != nullptr;
#define Q3_G1(EPI, NORM, COMB) hipLaunchKernelGGL((k_gemv1<MT, NCH, RW, EPI, NORM, COMB>), dim3(grid), dim3(256), 0, s, a.W, a.W2, a.x, a.gamma, \
        (a.epi == EPI_RESIDUAL ? a.res : a.bias), a.N, a.M, a.ldx, a.ldres, (int)a.nt, a)
    if (comb) { Q3_G1(EPI_RESIDUAL, false, true); return; }
    switch (a.epi) {
    case EPI_STORE: if (norm) Q3_G1(EPI_STORE, true, false); else Q3_G1(EPI_STORE, false, false); break;
    case EPI_SWIGLU:   // gemv1_rw never gives a SwiGLU launch four rows per wave (two weight matrices: the register file); not instantiated
        if constexpr (RW <= 3) { if (norm) Q3_G1(EPI_SWIGLU, true, false); else Q3_G1(EPI_SWIGLU, false, false); }
        else throw Error("gemv: SwiGLU fast path takes at most 3 rows per wave");
        break;
    case EPI_RESIDUAL: Q3_G1(EPI_RESIDUAL, false, false); break;
    case EPI_BIAS: Q3_G1(EPI_BIAS, false, false); break;
    default: Q3_G1(EPI_BIAS_SILU, false, false); break;
    }
#undef Q3_G1
}
template <int MT, int NCH>
static void gemv1_rwsel(const GemvArgs& a, hipStream_t s) {
    const int rw = gemv1_rw(a);
    const int grid = (a.N + 4 * rw - 1) / (4 * rw);
    if (rw == 1) gemv1_launch<MT, NCH, 1>(a, grid, s);
    else if (rw == 2) gemv1_launch<MT, NCH, 2>(a, grid, s);
    else if constexpr (NCH <= 6) {   // K = 6144 stops at two rows per wave (gemv1_rw): the wider variants would spill and are not instantiated
        if (rw == 3) gemv1_launch<MT, NCH, 3>(a, grid, s);
        else gemv1_launch<MT, NCH, 4>(a, grid, s);
    } else throw Error("gemv: K = 6144 fast path takes at most 2 rows per wave");
}
template <int MT>
static void gemv1_nch(const GemvArgs& a, hipStream_t s) {
    if (a.K == 1024) gemv1_rwsel<MT, 2>(a, s);
    else if (a.K == 2048) gemv1_rwsel<MT, 4>(a, s);
    else if (a.K == 3072) gemv1_rwsel<MT, 6>(a, s);
    else if constexpr (MT == 1) gemv1_rwsel<1, 12>(a, s);
    else throw Error("gemv: K = 6144 fast path is single-row");
}

void launch_gemv(const GemvArgs& a0, hipStream_t s) {
    if (gemv_fast_path(a0)) {
        if (a0.M == 1) gemv1_nch<1>(a0, s); else gemv1_nch<2>(a0, s);
        return;
    }
    if (gemv16_ok(a0)) { launch_gemv16(a0, s); return; }
    if (a0.po) throw Error("gemv: attention-partials prologue needs the fast path (combine separately)");
    if (a0.K % 8 != 0 || a0.ldx % 4 != 0) throw Error("gemv: K must be a multiple of 8 and ldx of 4");
    if ((a0.epi == EPI_RESIDUAL || a0.epi == EPI_BIAS || a0.epi == EPI_BIAS_SILU) && a0.gamma) throw Error("gemv: norm+epilogue combination not built");
    if (a0.M <= 0 || a0.N <= 0) return;
    for (int m0 = 0; m0 < a0.M; m0 += 8) { // M > 8: row chunks re-stream the weights from L2/MALL
        GemvArgs a = a0;
        a.M = a0.M - m0 < 8 ? a0.M - m0 : 8;
        a.x = a0.x + (size_t)m0 * a0.ldx;
        a.out = a0.out + (size_t)m0 * a0.ldo;
        if (a0.res) a.res = a0.res + (size_t)m0 * a0.ldres;
        if (a0.xn_out) a.xn_out = a0.xn_out + (size_t)m0 * a0.ld_xn;
        if (a.M == 1) gemv_r<1>(a, s);
        else if (a.M == 2) gemv_r<2>(a, s);
        else if (a.M <= 4) gemv_r<4>(a, s);
        else gemv_r<8>(a, s);
    }
}

// ================================================================================================
// k_attn — decode attention over the paged fp32 KV cache.  Workgroup = (kv head, new token x split,
// batch row).  A split covers `chunk` cache tokens; 16 lane-groups of 16 lanes each own every 16th
// token and keep a private online softmax; U tokens per group are loaded before any is consumed
// (the first batch before the q/k RMSNorm+RoPE prologue), so a split is ~one memory latency deep.
// Output: normalised rows (n_splits == 1, po == null) or un-normalised partials (m, l, sum p*v)
// that the o_proj GEMV prologue / k_attn_combine merge.
// ================================================================================================
#define ATT_MAX_NEW 16
#define ATT_MAX_GRP 4

// KVB: the cache holds bf16 (Q3TTS_FLAG_KV_BF16).  K / V rows are rounded to bf16 (RNE) where they enter the cache AND where this step
// uses them itself (the LDS copies of the new tokens), so every reader sees the same values — the oracle rounds at the same point
// (oracle/q3_oracle.c dec_forward, kv_bf16).  Ids are NOT bit-exact against the oracle in this mode: rounding is a discontinuity, the two
// implementations' logits sit ~4e-3 apart and the ids agree up to the first decision whose margin is below that noise; what IS bit-exact
// is this 16-bit storage path against fp32 storage of the same rounded rows (Q3TTS_FLAG_KV_ROUND_BF16).  Math stays fp32.
static __device__ __forceinline__ float bf16_round_f(float f) {
    uint32_t u = __float_as_uint(f);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}
// Deferred RMSNorm (split-K seam, GemmArgs::seam): the projection's input planes were gamma * x, NOT normalised; the row's 1 / rms comes
// from the producer's per-(row, 64-column tile) sums of squares.  Lane t holds partial t (loaded with the kernel's other operands);
// the sum runs in tile order through readlanes (wave-uniform, deterministic).  `have` false: 1.0f (the product is then exact).
static __device__ __forceinline__ float ssq_row_scale(float part, int nt, int K, float eps, bool have) {
    float tot = 0.f;
    for (int t = 0; t < nt; ++t) tot += lane_bcast(part, t);
    const float r = 1.0f / sqrtf(tot / (float)K + eps);
    return have ? r : 1.0f;
}
template <int D, int U, int G, bool IDENT = false, bool KVB = false>
__global__ __launch_bounds__(256) void k_attn(const int* ppage_table, const int* ppos_dev, const float* pqkv, const float* pkcache, const float* pvcache,
                                               const float* pcos, const float* psin, int ppos_scalar, int pn_splits, AttnArgs a) {
    // leading scalars: preloaded into SGPRs, so the first memory round (page ids, position) leaves at once (see k_gemv1)
    constexpr int EPL = D / 16;
    constexpr int HALF = D / 2;
    const int kvh = blockIdx.x, bi = blockIdx.z;
    const int S = pn_splits;
    const int inew = blockIdx.y / S, split = blockIdx.y % S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: pointer selects stay scalar
    const int grp = a.nq / a.nkv;
    const int slot = a.slot_map ? a.slot_map[bi] : a.slot_offset + bi;
    const int row = bi * a.n_new + inew;
    const int pshift = a.page_shift, page_tokens = 1 << pshift;

    __shared__ float q_s[G][D];
    __shared__ float knew[ATT_MAX_NEW][D];
    __shared__ float vnew[ATT_MAX_NEW][D];
    __shared__ float cm[16][G], cl[16][G];
    __shared__ float co[16][G][D];

    KP_MARK(24);
    // ---- round 1: position of new token 0 and the (at most 4) page ids this split can touch ----
    const int* pt = ppage_table + (size_t)slot * a.pages_per_slot;
    const int pbase = (split * a.chunk) >> pshift;
    int pg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pi = pbase + q < a.pages_per_slot ? pbase + q : a.pages_per_slot - 1;
        // IDENT (code predictor: one fixed page per slot): no table read, so the K/V batch is not a memory round behind the page ids
        pg[q] = IDENT ? slot * a.pages_per_slot + pi : pt[pi];
    }
    int base = ppos_scalar;
    if (ppos_dev) base = ppos_dev[slot];
    const int pos = base + inew;
    __builtin_amdgcn_sched_barrier(0);

    KP_MARK(25);
    // token range of this split
    int lo = split * a.chunk;
    int hi = lo + a.chunk < pos + 1 ? lo + a.chunk : pos + 1;
    if (a.window > 0 && pos - a.window + 1 > lo) lo = pos - a.window + 1;
    if (lo >= hi) return; // empty split: the combiner derives the active split count from pos
    const int n_cache = a.new_from_raw ? base : pos + 1;   // tokens < n_cache live in the cache
    const int cend = hi < n_cache ? hi : n_cache;           // cache tokens [lo, cend)
    const int nlo = lo > base ? lo : base;                  // new tokens [nlo, hi) come from qkv (new_from_raw)

    auto cache_off = [&](int t) -> size_t {
        const int q = (t >> pshift) - pbase;
        // IDENT: the page id is arithmetic, so a split may span any number of pages (one split over the whole context at large batch)
        const int page = IDENT ? slot * a.pages_per_slot + (t >> pshift) : (q <= 0 ? pg[0] : (q == 1 ? pg[1] : (q == 2 ? pg[2] : pg[3])));
        return ((((size_t)page * a.n_layers + a.layer) * a.nkv + kvh) * page_tokens + (t & (page_tokens - 1))) * D;
    };

    // ---- round 2: the first K/V batch and every prologue operand, all in flight together ----
    const int tg = wave * 4 + (lane >> 4), sub = lane & 15;
    float kr[U][EPL], vr[U][EPL];
    auto load_batch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int t = t0 + 16 * u;           // clamp instead of branching: a conditional load costs a serial round trip
            t = t < cend ? t : cend - 1;
            t = t > 0 ? t : 0;
            const size_t off = cache_off(t) + sub * EPL;
            if (KVB) {   // EPL bf16 values per lane: one 16-byte (EPL 8), 8-byte (4) or 2-byte (1) load per row
                const uint16_t* kc16 = reinterpret_cast<const uint16_t*>(pkcache) + off;
                const uint16_t* vc16 = reinterpret_cast<const uint16_t*>(pvcache) + off;
                if (EPL == 8) {
                    const u32x4 k4 = *reinterpret_cast<const u32x4*>(kc16), v4 = *reinterpret_cast<const u32x4*>(vc16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        kr[u][2 * e] = bf_lo(k4[e]); kr[u][2 * e + 1] = bf_hi(k4[e]);
                        vr[u][2 * e] = bf_lo(v4[e]); vr[u][2 * e + 1] = bf_hi(v4[e]);
                    }
                } else if (EPL == 4) {
                    const uint2 k2 = *reinterpret_cast<const uint2*>(kc16), v2 = *reinterpret_cast<const uint2*>(vc16);
                    kr[u][0] = bf_lo(k2.x); kr[u][1] = bf_hi(k2.x); kr[u][2] = bf_lo(k2.y); kr[u][3] = bf_hi(k2.y);
                    vr[u][0] = bf_lo(v2.x); vr[u][1] = bf_hi(v2.x); vr[u][2] = bf_lo(v2.y); vr[u][3] = bf_hi(v2.y);
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) { kr[u][e] = bf_lo(kc16[e]); vr[u][e] = bf_lo(vc16[e]); }
                }
            } else
            if (EPL >= 4) {
#pragma unroll
                for (int e = 0; e < EPL; e += 4) {
                    const float4 k4 = *reinterpret_cast<const float4*>(pkcache + off + e);
                    const float4 v4 = *reinterpret_cast<const float4*>(pvcache + off + e);
                    kr[u][e] = k4.x; kr[u][e + 1] = k4.y; kr[u][e + 2] = k4.z; kr[u][e + 3] = k4.w;
                    vr[u][e] = v4.x; vr[u][e + 1] = v4.y; vr[u][e + 2] = v4.z; vr[u][e + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) { kr[u][e] = pkcache[off + e]; vr[u][e] = pvcache[off + e]; }
            }
        }
    };
    // ---- 1. q heads of this group and this split's new keys: RMSNorm + RoPE; own K/V appended to the cache.
    // Operand loads of each wave's first vector are issued BEFORE the K/V batch (vmcnt is in-order), so the
    // prologue math runs while the batch is still in flight. ----
    const int jlo = nlo - base;                    // first new token of this split
    const int nnew = a.new_from_raw && hi > nlo ? hi - nlo : 0;
    const int nvec = grp + nnew;
    const int hl = lane < HALF ? lane : 0;         // clamped lane: loads stay unconditional
    struct VecOps { float x0, x1, v0, v1, n0, n1, cs, sn; };
    auto load_vec = [&](int v) -> VecOps {
        const bool is_q = v < grp;
        const int j = is_q ? inew : jlo + (v - grp);
        const float* rowp = pqkv + (size_t)(bi * a.n_new + j) * a.ld_qkv;
        const float* src = rowp + (is_q ? (kvh * grp + v) * D : (a.nq + kvh) * D);
        const float* vs = rowp + (a.nq + a.nkv + kvh) * D;
        const int p = base + j;
        VecOps r;
        // split-K partial slabs of the QKV projection (at most 4): every load first, summed in slab order — a runtime-count
        // load-then-add loop costs one L2 round trip per slab
        float px0[4], px1[4], pv0[4], pv1[4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const size_t so = (size_t)(sb < a.qkv_nslab ? sb : 0) * a.qkv_slab_stride;
            px0[sb] = src[so + hl]; px1[sb] = src[so + hl + HALF];
            pv0[sb] = vs[so + hl]; pv1[sb] = vs[so + hl + HALF];
        }
        // deferred RMSNorm of the projection's input rows (address select: the load stays unconditional)
        const bool have_ssq = a.ssq_in != nullptr;
        const int snt = have_ssq ? a.ssq_nt : 1;
        const float spart = (have_ssq ? a.ssq_in + (size_t)(bi * a.n_new + j) * a.ssq_nt : src)[lane < snt ? lane : 0];   // fallback address: the row itself (rope tables may be null: codec)
        r.x0 = px0[0]; r.x1 = px1[0]; r.v0 = pv0[0]; r.v1 = pv1[0];
#pragma unroll
        for (int sb = 1; sb < 4; ++sb)
            if (sb < a.qkv_nslab) { r.x0 += px0[sb]; r.x1 += px1[sb]; r.v0 += pv0[sb]; r.v1 += pv1[sb]; }
        const float rsc = ssq_row_scale(spart, snt, a.ssq_K, a.ssq_eps, have_ssq);
        r.x0 *= rsc; r.x1 *= rsc; r.v0 *= rsc; r.v1 *= rsc;
        const float* nw = is_q ? a.q_norm : a.k_norm;
        r.n0 = 1.f; r.n1 = 1.f; r.cs = 1.f; r.sn = 0.f;
        if (a.new_from_raw) {
            if (nw != nullptr) { r.n0 = nw[hl]; r.n1 = nw[hl + HALF]; }
            r.cs = pcos[(size_t)p * HALF + hl]; r.sn = psin[(size_t)p * HALF + hl];
        }
        return r;
    };
    auto finish_vec = [&](int v, VecOps r) {
        const bool is_q = v < grp;
        const int j = is_q ? inew : jlo + (v - grp);
        const int p = base + j;
        float x0 = lane < HALF ? r.x0 : 0.f, x1 = lane < HALF ? r.x1 : 0.f;
        if (!a.new_from_raw) { // q was roped by k_rope_store
            if (lane < HALF) { q_s[v][lane] = x0; q_s[v][lane + HALF] = x1; }
            return;
        }
        const float* nw = is_q ? a.q_norm : a.k_norm;
        if (nw != nullptr) {
            const float ss = wave_sum(x0 * x0 + x1 * x1);
            const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
            x0 = r.n0 * (x0 * rr); x1 = r.n1 * (x1 * rr);
        }
        if (lane < HALF) {
            float y0 = x0 * r.cs + (-x1) * r.sn;
            float y1 = x1 * r.cs + x0 * r.sn;
            if (is_q) { q_s[v][lane] = y0; q_s[v][lane + HALF] = y1; }
            else {
                float v0 = r.v0, v1 = r.v1;
                if (KVB || a.kv_round) { y0 = bf16_round_f(y0); y1 = bf16_round_f(y1); v0 = bf16_round_f(v0); v1 = bf16_round_f(v1); }
                knew[j][lane] = y0; knew[j][lane + HALF] = y1;
                vnew[j][lane] = v0; vnew[j][lane + HALF] = v1;
                if (j == inew) { // this workgroup owns position `pos`
                    const size_t off = cache_off(p);
                    if (KVB) {
                        uint16_t* kc16 = reinterpret_cast<uint16_t*>(a.kcache); uint16_t* vc16 = reinterpret_cast<uint16_t*>(a.vcache);
                        kc16[off + lane] = (uint16_t)(__float_as_uint(y0) >> 16); kc16[off + lane + HALF] = (uint16_t)(__float_as_uint(y1) >> 16);
                        vc16[off + lane] = (uint16_t)(__float_as_uint(v0) >> 16); vc16[off + lane + HALF] = (uint16_t)(__float_as_uint(v1) >> 16);
                    } else {
                        a.kcache[off + lane] = y0; a.kcache[off + lane + HALF] = y1;
                        a.vcache[off + lane] = v0; a.vcache[off + lane + HALF] = v1;
                    }
                }
            }
        }
    };
    KP_MARK(26);
    const VecOps first = load_vec(wave < nvec ? wave : 0);
    __builtin_amdgcn_sched_barrier(0);
    load_batch(lo + tg);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < nvec) finish_vec(wave, first);
    for (int v = wave + 4; v < nvec; v += 4) finish_vec(v, load_vec(v)); // prefill only (more than 4 vectors)
    __syncthreads();

    KP_MARK(27);
    // ---- 2. online softmax ----
    float qr[G][EPL], o[G][EPL], mrun[G], lrun[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        mrun[h] = -INFINITY; lrun[h] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { qr[h][e] = h < grp ? q_s[h][sub * EPL + e] : 0.f; o[h][e] = 0.f; }
    }
    auto consume = [&](const float (&kv)[EPL], const float (&vv)[EPL]) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
            if (h < grp) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) s = fmaf(qr[h][e], kv[e], s);
                s = row_sum16(s);
                s *= a.scale;
                const float mn = fmaxf(mrun[h], s);
                const float corr = __expf(mrun[h] - mn); // exp(-inf) = 0 on the first token
                const float pw = __expf(s - mn);
                lrun[h] = lrun[h] * corr + pw;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[h][e] = o[h][e] * corr + pw * vv[e];
                mrun[h] = mn;
            }
        }
    };
    for (int t0 = lo + tg; t0 < cend; t0 += 16 * U) {
        if (t0 != lo + tg) load_batch(t0);
        // all U scores first (independent dot products), then one running-max update per head
        float sc[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = t0 + 16 * u < cend;
#pragma unroll
            for (int h = 0; h < G; ++h) {
                float sdot = 0.f;
                if (h < grp) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) sdot = fmaf(qr[h][e], kr[u][e], sdot);
                    sdot = row_sum16(sdot) * a.scale;
                }
                sc[u][h] = valid ? sdot : -INFINITY;
            }
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            if (h < grp) {
                float mn = mrun[h];
#pragma unroll
                for (int u = 0; u < U; ++u) mn = fmaxf(mn, sc[u][h]);   // token u=0 is always valid -> finite
                const float corr = __expf(mrun[h] - mn);
                float l = lrun[h] * corr;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[h][e] *= corr;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float pw = __expf(sc[u][h] - mn);           // exp(-inf) = 0 for padding tokens
                    l += pw;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[h][e] = fmaf(pw, t0 + 16 * u < cend ? vr[u][e] : 0.f, o[h][e]);
                }
                lrun[h] = l;
                mrun[h] = mn;
            }
        }
    }
    for (int t = nlo + tg; t < hi && a.new_from_raw; t += 16) { // this split's new tokens (LDS)
        const int j = t - base;
        float kv[EPL], vv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) { kv[e] = knew[j][sub * EPL + e]; vv[e] = vnew[j][sub * EPL + e]; }
        consume(kv, vv);
    }
    KP_MARK(28);
    // ---- 3. combine the 16 token groups ----
#pragma unroll
    for (int h = 0; h < G; ++h) {
        if (h < grp) {
            if (sub == 0) { cm[tg][h] = mrun[h]; cl[tg][h] = lrun[h]; }
#pragma unroll
            for (int e = 0; e < EPL; ++e) co[tg][h][sub * EPL + e] = o[h][e];
        }
    }
    __syncthreads();
    // the 16 groups' weights exp(m_g - max) once per (group, head) instead of once per output element (round 4: 16 libm expf per thread were
    // ~0.8 us of the launch's 1.3 us tail; same values, same order of sums: bit-identical)
    __shared__ float cw[16][G];
    if (tid < 16 * grp) {
        const int g = tid / grp, h = tid % grp;
        float mxw = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) mxw = fmaxf(mxw, cm[q][h]);
        cw[g][h] = cm[g][h] == -INFINITY ? 0.f : expf(cm[g][h] - mxw);
    }
    __syncthreads();
    for (int idx = tid; idx < grp * D; idx += 256) {
        const int h = idx / D, e = idx % D;
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) mx = fmaxf(mx, cm[g][h]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float w = cw[g][h];
            L += w * cl[g][h];
            O += w * co[g][h][e];
        }
        const int head = kvh * grp + h;
        if (a.po == nullptr) {
            const float o = O / L;
            if (a.out) a.out[(size_t)row * a.ld_out + head * D + e] = o;
            if (a.oh) { // (hi, lo) bf16 planes for the MFMA o_proj: a one-split attention needs no combine launch
                const uint32_t u = __float_as_uint(o);
                const bf16_t hi = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                const float rem = o - __uint_as_float((uint32_t)hi << 16);
                const uint32_t v = __float_as_uint(rem);
                a.oh[(size_t)row * a.ldp + head * D + e] = hi;
                a.ol[(size_t)row * a.ldp + head * D + e] = (bf16_t)((v + 0x7FFFu + ((v >> 16) & 1u)) >> 16);
            }
        } else {
            const size_t pi = ((size_t)row * a.nq + head) * S + split;
            a.po[pi * D + e] = O;
            if (e == 0) { a.pm[pi] = mx; a.pl[pi] = L; }
        }
    }
    KP_MARK(29);
}

// ================================================================================================
// k_attn_stream (round 4) — the talker's decode attention of the BATCHED step at long contexts, where the launch is bound by the KV
// bytes it streams (64 rows x 1041 tokens: 546 MB fp32 / 273 MB bf16 per layer) instead of by its latency chain.  k_attn holds one batch
// of 8 tokens per lane group in ~250 registers — two workgroups per CU, each one memory round deep, nothing in flight while it computes
// or merges.  Here a workgroup = (kv head, split of `chunk` tokens, row) walks its split in batches of 16 U tokens through a two-deep
// register ring: batch j+2 is requested as soon as batch j has been consumed (loads return in issue order, so the wait for batch j
// leaves batch j+1 in flight), the rows stay PACKED in the ring (bf16 cache: 8 registers per token and lane for K and V, converted as
// they are consumed), and the kernel fits 128 registers: four workgroups per CU, each with 2 x 16 KiB of K/V in flight.
//   * one new token per row (decode), raw q/k/v slabs in, same prologue as k_attn: q heads and the new key through RMSNorm + RoPE, the
//     new K/V row appended by the split that holds position `pos` and consumed from LDS;
//   * a batch of 16 U tokens never crosses a 64-token page and a split starts on a page boundary, so the page of a batch is
//     wave-uniform: the load address is {scalar base, 32-bit lane offset};
//   * no load under a runtime condition: a batch past the split's end requests one hot line (every lane the split's first row) instead;
//   * MAPC: lane `sub` of a token's 16 lanes owns dims [8 sub, 8 sub + 8) — one 16-byte piece of a bf16 row, two adjacent pieces of an
//     fp32 row.  !MAPC (fp32 only): dims [4 sub, +4) and [64 + 4 sub, +4), so that each load instruction of the wave covers whole
//     256-byte runs.  Q3TTS_FLAG_KV_ROUND_BF16 keeps MAPC: its sums then associate exactly like the bf16 cache's (bit-for-bit test);
//   * output: un-normalised partials (m, l, sum p v) per split for k_attn_combine, or — one split — the normalised planes / rows.
// Reference semantics: run_decode's attention over the grown KVCache, /root/reference/src/tts_onnx.cpp:667-732, tts_onnx.h:108-115.
// ================================================================================================
template <int G, bool KVB, bool MAPC, bool IDENT, int U, int WPE>
__global__ __launch_bounds__(256, WPE) void k_attn_stream(const int* ppage_table, const int* ppos_dev, const float* pqkv, const float* pkcache, const float* pvcache,
                                                         const float* pcos, const float* psin, int pn_splits, int pchunk, AttnArgs a) {
    constexpr int D = 128, HALF = 64, EPL = 8;
    constexpr int KR = KVB ? 1 : 2;                 // 16-byte pieces per (row, lane)
    constexpr int ESZ = KVB ? 2 : 4;
    constexpr int BT = 16 * U;                      // tokens per batch
    constexpr unsigned PIECE1 = MAPC ? 16u : 256u;  // byte distance of a lane's second piece (fp32)
    static_assert(64 % BT == 0, "a batch must not cross a KV page");
    static_assert(!KVB || MAPC, "the bf16 cache has one piece per lane");
    const int kvh = blockIdx.x, split = blockIdx.y, bi = blockIdx.z;
    const int S = pn_splits;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = a.slot_offset + bi, row = bi;
    const int tg = wave * 4 + (lane >> 4), sub = lane & 15;

    __shared__ float q_s[G][D];
    __shared__ float knew[D], vnew[D];
    __shared__ float cm[16][G], cl[16][G];
    __shared__ float co[16][G][D];

    // ---- round 1: the row's position ----
    const int lo = split * pchunk;
    const int pbase = lo >> 6;
    // Pooled cache (!IDENT): a batch's page id is read from the slot's table row where the batch is requested — a scalar load from a
    // line the scalar cache holds after the first one.  (Page ids kept in a private array, or in four scalars behind a select chain,
    // made hipcc copy the whole argument block to scratch: 700-1100 bytes of private segment, every field reloaded from there.)
    const int* ptr = ppage_table + (size_t)slot * a.pages_per_slot;
    const int plast = a.pages_per_slot - 1;
    int base = a.pos_scalar;
    if (ppos_dev) base = ppos_dev[slot];
    __builtin_amdgcn_sched_barrier(0);
    const int hi = lo + pchunk < base + 1 ? lo + pchunk : base + 1;
    if (lo >= hi) return;                            // empty split: the combiner derives the active split count from pos
    const int cend = hi < base ? hi : base;          // cache tokens [lo, cend); position `base` is the new token
    const bool has_new = hi > base;
    const int nbat = (cend - lo + BT - 1) / BT;
    const int last = cend - 1 - lo;                  // last cached token of the split, relative to lo (< 0: none)

    const size_t unit = (size_t)64 * D * ESZ;        // bytes of one (page, layer, kv head) block
    const unsigned lane_b0 = (unsigned)((MAPC ? 8 * sub : 4 * sub) * ESZ);
    u32x4 kq[2][U][KR], vq[2][U][KR];
    auto issue = [&](auto bufc, int j) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const bool live = j < nbat;
        const int jj = live ? j : 0;
        const int pq = (jj * BT) >> 6;
        const int page = IDENT ? slot * a.pages_per_slot + pbase + pq : ptr[pbase + pq < plast ? pbase + pq : plast];
        const size_t ub = (((size_t)page * a.n_layers + a.layer) * a.nkv + kvh) * unit;
        const char* kb = reinterpret_cast<const char*>(pkcache) + ub;
        const char* vb = reinterpret_cast<const char*>(pvcache) + ub;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int tl = jj * BT + 16 * u + tg;
            tl = tl < last ? tl : last;
            tl = live ? tl : 0;                       // past the end: every lane group asks for the split's first row (one hot line)
            tl = tl > 0 ? tl : 0;
            const unsigned off = (unsigned)((tl & 63) * D * ESZ) + lane_b0;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const u32x4* kp = reinterpret_cast<const u32x4*>(kb + off + r * PIECE1);
                const u32x4* vp = reinterpret_cast<const u32x4*>(vb + off + r * PIECE1);
                kq[buf][u][r] = __builtin_nontemporal_load(kp);   // read once per step and far larger than L2 + MALL: nt (A/B below)
                vq[buf][u][r] = __builtin_nontemporal_load(vp);
            }
        }
    };
    auto unpack = [&](const u32x4 (&raw)[KR], float (&f)[EPL]) __attribute__((always_inline)) {
        if (KVB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(raw[0][e]); f[2 * e + 1] = bf_hi(raw[0][e]); }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = __uint_as_float(raw[0][e]); f[4 + e] = __uint_as_float(raw[KR - 1][e]); }
        }
    };
    auto dim_of = [&](int e) __attribute__((always_inline)) -> int { return MAPC ? 8 * sub + e : (e < 4 ? 4 * sub + e : HALF + 4 * sub + (e - 4)); };

    // ---- round 2: the prologue's operands first (vmcnt is in order: their math then runs under the K/V batches), then two batches ----
    constexpr int NVEC = G + 1;                     // G query heads + the new key
    const int hl = lane < HALF ? lane : 0;
    struct VecOps { float x0, x1, v0, v1, n0, n1, cs, sn; };
    auto load_vec = [&](int v) __attribute__((always_inline)) -> VecOps {
        const bool is_q = v < G;
        const float* rowp = pqkv + (size_t)row * a.ld_qkv;
        const float* src = rowp + (is_q ? (kvh * G + v) * D : (a.nq + kvh) * D);
        const float* vs = rowp + (a.nq + a.nkv + kvh) * D;
        VecOps r;
        float px0[4], px1[4], pv0[4], pv1[4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const size_t so = (size_t)(sb < a.qkv_nslab ? sb : 0) * a.qkv_slab_stride;
            px0[sb] = src[so + hl]; px1[sb] = src[so + hl + HALF];
            pv0[sb] = vs[so + hl]; pv1[sb] = vs[so + hl + HALF];
        }
        const bool have_ssq = a.ssq_in != nullptr;
        const int snt = have_ssq ? a.ssq_nt : 1;
        const float spart = (have_ssq ? a.ssq_in + (size_t)row * a.ssq_nt : src)[lane < snt ? lane : 0];
        r.x0 = px0[0]; r.x1 = px1[0]; r.v0 = pv0[0]; r.v1 = pv1[0];
#pragma unroll
        for (int sb = 1; sb < 4; ++sb)
            if (sb < a.qkv_nslab) { r.x0 += px0[sb]; r.x1 += px1[sb]; r.v0 += pv0[sb]; r.v1 += pv1[sb]; }
        const float rsc = ssq_row_scale(spart, snt, a.ssq_K, a.ssq_eps, have_ssq);
        r.x0 *= rsc; r.x1 *= rsc; r.v0 *= rsc; r.v1 *= rsc;
        const float* nw = is_q ? a.q_norm : a.k_norm;
        r.n0 = 1.f; r.n1 = 1.f;
        if (nw != nullptr) { r.n0 = nw[hl]; r.n1 = nw[hl + HALF]; }
        r.cs = pcos[(size_t)base * HALF + hl]; r.sn = psin[(size_t)base * HALF + hl];
        return r;
    };
    auto finish_vec = [&](int v, VecOps r) __attribute__((always_inline)) {
        const bool is_q = v < G;
        float x0 = lane < HALF ? r.x0 : 0.f, x1 = lane < HALF ? r.x1 : 0.f;
        const float* nw = is_q ? a.q_norm : a.k_norm;
        if (nw != nullptr) {
            const float ss = wave_sum(x0 * x0 + x1 * x1);
            const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
            x0 = r.n0 * (x0 * rr); x1 = r.n1 * (x1 * rr);
        }
        if (lane < HALF) {
            float y0 = x0 * r.cs + (-x1) * r.sn;
            float y1 = x1 * r.cs + x0 * r.sn;
            if (is_q) { q_s[v][lane] = y0; q_s[v][lane + HALF] = y1; }
            else if (has_new) {
                float v0 = r.v0, v1 = r.v1;
                if (KVB || a.kv_round) { y0 = bf16_round_f(y0); y1 = bf16_round_f(y1); v0 = bf16_round_f(v0); v1 = bf16_round_f(v1); }
                knew[lane] = y0; knew[lane + HALF] = y1;
                vnew[lane] = v0; vnew[lane + HALF] = v1;
                            const int page = IDENT ? slot * a.pages_per_slot + (base >> 6) : ptr[(base >> 6) < plast ? (base >> 6) : plast];
                const size_t off = ((((size_t)page * a.n_layers + a.layer) * a.nkv + kvh) * 64 + (base & 63)) * D;
                if (KVB) {
                    uint16_t* kc16 = reinterpret_cast<uint16_t*>(a.kcache); uint16_t* vc16 = reinterpret_cast<uint16_t*>(a.vcache);
                    kc16[off + lane] = (uint16_t)(__float_as_uint(y0) >> 16); kc16[off + lane + HALF] = (uint16_t)(__float_as_uint(y1) >> 16);
                    vc16[off + lane] = (uint16_t)(__float_as_uint(v0) >> 16); vc16[off + lane + HALF] = (uint16_t)(__float_as_uint(v1) >> 16);
                } else {
                    a.kcache[off + lane] = y0; a.kcache[off + lane + HALF] = y1;
                    a.vcache[off + lane] = v0; a.vcache[off + lane + HALF] = v1;
                }
            }
        }
    };
    const VecOps first = load_vec(wave < NVEC ? wave : 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(std::integral_constant<int, 0>{}, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < NVEC) finish_vec(wave, first);
    for (int v = wave + 4; v < NVEC; v += 4) finish_vec(v, load_vec(v));   // G = 4: five vectors over four waves
    __builtin_amdgcn_sched_barrier(0);
    issue(std::integral_constant<int, 1>{}, 1);       // behind the prologue: its operands' registers are free again
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    // ---- online softmax over the split, two batches in flight ----
    float qr[G][EPL], o[G][EPL], mrun[G], lrun[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        mrun[h] = -INFINITY; lrun[h] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { qr[h][e] = q_s[h][dim_of(e)]; o[h][e] = 0.f; }
    }
    auto compute = [&](auto bufc, int j) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        float pw[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = j * BT + 16 * u + tg <= last;
            float kf[EPL];
            unpack(kq[buf][u], kf);
#pragma unroll
            for (int h = 0; h < G; ++h) {
                float sdot = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) sdot = fmaf(qr[h][e], kf[e], sdot);
                sdot = row_sum16(sdot) * a.scale;
                pw[u][h] = valid ? sdot : -INFINITY;
            }
            __builtin_amdgcn_sched_barrier(0);   // one token's row unpacked at a time: left alone hipcc unpacks the whole batch first and spills
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float mn = mrun[h];
#pragma unroll
            for (int u = 0; u < U; ++u) mn = fmaxf(mn, pw[u][h]);
            const float mu = mn == -INFINITY ? 0.f : mn;   // a lane group without a valid token yet: exp(-inf - 0) = 0 everywhere, no NaN
            const float corr = __expf(mrun[h] - mu);
            float l = lrun[h] * corr;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[h][e] *= corr;
#pragma unroll
            for (int u = 0; u < U; ++u) { pw[u][h] = __expf(pw[u][h] - mu); l += pw[u][h]; }
            lrun[h] = l;
            mrun[h] = mn;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float vf[EPL];
            unpack(vq[buf][u], vf);        // a clamped token's row is a real row (finite), its weight is 0
#pragma unroll
            for (int h = 0; h < G; ++h)
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[h][e] = fmaf(pw[u][h], vf[e], o[h][e]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int j = 0; j < nbat; j += 2) {
        compute(std::integral_constant<int, 0>{}, j);
        issue(std::integral_constant<int, 0>{}, j + 2);
        compute(std::integral_constant<int, 1>{}, j + 1);   // past the end: every token invalid, the state does not move
        issue(std::integral_constant<int, 1>{}, j + 3);
    }
    if (has_new && tg == 0) {   // the new token, from LDS
        float kf[EPL], vf[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) { kf[e] = knew[dim_of(e)]; vf[e] = vnew[dim_of(e)]; }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s = fmaf(qr[h][e], kf[e], s);
            s = row_sum16(s) * a.scale;
            const float mn = fmaxf(mrun[h], s);
            const float corr = __expf(mrun[h] - mn);
            const float p1 = __expf(s - mn);
            lrun[h] = lrun[h] * corr + p1;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[h][e] = o[h][e] * corr + p1 * vf[e];
            mrun[h] = mn;
        }
    }
    // ---- the 16 lane groups meet in LDS ----
#pragma unroll
    for (int h = 0; h < G; ++h) {
        if (sub == 0) { cm[tg][h] = mrun[h]; cl[tg][h] = lrun[h]; }
#pragma unroll
        for (int e = 0; e < EPL; ++e) co[tg][h][dim_of(e)] = o[h][e];
    }
    __syncthreads();
    __shared__ float cw[16][G];            // exp(m_g - max) once per (group, head), as in k_attn
    if (tid < 16 * G) {
        const int g = tid / G, h = tid % G;
        float mxw = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) mxw = fmaxf(mxw, cm[q][h]);
        cw[g][h] = cm[g][h] == -INFINITY ? 0.f : expf(cm[g][h] - mxw);
    }
    __syncthreads();
    for (int idx = tid; idx < G * D; idx += 256) {
        const int h = idx / D, e = idx % D;
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) mx = fmaxf(mx, cm[g][h]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float w = cw[g][h];
            L += w * cl[g][h];
            O += w * co[g][h][e];
        }
        const int head = kvh * G + h;
        if (a.po == nullptr) {
            const float ov = O / L;
            if (a.out) a.out[(size_t)row * a.ld_out + head * D + e] = ov;
            if (a.oh) {
                const uint32_t u = __float_as_uint(ov);
                const bf16_t hb = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                const float rem = ov - __uint_as_float((uint32_t)hb << 16);
                const uint32_t v = __float_as_uint(rem);
                a.oh[(size_t)row * a.ldp + head * D + e] = hb;
                a.ol[(size_t)row * a.ldp + head * D + e] = (bf16_t)((v + 0x7FFFu + ((v >> 16) & 1u)) >> 16);
            }
        } else {
            const size_t pi = ((size_t)row * a.nq + head) * S + split;
            a.po[pi * D + e] = O;
            if (e == 0) { a.pm[pi] = mx; a.pl[pi] = L; }
        }
    }
}

// ================================================================================================
// k_attn_tiny — the code predictor's attention in the batched step: at most 32 cached tokens (one fixed page per slot), 1 or 2 new
// tokens, position a launch argument.  k_attn spreads one (kv head, row) over 256 threads — 16 lane groups, two block barriers, an LDS
// merge of 16 partial softmaxes — for a context of 1..17 tokens: 4.7 us in-kernel, 75 launches per step.  Here ONE WAVE owns a
// (kv head, row): every load is issued up front (raw q / k / v rows of the split-K slabs, norm and RoPE operands, the page's K rows as
// (token, 32-dim chunk) per lane and V rows as (dim, dim + 64) per lane), q goes through the wave's own LDS slice to change layout, the
// softmax is a handful of DPP reductions, p_t reaches the P.V loop through v_readlane.  No block barrier; two waves per workgroup.
// Same arithmetic as k_attn for the new token (RMSNorm, RoPE, cache append); the softmax sums associate differently (fp32, ~1e-7).
// ================================================================================================
static __device__ __forceinline__ void wave_lds_sync();
template <int G, int NN>
__global__ __launch_bounds__(128) void k_attn_tiny(const float* pqkv, const float* pkcache, const float* pvcache, const float* pcos, const float* psin,
                                                    int pbase, AttnArgs a) {
    constexpr int D = 128, HALF = 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.x * 2 + wave;
    if (pair >= a.nb * a.nkv) return;                                   // wave-uniform
    KP_MARK(12);
    const int bi = pair / a.nkv, kvh = pair - bi * a.nkv;
    const int slot = a.slot_offset + bi, base = pbase;
    const int PT = 1 << a.page_shift;
    const size_t cbase = ((((size_t)slot * a.pages_per_slot) * a.n_layers + a.layer) * a.nkv + kvh) * (size_t)PT * D;
    const float* kc = pkcache + cbase;
    const float* vc = pvcache + cbase;
    __shared__ __attribute__((aligned(16))) float q_sh[2][NN][G][D];
    float (*q_s)[G][D] = q_sh[wave];

    // ---- every load of the launch, before any use ----
    float qx0[NN][G][4], qx1[NN][G][4], kx0[NN][4], kx1[NN][4], vx0[NN][4], vx1[NN][4], cs[NN], sn[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const float* rowp = pqkv + (size_t)(bi * NN + j) * a.ld_qkv;
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const size_t so = (size_t)(sb < a.qkv_nslab ? sb : 0) * a.qkv_slab_stride;
#pragma unroll
            for (int h = 0; h < G; ++h) { qx0[j][h][sb] = rowp[so + (kvh * G + h) * D + lane]; qx1[j][h][sb] = rowp[so + (kvh * G + h) * D + lane + HALF]; }
            kx0[j][sb] = rowp[so + (a.nq + kvh) * D + lane]; kx1[j][sb] = rowp[so + (a.nq + kvh) * D + lane + HALF];
            vx0[j][sb] = rowp[so + (a.nq + a.nkv + kvh) * D + lane]; vx1[j][sb] = rowp[so + (a.nq + a.nkv + kvh) * D + lane + HALF];
        }
        cs[j] = pcos[(size_t)(base + j) * HALF + lane]; sn[j] = psin[(size_t)(base + j) * HALF + lane];
    }
    // deferred RMSNorm of the projection's input rows (split-K seam): per-tile partial sums of squares of each new row, one per lane
    const bool have_ssq = a.ssq_in != nullptr;
    const int snt = have_ssq ? a.ssq_nt : 1;
    float spart[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) spart[j] = (have_ssq ? a.ssq_in + (size_t)(bi * NN + j) * a.ssq_nt : pqkv + (size_t)(bi * NN + j) * a.ld_qkv)[lane < snt ? lane : 0];
    const float* qnw = a.q_norm ? a.q_norm : pcos;      // address select: the loads stay unconditional, the values are replaced below
    const float* knw = a.k_norm ? a.k_norm : pcos;
    const float qn0 = qnw[lane], qn1 = qnw[lane + HALF], kn0 = knw[lane], kn1 = knw[lane + HALF];
    // cached K as (token = lane / 4, 32-dim chunk = lane % 4), cached V as (dim = lane, lane + 64) per token; tokens past `base` repeat the last one
    const int tk = lane >> 2, ch = lane & 3;
    const int last = base > 0 ? base - 1 : 0;
    float4 kr[8];
    {
        const int t = tk < base ? tk : last;
#pragma unroll
        for (int e = 0; e < 8; ++e) kr[e] = *reinterpret_cast<const float4*>(kc + (size_t)t * D + ch * 32 + e * 4);
    }
    float vr0[16], vr1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int tt = t < base ? t : last;
        vr0[t] = vc[(size_t)tt * D + lane]; vr1[t] = vc[(size_t)tt * D + lane + HALF];
    }
    __builtin_amdgcn_sched_barrier(0);
    KP_MARK(13);   // every load issued
#pragma unroll
    for (int t = 0; t < 16; ++t) { vr0[t] = t < base ? vr0[t] : 0.f; vr1[t] = t < base ? vr1[t] : 0.f; }   // never-written cache rows may hold anything: 0 x NaN
    KP_MARK(14);   // the cached K / V rows have arrived (the selects above wait for the last V row)

    // ---- new tokens: slab sums (slab order), RMSNorm, RoPE; K / V appended to the cache; q to LDS ----
    float ky0[NN], ky1[NN], vn0[NN], vn1[NN], qy0[NN][G], qy1[NN][G];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const float rsc = ssq_row_scale(spart[j], snt, a.ssq_K, a.ssq_eps, have_ssq);
        auto slabsum = [&](const float (&p)[4]) { float t = p[0];
#pragma unroll
            for (int sb = 1; sb < 4; ++sb) if (sb < a.qkv_nslab) t += p[sb];
            return t * rsc; };
        float k0 = slabsum(kx0[j]), k1 = slabsum(kx1[j]);
        vn0[j] = slabsum(vx0[j]); vn1[j] = slabsum(vx1[j]);
        if (a.k_norm != nullptr) {
            const float ss = wave_sum(k0 * k0 + k1 * k1);
            const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
            k0 = kn0 * (k0 * rr); k1 = kn1 * (k1 * rr);
        }
        ky0[j] = k0 * cs[j] + (-k1) * sn[j];
        ky1[j] = k1 * cs[j] + k0 * sn[j];
        const size_t off = cbase + (size_t)(base + j) * D;
        a.kcache[off + lane] = ky0[j]; a.kcache[off + lane + HALF] = ky1[j];
        a.vcache[off + lane] = vn0[j]; a.vcache[off + lane + HALF] = vn1[j];
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float x0 = slabsum(qx0[j][h]), x1 = slabsum(qx1[j][h]);
            if (a.q_norm != nullptr) {
                const float ss = wave_sum(x0 * x0 + x1 * x1);
                const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
                x0 = qn0 * (x0 * rr); x1 = qn1 * (x1 * rr);
            }
            qy0[j][h] = x0 * cs[j] + (-x1) * sn[j];
            qy1[j][h] = x1 * cs[j] + x0 * sn[j];
            q_s[j][h][lane] = qy0[j][h]; q_s[j][h][lane + HALF] = qy1[j][h];
        }
    }
    wave_lds_sync();
    KP_MARK(15);   // slab sums, norms, RoPE done; q in LDS

    // ---- per new token: scores, softmax, P.V, output ----
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const int row = bi * NN + j;
#pragma unroll
        for (int h = 0; h < G; ++h) {
            // cached tokens 0..15 (and 16..31 below when the context is that long)
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 q4 = *reinterpret_cast<const float4*>(&q_s[j][h][ch * 32 + e * 4]);
                part = fmaf(q4.x, kr[e].x, part); part = fmaf(q4.y, kr[e].y, part); part = fmaf(q4.z, kr[e].z, part); part = fmaf(q4.w, kr[e].w, part);
            }
            part += dpp_f<Q3_DPP_XOR1, 0xF>(0.f, part);
            part += dpp_f<Q3_DPP_XOR2, 0xF>(0.f, part);
            float sc = tk < base ? part * a.scale : -INFINITY;            // all four lanes of a token hold its score
            float scB = -INFINITY;
            float4 krB[8];
            if (base > 16) {                                               // wave-uniform, rare (more than 16 cached tokens)
                const int t = 16 + tk < base ? 16 + tk : last;
                float pb = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    krB[e] = *reinterpret_cast<const float4*>(kc + (size_t)t * D + ch * 32 + e * 4);
                    const float4 q4 = *reinterpret_cast<const float4*>(&q_s[j][h][ch * 32 + e * 4]);
                    pb = fmaf(q4.x, krB[e].x, pb); pb = fmaf(q4.y, krB[e].y, pb); pb = fmaf(q4.z, krB[e].z, pb); pb = fmaf(q4.w, krB[e].w, pb);
                }
                pb += dpp_f<Q3_DPP_XOR1, 0xF>(0.f, pb);
                pb += dpp_f<Q3_DPP_XOR2, 0xF>(0.f, pb);
                scB = 16 + tk < base ? pb * a.scale : -INFINITY;
            }
            // the new tokens up to and including this one
            float sn_[NN];
#pragma unroll
            for (int jn = 0; jn < NN; ++jn) sn_[jn] = jn <= j ? wave_sum(qy0[j][h] * ky0[jn] + qy1[j][h] * ky1[jn]) * a.scale : -INFINITY;
            float m = wave_max(fmaxf(sc, scB));
#pragma unroll
            for (int jn = 0; jn < NN; ++jn) m = fmaxf(m, sn_[jn]);        // the token itself is always there: m is finite
            const float pA = __expf(sc - m), pB = __expf(scB - m);       // exp(-inf) = 0
            float l = wave_sum(ch == 0 ? pA + pB : 0.f);
            float pn[NN];
#pragma unroll
            for (int jn = 0; jn < NN; ++jn) { pn[jn] = __expf(sn_[jn] - m); l += pn[jn]; }
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float pt = lane_bcast(pA, 4 * t);
                o0 = fmaf(pt, vr0[t], o0); o1 = fmaf(pt, vr1[t], o1);
            }
            if (base > 16) {
                for (int t = 16; t < base; ++t) {
                    const float pt = lane_bcast(pB, 4 * (t - 16));
                    o0 = fmaf(pt, vc[(size_t)t * D + lane], o0); o1 = fmaf(pt, vc[(size_t)t * D + lane + HALF], o1);
                }
            }
#pragma unroll
            for (int jn = 0; jn < NN; ++jn) { o0 = fmaf(pn[jn], vn0[jn], o0); o1 = fmaf(pn[jn], vn1[jn], o1); }
            o0 /= l; o1 /= l;
            const int head = kvh * G + h;
            if (a.out) { a.out[(size_t)row * a.ld_out + head * D + lane] = o0; a.out[(size_t)row * a.ld_out + head * D + lane + HALF] = o1; }
            if (a.oh) {
                const float ov[2] = { o0, o1 };
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint32_t u = __float_as_uint(ov[q]);
                    const bf16_t hi = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                    const float rem = ov[q] - __uint_as_float((uint32_t)hi << 16);
                    const uint32_t v = __float_as_uint(rem);
                    a.oh[(size_t)row * a.ldp + head * D + lane + q * HALF] = hi;
                    a.ol[(size_t)row * a.ldp + head * D + lane + q * HALF] = (bf16_t)((v + 0x7FFFu + ((v >> 16) & 1u)) >> 16);
                }
            }
        }
    }
    KP_MARK(23);
}

// ================================================================================================
// k_attn_tiny2 (round 5) — k_attn_tiny for two query heads per kv head and at most 16 tokens in all (every launch of the code predictor in
// the batched step), with the wave's lanes laid out for 16-BYTE accesses.  k_attn_tiny asked for its operands as (lane, lane + 64) dwords:
// ~80 load instructions per wave — 1.14 us of ISSUE before the first byte was needed — and ran the three q / k RMSNorms of a new token
// as three 64-lane reductions one after the other, 1.9 us of single-wave vector issue (profiles/r03_attn_tiny_phase_stamps.txt).  Here a
// 16-lane row owns ONE vector of the kv group (row 0 / 1: the two query heads, row 2: the new key, row 3: the new value), lane s of a row
// the dims [4s, 4s + 4) and [64 + 4s, 64 + 4s + 4) — both halves of every rotate-half RoPE pair in the same lane:
//   * split-K slabs, RoPE tables, norm gains: two 16-byte loads per slab and lane instead of eight dword loads per vector: 8 + 2 + 2
//     instead of 32 + 2 + 4 instructions per new token;
//   * the four vectors are normalised and rotated AT ONCE (one 16-lane row sum, one rsqrt, one rotation) instead of one after the other;
//   * cached V rows as (token = row + 4 i, dims of lane s): 8 sixteen-byte loads instead of 32 dwords; a row accumulates P.V over its
//     four tokens and the rows meet through two permlane swaps per value; p_t arrives by ds_bpermute instead of 16 v_readlane;
//   * the new tokens go through the wave's LDS slice into the same (token, 32-dim chunk) score layout as the cached ones: one code path
//     for all <= 16 tokens, no per-new-token wave reductions.
// Same arithmetic per element (slab sums in slab order, deferred 1 / rms, RMSNorm, rotate-half RoPE, softmax in fp32); sums associate
// differently from k_attn_tiny's (fp32, ~1e-7).  Q3TTS_ATTN_TINY2=0 falls back to k_attn_tiny.
// ================================================================================================
template <int NN>
__global__ __launch_bounds__(128) void k_attn_tiny2(const float* pqkv, const float* pkcache, const float* pvcache, const float* pcos, const float* psin,
                                                     int pbase, AttnArgs a) {
    constexpr int D = 128, HALF = 64, G = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.x * 2 + wave;
    if (pair >= a.nb * a.nkv) return;                                   // wave-uniform
    const int bi = pair / a.nkv, kvh = pair - bi * a.nkv;
    const int slot = a.slot_offset + bi, base = pbase;
    const int PT = 1 << a.page_shift;
    const size_t cbase = ((((size_t)slot * a.pages_per_slot) * a.n_layers + a.layer) * a.nkv + kvh) * (size_t)PT * D;
    const float* kc = pkcache + cbase;
    const float* vc = pvcache + cbase;
    __shared__ __attribute__((aligned(16))) float new_sh[2][NN][4][D];   // [wave][new token][q head 0 | q head 1 | key | value][dim], after norm + RoPE
    float (*ns)[4][D] = new_sh[wave];
    const int row = lane >> 4, s = lane & 15;

    // ---- every load of the launch, before any use ----
    const int voff = row < 2 ? (kvh * G + row) * D : (row == 2 ? (a.nq + kvh) * D : (a.nq + a.nkv + kvh) * D);
    f32x4 xa[NN][4], xb[NN][4], cs4[NN], sn4[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const float* rowp = pqkv + (size_t)(bi * NN + j) * a.ld_qkv + voff + 4 * s;
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const size_t so = (size_t)(sb < a.qkv_nslab ? sb : 0) * a.qkv_slab_stride;
            xa[j][sb] = *reinterpret_cast<const f32x4*>(rowp + so);
            xb[j][sb] = *reinterpret_cast<const f32x4*>(rowp + so + HALF);
        }
        cs4[j] = *reinterpret_cast<const f32x4*>(pcos + (size_t)(base + j) * HALF + 4 * s);
        sn4[j] = *reinterpret_cast<const f32x4*>(psin + (size_t)(base + j) * HALF + 4 * s);
    }
    const bool have_ssq = a.ssq_in != nullptr;
    const int snt = have_ssq ? a.ssq_nt : 1;
    float spart[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) spart[j] = (have_ssq ? a.ssq_in + (size_t)(bi * NN + j) * a.ssq_nt : pqkv + (size_t)(bi * NN + j) * a.ld_qkv)[lane < snt ? lane : 0];
    const float* nwq = a.q_norm ? a.q_norm : pcos;      // address select: the loads stay unconditional, the values are replaced below
    const float* nwk = a.k_norm ? a.k_norm : pcos;
    const float* nw = row < 2 ? nwq : nwk;
    const f32x4 nwa = *reinterpret_cast<const f32x4*>(nw + 4 * s), nwb = *reinterpret_cast<const f32x4*>(nw + HALF + 4 * s);
    // cached K as (token = lane / 4, 32-dim chunk = lane % 4); cached V as (token = row + 4 i, this lane's 8 dims); tokens past `base` repeat the last one
    const int tk = lane >> 2, ch = lane & 3;
    const int last = base > 0 ? base - 1 : 0;
    f32x4 kk[8];
    {
        const int t = tk < base ? tk : last;
#pragma unroll
        for (int e = 0; e < 8; ++e) kk[e] = *reinterpret_cast<const f32x4*>(kc + (size_t)t * D + ch * 32 + e * 4);
    }
    f32x4 va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = row + 4 * i, tt = t < base ? t : last;
        va[i] = *reinterpret_cast<const f32x4*>(vc + (size_t)tt * D + 4 * s);
        vb[i] = *reinterpret_cast<const f32x4*>(vc + (size_t)tt * D + HALF + 4 * s);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- new tokens: slab sums (slab order), deferred 1 / rms, RMSNorm, RoPE — the row's vector; K / V appended to the cache; all four to LDS ----
    const bool do_norm = row < 2 ? a.q_norm != nullptr : (row == 2 && a.k_norm != nullptr);
    const bool do_rope = row < 3;
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const float rsc = ssq_row_scale(spart[j], snt, a.ssq_K, a.ssq_eps, have_ssq);
        f32x4 ya = xa[j][0], yb = xb[j][0];
#pragma unroll
        for (int sb = 1; sb < 4; ++sb) if (sb < a.qkv_nslab) { ya += xa[j][sb]; yb += xb[j][sb]; }
        ya *= rsc; yb *= rsc;
        float ss = (ya.x * ya.x + yb.x * yb.x) + (ya.y * ya.y + yb.y * yb.y);
        ss += (ya.z * ya.z + yb.z * yb.z) + (ya.w * ya.w + yb.w * yb.w);
        ss = row_sum16(ss);
        const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
        if (do_norm) { ya = nwa * (ya * rr); yb = nwb * (yb * rr); }
        const f32x4 za = do_rope ? ya * cs4[j] + (-yb) * sn4[j] : ya;
        const f32x4 zb = do_rope ? yb * cs4[j] + ya * sn4[j] : yb;
        *reinterpret_cast<f32x4*>(&ns[j][row][4 * s]) = za;
        *reinterpret_cast<f32x4*>(&ns[j][row][HALF + 4 * s]) = zb;
        if (row >= 2) {                                                  // rows 2 / 3: the cache append of the new key / value
            float* dst = (row == 2 ? a.kcache : a.vcache) + cbase + (size_t)(base + j) * D;
            *reinterpret_cast<f32x4*>(dst + 4 * s) = za;
            *reinterpret_cast<f32x4*>(dst + HALF + 4 * s) = zb;
        }
    }
    wave_lds_sync();

    // ---- all tokens in one layout: slot t < base from the cache, slots base .. base + NN - 1 the new tokens (from LDS) ----
    {
        const int jn = tk - base;
        const bool isnew = jn >= 0 && jn < NN;
        const int jc = isnew ? jn : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4 kn = *reinterpret_cast<const f32x4*>(&ns[jc][2][ch * 32 + e * 4]);
            kk[e] = isnew ? kn : kk[e];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = row + 4 * i, j2 = t - base;
            const bool vnew = j2 >= 0 && j2 < NN;
            const int j2c = vnew ? j2 : 0;
            const f32x4 na = *reinterpret_cast<const f32x4*>(&ns[j2c][3][4 * s]), nb2 = *reinterpret_cast<const f32x4*>(&ns[j2c][3][HALF + 4 * s]);
            const f32x4 zero = { 0.f, 0.f, 0.f, 0.f };
            va[i] = vnew ? na : (t < base ? va[i] : zero);              // never-written cache rows may hold anything: 0 x NaN
            vb[i] = vnew ? nb2 : (t < base ? vb[i] : zero);
        }
    }
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        const int orow = bi * NN + j;
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(&ns[j][h][ch * 32 + e * 4]);
                part = fmaf(q4.x, kk[e].x, part); part = fmaf(q4.y, kk[e].y, part); part = fmaf(q4.z, kk[e].z, part); part = fmaf(q4.w, kk[e].w, part);
            }
            part += dpp_f<Q3_DPP_XOR1, 0xF>(0.f, part);
            part += dpp_f<Q3_DPP_XOR2, 0xF>(0.f, part);
            const float sc = tk <= base + j ? part * a.scale : -INFINITY;   // causal: the cached tokens and the new ones up to this one; all four lanes of a token hold its score
            const float m = wave_max(sc);                                   // the token itself is always there: m is finite
            const float pt = __expf(sc - m);                                // exp(-inf) = 0
            const float l = wave_sum(ch == 0 ? pt : 0.f);
            f32x4 oa = { 0.f, 0.f, 0.f, 0.f }, ob = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((row + 4 * i) << 4, __builtin_bit_cast(int, pt)));   // lane 4 t holds p_t
                oa += pi * va[i]; ob += pi * vb[i];
            }
            float o[8] = { oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w };
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] += wave_xor_lane_f<16>(o[e], lane); o[e] += wave_xor_lane_f<32>(o[e], lane); o[e] /= l; }
            if (row == 0) {                                                 // every row holds the sums: row 0 stores
                const int head = kvh * G + h;
                if (a.out) {
                    float* dst = a.out + (size_t)orow * a.ld_out + head * D;
                    *reinterpret_cast<f32x4*>(dst + 4 * s) = f32x4{ o[0], o[1], o[2], o[3] };
                    *reinterpret_cast<f32x4*>(dst + HALF + 4 * s) = f32x4{ o[4], o[5], o[6], o[7] };
                }
                if (a.oh) {
                    uint32_t hh[8], ll[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t u = __float_as_uint(o[e]);
                        hh[e] = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                        const uint32_t v = __float_as_uint(o[e] - __uint_as_float(hh[e] << 16));
                        ll[e] = (v + 0x7FFFu + ((v >> 16) & 1u)) >> 16;
                    }
                    const size_t po = (size_t)orow * a.ldp + head * D + 4 * s;
                    *reinterpret_cast<uint2*>(a.oh + po) = make_uint2(hh[0] | hh[1] << 16, hh[2] | hh[3] << 16);
                    *reinterpret_cast<uint2*>(a.oh + po + HALF) = make_uint2(hh[4] | hh[5] << 16, hh[6] | hh[7] << 16);
                    *reinterpret_cast<uint2*>(a.ol + po) = make_uint2(ll[0] | ll[1] << 16, ll[2] | ll[3] << 16);
                    *reinterpret_cast<uint2*>(a.ol + po + HALF) = make_uint2(ll[4] | ll[5] << 16, ll[6] | ll[7] << 16);
                }
            }
        }
    }
}

// ================================================================================================
// k_attn_win (round 4) — the sliding-window attention of the codec decoder's pre-transformer (8 layers, 16 heads x 64, window 72; every
// row's K / V already in the layer's cache, q roped by k_rope_store).  k_attn served it with one workgroup per (head, QUERY): 2048 x 16
// workgroups per layer each fetching its own 72-row window — 37 KB for 9 K multiply-adds, 1.2 GB of L2 traffic and 146 us per layer at
// 2048 frames (1.2 ms of a 31 ms decode).  Here a workgroup takes 32 consecutive queries of one head: their windows overlap in all but 31
// rows, so 103 K rows and 103 V rows are staged once in LDS (16-byte loads, rows padded by 4 floats).  A query is shared by 8 lanes: lane
// j scores window offsets j, j + 8, ... against the query held in its registers, the 8 lanes agree on the maximum and the sum by DPP
// (quad sums + half-row mirror), accumulate P.V over their own offsets and add up the 64 output dims the same way; lane j stores dims
// [8 j, 8 j + 8).  Which lane handles which window offset depends on the offset alone, so a chunked decode (the carried-state stream)
// and the one-shot decode sum in the same order.  fp32 throughout, __expf as in k_attn.
// ================================================================================================
template <int D, int WMAX>
__global__ __launch_bounds__(256, 2) void k_attn_win(AttnArgs a) {
    constexpr int QB = 32, ROWS = QB + WMAX - 1, LDK = D + 4, NM = (WMAX + 7) / 8;
    __shared__ __attribute__((aligned(16))) float Ks[ROWS][LDK];
    __shared__ __attribute__((aligned(16))) float Vs[ROWS][LDK];
    __shared__ float Ss[NM][256];                                // a thread's scores between the two passes (its own column: no conflicts, no barrier)
    const int kvh = blockIdx.x, q0 = blockIdx.y * QB, bi = blockIdx.z;
    const int tid = threadIdx.x, qi = tid >> 3, j = tid & 7;
    const int base = a.pos_scalar, W = a.window;
    const int wbase = base + q0 - (W - 1);                       // position of staged row 0
    const int last = base + a.n_new - 1;                         // newest position the cache holds
    const int slot = a.slot_offset + bi;
    const int page = a.page_table[(size_t)slot * a.pages_per_slot];
    const int P = 1 << a.page_shift;
    const size_t cb = (((size_t)page * a.n_layers + a.layer) * a.nkv + kvh) * (size_t)P * D;
    const float* kc = a.kcache + cb;
    const float* vc = a.vcache + cb;
    const int nrows = QB + W - 1;
    constexpr int NIT = (ROWS * (D / 4) + 255) / 256;             // every staging load of the workgroup in flight at once (7 x 2 per thread)
    f32x4 kst[NIT], vst[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int r = i / (D / 4), c4 = (i % (D / 4)) * 4, pos = wbase + r;
        const int pc = pos < 0 ? 0 : (pos <= last ? pos : last);  // clamped address, zeroed below: rows before the utterance / past its end
        kst[it] = *reinterpret_cast<const f32x4*>(kc + (size_t)pc * D + c4);
        vst[it] = *reinterpret_cast<const f32x4*>(vc + (size_t)pc * D + c4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int r = i / (D / 4), c4 = (i % (D / 4)) * 4, pos = wbase + r;
        if (r < nrows) {
            const bool ok = pos >= 0 && pos <= last;
            *reinterpret_cast<f32x4*>(&Ks[r][c4]) = ok ? kst[it] : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&Vs[r][c4]) = ok ? vst[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int inew = q0 + qi;
    const int irow = inew < a.n_new ? inew : a.n_new - 1;         // clamped: the loads stay unconditional, the store is masked
    const float* qrow = a.qkv + (size_t)(bi * a.n_new + irow) * a.ld_qkv + kvh * D;
    float q[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + d);
        q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
    }
    __syncthreads();
    auto dpp8_sum = [&](float v) __attribute__((always_inline)) -> float {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad of the 8 lanes
        return v;
    };
    auto dpp8_max = [&](float v) __attribute__((always_inline)) -> float {
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
        return v;
    };
    const int p = base + irow;                                   // this query's position; its window = positions [p - W + 1, p]
    // The two passes over the window are ROLLED loops with the scores parked in the thread's own LDS column between them.  Unrolled (round
    // 4) hipcc hoisted the LDS reads of all nine offsets as far as 256 registers allow and past that: 68 bytes of scratch per lane, and —
    // with HIP's float4 struct split into scalars — 382 of the kernel's LDS reads came out as two-dword reads, each with its own address
    // register.  Same operations in the same order: the sums are bit-identical to the unrolled kernel's.
    float mx = -INFINITY;
#pragma unroll 1
    for (int m = 0; m < NM; ++m) {
        const int o = j + 8 * m;                                 // window offset; key position p - (W - 1) + o, staged row qi + o
        const bool valid = o < W && p - (W - 1) + o >= 0;
        const int row = o < W ? (irow - q0) + o : 0;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(&Ks[row][d]);
            s = fmaf(q[d], kv.x, s); s = fmaf(q[d + 1], kv.y, s); s = fmaf(q[d + 2], kv.z, s); s = fmaf(q[d + 3], kv.w, s);
        }
        const float scv = valid ? s * a.scale : -INFINITY;
        Ss[m][tid] = scv;
        mx = fmaxf(mx, scv);
    }
    mx = dpp8_max(mx);                                           // offset W - 1 (the query's own position) is always valid: finite
    float l = 0.f, oacc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) oacc[d] = 0.f;
#pragma unroll 1
    for (int m = 0; m < NM; ++m) {
        const int o = j + 8 * m;
        const float scv = Ss[m][tid];
        // masked offsets read the query's own row (always in range and finite) with weight exp(-inf) = 0: a padded utterance's rows
        // behind it may hold anything, and 0 x NaN would reach the sum
        const int row = scv != -INFINITY ? (irow - q0) + o : (irow - q0) + W - 1;
        const float pw = __expf(scv - mx);
        l += pw;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[row][d]);
            oacc[d] = fmaf(pw, vv.x, oacc[d]); oacc[d + 1] = fmaf(pw, vv.y, oacc[d + 1]); oacc[d + 2] = fmaf(pw, vv.z, oacc[d + 2]); oacc[d + 3] = fmaf(pw, vv.w, oacc[d + 3]);
        }
    }
    l = dpp8_sum(l);
#pragma unroll
    for (int d = 0; d < D; ++d) oacc[d] = dpp8_sum(oacc[d]);
    if (inew < a.n_new) {
        float* out = a.out + (size_t)(bi * a.n_new + inew) * a.ld_out + kvh * D;
        const float inv = 1.0f / l;
#pragma unroll
        for (int e8 = 0; e8 < D / 8; ++e8) {                     // lane j stores dims [8 j, 8 j + 8) (statically indexed registers: a select per 8-dim block)
            if (j == e8) {
#pragma unroll
                for (int e = 0; e < 8; e += 4)
                    *reinterpret_cast<f32x4*>(out + e8 * 8 + e) = f32x4{oacc[e8 * 8 + e] * inv, oacc[e8 * 8 + e + 1] * inv, oacc[e8 * 8 + e + 2] * inv, oacc[e8 * 8 + e + 3] * inv};
            }
        }
    }
}
static bool attn_win_ok(const AttnArgs& a) {
    const bool off = knob("Q3TTS_ATTN_WIN") && atoi(knob("Q3TTS_ATTN_WIN")) == 0;   // A/B knob: back to k_attn
    return !off && a.window > 0 && a.window <= 72 && !a.new_from_raw && a.d == 64 && a.nq == a.nkv && a.n_splits == 1 && a.out != nullptr && a.oh == nullptr &&
           a.pos_dev == nullptr && a.slot_map == nullptr && a.pages_per_slot == 1 && !a.kv_bf16 && a.n_new >= 1 && a.ld_qkv % 4 == 0 && a.ld_out % 4 == 0 &&
           (a.n_new + 31) / 32 <= 65535 && a.nb <= 65535 &&
           (size_t)a.n_new * a.nb >= 128;   // below that the launch is a handful of workgroups: k_attn's one-per-query grid is quicker (F = 64: 37.8 vs 38.6 us/frame)
}

static void launch_attn_stream(const AttnArgs& a, hipStream_t s) {
    const int grp = a.nq / a.nkv;
    if (a.d != 128 || a.n_new != 1 || !a.new_from_raw || a.window != 0 || a.slot_map != nullptr || a.nq % a.nkv || grp != 2)
        throw Error("attn (stream): one new token per row, head_dim 128, two query heads per kv head (the 0.6B and 1.7B talkers), no window, no slot map");
    if (a.n_splits < 1 || a.n_splits > 65535 || (a.n_splits > 1 && (a.po == nullptr || a.chunk % 64 != 0))) throw Error("attn (stream): splits start on page boundaries and need partial buffers");
    if (a.page_shift != 6) throw Error("attn (stream): 64-token KV pages");
    const int span = a.n_splits == 1 ? a.pages_per_slot << 6 : a.chunk;   // tokens one workgroup may walk
    if (a.kv_bf16 && a.kv_round) throw Error("attn (stream): kv_bf16 and kv_round exclude each other");
    const dim3 grid(a.nkv, a.n_splits, a.nb);
#define Q3_AS_ARGS a.page_table, a.pos_dev, a.qkv, (const float*)a.kcache, (const float*)a.vcache, a.rope_cos, a.rope_sin, a.n_splits, span, a
#define Q3_AS_GO(G_, KVB_, MAPC_, U_, WPE_) do { \
        if (a.identity_pages) hipLaunchKernelGGL((k_attn_stream<G_, KVB_, MAPC_, true, U_, WPE_>), grid, dim3(256), 0, s, Q3_AS_ARGS); \
        else hipLaunchKernelGGL((k_attn_stream<G_, KVB_, MAPC_, false, U_, WPE_>), grid, dim3(256), 0, s, Q3_AS_ARGS); } while (0)
    // Tokens per lane group and batch (U) x waves per SIMD, measured at 64 rows x 1024 tokens (profiles/r04_attn_stream_ab.txt; step time,
    // KV rate of the attention's share): bf16 cache 4 x 3 (a batch = one 64-token page); fp32 cache 4 x 2 across splits, 2 x 3 for the
    // one-split launch of short contexts.  Non-temporal loads are worth 0.2-0.3 ms per step (fp32 7.35 -> 7.08 ms, bf16 5.98 -> 5.85);
    // the bf16 cache's lane mapping on an fp32 cache costs 0.35 ms (used only under Q3TTS_FLAG_KV_ROUND_BF16, the bit-for-bit test aid).
    const bool one = a.n_splits == 1;
    // Q3TTS_FLAG_KV_ROUND_BF16 (test aid): the bf16 cache's lane mapping AND batch shape (4 tokens per lane group and batch) on fp32 storage
    // of the rounded rows, so that the online softmax associates exactly as the 16-bit cache's launch does (bit-for-bit test)
    if (a.kv_bf16) Q3_AS_GO(2, true, true, 4, 3);
    else if (a.kv_round) Q3_AS_GO(2, false, true, 4, 2);
    else if (one) Q3_AS_GO(2, false, false, 2, 3);
    else Q3_AS_GO(2, false, false, 4, 2);
#undef Q3_AS_GO
#undef Q3_AS_ARGS
    Q3_HIP_CHECK(hipGetLastError());
}

void launch_attn(const AttnArgs& a, hipStream_t s) {
    if (a.stream) { launch_attn_stream(a, s); return; }
    if (attn_win_ok(a)) {   // the codec decoder's windowed attention: 32 queries per workgroup over a shared K / V window
        hipLaunchKernelGGL((k_attn_win<64, 72>), dim3(a.nkv, (a.n_new + 31) / 32, a.nb), dim3(256), 0, s, a);
        Q3_HIP_CHECK(hipGetLastError());
        return;
    }
    const int grp = a.nq / a.nkv;
    if (grp < 1 || grp > ATT_MAX_GRP || a.nq % a.nkv) throw Error("attn: unsupported GQA group size");
    if (a.n_new > ATT_MAX_NEW && a.new_from_raw) throw Error("attn: too many new tokens per launch");
    if (a.n_splits < 1 || (a.n_splits > 1 && (a.po == nullptr || a.window > 0))) throw Error("attn: split mode needs partial buffers and no window");
    if ((size_t)a.n_new * a.n_splits > 65535) throw Error("attn: grid too large");
    if (a.n_splits > 1 && (a.chunk >> a.page_shift) + 1 > 4) throw Error("attn: a split may touch at most 4 KV pages");
    if (a.n_splits == 1 && !a.identity_pages && a.pages_per_slot > 4 && a.window == 0) throw Error("attn: one split over more than 4 table-mapped KV pages");
    const bool tiny_ctx0 = a.n_splits == 1 && a.window == 0 && (a.pages_per_slot << a.page_shift) <= 32;
    const bool no_tiny = knob("Q3TTS_NO_ATTN_TINY") != nullptr;   // A/B switch
    if (!no_tiny && a.d == 128 && tiny_ctx0 && a.identity_pages && a.pages_per_slot == 1 && a.n_new >= 1 && a.n_new <= 2 && a.slot_map == nullptr &&
        a.pos_dev == nullptr && a.new_from_raw && a.po == nullptr && (a.out || a.oh) && a.nb >= 2 && a.pos_scalar + a.n_new <= 32 && a.qkv_nslab >= 1 && a.qkv_nslab <= 4) {
        const dim3 g((unsigned)((a.nb * a.nkv + 1) / 2));
        const bool tiny2_off = knob("Q3TTS_ATTN_TINY2") && atoi(knob("Q3TTS_ATTN_TINY2")) == 0;   // A/B knob: back to k_attn_tiny
        if (!tiny2_off && grp == 2 && a.pos_scalar + a.n_new <= 16 && a.ld_qkv % 4 == 0 && a.qkv_slab_stride % 4 == 0 && (!a.out || a.ld_out % 4 == 0) && (!a.oh || a.ldp % 4 == 0)) {
            if (a.n_new == 1) hipLaunchKernelGGL((k_attn_tiny2<1>), g, dim3(128), 0, s, a.qkv, (const float*)a.kcache, (const float*)a.vcache, a.rope_cos, a.rope_sin, a.pos_scalar, a);
            else hipLaunchKernelGGL((k_attn_tiny2<2>), g, dim3(128), 0, s, a.qkv, (const float*)a.kcache, (const float*)a.vcache, a.rope_cos, a.rope_sin, a.pos_scalar, a);
            Q3_HIP_CHECK(hipGetLastError());
            return;
        }
#define Q3_TINY(G_, NN_) hipLaunchKernelGGL((k_attn_tiny<G_, NN_>), g, dim3(128), 0, s, a.qkv, (const float*)a.kcache, (const float*)a.vcache, a.rope_cos, a.rope_sin, a.pos_scalar, a)
        if (grp == 1) { if (a.n_new == 1) Q3_TINY(1, 1); else Q3_TINY(1, 2); }
        else if (grp == 2) { if (a.n_new == 1) Q3_TINY(2, 1); else Q3_TINY(2, 2); }
        else if (grp == 4) { if (a.n_new == 1) Q3_TINY(4, 1); else Q3_TINY(4, 2); }
        else goto generic;
#undef Q3_TINY
        return;
    }
generic:;
    dim3 grid(a.nkv, a.n_new * a.n_splits, a.nb);
#define Q3_ATT_ARGS a.page_table, a.pos_dev, a.qkv, (const float*)a.kcache, (const float*)a.vcache, a.rope_cos, a.rope_sin, a.pos_scalar, a.n_splits, a
#define Q3_ATT_I(D, U, I) do { if (grp == 1) hipLaunchKernelGGL((k_attn<D, U, 1, I>), grid, dim3(256), 0, s, Q3_ATT_ARGS); \
        else if (grp == 2) hipLaunchKernelGGL((k_attn<D, U, 2, I>), grid, dim3(256), 0, s, Q3_ATT_ARGS); \
        else hipLaunchKernelGGL((k_attn<D, U, 4, I>), grid, dim3(256), 0, s, Q3_ATT_ARGS); } while (0)
#define Q3_ATT_B(D, U, I) do { if (grp == 1) hipLaunchKernelGGL((k_attn<D, U, 1, I, true>), grid, dim3(256), 0, s, Q3_ATT_ARGS); \
        else if (grp == 2) hipLaunchKernelGGL((k_attn<D, U, 2, I, true>), grid, dim3(256), 0, s, Q3_ATT_ARGS); \
        else hipLaunchKernelGGL((k_attn<D, U, 4, I, true>), grid, dim3(256), 0, s, Q3_ATT_ARGS); } while (0)
#define Q3_ATT(D, U) Q3_ATT_I(D, U, false)
    const bool tiny_ctx = a.n_splits == 1 && a.window == 0 && (a.pages_per_slot << a.page_shift) <= 32; // code predictor
    // identity_pages: the decode stacks with a fixed run of pages per slot (slot_map == null: slot ids are computed too)
    if (a.kv_bf16) {   // the talker's cache under Q3TTS_FLAG_KV_BF16
        if (!a.new_from_raw || tiny_ctx) throw Error("attn: the bf16 cache is the talker's (raw new tokens, paged context)");
        if (a.d == 128 && a.identity_pages) Q3_ATT_B(128, 8, true);
        else if (a.d == 128) Q3_ATT_B(128, 8, false);
        else if (a.d == 64) Q3_ATT_B(64, 8, false);
        else if (a.d == 16) Q3_ATT_B(16, 4, false);
        else throw Error("attn: head_dim must be 16, 64 or 128");
    }
    else if (a.d == 128 && tiny_ctx && a.identity_pages) Q3_ATT_I(128, 2, true);
    else if (a.d == 128 && a.identity_pages && a.n_splits > 1 && a.chunk <= 64) Q3_ATT_I(128, 4, true);   // 64-token splits: 4 tokens per lane group in flight
    else if (a.d == 128 && a.identity_pages) Q3_ATT_I(128, 8, true);
    else if (a.d == 128 && tiny_ctx) Q3_ATT(128, 2);
    else if (a.d == 128) Q3_ATT(128, 8);
    else if (a.d == 64) Q3_ATT(64, 8);
    else if (a.d == 16) Q3_ATT(16, 4);
    else throw Error("attn: head_dim must be 16, 64 or 128");
#undef Q3_ATT_I
#undef Q3_ATT_B
#undef Q3_ATT
}

// partials -> normalised rows (used when the consumer GEMV cannot take the fused prologue)
__global__ __launch_bounds__(256) void k_attn_combine(AttnArgs a) {
    const int row = blockIdx.x, bi = row / a.n_new, inew = row % a.n_new;
    const int pos = (a.pos_dev ? a.pos_dev[a.slot_offset + bi] : a.pos_scalar) + inew;
    int nact = pos / a.chunk + 1;
    if (nact > a.n_splits) nact = a.n_splits;
    // columns over blockIdx.y as well: one row's 2048 columns x S splits on ONE workgroup were 8 dependent load rounds (9.8 us per launch at
    // 64 rows x 9 splits, 0.28 ms per step)
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < a.nq * a.d; idx += 256 * gridDim.y) {
        const int head = idx / a.d, e = idx % a.d;
        const size_t pi = ((size_t)row * a.nq + head) * a.n_splits;
        // online merge in branch-free batches of 4 splits (clamped addresses, zero weight past nact)
        float mx = -INFINITY, L = 0.f, O = 0.f;
        for (int sp0 = 0; sp0 < nact; sp0 += 4) {
            float pmv[4], plv[4], pov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sp = sp0 + q < nact ? sp0 + q : nact - 1;
                pmv[q] = a.pm[pi + sp]; plv[q] = a.pl[pi + sp]; pov[q] = a.po[(pi + sp) * a.d + e];
            }
            float mn = mx;
#pragma unroll
            for (int q = 0; q < 4; ++q) mn = fmaxf(mn, pmv[q]);
            const float corr = __expf(mx - mn);
            L *= corr; O *= corr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float w = sp0 + q < nact ? __expf(pmv[q] - mn) : 0.f;
                L += w * plv[q]; O += w * pov[q];
            }
            mx = mn;
        }
        const float o = O / L;
        if (a.out) a.out[(size_t)row * a.ld_out + idx] = o;
        if (a.oh) {
            const uint32_t u = __float_as_uint(o);
            const bf16_t hi = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
            const float rem = o - __uint_as_float((uint32_t)hi << 16);
            const uint32_t v = __float_as_uint(rem);
            a.oh[(size_t)row * a.ldp + idx] = hi;
            a.ol[(size_t)row * a.ldp + idx] = (bf16_t)((v + 0x7FFFu + ((v >> 16) & 1u)) >> 16);
        }
    }
}
void launch_attn_combine(const AttnArgs& a, hipStream_t s) {
    const int cols = a.nq * a.d, gy = std::max(1, std::min(8, cols / 256));
    hipLaunchKernelGGL(k_attn_combine, dim3(a.nb * a.n_new, gy), dim3(256), 0, s, a);
}

// ================================================================================================
// k_cp_attn_oproj — the code predictor's attention AND its output projection in one launch (b = 1).
// The predictor context is at most 17 tokens, so attention is a few KB of L2-resident K/V per head; what a
// separate attention launch costs is its dependent-launch slot (~4.6 us of the 2.67 ms step, 75 per frame).
// Here every workgroup of the o_proj GEMV (4 output rows x K) recomputes the whole attention vector itself:
// 16 waves, wave w = query head w (the two heads of a kv group both redo its new key), all addresses known at launch
// (contiguous per-slot cache, host-known position), so weights, q/k/v rows, norm/RoPE operands and the cached K/V are ONE
// memory round.  The 4 token groups of a wave meet in registers (permlane swaps), the 16 heads meet in LDS, then 4 waves
// per output row split K.  Workgroup 0 appends the new K/V rows to the cache.  NEW = new rows (1, or 2 for the predictor's
// first pass), U = cached tokens per 16-lane group in flight (cached tokens <= 4 U).
// ================================================================================================
template <int NEW, int U>
__global__ __launch_bounds__(1024) void k_cp_attn_oproj(const bf16_t* pW, const float* pqkv, const float* pkc, const float* pvc, const float* px,
                                                         const float* pcos, const float* psin, uint32_t pk0, uint32_t pk1, CpAttnOprojArgs a) {
    // leading scalars = everything the first memory round's addresses need, preloaded into SGPRs (see k_gemv1); 16 dwords at most, so the
    // small integers travel packed: pk0 = base | page_tokens << 16, pk1 = N | ldx << 16 (ld_qkv is (16 + 2*8) * 128 by construction)
    const int pbase = (int)(pk0 & 0xFFFFu), ppage_tokens = (int)(pk0 >> 16), pN = (int)(pk1 & 0xFFFFu), pldx = (int)(pk1 >> 16);
    constexpr int LDQ = 4096;
    constexpr int D = 128, HALF = 64, EPL = 8, G = 2, NKV = 8, NQ = 16, K = 2048;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 16 waves: one query head each (both heads of a kv group redo its new key)
    const int head = wave, kvh = wave / G;
    const int base = pbase;

    __shared__ float q_s[NQ][NEW][D];      // wave-private staging: (lane, lane+64) layout -> 8 contiguous dims per lane
    __shared__ float knew[NQ][NEW][D];
    __shared__ float vnew[NQ][NEW][D];
    __shared__ float attn_s[NEW][K];
    __shared__ float part[NEW][16];

    KP_MARK(16);
    // ---- the one memory round: o_proj weights, residual, q/k/v rows + their norm / RoPE operands, cached K/V ----
    const int orow = blockIdx.x * 4 + (wave & 3), kq = wave >> 2;
    const int orow_c = orow < pN ? orow : pN - 1;
    const uint4 w4 = ldw_rt(pW + (size_t)orow_c * K + kq * 512 + lane * 8, false);
    float resid[NEW];
#pragma unroll
    for (int m = 0; m < NEW; ++m) resid[m] = px[(size_t)m * pldx + orow_c];
    __builtin_amdgcn_sched_barrier(0);

    struct VecOps { float x0, x1, v0, v1, n0, n1, cs, sn; };
    constexpr int NVEC = 2 * NEW;                // this head's query rows, then the new keys of its kv group
    VecOps vec[NVEC];
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
        const bool is_q = v < NEW;
        const int j = is_q ? v : v - NEW;
        const float* rowp = pqkv + (size_t)j * LDQ;
        const float* src = rowp + (is_q ? head * D : (NQ + kvh) * D);
        const float* vs = rowp + (NQ + NKV + kvh) * D;
        const float* nw = is_q ? a.q_norm : a.k_norm;
        vec[v].x0 = src[lane]; vec[v].x1 = src[lane + HALF];
        vec[v].v0 = vs[lane]; vec[v].v1 = vs[lane + HALF];
        vec[v].n0 = nw[lane]; vec[v].n1 = nw[lane + HALF];
        vec[v].cs = pcos[(size_t)(base + j) * HALF + lane]; vec[v].sn = psin[(size_t)(base + j) * HALF + lane];
    }
    const int tg = lane >> 4, sub = lane & 15;
    float kr[U][EPL], vr[U][EPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int t = tg + 4 * u;                      // clamped, unconditional (a conditional load is a serial round trip)
        t = t < base ? t : base - 1;
        t = t > 0 ? t : 0;
        const size_t off = ((size_t)kvh * ppage_tokens + t) * D + sub * EPL;
        const float4 k0 = *reinterpret_cast<const float4*>(pkc + off), k1 = *reinterpret_cast<const float4*>(pkc + off + 4);
        const float4 v0 = *reinterpret_cast<const float4*>(pvc + off), v1 = *reinterpret_cast<const float4*>(pvc + off + 4);
        kr[u][0] = k0.x; kr[u][1] = k0.y; kr[u][2] = k0.z; kr[u][3] = k0.w; kr[u][4] = k1.x; kr[u][5] = k1.y; kr[u][6] = k1.z; kr[u][7] = k1.w;
        vr[u][0] = v0.x; vr[u][1] = v0.y; vr[u][2] = v0.z; vr[u][3] = v0.w; vr[u][4] = v1.x; vr[u][5] = v1.y; vr[u][6] = v1.z; vr[u][7] = v1.w;
    }
    __builtin_amdgcn_sched_barrier(0);

    KP_MARK(17);
    // ---- 1. q / k RMSNorm + RoPE (reference graphs: per-head norm, rotate-half RoPE); the staging is wave-private, so the
    // wave's in-order LDS queue is all the synchronisation it needs.  The even head of workgroup 0 appends K/V to the cache. ----
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
        const bool is_q = v < NEW;
        const int j = is_q ? v : v - NEW;
        const float ss = wave_sum(vec[v].x0 * vec[v].x0 + vec[v].x1 * vec[v].x1);
        const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
        const float x0 = vec[v].n0 * (vec[v].x0 * rr), x1 = vec[v].n1 * (vec[v].x1 * rr);
        const float y0 = x0 * vec[v].cs + (-x1) * vec[v].sn;
        const float y1 = x1 * vec[v].cs + x0 * vec[v].sn;
        if (is_q) { q_s[head][j][lane] = y0; q_s[head][j][lane + HALF] = y1; }
        else {
            knew[head][j][lane] = y0; knew[head][j][lane + HALF] = y1;
            vnew[head][j][lane] = vec[v].v0; vnew[head][j][lane + HALF] = vec[v].v1;
            if (blockIdx.x == 0 && (head & 1) == 0) {
                const size_t off = ((size_t)kvh * a.page_tokens + base + j) * D;
                a.kc[off + lane] = y0; a.kc[off + lane + HALF] = y1;
                a.vc[off + lane] = vec[v].v0; a.vc[off + lane + HALF] = vec[v].v1;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    KP_MARK(18);
    // ---- 2. scores and weighted values: token group tg owns cached tokens tg, tg+4, ... and new token j if (j & 3) == tg ----
    float kn[NEW][EPL], vn[NEW][EPL];
#pragma unroll
    for (int j = 0; j < NEW; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) { kn[j][e] = knew[head][j][sub * EPL + e]; vn[j][e] = vnew[head][j][sub * EPL + e]; }
    KP_MARK(19);
#pragma unroll
    for (int inew = 0; inew < NEW; ++inew) {
        float qr[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) qr[e] = q_s[head][inew][sub * EPL + e];
        float sc[U + NEW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) sdot = fmaf(qr[e], kr[u][e], sdot);
            sdot = row_sum16(sdot) * a.scale;
            sc[u] = tg + 4 * u < base ? sdot : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < NEW; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) sdot = fmaf(qr[e], kn[j][e], sdot);
            sdot = row_sum16(sdot) * a.scale;
            sc[U + j] = (j <= inew && (j & 3) == tg) ? sdot : -INFINITY;   // causal among the new rows
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < U + NEW; ++i) mx = fmaxf(mx, sc[i]);
        float l = 0.f, o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pw = mx == -INFINITY ? 0.f : __expf(sc[u] - mx);
            l += pw;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = fmaf(pw, tg + 4 * u < base ? vr[u][e] : 0.f, o[e]);   // never-written cache rows may hold NaN
        }
#pragma unroll
        for (int j = 0; j < NEW; ++j) {
            const float pw = mx == -INFINITY ? 0.f : __expf(sc[U + j] - mx);
            l += pw;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = fmaf(pw, vn[j][e], o[e]);
        }
        // merge the wave's 4 token groups in registers: lanes {sub, sub+16, sub+32, sub+48} hold the same 8 dims of different groups
        const float mall = wave_max(mx);                                  // every group's max is replicated over its 16 lanes
        const float wgt = mx == -INFINITY ? 0.f : __expf(mx - mall);
        l *= wgt;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] *= wgt;
        l += wave_xor_lane_f<16>(l, lane);
        l += wave_xor_lane_f<32>(l, lane);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { o[e] += wave_xor_lane_f<16>(o[e], lane); o[e] += wave_xor_lane_f<32>(o[e], lane); }
        if (tg == 0) {
            const float il = 1.0f / l;
            float* dst = &attn_s[inew][head * D + sub * EPL];
            *reinterpret_cast<float4*>(dst) = make_float4(o[0] * il, o[1] * il, o[2] * il, o[3] * il);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4] * il, o[5] * il, o[6] * il, o[7] * il);
        }
    }
    KP_MARK(22);
    __syncthreads();
    KP_MARK(20);
    // ---- 3. o_proj: wave (row = wave & 3, K quarter = wave >> 2), residual add ----
#pragma unroll
    for (int m = 0; m < NEW; ++m) {
        const float* xr = &attn_s[m][kq * 512 + lane * 8];
        const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 4);
        const float xv[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
        const uint32_t wu[4] = { w4.x, w4.y, w4.z, w4.w };
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 = fmaf(xv[2 * j], bf_lo(wu[j]), s1); s1 = fmaf(xv[2 * j + 1], bf_hi(wu[j]), s1); }
        s1 = wave_sum(s1);
        if (lane == 0) part[m][wave] = s1;
    }
    __syncthreads();
    KP_MARK(21);
    if (wave < 4 && lane < NEW && orow < a.N) {
        float r = resid[0];
        if (NEW > 1 && lane == 1) r = resid[NEW - 1];
        a.x[(size_t)lane * a.ldx + orow] = r + (((part[lane][wave] + part[lane][wave + 4]) + part[lane][wave + 8]) + part[lane][wave + 12]);
    }
}

void launch_cp_attn_oproj(const CpAttnOprojArgs& a, int n_new, hipStream_t s) {
    if (!cp_attn_oproj_ok(a, n_new)) throw Error("cp_attn_oproj: unsupported shape");
    const int U = a.base <= 4 ? 1 : (a.base <= 8 ? 2 : (a.base <= 12 ? 3 : 4));
    const dim3 grid((a.N + 3) / 4), block(1024);
    // Two re-layouts of this launch (the kv group's two waves splitting the tokens instead of the heads; k_attn_tiny's one-wave layout)
    // were built in round 3 and measured slower (b=1 step 2.10 ms against 2.29 / 2.30 ms): profiles/r03_negative_results.txt item 6.
#define Q3_CAO(NEW, UU) hipLaunchKernelGGL((k_cp_attn_oproj<NEW, UU>), grid, block, 0, s, a.W, a.qkv, (const float*)a.kc, (const float*)a.vc, (const float*)a.x, \
        a.rope_cos, a.rope_sin, (uint32_t)a.base | (uint32_t)a.page_tokens << 16, (uint32_t)a.N | (uint32_t)a.ldx << 16, a)
    if (n_new == 1) { if (U == 1) Q3_CAO(1, 1); else if (U == 2) Q3_CAO(1, 2); else if (U == 3) Q3_CAO(1, 3); else Q3_CAO(1, 4); }
    else { if (U == 1) Q3_CAO(2, 1); else if (U == 2) Q3_CAO(2, 2); else Q3_CAO(2, 3); }   // two new rows over > 12 cached tokens: refused by cp_attn_oproj_ok (the <2, 4> variant spilled; the predictor's two-row pass starts at base 0)
#undef Q3_CAO
    Q3_HIP_CHECK(hipGetLastError());
}
bool cp_attn_oproj_ok(const CpAttnOprojArgs& a, int n_new) {
    return (n_new == 1 || (n_new == 2 && a.base <= 12)) && a.nq == 16 && a.nkv == 8 && a.d == 128 && a.K == 2048 && a.base >= 0 && a.base <= 16 &&
           a.base + n_new <= a.page_tokens && a.q_norm && a.k_norm && a.ld_qkv == 4096 && a.N >= 1 && a.N < 65536 && a.ldx < 65536 && a.page_tokens < 65536;
}

// ================================================================================================
// k_sample — temperature / top-k / top-p sampling (reference src/tts_onnx.cpp:878-950) on device.
// ONE WAVE per batch row: the whole vocabulary (<= 4096) sits in the wave's registers, every
// reduction / scan is a cross-lane DPP sequence, so there is no block barrier chain; the epilogue
// gathers the sampled token's embedding row and maintains the frame's running embedding sum
// (tts_onnx.cpp:824-842), which keeps the generation loop free of host round trips.
// ================================================================================================
#define SAMP_MAXV 4096

static __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static __device__ __forceinline__ float rng_uniform_dev(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t group) {
    uint64_t k = mix64(seed ^ mix64(((uint64_t)stream << 32) | frame));
    k = mix64(k + group);
    return (float)(k >> 40) * (1.0f / 16777216.0f);
}
// order-preserving float -> uint key (larger float => larger key)
static __device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static __device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// inclusive prefix sum across the wave (lane order)
static __device__ __forceinline__ int wave_scan_i(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
    return v;
}
static __device__ __forceinline__ float wave_scan_f(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const float t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
    return v;
}

// exp() of the sampler, fully specified so that this kernel and the CPU oracle (oracle/q3_oracle.c: q3o_expf, the same lines) agree BIT FOR
// BIT: the device library's expf and libm's differ in the last place on a few per cent of the inputs, and a probability that differs in
// its last bit can flip a top-p cut or a draw that lands on a boundary.  IEEE-exact operations only (mul, fma, rint, power-of-two
// scaling); the file is built with -ffp-contract=off.  Within 1 ulp of expf.
static __device__ __forceinline__ float q3_expf(float x) {
    if (!(x > -103.0f)) return 0.0f;
    x = x > 43.0f ? 43.0f : x;   // domain x <= 43 (callers pass x - max <= 0): the 2^(n+64) scale factor below holds n <= 63; same clamp as q3o_expf
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, z, r) + 1.0f;
    const int ni = (int)n;
    return (y * __uint_as_float((uint32_t)(ni + 64 + 127) << 23)) * __uint_as_float((uint32_t)(-64 + 127) << 23);
}

// The reference's sampler sums in plain loops (tts_onnx.cpp:907-915, 893-898, 929-950): `sum += x[i]` in index order, one fp32 rounding
// per element.  A tree / scan order gives sums that differ in the last bits, and with thousands of candidates (top_k = 0) or exact ties
// (running sums landing ON top_p) that is enough to move a cut or a draw by one element.  So the sums that feed a decision are left
// folds here too: every lane of the calling wave walks the same LDS array (same address in every lane = a broadcast read) and performs
// the same dependent chain of adds; n is a multiple of 32, the entries past the last candidate are +0 (s + 0 == s).
static __device__ __forceinline__ float seq_sum_lds(const float* a, int n) {
    float s = 0.f;
    for (int i = 0; i < n; i += 32) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(a + i + 4 * q);
#pragma unroll
        for (int q = 0; q < 8; ++q) { s += v[q].x; s += v[q].y; s += v[q].z; s += v[q].w; }
    }
    return s;
}
// first index whose left-fold running sum exceeds `target` (entries are >= 0, so the running sum never decreases), -1 if none
static __device__ __forceinline__ int seq_find_lds(const float* a, int n, float target) {
    float s = 0.f;
    int found = 0x7FFFFFFF;
    for (int i = 0; i < n; i += 32) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(a + i + 4 * q);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s += v[q].x; found = s > target ? min(found, i + 4 * q) : found;
            s += v[q].y; found = s > target ? min(found, i + 4 * q + 1) : found;
            s += v[q].z; found = s > target ? min(found, i + 4 * q + 2) : found;
            s += v[q].w; found = s > target ? min(found, i + 4 * q + 3) : found;
        }
        if (__builtin_amdgcn_readfirstlane(found) != 0x7FFFFFFF) break;   // wave-uniform
    }
    found = __builtin_amdgcn_readfirstlane(found);
    return found == 0x7FFFFFFF ? -1 : found;
}
static __device__ __forceinline__ void wave_lds_sync() {   // wave-private LDS staging: the wave's own in-order LDS queue is the synchronisation
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// softmax -> top-p -> renormalise -> inverse-CDF draw (tts_onnx.cpp:886-904) over at most 64 candidates, one per lane IN INDEX ORDER
// (lanes >= nk hold e = 0): e = exp(logit - max) numerators, id = token ids.  Every sum is the reference's left fold.  sb: 3 x 64 floats
// of wave-private LDS.  Returns the drawn token id (wave-uniform).
// have: this lane holds a candidate (the candidates need not be packed to the front: the fast path leaves them where the index-ordered
// gather put them).
static __device__ __forceinline__ int draw_small_exact(float e, int id, bool have, int nk, float top_p, float u, int lane, float (*sb)[64]) {
    sb[0][lane] = e;
    wave_lds_sync();
    const float S = seq_sum_lds(sb[0], 64);
    float p = e / S;                                                          // :913-914
    if (top_p < 1.0f) {                                                       // :929-950: order (p desc, index asc), keep through the first running sum > top_p
        float sk = have ? p : -INFINITY;
        int tag = lane;
        wave_sort_desc_kv(sk, tag, lane);                                     // the nk candidates come first (everything else is -inf)
        sb[1][lane] = lane < nk ? sk : 0.f;
        wave_lds_sync();
        const int cut = seq_find_lds(sb[1], 64, top_p);
        const int keep_rank = (lane < nk && (cut < 0 || lane <= cut)) ? 1 : 0;
        const int keep_here = __builtin_amdgcn_ds_permute(tag << 2, keep_rank); // the tags are a permutation of 0..63: every lane receives its flag
        p = keep_here ? p : 0.f;
        sb[2][lane] = p;
        wave_lds_sync();
        const float s2 = seq_sum_lds(sb[2], 64);
        if (s2 > 0.f) p = p / s2;                                              // :893-898
    }
    wave_lds_sync();                                                           // sb[0] is rewritten: its readers are done (same wave, in order)
    sb[0][lane] = p;
    wave_lds_sync();
    const float total = seq_sum_lds(sb[0], 64);
    const float target = u * total;
    int pick = seq_find_lds(sb[0], 64, target);
    if (pick < 0) {                                                            // rounding left target >= the last running sum: last positive entry
        const unsigned long long pos = __ballot(p > 0.f);
        pick = pos ? 63 - __clzll((long long)pos) : 0;
    }
    return lane_bcast_i(id, pick);
}

// The same decisions from tree-ordered sums (DPP reductions / scans, ~1 us instead of ~5 for the left folds), taken only when they
// provably equal the left-fold ones: a sum of <= 64 non-negative terms differs between any two summation orders by at most
// 2 * 63 * 2^-24 = 7.6e-6 relative, which bounds every quantity a decision compares — running sums of p = e / S against top_p (<= 1.6e-5
// apart between the two evaluations) and against u * total after the renormalisation (<= 5e-5 * total apart).  If no comparison is
// closer to its threshold than that (with a 2x margin), and the element right behind the top-p cut is not within 1e-5 relative of the cut
// element (equal-p elements are ordered by index, and whether two p values collide may differ between the evaluations), both
// evaluations decide alike; otherwise (a few per cent of the calls) the left-fold evaluation decides.
static __device__ __forceinline__ int draw_small(float e, int id, int nk, float top_p, float u, int lane, float (*sb)[64]) {
    const float Sa = wave_sum(e);
    float p = e / Sa;
    bool amb = false;
    if (top_p < 1.0f) {
        float sk = lane < nk ? p : -INFINITY;
        int tag = lane;
        wave_sort_desc_kv(sk, tag, lane);
        const float cs = wave_scan_incl_f(lane < nk ? sk : 0.f);
        const unsigned long long over = __ballot(lane < nk && cs > top_p);
        const int rcut = over ? __ffsll((long long)over) - 1 : 63;
        amb = __ballot(lane < nk && fabsf(cs - top_p) <= 4e-5f) != 0ull;
        const float pcut = lane_bcast(sk, rcut), pnext = lane_bcast(sk, rcut < 63 ? rcut + 1 : 63);
        amb = amb || (rcut + 1 < nk && pnext >= pcut * (1.0f - 1e-5f));
        const int keep_rank = (lane < nk && lane <= rcut) ? 1 : 0;
        const int keep_here = __builtin_amdgcn_ds_permute(tag << 2, keep_rank);
        p = keep_here ? p : 0.f;
        const float s2 = wave_sum(p);
        if (s2 > 0.f) p = p / s2;
    }
    const float total = wave_sum(p);
    const float target = u * total;
    const float cum = wave_scan_incl_f(p);
    amb = amb || __ballot(p > 0.f && fabsf(cum - target) <= 1e-4f * total) != 0ull;
    if (__builtin_expect(!amb, 1)) {
        const unsigned long long hit = __ballot(p > 0.f && cum > target);
        const unsigned long long pos = __ballot(p > 0.f);
        const int pick = hit ? __ffsll((long long)hit) - 1 : (pos ? 63 - __clzll((long long)pos) : 0);
        return lane_bcast_i(id, pick);
    }
    return draw_small_exact(e, id, lane < nk, nk, top_p, u, lane, sb);
}

// draw_small for candidates that arrive sorted by (logit desc, index asc), one per lane, the kept ones in lanes [0, nk) — the order the
// fast path's threshold sort leaves them in, which is also the top-p order (p is monotone in the logit; two different logits whose
// probabilities collide are covered by the neighbour test at the cut).  `tag` = the lane each candidate came from: the gather put the
// candidates there IN INDEX ORDER (contiguous slices per wave), so the draw's order is restored by pushing every probability back to
// its lane (one ds_permute; rounds 1-4 sorted a second time, by index).  id0 = the id this lane held before the sort.
static __device__ __forceinline__ int draw_sorted(float v, int tag, int id0, int nk, float mx, float top_p, float u, int lane, float (*sb)[64]) {
    const bool kept = lane < nk;
    const float e = kept ? q3_expf(v - mx) : 0.f;                              // softmax numerators over the kept entries (:907-915)
    const float Sa = wave_sum(e);
    float p = e / Sa;
    bool amb = false;
    if (top_p < 1.0f) {
        const float cs = wave_scan_incl_f(p);
        const unsigned long long over = __ballot(kept && cs > top_p);
        const int rcut = over ? __ffsll((long long)over) - 1 : 63;
        amb = __ballot(kept && fabsf(cs - top_p) <= 4e-5f) != 0ull;
        const float pcut = lane_bcast(p, rcut), pnext = lane_bcast(p, rcut < 63 ? rcut + 1 : 63);
        amb = amb || (rcut + 1 < nk && pnext >= pcut * (1.0f - 1e-5f));
        p = lane <= rcut ? p : 0.f;
        const float s2 = wave_sum(p);
        if (s2 > 0.f) p = p / s2;
    }
    // index order for the draw: the tags are a permutation of 0..63, every lane receives the probability of the candidate it gathered
    const float pi = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(tag << 2, __builtin_bit_cast(int, p)));
    const float total = wave_sum(pi);
    const float target = u * total;
    const float cum = wave_scan_incl_f(pi);
    amb = amb || __ballot(pi > 0.f && fabsf(cum - target) <= 1e-4f * total) != 0ull;
    if (__builtin_expect(!amb, 1)) {   // the left-fold evaluation is the cold path: laid out behind the quick one
        const unsigned long long hit = __ballot(pi > 0.f && cum > target);
        const unsigned long long pos = __ballot(pi > 0.f);
        const int pick = hit ? __ffsll((long long)hit) - 1 : (pos ? 63 - __clzll((long long)pos) : 0);
        return lane_bcast_i(id0, pick);
    }
    const float ei = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(tag << 2, __builtin_bit_cast(int, e)));
    const int havei = __builtin_amdgcn_ds_permute(tag << 2, kept ? 1 : 0);
    return draw_small_exact(ei, id0, havei != 0, nk, top_p, u, lane, sb);
}

// k-th largest of one value per lane (ties allowed), -inf when fewer than k lanes hold a finite value: the wave sorts its 64
// values in registers (q3_wave_sort.h) and reads rank k-1.  (A 64-broadcast rank count did the same in 1.4 us; this is ~0.3.)
static __device__ __forceinline__ float kth_largest_of_lanes(float v, int k) {
    const int lane = threadIdx.x & 63;
    const float sorted = wave_sort_desc(v, lane);
    const int r = __builtin_amdgcn_readfirstlane(k) - 1;
    return lane_bcast(sorted, r < 0 ? 0 : (r > 63 ? 63 : r));
}

// Four waves per utterance: wave w owns the CONTIGUOUS 64-element slices j = w PW .. w PW + PW - 1 of the logits row (registers; round 5 —
// they were interleaved, j = w, w + 4, ...: with contiguous ranges the four waves' survivor lists concatenate in index order, so the
// draw's index order is the gather order and no second sort is needed),
// the few cross-wave hand-offs go through LDS; the serial tail (threshold rounds, top-p, draw) runs on wave 0
// only, the embedding epilogue on all 256 threads.
// PW = 64-element slices per wave: 16 covers 4096 logits; 8 (vocabularies up to 2048: fifteen of a frame's sixteen samplers) and 12 (up to
// 3072: the code0 sampler) halve / trim every per-element loop — loads, suppression, the temperature division, survivor scans
#ifdef Q3_SAMPLE_PROF
#define SP_MARK(k) do { if (threadIdx.x == 0) g_kernel_prof[k] = wall_clock64(); } while (0)
#else
#define SP_MARK(k) do { } while (0)
#endif
template <bool SLABS, int PW>   // SLABS: the logits row is the ordered sum of a.nslab (<= 4) split-K partial slabs of the head projection
__global__ __launch_bounds__(256) void k_sample(const float* plogits, SlotState* pst, int pld, int pV, SampleArgs a) {   // leading scalars: preloaded (see k_gemv1)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = pV;
    __shared__ float gmax[4][64];                // per-wave lane maxima
    __shared__ float svw[4][256 + 64];           // per-wave survivor staging (+64 dump slots)
    __shared__ int svn[4];
    __shared__ float svb[256 + 64];              // wave 0: survivor rounds
    __shared__ int cnt_s[SAMP_MAXV / 64];        // survivors per 64-slice (index-ordered compaction)
    __shared__ int cand_idx[SAMP_MAXV + 64];
    __shared__ __attribute__((aligned(16))) float cand_p[SAMP_MAXV + 64];
    __shared__ __attribute__((aligned(16))) float sorted_p[SAMP_MAXV];        // general path only
    __shared__ int cand_rank[SAMP_MAXV];         // general path only: position in (p desc, index asc) order
    __shared__ __attribute__((aligned(16))) float sb[3][64];                  // wave 0: staging of the left-fold sums (draw_small)
    __shared__ float sh_f[4];
    __shared__ int sh_i[4];

    SP_MARK(0);
    if (a.step_gen != nullptr && b == 0 && tid == 0) *a.step_gen += 1u;   // first launch of a step: the generation its seam flags will carry
    // ---- round trip 1: slot state (one 64-byte struct) and this wave's logits slices, all in flight ----
    float temperature = a.temperature, top_p = a.top_p, u = a.u_dev ? a.u_dev[b] : a.u;
    int top_k = a.top_k, suppress = a.suppress, keep_eos = 1;
    int frame = 0;
    SlotState* st = pst ? pst + b : nullptr;
    SlotState sl;
    if (st) {
        const uint4* sp = reinterpret_cast<const uint4*>(st);
        uint4 raw[4] = { sp[0], sp[1], sp[2], sp[3] };
        __builtin_memcpy(&sl, raw, sizeof sl);
    }
    const float* lg = plogits + (size_t)b * pld;
    const int PER = (V + 63) / 64;               // 64-element slices in the row
    float x[PW];
    if (SLABS) {   // every load first (clamped slab index: a load under a runtime condition is a serial round trip), then the sums in slab order
        float xp[PW][4];
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const int i = (wave * PW + jj) * 64 + lane;
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) xp[jj][sb] = lg[(size_t)(sb < a.nslab ? sb : 0) * a.slab_stride + (i < V ? i : V - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            float t = xp[jj][0];
#pragma unroll
            for (int sb = 1; sb < 4; ++sb) t = sb < a.nslab ? t + xp[jj][sb] : t;
            x[jj] = t;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const int i = (wave * PW + jj) * 64 + lane;
            x[jj] = lg[i < V ? i : V - 1];           // clamped, unconditional
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st) {
        if (!sl.active || sl.finished) return;
        if (a.group == 0 && sl.n_frames >= sl.max_frames) { if (tid == 0) st->finished = 1; return; }
        temperature = sl.temperature; top_p = sl.top_p; top_k = sl.top_k;
        frame = sl.n_frames;
        u = rng_uniform_dev(sl.seed, sl.stream_id, (uint32_t)frame, (uint32_t)a.group);
        suppress = a.group == 0;
        keep_eos = !sl.ignore_eos;
    }
    SP_MARK(1);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    // suppress (:803-807) + temperature (:882-884)
    const bool use_temp = temperature > 0.0f && temperature != 1.0f;
    float lmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < PW; ++jj) {
        const int i = (wave * PW + jj) * 64 + lane;
        float v = x[jj];
        const bool sup = suppress && i >= a.sup_begin && i < a.sup_end && !(i == a.eos_id && keep_eos);
        const float vt = v / temperature;
        v = use_temp ? vt : v;
        v = (sup || i >= V) ? -INFINITY : v;
        x[jj] = v;
        lmax = fmaxf(lmax, v);
    }
    int tok = 0;
    // ---- fast path (2 <= top_k <= 64): one bound, one survivor list, two register sorts ----
    // The k-th largest logit is at least B = the k-th largest of the 64 values {16 largest THREAD maxima of each wave} (a subset of
    // the logits, so its k-th largest cannot exceed theirs).  Thread maxima are a much tighter net than the 64 lane-group maxima:
    // typically k+5..k+10 values survive, i.e. they fit one value per lane of wave 0, where a (value desc, index asc) sort yields
    // the threshold, the kept set and the top-p order at once, and a second sort by index restores the order the draw walks in.
    bool fast_done = false;
    if (top_k >= 2 && top_k <= 64 && top_k < V) {
        const float tsort = wave_sort_desc(lmax, lane);
        if (lane < 16) svb[wave * 16 + ((wave & 1) ? 15 - lane : lane)] = tsort;   // odd waves' runs reversed: runs 0|1 and 2|3 form bitonic blocks of 32
        __syncthreads();
        const float msort = wave_merge4x16_desc(svb[lane], lane);             // every wave redoes the 64-value merge (11 stages: the runs are sorted): no second barrier
        const float B = lane_bcast(msort, __builtin_amdgcn_readfirstlane(top_k) - 1);
        const float mxf = lane_bcast(msort, 0);
        SP_MARK(3);
        int nw = 0;
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const bool sv = x[jj] >= B && x[jj] != -INFINITY;
            const unsigned long long m = __ballot(sv);
            const int ppos = nw + __popcll(m & lt_mask);
            const int slot = (sv && ppos < 256) ? ppos : 256 + lane;
            svw[wave][slot] = x[jj];
            cand_idx[wave * (256 + 64) + slot] = (wave * PW + jj) * 64 + lane;
            nw += __popcll(m);
        }
        if (lane == 0) svn[wave] = nw;
        __syncthreads();
        SP_MARK(4);
        const int n0 = svn[0], n1 = svn[1], n2 = svn[2], n3 = svn[3], n = n0 + n1 + n2 + n3;
        if (B != -INFINITY && n >= 1 && n <= 64) {       // block-uniform
            fast_done = true;
            if (wave == 0) {
                float v = -INFINITY;
                int id = 0x7FFFFFFF;
                const int w = lane < n0 ? 0 : (lane < n0 + n1 ? 1 : (lane < n0 + n1 + n2 ? 2 : 3));
                const int off = lane - (w == 0 ? 0 : (w == 1 ? n0 : (w == 2 ? n0 + n1 : n0 + n1 + n2)));
                if (lane < n) { v = svw[w][off]; id = cand_idx[w * (256 + 64) + off]; }   // index order: wave w's list precedes wave w + 1's
                int tag = lane;
                wave_sort_desc_kv(v, tag, lane);                                    // (logit desc, index asc): lane order IS index order
                const int kk = __builtin_amdgcn_readfirstlane(top_k);
                const float thrf = n >= kk ? lane_bcast(v, kk - 1) : -INFINITY;     // ties at the threshold stay (:917-927)
                const bool kept = lane < n && v >= thrf;
                const int nk = __popcll(__ballot(kept));
                SP_MARK(5);
                SP_MARK(6);
                tok = draw_sorted(v, tag, id, nk, mxf, top_p, u, lane, sb);   // lanes [0, nk) are the kept ones: the sort put them first
                SP_MARK(7);
                if (lane == 0) sh_i[3] = tok;
            }
        }
    }
    if (__builtin_expect(!fast_done, 0)) {   // top_k == 1, 0 or > 64, or more than 64 survivors (mass ties): the general machinery
        gmax[wave][lane] = lmax;
        __syncthreads();
        // 64 group maxima (group = lane, over all four waves) -> global max and the prefilter bound
        const float gm = fmaxf(fmaxf(gmax[0][lane], gmax[1][lane]), fmaxf(gmax[2][lane], gmax[3][lane]));
        const float mx = wave_max(gm);
        SP_MARK(2);

        // ---- top-k threshold = k-th largest value, ties kept (:917-927) ----
        float thr = -INFINITY;
        if (top_k > 0 && top_k < V) {
            bool done = false;
            if (top_k == 1) { thr = mx; done = true; }
            else if (top_k <= 64) {
                // Iterated prefilter: the k-th largest of the 64 group maxima is a lower bound L of the k-th
                // largest overall, so only elements >= L can matter (~100 of 3072); wave 0 re-deals the survivors
                // over its lanes and repeats until at most one candidate per lane is left, ranked exactly.
                const float L1 = kth_largest_of_lanes(gm, top_k);   // every wave computes the same L1
                SP_MARK(3);
                int nw = 0;
    #pragma unroll
                for (int jj = 0; jj < PW; ++jj) {
                    const bool sv = x[jj] >= L1 && x[jj] != -INFINITY;
                    const unsigned long long m = __ballot(sv);
                    const int ppos = nw + __popcll(m & lt_mask);
                    svw[wave][(sv && ppos < 256) ? ppos : 256 + lane] = x[jj];
                    nw += __popcll(m);
                }
                if (lane == 0) svn[wave] = nw;
                __syncthreads();
                SP_MARK(4);
                if (wave == 0) {
                    const int n0 = svn[0], n1 = svn[1], n2 = svn[2], n3 = svn[3];
                    int n = n0 + n1 + n2 + n3;
                    bool ok = L1 != -INFINITY && n <= 256;
                    if (ok) { // gather the four wave-local lists into svb[0..n)
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int i = q * 64 + lane;
                            float v = -INFINITY;
                            if (i < n0) v = svw[0][i];
                            else if (i < n0 + n1) v = svw[1][i - n0];
                            else if (i < n0 + n1 + n2) v = svw[2][i - n0 - n1];
                            else if (i < n) v = svw[3][i - n0 - n1 - n2];
                            svb[i] = v;
                        }
                    }
                    for (int round = 0; ok && round < 8; ++round) {
                        float s4[4];
    #pragma unroll
                        for (int q = 0; q < 4; ++q) { const float v = svb[q * 64 + lane]; s4[q] = q * 64 + lane < n ? v : -INFINITY; }
                        if (n <= 64) { thr = kth_largest_of_lanes(s4[0], top_k); done = thr != -INFINITY; break; }
                        const float L2 = kth_largest_of_lanes(fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3])), top_k);
                        int n2c = 0;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool sv = s4[q] >= L2 && s4[q] != -INFINITY;
                            const unsigned long long m = __ballot(sv);
                            const int ppos = n2c + __popcll(m & lt_mask);
                            svb[sv ? ppos : 256 + lane] = s4[q];
                            n2c += __popcll(m);
                        }
                        ok = L2 != -INFINITY && n2c < n; // no progress (mass ties): exact fallback below
                        n = n2c;
                    }
                    if (lane == 0) { sh_f[0] = thr; sh_i[0] = done ? 1 : 0; }
                }
                __syncthreads();
                SP_MARK(5);
                thr = sh_f[0];
                done = sh_i[0] != 0;
            }
            if (!done) { // rare fallback (top_k > 64, or pathological ties): bitwise search of the k-th largest key
                uint32_t prefix = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = prefix | (1u << bit);
                    int c = 0;
    #pragma unroll
                    for (int jj = 0; jj < PW; ++jj) c += ((wave * PW + jj) * 64 + lane < V && fkey(x[jj]) >= cand) ? 1 : 0;
                    c = wave_sum_i(c);
                    __syncthreads();
                    if (lane == 0) svn[wave] = c;
                    __syncthreads();
                    if (svn[0] + svn[1] + svn[2] + svn[3] >= top_k) prefix = cand;
                }
                const uint32_t ku = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
                thr = __uint_as_float(ku);
            }
        }

        // ---- index-ordered compaction of the kept entries; softmax numerators exp(x - max) (:907-915) ----
        unsigned long long km[PW];
    #pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const bool keep = x[jj] >= thr && x[jj] != -INFINITY;
            km[jj] = __ballot(keep);
            const int j = wave * PW + jj;
            if (lane == 0 && j < PER) cnt_s[j] = __popcll(km[jj]);
        }
        __syncthreads();
        // exclusive prefix of the per-slice counts (slice order = index order); lane j holds slice j
        const int cmine = lane < PER ? cnt_s[lane] : 0;
        const int cincl = wave_scan_incl_i(cmine);
        const int n_kept = lane_bcast_i(cincl, 63);
    #pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const int j = wave * PW + jj;
            if (km[jj]) { // wave-uniform: most slices hold no survivor
                const int base = lane_bcast_i(cincl, j) - lane_bcast_i(cmine, j);
                const bool keep = (km[jj] >> lane) & 1ull;
                const int wpos = keep ? base + __popcll(km[jj] & lt_mask) : SAMP_MAXV + lane;
                cand_idx[wpos] = j * 64 + lane;
                cand_p[wpos] = keep ? q3_expf(x[jj] - mx) : 0.f;                      // softmax numerators, index order (:907-915)
            }
        }
        __syncthreads();
        SP_MARK(6);

        if (wave == 0) {
            if (n_kept <= 64) {
                // ---- wave path: one candidate per lane (already in index order) ----
                const bool have = lane < n_kept;
                tok = draw_small(have ? cand_p[lane] : 0.f, have ? cand_idx[lane] : 0, n_kept, top_p, u, lane, sb);
                SP_MARK(7);
            } else {
                // ---- general path (top_k == 0 or > 64, or many ties): LDS-resident candidates in index order, wave 0 only; the same
                // left-fold sums as draw_small, over arrays padded with +0 to a multiple of 32 ----
                const int npad = (n_kept + 31) & ~31;
                for (int c = n_kept + lane; c < npad; c += 64) cand_p[c] = 0.f;
                wave_lds_sync();
                const float S = seq_sum_lds(cand_p, npad);
                for (int c = lane; c < n_kept; c += 64) cand_p[c] = cand_p[c] / S;
                wave_lds_sync();
                if (top_p < 1.0f) {   // :929-950 — rank = position in (p desc, index asc) order; keep the ranks through the first running sum > top_p
                    for (int c = lane; c < n_kept; c += 64) {
                        const float pc = cand_p[c];
                        int rank = 0;
                        for (int o2 = 0; o2 < n_kept; ++o2) {
                            const float po = cand_p[o2];
                            rank += (po > pc || (po == pc && o2 < c)) ? 1 : 0;
                        }
                        sorted_p[rank] = pc;
                        cand_rank[c] = rank;
                    }
                    for (int c = n_kept + lane; c < npad; c += 64) sorted_p[c] = 0.f;
                    wave_lds_sync();
                    const int cut = seq_find_lds(sorted_p, npad, top_p);
                    const int cutoff = cut < 0 ? n_kept : cut + 1;
                    for (int c = lane; c < n_kept; c += 64) if (cand_rank[c] >= cutoff) cand_p[c] = 0.f;
                    wave_lds_sync();
                    const float s2 = seq_sum_lds(cand_p, npad);
                    if (s2 > 0.f) for (int c = lane; c < n_kept; c += 64) cand_p[c] = cand_p[c] / s2; // :893-898
                    wave_lds_sync();
                }
                const float total = seq_sum_lds(cand_p, npad);
                const float target = u * total;
                int pick = seq_find_lds(cand_p, npad, target);
                if (pick < 0) {   // rounding left target >= the last running sum: last positive entry
                    float last = -1.f;
                    for (int c = lane; c < n_kept; c += 64) if (cand_p[c] > 0.f) last = (float)c;
                    last = wave_max(last);
                    pick = last < 0.f ? 0 : (int)last;
                }
                tok = cand_idx[pick];
            }
            if (lane == 0) sh_i[3] = tok;
        }
    }
    __syncthreads();
    tok = sh_i[3];
    SP_MARK(8);

    if (!st) { if (tid == 0) a.token_out[b] = tok; return; }

    // ---- fused epilogue of the generation loop (tts_onnx.cpp:812-842, 864-868), all 256 threads ----
    if (a.group == 0 && tok == a.eos_id) { if (tid == 0) st->finished = 1; return; } // :812 — no frame recorded
    if (tid == 0) a.codes[((size_t)b * a.max_frames_cap + frame) * a.n_groups + a.group] = tok;
    const bf16_t* er = a.embed + (size_t)tok * a.H;
    const bool last_group = a.group == a.n_groups - 1;
    const float* text = nullptr;
    if (last_group) text = frame < sl.trailing_len ? a.trailing + ((size_t)b * a.max_trailing + frame) * a.H : a.tts_pad; // :833-842
    const float* sum_r = a.group != 0 ? a.sum + (size_t)b * a.H : nullptr;
    constexpr int EP_MAX = 2; // H <= 2048: 256 threads x 4 floats x 2
    const bool planes = a.pl_h != nullptr && a.x_next != nullptr;   // uniform: the next pass's input planes are made here (launcher: H <= 2048)
    const float* lh_r = planes && a.lh ? a.lh + (size_t)b * a.ld_lh : nullptr;
    float ss_e = 0.f, ss_l = 0.f;
    for (int h0 = 0; h0 < a.H; h0 += 1024 * EP_MAX) {
        float e[EP_MAX][4], sm[EP_MAX][4], tx[EP_MAX][4], g0[EP_MAX][4], lhv[EP_MAX][4];
#pragma unroll
        for (int it = 0; it < EP_MAX; ++it) { // every load first (clamped, unconditional), stores below
            int h = h0 + (it * 256 + tid) * 4;
            h = h < a.H ? h : a.H - 4;
            const uint2 raw = *reinterpret_cast<const uint2*>(er + h);
            e[it][0] = __uint_as_float(raw.x << 16); e[it][1] = __uint_as_float(raw.x & 0xFFFF0000u);
            e[it][2] = __uint_as_float(raw.y << 16); e[it][3] = __uint_as_float(raw.y & 0xFFFF0000u);
            if (sum_r) { const float4 v = *reinterpret_cast<const float4*>(sum_r + h); sm[it][0] = v.x; sm[it][1] = v.y; sm[it][2] = v.z; sm[it][3] = v.w; }
            if (last_group) { const float4 v = *reinterpret_cast<const float4*>(text + h); tx[it][0] = v.x; tx[it][1] = v.y; tx[it][2] = v.z; tx[it][3] = v.w; }
            if (planes) { const float4 v = *reinterpret_cast<const float4*>(a.gamma0 + h); g0[it][0] = v.x; g0[it][1] = v.y; g0[it][2] = v.z; g0[it][3] = v.w; }
            if (lh_r) { const float4 v = *reinterpret_cast<const float4*>(lh_r + h); lhv[it][0] = v.x; lhv[it][1] = v.y; lhv[it][2] = v.z; lhv[it][3] = v.w; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < EP_MAX; ++it) {
            const int h = h0 + (it * 256 + tid) * 4;
            if (h < a.H) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sacc = a.group == 0 ? e[it][q] : sm[it][q] + e[it][q]; // fp32, order code0, sub0..sub14 (:824-830)
                    o[q] = last_group ? sacc + tx[it][q] : sacc;
                }
                if (a.x_next) *reinterpret_cast<float4*>(a.x_next + (size_t)b * a.ld_xnext + h) = make_float4(e[it][0], e[it][1], e[it][2], e[it][3]);
                if (planes) {   // gamma0 * row as (hi, lo) bf16 planes; 1 / rms is the consumer's (deferred RMSNorm)
                    const size_t prow = (size_t)b * a.pl_row_mul + a.pl_row_add;
                    uint32_t hh[4], ll[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float y = g0[it][q] * e[it][q];
                        ss_e = fmaf(e[it][q], e[it][q], ss_e);
                        const uint32_t u = __float_as_uint(y);
                        hh[q] = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                        const uint32_t v = __float_as_uint(y - __uint_as_float(hh[q] << 16));
                        ll[q] = (v + 0x7FFFu + ((v >> 16) & 1u)) >> 16;
                    }
                    *reinterpret_cast<uint2*>(a.pl_h + prow * a.pl_ldp + h) = make_uint2(hh[0] | hh[1] << 16, hh[2] | hh[3] << 16);
                    *reinterpret_cast<uint2*>(a.pl_l + prow * a.pl_ldp + h) = make_uint2(ll[0] | ll[1] << 16, ll[2] | ll[3] << 16);
                    if (lh_r) {   // pass 0's first row: the talker's last_hidden of this utterance
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float y = g0[it][q] * lhv[it][q];
                            ss_l = fmaf(lhv[it][q], lhv[it][q], ss_l);
                            const uint32_t u = __float_as_uint(y);
                            hh[q] = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                            const uint32_t v = __float_as_uint(y - __uint_as_float(hh[q] << 16));
                            ll[q] = (v + 0x7FFFu + ((v >> 16) & 1u)) >> 16;
                        }
                        *reinterpret_cast<uint2*>(a.pl_h + (prow - 1) * a.pl_ldp + h) = make_uint2(hh[0] | hh[1] << 16, hh[2] | hh[3] << 16);
                        *reinterpret_cast<uint2*>(a.pl_l + (prow - 1) * a.pl_ldp + h) = make_uint2(ll[0] | ll[1] << 16, ll[2] | ll[3] << 16);
                    }
                }
                if (last_group) *reinterpret_cast<float4*>(a.x_talk + (size_t)b * a.H + h) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<float4*>(a.sum + (size_t)b * a.H + h) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (planes) {   // the rows' sums of squares: waves in order, then one partial per row (the consumer sums ssq_nt partials: the rest are zero)
        ss_e = wave_sum(ss_e); ss_l = wave_sum(ss_l);
        __syncthreads();                       // sh_f is free again (the token has been read)
        if (lane == 0) { sh_f[wave] = ss_e; gmax[0][wave] = ss_l; }
        __syncthreads();
        const size_t prow = (size_t)b * a.pl_row_mul + a.pl_row_add;
        if (tid < a.ssq_nt) {
            a.ssq_out[prow * a.ssq_nt + tid] = tid == 0 ? ((sh_f[0] + sh_f[1]) + sh_f[2]) + sh_f[3] : 0.f;
            if (lh_r) a.ssq_out[(prow - 1) * a.ssq_nt + tid] = tid == 0 ? ((gmax[0][0] + gmax[0][1]) + gmax[0][2]) + gmax[0][3] : 0.f;
        }
    }
    if (last_group && tid == 0) {
        st->n_frames = frame + 1;
        a.talker_pos[b] = sl.prompt_len + frame; // position of the token the talker decodes next
    }
}

void launch_sample(const SampleArgs& a, hipStream_t s) {
    if (a.V > SAMP_MAXV) throw Error("sample: vocabulary larger than 4096");
    if (a.nslab < 1 || a.nslab > 4) throw Error("sample: 1..4 logits slabs");
    if (a.pl_h && (a.H > 2048 || a.H % 4 || !a.pl_l || !a.gamma0 || !a.ssq_out || a.ssq_nt < 1 || a.ssq_nt > 256 || a.pl_ldp % 4 || (a.lh && a.pl_row_add < 1)))
        throw Error("sample: bad plane-output arguments");
#define Q3_SAMP(SL, PW_) hipLaunchKernelGGL((k_sample<SL, PW_>), dim3(a.nb), dim3(256), 0, s, a.logits, a.st, a.ld, a.V, a)
    if (a.nslab > 1) { if (a.V <= 2048) Q3_SAMP(true, 8); else if (a.V <= 3072) Q3_SAMP(true, 12); else Q3_SAMP(true, 16); }
    else { if (a.V <= 2048) Q3_SAMP(false, 8); else if (a.V <= 3072) Q3_SAMP(false, 12); else Q3_SAMP(false, 16); }
#undef Q3_SAMP
}

// ================================================================================================
// small helpers
// ================================================================================================
__global__ void k_gather_rows_bf16(const bf16_t* table, int H, const int64_t* ids, float* out, int ldo) {
    const int r = blockIdx.x;
    const bf16_t* src = table + (size_t)ids[r] * H;
    for (int h = threadIdx.x; h < H; h += blockDim.x) out[(size_t)r * ldo + h] = __uint_as_float((uint32_t)src[h] << 16);
}
void launch_gather_rows_bf16(const bf16_t* table, int H, const int64_t* ids_dev, int n, float* out, int ldo, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_rows_bf16, dim3(n), dim3(256), 0, s, table, H, ids_dev, out, ldo);
}

__global__ void k_copy_rows(const float* src, int lds, float* dst, int ldd, int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}
void launch_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    if (rows > 0) hipLaunchKernelGGL(k_copy_rows, dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cols);
}

__global__ void k_copy_rows_masked(const float* src, int lds, float* dst, int ldd, int cols, const int* flags) {
    const int r = blockIdx.x;
    if (flags[r] == 0) return;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}
void launch_copy_rows_masked(const float* src, int lds, float* dst, int ldd, int rows, int cols, const int* flags_dev, hipStream_t s) {
    if (rows > 0) hipLaunchKernelGGL(k_copy_rows_masked, dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cols, flags_dev);
}
__global__ void k_bump_u32(unsigned* p) { *p += 1u; }
void launch_bump_u32(unsigned* counter, hipStream_t s) { hipLaunchKernelGGL(k_bump_u32, dim3(1), dim3(1), 0, s, counter); }

__global__ void k_count_active(const SlotState* st, int nb, int32_t* out) {
    int n = 0;
    for (int b = 0; b < nb; ++b) n += (st[b].active && !st[b].finished && st[b].n_frames < st[b].max_frames) ? 1 : 0;
    *out = n;
}
void launch_count_active(const SlotState* st, int nb, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_active, dim3(1), dim3(1), 0, s, st, nb, out);
}

// Synthetic weights: value i of a tensor = mean + stddev * z, z ~ Irwin-Hall(4 x u16) standardised,
// rounded to bf16.  Integer hashing + two fp32 roundings: reproducible bit-for-bit anywhere.
__global__ void k_fill_synth(void* dst, int is_bf16, int64_t n, uint64_t key, float mean, float stddev) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = mix64(key + (uint64_t)i);
        const int sum = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)((h >> 48) & 0xFFFF);
        const float z = (float)(sum - 131070) * (1.0f / 37837.225f); // sqrt(4 * (65536^2 - 1) / 12)
        const float v = mean + stddev * z;
        uint32_t u = __float_as_uint(v);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        if (is_bf16) reinterpret_cast<bf16_t*>(dst)[i] = (bf16_t)(u >> 16);
        else reinterpret_cast<float*>(dst)[i] = __uint_as_float(u);
    }
}
void launch_fill_synth(void* dst, int is_bf16, int64_t n, uint64_t key, float mean, float stddev, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_synth, dim3((int)blocks), dim3(256), 0, s, dst, is_bf16, n, key, mean, stddev);
}

} // namespace q3
