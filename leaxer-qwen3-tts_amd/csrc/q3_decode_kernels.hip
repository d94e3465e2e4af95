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

namespace q3 {

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
static __device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
static __device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }
static __device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ================================================================================================
// k_gemv
// ================================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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

void launch_gemv(const GemvArgs& a0, hipStream_t s) {
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
// k_attn — one workgroup per (kv head, new token, batch row)
// ================================================================================================
#define ATT_MAX_NEW 16
#define ATT_MAX_GRP 4

// D = head_dim; 16 lanes share a token, EPL = D/16 elements per lane
template <int D>
__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
    constexpr int EPL = D / 16;
    constexpr int HALF = D / 2;
    const int kvh = blockIdx.x, inew = blockIdx.y, bi = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = a.nq / a.nkv;
    const int slot = a.slot_offset + bi;
    const int base = a.pos_dev ? a.pos_dev[slot] : a.pos_scalar; // position of new token 0
    const int row = bi * a.n_new + inew;
    const int pos = base + inew;
    const int page_tokens = 1 << a.page_shift;

    __shared__ float q_s[ATT_MAX_GRP][D];
    __shared__ float knew[ATT_MAX_NEW][D];
    __shared__ float vnew[ATT_MAX_NEW][D];
    __shared__ float cm[16][ATT_MAX_GRP], cl[16][ATT_MAX_GRP];
    __shared__ float co[16][ATT_MAX_GRP][D];

    const int* pt = a.page_table + (size_t)slot * a.pages_per_slot;
    auto cache_off = [&](int t) -> size_t {
        const int page = pt[t >> a.page_shift];
        return ((((size_t)page * a.n_layers + a.layer) * a.nkv + kvh) * page_tokens + (t & (page_tokens - 1))) * D;
    };

    // ---- 1. q heads of this group and the new tokens' keys: RMSNorm + RoPE; own K/V appended to the cache ----
    const int nvec = a.new_from_raw ? grp + inew + 1 : grp;
    for (int v = wave; v < nvec; v += 4) {
        const bool is_q = v < grp;
        const int j = is_q ? inew : v - grp;                 // which new token the vector belongs to
        const float* src = a.qkv + (size_t)(bi * a.n_new + j) * a.ld_qkv + (is_q ? (kvh * grp + v) * D : (a.nq + kvh) * D);
        const int p = base + j;
        float x0 = 0.f, x1 = 0.f;
        if (lane < HALF) { x0 = src[lane]; x1 = src[lane + HALF]; }
        if (!a.new_from_raw) { // q was normalised + roped by k_rope_store
            if (lane < HALF) { q_s[v][lane] = x0; q_s[v][lane + HALF] = x1; }
            continue;
        }
        const float* nw = is_q ? a.q_norm : a.k_norm;
        if (nw != nullptr) {
            float ss = wave_sum(x0 * x0 + x1 * x1);
            // oracle order: sum of squares, mean, +eps, 1/sqrt; then w * (x * r)
            float r = 1.0f / sqrtf(ss / (float)D + a.eps);
            if (lane < HALF) { x0 = nw[lane] * (x0 * r); x1 = nw[lane + HALF] * (x1 * r); }
        }
        if (lane < HALF) {
            const float cs = a.rope_cos[(size_t)p * HALF + lane], sn = a.rope_sin[(size_t)p * HALF + lane];
            const float y0 = x0 * cs + (-x1) * sn;
            const float y1 = x1 * cs + x0 * sn;
            if (is_q) { q_s[v][lane] = y0; q_s[v][lane + HALF] = y1; }
            else {
                knew[j][lane] = y0; knew[j][lane + HALF] = y1;
                const float* vs = a.qkv + (size_t)(bi * a.n_new + j) * a.ld_qkv + (a.nq + a.nkv + kvh) * D;
                const float v0 = vs[lane], v1 = vs[lane + HALF];
                vnew[j][lane] = v0; vnew[j][lane + HALF] = v1;
                if (j == inew) { // this workgroup owns position `pos`
                    const size_t off = cache_off(p);
                    a.kcache[off + lane] = y0; a.kcache[off + lane + HALF] = y1;
                    a.vcache[off + lane] = v0; a.vcache[off + lane + HALF] = v1;
                }
            }
        }
    }
    __syncthreads();

    // ---- 2. online softmax over the visible tokens; 16 token groups of 16 lanes ----
    const int tg = wave * 4 + (lane >> 4), sub = lane & 15;
    float qr[ATT_MAX_GRP][EPL], o[ATT_MAX_GRP][EPL], mrun[ATT_MAX_GRP], lrun[ATT_MAX_GRP];
#pragma unroll
    for (int h = 0; h < ATT_MAX_GRP; ++h) {
        mrun[h] = -INFINITY; lrun[h] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { qr[h][e] = h < grp ? q_s[h][sub * EPL + e] : 0.f; o[h][e] = 0.f; }
    }
    int t_lo = 0;
    if (a.window > 0 && pos - a.window + 1 > 0) t_lo = pos - a.window + 1;
    const int n_cache = a.new_from_raw ? base : pos + 1; // tokens < n_cache come from the cache
    for (int t = t_lo + tg; t <= pos; t += 16) {
        float kv[EPL], vv[EPL];
        if (t < n_cache) {
            const size_t off = cache_off(t) + sub * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e) { kv[e] = a.kcache[off + e]; vv[e] = a.vcache[off + e]; }
        } else {
            const int j = t - base;
#pragma unroll
            for (int e = 0; e < EPL; ++e) { kv[e] = knew[j][sub * EPL + e]; vv[e] = vnew[j][sub * EPL + e]; }
        }
#pragma unroll
        for (int h = 0; h < ATT_MAX_GRP; ++h) {
            if (h < grp) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) s = fmaf(qr[h][e], kv[e], s);
                s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
                s *= a.scale;
                const float mn = fmaxf(mrun[h], s);
                const float corr = expf(mrun[h] - mn); // exp(-inf) = 0 on the first token
                const float p = expf(s - mn);
                lrun[h] = lrun[h] * corr + p;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[h][e] = o[h][e] * corr + p * vv[e];
                mrun[h] = mn;
            }
        }
    }
    // ---- 3. combine the 16 token groups ----
#pragma unroll
    for (int h = 0; h < ATT_MAX_GRP; ++h) {
        if (h < grp) {
            if (sub == 0) { cm[tg][h] = mrun[h]; cl[tg][h] = lrun[h]; }
#pragma unroll
            for (int e = 0; e < EPL; ++e) co[tg][h][sub * EPL + e] = o[h][e];
        }
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
            const float w = cm[g][h] == -INFINITY ? 0.f : expf(cm[g][h] - mx);
            L += w * cl[g][h];
            O += w * co[g][h][e];
        }
        a.out[(size_t)row * a.ld_out + (kvh * grp + h) * D + e] = O / L;
    }
}

void launch_attn(const AttnArgs& a, hipStream_t s) {
    const int grp = a.nq / a.nkv;
    if (grp < 1 || grp > ATT_MAX_GRP || a.nq % a.nkv) throw Error("attn: unsupported GQA group size");
    if (a.n_new > ATT_MAX_NEW && a.new_from_raw) throw Error("attn: too many new tokens per launch");
    dim3 grid(a.nkv, a.n_new, a.nb);
    if (a.d == 128) hipLaunchKernelGGL((k_attn<128>), grid, dim3(256), 0, s, a);
    else if (a.d == 64) hipLaunchKernelGGL((k_attn<64>), grid, dim3(256), 0, s, a);
    else if (a.d == 16) hipLaunchKernelGGL((k_attn<16>), grid, dim3(256), 0, s, a);
    else throw Error("attn: head_dim must be 16, 64 or 128");
}

// ================================================================================================
// k_sample — temperature / top-k / top-p sampling (reference src/tts_onnx.cpp:878-950) on device,
// one workgroup per batch row, V <= 4096.
// ================================================================================================
#define SAMP_MAXV 4096
#define SAMP_PER (SAMP_MAXV / 256)

static __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static __host__ __device__ __forceinline__ uint64_t mix64_hd(uint64_t z) {
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

static __device__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
static __device__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void k_sample(SampleArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int V = a.V;
    __shared__ float red[4];
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int cand_idx[SAMP_MAXV];
    __shared__ float cand_p[SAMP_MAXV];
    __shared__ float sorted_p[SAMP_MAXV];
    __shared__ int sh_i[4];
    __shared__ float sh_f[2];

    float temperature = a.temperature, top_p = a.top_p, u = a.u;
    int top_k = a.top_k, suppress = a.suppress, keep_eos = 1;
    int frame = 0;
    SlotState* st = a.st ? a.st + b : nullptr;
    if (st) {
        if (!st->active) return;
        if (a.group == 0 && !st->finished && st->n_frames >= st->max_frames) st->finished = 1; // tid-uniform value; benign race
        __syncthreads();
        if (st->finished) return;
        temperature = st->temperature; top_p = st->top_p; top_k = st->top_k;
        frame = st->n_frames;
        u = rng_uniform_dev(st->seed, st->stream_id, (uint32_t)frame, (uint32_t)a.group);
        suppress = a.group == 0;
        keep_eos = !st->ignore_eos;
    }

    // ---- load: thread owns indices [tid*PER, tid*PER+PER) so compaction preserves index order ----
    const int PER = (V + 255) / 256;
    float x[SAMP_PER];
    const float* lg = a.logits + (size_t)b * a.ld;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) {
        const int i = tid * PER + j;
        float v = -INFINITY;
        if (j < PER && i < V) {
            v = lg[i];
            if (suppress && i >= a.sup_begin && i < a.sup_end && !(i == a.eos_id && keep_eos)) v = -INFINITY; // :803-807
            if (temperature > 0.0f && temperature != 1.0f) v = v / temperature;                              // :882-884
        }
        x[j] = v;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) mx = fmaxf(mx, x[j]);
    mx = block_max(mx, red);

    // ---- top-k threshold = k-th largest value, ties kept (:917-927) ----
    float thr = -INFINITY;
    if (top_k > 0 && top_k < V) {
        if (top_k == 1) thr = mx;
        else { // 4-pass MSB radix select on order-preserving keys
            uint32_t prefix = 0, mask = 0;
            int remaining = top_k;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                hist[tid] = 0;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < SAMP_PER; ++j) {
                    const int i = tid * PER + j;
                    if (j < PER && i < V) {
                        const uint32_t key = fkey(x[j]);
                        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
                    }
                }
                __syncthreads();
                // suffix sums: scan[d] = #keys with digit >= d
                scan[tid] = hist[tid];
                __syncthreads();
                for (int off = 1; off < 256; off <<= 1) {
                    int v = scan[tid] + (tid + off < 256 ? scan[tid + off] : 0);
                    __syncthreads();
                    scan[tid] = v;
                    __syncthreads();
                }
                // digit d is selected when scan[d] >= remaining > scan[d+1]
                const int above = tid + 1 < 256 ? scan[tid + 1] : 0;
                if (scan[tid] >= remaining && above < remaining) { sh_i[0] = tid; sh_i[1] = above; }
                __syncthreads();
                prefix |= (uint32_t)sh_i[0] << shift;
                mask |= 255u << shift;
                remaining -= sh_i[1];
                __syncthreads();
            }
            // prefix is the key of the k-th largest element
            const uint32_t ku = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
            thr = __uint_as_float(ku);
        }
    }

    // ---- ordered compaction of survivors (x >= thr and finite) ----
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) cnt += (x[j] >= thr && x[j] != -INFINITY) ? 1 : 0;
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { // inclusive prefix sum
        int v = scan[tid] + (tid >= off ? scan[tid - off] : 0);
        __syncthreads();
        scan[tid] = v;
        __syncthreads();
    }
    const int n_kept = scan[255];
    int wpos = scan[tid] - cnt;
    // softmax numerators (:907-915): exp(x - max); dropped entries contribute exp(-inf) = 0
    float esum = 0.f;
#pragma unroll
    for (int j = 0; j < SAMP_PER; ++j) {
        if (x[j] >= thr && x[j] != -INFINITY) {
            const float e = expf(x[j] - mx);
            cand_idx[wpos] = tid * PER + j;
            cand_p[wpos] = e;
            ++wpos;
            esum += e;
        }
    }
    esum = block_sum(esum, red);
    for (int c = tid; c < n_kept; c += 256) cand_p[c] = cand_p[c] / esum;
    __syncthreads();

    // ---- top-p (:929-950): sort descending (ties: lower index first), keep through the first cumsum > p ----
    if (top_p < 1.0f) {
        for (int c = tid; c < n_kept; c += 256) {
            const float pc = cand_p[c];
            int rank = 0;
            for (int o2 = 0; o2 < n_kept; ++o2) {
                const float po = cand_p[o2];
                rank += (po > pc || (po == pc && o2 < c)) ? 1 : 0;
            }
            sorted_p[rank] = pc;
        }
        __syncthreads();
        if (tid == 0) {
            float cum = 0.f;
            int cutoff = n_kept;
            for (int i = 0; i < n_kept; ++i) { cum += sorted_p[i]; if (cum > top_p) { cutoff = i + 1; break; } }
            sh_i[2] = cutoff;
        }
        __syncthreads();
        const int cutoff = sh_i[2];
        // keep candidate c iff rank(c) < cutoff; recompute rank (n_kept is ~top_k)
        float s2 = 0.f;
        for (int c = tid; c < n_kept; c += 256) {
            const float pc = cand_p[c];
            int rank = 0;
            for (int o2 = 0; o2 < n_kept; ++o2) {
                const float po = cand_p[o2];
                rank += (po > pc || (po == pc && o2 < c)) ? 1 : 0;
            }
            sorted_p[c] = rank < cutoff ? pc : 0.f; // now indexed by candidate
        }
        __syncthreads();
        for (int c = tid; c < n_kept; c += 256) { cand_p[c] = sorted_p[c]; s2 += sorted_p[c]; }
        s2 = block_sum(s2, red);
        if (s2 > 0.f) for (int c = tid; c < n_kept; c += 256) cand_p[c] = cand_p[c] / s2; // :893-898
        __syncthreads();
    }

    // ---- draw: inverse CDF in index order ----
    if (tid == 0) {
        float total = 0.f;
        for (int c = 0; c < n_kept; ++c) total += cand_p[c];
        const float target = u * total;
        float cum = 0.f;
        int pick = -1, last = -1;
        for (int c = 0; c < n_kept; ++c) {
            if (cand_p[c] > 0.f) { last = c; cum += cand_p[c]; if (cum > target) { pick = c; break; } }
        }
        if (pick < 0) pick = last;
        sh_i[3] = pick >= 0 ? cand_idx[pick] : 0;
    }
    __syncthreads();
    const int tok = sh_i[3];

    if (!st) { if (tid == 0) a.token_out[b] = tok; return; }

    // ---- fused epilogue of the generation loop (tts_onnx.cpp:812-842, 864-868) ----
    if (a.group == 0 && tok == a.eos_id) { if (tid == 0) st->finished = 1; return; } // :812 — no frame recorded
    if (tid == 0) a.codes[((size_t)b * a.max_frames_cap + frame) * a.n_groups + a.group] = tok;
    const bf16_t* er = a.embed + (size_t)tok * a.H;
    const bool last_group = a.group == a.n_groups - 1;
    const float* text = nullptr;
    if (last_group) text = frame < st->trailing_len ? a.trailing + ((size_t)b * a.max_trailing + frame) * a.H : a.tts_pad; // :833-842
    for (int h = tid; h < a.H; h += 256) {
        const float e = __uint_as_float((uint32_t)er[h] << 16);
        if (a.x_next) a.x_next[(size_t)b * a.ld_xnext + h] = e;
        float sacc = a.group == 0 ? e : a.sum[(size_t)b * a.H + h] + e; // fp32, order code0, sub0..sub14 (:824-830)
        if (last_group) a.x_talk[(size_t)b * a.H + h] = sacc + text[h];
        else a.sum[(size_t)b * a.H + h] = sacc;
    }
    if (last_group && tid == 0) {
        st->n_frames = frame + 1;
        a.talker_pos[b] = st->prompt_len + frame; // position of the token the talker decodes next
    }
}

void launch_sample(const SampleArgs& a, hipStream_t s) {
    if (a.V > SAMP_MAXV) throw Error("sample: vocabulary larger than 4096");
    hipLaunchKernelGGL(k_sample, dim3(a.nb), dim3(256), 0, s, a);
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

__global__ void k_count_active(const SlotState* st, int nb, int32_t* out) {
    int n = 0;
    for (int b = 0; b < nb; ++b) n += (st[b].active && !st[b].finished) ? 1 : 0;
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
