// q3_speaker_kernels.hip — kernels of the ECAPA-TDNN speaker encoder (voice-clone path, SURVEY.md 8f-2;
// the reference runs speaker_encoder.onnx through ORT, src/tts_onnx.cpp:367-403).  One pass per reference
// clip (not per frame): ~5 MFLOP per mel frame, so these are plain fp32 FMA kernels with LDS-staged input
// tiles — the work is far below anything worth an MFMA pipeline.  Activations are time-major [T][C].
#include "q3_common.h"

namespace q3 {

// "same" Conv1d with reflect padding: y[t][co] = act(b[co] + sum_{ci,j} W[j][ci][co] * in[reflect(t + (j - k/2) dil)][ci]),
// in = x (+ x2).  Tile: 64 output channels x 16 time steps per workgroup, 32 input channels per LDS stage.
__global__ __launch_bounds__(256) void k_spk_conv(SpkConvArgs a) {
    __shared__ float xs[5][16][33];
    const int co = blockIdx.x * 64 + (threadIdx.x & 63), tg = threadIdx.x >> 6;
    const int t0 = blockIdx.y * 16, half = a.k / 2;
    float acc[4] = { 0.f, 0.f, 0.f, 0.f };
    for (int c0 = 0; c0 < a.Cin; c0 += 32) {
        for (int e = threadIdx.x; e < a.k * 512; e += 256) {
            const int c = e & 31, tt = (e >> 5) & 15, j = e >> 9;
            const int t = t0 + tt;
            float v = 0.f;
            if (t < a.T && c0 + c < a.Cin) {
                int src = t + (j - half) * a.dil;
                src = src < 0 ? -src : (src >= a.T ? 2 * (a.T - 1) - src : src);
                v = a.x_channel_major ? a.x[(size_t)(c0 + c) * a.ldx + src] : a.x[(size_t)src * a.ldx + c0 + c];
                if (a.x2) v += a.x2[(size_t)src * a.ldx2 + c0 + c];
            }
            xs[j][tt][c] = v;
        }
        __syncthreads();
        if (co < a.Cout) {
            const int cn = a.Cin - c0 < 32 ? a.Cin - c0 : 32;
            for (int c = 0; c < cn; ++c)
                for (int j = 0; j < a.k; ++j) {
                    const float w = a.W[((size_t)j * a.Cin + c0 + c) * a.Cout + co];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(w, xs[j][tg * 4 + q][c], acc[q]);
                }
        }
        __syncthreads();
    }
    if (co >= a.Cout) return;
    const float b = a.bias[co];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = t0 + tg * 4 + q;
        if (t >= a.T) continue;
        float v = acc[q] + b;
        if (a.act >= 1) v = v > 0.f ? v : 0.f;
        if (a.act == 2) v = tanhf(v);
        a.y[(size_t)t * a.ldy + co] = v;
    }
}

// torch Conv1d weight [Cout][Cin][k] -> [k][Cin][Cout]
__global__ void k_spk_repack(const float* w, float* out, int cout, int cin, int k) {
    const size_t n = (size_t)cout * cin * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout), ci = (int)((i / cout) % cin), j = (int)(i / ((size_t)cout * cin));
        out[i] = w[((size_t)co * cin + ci) * k + j];
    }
}

// per-channel reductions over time; 64 channels x 4 time lanes per workgroup, fixed combination order
template <typename F>
static __device__ __forceinline__ float col_reduce(int T, int tl, F f, float (*red)[64], int cl, bool is_max) {
    float s = is_max ? -INFINITY : 0.f;
    for (int t = tl; t < T; t += 4) { const float v = f(t); s = is_max ? fmaxf(s, v) : s + v; }
    red[tl][cl] = s;
    __syncthreads();
    const float r = is_max ? fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl])) : ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    __syncthreads();
    return r;
}

// mean[c] (and, if sd != null, sqrt(max(mean of squared deviations, 1e-12))) of x[T][ld]
__global__ __launch_bounds__(256) void k_spk_colstats(const float* x, int ld, int T, int C, float* mean, float* sd) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, ch = blockIdx.x * 64 + cl;
    const int cc = ch < C ? ch : C - 1;
    const float mu = col_reduce(T, tl, [&](int t) { return x[(size_t)t * ld + cc]; }, red, cl, false) / (float)T;
    float var = 0.f;
    if (sd) var = col_reduce(T, tl, [&](int t) { const float d = x[(size_t)t * ld + cc] - mu; return d * d; }, red, cl, false) / (float)T;
    if (ch < C && tl == 0) {
        mean[ch] = mu;
        if (sd) sd[ch] = sqrtf(var > 1e-12f ? var : 1e-12f);
    }
}

// squeeze-excitation gate + block residual: h[t][c] = y[t][c] * sigmoid(g[c]) + h[t][c]; the same value goes to cat[t][c]
__global__ void k_spk_se_gate(const float* y, const float* g, float* h, float* cat, int ld_cat, int T, int C) {
    const size_t n = (size_t)T * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const float v = y[i] * (1.0f / (1.0f + expf(-g[c]))) + h[i];
        h[i] = v;
        cat[t * ld_cat + c] = v;
    }
}

// attention input of the pooling layer: rows [x[t] | mean | sd]  (C each)
__global__ void k_spk_asp_input(const float* x, const float* mean, const float* sd, float* out, int T, int C) {
    const size_t n = (size_t)T * 3 * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (3 * C));
        const size_t t = i / (3 * C);
        out[i] = c < C ? x[t * C + c] : (c < 2 * C ? mean[c - C] : sd[c - 2 * C]);
    }
}

// attentive statistics: per channel, softmax over time of the scores s[T][C], then the weighted mean and standard
// deviation of x[T][C]; out[c] = mean, out[C + c] = sd
__global__ __launch_bounds__(256) void k_spk_asp_pool(const float* s, const float* x, int T, int C, float* out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, ch = blockIdx.x * 64 + cl;
    const int cc = ch < C ? ch : C - 1;
    const float mx = col_reduce(T, tl, [&](int t) { return s[(size_t)t * C + cc]; }, red, cl, true);
    const float den = col_reduce(T, tl, [&](int t) { return expf(s[(size_t)t * C + cc] - mx); }, red, cl, false);
    const float mu = col_reduce(T, tl, [&](int t) { return expf(s[(size_t)t * C + cc] - mx) / den * x[(size_t)t * C + cc]; }, red, cl, false);
    const float var = col_reduce(T, tl, [&](int t) { const float d = x[(size_t)t * C + cc] - mu; return expf(s[(size_t)t * C + cc] - mx) / den * d * d; }, red, cl, false);
    if (ch < C && tl == 0) {
        out[ch] = mu;
        out[C + ch] = sqrtf(var > 1e-12f ? var : 1e-12f);
    }
}

void launch_spk_conv(const SpkConvArgs& a, hipStream_t s) {
    if (a.k < 1 || a.k > 5 || !(a.k & 1)) throw Error("speaker conv: kernel size must be 1, 3 or 5");
    if (a.T < 1 || (a.k > 1 && (a.k / 2) * a.dil >= a.T)) throw Error("speaker conv: reflect padding needs more frames");
    hipLaunchKernelGGL(k_spk_conv, dim3((a.Cout + 63) / 64, (a.T + 15) / 16), dim3(256), 0, s, a);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_spk_repack(const float* w, float* out, int cout, int cin, int k, hipStream_t s) {
    const size_t n = (size_t)cout * cin * k;
    hipLaunchKernelGGL(k_spk_repack, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, w, out, cout, cin, k);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_spk_colstats(const float* x, int ld, int T, int C, float* mean, float* sd, hipStream_t s) {
    hipLaunchKernelGGL(k_spk_colstats, dim3((C + 63) / 64), dim3(256), 0, s, x, ld, T, C, mean, sd);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_spk_se_gate(const float* y, const float* g, float* h, float* cat, int ld_cat, int T, int C, hipStream_t s) {
    const size_t n = (size_t)T * C;
    hipLaunchKernelGGL(k_spk_se_gate, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, y, g, h, cat, ld_cat, T, C);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_spk_asp_input(const float* x, const float* mean, const float* sd, float* out, int T, int C, hipStream_t s) {
    const size_t n = (size_t)T * 3 * C;
    hipLaunchKernelGGL(k_spk_asp_input, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, x, mean, sd, out, T, C);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_spk_asp_pool(const float* sc, const float* x, int T, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_spk_asp_pool, dim3((C + 63) / 64), dim3(256), 0, s, sc, x, T, C, out);
    Q3_HIP_CHECK(hipGetLastError());
}

} // namespace q3
