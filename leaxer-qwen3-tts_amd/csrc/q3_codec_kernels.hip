// q3_codec_kernels.hip — gfx950 kernels of the 12 Hz codec decoder (the reference's
// tokenizer12hz_decode.onnx session, src/tts_onnx.cpp:759-776): codebook-embedding mean,
// sliding-window pre-transformer, ConvNeXt upsampling and the SnakeBeta transposed-conv decoder.
//
// Every convolution / linear layer is one implicit GEMM on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 155 TF peak on MI355X), so the PCM matches the
// fp32 oracle to rounding.  Activations are time-major [T][C]; a k-tap causal conv is k shifted
// GEMMs accumulated in registers; a stride-s transposed conv is s phase GEMMs (blockIdx.z).
// SnakeBeta is applied in the PRODUCER's epilogue (second output), never on the k-times-re-read
// operand loads.
#include "q3_common.h"

namespace q3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static __device__ __forceinline__ float silu2_f(float x) { return x / (1.0f + expf(-x)); }

struct ConvKArgs {
    const float* in; int T_in, C_in;
    float* out; int T_out, C_out;
    const float* W;       // [taps][C_out][C_in]
    const float* bias;
    int taps, dil, transposed, stride, left;
    const float* res; const float* res_scale; const float* mul;
    int act, clamp;
    float* out2; const float* s2_alpha; const float* s2_beta; // out2 = snake(out value)
};

#define CT_M 64
#define CT_N 64
#define CT_K 32
#define CT_LD 33 // padded LDS row (floats): ds_read_b32 of a column is conflict-free

__global__ __launch_bounds__(256) void k_conv_mfma(ConvKArgs a) {
    __shared__ float As[CT_M][CT_LD];
    __shared__ float Bs[CT_N][CT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * CT_M, co0 = blockIdx.y * CT_N, phase = blockIdx.z;
    const int NT = a.transposed ? a.taps / a.stride : a.taps;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // staging assignment: 64 rows x 8 float4 per tile -> 2 float4 per thread per tile
    const int srow = tid >> 3, scol = (tid & 7) * 4; // rows srow and srow+32

    for (int ti = 0; ti < NT; ++ti) {
        const int shift = a.transposed ? ti : (a.taps - 1 - ti) * a.dil;
        const int wtap = a.transposed ? phase + ti * a.stride : ti;
        const float* Wt = a.W + (size_t)wtap * a.C_out * a.C_in;
        for (int ci0 = 0; ci0 < a.C_in; ci0 += CT_K) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = srow + 32 * h;
                const int src = m0 + r - shift;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src >= 0 && src < a.T_in && ci0 + scol < a.C_in) v = *reinterpret_cast<const float4*>(a.in + (size_t)src * a.C_in + ci0 + scol);
                As[r][scol] = v.x; As[r][scol + 1] = v.y; As[r][scol + 2] = v.z; As[r][scol + 3] = v.w;
                const int co = co0 + r;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < a.C_out && ci0 + scol < a.C_in) w = *reinterpret_cast<const float4*>(Wt + (size_t)co * a.C_in + ci0 + scol);
                Bs[r][scol] = w.x; Bs[r][scol + 1] = w.y; Bs[r][scol + 2] = w.z; Bs[r][scol + 3] = w.w;
            }
            __syncthreads();
            const float* ap = &As[wr * 32 + (lane & 31)][lane >> 5];
            const float* bp = &Bs[wc * 32 + (lane & 31)][lane >> 5];
#pragma unroll
            for (int kk = 0; kk < CT_K / 2; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * 2], bp[kk * 2], acc, 0, 0, 0);
            __syncthreads();
        }
    }

    // epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int co = co0 + wc * 32 + (lane & 31);
    if (co >= a.C_out) return;
    const float bias = a.bias ? a.bias[co] : 0.f;
    const float rs = a.res_scale ? a.res_scale[co] : 1.f;
    float ea = 0.f, ib = 0.f;
    if (a.out2) { ea = expf(a.s2_alpha[co]); ib = 1.0f / (expf(a.s2_beta[co]) + 0.000000001f); }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        int t = m;
        if (a.transposed) t = m * a.stride + phase - a.left;
        if (t < 0 || t >= a.T_out) continue;
        if (a.transposed && m >= a.T_in + NT - 1) continue;
        float v = acc[reg] + bias;
        if (a.act == 1) v = gelu_f(v);
        else if (a.act == 2) v = silu2_f(v);
        const size_t o = (size_t)t * a.C_out + co;
        if (a.mul) v = v * a.mul[o];
        if (a.res_scale) v = rs * v;
        if (a.res) v = a.res[o] + v;
        if (a.clamp) v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        if (a.out) a.out[o] = v;
        if (a.out2) { const float sn = sinf(v * ea); a.out2[o] = v + ib * (sn * sn); }
    }
}

void launch_conv(const ConvArgs& c, hipStream_t s) {
    if (c.C_in % 4 != 0) throw Error("conv: C_in must be a multiple of 4");
    ConvKArgs a;
    a.in = c.in; a.T_in = c.T_in; a.C_in = c.C_in; a.out = c.out; a.T_out = c.T_out; a.C_out = c.C_out;
    a.W = c.W; a.bias = c.bias; a.taps = c.taps; a.dil = c.dil; a.transposed = c.transposed; a.stride = c.stride; a.left = c.left;
    a.res = c.res; a.res_scale = c.res_scale; a.mul = c.mul; a.act = c.act; a.clamp = c.clamp;
    a.out2 = c.out2; a.s2_alpha = c.snake_alpha; a.s2_beta = c.snake_beta;
    if (c.transposed && c.taps % c.stride != 0) throw Error("conv: transposed kernel must be a multiple of the stride");
    const int rows = c.transposed ? c.T_in + c.taps / c.stride - 1 : c.T_out;
    dim3 grid((rows + CT_M - 1) / CT_M, (c.C_out + CT_N - 1) / CT_N, c.transposed ? c.stride : 1);
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_conv_mfma, grid, dim3(256), 0, s, a);
}

// ---- weight repack: PyTorch Conv1d [co][ci][k] / ConvTranspose1d [ci][co][k] -> [k][co][ci] ----
__global__ void k_repack_conv(const float* w, float* out, int cin, int cout, int k, int transposed) {
    const int64_t n = (int64_t)cin * cout * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const int co = (int)((i / cin) % cout);
        const int t = (int)(i / ((int64_t)cin * cout));
        out[i] = transposed ? w[((size_t)ci * cout + co) * k + t] : w[((size_t)co * cin + ci) * k + t];
    }
}
void launch_repack_conv(const float* w, float* out, int cin, int cout, int k, int transposed, hipStream_t s) {
    hipLaunchKernelGGL(k_repack_conv, dim3(1024), dim3(256), 0, s, w, out, cin, cout, k, transposed);
}

// ---- code_embedding(codes + g*codebook).mean over the G quantizers (Code2Wav.forward) ----
__global__ void k_code_embed_mean(const float* table, const int32_t* codes, int G, int codebook, int C, float* out) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += table[((size_t)g * codebook + codes[(size_t)t * G + g]) * C + c];
        out[(size_t)t * C + c] = s / (float)G;
    }
}
void launch_code_embed_mean(const float* table, const int32_t* codes, int F, int G, int codebook, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_code_embed_mean, dim3(F), dim3(256), 0, s, table, codes, G, codebook, C, out);
}

static __device__ float block_sum256(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_rmsnorm_rows(const float* x, const float* w, float eps, int C, float* out) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * C;
    float ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) ss += xr[c] * xr[c];
    ss = block_sum256(ss, red);
    const float r = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)blockIdx.x * C + c] = w[c] * (xr[c] * r);
}
void launch_rmsnorm_rows(const float* x, const float* w, float eps, int rows, int C, float* out, hipStream_t s) {
    if (rows > 0) hipLaunchKernelGGL(k_rmsnorm_rows, dim3(rows), dim3(256), 0, s, x, w, eps, C, out);
}

// RoPE on q (in place) and k, K/V copied into the attention cache layout [kvh][P][d] (one layer at a time)
__global__ void k_rope_store(float* qkv, int ld, int nq, int nkv, int d, const float* cs, const float* sn,
                             float* kc, float* vc, int P) {
    const int t = blockIdx.x, half = d / 2;
    float* row = qkv + (size_t)t * ld;
    for (int i = threadIdx.x; i < (nq + nkv) * half; i += blockDim.x) {
        const int h = i / half, e = i % half;
        float* v = row + (size_t)h * d;
        const float c = cs[(size_t)t * half + e], s = sn[(size_t)t * half + e];
        const float x0 = v[e], x1 = v[e + half];
        const float y0 = x0 * c + (-x1) * s, y1 = x1 * c + x0 * s;
        if (h < nq) { v[e] = y0; v[e + half] = y1; }
        else {
            const int kh = h - nq;
            float* dst = kc + ((size_t)kh * P + t) * d;
            dst[e] = y0; dst[e + half] = y1;
        }
    }
    for (int i = threadIdx.x; i < nkv * d; i += blockDim.x) {
        const int kh = i / d, e = i % d;
        vc[((size_t)kh * P + t) * d + e] = row[(size_t)(nq + nkv + kh) * d + e];
    }
}
void launch_rope_store(float* qkv, int ld, int T, int nq, int nkv, int d, const float* cs, const float* sn,
                       float* kc, float* vc, int P, hipStream_t s) {
    if (T > 0) hipLaunchKernelGGL(k_rope_store, dim3(T), dim3(256), 0, s, qkv, ld, nq, nkv, d, cs, sn, kc, vc, P);
}

// ConvNeXt front half: depthwise causal k7 conv + LayerNorm(eps 1e-6) over channels, one row per block
__global__ __launch_bounds__(256) void k_dwconv_ln(const float* x, int T, int C, const float* dw_w, const float* dw_b,
                                                   const float* ln_w, const float* ln_b, float* out) {
    __shared__ float red[4];
    extern __shared__ float hbuf[];
    const int t = blockIdx.x;
    float s1 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = dw_b[c];
        for (int tap = 0; tap < 7; ++tap) {
            const int ts = t - (6 - tap);
            if (ts >= 0) acc += dw_w[c * 7 + tap] * x[(size_t)ts * C + c];
        }
        hbuf[c] = acc;
        s1 += acc;
    }
    const float mean = block_sum256(s1, red) / (float)C;
    float s2 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float d = hbuf[c] - mean; s2 += d * d; }
    const float var = block_sum256(s2, red) / (float)C;
    const float r = 1.0f / sqrtf(var + 1e-6f);
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)t * C + c] = (hbuf[c] - mean) * r * ln_w[c] + ln_b[c];
}
void launch_dwconv_ln(const float* x, int T, int C, const float* dw_w, const float* dw_b, const float* ln_w,
                      const float* ln_b, float* out, hipStream_t s) {
    if (T > 0) hipLaunchKernelGGL(k_dwconv_ln, dim3(T), dim3(256), (size_t)C * sizeof(float), s, x, T, C, dw_w, dw_b, ln_w, ln_b, out);
}

} // namespace q3
