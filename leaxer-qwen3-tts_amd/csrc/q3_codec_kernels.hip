// q3_codec_kernels.hip — gfx950 kernels of the 12 Hz codec decoder (the reference's
// tokenizer12hz_decode.onnx session, src/tts_onnx.cpp:759-776): codebook-embedding mean,
// sliding-window pre-transformer, ConvNeXt upsampling and the SnakeBeta transposed-conv decoder.
//
// Every convolution / linear layer is one implicit GEMM on the matrix cores.  Default path, k_conv_split: both operands as fp16
// (hi, lo) planes on v_mfma_f32_32x32x16_f16 with fp32 accumulate — three products per fp32 product (lo.hi + hi.lo + hi.hi), two when
// the weight is exact in fp16 (every bf16-origin tensor: its lo plane is empty) — PCM within 2e-7 RMS of the fp32 oracle.  k_conv_mfma,
// the exact-fp32 path (v_mfma_f32_32x32x2_f32: fp32 fmaf chains, 155 TF peak on MI355X), serves Q3TTS_FLAG_FP32_CODEC and channel
// counts that are not multiples of 32.  Activations are time-major [T][C]; a k-tap causal conv is k shifted GEMMs accumulated in
// registers; a stride-s transposed conv is s phase GEMMs (blockIdx.z).  SnakeBeta is applied in the PRODUCER's epilogue (second
// output), never on the k-times-re-read operand loads.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "q3_common.h"

namespace q3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DQ3_SAMPLE_PROF: wall-clock stamps inside k_conv_split (workgroup (0,0,0), thread 0), tools/ only
#ifdef Q3_SAMPLE_PROF
__device__ long long g_conv_prof[64];
void conv_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_prof), sizeof(long long) * 64); }
// the stamped launch: the pre-transformer's down projection (C_in 3072 -> C_out 1024, one tap), i.e. the longest serial K walk
// (-DQ3_CONV_PROF_SEL="..." / -DQ3_CONV_PROF_WG=n pick another launch shape / another workgroup, e.g. a mid-grid one of a 7-tap conv)
#ifndef Q3_CONV_PROF_SEL
#define Q3_CONV_PROF_SEL (a.taps == 1 && a.C_in == 3072 && a.C_out == 1024)
#endif
#ifndef Q3_CONV_PROF_WG
#define Q3_CONV_PROF_WG 0
#endif
// slots 0..45: the loop (4 per iteration, first 11 iterations); 48..55: the fused tail (4 per row block); 60..63: after the loop.
// Pinned on both sides: the clock read has no dependences, left alone it is scheduled far from where it is written.
#define CP_MARK_IF(k, lim) do { __builtin_amdgcn_sched_barrier(0); \
        if ((k) < (lim) && blockIdx.x == Q3_CONV_PROF_WG && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (Q3_CONV_PROF_SEL)) g_conv_prof[k] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define CP_MARK(k) CP_MARK_IF(k, 64)
#define CP_MARK_L(k) CP_MARK_IF(k, 46)
#else
#define CP_MARK(k) do { } while (0)
#define CP_MARK_L(k) do { } while (0)
#endif

static __device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static __device__ __forceinline__ float silu2_f(float x) { return x / (1.0f + expf(-x)); }
// sin(x)^2 for SnakeBeta: three-term FMA reduction by pi/2 (exact products; good to |x| ~ 1e5, far past anything a*x reaches here),
// then the fdlibm single-precision kernels on [-pi/4, pi/4]; the quadrant only picks sine or cosine since the square drops the sign.
// ~25 instructions instead of libm sinf's ~65 (the snake is evaluated 5e9 times per 2048-frame utterance); within 2 ulp of it.
// Q3_SIN_SQ_EXACT (build knob): the 25-instruction evaluation below.  Default: the hardware's v_sin_f32 on the fractional number of
// revolutions (4 instructions).  The snake's VALU work had become the bound of the fused residual unit (two SnakeBeta evaluations per
// output element: 34k VALU cycles per tile against 28k matrix-core cycles); PCM error vs the fp32 oracle with the hardware sine is
// measured by tests/test_gpu_full.py::test_codec_split_precision_matrix_path_full_size (budget 1e-4 RMS; the exact-fp32 codec,
// Q3TTS_FLAG_FP32_CODEC, keeps libm's sinf).
#ifndef Q3_SIN_SQ_EXACT
static __device__ __forceinline__ float sin_sq(float x) {
    const float s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f));   // sin(2 pi frac(x / 2 pi))
    return s * s;
}
#else
static __device__ __forceinline__ float sin_sq(float x) {
    const float kf = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-kf, 1.57079637050628662109375f, x);
    r = fmaf(-kf, -4.37113900018624283e-8f, r);
    r = fmaf(-kf, -1.71512451000590e-15f, r);
    const float z = r * r;
    const float sp = fmaf(r * z, fmaf(z, fmaf(z, fmaf(z, 2.7183114939898219064e-6f, -1.9839334836096632650e-4f), 8.3333293858894631756e-3f), -1.6666666641626524e-1f), r);
    const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.4390448796277409065e-5f, -1.3886763774609869e-3f), 4.1666623323739063e-2f), fmaf(-0.5f, z, 1.0f));
    const float v = ((int)kf & 1) ? cp : sp;
    return v * v;
}
#endif

struct ConvKArgs {
    const float* in; int T_in, C_in;
    float* out; int T_out, C_out;
    const float* W;       // [taps][C_out][C_in]
    const float* bias;
    int taps, dil, transposed, stride, left;
    const float* res; const float* res_scale; const float* mul;
    int act, clamp;
    float* out2; const float* s2_alpha; const float* s2_beta; // out2 = snake(out value)
    const bf16_t* Wh; const bf16_t* Wl; // optional (hi, lo) fp16 planes of W * 2^k for k_conv_split (16-bit storage)
    float acc_scale;                    // 2^-k (1 on the fp32 path)
    int ksplit; float* slab;            // k_conv_split, 1-tap GEMMs: blockIdx.z = K slice, raw partial sums -> slab[z][T_out][C_out] (k_conv_finish)
    int xcd_map;                        // k_conv_split: XCD-aware tile ids (conv_tile_ids)
    int no_fast_epi;                    // A/B switch: decoder convs through the generic epilogue
    int peel_taps;                      // k_conv_split: >= 3-tap convs request the next chunk's rows two iterations ahead (A/B switch)
    // SnakeBeta outputs between the decoder's convs as (hi, lo) fp16 pairs instead of fp32: every group of 4 channels of `in` / `out2`
    // holds its 4 hi halves followed by its 4 lo halves in the 16 bytes the 4 floats would take — the same addresses, the same 16-byte
    // accesses — split ONCE by the producer's epilogue instead of by every consuming workgroup while it stages its rows (2-8
    // output-channel tiles x 1.2 halo re-split each element; ~1.4 us per chunk).  (Two separate planes were tried first: the 8-byte
    // accesses made every conv 5-18 % slower.)
    int in_planes, out2_planes;
    const float* s2_pre; const float* s1_pre;   // SnakeBeta constants [2][C_out] (exp(alpha) | 1 / (exp(beta) + 1e-9)), precomputed at finalize
    int batch_tiles;                    // > 0: blockIdx.x = sequence * batch_tiles + row tile; sequences are in_ustride / T_out * C_out floats apart
    size_t in_ustride;
    // fused residual unit (k_conv_split<..., F2 = true>): second (1x1) conv behind a SnakeBeta on the first conv's output
    const bf16_t* W2h; const bf16_t* W2l; float acc_scale2; const float* bias2; const float* s1_alpha; const float* s1_beta;
    int wlo;                            // 0: every weight plane `lo` of this launch is identically zero (bf16- / fp16-origin weights) -> the WLO = false kernels
    const bf16_t* W2fh; const bf16_t* W2fl;   // fused unit: W2's planes in B-fragment order (ConvArgs::W2fh), or null
};

// batched launch: rebase the sequence-shaped pointers to this workgroup's sequence and return its row-tile index
// XCD-aware tile ids.  Workgroups leave the dispatcher in linear order (x fastest) and go round-robin to the 8 XCDs, each with its own L2.
// With the launch grid read literally, the gridDim.y x gridDim.z workgroups that walk the SAME input rows (other output-channel tiles,
// other phases of a transposed conv) are a whole grid row apart: each fetches its rows from HBM again (2-8x the input traffic; a block's
// activation, 0.2-1.5 GB, does not fit MALL either).  Remapped, every run of 8 * ny * nz consecutive workgroups covers 8 row tiles, and
// the ny * nz workgroups of one row tile sit 8 ids apart — same XCD, same moment: one fetches, the rest hit L2.
static __device__ __forceinline__ void conv_tile_ids(bool remap, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    const int nx = gridDim.x, ny = gridDim.y, nyz = ny * gridDim.z;
    if (!remap || nyz == 1) return;
    const int L = bx + nx * (by + ny * bz), gsz = 8 * nyz, full = nx >> 3;
    int G = L / gsz, w = 8;
    if (G >= full) { G = full; w = nx & 7; }              // the last nx % 8 row tiles form a narrower group
    const int r = L - G * gsz;
    bx = G * 8 + r % w;
    const int yz = r / w;
    by = yz % ny; bz = yz / ny;
}

static __device__ __forceinline__ int conv_batch_rebase(ConvKArgs& a, int bx) {
    if (a.batch_tiles > 0) {
        const int u = bx / a.batch_tiles;
        bx -= u * a.batch_tiles;
        const size_t io = (size_t)u * a.in_ustride, oo = (size_t)u * a.T_out * a.C_out;
        a.in += io;
        if (a.out) a.out += oo;
        if (a.out2) a.out2 += oo;
        if (a.res) a.res += oo;
        if (a.mul) a.mul += oo;
    }
    return bx;
}

// Epilogue of one 32x32 accumulator block: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
static __device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, const f32x16 acc, int mrow0, int co, int lane, int phase, int NT) {
    if (co >= a.C_out) return;
    const float bias = a.bias ? a.bias[co] : 0.f;
    const float rs = a.res_scale ? a.res_scale[co] : 1.f;
    float ea = 0.f, ib = 0.f;
    if (a.out2) { ea = expf(a.s2_alpha[co]); ib = 1.0f / (expf(a.s2_beta[co]) + 0.000000001f); }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = mrow0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        int t = m;
        if (a.transposed) t = m * a.stride + phase - a.left;
        if (t < 0 || t >= a.T_out) continue;
        if (a.transposed && m >= a.T_in + NT - 1) continue;
        float v = acc[reg] * a.acc_scale + bias;   // acc_scale undoes the power-of-two weight pre-scale of the split path (1 otherwise)
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

    conv_epilogue(a, acc, m0 + wr * 32, co0 + wc * 32 + (lane & 31), lane, phase, NT);
}

// ================================================================================================
// k_conv_split — the same implicit GEMM on the 16-bit matrix cores with fp32-grade accuracy.  Both operands are
// split x = hi + lo into two fp16 planes (11 + 11 significant bits; the matrix core multiplies fp16 subnormals
// exactly — tools/mfma_f16_denorm.hip — so small values keep an absolute resolution of 2^-24) and three
// v_mfma_f32_32x32x16_f16 products per tile step (lo*hi + hi*lo + hi*hi, fp32 accumulate) stand in for one fp32
// product: 1/3 of the 16-bit peak = 5x the fp32 matrix peak, dropped terms 2^-22 relative.  (bf16 planes carry only
// 8 + 8 bits: measured PCM error 8e-5 RMS on the tiny golden, too close to the 1e-4 budget; fp16 planes: ~1e-6.)
// Range: weights are pre-scaled per tensor by a power of two (k_split_planes; undone in the epilogue) so their
// low plane stays normal; activations beyond +-65504 would saturate the high plane — it is clamped, the low plane
// takes the remainder (exact to +-131008), values the codec's audio-scale features never approach.
// Workgroup tile TM x TN = (WM*MB*32) x (WN*NB*32); K is walked as (C_in chunk of 32) x (tap).  The activation
// rows of a chunk are staged ONCE with their tap halo (rows m0-halo .. m0+TM) and split to hi/lo on the way into
// LDS; every tap reads the same rows at a shifted offset, so a 7-tap conv reads and splits its input once, not
// seven times.  Weight planes are pre-split, double-buffered per tap.  LDS rows are 40 halves (80 B): the 16 rows
// of a ds_read_b128 group land on disjoint banks.  Next tap's weights / next chunk's rows are in registers
// (global loads in flight) during the current tap's MFMAs.
// ================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define B3_LD 40

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// x = hi + lo in two fp16 values, two elements at a time: v_med3_f32 clamps (one instruction instead of two compare + select pairs),
// v_cvt_pk_f16_f32 rounds both to nearest-even at once.  This runs for every staged input element and for every element of the fused
// unit's intermediate: the scalar version was ~15 % of the fused tail's VALU work.
static __device__ __forceinline__ void split_f16x2(float a, float b, f16x2& hi, f16x2& lo) {
    const f32x2 c = { __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f) };
    hi = __builtin_convertvector(c, f16x2);
    const f32x2 back = __builtin_convertvector(hi, f32x2);
    const f32x2 r = { __builtin_amdgcn_fmed3f(a - back.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b - back.y, -65504.f, 65504.f) };
    lo = __builtin_convertvector(r, f16x2);
}
// Epilogue of one 32-row block of a wave's tile (32 x NB*32 outputs).  The accumulator layout (lane = column) would
// make every global access a 4-byte-per-lane, two-rows-per-instruction affair — measured: half of the decoder's time.
// The block is transposed through the wave's LDS slice instead, so each lane owns 4 consecutive channels of a row:
// residual / multiplier rows come in as 16-byte loads issued up front (clamped addresses, never conditional), results
// leave as 16-byte stores.
// The decoder's own convs (no activation, no multiplier, no LayerScale; SnakeBeta of the result for the next layer): everything the
// generic epilogue below decides per element at run time is a template parameter here, so the residual rows are plain 16-byte loads —
// ALL 16 of the block issued before the first is needed — and the row loop is straight-line.  (The generic version's `has_res ? load :
// zero` compiled to four 4-byte loads per row, each behind its own uniform branch, one group of rows ahead: 11-16 us per 32-row block
// measured with in-kernel stamps, as long as the block's whole main loop.)  Same operations in the same order as the generic path.
// RH = residual rows in flight (16: the whole block up front; 8: two halves, the second requested after the first four rows are
// done — for the fused residual unit, whose other row block's accumulators are still live).
// No branch inside the row loop: rows outside the output (tile tails, the rows a transposed conv's last phase does not own) are stored
// to a dump line instead of being predicated.  Behind a branch hipcc cannot count the stores in flight, so every wait for an
// (older) operand load — s_waitcnt counts in order — also waited for the previous rows' STORES to complete: ~0.4 us per row.
__device__ float g_conv_dump[2][64 * 4];
template <int NB, bool OUT, bool RES, int RH, bool O2P = false>
static __device__ __forceinline__ void split_epilogue_snake(const ConvKArgs& a, const f32x16 (&acc)[NB], float* stage, int mrow0, int co0w,
                                                            int lane, int phase, int NT) {
    constexpr int W = NB * 32, LDE = W + 8;
    const int c4 = lane & 31, rsel = lane >> 5;
    const int co = co0w + c4 * 4;
    const bool colok = c4 * 4 < W && co < a.C_out;
    const int coc = colok ? co : 0, lc = c4 * 4 < W ? c4 * 4 : 0;
    // output row of row pair p: t = t0 + p * tstep, valid while 0 <= t < T_out and the input row m0r + 2 p is below mlim
    const int m0r = mrow0 + rsel;
    const int t0 = a.transposed ? m0r * a.stride + phase - a.left : m0r, tstep = a.transposed ? 2 * a.stride : 2;
    const int mlim = a.transposed ? a.T_in + NT - 1 : 0x7fffffff;
    auto row_ok = [&](int p) { const int t = t0 + p * tstep; return colok && t >= 0 && t < a.T_out && m0r + 2 * p < mlim; };
    auto row_off = [&](int p) -> size_t { return row_ok(p) ? (size_t)(t0 + p * tstep) * a.C_out + coc : (size_t)coc; };   // row 0 stands in for a row outside (loads only)
    float4 resv[RH];
    auto load_res = [&](int slot0, int p0, int n) {
#pragma unroll
        for (int k = 0; k < n; ++k) resv[slot0 + k] = *reinterpret_cast<const float4*>(a.res + row_off(p0 + k));
    };
    if (RES) load_res(0, 0, RH);
    const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + coc);
    // exp(alpha) and 1 / (exp(beta) + 1e-9) come precomputed (k_snake_pre, same expressions): 8 expf + 4 divisions per block were ~10 % of the tail
    const float4 ea4 = *reinterpret_cast<const float4*>(a.s2_pre + coc), ib4 = *reinterpret_cast<const float4*>(a.s2_pre + a.C_out + coc);
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * rsel) * LDE + j * 32 + c4] = acc[j][reg];
    const float ea[4] = { ea4.x, ea4.y, ea4.z, ea4.w }, ib[4] = { ib4.x, ib4.y, ib4.z, ib4.w };
    const float bs[4] = { bias4.x, bias4.y, bias4.z, bias4.w };
    float* const dump1 = &g_conv_dump[0][lane * 4];
    float* const dump2 = &g_conv_dump[1][lane * 4];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const bool okp = row_ok(p);
        const size_t o = row_off(p);
        const float4 raw = *reinterpret_cast<const float4*>(&stage[(p * 2 + rsel) * LDE + lc]);
        float v[4] = { raw.x, raw.y, raw.z, raw.w }, s2[4];
        const float4 r4 = RES ? resv[p % RH] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float rv[4] = { r4.x, r4.y, r4.z, r4.w };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e] * a.acc_scale + bs[e];
            if (RES) x = rv[e] + x;
            v[e] = x;
            s2[e] = x + ib[e] * sin_sq(x * ea[e]);
        }
        if (OUT) *reinterpret_cast<float4*>(okp ? a.out + o : dump1) = make_float4(v[0], v[1], v[2], v[3]);
        if (O2P) {   // 4 hi halves | 4 lo halves in the 16 bytes of the 4 floats
            f16x2 h01, h23, l01, l23;
            split_f16x2(s2[0], s2[1], h01, l01);
            split_f16x2(s2[2], s2[3], h23, l23);
            *reinterpret_cast<uint4*>(okp ? a.out2 + o : dump2) =
                make_uint4(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        } else
        *reinterpret_cast<float4*>(okp ? a.out2 + o : dump2) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        if ((p & 3) == 3) {                               // groups of four rows stay groups: left alone hipcc hoists all 16 LDS reads and addresses
            __builtin_amdgcn_sched_barrier(0);
            if (RES && RH == 8 && p < 8) load_res(p - 3, p + 5, 4);      // slots of the rows just finished take rows 8..11 / 12..15
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NB, int RH = 16>
static __device__ __forceinline__ void split_epilogue_block(const ConvKArgs& a, const f32x16 (&acc)[NB], float* stage, int mrow0, int co0w,
                                                            int lane, int phase, int NT) {
    if (a.out2 && a.s2_pre && a.bias && a.act == 0 && !a.mul && !a.res_scale && !a.no_fast_epi) {   // uniform: the decoder's convs
        if (a.out2_planes) {
            if (a.res) { if (a.out) split_epilogue_snake<NB, true, true, RH, true>(a, acc, stage, mrow0, co0w, lane, phase, NT);
                         else split_epilogue_snake<NB, false, true, RH, true>(a, acc, stage, mrow0, co0w, lane, phase, NT); }
            else { if (a.out) split_epilogue_snake<NB, true, false, RH, true>(a, acc, stage, mrow0, co0w, lane, phase, NT);
                   else split_epilogue_snake<NB, false, false, RH, true>(a, acc, stage, mrow0, co0w, lane, phase, NT); }
            return;
        }
        if (a.res) { if (a.out) split_epilogue_snake<NB, true, true, RH>(a, acc, stage, mrow0, co0w, lane, phase, NT);
                     else split_epilogue_snake<NB, false, true, RH>(a, acc, stage, mrow0, co0w, lane, phase, NT); }
        else { if (a.out) split_epilogue_snake<NB, true, false, RH>(a, acc, stage, mrow0, co0w, lane, phase, NT);
               else split_epilogue_snake<NB, false, false, RH>(a, acc, stage, mrow0, co0w, lane, phase, NT); }
        return;
    }
    constexpr int W = NB * 32, LDE = W + 8;      // padded row: the two 32-lane halves of a transposing write hit disjoint banks
    const int c4 = lane & 31, rsel = lane >> 5;
    const int co = co0w + c4 * 4;
    const bool colok = c4 * 4 < W && co < a.C_out;
    const int coc = colok ? co : 0, lc = c4 * 4 < W ? c4 * 4 : 0;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bias4 = zero4, rs4 = one4, al4 = zero4, be4 = zero4;
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + coc);
    if (a.res_scale) rs4 = *reinterpret_cast<const float4*>(a.res_scale + coc);
    if (a.out2) { al4 = *reinterpret_cast<const float4*>(a.s2_alpha + coc); be4 = *reinterpret_cast<const float4*>(a.s2_beta + coc); }
    // accumulators -> LDS (transposing write), then rows come back 4 at a time; the residual / multiplier rows of the NEXT
    // group of 4 are loaded while the current group is computed (rolled loop: the activation code is emitted once)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * rsel) * LDE + j * 32 + c4] = acc[j][reg];
    const float ea[4] = { expf(al4.x), expf(al4.y), expf(al4.z), expf(al4.w) };
    const float ib[4] = { 1.0f / (expf(be4.x) + 0.000000001f), 1.0f / (expf(be4.y) + 0.000000001f), 1.0f / (expf(be4.z) + 0.000000001f),
                          1.0f / (expf(be4.w) + 0.000000001f) };
    const float bs[4] = { bias4.x, bias4.y, bias4.z, bias4.w }, rs[4] = { rs4.x, rs4.y, rs4.z, rs4.w };
    auto row_off = [&](int p, bool* okp) -> size_t {
        const int m = mrow0 + p * 2 + rsel;
        int t = a.transposed ? m * a.stride + phase - a.left : m;
        *okp = colok && t >= 0 && t < a.T_out && !(a.transposed && m >= a.T_in + NT - 1);
        t = t < 0 ? 0 : (t < a.T_out ? t : a.T_out - 1);
        return (size_t)t * a.C_out + coc;
    };
    // The per-element chain is branch-free: absent operands are neutral (bias 0, multiplier 1, scale 1, residual 0), the activation
    // and the SnakeBeta second output are compile-time variants picked ONCE per block — six uniform branches per element cost a lone
    // wave more than the arithmetic (23 us per 64 x 128 tile before).
    const bool has_res = a.res != nullptr, has_mul = a.mul != nullptr;
    auto rows = [&](auto act_tag, auto out2_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool OUT2 = decltype(out2_tag)::value;
        float4 resv[4], mulv[4], resn[4], muln[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool okk;
            const size_t o = row_off(k, &okk);
            resv[k] = has_res ? *reinterpret_cast<const float4*>(a.res + o) : zero4;
            mulv[k] = has_mul ? *reinterpret_cast<const float4*>(a.mul + o) : one4;
        }
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                      // prefetch the next group (clamped to the last one on the final trip)
                bool okk;
                const size_t o = row_off((q < 3 ? q + 1 : q) * 4 + k, &okk);
                resn[k] = has_res ? *reinterpret_cast<const float4*>(a.res + o) : zero4;
                muln[k] = has_mul ? *reinterpret_cast<const float4*>(a.mul + o) : one4;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = q * 4 + k;
                bool okp;
                const size_t o = row_off(p, &okp);
                const float4 raw = *reinterpret_cast<const float4*>(&stage[(p * 2 + rsel) * LDE + lc]);
                float v[4] = { raw.x, raw.y, raw.z, raw.w }, s2[4] = { 0.f, 0.f, 0.f, 0.f };
                const float rv[4] = { resv[k].x, resv[k].y, resv[k].z, resv[k].w }, mv[4] = { mulv[k].x, mulv[k].y, mulv[k].z, mulv[k].w };
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e] * a.acc_scale + bs[e];
                    if (ACT == 1) x = gelu_f(x);
                    else if (ACT == 2) x = silu2_f(x);
                    x = rv[e] + rs[e] * (x * mv[e]);
                    v[e] = x;
                    if (OUT2) s2[e] = x + ib[e] * sin_sq(x * ea[e]);
                }
                if (okp) {
                    if (a.out) *reinterpret_cast<float4*>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    if (OUT2) *reinterpret_cast<float4*>(a.out2 + o) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { resv[k] = resn[k]; mulv[k] = muln[k]; }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    if (a.out2) { if (a.act == 0) rows(I0{}, std::true_type{}); else if (a.act == 1) rows(I1{}, std::true_type{}); else rows(I2{}, std::true_type{}); }
    else { if (a.act == 0) rows(I0{}, std::false_type{}); else if (a.act == 1) rows(I1{}, std::false_type{}); else rows(I2{}, std::false_type{}); }
}

// KC = C_in columns per staged chunk (32, or 128 for the short-and-wide GEMMs of the pre-transformer, whose few workgroups walk K
// serially: a chunk costs one memory latency whatever its size, so 4x wider chunks are 4x fewer latencies); NBUF = weight-tile buffers.
// PEEL: plain convs with >= 3 taps, next chunk's rows two iterations ahead (see the loop)
// APL: the input is (hi, lo) fp16 planes (ConvKArgs::in_planes): staged without conversion
// WLO: the weight has a non-zero lo plane.  false (weights that are exact in fp16 after the power-of-two pre-scale: every bf16-origin
// tensor): w = hi exactly, so x . w = x_hi . w_hi + x_lo . w_hi — two products instead of three, exact, and the lo plane is neither
// loaded nor staged
template <int MB, int NB, int WM, int WN, int PA, int KC = 32, int NBUF = 2, bool F2 = false, bool PEEL = false, bool APL = false, bool WLO = true>
// <= 96 accumulator registers and 32-column chunks: two workgroups per CU (256 registers each); otherwise ONE wave per SIMD with the
// whole 512-register file — said explicitly, or hipcc still budgets 256 and spills the prefetch registers right behind their loads
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MB * NB <= 6 && KC == 32 ? 2 : 1), (MB * NB <= 6 && KC == 32 ? 2 : 1))))
void k_conv_split(ConvKArgs a0) {
    ConvKArgs a = a0;
    int bx, by, bz;
    conv_tile_ids(a.xcd_map != 0, bx, by, bz);
    bx = conv_batch_rebase(a, bx);
    constexpr int TM = WM * MB * 32, TN = WN * NB * 32, AROWS = PA * 32;
    constexpr int CB = KC / 32, SEG = KC / 8, LD = KC + 8;       // 32-column blocks, 16-byte segments and padded halves per staged row
    constexpr int BPL = WLO ? 2 : 1, PBT = TN * SEG * BPL;      // weight planes staged; their 16-B segments per tile
    constexpr int PB = (PBT + 255) / 256;                        // segments per thread (a ragged last pass re-stages the tile's first segments: same bytes, same place)
    constexpr int A_BYTES = 2 * AROWS * LD * 2, B_BYTES = NBUF * 2 * TN * LD * 2;
    constexpr int E_BYTES = 4 * 32 * (NB * 32 + 8) * 4;          // epilogue staging: 32 rows per wave
    constexpr int SM_BYTES = A_BYTES + B_BYTES > E_BYTES ? A_BYTES + B_BYTES : E_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    _Float16 (*As)[AROWS][LD] = reinterpret_cast<_Float16 (*)[AROWS][LD]>(smem);
    _Float16 (*Bs)[2][TN][LD] = reinterpret_cast<_Float16 (*)[2][TN][LD]>(smem + A_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool ksp = KC == 128 && a.ksplit > 1;               // split-K (short pre-transformer GEMMs): blockIdx.z is the K slice, not a phase
    const int m0 = bx * TM, co0 = by * TN, phase = ksp ? 0 : bz;
    const int NT = a.transposed ? a.taps / a.stride : a.taps;
    const int halo = a.transposed ? NT - 1 : (a.taps - 1) * a.dil;
    const int n_chunks = ksp ? a.C_in / KC / a.ksplit : a.C_in / KC, total = n_chunks * NT;
    const int c_first = ksp ? bz * n_chunks : 0;

    f32x16 acc[MB][NB];
    const f32x16 zero16 = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = zero16;

    const int arow = tid >> 3, acol = (tid & 7) * 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v areg[PA][CB];   // native vector types: hipcc keeps arrays of the HIP uint4 / float4 structs beyond 128 bytes in scratch
    u32x4 breg[PB];
    // row addresses as a uniform base (SGPR pair, moves with the chunk) + a 32-bit byte offset per staged row (loop-invariant): half the
    // address registers of 64-bit pointers, which is what kept the peeled loop from fitting 256 registers (launch_conv checks < 4 GB)
    const char* const in_bytes = reinterpret_cast<const char*>(a.in);
    unsigned aoff[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int src = m0 - halo + arow + 32 * i;
        const int sc = src < 0 ? 0 : (src < a.T_in ? src : a.T_in - 1);          // clamped address; out-of-range rows are zeroed in storeA
        aoff[i] = ((unsigned)sc * (unsigned)a.C_in + (unsigned)acol) * 4u;
    }
    auto loadA = [&](int ci0) {
        const char* const base = in_bytes + (size_t)ci0 * 4;
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) areg[i][cb] = *reinterpret_cast<const f32x4v*>(base + cb * 128 + aoff[i]);   // APL: .xy = 4 hi halves, .zw = 4 lo halves
    };
    // weights likewise: uniform base (tap, chunk) + a 32-bit byte offset per staged segment (plane, row, segment); the lo plane sits
    // behind the hi plane in one allocation (launch_conv checks the distance)
    const char* const w_bytes = reinterpret_cast<const char*>(a.Wh);
    const unsigned w_lo = (unsigned)(reinterpret_cast<const char*>(a.Wl) - reinterpret_cast<const char*>(a.Wh));
    // 32-wide-chunk kernels read the CHUNK-major planes (launch_conv hands them ConvArgs::Whc): a weight row of a (tap, chunk) tile is 64
    // bytes and the tile's rows lie back to back; the 128-wide-chunk kernels read the row-major planes.  Compile-time: a run-time choice
    // cost the peeled 7-tap kernels 12-20 bytes of scratch.
    constexpr bool w_cm = KC == 32;
    const unsigned w_rowb = w_cm ? 64u : (unsigned)a.C_in * 2u;        // bytes between the tile's consecutive weight rows
    unsigned boff[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        int idx = tid + 256 * i;
        if (PBT % 256 != 0 && idx >= PBT) idx -= PBT;
        const int plane = idx / (TN * SEG), rem = idx % (TN * SEG), brow = rem / SEG, bseg = rem % SEG;
        const int co = co0 + brow;
        const int cc = co < a.C_out ? co : a.C_out - 1;                   // rows past C_out repeat the last one; their columns are never stored
        boff[i] = (plane ? w_lo : 0u) + (unsigned)cc * w_rowb + (unsigned)bseg * 16u;
    }
    auto loadB = [&](int ci0, int ti) {
        const int wtap = a.transposed ? phase + ti * a.stride : ti;
        const char* const base = w_bytes + (w_cm ? ((size_t)wtap * (a.C_in >> 5) + (ci0 >> 5)) * a.C_out * 64 : ((size_t)wtap * a.C_out * a.C_in + ci0) * 2);
#pragma unroll
        for (int i = 0; i < PB; ++i) breg[i] = *reinterpret_cast<const u32x4*>(base + boff[i]);
    };
    auto storeA = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int src = m0 - halo + arow + 32 * i;
            const bool inr = src >= 0 && src < a.T_in;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f32x4v r = areg[i][cb];
                const float4 v = make_float4(inr ? r.x : 0.f, inr ? r.y : 0.f, inr ? r.z : 0.f, inr ? r.w : 0.f);
                if (APL) {   // already (hi, lo) halves: zeroed rows are 0.0 in both planes
                    *reinterpret_cast<uint2*>(&As[0][arow + 32 * i][cb * 32 + acol]) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
                    *reinterpret_cast<uint2*>(&As[1][arow + 32 * i][cb * 32 + acol]) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
                    continue;
                }
                f16x2 h01, h23, l01, l23;
                split_f16x2(v.x, v.y, h01, l01);
                split_f16x2(v.z, v.w, h23, l23);
                *reinterpret_cast<uint2*>(&As[0][arow + 32 * i][cb * 32 + acol]) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2*>(&As[1][arow + 32 * i][cb * 32 + acol]) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
        }
    };
    auto storeB = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int idx = tid + 256 * i;
            if (PBT % 256 != 0 && idx >= PBT) idx -= PBT;
            const int plane = idx / (TN * SEG), rem = idx % (TN * SEG), brow = rem / SEG, bseg = rem % SEG;
            *reinterpret_cast<u32x4*>(&Bs[buf][plane][brow][bseg * 8]) = breg[i];
        }
    };

    // one (chunk, tap) step of the K walk on the staged tiles: KC / 16 k-steps of MB x NB x 3 MFMAs
    auto mfma_step = [&](int buf, int shift) {
        const int roff = halo - shift + wm * MB * 32 + (lane & 31);
#pragma unroll
        for (int st = 0; st < KC / 16; ++st) {
            const int kof = st * 16 + 8 * (lane >> 5);
            f16x8 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(&As[0][roff + i * 32][kof]);
                al[i] = *reinterpret_cast<const f16x8*>(&As[1][roff + i * 32][kof]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int brow = wn * NB * 32 + j * 32 + (lane & 31);
                bh[j] = *reinterpret_cast<const f16x8*>(&Bs[buf][0][brow][kof]);
                if constexpr (WLO) bl[j] = *reinterpret_cast<const f16x8*>(&Bs[buf][1][brow][kof]);
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    if constexpr (WLO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    CP_MARK(0);
    loadA(c_first * KC);
    loadB(c_first * KC, 0);
    if constexpr (PEEL) {
        static_assert(KC == 32 && NBUF == 2, "peeled taps: 32-wide chunks, double-buffered weights");   // launch_conv: !transposed, taps >= 3
        // Plain convs with >= 3 taps (the decoder's 7-tap convs).  The NEXT chunk's activation rows are requested during tap 1, behind
        // tap 2's weights in program order, and have until tap 3 to arrive: two iterations instead of one (the stamps showed 1.3-2.7 us
        // of exposed HBM latency at every chunk boundary).  s_waitcnt counts in order and hipcc assumes the fewest loads in flight
        // wherever control flow joins, so this only works if every wait sees the same loads on every path to it: taps 0, 1 and 2 are
        // peeled (straight-line), the remaining taps run in a loop whose entry and back edge both have nothing but the weights younger
        // than what is waited for, and nothing in a chunk's body is conditional — past the last chunk the loads are issued for the last
        // chunk again (L2 hits, never used) instead of being skipped.
        const int c_last = c_first + n_chunks - 1;
        int it = 0;
        // the fused unit's dilation-9 variant (PA = 10) keeps one tap per barrier: with the pairs' third weight register it spills
        constexpr bool PAIRS = !WLO && !(F2 && PA == 10);
        if constexpr (PAIRS) {   // launch_conv: the two-product kernels are peeled for 7-tap convs only
            // Two taps per block barrier (7-tap convs, weights exact in fp16): without a lo plane half of each weight buffer is idle, so
            // it takes a second TAP's tile: Bs[buf][0 | 1] = taps (2 p, 2 p + 1), and the seventh tap travels as the second tile of the
            // pair (5, 6) whose first tile is not used.  4 barriers and 4 weight hand-overs per chunk instead of 7, 48 MFMAs between
            // them instead of 24; same products in the same order (chunk-major, taps ascending): bit-identical sums.  Straight-line
            // per chunk like the tap form below, for the same reason (in-order s_waitcnt): the next chunk's rows go out in pair 1.
            constexpr int PBT2 = 2 * TN * SEG, PB2 = (PBT2 + 255) / 256;
            static_assert(PBT2 % 256 == 0, "paired taps: the two tiles' segments divide over the 256 threads");
            u32x4 breg2[PB2];
            unsigned boff2[PB2];
            const unsigned tap_bytes = (unsigned)a.C_out * (unsigned)a.C_in * 2u;
#pragma unroll
            for (int i = 0; i < PB2; ++i) {
                const int idx = tid + 256 * i;
                const int t2 = idx / (TN * SEG), rem = idx % (TN * SEG), brow = rem / SEG, bseg = rem % SEG;
                const int co = co0 + brow;
                const int cc = co < a.C_out ? co : a.C_out - 1;
                boff2[i] = (t2 ? tap_bytes : 0u) + (unsigned)cc * w_rowb + (unsigned)bseg * 16u;   // the next tap's tile is tap_bytes on in either layout
            }
            auto loadB2 = [&](int ci0, int tbase) __attribute__((always_inline)) {
                const char* const base = w_bytes + (w_cm ? ((size_t)tbase * (a.C_in >> 5) + (ci0 >> 5)) * a.C_out * 64 : ((size_t)tbase * a.C_out * a.C_in + ci0) * 2);
#pragma unroll
                for (int i = 0; i < PB2; ++i) breg2[i] = *reinterpret_cast<const u32x4*>(base + boff2[i]);
            };
            auto storeB2 = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < PB2; ++i) {
                    const int idx = tid + 256 * i;
                    const int t2 = idx / (TN * SEG), rem = idx % (TN * SEG), brow = rem / SEG, bseg = rem % SEG;
                    *reinterpret_cast<u32x4*>(&Bs[buf][t2][brow][bseg * 8]) = breg2[i];
                }
            };
            auto mfma_tap = [&](int buf, int t2, int shift) __attribute__((always_inline)) {
                const int roff = halo - shift + wm * MB * 32 + (lane & 31);
#pragma unroll
                for (int st = 0; st < KC / 16; ++st) {
                    const int kof = st * 16 + 8 * (lane >> 5);
                    f16x8 ah[MB], al[MB], bh[NB];
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        ah[i] = *reinterpret_cast<const f16x8*>(&As[0][roff + i * 32][kof]);
                        al[i] = *reinterpret_cast<const f16x8*>(&As[1][roff + i * 32][kof]);
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j) bh[j] = *reinterpret_cast<const f16x8*>(&Bs[buf][t2][wn * NB * 32 + j * 32 + (lane & 31)][kof]);
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            };
            // pair p of a chunk: tiles from tap tb(p) = 0, 2, 4, 5; pair 3 uses its second tile only (tap 6)
            auto pair = [&](int chunk, int nchunk, int p, bool rows_in, bool rows_out) __attribute__((always_inline)) {
                const int buf = it & 1;
                CP_MARK_L(1 + it * 4);
                if (rows_in) { __syncthreads(); storeA(); }
                storeB2(buf);
                CP_MARK_L(2 + it * 4);
                __syncthreads();
                CP_MARK_L(3 + it * 4);
                const bool more = p < 3;
                loadB2((more ? chunk : nchunk) * KC, more ? (p == 2 ? 5 : 2 * (p + 1)) : 0);
                __builtin_amdgcn_sched_barrier(0);
                if (rows_out) loadA(nchunk * KC);
                __builtin_amdgcn_sched_barrier(0);
                if (p < 3) { mfma_tap(buf, 0, (6 - 2 * p) * a.dil); mfma_tap(buf, 1, (5 - 2 * p) * a.dil); }
                else mfma_tap(buf, 1, 0);
                CP_MARK_L(4 + it * 4);
                ++it;
            };
            loadB2(c_first * KC, 0);      // the prologue's loadB fetched tap 0's tile only: both tiles of pair 0 again, behind the rows
            for (int chunk = c_first; chunk <= c_last; ++chunk) {
                const int nchunk = chunk < c_last ? chunk + 1 : c_last;
                pair(chunk, nchunk, 0, true, false);
                pair(chunk, nchunk, 1, false, true);
                pair(chunk, nchunk, 2, false, false);
                pair(chunk, nchunk, 3, false, false);
            }
        } else {
        auto tap = [&](int chunk, int nchunk, int ti, bool rows_in, bool rows_out) {
            const int buf = it & 1;
            CP_MARK_L(1 + it * 4);
            if (rows_in) { __syncthreads(); storeA(); }
            storeB(buf);
            CP_MARK_L(2 + it * 4);
            __syncthreads();
            CP_MARK_L(3 + it * 4);
            const bool more = ti + 1 < NT;
            loadB((more ? chunk : nchunk) * KC, more ? ti + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);             // the rows go out BEHIND the weights, whatever the scheduler prefers
            if (rows_out) loadA(nchunk * KC);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(buf, (a.taps - 1 - ti) * a.dil);
            CP_MARK_L(4 + it * 4);
            ++it;
        };
        for (int chunk = c_first; chunk <= c_last; ++chunk) {
            const int nchunk = chunk < c_last ? chunk + 1 : c_last;
            tap(chunk, nchunk, 0, true, false);
            tap(chunk, nchunk, 1, false, true);
            tap(chunk, nchunk, 2, false, false);
            for (int ti = 3; ti < NT; ++ti) tap(chunk, nchunk, ti, false, false);
        }
        }
    } else {
    int chunk = c_first, ti = 0;
    for (int it = 0; it < total; ++it) {
        const int buf = NBUF == 2 ? (it & 1) : 0;
        CP_MARK_L(1 + it * 4);
        if (ti == 0 || NBUF == 1) __syncthreads();            // every wave is done with the rows (and, single-buffered, the weight tile) it read last
        if (ti == 0) storeA();
        storeB(buf);                                          // double-buffered: this buffer was last read two taps ago
        CP_MARK_L(2 + it * 4);
        __syncthreads();
        CP_MARK_L(3 + it * 4);
        int nchunk = chunk, nti = ti + 1;
        if (nti == NT) { nti = 0; ++nchunk; }
        if (it + 1 < total) {                                 // next tap's weights (and next chunk's rows) fly during the MFMAs below
            loadB(nchunk * KC, nti);
            if (nti == 0) loadA(nchunk * KC);
        }
        mfma_step(buf, a.transposed ? ti : (a.taps - 1 - ti) * a.dil);
        chunk = nchunk; ti = nti;
        CP_MARK_L(4 + it * 4);
    }
    }
    CP_MARK(62);
    __syncthreads();                                          // the staging slices below overlay the operand tiles
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * (NB * 32 + 8);
    if constexpr (F2) {
        // ---- fused residual unit: t = SnakeBeta(conv1 + bias) never leaves the CU.  A 1x1 conv needs, for a wave's 32 time rows, those
        // same 32 rows of t and nothing else: each wave turns its accumulator block into (hi, lo) fp16 planes in its OWN LDS slice (the
        // bytes its epilogue staging uses afterwards), multiplies them with the 96 x 96 weight planes (B fragments straight from
        // global: 36 KB, L1 / L2 resident) and runs the ordinary epilogue (bias2, residual, SnakeBeta for the next layer) on the result.
        // No block barrier past this point.  TN == C_out == the whole channel range (grid.y == 1).
        static_assert(WN == 1 && KC == 32, "fused residual unit: one column tile, 32-wide chunks");
        constexpr int C2 = NB * 32, TLD = C2 + 8;
        _Float16* ts = reinterpret_cast<_Float16*>(stage);                       // [2][32][TLD] halves == 32 x (C2 + 8) floats: the wave's slice
        const int col0 = lane & 31, rsel = lane >> 5;
        float b1[NB], ea1[NB], ib1[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = j * 32 + col0;
            b1[j] = a.bias ? a.bias[co] : 0.f;
            ea1[j] = a.s1_pre[co];
            ib1[j] = a.s1_pre[C2 + co];
        }
#ifdef Q3_FORCE_SPILL   // tools/build_spill_lib.sh only: the scale takes a round trip through a private (scratch) segment, so that the fused
        volatile float q3_sp[4];   // units run WITH scratch on the concurrent vocoder lanes (round 4's suspect; tests/test_gpu_codec_stress.py under Q3TTS_LIB)
        q3_sp[lane & 3] = a.acc_scale;
        const float sc1 = q3_sp[lane & 3];
#else
        const float sc1 = a.acc_scale;
#endif
        const bool w2f = a.W2fh != nullptr;
        const _Float16* const w2hp = reinterpret_cast<const _Float16*>(w2f ? a.W2fh : a.W2h);
        const _Float16* const w2lp = reinterpret_cast<const _Float16*>(w2f ? a.W2fl : a.W2l);
        a.bias = a.bias2; a.acc_scale = a.acc_scale2;          // from here on `a` describes the second conv's epilogue (no struct copy: it would live in scratch)
        // statically indexed row blocks (a rolled loop over i gives the accumulators a scratch home that the main loop keeps in sync)
        auto unit = [&](auto itag) {
            constexpr int i = decltype(itag)::value;
            CP_MARK(48 + 4 * i);
            // accumulator layout: column = lane & 31 (+ 32 j), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int reg = 0; reg < 16; reg += 2) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * rsel;      // reg, reg + 1: rows row, row + 1
                    const float v0 = acc[i][j][reg] * sc1 + b1[j], v1 = acc[i][j][reg + 1] * sc1 + b1[j];
                    const float t0 = v0 + ib1[j] * sin_sq(v0 * ea1[j]), t1 = v1 + ib1[j] * sin_sq(v1 * ea1[j]);
                    f16x2 h, l;
                    split_f16x2(t0, t1, h, l);
                    ts[row * TLD + j * 32 + col0] = h.x;
                    ts[(row + 1) * TLD + j * 32 + col0] = h.y;
                    ts[(32 + row) * TLD + j * 32 + col0] = l.x;
                    ts[(33 + row) * TLD + j * 32 + col0] = l.y;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CP_MARK(49 + 4 * i);
            f32x16 acc2[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc2[j] = zero16;
            // weight fragments one k-step ahead, pinned: left alone hipcc hoists all 36 loads (144 registers) above the loop and spills
            f16x8 wbh[2][NB], wbl[2][NB];
            auto loadW = [&](int buf, int st) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // fragment order (a.W2fh: one contiguous KB per load) or weight row = output channel, [C_out][C_in] with C_in == C2
                    const size_t wo = w2f ? ((size_t)(st * NB + j) * 64 + lane) * 8 : (size_t)(j * 32 + col0) * C2 + st * 16 + 8 * rsel;
                    wbh[buf][j] = *reinterpret_cast<const f16x8*>(w2hp + wo);
                    if constexpr (WLO) wbl[buf][j] = *reinterpret_cast<const f16x8*>(w2lp + wo);
                }
            };
            loadW(0, 0);
#pragma unroll
            for (int st = 0; st < C2 / 16; ++st) {
                if (st + 1 < C2 / 16) loadW((st + 1) & 1, st + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int kof = st * 16 + 8 * rsel;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ts[col0 * TLD + kof]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&ts[(32 + col0) * TLD + kof]);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wbh[st & 1][j], acc2[j], 0, 0, 0);
                    if constexpr (WLO) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wbl[st & 1][j], acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wbh[st & 1][j], acc2[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                     // the plane reads are done: the same bytes become the epilogue's staging
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CP_MARK(50 + 4 * i);
            if (a.out2_planes) split_epilogue_snake<NB, true, true, 8, true>(a, acc2, stage, m0 + wm * MB * 32 + i * 32, co0, lane, 0, NT);
            else split_epilogue_snake<NB, true, true, 8>(a, acc2, stage, m0 + wm * MB * 32 + i * 32, co0, lane, 0, NT);   // launch_conv guarantees out, out2, res, no activation
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CP_MARK(51 + 4 * i);
        };
        unit(std::integral_constant<int, 0>{});
        if constexpr (MB > 1) unit(std::integral_constant<int, MB - 1>{});
        static_assert(MB <= 2, "fused residual unit: at most two row blocks per wave");
        CP_MARK(63);
        return;
    }
    // statically indexed row blocks (a rolled loop would give the accumulators a scratch home that the main loop keeps in sync)
    if constexpr (KC == 128) {
        if (ksp) {   // raw partial sums of this K slice; bias / activation / residual belong to k_conv_finish
            ConvKArgs b = a;
            b.out = a.slab + (size_t)bz * a.T_out * a.C_out;
            b.bias = nullptr; b.res = nullptr; b.res_scale = nullptr; b.mul = nullptr; b.out2 = nullptr; b.act = 0; b.acc_scale = 1.0f;
            split_epilogue_block<NB>(b, acc[0], stage, m0 + wm * MB * 32, co0 + wn * NB * 32, lane, 0, NT);
            if (MB > 1) split_epilogue_block<NB>(b, acc[MB - 1], stage, m0 + wm * MB * 32 + 32, co0 + wn * NB * 32, lane, 0, NT);
            return;
        }
    }
    CP_MARK(60);
    split_epilogue_block<NB>(a, acc[0], stage, m0 + wm * MB * 32, co0 + wn * NB * 32, lane, phase, NT);
    CP_MARK(61);
    if (MB > 1) split_epilogue_block<NB>(a, acc[MB - 1], stage, m0 + wm * MB * 32 + 32, co0 + wn * NB * 32, lane, phase, NT);
    CP_MARK(63);
}

// Split-K tail: out = epilogue(sum of the K slices' partial sums, slice order).  One thread owns 4 consecutive channels of one row;
// the arithmetic is split_epilogue_block's: x = acc * scale + bias, activation, x = res + res_scale * (x * mul), SnakeBeta second output.
__global__ __launch_bounds__(256) void k_conv_finish(ConvKArgs a) {
    const size_t n4 = (size_t)a.T_out * a.C_out / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t o = i * 4, plane = (size_t)a.T_out * a.C_out;
    const int co = (int)(o % a.C_out);
    float4 s4 = *reinterpret_cast<const float4*>(a.slab + o);
    for (int z = 1; z < a.ksplit; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(a.slab + z * plane + o);
        s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w;
    }
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : zero4;
    const float4 r4 = a.res_scale ? *reinterpret_cast<const float4*>(a.res_scale + co) : one4;
    const float4 m4 = a.mul ? *reinterpret_cast<const float4*>(a.mul + o) : one4;
    const float4 e4 = a.res ? *reinterpret_cast<const float4*>(a.res + o) : zero4;
    const float v[4] = { s4.x, s4.y, s4.z, s4.w }, bs[4] = { b4.x, b4.y, b4.z, b4.w }, rs[4] = { r4.x, r4.y, r4.z, r4.w };
    const float mv[4] = { m4.x, m4.y, m4.z, m4.w }, rv[4] = { e4.x, e4.y, e4.z, e4.w };
    float x[4], s2[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = v[e] * a.acc_scale + bs[e];
        if (a.act == 1) t = gelu_f(t);
        else if (a.act == 2) t = silu2_f(t);
        x[e] = rv[e] + rs[e] * (t * mv[e]);
    }
    if (a.out2) {
        const float4 al4 = *reinterpret_cast<const float4*>(a.s2_alpha + co), be4 = *reinterpret_cast<const float4*>(a.s2_beta + co);
        const float al[4] = { al4.x, al4.y, al4.z, al4.w }, be[4] = { be4.x, be4.y, be4.z, be4.w };
#pragma unroll
        for (int e = 0; e < 4; ++e) s2[e] = x[e] + (1.0f / (expf(be[e]) + 0.000000001f)) * sin_sq(x[e] * expf(al[e]));
        if (a.out2_planes) {
            f16x2 h01, h23, l01, l23;
            split_f16x2(s2[0], s2[1], h01, l01);
            split_f16x2(s2[2], s2[3], h23, l23);
            *reinterpret_cast<uint4*>(a.out2 + o) =
                make_uint4(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        } else
        *reinterpret_cast<float4*>(a.out2 + o) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    if (a.out) *reinterpret_cast<float4*>(a.out + o) = make_float4(x[0], x[1], x[2], x[3]);
}

// fp32 weights * scale -> (hi, lo) fp16 planes; absmax for choosing the power-of-two scale
__global__ void k_split_planes(const float* w, bf16_t* hi, bf16_t* lo, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = w[i] * scale;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        hi[i] = __builtin_bit_cast(bf16_t, h);
        lo[i] = __builtin_bit_cast(bf16_t, l);
    }
}
__global__ void k_absmax(const float* w, size_t n, unsigned* out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bit patterns
}
// SnakeBeta constants of one activation, with the expressions the epilogues used to evaluate per block
__global__ void k_snake_pre(const float* alpha, const float* beta, float* pre, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { pre[c] = expf(alpha[c]); pre[C + c] = 1.0f / (expf(beta[c]) + 0.000000001f); }
}
void launch_snake_pre(const float* alpha, const float* beta, float* pre, int C, hipStream_t s) {
    hipLaunchKernelGGL(k_snake_pre, dim3((C + 255) / 256), dim3(256), 0, s, alpha, beta, pre, C);
}

void launch_split_planes(const float* w, bf16_t* hi, bf16_t* lo, size_t n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, w, hi, lo, n, scale);
    Q3_HIP_CHECK(hipGetLastError());
}
// OR of the magnitude bits of a 16-bit plane: 0 <=> every element is +-0 (is a weight's lo plane empty?)
__global__ void k_or_mag16(const bf16_t* p, size_t n, unsigned* out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m |= (unsigned)(p[i] & 0x7FFFu);
    if (m) atomicOr(out, m);
}
void launch_or_mag16(const bf16_t* p, size_t n, unsigned* out, hipStream_t s) {
    hipLaunchKernelGGL(k_or_mag16, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, p, n, out);
    Q3_HIP_CHECK(hipGetLastError());
}
void launch_absmax(const float* w, size_t n, unsigned* out, hipStream_t s) {
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, w, n, out);
    Q3_HIP_CHECK(hipGetLastError());
}

// every k_conv_split launch: the WLO = false twin when the launch's weight planes `lo` are all zero (ConvKArgs::wlo == 0)
#define Q3_CS(GRID, S, A, ...) do { if ((A).wlo) hipLaunchKernelGGL((k_conv_split<__VA_ARGS__, true>), GRID, dim3(256), 0, S, A); \
                                     else hipLaunchKernelGGL((k_conv_split<__VA_ARGS__, false>), GRID, dim3(256), 0, S, A); } while (0)
template <int MB, int NB, int WM, int WN, bool APL = false>
static void launch_split_pa(const ConvKArgs& a, dim3 grid, int extra, hipStream_t s) {
    constexpr int P0 = WM * MB;
    if constexpr (MB == 2 && NB == 3) {   // the decoder's 7-tap convs on 256 x 96 tiles
        if (a.peel_taps && !a.transposed && (a.wlo ? a.taps >= 3 : a.taps == 7) && extra >= 1) {   // two-product kernels: the peeled loop is built for 7 taps (tap pairs)
            if (extra == 1) Q3_CS(grid, s, a, MB, NB, WM, WN, P0 + 1, 32, 2, false, true, APL);
            else Q3_CS(grid, s, a, MB, NB, WM, WN, P0 + 2, 32, 2, false, true, APL);
            return;
        }
    }
    if (extra == 0) Q3_CS(grid, s, a, MB, NB, WM, WN, P0, 32, 2, false, false, APL);
    else if (extra == 1) Q3_CS(grid, s, a, MB, NB, WM, WN, P0 + 1, 32, 2, false, false, APL);
    else Q3_CS(grid, s, a, MB, NB, WM, WN, P0 + 2, 32, 2, false, false, APL);
}
// (hi, lo)-plane inputs exist for the tile shapes the decoder's convs use (96-multiples); anything else reads fp32
template <int MB, int NB, int WM, int WN>
static void launch_split_in(const ConvKArgs& a, dim3 grid, int extra, hipStream_t s) {
    if (a.in_planes) launch_split_pa<MB, NB, WM, WN, true>(a, grid, extra, s);
    else launch_split_pa<MB, NB, WM, WN, false>(a, grid, extra, s);
}

// C_out == 1 (the decoder's last conv): one output sample per thread, the input rows of a 128-sample tile (+ tap halo) staged in
// LDS with an odd row stride; a matrix-core tile would be 1/64 full.  Causal taps, optional clamp.
#define CO1_T 128
__global__ __launch_bounds__(CO1_T) void k_conv_cout1(ConvKArgs a0) {
    extern __shared__ float xs[];                    // [(CO1_T + halo)][C_in + 1]
    ConvKArgs a = a0;
    const int bx = conv_batch_rebase(a, blockIdx.x);
    const int ld = a.C_in + 1, halo = (a.taps - 1) * a.dil, t0 = bx * CO1_T, rows = CO1_T + halo;
    for (int i = threadIdx.x; i < rows * (a.C_in / 4); i += CO1_T) {
        const int r = i / (a.C_in / 4), c4 = (i % (a.C_in / 4)) * 4, src = t0 - halo + r;
        const int sc = src < 0 ? 0 : (src < a.T_in ? src : a.T_in - 1);
        const float4 v = *reinterpret_cast<const float4*>(a.in + (size_t)sc * a.C_in + c4);
        const bool inr = src >= 0 && src < a.T_in;
        xs[r * ld + c4] = inr ? v.x : 0.f; xs[r * ld + c4 + 1] = inr ? v.y : 0.f; xs[r * ld + c4 + 2] = inr ? v.z : 0.f; xs[r * ld + c4 + 3] = inr ? v.w : 0.f;
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    float acc = 0.f;
    for (int tap = 0; tap < a.taps; ++tap) {
        const float* xr = xs + (threadIdx.x + tap * a.dil) * ld;   // row t - (taps-1-tap)*dil
        const float* w = a.W + (size_t)tap * a.C_in;                // [tap][0][ci]: wave-uniform, scalar loads
        for (int ci = 0; ci < a.C_in; ++ci) acc = fmaf(w[ci], xr[ci], acc);
    }
    if (t < a.T_out) {
        float v = acc + (a.bias ? a.bias[0] : 0.f);
        if (a.clamp) v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        a.out[t] = v;
    }
}

// C_out == 1 with C_in a multiple of 32 (the decoder's last conv: 96 channels, 7 taps) — round 4.  k_conv_cout1 above reads every staged
// element from LDS once per tap (672 four-byte LDS reads per output sample: the launch ran at 1.6 TB/s of HBM traffic).  Here nothing is
// staged: eight lanes share an input row, lane s owning channels [s CPT, (s + 1) CPT) of it — straight from global memory, 16 bytes per
// load, a row's eight lanes covering its C_in floats contiguously — with the `taps` x CPT weights of its slice resident in registers.  A
// workgroup walks 256 rows in 8 passes of 32 (every load of the tile issued up front), each lane forms its slice of the row's per-tap
// dot products d[tap] = sum_c W[tap][c] x[row][c], the eight slices meet by DPP (quad sums, then the half-row mirror), and the partial sums
// change hands through LDS: output j adds d[tap] of rows j + tap * dil.  Causal taps, optional clamp.  A tile yields 256 - halo outputs.
//
// Round 5: every sum here is a SCALAR fp32 instruction (opaque() below keeps the compiler from pairing two taps into v_pk_fma_f32 /
// v_pk_add_f32).  With the packed form — what the compiler emits by itself, and what rounds 4-5 shipped until this fix — the kernel
// returned a wrong partial sum now and then: the low half of a packed pair (an even tap), in lanes 48..63 of a wave (the rows 8w + 6 and
// 8w + 7 of a pass, 9 times in 10 of the first pass, i.e. at the onset of the FMA burst behind the kernel's one memory wait), the input
// rows bit-identical to a clean run's — two adjacent PCM samples off by up to 1.7e-2 (round 4's "one-off" 3.8e-3).  Which launch of a
// job it hits depends on the job's composition (0 % to 90 % of the jobs: profiles/r05_hunt/README.txt); a decode on its own never
// fails.  Of twelve variants of this kernel the six with v_pk_fma_f32 fail and the six without never do (0 of 1000-3000 jobs each; LDS
// layout, DPP vs ds_bpermute, wait states before the DPP reads, launching it twice do not matter): tools/vocoder_stress.py, DESIGN.md
// section 8.  PACKED = the old code, kept as the reproducer behind the A/B knob Q3TTS_COUT1_PACKED (=2: also dumps each tile's LDS
// partial sums, DUMP).
#define CO1R_ROWS 256
static __device__ __forceinline__ float opaque(float v) { asm volatile("" : "+v"(v)); return v; }
template <int CPT, bool PACKED = false, bool DUMP = false>
__global__ __launch_bounds__(256) void k_conv_cout1_reg(ConvKArgs a0) {
    constexpr int MAXT = 8, NP = CO1R_ROWS / 32;
    __shared__ float ds[CO1R_ROWS][MAXT + 1];
    ConvKArgs a = a0;
    if constexpr (DUMP) a.out2 = nullptr;   // a0.out2 carries the dump buffer: not a second output
    const int bx = conv_batch_rebase(a, blockIdx.x);
    const int tid = threadIdx.x, s8 = tid & 7, rl = tid >> 3;
    const int halo = (a.taps - 1) * a.dil, TO = CO1R_ROWS - halo, t0 = bx * TO;
    // every row of the tile first (clamped addresses; rows outside the sequence are zeroed below)
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
        const float* wr = a.W + (size_t)(tap < a.taps ? tap : 0) * a.C_in + s8 * CPT;   // [tap][0][ci]
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wr + c);
            const bool on = tap < a.taps;
            w[tap][c] = on ? v.x : 0.f; w[tap][c + 1] = on ? v.y : 0.f; w[tap][c + 2] = on ? v.z : 0.f; w[tap][c + 3] = on ? v.w : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (DUMP) {   // shader clock over the FMA passes: s_memtime ticks per s_memrealtime tick (100 MHz), both taken behind the loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime();
    }
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
            if constexpr (!PACKED) acc = opaque(acc);   // one tap's chain is one value to the vectoriser: v_fmac_f32, never v_pk_fma_f32
            // the row's eight channel slices: quad sums, then lanes 0..3 take lanes 4..7 (lane s8 == 0 ends up with the row's sum)
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            if constexpr (!PACKED) acc = opaque(acc);
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]: every lane holds its quad's sum
            if constexpr (!PACKED) acc = opaque(acc);
            acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x141, 0xF, 0xF, true));   // row_half_mirror: lane i <-> lane 7 - i of its 8 lanes = the other quad
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
            if (tap < a.taps) acc += ds[tid + tap * a.dil][tap];   // row t - (taps - 1 - tap) * dil
        float v = acc + (a.bias ? a.bias[0] : 0.f);
        if (a.clamp) v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        a.out[t] = v;
    }
    if constexpr (DUMP) {   // the tile's partial sums as its output phase read them -> a0.out2[blockIdx.x][row][tap]
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        float* dbg = a0.out2 + (size_t)blockIdx.x * CO1R_ROWS * MAXT;
        for (int i = tid; i < CO1R_ROWS * MAXT; i += 256) dbg[i] = ds[i / MAXT][i % MAXT];
        __syncthreads();
        // tap 7 does not exist (its sums are 0): rows 0..2 of that column carry wave 0's shader ticks, 100 MHz ticks and the start time
        if (tid == 0) { dbg[0 * MAXT + 7] = (float)(clk1 - clk0); dbg[1 * MAXT + 7] = (float)(rt1 - rt0); dbg[2 * MAXT + 7] = (float)(rt0 & 0xFFFFFF); }
    }
}

// experiment of the packed-FMA hunt (Q3TTS_COUT1_DELAY_US, with Q3TTS_COUT1_PACKED): one wave that idles for `us` microseconds in front of the conv
__global__ void k_idle_us(int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
}

// dump buffer of the DUMP variant (Q3TTS_COUT1_PACKED=2 on a test-hook engine): one allocation, owned by the last batched launch
static float* g_cout1_dbg = nullptr;
static size_t g_cout1_dbg_floats = 0, g_cout1_dbg_used = 0;
void cout1_debug_buffer(const float** p, size_t* n) { *p = g_cout1_dbg; *n = g_cout1_dbg_used; }

void launch_conv(const ConvArgs& c, hipStream_t s) {
    if (c.C_in % 4 != 0) throw Error("conv: C_in must be a multiple of 4");
    ConvKArgs a;
    a.in = c.in; a.T_in = c.T_in; a.C_in = c.C_in; a.out = c.out; a.T_out = c.T_out; a.C_out = c.C_out;
    a.W = c.W; a.bias = c.bias; a.taps = c.taps; a.dil = c.dil; a.transposed = c.transposed; a.stride = c.stride; a.left = c.left;
    a.res = c.res; a.res_scale = c.res_scale; a.mul = c.mul; a.act = c.act; a.clamp = c.clamp;
    a.out2 = c.out2; a.s2_alpha = c.snake_alpha; a.s2_beta = c.snake_beta; a.s2_pre = c.snake_pre; a.s1_pre = nullptr;
    a.Wh = c.Wh; a.Wl = c.Wl;
    const bool force_wlo = knob("Q3TTS_CONV_3PRODUCT") != nullptr;   // A/B knob: always the three-product kernels
    a.wlo = (c.w_lo_zero && (c.W2h == nullptr || c.w2_lo_zero) && !force_wlo) ? 0 : 1;
    a.acc_scale = 1.0f;
    a.ksplit = 0; a.slab = nullptr; a.batch_tiles = 0;
    const bool no_xcd_map = knob("Q3TTS_CONV_NO_XCD_MAP") != nullptr;   // A/B switch
    a.xcd_map = no_xcd_map ? 0 : 1;
    const bool no_fast_epi = knob("Q3TTS_CONV_GENERIC_EPILOGUE") != nullptr;
    a.no_fast_epi = no_fast_epi ? 1 : 0;
    const bool no_peel = knob("Q3TTS_CONV_NO_PEEL") != nullptr;
    a.peel_taps = no_peel ? 0 : 1;
    a.in_planes = c.in_planes ? 1 : 0; a.out2_planes = c.out2_planes ? 1 : 0;
    if (c.in_planes || c.out2_planes) {   // only between convs of the split-precision path whose tiles have the plane variants
        const bool split_ok = c.Wh && c.Wl && c.C_in % 32 == 0 && c.C_out >= 32 && c.C_out % 96 == 0 && !c.clamp;
        if (!split_ok || (c.out2_planes && !(c.out2 && c.snake_pre && c.bias && c.act == 0 && !c.mul && !c.res_scale && !no_fast_epi)))
            throw Error("conv: (hi, lo)-plane activations need a 96-multiple decoder conv on the split-precision path");
    }
    a.W2h = nullptr; a.W2l = nullptr; a.acc_scale2 = 1.0f; a.bias2 = nullptr; a.s1_alpha = nullptr; a.s1_beta = nullptr;
    a.W2fh = nullptr; a.W2fl = nullptr;
    // the 32-wide-chunk kernels read the chunk-major planes (k_conv_split: w_cm), the 128-wide-chunk kernels the row-major ones
    auto use_cm = [&]() {
        if (c.Whc == nullptr || c.Wh == nullptr || c.Wl == nullptr) throw Error("conv: the 32-wide-chunk kernels need the chunk-major weight planes (ConvArgs::Whc)");
        a.Wl = c.Whc + (c.Wl - c.Wh); a.Wh = c.Whc;
    };
    a.in_ustride = c.in_ustride ? c.in_ustride : (size_t)c.T_in * c.C_in;
    const int nb = c.batch > 1 ? c.batch : 1;
    if (c.transposed && c.taps % c.stride != 0) throw Error("conv: transposed kernel must be a multiple of the stride");
    const int rows = c.transposed ? c.T_in + c.taps / c.stride - 1 : c.T_out;
    dim3 grid((rows + CT_M - 1) / CT_M, (c.C_out + CT_N - 1) / CT_N, c.transposed ? c.stride : 1);
    if (rows <= 0) return;
    if (c.C_out == 1 && !c.transposed && c.C_in == 96 && c.taps <= 8 && (c.taps - 1) * c.dil <= 64 && c.out && !c.out2 && !c.res && !c.mul &&
        !c.res_scale && c.act == 0 && !knob("Q3TTS_COUT1_LDS")) {   // Q3TTS_COUT1_LDS: the LDS-staged kernel (A/B knob)
        const int to1 = CO1R_ROWS - (c.taps - 1) * c.dil, tiles = (c.T_out + to1 - 1) / to1;
        if (nb > 1) a.batch_tiles = tiles;
        const dim3 g1((unsigned)(tiles * nb));
        if (const char* v = knob("Q3TTS_COUT1_PACKED")) {   // A/B knob: the packed-fp32 code of rounds 4-5 (the reproducer of the two-sample mismatch)
            if (const char* d = knob("Q3TTS_COUT1_DELAY_US")) hipLaunchKernelGGL(k_idle_us, dim3(1), dim3(64), 0, s, atoi(d));
            if (atoi(v) == 2 && nb > 1) {
                const size_t need = (size_t)tiles * nb * CO1R_ROWS * 8;
                if (need > g_cout1_dbg_floats) { Q3_HIP_CHECK(hipDeviceSynchronize()); if (g_cout1_dbg) (void)hipFree(g_cout1_dbg); g_cout1_dbg = nullptr; g_cout1_dbg_floats = 0; Q3_HIP_CHECK(hipMalloc((void**)&g_cout1_dbg, need * sizeof(float))); g_cout1_dbg_floats = need; }
                g_cout1_dbg_used = need;
                a.out2 = g_cout1_dbg;
                hipLaunchKernelGGL((k_conv_cout1_reg<12, true, true>), g1, dim3(256), 0, s, a);
            } else hipLaunchKernelGGL((k_conv_cout1_reg<12, true, false>), g1, dim3(256), 0, s, a);
            Q3_HIP_CHECK(hipGetLastError());
            return;
        }
        hipLaunchKernelGGL(k_conv_cout1_reg<12>, g1, dim3(256), 0, s, a);   // 96 channels (the shipped decoders' last block); other widths take the LDS kernel below
        Q3_HIP_CHECK(hipGetLastError());
        return;
    }
    if (c.C_out == 1 && !c.transposed && c.C_in % 4 == 0 && c.out && !c.out2 && !c.res && !c.mul && !c.res_scale && c.act == 0) {
        const size_t lds = (size_t)(CO1_T + (c.taps - 1) * c.dil) * (c.C_in + 1) * sizeof(float);
        if (lds <= 60 * 1024) {
            const int tiles = (c.T_out + CO1_T - 1) / CO1_T;
            if (nb > 1) a.batch_tiles = tiles;
            hipLaunchKernelGGL(k_conv_cout1, dim3(tiles * nb), dim3(CO1_T), lds, s, a);
            Q3_HIP_CHECK(hipGetLastError());
            return;
        }
    }
    const int NTt = c.transposed ? c.taps / c.stride : c.taps;
    const int halo = c.transposed ? NTt - 1 : (c.taps - 1) * c.dil;
    if (c.Wh && c.Wl && (size_t)c.T_in * c.C_in * sizeof(float) >= ((size_t)1 << 32))
        throw Error("conv: one sequence's activation exceeds 4 GB (beyond ~5500 frames): decode it in chunks (q3tts_codec_decode_chunked_host)");
    if (c.Wh && c.Wl && !(c.Wl > c.Wh && (size_t)((const char*)c.Wl - (const char*)c.Wh) < ((size_t)1 << 31)))
        throw Error("conv: the lo weight plane must follow the hi plane within 2 GB (one allocation)");
    if (c.W2h != nullptr) {   // fused residual unit: 7-tap conv -> SnakeBeta -> 1x1 conv -> + residual, 96 channels, 256-row tiles
        if (!(c.Wh && c.Wl && c.W2l && c.C_in == 96 && c.C_out == 96 && !c.transposed && halo <= 64 && c.mid_pre && c.snake_pre && c.bias2 && c.res && c.out && c.out2 && !c.res_scale && !c.clamp && c.act == 0 && !c.mul))
            throw Error("conv: fused residual unit needs 96 -> 96 channels on the split-precision path");
        a.acc_scale = c.w_scale_inv;
        a.W2h = c.W2h; a.W2l = c.W2l; a.acc_scale2 = c.w2_scale_inv; a.bias2 = c.bias2; a.s1_alpha = c.mid_alpha; a.s1_beta = c.mid_beta; a.s1_pre = c.mid_pre;
        if (c.W2fh && c.W2fl && !knob("Q3TTS_CONV_W2_ROWMAJOR")) { a.W2fh = c.W2fh; a.W2fl = c.W2fl; }   // A/B knob: the second conv's fragments from the row-major planes
        use_cm();
        const int extra = (halo + 31) / 32, tiles = (rows + 255) / 256;
        if (nb > 1) a.batch_tiles = tiles;
        const dim3 g((unsigned)(tiles * nb), 1, 1);
#define Q3_FUSED(PA_, PEEL_, APL_) Q3_CS(g, s, a, 2, 3, 4, 1, PA_, 32, 2, true, PEEL_, APL_)
        const bool peel = a.peel_taps && (a.wlo ? c.taps >= 3 : c.taps == 7), wide = extra > 1;
        if (a.in_planes) { if (peel) { if (wide) Q3_FUSED(10, true, true); else Q3_FUSED(9, true, true); } else { if (wide) Q3_FUSED(10, false, true); else Q3_FUSED(9, false, true); } }
        else { if (peel) { if (wide) Q3_FUSED(10, true, false); else Q3_FUSED(9, true, false); } else { if (wide) Q3_FUSED(10, false, false); else Q3_FUSED(9, false, false); } }
#undef Q3_FUSED
        Q3_HIP_CHECK(hipGetLastError());
        return;
    }
    if (c.Wh && c.Wl && c.C_in % 32 == 0 && c.C_out >= 32 && c.C_out % 4 == 0 && halo <= 64 && !c.clamp) {   // fp16 hi/lo split path
        a.acc_scale = c.w_scale_inv;
        const int extra = (halo + 31) / 32, z = c.transposed ? c.stride : 1;
        const bool n96 = c.C_out % 96 == 0;   // every decoder width (1536 .. 96) and the FFN; 96-wide tiles fit two workgroups per CU
        const int ntile = n96 ? c.C_out / 96 : (c.C_out + 127) / 128;
        const long n_big = (long)((rows + 255) / 256) * ntile * z * nb, n_thin = (long)((rows + 127) / 128) * ntile * z * nb;
        auto bgrid = [&](int tiles, int gy, int gz) { if (nb > 1) a.batch_tiles = tiles; return dim3((unsigned)(tiles * nb), (unsigned)gy, (unsigned)gz); };
        // 256-row tiles when the grid still fills the chip a few times over and K is deep enough to be compute-bound; 128-row tiles
        // (2-3 workgroups per CU) for bandwidth-bound or mid-sized launches; 64-row tiles for the short pre-transformer GEMMs
        const bool deep = NTt * c.C_in > 512;
        if (n_thin < 256) {   // fewer 128-row tiles than CUs: 64-row tiles, 128-column chunks when the channel count allows (swept: 64 .. 512)
            const dim3 g = bgrid((rows + 63) / 64, (c.C_out + 127) / 128, z);
            // a short utterance's pre-transformer GEMM (<= 128 frames x 1024..3072 channels) has 8-48 of these workgroups, each streaming
            // 0.5-1.5 MB of weights alone (47-75 us): cut K into 128-wide slices across workgroups, sum the slices in a tail kernel
            int ks = 1;
            if (nb == 1 && c.taps == 1 && !c.transposed && c.C_in % 128 == 0 && c.C_in >= 512 && c.slab) {   // fewest slices that give >= 256 workgroups
                const int nch = c.C_in / 128;
                for (ks = 1; ks < nch; ++ks)
                    if (nch % ks == 0 && (long)g.x * g.y * ks >= 256) break;
                while (ks > 1 && (nch % ks != 0 || (size_t)ks * c.T_out * c.C_out > c.slab_floats)) --ks;
            }
            if (ks > 1) {
                a.ksplit = ks; a.slab = c.slab;
                if (a.in_planes) Q3_CS(dim3(g.x, g.y, ks), s, a, 2, 1, 1, 4, 2, 128, 1, false, false, true);
                else Q3_CS(dim3(g.x, g.y, ks), s, a, 2, 1, 1, 4, 2, 128, 1, false, false, false);
                const size_t n4 = (size_t)c.T_out * c.C_out / 4;
                hipLaunchKernelGGL(k_conv_finish, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a);
                Q3_HIP_CHECK(hipGetLastError());
                return;
            }
            if (c.C_in % 128 == 0) {
                if (a.in_planes) {
                    if (extra == 0) Q3_CS(g, s, a, 2, 1, 1, 4, 2, 128, 1, false, false, true);
                    else if (extra == 1) Q3_CS(g, s, a, 2, 1, 1, 4, 3, 128, 1, false, false, true);
                    else Q3_CS(g, s, a, 2, 1, 1, 4, 4, 128, 1, false, false, true);
                }
                else if (extra == 0) Q3_CS(g, s, a, 2, 1, 1, 4, 2, 128, 1, false, false, false);
                else if (extra == 1) Q3_CS(g, s, a, 2, 1, 1, 4, 3, 128, 1, false, false, false);
                else Q3_CS(g, s, a, 2, 1, 1, 4, 4, 128, 1, false, false, false);
            } else { use_cm(); launch_split_in<2, 1, 1, 4>(a, g, extra, s); }
        }
        else if (!deep || n_big < 1024) {
            use_cm();
            const dim3 g2 = bgrid((rows + 127) / 128, ntile, z);
            if (n96) launch_split_in<1, 3, 4, 1>(a, g2, extra, s);
            else launch_split_pa<1, 4, 4, 1>(a, g2, extra, s);
        } else {
            use_cm();
            const dim3 g2 = bgrid((rows + 255) / 256, ntile, z);
            if (n96) launch_split_in<2, 3, 4, 1>(a, g2, extra, s);
            else launch_split_pa<2, 4, 4, 1>(a, g2, extra, s);
        }
        (void)n_thin;
        Q3_HIP_CHECK(hipGetLastError());
        return;
    }
    if (nb > 1) throw Error("conv: a batched launch needs the split-precision path (or C_out == 1)");
    hipLaunchKernelGGL(k_conv_mfma, grid, dim3(256), 0, s, a);
}

// ---- line-friendly copies of the split-precision weight planes (ConvArgs::Whc, W2fh / W2fl) ----
// planes [2][taps][cout][cin] -> [2][taps][cin / 32][cout][32]; plane_elems = the distance between the two planes (both layouts)
__global__ void k_repack_planes_cm(const bf16_t* in, bf16_t* out, int taps, int cout, int cin, size_t plane_elems) {
    const size_t n8 = (size_t)taps * cout * cin / 8;                  // 16-byte pieces per plane
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = i / n8, r = i % n8;
        const int seg = (int)(r % 4);                                 // 8 halves of the 32-channel chunk
        const size_t q = r / 4;
        const int co = (int)(q % cout);
        const size_t q2 = q / cout;
        const int ch = (int)(q2 % (cin / 32)), tap = (int)(q2 / (cin / 32));
        const uint4 v = *reinterpret_cast<const uint4*>(in + plane * plane_elems + ((size_t)tap * cout + co) * cin + ch * 32 + seg * 8);
        *reinterpret_cast<uint4*>(out + plane * plane_elems + r * 8) = v;
    }
}
void launch_repack_planes_cm(const bf16_t* planes, bf16_t* out, int taps, int cout, int cin, size_t plane_elems, hipStream_t s) {
    if (cin % 32 != 0) throw Error("repack_planes_cm: C_in must be a multiple of 32");
    hipLaunchKernelGGL(k_repack_planes_cm, dim3(1024), dim3(256), 0, s, planes, out, taps, cout, cin, plane_elems);
}
// plane [C][C] (row = output channel) -> B fragments of v_mfma_f32_32x32x16_f16: [k-step st][32-column block j][lane][8],
// lane (col0 = lane & 31, rsel = lane >> 5) holds W[j 32 + col0][st 16 + 8 rsel .. + 7]
__global__ void k_pack_w2_frags(const bf16_t* in, bf16_t* out, int C) {
    const int NB = C / 32, KS = C / 16;
    const int n = KS * NB * 64;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
        const int lane = f & 63, j = (f >> 6) % NB, st = (f >> 6) / NB;
        const uint4 v = *reinterpret_cast<const uint4*>(in + (size_t)(j * 32 + (lane & 31)) * C + st * 16 + 8 * (lane >> 5));
        *reinterpret_cast<uint4*>(out + (size_t)f * 8) = v;
    }
}
void launch_pack_w2_frags(const bf16_t* plane, bf16_t* out, int C, hipStream_t s) {
    if (C % 32 != 0) throw Error("pack_w2_frags: C must be a multiple of 32");
    hipLaunchKernelGGL(k_pack_w2_frags, dim3(16), dim3(256), 0, s, plane, out, C);
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
// grid (rows per utterance, utterances): utterance u reads codes + u * codes_stride (ints) and writes rows u * gridDim.x + t
__global__ void k_code_embed_mean(const float* table, const int32_t* codes0, int G, int codebook, int C, float* out0, size_t codes_stride, const int* perm) {
    const int t = blockIdx.x;
    const int32_t* codes = codes0 + (size_t)(perm ? perm[blockIdx.y] : (int)blockIdx.y) * codes_stride;   // perm: row block y holds utterance perm[y]
    float* out = out0 + (size_t)blockIdx.y * gridDim.x * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) {
            int code = codes[(size_t)t * G + g];
            code = code < 0 ? 0 : (code < codebook ? code : codebook - 1);   // device-resident codes are not validated on the host: never index outside the table
            s += table[((size_t)g * codebook + code) * C + c];
        }
        out[(size_t)t * C + c] = s / (float)G;
    }
}
void launch_code_embed_mean(const float* table, const int32_t* codes, int F, int G, int codebook, int C, float* out, hipStream_t s,
                            int n_utt, size_t codes_stride, const int* perm) {
    if (F > 0 && n_utt > 0) hipLaunchKernelGGL(k_code_embed_mean, dim3(F, n_utt), dim3(256), 0, s, table, codes, G, codebook, C, out, codes_stride, perm);
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
// grid (positions, utterances): utterance u owns rows [u * gridDim.x, (u + 1) * gridDim.x) of qkv and cache block u ([kvh][P][d])
__global__ void k_rope_store(float* qkv, int ld, int nq, int nkv, int d, const float* cs, const float* sn,
                             float* kc0, float* vc0, int P) {
    const int t = blockIdx.x, half = d / 2;
    float* row = qkv + ((size_t)blockIdx.y * gridDim.x + t) * ld;
    float* kc = kc0 + (size_t)blockIdx.y * nkv * P * d;
    float* vc = vc0 + (size_t)blockIdx.y * nkv * P * d;
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
                       float* kc, float* vc, int P, hipStream_t s, int n_utt) {
    if (T > 0 && n_utt > 0) hipLaunchKernelGGL(k_rope_store, dim3(T, n_utt), dim3(256), 0, s, qkv, ld, nq, nkv, d, cs, sn, kc, vc, P);
}

// ConvNeXt front half: depthwise causal k7 conv + LayerNorm(eps 1e-6) over channels, one row per block
// rows_per_utt > 0: the T rows are [utterance][rows_per_utt] blocks, the causal taps stop at the utterance's first row
__global__ __launch_bounds__(256) void k_dwconv_ln(const float* x, int T, int C, const float* dw_w, const float* dw_b,
                                                   const float* ln_w, const float* ln_b, float* out, int rows_per_utt) {
    __shared__ float red[4];
    extern __shared__ float hbuf[];
    const int t = blockIdx.x;
    const int t_first = rows_per_utt > 0 ? t - t % rows_per_utt : 0;
    float s1 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = dw_b[c];
        for (int tap = 0; tap < 7; ++tap) {
            const int ts = t - (6 - tap);
            if (ts >= t_first) acc += dw_w[c * 7 + tap] * x[(size_t)ts * C + c];
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
                      const float* ln_b, float* out, hipStream_t s, int rows_per_utt) {
    if (T > 0) hipLaunchKernelGGL(k_dwconv_ln, dim3(T), dim3(256), (size_t)C * sizeof(float), s, x, T, C, dw_w, dw_b, ln_w, ln_b, out, rows_per_utt);
}

} // namespace q3
