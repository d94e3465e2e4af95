// q3_common.h — shared declarations of libq3tts_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/q3tts.h"

typedef uint16_t bf16_t; // raw bf16 bits

#define Q3_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) throw q3::Error(std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

namespace q3 {

struct Error {
    std::string msg;
    explicit Error(std::string m) : msg(std::move(m)) {}
};

// A/B knobs of the measurement scripts and the test suite (Q3TTS_SEAM, Q3TTS_GEMM3_LA, Q3TTS_CONV_NO_PEEL, Q3TTS_ATTN_STREAM_*, ...; the
// full list is in INTEGRATION.md).  knob() answers like getenv() ONLY while at least one engine created with Q3TTS_FLAG_TEST_HOOKS is
// alive in the process, and nullptr otherwise: a stray environment variable cannot change which kernels a production engine launches.
// Read at every use (no caching): a test may flip a knob between two calls.  Defined in q3_engine.cpp.
const char* knob(const char* name);
struct KnobScope {   // member of Engine: holds the process-wide count of hook-enabled engines for the engine's lifetime
    bool on = false;
    void enable();
    ~KnobScope();
};

// fp32 -> bf16, round-to-nearest-even (inputs are finite weights)
inline bf16_t f32_to_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    return (bf16_t)u;
}
inline float bf16_to_f32(bf16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// ------------------------------------------------------------------------------------------------
// kernel argument blocks + launchers (q3_decode_kernels.hip)
// ------------------------------------------------------------------------------------------------

enum GemvEpi { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_BIAS = 3, EPI_BIAS_SILU = 4, EPI_SLAB = 5, EPI_SLAB2 = 6 };

// out[m][n] = epi( sum_k xin[m][k] * W[n][k] ),  W row-major bf16 [N][K] (nn.Linear.weight layout)
struct GemvArgs {
    const bf16_t* W = nullptr;   // [N][K]
    const bf16_t* W2 = nullptr;  // EPI_SWIGLU: up_proj rows (W = gate_proj)
    const float* x = nullptr;    // [M][ldx]
    int ldx = 0;
    const float* gamma = nullptr; // non-null: fused RMSNorm of x rows (gamma[K], eps)
    float eps = 0.f;
    float* xn_out = nullptr;      // optional: normalised rows written by block 0 (ld_xn)
    int ld_xn = 0;
    const float* bias = nullptr;  // EPI_BIAS / EPI_BIAS_SILU
    const float* res = nullptr;   // EPI_RESIDUAL (may alias out)
    int ldres = 0;
    float* out = nullptr;
    int ldo = 0;
    int M = 0, N = 0, K = 0;
    int epi = EPI_STORE;
    bool nt = false;              // non-temporal weight loads (streamed-once weights)
    // optional: x rows are the combination of split-T attention partials (o_proj prologue)
    const float* po = nullptr;    // [(row*heads + h)*S + s][d] un-normalised sum(p*v)
    const float* pm = nullptr;    // [(row*heads + h)*S + s] running max
    const float* pl = nullptr;    // [(row*heads + h)*S + s] running sum
    int pS = 0, pchunk = 0, pn_new = 1, pslot_offset = 0, pheads = 0, pd = 0;
    const int* ppos_dev = nullptr;
    int ppos_scalar = 0;
};
void launch_gemv(const GemvArgs& a, hipStream_t s);
bool gemv_fast_path(const GemvArgs& a); // single-pass kernel available (M <= 2, K in {1024,2048,3072})
bool gemv16_ok(const GemvArgs& a);        // 3..16 rows on the matrix cores, same contract (q3_gemm_kernels.hip); launch_gemv picks it
void launch_gemv16(const GemvArgs& a, hipStream_t s);
// Fragment-packed copies of the bf16 weight matrices for the matrix-core decode kernels (k_gemv16, k_gemm3; round 5).  The B operand of
// v_mfma_f32_16x16x32_bf16 wants lane (r16, q) to hold row n0 + r16, k = 32 s + 8 q .. + 7: read from the row-major [N][K] matrix that is
// 16 rows x 64 bytes per load instruction — sixteen half-used 128-byte lines.  The packed copy stores, for every (16-row tile t, k-step s),
// the 64 lanes' 16 bytes back to back: P[((t K/32 + s) 64 + lane) 8 + e] = W[min(16 t + (lane & 15), N - 1)][32 s + 8 (lane >> 4) + e], so a
// load instruction is ONE contiguous KB (eight whole lines).  Same values into the same registers: results are bit-identical.  The kernels
// find the copy through a process-wide registry keyed by the row-major pointer (engines register at finalize, unregister when destroyed);
// Q3TTS_PACKED_W=0 (A/B knob) ignores it.
void launch_pack_mfma_b(const bf16_t* W, bf16_t* P, int N, int K, hipStream_t s);
size_t packed_mfma_b_elems(int N, int K);
void register_packed_weight(const bf16_t* W, const bf16_t* P);
void unregister_packed_weight(const bf16_t* W);
const bf16_t* find_packed_weight(const bf16_t* W);   // nullptr: none (or the knob says no)

// Decode-time attention over a paged fp32 KV cache with the new tokens' q/k-norm + RoPE + append fused.
struct AttnArgs {
    const float* qkv = nullptr; // [nb*n_new][ld_qkv]: q heads | k heads | v heads (raw projections)
    int ld_qkv = 0;
    // deferred RMSNorm of the projection's input (split-K seam, GemmArgs::seam): the raw q / k / v sums are scaled by
    // 1 / sqrt(sum_t ssq_in[row][t] / ssq_K + ssq_eps) before anything else.  Null: the input planes were normalised.
    const float* ssq_in = nullptr; int ssq_nt = 0, ssq_K = 0; float ssq_eps = 0.f;
    int qkv_nslab = 1; size_t qkv_slab_stride = 0; // > 1: qkv is the sum of that many split-K partial slabs
    float* out = nullptr;       // [nb*n_new][nq*d]
    int ld_out = 0;
    float* kcache = nullptr;    // [page][layer][kvh][page_tokens][d]
    float* vcache = nullptr;
    bool kv_bf16 = false;       // the caches hold bf16 (2 bytes per element, same element layout); rounded on append, fp32 math
    bool kv_round = false;      // fp32 caches holding bf16-ROUNDED values (Q3TTS_FLAG_KV_ROUND_BF16): what kv_bf16 computes, without the 16-bit storage
    const int* page_table = nullptr; // [slot][pages_per_slot]
    int pages_per_slot = 0, page_shift = 0;
    bool identity_pages = false;     // page_table[slot][i] == slot * pages_per_slot + i by construction (code predictor): kernels may skip the table
    int layer = 0, n_layers = 0;
    const float* q_norm = nullptr; // [d] or null
    const float* k_norm = nullptr;
    float eps = 0.f;
    const float* rope_cos = nullptr; // [max_pos][d/2]
    const float* rope_sin = nullptr;
    const int* pos_dev = nullptr;    // per-slot position of new token 0 (device) or null -> pos_scalar
    int pos_scalar = 0;
    int slot_offset = 0, nb = 0, n_new = 0;
    const int* slot_map = nullptr; // optional: batch row bi belongs to slot slot_map[bi] instead of slot_offset + bi (prefill of scattered slots)
    int nq = 0, nkv = 0, d = 0;
    float scale = 0.f;
    int window = 0;
    int new_from_raw = 1; // 1: new tokens' K/V come from qkv (and are appended); 0: everything is in the cache, q pre-roped
    // split-T mode: n_splits workgroups per (kv head, new token, row), each over `chunk` tokens, write
    // un-normalised partials instead of `out`; combined by the o_proj GEMV prologue or launch_attn_combine
    int n_splits = 1, chunk = 1 << 30;
    float* po = nullptr; float* pm = nullptr; float* pl = nullptr;
    // optional (hi, lo) bf16 plane outputs of launch_attn_combine for the MFMA GEMM path
    bf16_t* oh = nullptr; bf16_t* ol = nullptr; int ldp = 0;
    // 1: k_attn_stream — the batched step's long-context kernel (one new token per row, d = 128, splits on page boundaries, K/V walked in
    // a two-deep register ring); the engine asks for it where the launch is bound by the KV bytes it streams
    int stream = 0;
};
void launch_attn(const AttnArgs& a, hipStream_t s);
// code-predictor attention + o_proj (+ residual) for one utterance: see k_cp_attn_oproj
struct CpAttnOprojArgs {
    const float* qkv = nullptr; int ld_qkv = 0; // [n_new][ld_qkv] raw q | k | v projections
    float* kc = nullptr; float* vc = nullptr;    // this (slot, layer)'s cache rows: [nkv][page_tokens][d]
    int page_tokens = 0;
    int base = 0;                                // tokens already cached = position of new row 0
    const float* q_norm = nullptr; const float* k_norm = nullptr; float eps = 0.f;
    const float* rope_cos = nullptr; const float* rope_sin = nullptr;
    float scale = 0.f;
    int nq = 0, nkv = 0, d = 0;
    const bf16_t* W = nullptr; int K = 0, N = 0; // o_proj [N][K]
    float* x = nullptr; int ldx = 0;             // residual stream rows [n_new][ldx], updated in place
};
bool cp_attn_oproj_ok(const CpAttnOprojArgs& a, int n_new);
void launch_cp_attn_oproj(const CpAttnOprojArgs& a, int n_new, hipStream_t s);
void launch_attn_combine(const AttnArgs& a, hipStream_t s); // partials -> a.out

// Skinny-M bf16-MFMA GEMM (q3_gemm_kernels.hip): activations as (hi, lo) bf16 planes, fp32 accumulate
struct GemmArgs {
    const bf16_t* W = nullptr;   // [N][K]
    const bf16_t* W2 = nullptr;  // EPI_SWIGLU: up_proj rows
    const bf16_t* xh = nullptr;  // [M][ldx] hi plane
    const bf16_t* xl = nullptr;  // [M][ldx] lo plane
    int ldx = 0;
    const float* res = nullptr; int ldres = 0;
    const float* bias = nullptr;
    float* out = nullptr; int ldo = 0;        // fp32 output (may be null when only planes are wanted)
    float* out2 = nullptr;                    // EPI_SLAB2: slabs of the W2 product
    bf16_t* oh = nullptr; bf16_t* ol = nullptr; int ldp = 0; // optional plane outputs (input of the next GEMM)
    int M = 0, N = 0, K = 0, epi = EPI_STORE;
    int slab_rows = 0;   // EPI_SLAB / EPI_SLAB2: rows per slab (0 = M); launch_gemm2 sets it when it cuts M into 128-row blocks
    bool nt = false;     // non-temporal weight loads (weights this step reads once: the talker's)
    bool w_packed = false;      // W / W2 point at fragment-packed copies (set by launch_gemm3 from the registry)
    bool plain_slabs = false;   // A/B knob Q3TTS_GEMM_PLAIN_SLABS: k_gemm3's slabs as plain stores instead of write-through (sc1)
    // ---- split-K seam (k_gemm3 only): the slabs are reduced INSIDE the launch by the K-slice workgroups of a column tile themselves
    // (sc1 slab stores, a flag per slice, slice s < 16-row-chunk count owns chunk s once every flag is set), instead of by a k_finish* launch.
    // seam 1: x += sum(slabs); planes = split(gamma * x) — NOT normalised: the consumer applies 1/rms from the per-(row, tile) sums of
    //         squares written to ssq_out (deferred RMSNorm, as k_gemv16 does);
    // seam 2: planes = split(silu(r * sum(gate slabs)) * (r * sum(up slabs))), r = 1/rms of the INPUT planes' rows from ssq_in.
    int seam = 0;
    unsigned* seam_cnt = nullptr;                    // one 64-byte line per (column tile, row block): words 0..11 slice flags, 12..15 abandoned-chunk marks
    const unsigned* seam_gen = nullptr;              // the step's generation (bumped once per step by the first sampler): every word a launch writes carries it,
                                                     // so nothing is ever reset and a copy of the line left over from an earlier step can never read as set
    int seam_spin = 512;                             // polls an owner makes before it abandons its chunk (~0.7 us each)
    float* sx = nullptr; int sldx = 0;               // seam 1: residual stream rows, updated in place
    const float* sgamma = nullptr;                   // seam 1: the consumer's RMSNorm gain
    float* ssq_out = nullptr; int ssq_nt = 0;        // seam 1: [M][ssq_nt] partial sums of squares, one per 64-column tile
    const float* ssq_in = nullptr; int ssq_in_nt = 0; float seps = 0.f;   // seam 2 (K = this GEMM's K): the input rows' partial sums of squares
};
void launch_gemm2(const GemmArgs& a, int ksplit, int nw, hipStream_t s); // EPI_SLAB: out = slabs [ksplit][M][ldo]
bool gemm_seam_ok(const GemmArgs& a, int ksplit);                        // the in-launch split-K reduction (GemmArgs::seam) covers this launch
void launch_finish(float* x, int ldx, const float* slab, int nslab, size_t slab_stride, int ld_slab, const float* gamma, float eps,
                   int rows, int K, bf16_t* oh, bf16_t* ol, int ldp, float* xn_out, int ld_xn, hipStream_t s);
void launch_finish_swiglu(const float* gs, const float* us, int nslab, size_t slab_stride, int rows, int N,
                          bf16_t* oh, bf16_t* ol, int ldp, hipStream_t s);

struct SlotState { // device-resident per-slot generation state
    int32_t n_frames;     // frames recorded so far
    int32_t finished;     // 1 after EOS / max_frames
    int32_t active;       // slot armed
    int32_t prompt_len;
    int32_t trailing_len;
    int32_t max_frames;
    int32_t top_k;
    int32_t ignore_eos;
    float temperature, top_p;
    uint32_t stream_id;
    uint32_t pad0;
    uint64_t seed;
    uint32_t pad1[2]; // 64 bytes: read by the sampler as four 16-byte loads
};
static_assert(sizeof(SlotState) == 64, "SlotState must be 64 bytes");

struct SampleArgs {
    const float* logits = nullptr; // [nb][ld]
    int ld = 0, V = 0, nb = 0;
    int nslab = 1; size_t slab_stride = 0; // > 1: the logits are the sum of that many split-K partial slabs (batched predictor heads), summed in slab order
    int sup_begin = 0, sup_end = 0, eos_id = -1; // suppress [sup_begin,sup_end) except eos_id (group 0 only)
    int group = 0, n_groups = 0;
    SlotState* st = nullptr;       // [nb]; null -> standalone mode (params below, token_out)
    float temperature = 1.f, top_p = 1.f;
    int top_k = 0;
    float u = 0.f;
    const float* u_dev = nullptr;   // standalone mode, optional: row b draws with u_dev[b] instead of u (q3tts_sample_dev)
    int suppress = 0;
    int64_t* token_out = nullptr;
    // fused epilogue (generation mode)
    const bf16_t* embed = nullptr; // [V][H] embedding table of the sampled codebook
    int H = 0;
    float* x_next = nullptr;       // row b*ld_xnext: embedding of the sampled token (next predictor input)
    int ld_xnext = 0;
    float* sum = nullptr;          // [nb][H] running fp32 sum of the frame's 16 embeddings
    float* x_talk = nullptr;       // [nb][H], last group only: sum + text row | tts_pad
    const float* trailing = nullptr; // [nb][max_trailing][H]
    int max_trailing = 0;
    const float* tts_pad = nullptr;  // [H]
    int32_t* codes = nullptr;      // [nb][max_frames_cap][n_groups]
    int max_frames_cap = 0;
    int32_t* talker_pos = nullptr; // [nb], last group only: position of the token the talker decodes next
    // optional (batched step with the split-K seam): the sampler also prepares the NEXT predictor pass's input planes, so that the pass
    // needs no RMSNorm launch in front of its first projection — (hi, lo) planes of gamma0 * row (NOT normalised: deferred RMSNorm) and
    // the row's sum of squares in ssq_out[row * ssq_nt + 0] (the other ssq_nt - 1 partials zero).  Row of utterance b: b * pl_row_mul +
    // pl_row_add.  lh (group 0 only): the talker's last_hidden rows [nb][ld_lh], which pass 0 takes as row b * 2.
    bf16_t* pl_h = nullptr; bf16_t* pl_l = nullptr; int pl_ldp = 0;
    const float* gamma0 = nullptr;
    float* ssq_out = nullptr; int ssq_nt = 0;
    int pl_row_mul = 1, pl_row_add = 0;
    const float* lh = nullptr; int ld_lh = 0;
    unsigned* step_gen = nullptr;  // group 0 only: the step's generation counter (split-K seam flags), bumped by workgroup 0
};
void launch_sample(const SampleArgs& a, hipStream_t s);

void launch_gather_rows_bf16(const bf16_t* table, int H, const int64_t* ids_dev, int n, float* out, int ldo, hipStream_t s);
void launch_fill_synth(void* dst, int is_bf16, int64_t n, uint64_t key, float mean, float stddev, hipStream_t s);
void launch_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);
// rows whose flag is 0 are skipped (flags: device ints, one per row)
void launch_copy_rows_masked(const float* src, int lds, float* dst, int ldd, int rows, int cols, const int* flags_dev, hipStream_t s);
void launch_bump_u32(unsigned* counter, hipStream_t s);   // *counter += 1 (the split-K seam's step generation outside the fused step)
void launch_count_active(const SlotState* st, int nb, int32_t* out, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// codec decoder launchers (q3_codec_kernels.hip)
// ------------------------------------------------------------------------------------------------
struct ConvArgs { // out[t][co] = epi(bias[co] + sum_{tap,ci} W[tap][co][ci] * in[src_t(t,tap)][ci])
    const float* in = nullptr; int T_in = 0, C_in = 0;
    float* out = nullptr;      int T_out = 0, C_out = 0; // out may be null when only out2 is wanted
    const float* W = nullptr;  // [taps][C_out][C_in] fp32
    const bf16_t* Wh = nullptr; const bf16_t* Wl = nullptr; // optional (hi, lo) fp16 planes of W * 2^k: selects the split-precision MFMA kernel
    float w_scale_inv = 1.0f;                                  // 2^-k
    bool w_lo_zero = false, w2_lo_zero = false;                // the lo plane of W (W2) is identically zero: two matrix-core products instead of three, exact
    const float* bias = nullptr;
    int taps = 1, dil = 1;
    int transposed = 0, stride = 1, left = 0; // transposed: out index jo = m*stride + phase - left
    int act = 0;                 // 0 none, 1 GELU(erf), 2 SiLU, applied to (acc + bias)
    const float* mul = nullptr;  // optional elementwise multiplier after act (SwiGLU up branch)
    const float* res_scale = nullptr; // optional per-channel scale (LayerScale / ConvNeXt gamma)
    const float* res = nullptr;  // optional residual (same shape as out, may alias out)
    int clamp = 0;               // clamp to [-1, 1]
    float* out2 = nullptr;       // optional second output: SnakeBeta(value) for the NEXT layer
    const float* snake_alpha = nullptr;
    const float* snake_beta = nullptr;
    bool in_planes = false, out2_planes = false;   // `in` / `out2` hold, per 4 channels, 4 hi + 4 lo fp16 halves in place of the 4 floats (k_conv_split; see ConvKArgs)
    const float* snake_pre = nullptr;   // [2][C_out]: exp(alpha) | 1 / (exp(beta) + 1e-9), launch_snake_pre (split-precision path)
    float* slab = nullptr; size_t slab_floats = 0; // optional scratch for split-K partial sums of short 1-tap GEMMs ([slice][T_out][C_out])
    // fused residual unit (96-channel decoder block): out = res + bias2 + W2 . snake_mid(bias + W . in) — the 7-tap conv, the SnakeBeta
    // between, the 1x1 conv and the residual add in ONE launch; the intermediate never leaves the CU.  W2 = the 1x1 conv's [1][C_out][C_out]
    const float* W2 = nullptr; const bf16_t* W2h = nullptr; const bf16_t* W2l = nullptr; float w2_scale_inv = 1.0f;
    // Round 5, line-friendly copies of the weight planes (same values; launch_repack_planes_cm / launch_pack_w2_frags at finalize):
    // Whc = the (hi, lo) planes CHUNK-major, [plane][tap][C_in / 32][C_out][32]: a workgroup's weight tile of one (tap, 32-channel chunk)
    // is contiguous (96 rows x 64 B = 6 KB) instead of 96 separate 64-byte row pieces — k_conv_split's 32-wide-chunk kernels stage it with
    // whole-line loads (the 128-wide-chunk kernels of the short GEMMs keep the row-major planes: their rows are 256 B already);
    // W2fh / W2fl = the fused unit's 96 x 96 second conv in MFMA B-fragment order, [k-step][32-column block][lane][8]: the unit's weight
    // fragments come straight from global memory, one contiguous KB per load instead of 32 rows x 32 bytes.
    const bf16_t* Whc = nullptr; const bf16_t* W2fh = nullptr; const bf16_t* W2fl = nullptr;
    const float* bias2 = nullptr; const float* mid_alpha = nullptr; const float* mid_beta = nullptr; const float* mid_pre = nullptr;
    int batch = 1;               // independent sequences of the same shape: sequence u at in + u * in_ustride, out / out2 / res / mul at + u * T_out * C_out
    size_t in_ustride = 0;       // floats between the sequences' inputs (0: T_in * C_in, i.e. densely packed)
};
void launch_conv(const ConvArgs& a, hipStream_t s);
void cout1_debug_buffer(const float** p, size_t* n);   // experiment hook (Q3TTS_COUT1_PACKED=2)
void launch_split_planes(const float* w, bf16_t* hi, bf16_t* lo, size_t n, float scale, hipStream_t s);
void launch_snake_pre(const float* alpha, const float* beta, float* pre /* [2][C] */, int C, hipStream_t s);
void launch_repack_planes_cm(const bf16_t* planes /* [2][taps][cout][cin] */, bf16_t* out /* [2][taps][cin/32][cout][32] */, int taps, int cout, int cin, size_t plane_elems, hipStream_t s);
void launch_pack_w2_frags(const bf16_t* plane /* [C][C] */, bf16_t* out /* [C/16][C/32][64][8] */, int C, hipStream_t s);
void launch_absmax(const float* w, size_t n, unsigned* out, hipStream_t s);
void launch_or_mag16(const bf16_t* p, size_t n, unsigned* out /* |= magnitude bits */, hipStream_t s);
void launch_repack_conv(const float* w, float* out, int cin, int cout, int k, int transposed, hipStream_t s);
void launch_code_embed_mean(const float* table, const int32_t* codes, int F, int G, int codebook, int C, float* out, hipStream_t s,
                            int n_utt = 1, size_t codes_stride = 0, const int* perm = nullptr);
void launch_rmsnorm_rows(const float* x, const float* w, float eps, int rows, int C, float* out, hipStream_t s);
void launch_rope_store(float* qkv, int ld, int T, int nq, int nkv, int d, const float* cs, const float* sn,
                       float* kc, float* vc, int P, hipStream_t s, int n_utt = 1);
void launch_dwconv_ln(const float* x, int T, int C, const float* dw_w, const float* dw_b, const float* ln_w,
                      const float* ln_b, float* out, hipStream_t s, int rows_per_utt = 0);


// ---- speaker encoder (q3_speaker_kernels.hip) ----
struct SpkConvArgs {
    const float* x = nullptr; int ldx = 0;   // [T][ldx] time-major (or [Cin][ldx] when x_channel_major)
    const float* x2 = nullptr; int ldx2 = 0; // optional second input added to x before the convolution
    int x_channel_major = 0;
    int T = 0, Cin = 0, Cout = 0, k = 1, dil = 1;
    int act = 0;                              // 0 none, 1 ReLU, 2 tanh(ReLU)
    const float* W = nullptr;                 // [k][Cin][Cout]
    const float* bias = nullptr;
    float* y = nullptr; int ldy = 0;
};
void launch_spk_conv(const SpkConvArgs& a, hipStream_t s);
void launch_spk_repack(const float* w, float* out, int cout, int cin, int k, hipStream_t s);
void launch_spk_colstats(const float* x, int ld, int T, int C, float* mean, float* sd, hipStream_t s);
void launch_spk_se_gate(const float* y, const float* g, float* h, float* cat, int ld_cat, int T, int C, hipStream_t s);
void launch_spk_asp_input(const float* x, const float* mean, const float* sd, float* out, int T, int C, hipStream_t s);
void launch_spk_asp_pool(const float* scores, const float* x, int T, int C, float* out, hipStream_t s);

} // namespace q3
