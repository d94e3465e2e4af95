/*
 * q3_oracle.h — CPU ORACLE for the Qwen3-TTS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (leaxer-qwen3-tts_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the network arithmetic.  The reference
 * (/root/reference, leaxer-ai/leaxer-qwen3-tts v0.2.0) owns no neural-network arithmetic:
 * every matmul runs inside ONNX Runtime 1.20.0 (CMakeLists.txt:23-46, ci.yml:10) executing
 * seven .onnx graphs from HF zukky/Qwen3-TTS-ONNX-DLL (README.md:71-89); neither ORT nor the
 * graphs exist in this image, and the reference's tests hold no golden vector for this path
 * (tests/test_onnx.cpp only checks constants).  What this file restates:
 *   - the reference's HOST logic, behaviour-for-behaviour, with file:line citations
 *     (prompt assembly src/tts_onnx.cpp:442-539, generation loop :782-849, predict_subcodes
 *     :851-872, sampler :878-950, session tensor contracts :545-776);
 *   - the published architecture the opaque graphs implement (Qwen3 decoder layer,
 *     Qwen3-Omni talker code predictor and Code2Wav), cross-checked on seeded tiny configs
 *     against the `transformers` implementation installed in the build container
 *     (tests/golden/make_hf_goldens.py -> tests/golden/hf_*.npz).
 *
 * All arithmetic is fp32 (the reference's ORT CPU path is fp32), weights are whatever the
 * caller uploads (tests upload bf16-representable values so the GPU's bf16 weight storage is
 * lossless).
 */
#ifndef Q3_ORACLE_H
#define Q3_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as q3tts_config in include/q3tts.h (kept standalone on purpose). */
typedef struct q3o_config {
    /* talker (reference constants: src/tts_onnx.h:31-37) */
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, ffn, vocab;
    float rope_theta, rms_eps;
    /* code predictor */
    int32_t cp_layers, cp_heads, cp_kv_heads, cp_head_dim, cp_ffn, n_groups, sub_vocab;
    float cp_rope_theta, cp_rms_eps;
    /* text_project */
    int32_t text_vocab, text_hidden;
    /* 12 Hz codec decoder */
    int32_t cd_codebook, cd_hidden, cd_layers, cd_heads, cd_head_dim, cd_ffn, cd_window;
    float cd_rope_theta, cd_rms_eps;
    int32_t cd_n_up;          /* ConvNeXt upsample stages (2) */
    int32_t cd_up_ratios[4];  /* (2,2) */
    int32_t cd_decoder_dim;   /* 1536 */
    int32_t cd_n_blocks;      /* 4 */
    int32_t cd_up_rates[8];   /* (8,5,4,3) */
    int32_t cd_tconv_trim;    /* 0: trim k-s on BOTH sides (transformers code as written), 1: right only */
    /* generation-loop constants (src/tts_onnx.h:50-51, tts_onnx.cpp:803-807) */
    int32_t codec_eos, suppress_begin, suppress_end;
    /* speaker encoder of the voice-clone path (speaker_encoder.onnx, src/tts_onnx.cpp:367-403): ECAPA-TDNN
     * [HINT: transformers qwen2_5_omni ECAPA_TimeDelayNet].  spk_enc_dim == 0: no speaker encoder.
     * Channel plan (C, C, C, C, 3C), kernels (5,3,3,3,1), dilations (1,2,3,4,1). */
    int32_t spk_enc_dim, spk_mel, spk_channels, spk_scale, spk_se, spk_att;
    /* width of the code predictor's layers; 0 or == hidden: same as the talker (0.6B).  Otherwise (1.7B: 2048 -> 1024) every predictor
     * input row goes through cp.proj (Linear with bias) first; predictor embeddings stay at the talker's width. */
    int32_t cp_hidden;
} q3o_config;

/* src/tts_onnx.h:99-105 */
typedef struct q3o_sampling {
    float temperature, top_p;
    int32_t top_k;
    float repetition_penalty; /* declared, never read by the reference */
    int32_t max_new_tokens;
} q3o_sampling;

typedef struct q3o_model q3o_model;

q3o_model* q3o_create(const q3o_config* cfg, int max_ctx);
/* talker K / V rows rounded to bf16 (round-to-nearest-even) on append, like the product under Q3TTS_FLAG_KV_BF16; off by default (fp32 cache) */
void q3o_set_kv_bf16(q3o_model* m, int on);
/* diagnostic: the sampler's softmax with libm expf (the reference's std::exp, tts_onnx.cpp:912) instead of q3o_expf; process-global, off by default */
void q3o_set_sampler_exp_libm(int on);
void q3o_destroy(q3o_model* m);
const char* q3o_last_error(void);
/* copies n floats; returns 0 ok, <0 unknown name / wrong element count */
int q3o_set_tensor(q3o_model* m, const char* name, const float* data, int64_t n);
int64_t q3o_tensor_numel(q3o_model* m, const char* name);
void q3o_set_threads(int n);

/* ---- session-shaped entry points (tensor contracts: src/tts_onnx.cpp:545-776) ---- */
int q3o_text_project(q3o_model* m, const int64_t* ids, int n, float* out /*[n][H]*/);
int q3o_codec_embed(q3o_model* m, const int64_t* ids, int n, float* out /*[n][H]*/);
int q3o_cp_embed(q3o_model* m, int64_t id, int step, float* out /*[H]*/);
/* prefill resets the KV cache; logits for every row, last_hidden for the final row */
int q3o_prefill(q3o_model* m, const float* embeds, int S, float* logits /*[S][V]*/, float* last_hidden /*[H]*/);
int q3o_decode(q3o_model* m, const float* embed /*[H]*/, float* logits /*[V]*/, float* last_hidden /*[H]*/);
/* full re-run over n rows, no KV cache, head #step on the last row (tts_onnx.cpp:734-757) */
int q3o_code_predictor(q3o_model* m, const float* seq /*[n][H]*/, int n, int step, float* logits /*[SV]*/);
/* returns number of samples written to pcm (capacity cap); *out_len = lengths[0] */
int64_t q3o_vocoder(q3o_model* m, const int64_t* codes /*[F][G]*/, int F, float* pcm, int64_t cap);
int64_t q3o_vocoder_len(const q3o_config* cfg, int F);
/* codec-decoder intermediate taps for kernel-level parity: stage 0 = after pre-transformer+norm
 * [F][H]; 1 = after upsample stages [4F][H]; 2 = after decoder.conv_in [4F][D]; 3.. = after block i */
int64_t q3o_vocoder_tap(q3o_model* m, const int64_t* codes, int F, int stage, float* out, int64_t cap);

/* run_speaker_encoder (tts_onnx.cpp:367-403): mel [n_mels][frames] as MelExtractor::extract returns it
 * (the session input is its transpose [1, frames, n_mels]); out [spk_enc_dim].  frames >= 5. */
int q3o_speaker_encoder(q3o_model* m, const float* mel, int frames, float* out);

/* ---- sampler (tts_onnx.cpp:878-950) with a counter-based RNG instead of mt19937 ---- */
float q3o_rng_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t group);
int64_t q3o_sample(const float* logits, int n, const q3o_sampling* p, float u);
/* the same, also reporting how far the decision was from flipping (top-k gap, top-p cut, draw edge; relative to total probability 1) */
int64_t q3o_sample_margin(const float* logits, int n, const q3o_sampling* p, float u, float* margin);
/* debugging aid: the next q3o_generate* call copies the logits row of decision (frame, group) into buf (NULL: off) */
void q3o_set_logits_dump(int frame, int group, float* buf);
/* the running sums behind q3o_sample's decisions (top-p: sorted order; draw: index order, -1 where p == 0) and the draw's total */
void q3o_sample_trace(const float* logits, int n, const q3o_sampling* p, float* topp_cum, float* draw_cum, float* total);
void q3o_softmax(float* x, int n);
/* the sampler's exp: IEEE-exact operations only, bit-identical to the HIP sampler's q3_expf */
float q3o_expf(float x);
void q3o_top_k_filter(float* x, int n, int k);
void q3o_top_p_filter(float* probs, int n, float p);

/* ---- host logic (tts_onnx.cpp:442-539, 782-872) ---- */
/* lang: 0 Auto, 1 English, 2 Chinese, 3 Japanese, 4 Korean (tts_onnx.h:73-79) */
int q3o_build_prompt(q3o_model* m, const int64_t* ids, int n_ids, int lang, const float* speaker /*[H] or NULL*/,
                     float* prompt /*[<=16][H]*/, int* S);
int q3o_trailing(q3o_model* m, float* out /*[trailing_len][H]*/, int cap_rows, float* pad /*[H]*/);
/* cp_cached: 0 = reference call pattern (re-run 2..16 rows, no cache), 1 = KV-cached predictor.
 * ignore_eos: keep generating past CODEC_EOS (benchmark mode, never samples EOS as a frame) */
int q3o_generate(q3o_model* m, const float* prompt, int S, const q3o_sampling* p, uint64_t seed, uint32_t stream,
                 int cp_cached, int ignore_eos, int64_t* codes /*[max_new][G]*/);
/* the same, also reporting per frame the top-2 logit margin of the code0 decision and the smallest margin over its sub-codes */
int q3o_generate_margins(q3o_model* m, const float* prompt, int S, const q3o_sampling* p, uint64_t seed, uint32_t stream,
                         int cp_cached, int ignore_eos, int64_t* codes, float* margins /*[max_new][2 + n_groups]*/);
int64_t q3o_synthesize_tokens(q3o_model* m, const int64_t* ids, int n_ids, int lang, const q3o_sampling* p,
                              uint64_t seed, uint32_t stream, float* pcm, int64_t cap, int64_t* codes, int* n_frames);

#ifdef __cplusplus
}
#endif
#endif
