/*
 * q3tts.h — C-ABI of libq3tts_hip.so: the MI355X (gfx950) replacement for the seven ONNX Runtime
 * sessions of leaxer-ai/leaxer-qwen3-tts.
 *
 * The reference has no plugin/FFI layer; its narrowest seam is the private run_* family of
 * TTSEngine (reference src/tts_onnx.h:196-212, src/tts_onnx.cpp:545-776), each a named-tensor
 * ONNX Runtime Session::Run over host vectors.  Every entry point below cites the run_* call it
 * replaces.  Conventions: int return (0 ok, <0 error; message via q3tts_last_error), no exceptions
 * cross the ABI, plain pointers and sizes only.  "_host" entry points take HOST pointers exactly
 * like the reference's run_* (inputs caller-owned, outputs copied out); the batched generation
 * entry points keep everything (KV cache, logits, codes) resident in HBM.  A handle is not
 * thread-safe: one in-flight call per handle, like one TTSEngine (tts_onnx.h:182-186).
 */
#ifndef Q3TTS_H
#define Q3TTS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Model dimensions.  The reference hard-codes the 0.6B talker dims (tts_onnx.h:31-37); the rest
 * are properties of the opaque graphs ([HINT]).  Runtime-configurable so the
 * same library serves the 1.7B export and the tiny test configs. */
typedef struct q3tts_config {
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, ffn, vocab;
    float rope_theta, rms_eps;
    int32_t cp_layers, cp_heads, cp_kv_heads, cp_head_dim, cp_ffn, n_groups, sub_vocab;
    float cp_rope_theta, cp_rms_eps;
    int32_t text_vocab, text_hidden;
    int32_t cd_codebook, cd_hidden, cd_layers, cd_heads, cd_head_dim, cd_ffn, cd_window;
    float cd_rope_theta, cd_rms_eps;
    int32_t cd_n_up;
    int32_t cd_up_ratios[4];
    int32_t cd_decoder_dim;
    int32_t cd_n_blocks;
    int32_t cd_up_rates[8];
    int32_t cd_tconv_trim; /* 0: trim k-s on both sides, 1: right side only */
    int32_t codec_eos, suppress_begin, suppress_end; /* tts_onnx.h:51, tts_onnx.cpp:803-807 */
    /* speaker encoder of the clone path (speaker_encoder.onnx, tts_onnx.cpp:367-403): ECAPA-TDNN, channel plan
     * (C, C, C, C, 3C), kernels (5,3,3,3,1), dilations (1,2,3,4,1).  spk_enc_dim == 0: no speaker encoder
     * (has_speaker_encoder() false, as when the reference finds no speaker_encoder.onnx). */
    int32_t spk_enc_dim, spk_mel, spk_channels, spk_scale, spk_se, spk_att;
    /* width of the code predictor's layers; 0 (or == hidden): the talker's width, as in the 0.6B export.  Otherwise (1.7B: hidden 2048,
     * cp_hidden 1024) every predictor input row passes through cp.proj (Linear + bias, [HINT] Qwen3-TTS small_to_mtp_projection) first;
     * the predictor's code embeddings stay talker-wide.  The code_predictor session contract (tts_onnx.cpp:734-757) is unchanged. */
    int32_t cp_hidden;
} q3tts_config;

/* SamplingParams, reference src/tts_onnx.h:99-105 (repetition_penalty is never read there) */
typedef struct q3tts_sampling {
    float temperature, top_p;
    int32_t top_k;
    float repetition_penalty;
    int32_t max_new_tokens;
} q3tts_sampling;

typedef struct q3tts_engine q3tts_engine;

/* flags for q3tts_create */
#define Q3TTS_FLAG_NO_GRAPH 1u   /* launch the decode step eagerly instead of replaying a hipGraph */
#define Q3TTS_FLAG_NO_FUSED_CP 2u /* b = 1: keep code-predictor attention and o_proj as separate launches (A/B testing) */
#define Q3TTS_FLAG_KV_BF16 8u     /* talker KV cache in bf16: K / V rows rounded to bf16 (round-to-nearest-even) where they enter the cache, fp32 attention
                                   * math on the rounded rows — half the cache bytes of the default fp32 cache (which mirrors the reference's fp32
                                   * KVCache, src/tts_onnx.h:108-115).  The CPU oracle has the same switch; rounding being a discontinuity, two implementations
                                   * agree to ~4e-3 on logits in this mode (2e-5 with fp32 caches), ids to the first sub-noise decision. */
#define Q3TTS_FLAG_KV_ROUND_BF16 16u /* test aid: fp32 KV storage holding the bf16-ROUNDED rows — the arithmetic of Q3TTS_FLAG_KV_BF16 without its 16-bit storage;
                                     * the two modes must agree bit for bit (tests/test_gpu_full.py), which pins the bf16 load / store / convert path */
#define Q3TTS_FLAG_TEST_HOOKS 32u  /* honour the fault-injection environment hooks of the test suite (Q3TTS_TEST_FAIL_VOCODER_SUBMIT); without it they are ignored */
#define Q3TTS_FLAG_FP32_CODEC 4u  /* codec decoder on the exact-fp32 matrix-core path instead of the fp16 (hi, lo) split-operand path */

/* ---- lifecycle (replaces TTSEngine ctor / load_model, tts_onnx.cpp:84-232) ---- */
int q3tts_default_config(const char* name /* "0.6b" | "1.7b" */, q3tts_config* out);
q3tts_engine* q3tts_create(const q3tts_config* cfg, int device, int max_batch, int max_ctx, uint32_t flags);
/* The same with a bounded KV page pool.  The talker's cache (the reference's KVCache, tts_onnx.h:108-115, grown by one token per run_decode)
 * is a pool of 64-token pages; a slot takes pages for prompt + max_new_tokens when it is armed (q3tts_slot_begin / the scheduler) or as its
 * context grows (q3tts_talker_prefill_host / q3tts_talker_decode_host) and returns them at q3tts_slot_release.  kv_pool_tokens = 0 sizes the
 * pool for max_batch x max_ctx (q3tts_create); a smaller pool admits as many utterances as fit — q3tts_slot_begin fails with "KV page pool
 * exhausted" and arms nothing; the scheduler (q3tts_synthesize_*) keeps the rest queued.  There, with ignore_eos (lengths known) an
 * utterance is admitted when prompt + cap fit; otherwise slots grow page by page as they generate, so utterances that stop early never
 * hold the pages of their cap, and when the pool runs dry the youngest running utterance is preempted and generated again later. */
q3tts_engine* q3tts_create_pooled(const q3tts_config* cfg, int device, int max_batch, int max_ctx, int64_t kv_pool_tokens, uint32_t flags);
int q3tts_kv_pool_info(q3tts_engine* e, int* page_tokens, int* total_pages, int* free_pages);
/* The last q3tts_synthesize_* call on this engine: utterances admitted to a slot (re-admissions count), utterances preempted because the
 * pool ran dry (each is generated again from its prompt: same RNG stream, same codes), most utterances running at once. */
int q3tts_sched_stats(q3tts_engine* e, int64_t* admitted, int64_t* preempted, int* peak_live);
void q3tts_destroy(q3tts_engine* e);
const char* q3tts_last_error(q3tts_engine* e); /* e may be NULL: error of the last failed create */

/* ---- weights: tensors by name (names = the oracle's / DESIGN.md section 3) ---- */
/* The registry a config implies, without an engine (host-only, no GPU needed): what a checkpoint converter must provide.
 * kind: 0 matrix / conv weight, 1 norm weight, 2 bias, 3 LayerScale / gamma, 4 SnakeBeta alpha / beta. */
int q3tts_config_num_tensors(const q3tts_config* cfg);
int q3tts_config_tensor_info(const q3tts_config* cfg, int index, char* name, int name_cap, int64_t* shape4, int* ndim, int* kind);
int q3tts_num_tensors(q3tts_engine* e);
int q3tts_tensor_info(q3tts_engine* e, int index, char* name, int name_cap, int64_t* shape4, int* ndim);
int q3tts_set_tensor_host(q3tts_engine* e, const char* name, const float* data, int64_t numel);
int q3tts_get_tensor_host(q3tts_engine* e, const char* name, float* out, int64_t numel);
/* seeded on-device synthetic weights (integer hash -> Irwin-Hall normal, bf16-representable) */
int q3tts_fill_synthetic(q3tts_engine* e, uint64_t seed);
/* call after the last set_tensor / fill: builds RoPE tables, packed conv weights, tts_pad row */
int q3tts_finalize(q3tts_engine* e);

/* Weight files ("Q3TW0001": config + named tensors, fp32 or bf16 payloads; written by
 * q3tts_save_weights_file or tools/pack_weights.py).  This is what TTSEngine(model_dir) loads in
 * place of the reference's seven .onnx files (tts_onnx.cpp:91-107). */
int q3tts_read_weights_config(const char* path, q3tts_config* out);
int q3tts_load_weights_file(q3tts_engine* e, const char* path);   /* set_tensor for every entry + finalize */
int q3tts_save_weights_file(q3tts_engine* e, const char* path);

/* ---- session-shaped entry points, host I/O, one per reference run_* ---- */
/* run_text_project, tts_onnx.cpp:545-559: ids[n] -> out[n][hidden] */
int q3tts_text_project_host(q3tts_engine* e, const int64_t* ids, int n, float* out);
/* run_codec_embed / run_codec_embed_batch, tts_onnx.cpp:561-590 */
int q3tts_codec_embed_host(q3tts_engine* e, const int64_t* ids, int n, float* out);
/* run_code_predictor_embed, tts_onnx.cpp:592-613 */
int q3tts_cp_embed_host(q3tts_engine* e, int64_t id, int generation_step, float* out);
/* run_prefill, tts_onnx.cpp:615-665: embeds[S][hidden] -> logits[S][vocab], last_hidden[hidden];
 * the KV cache of `slot` is reset and stays device-resident (replaces KVCache, tts_onnx.h:108-115) */
int q3tts_talker_prefill_host(q3tts_engine* e, int slot, const float* embeds, int S, float* logits, float* last_hidden);
/* run_decode, tts_onnx.cpp:667-732: one token appended to `slot` */
int q3tts_talker_decode_host(q3tts_engine* e, int slot, const float* embed, float* logits, float* last_hidden);
/* run_code_predictor, tts_onnx.cpp:734-757: seq[n][hidden], head #generation_step on the last row */
int q3tts_code_predictor_host(q3tts_engine* e, const float* seq, int n, int generation_step, float* logits);
/* run_vocoder, tts_onnx.cpp:759-776: codes[F][n_groups] (frame-major) -> pcm; *out_len = lengths[0].
 * One call decodes at most ~5500 frames (7 minutes: a decoder activation must stay below 4 GB, the conv kernels address it with 32-bit
 * offsets); longer utterances go through q3tts_codec_decode_chunked_host. */
int q3tts_codec_decode_host(q3tts_engine* e, const int64_t* codes, int F, float* pcm, int64_t cap, int64_t* out_len);
int64_t q3tts_codec_decode_len(const q3tts_config* cfg, int F);
/* run_vocoder for a whole job in one call (the reference calls run_vocoder once per utterance, tts_onnx.cpp:418-432 / :759-776; this is the
 * vocoder phase of q3tts_synthesize_schedule_host on its own): utterance u's codes are codes[frame_offsets[u] .. frame_offsets[u+1])
 * frames of n_groups ids each.  Utterances of similar length share one batched pass, the rest go one at a time over the side lanes;
 * pcm_out[u] receives up to pcm_cap samples, pcm_len[u] the utterance's sample count (q3tts_codec_decode_len of its frames; 0 for an
 * utterance without frames).  Each result equals q3tts_codec_decode_host of the same utterance to fp32 rounding; ids outside
 * [0, codebook) and utterances beyond the engine's frame capacity are rejected like there. */
int q3tts_codec_decode_batch_host(q3tts_engine* e, int n_utt, const int64_t* codes, const int32_t* frame_offsets, float* const* pcm_out,
                                  int64_t pcm_cap, int64_t* pcm_len);
/* the same with both ends in HBM: codes_dev int32 [F][n_groups] and pcm_dev float [cap] are DEVICE pointers on the engine's GPU (any
 * allocator: hipMalloc, a torch tensor's data_ptr); values outside [0, codebook) are clamped.  Returns after the work has completed. */
int q3tts_codec_decode_dev(q3tts_engine* e, const int32_t* codes_dev, int F, float* pcm_dev, int64_t cap, int64_t* out_len);
/* the engine's HIP stream as an opaque pointer (hipStream_t): every launch of this handle is ordered on it, so a host application can
 * record events on / wait for it instead of relying on the blocking entry points */
void* q3tts_stream(q3tts_engine* e);
/* Streaming / chunked decode (the reference decodes the whole utterance in one run_vocoder call, tts_onnx.cpp:430).
 * The decoder is causal: frames [a, b) own the samples [L(a), L(b)) of the full decode (L = q3tts_codec_decode_len, L(0) = 0), and
 * they are final as soon as frame b-1 exists.  With left_context >= a (exact mode) the call runs on a carried-state stream (below):
 * O(b - a) work, and the concatenation over chunks equals the whole-utterance decode.  A smaller left_context decodes the window
 * [a - left_context, b) instead and returns those samples: bounded memory of the past at the price of a truncated history (the
 * pre-transformer looks back 72 frames per layer, 568 in all). */
int q3tts_codec_decode_chunked_host(q3tts_engine* e, const int64_t* codes, int F, int chunk_frames, int left_context, float* pcm, int64_t cap,
                                    int64_t* out_len);
/* Streaming decode with CARRIED state (round 4): a stream keeps the pre-transformer's K / V rows of every layer and its output rows, so a
 * push of n new frames costs O(n + a few frames of conv look-back) instead of a decode of the history.  The concatenation of the pushes
 * equals q3tts_codec_decode_host of all the frames to fp32 rounding (<= 2e-5 asserted in tests/test_gpu_codec.py, ~1e-6 measured): the
 * arithmetic is the same, but a short push picks other GEMM tile shapes / split-K and, under 128 rows, k_attn instead of k_attn_win, so
 * sums associate differently.  max_frames bounds the stream's length (and sizes the shared RoPE tables), NOT its memory: the state is a
 * sliding buffer of the last window - 1 = 71 K / V rows per layer and the last 12 output rows plus room for the largest push so far
 * (round 5; 16.8 MB of K / V + 0.4 MB of rows per stream at 0.6B dims for pushes of up to 114 frames, whatever max_frames is; a larger push
 * grows it once).  A stream's buffers are kept when it ends and reused by the next stream; they are freed with the engine.
 * q3tts_codec_decode_chunked_host with left_context >= F and q3tts_slot_codec_decode_range_host with left_context >= frame_begin run on
 * such a stream by themselves (one per slot, created at the slot's first exact range). */
int q3tts_codec_stream_begin(q3tts_engine* e, int max_frames, int* stream_id);
/* codes[n_frames][n_groups] of the NEXT n_frames frames of the stream -> the samples those frames own (*out_len of them) */
int q3tts_codec_stream_push_host(q3tts_engine* e, int stream_id, const int64_t* codes, int n_frames, float* pcm, int64_t cap, int64_t* out_len);
int q3tts_codec_stream_end(q3tts_engine* e, int stream_id);
/* the same for frames of a slot that is still generating: call after q3tts_decode_steps has produced frame_end frames */
int q3tts_slot_codec_decode_range_host(q3tts_engine* e, int slot, int frame_begin, int frame_end, int left_context, float* pcm, int64_t cap,
                                       int64_t* out_len);
/* ---- the same sessions, batch-first, on DEVICE pointers ----
 * For a host application that keeps embeddings, logits and ids in HBM: no PCIe round trip per call.  Row b of a call is slot b of the
 * engine (batch <= max_batch).  Tensors (float / id buffers) are device pointers on the engine's GPU, any allocator; control arrays
 * (`lens`, `active_mask`) are host pointers.  `stream` is the caller's hipStream_t (NULL: the engine's own stream, q3tts_stream): the
 * engine's stream first waits for everything the caller has enqueued on it, and the caller's stream is made to wait for the call's
 * work — the call is ordered inside the caller's stream like a kernel launch.  The calls that track positions on the host
 * (prefill / decode / code_predictor) return after their launches have completed; q3tts_sample_dev returns at once.
 * Per-slot state is shared with the "_host" entry points and the fused generation (positions, KV cache, the armed logits row). */
/* run_prefill, tts_onnx.cpp:615-665: embeds[batch][S][hidden] (row block b: lens[b] <= S <= 16 rows; lens NULL = S for all) ->
 * logits_last[batch][vocab] (the last prompt row's, all the reference consumes, :797-798), last_hidden[batch][hidden]; either may be NULL.
 * Consecutive slots with equal lengths share one pass through the layers; a slot on its own takes q3tts_talker_prefill_host's launches. */
int q3tts_talker_prefill_dev(q3tts_engine* e, const float* embeds, int batch, int S, const int32_t* lens, float* logits_last, float* last_hidden, void* stream);
/* run_decode, tts_onnx.cpp:667-732: embeds[batch][hidden] -> logits[batch][vocab], last_hidden[batch][hidden]; one token appended to every
 * slot whose active_mask[b] != 0 (NULL: all).  Masked rows keep the batch's shape and leave outputs, position and slot state untouched. */
int q3tts_talker_decode_dev(q3tts_engine* e, const float* embeds, int batch, const uint8_t* active_mask, float* logits, float* last_hidden, void* stream);
/* predict_subcodes, tts_onnx.cpp:851-872, fused: last_hidden[batch][hidden] + code0[batch] (int64, as the reference holds ids) ->
 * sub[batch][n_groups - 1] int32: 15 KV-cached run_code_predictor passes (:734-757) with sample_token (:878-950) on device; row b draws
 * sub-code j with q3tts_rng_uniform(seed, stream_id0 + b, frame, j + 1), the fused generation loop's draw for that utterance and frame. */
int q3tts_code_predictor_dev(q3tts_engine* e, const float* last_hidden, const int64_t* code0, int batch, const q3tts_sampling* p, uint64_t seed,
                             uint32_t stream_id0, uint32_t frame, int32_t* sub, void* stream);
/* sample_token, tts_onnx.cpp:878-950, for a batch: logits[batch][n], u[batch] (uniforms in [0,1), one per row) -> ids[batch] int64 */
int q3tts_sample_dev(q3tts_engine* e, const float* logits, int batch, int n, const q3tts_sampling* p, const float* u, int suppress, int64_t* ids, void* stream);
/* sample_token, tts_onnx.cpp:878-905, on device; u in [0,1) replaces the mt19937 draw.
 * suppress != 0 applies the special-token suppression of tts_onnx.cpp:803-807 first. */
int q3tts_sample_host(q3tts_engine* e, const float* logits, int n, const q3tts_sampling* p, float u, int suppress, int64_t* token);
float q3tts_rng_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t group);

/* ---- host logic of the path, mirrored (build_prompt_embeddings, tts_onnx.cpp:442-539) ---- */
/* lang: 0 Auto, 1 English, 2 Chinese, 3 Japanese, 4 Korean.  prompt[<=16][hidden], *S rows;
 * trailing[cap_rows][hidden] receives trailing_text_hidden_, *n_trailing its row count. */
int q3tts_build_prompt_host(q3tts_engine* e, const int64_t* ids, int n_ids, int lang, const float* speaker,
                            float* prompt, int* S, float* trailing, int cap_rows, int* n_trailing);

/* ---- fused, batched generation (generate_codes + predict_subcodes, tts_onnx.cpp:782-872) ---- */
/* Admit an utterance into `slot`: uploads prompt + trailing rows, runs prefill, arms the slot.
 * stream_id selects the RNG stream; ignore_eos keeps EOS suppressed (fixed-length benchmark mode). */
int q3tts_slot_begin(q3tts_engine* e, int slot, const float* prompt, int S, const float* trailing, int n_trailing,
                     const q3tts_sampling* p, uint64_t seed, uint32_t stream_id, int ignore_eos);
/* Advance every armed slot by n_steps frames (one hipGraph replay per frame).  Returns the number
 * of slots still active, <0 on error. */
int q3tts_decode_steps(q3tts_engine* e, int n_steps);
int q3tts_slot_status(q3tts_engine* e, int slot, int* n_frames, int* finished);
/* codes[cap_frames][n_groups], int64 like the reference (tts_onnx.cpp:421-427) */
int q3tts_slot_codes_host(q3tts_engine* e, int slot, int64_t* codes, int cap_frames);
/* run_decode's outputs (tts_onnx.cpp:714-719) as the fused path holds them for the slot: logits[vocab] the slot's next code0 will be
 * sampled from and last_hidden[hidden] the code predictor's first input row (either may be NULL).  For teacher-forced parity checks
 * of the batched step; the data never leaves HBM in normal operation. */
int q3tts_slot_logits_host(q3tts_engine* e, int slot, float* logits, float* last_hidden);
/* vocoder over the slot's device-resident codes */
int q3tts_slot_codec_decode_host(q3tts_engine* e, int slot, float* pcm, int64_t cap, int64_t* out_len);
int q3tts_slot_release(q3tts_engine* e, int slot);

/* synthesize_tokens (tts_onnx.cpp:405-436) for a batch: utterance u has token ids
 * ids[offsets[u] .. offsets[u+1]).  pcm_out[u] receives up to pcm_cap samples, pcm_len[u] the
 * sample count, n_frames[u] the frames generated; codes_out (optional) [n_utt][max_new][n_groups]. */
int q3tts_synthesize_batch_host(q3tts_engine* e, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                const q3tts_sampling* p, uint64_t seed, int ignore_eos,
                                float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames,
                                int64_t* codes_out);

/* ---- voice-clone front end ---- */
/* synthesize_tokens with one speaker embedding per utterance spliced before CODEC_BOS (synthesize_clone,
 * tts_onnx.cpp:264-318; splice :481-498).  speakers[u] = [hidden] floats or NULL; speakers == NULL is
 * q3tts_synthesize_batch_host. */
int q3tts_synthesize_clone_batch_host(q3tts_engine* e, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                      const float* const* speakers, const q3tts_sampling* p, uint64_t seed, int ignore_eos,
                                      float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames,
                                      int64_t* codes_out);
/* The scheduler behind both: n_utt may exceed max_batch — utterances queue for the slots, a slot that finishes is re-armed with the
 * next one (continuous batching) while its codes are vocoded on a side stream.  max_new_per_utt (NULL: p->max_new_tokens for all) caps
 * each utterance separately: with ignore_eos it fixes ragged lengths for benchmarks.  Results do not depend on
 * the schedule (RNG stream = utterance index). */
int q3tts_synthesize_schedule_host(q3tts_engine* e, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                   const float* const* speakers, const q3tts_sampling* p, const int32_t* max_new_per_utt, uint64_t seed, int ignore_eos,
                                   float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames, int64_t* codes_out);
/* io::read_wav (src/io/wav_reader.h:13, wav_reader.cpp:28-143): mono float samples; -1 when the reference
 * returns an empty vector.  Call with out == NULL to learn *n_samples. */
int q3tts_read_wav_host(const char* path, float* out, int64_t cap, int64_t* n_samples, int32_t* sample_rate);
/* io::resample (wav_reader.cpp:145-164): linear interpolation; returns the output length */
int64_t q3tts_resample_host(const float* in, int64_t n, int32_t src_rate, int32_t dst_rate, float* out, int64_t cap);
/* MelExtractor::extract with the settings of tts_onnx.cpp:347-354 (24 kHz, n_fft = win = 1024, hop 256, 128 HTK
 * mels, 0-12 kHz, log power): mel[128][*frames].  Call with mel == NULL to learn *frames. */
int q3tts_mel_host(const float* audio, int64_t n, float* mel, int64_t cap, int32_t* frames);
/* has_speaker_encoder (tts_onnx.h:172) */
int q3tts_has_speaker_encoder(q3tts_engine* e);
/* run_speaker_encoder (tts_onnx.cpp:367-403): mel[128][frames] (MelExtractor layout) -> embed[spk_enc_dim], on the GPU */
int q3tts_speaker_encoder_host(q3tts_engine* e, const float* mel, int frames, float* embed);
/* extract_speaker_embedding (tts_onnx.cpp:331-365): wav -> 24 kHz -> mel -> speaker encoder */
int q3tts_extract_speaker_embedding_host(q3tts_engine* e, const char* wav_path, float* embed);

/* ---- text front end: the reference's byte-level BPE tokenizer ---- */
/* Replaces leaxer_qwen::io::load_vocab / load_merges / is_tokenizer_ready / tokenize (reference
 * src/io/tokenizer.h:13-22, src/io/tokenizer.cpp:538-561) with the same ids for the same files and text.
 * The reference keeps one process-global tokenizer; here it is a handle (host-only, no GPU work).
 * load_* return 0 on success, -1 on failure (the reference's `false`).  q3tts_tokenize writes up to
 * `cap` ids and returns the number of ids the text produces (call with cap=0 to size the buffer). */
typedef struct q3tts_tokenizer q3tts_tokenizer;
q3tts_tokenizer* q3tts_tokenizer_create(void);
void q3tts_tokenizer_destroy(q3tts_tokenizer* t);
int q3tts_tokenizer_load_vocab(q3tts_tokenizer* t, const char* vocab_json_path);
int q3tts_tokenizer_load_merges(q3tts_tokenizer* t, const char* merges_txt_path);
int q3tts_tokenizer_ready(const q3tts_tokenizer* t);
int64_t q3tts_tokenize(const q3tts_tokenizer* t, const char* text, int64_t len, int32_t* ids, int64_t cap);

/* ---- measurement hooks (bench.py) ---- */
/* device time in ms of the last q3tts_decode_steps call, from HIP events on the engine's stream */
int q3tts_last_decode_ms(q3tts_engine* e, float* ms, int* steps);
/* device time in ms of the last codec decode */
int q3tts_last_codec_ms(q3tts_engine* e, float* ms);
/* accumulated device time since the last reset: decode steps (HIP events around the graph
 * replays, on the engine's stream) and codec decodes */
int q3tts_counters(q3tts_engine* e, double* decode_ms, int64_t* decode_steps, double* codec_ms, int64_t* codec_frames, int reset);
/* Codec decoder (run_vocoder's graph, /root/reference/src/tts_onnx.cpp:759-776): how many of its conv / linear weight tensors are exact
 * in fp16 after the power-of-two pre-scale (every bf16- or fp16-origin tensor) and therefore run two matrix-core products per fp32
 * product, and how many keep a non-zero lo plane and run three.  Both 0 under Q3TTS_FLAG_FP32_CODEC. */
int q3tts_codec_plane_stats(q3tts_engine* e, int* two_product, int* three_product);
/* Per-stage device time of the decode step: runs n_steps EAGER steps of the armed slots (they advance like q3tts_decode_steps) with HIP
 * events at the stage boundaries.  out_ms[0] sampler (n_groups launches), [1] code predictor (layer passes + heads; predict_subcodes,
 * tts_onnx.cpp:851-872), [2] talker decode (layers + codec head; run_decode :667-732), [3] their sum — milliseconds per step. */
int q3tts_stage_profile(q3tts_engine* e, int n_steps, double* out_ms /* [4] */);
/* Device time of the prefill stage (run_prefill, tts_onnx.cpp:615-665): `reps` batched prefill passes of slots 0..n_slots-1 (all free)
 * over n_rows synthetic prompt rows each, already in HBM — the launches a job's equal-length prompts take (groups of up to 128 rows
 * share one pass over the talker's weights) — with HIP events around each pass; *ms_per_pass = mean device milliseconds.  The slots
 * are released again.  bench.py's stages.prefill. */
int q3tts_prefill_profile(q3tts_engine* e, int n_slots, int n_rows, int reps, double* ms_per_pass);
/* Measurement aid (engines created with Q3TTS_FLAG_TEST_HOOKS only): every armed slot jumps n_frames ahead without generating them — frame
 * counters and talker positions advance, the skipped frames' codes are zero and the talker's KV cache is refilled with seeded synthetic rows.
 * What the slots emit afterwards is numerically meaningless; the decode step streams a context of the requested depth, which is what the
 * rocprofv3 passes over run_decode's attention (tts_onnx.cpp:667-732) at 1000-2000 tokens of context need (tools/ctx_bench.py). */
int q3tts_measure_skip_frames(q3tts_engine* e, int n_frames);
/* Test hook (engines created with Q3TTS_FLAG_TEST_HOOKS only): fills the vocoder's reusable workspace (lane arenas, batched-front arena,
 * streaming arena, pinned PCM staging, the job's code rows) with 0xFF bytes = NaN, so that a later decode which reads anything it did
 * not write itself produces NaN instead of plausible stale samples (tests/test_gpu_codec_stress.py). */
int q3tts_test_poison_workspace(q3tts_engine* e);
/* Test hook (Q3TTS_FLAG_TEST_HOOKS engines): the last batched vocoder group's final conv as it lies in its lane's workspace — input rows
 * sx_out[nb][T][C] (cap_floats >= nb*T*C, else skipped) and output pcm_out[nb][T]; either may be NULL.  tools/vocoder_stress.py uses it to
 * tell a wrong input from a wrong conv when a job's PCM differs from the utterance's own decode. */
/* ... and, when the A/B knob Q3TTS_COUT1_PACKED=2 selected the dumping variant of that conv, the per-row per-tap partial sums each tile's
 * output phase read from LDS: out[tile][256][8]; returns the float count (call with out == NULL to size). */
int64_t q3tts_test_final_conv_partials(q3tts_engine* e, float* out, int64_t cap_floats);
int q3tts_test_group_final_conv(q3tts_engine* e, float* sx_out, float* pcm_out, int64_t cap_floats, int32_t* T, int32_t* C, int32_t* nb);
/* Parity aid: ONE eager decode step of the armed slots (they advance like q3tts_decode_steps(1)) that also returns, for `slot`, the
 * logits row each of the frame's n_groups decisions was sampled from — out[n_groups][cols], cols >= max(vocab, sub_vocab); row 0 the
 * code0 logits (run_decode's output, before suppression), row j the code predictor's logits for sub-code j-1 (run_code_predictor,
 * tts_onnx.cpp:734-757).  Lets a test compare every head of the fused step with the oracle at a chosen frame. */
int q3tts_step_logits_host(q3tts_engine* e, int slot, float* out, int cols);
/* algorithmic bytes one decode step streams (weights + KV at the slots' current contexts) */
int q3tts_decode_step_bytes(q3tts_engine* e, double* weight_bytes, double* kv_bytes);

#ifdef __cplusplus
}
#endif
#endif
