// q3_engine.h — the device-resident engine behind the C-ABI (include/q3tts.h).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "q3_common.h"
#include "q3_kvpool.h"

namespace q3 {

enum TensorKind { TK_W = 0, TK_NORM = 1, TK_BIAS = 2, TK_SCALE = 3, TK_SNAKE = 4 };

struct Tensor {
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0;
    int kind = TK_W;
    bool bf16 = false;  // storage: bf16 for talker / predictor / text matrices, fp32 otherwise
    void* dev = nullptr; // may point into a fused parent allocation (qkv)
    int64_t numel = 0;
    float synth_std = 0.02f;
};

struct TensorSpec { // registry entry without storage
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0, kind = TK_W;
    bool bf16 = false;
    int fuse = 0;           // 1/2/3: q/k/v of a layer, allocated as one block
    float synth_std = 0.02f;
};
std::vector<TensorSpec> tensor_specs(const q3tts_config& c); // host-only

struct DecLayerW { // one decoder layer, bf16 matrices
    const float *in_norm = nullptr, *post_norm = nullptr, *q_norm = nullptr, *k_norm = nullptr;
    const bf16_t *qkv = nullptr, *o = nullptr, *gate = nullptr, *up = nullptr, *down = nullptr;
};
struct DecStack {
    int H = 0, L = 0, nq = 0, nkv = 0, d = 0, ffn = 0;
    float eps = 0.f;
    std::vector<DecLayerW> layers;
    float *kc = nullptr, *vc = nullptr; // paged cache (fp32, or bf16 behind the same pointers when kv_bf16)
    bool kv_bf16 = false, kv_round = false;
    int* page_table = nullptr;
    int pages_per_slot = 0, page_shift = 0;
    bool identity_pages = false;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    bool nt = false; // weights streamed once per step -> non-temporal loads
    int n_splits = 1, chunk = 1 << 30; // split-T attention
    int n_splits_stream = 0, chunk_stream = 0; // k_attn_stream's splits (batched step, long contexts): whole pages, 256 tokens by default
    float *po = nullptr, *pm = nullptr, *pl = nullptr; // attention partials [rows][nq][n_splits]([d])
};

struct CodecW; // q3_codec.cpp
struct SpeakerW; // q3_speaker.cpp

class Engine {
public:
    // kv_pool_tokens: capacity of the talker's KV page pool in tokens (0: max_batch x max_ctx, every slot can reach max_ctx at once)
    Engine(const q3tts_config& cfg, int device, int max_batch, int max_ctx, uint32_t flags, int64_t kv_pool_tokens = 0);
    ~Engine();

    // ---- talker KV page pool: 64-token pages handed to slots on demand (page 0 is a scratch page every unowned table entry points at,
    // so masked rows of unarmed / released slots keep writing somewhere harmless).  Host-side free list; the device sees only the table.
    int kv_total_pages() const { return kv.total; }
    int kv_free_pages() const { return kv.free_count; }
    int kv_pages_for(int tokens) const { return kv.pages_for(tokens); }
    int kv_slot_pages(int slot) const { return kv.slot_pages(slot); }
    void kv_reserve(int slot, int tokens, bool exact);   // the slot owns pages for positions [0, tokens): grows (and with `exact` shrinks) to that
    void kv_upload_row(int slot);                        // after KvPool changed a slot's table row behind the engine's back (scheduler policy)
    void kv_release(int slot);

    KnobScope knob_scope;                      // first member: counts this engine among the hook-enabled ones until it is destroyed (also when the constructor throws)
    q3tts_config c;
    int device, B, max_ctx;
    KvPool kv;                                 // q3_kvpool.h: free list, per-slot pages, host mirror of talker.page_table
    int64_t sched_admitted = 0, sched_preempted = 0; int sched_peak_live = 0;   // the last scheduler call (q3tts_sched_stats)
    uint32_t flags;
    hipStream_t stream = nullptr;
    bool null_stream = false;
    std::string err;

    // ---- weights ----
    std::vector<Tensor> tensors;
    std::unordered_map<std::string, int> tindex;
    std::vector<void*> allocs;
    bool finalized = false;
    Tensor& T(const std::string& n);
    void set_tensor(const std::string& name, const float* data, int64_t n);
    void get_tensor(const std::string& name, float* out, int64_t n);
    void fill_synthetic(uint64_t seed);
    void finalize();

    // ---- session-shaped ops (host I/O) ----
    void text_project(const int64_t* ids, int n, float* out);
    void codec_embed(const int64_t* ids, int n, float* out);
    void cp_embed(int64_t id, int step, float* out);
    void talker_prefill(int slot, const float* embeds, int S, float* logits, float* last_hidden);
    void prefill_rows_in_xp(int slot, int S);   // run_prefill's device work for the S rows in xp (shared by the host and the device-pointer entry)
    void talker_decode(int slot, const float* embed, float* logits, float* last_hidden);
    void code_predictor(const float* seq, int n, int step, float* logits);
    void sample(const float* logits, int n, const q3tts_sampling& p, float u, int suppress, int64_t* tok);
    void build_prompt(const int64_t* ids, int n_ids, int lang, const float* speaker, float* prompt, int* S,
                      float* trailing, int cap_rows, int* n_trailing);
    void build_prompts(const int64_t* ids, const int32_t* offsets, int n_utt, int lang, const float* const* speakers,
                       float* prompts, int* S_out, float* trailing, const size_t* toff, int* nt_out);
    int64_t* proj_ids_d = nullptr; float* proj_out_d = nullptr; size_t proj_cap = 0;   // text_project staging (grow-only)
    int64_t codec_decode_host(const int64_t* codes, int F, float* pcm, int64_t cap);
    int64_t codec_decode_dev(const int32_t* codes_dev, int F, float* pcm_dev, int64_t cap);
    // exact chunked / streaming decode: samples owned by frames [a, b), decoded from the window [a - left_context, b)
    int64_t codec_decode_range_dev(const int32_t* codes_dev, int a, int b, int left_context, float* pcm, int64_t cap);
    int64_t slot_codec_decode_range(int slot, int a, int b, int left_context, float* pcm, int64_t cap);
    int64_t codec_decode_chunked_host(const int64_t* codes, int F, int chunk, int left_context, float* pcm, int64_t cap);
    // streaming decode with carried state (q3_codec.cpp): a stream keeps the pre-transformer's K / V rows and output rows, a push decodes
    // n new frames in O(n + stage_b_context) work, exactly
    void codec_debug_group(float* sx_out, float* pcm_out, int64_t cap_floats, int* T, int* C, int* nb);
    int64_t codec_debug_partials(float* out, int64_t cap_floats);
    void codec_poison();                                // test hook: the vocoder's reusable workspace filled with NaN bytes
    int codec_stream_begin(int max_frames);
    void codec_stream_fit(int sid, int n);              // room for a push of n frames in the stream's sliding K / V and row buffers
    int64_t codec_stream_push_dev(int sid, const int32_t* codes_dev, int n, float** pcm_dev);
    int64_t codec_stream_push_host(int sid, const int64_t* codes, int n, float* pcm, int64_t cap);
    void codec_stream_end(int sid);
    int codec_stream_frames(int sid) const;
    int codec_stage_b_context() const;          // frames the stages behind the pre-transformer look back
    int64_t slot_codec_stream_range(int slot, int a, int b, float* pcm, int64_t cap);
    void slot_codec_stream_reset(int slot);
    void codec_rope_tables(int P);

    // ---- batch-first session ops on DEVICE pointers (include/q3tts.h "_dev" entry points): row b <-> slot b ----
    // caller stream: the engine's stream first waits for what the caller has enqueued, the caller's stream then waits for the call's work
    void stream_join(hipStream_t caller);
    void stream_fork(hipStream_t caller);
    void talker_prefill_dev(const float* embeds, int nb, int S, const int32_t* lens, float* logits_last, float* last_hidden);
    void talker_decode_dev(const float* embeds, int nb, const uint8_t* active, float* logits, float* last_hidden);
    void code_predictor_dev(const float* last_hidden, const int64_t* code0, int nb, const q3tts_sampling& p, uint64_t seed, uint32_t stream0,
                            uint32_t frame, int32_t* sub);
    void sample_dev(const float* logits, int nb, int V, const q3tts_sampling& p, const float* u, int suppress, int64_t* ids);
    void dev_scratch(int nb);                  // lazily allocated workspaces of the four calls above
    float* dev_logits_d = nullptr; int* dev_flags_d = nullptr; int32_t* dev_pos_d = nullptr; int32_t* dev_pos_dummy_d = nullptr;
    SlotState* dev_st_d = nullptr; int32_t* dev_codes_d = nullptr;
    hipEvent_t ev_join = nullptr, ev_fork = nullptr;
    void predictor_passes(int nb, const SampleArgs& s0, bool sp0, bool spn, const std::function<void()>& mark);

    // ---- fused generation ----
    struct SlotInit { int slot = 0; const float* prompt = nullptr; int S = 0; const float* trailing = nullptr; int n_trailing = 0; uint32_t stream_id = 0;
                      int max_frames = 0; /* 0: the call's max_new_tokens */
                      int kv_tokens = 0;  /* KV pages reserved now, in tokens; 0: prompt + max_frames (the slot never needs more) */ };
    void slots_begin(const SlotInit* in, int n, const q3tts_sampling& p, uint64_t seed, int ignore_eos); // batched prefill of equal-length prompts
    void slot_begin(int slot, const float* prompt, int S, const float* trailing, int n_trailing,
                    const q3tts_sampling& p, uint64_t seed, uint32_t stream_id, int ignore_eos);
    int decode_steps(int n_steps);
    void slot_status(int slot, int* n_frames, int* finished);
    void slot_codes(int slot, int64_t* codes, int cap_frames);
    void slot_logits(int slot, float* logits, float* last_hidden);
    int64_t slot_codec_decode(int slot, float* pcm, int64_t cap);
    void slot_release(int slot);
    void step_bytes(double* wbytes, double* kvbytes);
    void measure_skip_frames(int n);                    // measurement aid: armed slots jump n frames ahead over a synthetic KV cache
    void pack_mfma_weights();                           // fragment-packed copies of the projection matrices for k_gemv16 / k_gemm3 (finalize)
    void free_packed_weights();
    std::vector<std::pair<const bf16_t*, bf16_t*>> packed_w;
    void prefill_profile(int nb, int S, int reps, double* ms_per_pass);   // device time of a batched prefill pass (diagnostic)
    void stage_profile(int n_steps, double* out_ms4);   // eager steps with events at the stage boundaries (diagnostic)
    // one EAGER step of the armed slots that also keeps, for slot `slot`, the logits row every one of the frame's n_groups decisions was
    // sampled from (out: [n_groups][cols], cols >= max(vocab, sub_vocab)); the slots advance like decode_steps(1)
    void step_logits(int slot, float* out, int cols);
    float* trace_d = nullptr; int trace_slot = 0, trace_cols = 0;
    std::vector<hipEvent_t> stage_ev;

    float last_decode_ms = 0.f;
    int last_decode_steps = 0;
    float last_codec_ms = 0.f;
    double total_decode_ms = 0.0, total_codec_ms = 0.0;
    int64_t total_decode_steps = 0, total_codec_frames = 0;

    // ---- codec decoder (q3_codec.cpp) ----
    CodecW* codec = nullptr;
    void codec_finalize();
    void codec_plane_stats(int* two_product, int* three_product) const;   // weight tensors on the 2-product (lo plane empty) / 3-product split path
    // returns the sample count; h_in: this utterance's rows from codec_pre_batch (h_stage 1: after the pre-transformer, 2: after the upsampling stages too)
    int64_t codec_run(const int32_t* codes_dev, int F, float** pcm_dev, int lane = 0, const float* h_in = nullptr, int h_stage = 1, int nbatch = 1, size_t h_ustride = 0);
    void codec_async_submit_group(const float* h_group, size_t h_ustride, int Fg, int g, const int* nf, float* const* user_pcm, int64_t cap, int64_t* const* len_out);
    bool codec_batchable() const;   // every conv of the decoder takes the split-precision path (the batched kernels)
    const float* codec_pre_batch(const int32_t* codes_dev, int codes_stride_frames, int n, int Fp, bool with_upsampling, int* rows_per_utt_out,
                                 const int* perm_host = nullptr);
    // vocoder side of the scheduler: stash a finished slot's codes, vocode the job's utterances over the side lanes at the end
    void codec_async_prepare(int max_frames, int n_utt);
    const int32_t* codec_stash(int slot, int nf, int utt, int row_frames);
    const int32_t* codec_job_codes(int utt, int row_frames);
    void codec_job_upload(const int32_t* host, int n_utt, int row_frames);
    void codec_async_submit_dev(const int32_t* codes_dev, int nf, float* user_pcm, int64_t cap, int64_t* len_out, const float* h_in = nullptr, int h_stage = 1);
    void codec_async_drain_lane(int lane);
    void codec_async_drain();
    void codec_async_abort();   // drop every pending vocoder result (error path: their host pointers belong to a failed job)
    void codec_lanes_join();
    void slots_state(int nb, std::vector<SlotState>& out);
    void codec_free();

    // ---- speaker encoder of the clone path (q3_speaker.cpp) ----
    SpeakerW* spk = nullptr;
    bool has_speaker() const { return c.spk_enc_dim > 0; }
    void speaker_finalize();
    void speaker_free();
    // mel [spk_mel][frames] (the reference MelExtractor layout) on the host -> embedding [spk_enc_dim] on the host
    void speaker_encode(const float* mel, int frames, float* out);

    // ---- internals ----
    DecStack talker, cp;
    const float* talker_norm = nullptr; const bf16_t* codec_head = nullptr; const bf16_t* codec_embed_w = nullptr;
    const bf16_t *text_embed = nullptr, *fc1_w = nullptr, *fc2_w = nullptr; const float *fc1_b = nullptr, *fc2_b = nullptr;
    const float* cp_norm = nullptr;
    const bf16_t* cp_proj_w = nullptr; const float* cp_proj_b = nullptr; // talker width -> predictor width (1.7B), null when equal
    float* x_cpp = nullptr;                                               // projected predictor input rows [<= rows_max][cp_width]
    int cp_width() const { return c.cp_hidden > 0 ? c.cp_hidden : c.hidden; }
    bool cp_projected() const { return cp_width() != c.hidden; }
    float* cp_project(float* rows, int ld, int M);
    std::vector<const bf16_t*> cp_head, cp_embed_w;

    int rows_max = 0, max_trailing = 0, max_frames_cap = 0;
    // rows from which a projection takes the split-K slab GEMM (k_gemm3 with the in-launch seam, 5 launches per layer; k_gemm2 + finish
    // kernels, 8 launches, where the seam does not apply).  Below it the GEMV-family contract holds (5 launches per layer): 1-2 rows
    // single-pass GEMV, 3..11 rows k_gemv16 on the matrix cores.  Crossover measured with the seam (round 3): b=8 3.76 (gemv16) vs 3.87 ms,
    // b=12 3.99 vs 3.89, b=16 4.22 vs 3.93; it was 17 rows with the finish launches.
    int mfma_min_rows = 12;
    float *x_talk = nullptr, *qkv = nullptr, *attn = nullptr, *act = nullptr, *logits_t = nullptr, *logits_cp = nullptr;
    float *x_cp = nullptr, *x_cp1 = nullptr, *sum = nullptr, *xp = nullptr, *hn = nullptr, *logits_p = nullptr;
    float *trailing_d = nullptr, *tts_pad_d = nullptr, *text_tmp = nullptr, *text_tmp2 = nullptr;
    int64_t* ids_d = nullptr;
    bf16_t *pl0h = nullptr, *pl0l = nullptr, *pl1h = nullptr, *pl1l = nullptr; // (hi, lo) activation planes for the MFMA GEMM path
    int ldp = 0;
    float* slab_d = nullptr; // split-K partial sums [ks][rows][H]
    float* gu_slab_d = nullptr;  // gate | up split-K partial sums, 2 x [<=4][rows][ffn]
    float* cp_logit_slab_d = nullptr; // split-K partial sums of the batched predictor heads [4][B][sub_vocab]: the sampler sums them
    float* qkv_slab_d = nullptr; // split-K partial sums of the QKV projection [<=4][rows][QKV]
    // split-K seam of the batched step (GemmArgs::seam): arrival / claim counters, one region per seam launch of the step (generation-valued words:
    // never reset), and the per-(row, 64-column tile) sums of squares behind the two residual seams of a layer
    unsigned* seam_cnt_d = nullptr; size_t seam_cnt_words = 0, seam_cnt_used = 0;
    unsigned* seam_gen_d = nullptr;   // the step generation the flag words carry (bumped by the step's first sampler launch)
    float *ssq_a_d = nullptr, *ssq_b_d = nullptr;
    bool seam_step = false;      // inside record_step: run_layers may fold the finish launches into the GEMMs
    bool seam_on = true;         // Q3TTS_SEAM=0 at engine creation keeps the finish launches (the A/B knob and the tests' second path)
    bool attn_stream = true;         // Q3TTS_ATTN_STREAM=0 at engine creation: the batched step's long-context attention stays on k_attn (A/B knob, tests' second path)
    bool attn_stream_one = true;     // Q3TTS_ATTN_STREAM_ONE=0: the one-split case (contexts <= 512 at >= 256 (row, kv head) pairs) back on k_attn (A/B knob): b=64 x 256 frames 4.67 -> 4.59 ms per step with it
    bool attn_keep_splits = false;   // Q3TTS_ATTN_KEEP_SPLITS at engine creation: the batched step keeps split-T attention + the combine launch (A/B knob, tests' second path)
    int seam_spin = 512;         // Q3TTS_SEAM_SPIN: polls (~0.7 us each) before an owner abandons its chunk to whoever sees the tile complete (1 forces that rescue path in the tests).  A fast-path seam completes within a few polls; the bound only matters when the launch's workgroups are not co-resident (vocoder lanes, other engines, a CU mask): 512 keeps a stalled owner off its CU after ~0.35 ms instead of round 3's ~2.9 ms
    int32_t* codes_d = nullptr;
    int32_t* codes_scratch_d = nullptr;
    int32_t* talker_pos_d = nullptr;
    int* slot_map_d = nullptr;      // [128] slots of a batched prefill over scattered slots
    float* logits_g = nullptr;      // [128][vocab] its head output before the scatter
    SlotState* st_d = nullptr;
    std::vector<SlotState> st_h;
    int32_t* active_d = nullptr;
    int32_t* active_h = nullptr; // pinned
    int64_t* tok_d = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::unordered_map<int, hipGraphExec_t> graphs; // keyed by nb

    void* dmalloc(size_t bytes);
    // final_gamma != null (MFMA path only): the last layer's finish kernel also applies the stack's final RMSNorm,
    // leaving (hi, lo) planes of the normalised rows in pl0 (+ fp32 rows in final_xn); returns true in that case
    bool run_layers(const DecStack& W, float* x, int ldx, int nb, int n_new, int slot_offset, const int* pos_dev, int pos_scalar,
                    const float* final_gamma = nullptr, float final_eps = 0.f, float* final_xn = nullptr, int final_ld_xn = 0,
                    const int* slot_map = nullptr);   // slot_map (device, nb ints): row group bi belongs to slot slot_map[bi]
    void record_step(int nb);
    bool seam_applies(const DecStack& W, int M, float* x, int ldx, bool has_slot_map) const;   // run_layers folds the finish launches of this stack pass into its GEMMs
    bool planes_in_ready = false;   // record_step -> run_layers: the sampler already wrote planes0 (gamma0 * rows) + their sums of squares (ssq_b_d)
    // returns the number of split-K slabs `out` was written as (1: plain rows).  slab_out non-null: the caller's consumer can sum slabs
    // ([nslab][M][ldo] at slab_out), which lets a 17..128-row head split K over 4x the workgroups
    int head_proj(const bf16_t* Wm, const float* x, int ldx, const float* gamma, float eps, float* xn_out, int ld_xn,
                  float* out, int ldo, int M, int N, int K, bool nt, bool planes_ready = false, int plane_row0 = 0, int plane_row_stride = 1,
                  float* slab_out = nullptr);
    int nb_in_use() const;
    void sync();
};

} // namespace q3
