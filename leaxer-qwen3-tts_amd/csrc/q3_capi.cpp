// q3_capi.cpp — extern "C" surface declared in include/q3tts.h.  No exception crosses the ABI.
#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "q3_audio.h"
#include "q3_bpe.h"
#include "q3_engine.h"

using q3::Engine;

struct q3tts_engine {
    Engine* e = nullptr;
    std::string err;
};
static std::string g_create_err;

#define Q3_API_BEGIN(h)                                       \
    if (!(h) || !(h)->e) return -1;                           \
    try {                                                     \
        (void)hipSetDevice((h)->e->device);
#define Q3_API_END(h)                                         \
    }                                                         \
    catch (const q3::Error& ex) { (h)->err = ex.msg; return -1; } \
    catch (const std::exception& ex) { (h)->err = ex.what(); return -1; } \
    catch (...) { (h)->err = "unknown error"; return -1; }

// The vocoder phase of a job: utterance u's codes are row u of the engine's job buffer (Engine::codec_job_codes, stride row_frames frames),
// got_frames[u] of them valid.  Shared by the scheduler (below) and q3tts_codec_decode_batch_host.
static void vocoder_job(q3::Engine& e, int n_utt, const std::vector<int32_t>& got_frames, int row_frames, float* const* pcm_out, int64_t pcm_cap,
                        int64_t* pcm_len) {
    // Vocoder, once the decode queue is empty.  Utterances are taken longest first, in blocks of similar length (the shortest at least half
    // the longest, at most 2^16 padded frames): a block's pre-transformer and upsampling stages run as one batched pass, its conv decoder in
    // groups of up to 32 utterances per set of launches (about 4.7 MB of workspace per frame: groups sized to ~8 GB), each group padded to its
    // own longest member.  Padding is exact: every layer is causal.  Single utterances, the exact-fp32 codec and configs whose decoder the
    // batched kernels do not cover go one utterance at a time over the side lanes.
    // A failure in here (arena / pinned-buffer allocation, a launch error) must not leave lanes holding this job's pcm_out / pcm_len
    // pointers: the next job's drain would write through them.  codec_async_abort() waits for the lanes and forgets their items.
    try {
    std::vector<int> order((size_t)n_utt);
    for (int u = 0; u < n_utt; ++u) { order[(size_t)u] = u; if (pcm_len) pcm_len[u] = 0; }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return got_frames[(size_t)x] > got_frames[(size_t)y]; });
    const bool batchable = e.codec_batchable();
    std::vector<int> nf;
    std::vector<float*> up;
    std::vector<int64_t*> lp;
    for (int y0 = 0; y0 < n_utt;) {
        const int F0 = got_frames[(size_t)order[(size_t)y0]];
        if (F0 <= 0) break;                                            // sorted: the rest of the job produced no frame
        int y1 = y0 + 1;
        // a block is a batch dimension of the batched kernels (gridDim.y / .z <= 65535): at most 4096 sequences and 2^16 padded frames, and
        // no activation matrix of its batched front may reach 4 GB (k_conv_split addresses rows with 32-bit byte offsets): the widest is
        // the ConvNeXt hidden [frames x upsampling][4 x cd_hidden] — at 0.6B dims exactly 4 GB at 2^16 frames, hence the strict bound
        int64_t up_front = 1;
        for (int i = 0; i < e.c.cd_n_up; ++i) up_front *= e.c.cd_up_ratios[i];
        const int64_t widest = std::max<int64_t>((int64_t)4 * e.c.cd_hidden * up_front, std::max<int64_t>(e.c.cd_ffn, (int64_t)3 * e.c.cd_hidden));
        const int64_t frame_cap = std::min<int64_t>((int64_t)1 << 16, (((int64_t)1 << 32) - 1) / ((int64_t)sizeof(float) * widest));
        while (y1 < n_utt && y1 - y0 < 4096 && (int64_t)(y1 - y0 + 1) * F0 <= frame_cap && got_frames[(size_t)order[(size_t)y1]] * 2 >= F0) ++y1;
        const int nblk = y1 - y0;
        if (nblk >= 2 && batchable && F0 <= frame_cap / 2) {
            int rows = 0;
            e.codec_lanes_join();                                      // the previous block's groups still read the batched buffers
            const float* hb = e.codec_pre_batch(e.codec_job_codes(0, row_frames), row_frames, nblk, F0, true, &rows, order.data() + y0);
            const size_t ustride = (size_t)rows * e.c.cd_hidden;
            const int group = (int)std::max<int64_t>(1, std::min<int64_t>(32, ((int64_t)8 << 30) / ((int64_t)4700000 * F0)));
            for (int g0 = 0; g0 < nblk; g0 += group) {
                const int g = std::min(group, nblk - g0);
                nf.assign((size_t)g, 0); up.assign((size_t)g, nullptr); lp.assign((size_t)g, nullptr);
                for (int k = 0; k < g; ++k) {
                    const int u = order[(size_t)(y0 + g0 + k)];
                    nf[(size_t)k] = got_frames[(size_t)u];
                    if (pcm_len) lp[(size_t)k] = pcm_len + u;
                    if (pcm_out) up[(size_t)k] = pcm_out[u];
                }
                e.codec_async_submit_group(hb + (size_t)g0 * ustride, ustride, nf[0], g, nf.data(), up.data(), pcm_cap, lp.data());
            }
        } else {
            for (int y = y0; y < y1; ++y) {
                const int u = order[(size_t)y];
                e.codec_async_submit_dev(e.codec_job_codes(u, row_frames), got_frames[(size_t)u], pcm_out ? pcm_out[u] : nullptr, pcm_cap, pcm_len ? pcm_len + u : nullptr);
            }
        }
        y0 = y1;
    }
    e.codec_async_drain();
    } catch (...) {
        e.codec_async_abort();
        for (int u = 0; u < n_utt; ++u) if (pcm_len) pcm_len[u] = 0;
        throw;
    }
}

extern "C" {

int q3tts_default_config(const char* name, q3tts_config* o) {
    if (!o || !name) return -1;
    const bool big = strcmp(name, "1.7b") == 0 || strcmp(name, "1.7B") == 0;
    if (!big && strcmp(name, "0.6b") != 0 && strcmp(name, "0.6B") != 0) return -1;
    memset(o, 0, sizeof *o);
    // talker dims: reference src/tts_onnx.h:31-37; the rest [HINT]
    o->hidden = 1024; o->n_layers = 28; o->n_heads = 16; o->n_kv_heads = 8; o->head_dim = 128; o->ffn = 3072; o->vocab = 3072;
    o->rope_theta = 1e6f; o->rms_eps = 1e-6f;
    o->cp_layers = 5; o->cp_heads = 16; o->cp_kv_heads = 8; o->cp_head_dim = 128; o->cp_ffn = 3072; o->n_groups = 16; o->sub_vocab = 2048;
    o->cp_rope_theta = 1e6f; o->cp_rms_eps = 1e-6f;
    o->text_vocab = 151936; o->text_hidden = 2048;
    o->cd_codebook = 2048; o->cd_hidden = 1024; o->cd_layers = 8; o->cd_heads = 16; o->cd_head_dim = 64; o->cd_ffn = 3072; o->cd_window = 72;
    o->cd_rope_theta = 10000.0f; o->cd_rms_eps = 1e-5f;
    o->cd_n_up = 2; o->cd_up_ratios[0] = 2; o->cd_up_ratios[1] = 2;
    o->cd_decoder_dim = 1536; o->cd_n_blocks = 4;
    o->cd_up_rates[0] = 8; o->cd_up_rates[1] = 5; o->cd_up_rates[2] = 4; o->cd_up_rates[3] = 3;
    o->cd_tconv_trim = 0;
    o->codec_eos = 2150; o->suppress_begin = 2048; o->suppress_end = 3072;
    // speaker encoder of the Base checkpoints [HINT: Qwen3-TTS speaker_encoder_config]: ECAPA-TDNN 512/1536 channels -> hidden
    o->spk_enc_dim = 1024; o->spk_mel = 128; o->spk_channels = 512; o->spk_scale = 8; o->spk_se = 128; o->spk_att = 128;
    if (big) { // [HINT: the public 1.7B checkpoints] talker twice as wide, predictor at the 0.6B width behind cp.proj, speaker row talker-wide.
               // Beyond the reference: README.md:125 lists 1.7B as planned and tts_onnx.h:31-37 hard-codes the 0.6B dims.
        o->hidden = 2048; o->ffn = 6144; o->cp_hidden = 1024; o->spk_enc_dim = 2048;
    }
    return 0;
}

int q3tts_config_num_tensors(const q3tts_config* cfg) {
    if (!cfg) return -1;
    try { return (int)q3::tensor_specs(*cfg).size(); } catch (...) { return -1; }
}
int q3tts_config_tensor_info(const q3tts_config* cfg, int index, char* name, int name_cap, int64_t* shape4, int* ndim, int* kind) {
    if (!cfg) return -1;
    try {
        const std::vector<q3::TensorSpec> specs = q3::tensor_specs(*cfg);
        if (index < 0 || index >= (int)specs.size()) return -1;
        const q3::TensorSpec& t = specs[index];
        if (name && name_cap > 0) { snprintf(name, (size_t)name_cap, "%s", t.name.c_str()); }
        if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
        if (ndim) *ndim = t.ndim;
        if (kind) *kind = t.kind;
        return 0;
    } catch (...) { return -1; }
}

q3tts_engine* q3tts_create(const q3tts_config* cfg, int device, int max_batch, int max_ctx, uint32_t flags) {
    return q3tts_create_pooled(cfg, device, max_batch, max_ctx, 0, flags);
}

q3tts_engine* q3tts_create_pooled(const q3tts_config* cfg, int device, int max_batch, int max_ctx, int64_t kv_pool_tokens, uint32_t flags) {
    if (!cfg) { g_create_err = "null config"; return nullptr; }
    try {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw q3::Error("no HIP device: libq3tts_hip needs an MI355X (gfx950); there is no CPU fallback");
        if (device < 0 || device >= n) throw q3::Error("device index out of range");
        if (const char* ng = getenv("Q3TTS_NO_GRAPH")) if (ng[0] == '1') flags |= Q3TTS_FLAG_NO_GRAPH; // profiling aid
        q3tts_engine* h = new q3tts_engine;
        h->e = new Engine(*cfg, device, max_batch, max_ctx, flags, kv_pool_tokens);
        return h;
    } catch (const q3::Error& ex) { g_create_err = ex.msg; }
    catch (const std::exception& ex) { g_create_err = ex.what(); }
    catch (...) { g_create_err = "unknown error"; }
    return nullptr;
}

void q3tts_destroy(q3tts_engine* h) {
    if (!h) return;
    delete h->e;
    delete h;
}

const char* q3tts_last_error(q3tts_engine* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int q3tts_num_tensors(q3tts_engine* h) { return h && h->e ? (int)h->e->tensors.size() : -1; }

int q3tts_tensor_info(q3tts_engine* h, int i, char* name, int cap, int64_t* shape4, int* ndim) {
    Q3_API_BEGIN(h)
    if (i < 0 || i >= (int)h->e->tensors.size()) throw q3::Error("tensor index out of range");
    const q3::Tensor& t = h->e->tensors[i];
    if (name && cap > 0) { strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
    if (ndim) *ndim = t.ndim;
    return 0;
    Q3_API_END(h)
}

int q3tts_set_tensor_host(q3tts_engine* h, const char* name, const float* data, int64_t n) {
    Q3_API_BEGIN(h) h->e->set_tensor(name, data, n); return 0; Q3_API_END(h)
}
int q3tts_get_tensor_host(q3tts_engine* h, const char* name, float* out, int64_t n) {
    Q3_API_BEGIN(h) h->e->get_tensor(name, out, n); return 0; Q3_API_END(h)
}
int q3tts_fill_synthetic(q3tts_engine* h, uint64_t seed) {
    Q3_API_BEGIN(h) h->e->fill_synthetic(seed); return 0; Q3_API_END(h)
}
int q3tts_finalize(q3tts_engine* h) {
    Q3_API_BEGIN(h) h->e->finalize(); return 0; Q3_API_END(h)
}

int q3tts_text_project_host(q3tts_engine* h, const int64_t* ids, int n, float* out) {
    Q3_API_BEGIN(h) h->e->text_project(ids, n, out); return 0; Q3_API_END(h)
}
int q3tts_codec_embed_host(q3tts_engine* h, const int64_t* ids, int n, float* out) {
    Q3_API_BEGIN(h) h->e->codec_embed(ids, n, out); return 0; Q3_API_END(h)
}
int q3tts_cp_embed_host(q3tts_engine* h, int64_t id, int step, float* out) {
    Q3_API_BEGIN(h) h->e->cp_embed(id, step, out); return 0; Q3_API_END(h)
}
int q3tts_talker_prefill_host(q3tts_engine* h, int slot, const float* embeds, int S, float* logits, float* last_hidden) {
    Q3_API_BEGIN(h) h->e->talker_prefill(slot, embeds, S, logits, last_hidden); return 0; Q3_API_END(h)
}
int q3tts_talker_decode_host(q3tts_engine* h, int slot, const float* embed, float* logits, float* last_hidden) {
    Q3_API_BEGIN(h) h->e->talker_decode(slot, embed, logits, last_hidden); return 0; Q3_API_END(h)
}
int q3tts_code_predictor_host(q3tts_engine* h, const float* seq, int n, int step, float* logits) {
    Q3_API_BEGIN(h) h->e->code_predictor(seq, n, step, logits); return 0; Q3_API_END(h)
}
int q3tts_codec_decode_host(q3tts_engine* h, const int64_t* codes, int F, float* pcm, int64_t cap, int64_t* out_len) {
    Q3_API_BEGIN(h)
    const int64_t n = h->e->codec_decode_host(codes, F, pcm, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}

int q3tts_codec_decode_batch_host(q3tts_engine* h, int n_utt, const int64_t* codes, const int32_t* frame_offsets, float* const* pcm_out, int64_t pcm_cap,
                                  int64_t* pcm_len) {
    Q3_API_BEGIN(h)
    Engine& e = *h->e;
    if (n_utt <= 0) return 0;
    if (!codes || !frame_offsets) throw q3::Error("codec_decode_batch: null argument");
    if (pcm_cap < 0) throw q3::Error("codec_decode_batch: negative pcm_cap");
    if (!e.finalized) throw q3::Error("weights not finalized");
    const int G = e.c.n_groups;
    int row_frames = 1;
    std::vector<int32_t> nf((size_t)n_utt);
    for (int u = 0; u < n_utt; ++u) {
        const int64_t f = (int64_t)frame_offsets[u + 1] - frame_offsets[u];
        if (f < 0) throw q3::Error("codec_decode_batch: frame_offsets must not decrease");
        if (f > e.max_frames_cap) throw q3::Error("codec_decode: F out of range");   // the single-utterance entry point's bound and message
        nf[(size_t)u] = (int32_t)f;
        row_frames = std::max(row_frames, (int)f);
    }
    e.codec_async_prepare(row_frames, n_utt);
    std::vector<int32_t> rows((size_t)n_utt * row_frames * G, 0);
    for (int u = 0; u < n_utt; ++u) {
        const int64_t* src = codes + (size_t)frame_offsets[u] * G;
        int32_t* dst = rows.data() + (size_t)u * row_frames * G;
        for (size_t i = 0; i < (size_t)nf[(size_t)u] * G; ++i) {
            if (src[i] < 0 || src[i] >= e.c.cd_codebook) throw q3::Error("codec_decode: code out of range");   // as q3tts_codec_decode_host
            dst[i] = (int32_t)src[i];
        }
    }
    e.codec_job_upload(rows.data(), n_utt, row_frames);
    vocoder_job(e, n_utt, nf, row_frames, pcm_out, pcm_cap, pcm_len);
    return 0;
    Q3_API_END(h)
}

int q3tts_codec_decode_dev(q3tts_engine* h, const int32_t* codes_dev, int F, float* pcm_dev, int64_t cap, int64_t* out_len) {
    Q3_API_BEGIN(h)
    const int64_t n = h->e->codec_decode_dev(codes_dev, F, pcm_dev, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}
void* q3tts_stream(q3tts_engine* h) { return h && h->e ? (void*)h->e->stream : nullptr; }

int q3tts_codec_decode_chunked_host(q3tts_engine* h, const int64_t* codes, int F, int chunk_frames, int left_context, float* pcm, int64_t cap,
                                    int64_t* out_len) {
    Q3_API_BEGIN(h)
    const int64_t n = h->e->codec_decode_chunked_host(codes, F, chunk_frames, left_context, pcm, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}
int q3tts_codec_stream_begin(q3tts_engine* h, int max_frames, int* stream_id) {
    Q3_API_BEGIN(h)
    if (!stream_id) throw q3::Error("codec_stream_begin: null output");
    *stream_id = h->e->codec_stream_begin(max_frames);
    return 0;
    Q3_API_END(h)
}
int q3tts_codec_stream_push_host(q3tts_engine* h, int stream_id, const int64_t* codes, int n_frames, float* pcm, int64_t cap, int64_t* out_len) {
    Q3_API_BEGIN(h)
    if (!codes) throw q3::Error("codec_stream_push: null codes");
    const int64_t n = h->e->codec_stream_push_host(stream_id, codes, n_frames, pcm, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}
int q3tts_codec_stream_end(q3tts_engine* h, int stream_id) {
    Q3_API_BEGIN(h)
    h->e->codec_stream_end(stream_id);
    return 0;
    Q3_API_END(h)
}
int q3tts_slot_codec_decode_range_host(q3tts_engine* h, int slot, int frame_begin, int frame_end, int left_context, float* pcm, int64_t cap,
                                       int64_t* out_len) {
    Q3_API_BEGIN(h)
    const int64_t n = h->e->slot_codec_decode_range(slot, frame_begin, frame_end, left_context, pcm, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}

int64_t q3tts_codec_decode_len(const q3tts_config* c, int F) {
    int64_t T = F;
    for (int s = 0; s < c->cd_n_up; ++s) T *= c->cd_up_ratios[s];
    for (int i = 0; i < c->cd_n_blocks; ++i) {
        const int r = c->cd_up_rates[i], k = 2 * r, pad = k - r;
        const int left = c->cd_tconv_trim == 0 ? pad : 0;
        T = (T - 1) * r + k - left - pad;
    }
    return T;
}

int q3tts_sample_host(q3tts_engine* h, const float* logits, int n, const q3tts_sampling* p, float u, int suppress, int64_t* token) {
    Q3_API_BEGIN(h) h->e->sample(logits, n, *p, u, suppress, token); return 0; Q3_API_END(h)
}

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
float q3tts_rng_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t group) {
    uint64_t k = mix64(seed ^ mix64(((uint64_t)stream << 32) | frame));
    k = mix64(k + group);
    return (float)(k >> 40) * (1.0f / 16777216.0f);
}

int q3tts_build_prompt_host(q3tts_engine* h, const int64_t* ids, int n_ids, int lang, const float* speaker,
                            float* prompt, int* S, float* trailing, int cap_rows, int* n_trailing) {
    Q3_API_BEGIN(h) h->e->build_prompt(ids, n_ids, lang, speaker, prompt, S, trailing, cap_rows, n_trailing); return 0; Q3_API_END(h)
}

int q3tts_slot_begin(q3tts_engine* h, int slot, const float* prompt, int S, const float* trailing, int n_trailing,
                     const q3tts_sampling* p, uint64_t seed, uint32_t stream_id, int ignore_eos) {
    Q3_API_BEGIN(h) h->e->slot_begin(slot, prompt, S, trailing, n_trailing, *p, seed, stream_id, ignore_eos); return 0; Q3_API_END(h)
}
int q3tts_decode_steps(q3tts_engine* h, int n_steps) {
    Q3_API_BEGIN(h) return h->e->decode_steps(n_steps); Q3_API_END(h)
}
int q3tts_slot_status(q3tts_engine* h, int slot, int* n_frames, int* finished) {
    Q3_API_BEGIN(h) h->e->slot_status(slot, n_frames, finished); return 0; Q3_API_END(h)
}
int q3tts_slot_codes_host(q3tts_engine* h, int slot, int64_t* codes, int cap_frames) {
    Q3_API_BEGIN(h) h->e->slot_codes(slot, codes, cap_frames); return 0; Q3_API_END(h)
}
int q3tts_slot_logits_host(q3tts_engine* h, int slot, float* logits, float* last_hidden) {
    Q3_API_BEGIN(h) h->e->slot_logits(slot, logits, last_hidden); return 0; Q3_API_END(h)
}
int q3tts_slot_codec_decode_host(q3tts_engine* h, int slot, float* pcm, int64_t cap, int64_t* out_len) {
    Q3_API_BEGIN(h)
    const int64_t n = h->e->slot_codec_decode(slot, pcm, cap);
    if (out_len) *out_len = n;
    return 0;
    Q3_API_END(h)
}
int q3tts_kv_pool_info(q3tts_engine* h, int* page_tokens, int* total_pages, int* free_pages) {
    Q3_API_BEGIN(h)
    if (page_tokens) *page_tokens = 1 << h->e->talker.page_shift;
    if (total_pages) *total_pages = h->e->kv_total_pages();
    if (free_pages) *free_pages = h->e->kv_free_pages();
    return 0;
    Q3_API_END(h)
}

int q3tts_sched_stats(q3tts_engine* h, int64_t* admitted, int64_t* preempted, int* peak_live) {
    Q3_API_BEGIN(h)
    if (admitted) *admitted = h->e->sched_admitted;
    if (preempted) *preempted = h->e->sched_preempted;
    if (peak_live) *peak_live = h->e->sched_peak_live;
    return 0;
    Q3_API_END(h)
}

int q3tts_slot_release(q3tts_engine* h, int slot) {
    Q3_API_BEGIN(h) h->e->slot_release(slot); return 0; Q3_API_END(h)
}

// synthesize_tokens (reference src/tts_onnx.cpp:405-436) over a batch of independent utterances:
// waves of up to max_batch slots; prompt assembly -> prefill -> fused decode -> vocoder.
int q3tts_synthesize_batch_host(q3tts_engine* h, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                const q3tts_sampling* p, uint64_t seed, int ignore_eos,
                                float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames,
                                int64_t* codes_out) {
    return q3tts_synthesize_clone_batch_host(h, n_utt, ids, offsets, lang, nullptr, p, seed, ignore_eos, pcm_out, pcm_cap, pcm_len, n_frames, codes_out);
}

// Continuous batching: utterances queue for the engine's max_batch slots; a slot that finishes (EOS or max_new_tokens) stashes its codes
// and is re-armed with the next utterance at the following look, so ragged lengths do not idle the batch; the vocoder runs over the
// side lanes once the decode queue is empty.  Results do not depend on the schedule: the RNG stream of an utterance is its index,
// never its slot.
int q3tts_synthesize_clone_batch_host(q3tts_engine* h, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                      const float* const* speakers, const q3tts_sampling* p, uint64_t seed, int ignore_eos,
                                      float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames,
                                      int64_t* codes_out) {
    return q3tts_synthesize_schedule_host(h, n_utt, ids, offsets, lang, speakers, p, nullptr, seed, ignore_eos, pcm_out, pcm_cap, pcm_len, n_frames, codes_out);
}

int q3tts_synthesize_schedule_host(q3tts_engine* h, int n_utt, const int64_t* ids, const int32_t* offsets, int lang,
                                   const float* const* speakers, const q3tts_sampling* p, const int32_t* max_new_per_utt, uint64_t seed, int ignore_eos,
                                   float* const* pcm_out, int64_t pcm_cap, int64_t* pcm_len, int32_t* n_frames,
                                   int64_t* codes_out) {
    Q3_API_BEGIN(h)
    Engine& e = *h->e;
    const int H = e.c.hidden, G = e.c.n_groups, B = e.B;
    if (n_utt <= 0) return 0;
    if (!ids || !offsets || !p) throw q3::Error("synthesize: null argument");
    if (pcm_cap < 0) throw q3::Error("synthesize: negative pcm_cap");
    for (int b = 0; b < B; ++b) e.slot_release(b);
    const int row_frames = std::max(1, std::min(p->max_new_tokens, e.max_frames_cap));
    e.codec_async_prepare(row_frames, n_utt);
    std::vector<int32_t> got_frames((size_t)n_utt, 0);
    // Prompt rows of every utterance are assembled before the first step, all in one projection pass (Engine::build_prompts): one
    // utterance at a time it is a handful of small synchronous device round trips each.
    struct Prep { int S = 0, nt = 0; size_t poff = 0, toff = 0; };
    std::vector<Prep> prep((size_t)n_utt);
    std::vector<float> prompts, trailing;
    {
        size_t prow = 0, trow = 0;
        for (int u = 0; u < n_utt; ++u) {
            prep[(size_t)u].poff = prow; prep[(size_t)u].toff = trow;
            prow += 16; trow += (size_t)std::max(1, offsets[u + 1] - offsets[u] - 3);   // trailing rows: text tokens minus the first, plus tts_eos
        }
        prompts.resize(prow * H); trailing.resize(trow * H);
        std::vector<size_t> toffs((size_t)n_utt);
        std::vector<int> Ss((size_t)n_utt), nts((size_t)n_utt);
        for (int u = 0; u < n_utt; ++u) toffs[(size_t)u] = prep[(size_t)u].toff;
        e.build_prompts(ids, offsets, n_utt, lang, speakers, prompts.data(), Ss.data(), trailing.data(), toffs.data(), nts.data());
        for (int u = 0; u < n_utt; ++u) { prep[(size_t)u].S = Ss[(size_t)u]; prep[(size_t)u].nt = nts[(size_t)u]; }
    }
    // KV pages.  With EOS suppressed every length is known: an utterance is admitted when the pool holds prompt + cap and never waits again.
    // Otherwise a slot owns what its context has reached plus the coming look (on-demand growth), so utterances that end early never hold
    // the pages of their cap; when the pool runs dry the YOUNGEST live utterance is preempted — its pages go back, it returns to the head
    // of the queue and is generated again from its prompt later (same RNG stream, same codes) — so the oldest always finishes.
    std::vector<Engine::SlotInit> init;
    std::vector<int> slot_utt((size_t)B, -1), done_frames((size_t)B, 0), fresh, retired;
    std::vector<q3::SlotState> st;
    std::deque<int> pending;
    auto cap_of = [&](int u) { return max_new_per_utt ? std::min(std::max(1, (int)max_new_per_utt[u]), p->max_new_tokens) : p->max_new_tokens; };
    for (int u = 0; u < n_utt; ++u) {
        if (e.kv_pages_for(prep[(size_t)u].S + cap_of(u)) > e.kv_total_pages())
            throw q3::Error("synthesize: one utterance (prompt + max_new_tokens) needs more KV pages than the pool holds");
        pending.push_back(u);
    }
    const bool reserve_all = ignore_eos != 0;
    const int first_look = 8;
    int live = 0;
    // A preempted utterance is not re-admitted on the pages it just gave back: it waits until a live utterance has retired or the pool
    // holds what it owned when it was preempted plus one page of head-room — otherwise a pool just above one utterance's cap re-admits
    // and preempts the same utterance look after look, paying a prefill and discarding its frames each time.
    std::vector<int> hold_pages((size_t)n_utt, 0);
    std::vector<int64_t> hold_mark((size_t)n_utt, 0);
    int64_t n_retired = 0;
    e.sched_admitted = e.sched_preempted = 0; e.sched_peak_live = 0;
    try {
        while (!pending.empty() || live > 0) {
            fresh.clear();
            {   // admission in queue order (q3_kvpool.h: sched_admit_count)
                std::vector<int> free_slots, need;
                for (int b = 0; b < B; ++b) if (slot_utt[(size_t)b] < 0) free_slots.push_back(b);
                for (size_t i = 0; i < pending.size() && i < free_slots.size(); ++i) {
                    const int u = pending[i], all = prep[(size_t)u].S + cap_of(u);
                    int pages = e.kv_pages_for(reserve_all ? all : std::min(all, prep[(size_t)u].S + first_look));
                    if (hold_pages[(size_t)u] > 0 && n_retired == hold_mark[(size_t)u] && live > 0) pages = std::max(pages, std::min(hold_pages[(size_t)u], e.kv_total_pages()));
                    need.push_back(pages);
                }
                const int n_adm = q3::sched_admit_count(e.kv, need, (int)free_slots.size(), live, reserve_all);
                for (int i = 0; i < n_adm; ++i) { slot_utt[(size_t)free_slots[(size_t)i]] = pending.front(); pending.pop_front(); fresh.push_back(free_slots[(size_t)i]); }
            }
            if (!fresh.empty()) {
                init.assign(fresh.size(), Engine::SlotInit());
                for (size_t i = 0; i < fresh.size(); ++i) {
                    const int b = fresh[i], u = slot_utt[(size_t)b];
                    const Prep& pr = prep[(size_t)u];
                    Engine::SlotInit& q = init[i];
                    q.slot = b; q.prompt = prompts.data() + pr.poff * H; q.S = pr.S; q.trailing = trailing.data() + pr.toff * H; q.n_trailing = pr.nt;
                    q.stream_id = (uint32_t)u;
                    q.max_frames = max_new_per_utt ? std::max(1, (int)max_new_per_utt[u]) : 0;
                    q.kv_tokens = reserve_all ? 0 : pr.S + first_look;
                }
                e.slots_begin(init.data(), (int)init.size(), *p, seed, ignore_eos);   // equal-length prompts in consecutive slots share one prefill pass
                live += (int)fresh.size();
                e.sched_admitted += (int64_t)fresh.size();
                e.sched_peak_live = std::max(e.sched_peak_live, live);
            }
            // steps until the next look: never past the earliest slot that can reach max_new_tokens, short when few utterances are live
            int rem = p->max_new_tokens;
            for (int b = 0; b < B; ++b) {
                const int u = slot_utt[(size_t)b];
                if (u >= 0) rem = std::min(rem, cap_of(u) - done_frames[(size_t)b]);
            }
            // (with EOS suppressed nothing can finish earlier than that, so the look-ahead only bounds how long the host is away)
            // While utterances wait in the queue a look is at least `quantum` steps away: a finished slot idles a few (masked, nearly free)
            // steps, and the slots that finish within the quantum are re-armed together, sharing one prefill pass.
            static const int quantum = getenv("Q3TTS_SCHED_QUANTUM") ? std::max(1, atoi(getenv("Q3TTS_SCHED_QUANTUM"))) : 4;
            const int look = std::min(rem, ignore_eos ? 64 : (live <= 16 ? 4 : 8));
            const int steps = std::max(!pending.empty() && live > 16 ? quantum : 1, look);
            if (!reserve_all) {
                // every live slot gets pages for the positions these steps write; oldest first, so that when the pool runs dry it is the
                // youngest that goes back to the queue (q3_kvpool.h: sched_grow)
                std::vector<int> order, want, changed;
                for (int b = 0; b < B; ++b) if (slot_utt[(size_t)b] >= 0) order.push_back(b);
                std::sort(order.begin(), order.end(), [&](int x, int y) {
                    return done_frames[(size_t)x] != done_frames[(size_t)y] ? done_frames[(size_t)x] > done_frames[(size_t)y] : slot_utt[(size_t)x] < slot_utt[(size_t)y]; });
                for (int b : order) {
                    const int u = slot_utt[(size_t)b];
                    want.push_back(std::min(prep[(size_t)u].S + cap_of(u), prep[(size_t)u].S + done_frames[(size_t)b] + steps));
                }
                const std::vector<int> victims = q3::sched_grow(e.kv, order, want, &changed);
                for (int vb : victims) {          // youngest first: pushed to the front one by one, the oldest of them ends up first in the queue
                    const int vu = slot_utt[(size_t)vb];
                    hold_pages[(size_t)vu] = e.kv_pages_for(prep[(size_t)vu].S + done_frames[(size_t)vb]) + 1;
                    hold_mark[(size_t)vu] = n_retired;
                    e.slot_release(vb);           // deactivates the slot (its pages are already back in the pool)
                    pending.push_front(vu);
                    slot_utt[(size_t)vb] = -1; done_frames[(size_t)vb] = 0;
                    --live; ++e.sched_preempted;
                }
                for (int b : changed) e.kv_upload_row(b);
            }
            e.decode_steps(steps);
            e.slots_state(B, st);
            retired.clear();
            for (int b = 0; b < B; ++b) {
                const int u = slot_utt[(size_t)b];
                if (u < 0) continue;
                const q3::SlotState& s = st[(size_t)b];
                done_frames[(size_t)b] = s.n_frames;
                if (!(s.finished || s.n_frames >= s.max_frames)) continue;
                got_frames[(size_t)u] = s.n_frames;
                if (n_frames) n_frames[u] = s.n_frames;
                e.codec_stash(b, s.n_frames, u, row_frames);
                retired.push_back(b);
            }
            for (int b : retired) {
                const int u = slot_utt[(size_t)b];
                if (codes_out) e.slot_codes(b, codes_out + (size_t)u * p->max_new_tokens * G, p->max_new_tokens);
                e.slot_release(b);
                slot_utt[(size_t)b] = -1; done_frames[(size_t)b] = 0;
                --live; ++n_retired;
            }
        }
    } catch (...) {
        try { e.codec_async_drain(); } catch (...) { }
        for (int b = 0; b < B; ++b) { try { e.slot_release(b); } catch (...) { } }
        throw;
    }
    vocoder_job(e, n_utt, got_frames, row_frames, pcm_out, pcm_cap, pcm_len);
    return 0;
    Q3_API_END(h)
}

// ---- voice-clone front end (host audio code + the speaker encoder on the GPU) ----
int q3tts_read_wav_host(const char* path, float* out, int64_t cap, int64_t* n_samples, int32_t* sample_rate) {
    if (!path || !n_samples || !sample_rate) return -1;
    try {
        int sr = 0;
        const std::vector<float> a = q3::read_wav(path, &sr);
        if (a.empty()) return -1;
        *n_samples = (int64_t)a.size();
        *sample_rate = sr;
        for (int64_t i = 0; i < (int64_t)a.size() && i < cap && out; ++i) out[i] = a[(size_t)i];
        return 0;
    } catch (...) { return -1; }
}
int64_t q3tts_resample_host(const float* in, int64_t n, int32_t src_rate, int32_t dst_rate, float* out, int64_t cap) {
    if (n < 0 || (n > 0 && !in) || src_rate <= 0 || dst_rate <= 0) return -1;
    try {
        const std::vector<float> r = q3::resample_linear(std::vector<float>(in, in + n), src_rate, dst_rate);
        for (int64_t i = 0; i < (int64_t)r.size() && i < cap && out; ++i) out[i] = r[(size_t)i];
        return (int64_t)r.size();
    } catch (...) { return -1; }
}
int q3tts_mel_host(const float* audio, int64_t n, float* mel, int64_t cap, int32_t* frames) {
    if (n < 0 || (n > 0 && !audio) || !frames) return -1;
    try {
        int f = 0;
        const std::vector<float> m = q3::log_mel(std::vector<float>(audio, audio + n), q3::MelSpec(), &f);
        *frames = f;
        if (m.empty()) return -1;
        if (mel) {
            if ((int64_t)m.size() > cap) return -1;
            memcpy(mel, m.data(), m.size() * sizeof(float));
        }
        return 0;
    } catch (...) { return -1; }
}
int q3tts_has_speaker_encoder(q3tts_engine* h) { return h && h->e && h->e->has_speaker() ? 1 : 0; }
int q3tts_speaker_encoder_host(q3tts_engine* h, const float* mel, int frames, float* embed) {
    Q3_API_BEGIN(h) h->e->speaker_encode(mel, frames, embed); return 0; Q3_API_END(h)
}
int q3tts_extract_speaker_embedding_host(q3tts_engine* h, const char* wav_path, float* embed) {
    Q3_API_BEGIN(h)
    if (!h->e->has_speaker()) throw q3::Error("model has no speaker encoder");
    int sr = 0;
    std::vector<float> a = q3::read_wav(wav_path, &sr);
    if (a.empty()) throw q3::Error(std::string("Failed to read audio: ") + wav_path);
    // the header's 32-bit rate is untrusted: >= 2^31 reads as a negative int (a negative resampling ratio is undefined behaviour)
    if (sr <= 0 || sr > 768000) throw q3::Error(std::string("Failed to read audio: ") + wav_path + " (sample rate " + std::to_string((unsigned)sr) + " out of range)");
    if (sr != 24000) a = q3::resample_linear(a, sr, 24000);
    int frames = 0;
    const std::vector<float> m = q3::log_mel(a, q3::MelSpec(), &frames);
    if (m.empty()) throw q3::Error("Failed to extract mel spectrogram");
    h->e->speaker_encode(m.data(), frames, embed);
    return 0;
    Q3_API_END(h)
}

int q3tts_last_decode_ms(q3tts_engine* h, float* ms, int* steps) {
    Q3_API_BEGIN(h)
    if (ms) *ms = h->e->last_decode_ms;
    if (steps) *steps = h->e->last_decode_steps;
    return 0;
    Q3_API_END(h)
}
int q3tts_last_codec_ms(q3tts_engine* h, float* ms) {
    Q3_API_BEGIN(h) if (ms) *ms = h->e->last_codec_ms; return 0; Q3_API_END(h)
}
int q3tts_counters(q3tts_engine* h, double* dms, int64_t* dsteps, double* cms, int64_t* cframes, int reset) {
    Q3_API_BEGIN(h)
    if (dms) *dms = h->e->total_decode_ms;
    if (dsteps) *dsteps = h->e->total_decode_steps;
    if (cms) *cms = h->e->total_codec_ms;
    if (cframes) *cframes = h->e->total_codec_frames;
    if (reset) { h->e->total_decode_ms = 0; h->e->total_decode_steps = 0; h->e->total_codec_ms = 0; h->e->total_codec_frames = 0; }
    return 0;
    Q3_API_END(h)
}
int q3tts_codec_plane_stats(q3tts_engine* h, int* two_product, int* three_product) {
    Q3_API_BEGIN(h)
    h->e->codec_plane_stats(two_product, three_product);
    return 0;
    Q3_API_END(h)
}
int q3tts_prefill_profile(q3tts_engine* h, int n_slots, int n_rows, int reps, double* ms_per_pass) {
    Q3_API_BEGIN(h)
    if (!ms_per_pass) throw q3::Error("prefill_profile: null output");
    h->e->prefill_profile(n_slots, n_rows, reps, ms_per_pass);
    return 0;
    Q3_API_END(h)
}
int q3tts_stage_profile(q3tts_engine* h, int n_steps, double* out_ms) {
    Q3_API_BEGIN(h)
    if (!out_ms) throw q3::Error("stage_profile: null output");
    h->e->stage_profile(n_steps, out_ms);
    return 0;
    Q3_API_END(h)
}
// ---- batch-first device-pointer entry points ----
#define Q3_DEV_CALL(h, strm, body)                                       \
    Q3_API_BEGIN(h)                                                      \
    hipStream_t caller_ = (strm) ? (hipStream_t)(strm) : (h)->e->stream; \
    (h)->e->stream_join(caller_);                                        \
    body;                                                                \
    (h)->e->stream_fork(caller_);                                        \
    return 0;                                                            \
    Q3_API_END(h)
int q3tts_talker_prefill_dev(q3tts_engine* h, const float* embeds, int batch, int S, const int32_t* lens, float* logits_last, float* last_hidden, void* stream) {
    Q3_DEV_CALL(h, stream, h->e->talker_prefill_dev(embeds, batch, S, lens, logits_last, last_hidden))
}
int q3tts_talker_decode_dev(q3tts_engine* h, const float* embeds, int batch, const uint8_t* active_mask, float* logits, float* last_hidden, void* stream) {
    Q3_DEV_CALL(h, stream, h->e->talker_decode_dev(embeds, batch, active_mask, logits, last_hidden))
}
int q3tts_code_predictor_dev(q3tts_engine* h, const float* last_hidden, const int64_t* code0, int batch, const q3tts_sampling* p, uint64_t seed,
                             uint32_t stream_id0, uint32_t frame, int32_t* sub, void* stream) {
    if (!p) return -1;
    Q3_DEV_CALL(h, stream, h->e->code_predictor_dev(last_hidden, code0, batch, *p, seed, stream_id0, frame, sub))
}
int q3tts_sample_dev(q3tts_engine* h, const float* logits, int batch, int n, const q3tts_sampling* p, const float* u, int suppress, int64_t* ids, void* stream) {
    if (!p) return -1;
    Q3_DEV_CALL(h, stream, h->e->sample_dev(logits, batch, n, *p, u, suppress, ids))
}
int q3tts_measure_skip_frames(q3tts_engine* h, int n_frames) {
    Q3_API_BEGIN(h)
    h->e->measure_skip_frames(n_frames);
    return 0;
    Q3_API_END(h)
}
int q3tts_test_group_final_conv(q3tts_engine* h, float* sx_out, float* pcm_out, int64_t cap_floats, int32_t* T, int32_t* C, int32_t* nb) {
    Q3_API_BEGIN(h)
    int t = 0, c = 0, n = 0;
    h->e->codec_debug_group(sx_out, pcm_out, cap_floats, &t, &c, &n);
    if (T) *T = t; if (C) *C = c; if (nb) *nb = n;
    return 0;
    Q3_API_END(h)
}
int64_t q3tts_test_final_conv_partials(q3tts_engine* h, float* out, int64_t cap_floats) {
    if (!h || !h->e) return -1;
    try { return h->e->codec_debug_partials(out, cap_floats); } catch (...) { return -1; }
}
int q3tts_test_poison_workspace(q3tts_engine* h) {
    Q3_API_BEGIN(h)
    h->e->codec_poison();
    return 0;
    Q3_API_END(h)
}
int q3tts_step_logits_host(q3tts_engine* h, int slot, float* out, int cols) {
    Q3_API_BEGIN(h)
    if (!out) throw q3::Error("step_logits: null output");
    h->e->step_logits(slot, out, cols);
    return 0;
    Q3_API_END(h)
}
int q3tts_decode_step_bytes(q3tts_engine* h, double* wb, double* kvb) {
    Q3_API_BEGIN(h) h->e->step_bytes(wb, kvb); return 0; Q3_API_END(h)
}

} // extern "C"

// ---- weight files -----------------------------------------------------------------------------
namespace {
const char kMagic[8] = { 'Q', '3', 'T', 'W', '0', '0', '0', '1' };
struct File {
    FILE* f;
    explicit File(const char* p, const char* mode) : f(fopen(p, mode)) {}
    ~File() { if (f) fclose(f); }
};
void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) throw q3::Error("weights file truncated"); }
void wr(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) throw q3::Error("weights file write failed"); }
void read_header(FILE* f, q3tts_config* cfg, uint32_t* n) {
    char magic[8];
    rd(f, magic, 8);
    if (memcmp(magic, kMagic, 8) != 0) throw q3::Error("not a Q3TW0001 weights file");
    uint32_t cfg_bytes = 0;
    rd(f, &cfg_bytes, 4);
    // files written before the speaker-encoder fields existed carry a shorter config: the missing tail reads as zero
    if (cfg_bytes > sizeof(q3tts_config) || cfg_bytes < offsetof(q3tts_config, spk_enc_dim)) throw q3::Error("weights file: config struct size mismatch");
    memset(cfg, 0, sizeof *cfg);
    rd(f, cfg, cfg_bytes);
    if (cfg->cp_hidden == cfg->hidden) cfg->cp_hidden = 0; // one spelling of "same width" (the engine normalises likewise)
    rd(f, n, 4);
}
} // namespace

extern "C" {

int q3tts_read_weights_config(const char* path, q3tts_config* out) {
    try {
        File fl(path, "rb");
        if (!fl.f) throw q3::Error(std::string("cannot open ") + path);
        uint32_t n = 0;
        read_header(fl.f, out, &n);
        return 0;
    } catch (const q3::Error& ex) { g_create_err = ex.msg; return -1; }
}

int q3tts_load_weights_file(q3tts_engine* h, const char* path) {
    Q3_API_BEGIN(h)
    File fl(path, "rb");
    if (!fl.f) throw q3::Error(std::string("cannot open ") + path);
    q3tts_config cfg;
    uint32_t n = 0;
    read_header(fl.f, &cfg, &n);
    if (memcmp(&cfg, &h->e->c, sizeof cfg) != 0) throw q3::Error("weights file was written for a different model config");
    std::vector<float> f32;
    std::vector<uint16_t> b16;
    // every registry tensor exactly once: storage is a bare hipMalloc, so a tensor the file does not carry would be synthesized from
    // uninitialised HBM (the reference refuses to start when a model file is missing, tts_onnx.cpp:91-107)
    std::vector<char> seen(h->e->tensors.size(), 0);
    for (uint32_t t = 0; t < n; ++t) {
        uint16_t nl = 0;
        rd(fl.f, &nl, 2);
        std::string name(nl, '\0');
        rd(fl.f, &name[0], nl);
        uint8_t dtype = 0;
        rd(fl.f, &dtype, 1);
        uint64_t numel = 0;
        rd(fl.f, &numel, 8);
        // the header is untrusted: size the buffers from the registry, not from the file
        const q3::Tensor& want = h->e->T(name);
        char& mark = seen[(size_t)h->e->tindex.at(name)];
        if (mark) throw q3::Error("weights file: tensor '" + name + "' appears twice");
        mark = 1;
        if ((uint64_t)want.numel != numel) throw q3::Error("weights file: tensor '" + name + "' has " + std::to_string(numel) + " elements, expected " + std::to_string(want.numel));
        f32.resize(numel);
        if (dtype == 0) rd(fl.f, f32.data(), numel * 4);
        else if (dtype == 1) {
            b16.resize(numel);
            rd(fl.f, b16.data(), numel * 2);
            for (uint64_t i = 0; i < numel; ++i) f32[i] = q3::bf16_to_f32(b16[i]);
        } else throw q3::Error("weights file: unknown dtype for " + name);
        h->e->set_tensor(name, f32.data(), (int64_t)numel);
    }
    {   // name every missing tensor at once (an importer with a wrong key prefix drops whole components)
        std::string missing;
        size_t n_missing = 0;
        for (size_t i = 0; i < seen.size(); ++i)
            if (!seen[i]) { if (n_missing++ < 24) missing += (missing.empty() ? "" : ", ") + h->e->tensors[i].name; }
        if (n_missing) throw q3::Error("weights file: " + std::to_string(n_missing) + " of " + std::to_string(seen.size()) + " tensors missing: " + missing + (n_missing > 24 ? ", ..." : ""));
    }
    h->e->finalize();
    return 0;
    Q3_API_END(h)
}

int q3tts_save_weights_file(q3tts_engine* h, const char* path) {
    Q3_API_BEGIN(h)
    File fl(path, "wb");
    if (!fl.f) throw q3::Error(std::string("cannot create ") + path);
    wr(fl.f, kMagic, 8);
    const uint32_t cfg_bytes = sizeof(q3tts_config), n = (uint32_t)h->e->tensors.size();
    wr(fl.f, &cfg_bytes, 4);
    wr(fl.f, &h->e->c, sizeof(q3tts_config));
    wr(fl.f, &n, 4);
    std::vector<float> f32;
    std::vector<uint16_t> b16;
    for (const q3::Tensor& t : h->e->tensors) {
        const uint16_t nl = (uint16_t)t.name.size();
        wr(fl.f, &nl, 2);
        wr(fl.f, t.name.data(), nl);
        const uint8_t dtype = t.bf16 ? 1 : 0;
        wr(fl.f, &dtype, 1);
        const uint64_t numel = (uint64_t)t.numel;
        wr(fl.f, &numel, 8);
        f32.resize(numel);
        h->e->get_tensor(t.name, f32.data(), t.numel);
        if (t.bf16) {
            b16.resize(numel);
            for (uint64_t i = 0; i < numel; ++i) b16[i] = q3::f32_to_bf16(f32[i]);
            wr(fl.f, b16.data(), numel * 2);
        } else wr(fl.f, f32.data(), numel * 4);
    }
    return 0;
    Q3_API_END(h)
}

// ---- text front end (host only) ----
struct q3tts_tokenizer {
    q3::BpeTokenizer t;
};

q3tts_tokenizer* q3tts_tokenizer_create(void) {
    try { return new q3tts_tokenizer(); } catch (...) { return nullptr; }
}
void q3tts_tokenizer_destroy(q3tts_tokenizer* t) { delete t; }
int q3tts_tokenizer_load_vocab(q3tts_tokenizer* t, const char* path) {
    if (!t || !path) return -1;
    try { return t->t.load_vocab(path) ? 0 : -1; } catch (...) { return -1; }
}
int q3tts_tokenizer_load_merges(q3tts_tokenizer* t, const char* path) {
    if (!t || !path) return -1;
    try { return t->t.load_merges(path) ? 0 : -1; } catch (...) { return -1; }
}
int q3tts_tokenizer_ready(const q3tts_tokenizer* t) { return t && t->t.ready() ? 1 : 0; }
int64_t q3tts_tokenize(const q3tts_tokenizer* t, const char* text, int64_t len, int32_t* ids, int64_t cap) {
    if (!t || len < 0 || (len > 0 && !text)) return -1;
    try {
        std::vector<int32_t> v;
        t->t.encode(text, (size_t)len, v);
        for (int64_t i = 0; i < (int64_t)v.size() && i < cap && ids; ++i) ids[i] = v[(size_t)i];
        return (int64_t)v.size();
    } catch (...) { return -1; }
}

} // extern "C"
