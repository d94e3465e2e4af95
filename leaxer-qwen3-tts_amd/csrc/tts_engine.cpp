// tts_engine.cpp — leaxer_qwen::TTSEngine over the C-ABI (include/q3tts.h).
#include "tts_engine.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>

#include "../../include/q3tts.h"

namespace leaxer_qwen {

Speaker parse_speaker(const std::string& name) { // reference src/tts_onnx.cpp:54-68: case-insensitive names
    std::string s;
    for (char ch : name) s.push_back((char)std::tolower((unsigned char)ch));
    static const struct { const char* n; Speaker v; } table[] = {
        { "serena", Speaker::Serena }, { "vivian", Speaker::Vivian }, { "uncle_fu", Speaker::Uncle_Fu },
        { "dylan", Speaker::Dylan }, { "eric", Speaker::Eric }, { "ryan", Speaker::Ryan }, { "aiden", Speaker::Aiden },
        { "ono_anna", Speaker::Ono_Anna }, { "sohee", Speaker::Sohee } };
    for (const auto& e : table) if (s == e.n) return e.v;
    return Speaker::None;
}

static int lang_index(Language l) { return l == Language::Auto ? 0 : (int)(language_to_codec_id(l) - config::LANG_ENGLISH) + 1; }

TTSEngine::TTSEngine(const std::string& model_dir) {
    const char* env_b = std::getenv("Q3TTS_MAX_BATCH");
    max_batch_ = env_b ? std::max(1, std::atoi(env_b)) : 1;
    const int device = std::getenv("Q3TTS_DEVICE") ? std::atoi(std::getenv("Q3TTS_DEVICE")) : 0;
    const int max_ctx = config::MAX_NEW_TOKENS + 64;
    q3tts_config cfg;
    if (model_dir.rfind("synthetic:", 0) == 0 || model_dir.rfind("synthetic-1.7b:", 0) == 0) {   // seeded weights at the 0.6B / 1.7B dims
        const bool big = model_dir[9] == '-';
        q3tts_default_config(big ? "1.7b" : "0.6b", &cfg);
        h_ = q3tts_create(&cfg, device, max_batch_, max_ctx, 0);
        if (!h_) { error_msg_ = q3tts_last_error(nullptr); return; }
        if (q3tts_fill_synthetic(h_, std::strtoull(model_dir.c_str() + (big ? 15 : 10), nullptr, 10)) != 0 || q3tts_finalize(h_) != 0) { error_msg_ = q3tts_last_error(h_); return; }
    } else {
        const std::string path = model_dir + "/model.q3w";
        if (q3tts_read_weights_config(path.c_str(), &cfg) != 0) { error_msg_ = std::string("Failed to load ") + path + ": " + q3tts_last_error(nullptr); return; }
        h_ = q3tts_create(&cfg, device, max_batch_, max_ctx, 0);
        if (!h_) { error_msg_ = q3tts_last_error(nullptr); return; }
        if (q3tts_load_weights_file(h_, path.c_str()) != 0) { error_msg_ = q3tts_last_error(h_); return; }
    }
    spk_dim_ = cfg.spk_enc_dim;
    cfg_hidden_ = cfg.hidden;
    // tokenizer files: where the reference looks (tts_onnx.cpp:110-121: <parent of model_dir>/models/
    // Qwen3-TTS-12Hz-0.6B-Base/{vocab.json,merges.txt}), then model_dir itself.  Present but unreadable is
    // an error, absent is a warning and text synthesis stays unavailable — as in the reference.
    namespace fs = std::filesystem;
    tok_ = q3tts_tokenizer_create();
    const fs::path ref_base = fs::path(model_dir).parent_path() / "models" / "Qwen3-TTS-12Hz-0.6B-Base";
    fs::path base = ref_base;
    std::error_code ec;
    if (!(fs::exists(base / "vocab.json", ec) && fs::exists(base / "merges.txt", ec)) && model_dir.rfind("synthetic:", 0) != 0)
        base = fs::path(model_dir);
    if (fs::exists(base / "vocab.json", ec) && fs::exists(base / "merges.txt", ec)) {
        if (!tok_ || q3tts_tokenizer_load_vocab(tok_, (base / "vocab.json").string().c_str()) != 0 ||
            q3tts_tokenizer_load_merges(tok_, (base / "merges.txt").string().c_str()) != 0) {
            error_msg_ = "Failed to load tokenizer";
            return;
        }
    } else {
        std::cerr << "[TTSEngine] Warning: Tokenizer not found at " << ref_base << std::endl;
    }
    ready_ = true;
}

TTSEngine::~TTSEngine() {
    if (tok_) q3tts_tokenizer_destroy(tok_);
    if (h_) q3tts_destroy(h_);
}

bool TTSEngine::wrap_text(const std::string& text, std::vector<int64_t>& ids) const {
    // reference tts_onnx.cpp:243-259: [IM_START, ASSISTANT, TTS_BOS, ...text..., TTS_EOS, IM_END]
    if (!q3tts_tokenizer_ready(tok_)) {
        std::cerr << "[TTSEngine] Tokenizer not ready" << std::endl;
        return false;
    }
    const int64_t n = q3tts_tokenize(tok_, text.data(), (int64_t)text.size(), nullptr, 0);
    if (n < 0) return false;
    std::vector<int32_t> t((size_t)n);
    q3tts_tokenize(tok_, text.data(), (int64_t)text.size(), t.data(), n);
    ids = { config::IM_START, config::ASSISTANT, config::TTS_BOS };
    ids.insert(ids.end(), t.begin(), t.end());
    ids.push_back(config::TTS_EOS);
    ids.push_back(config::IM_END);
    return true;
}

std::vector<int32_t> TTSEngine::tokenize(const std::string& text) const {
    std::vector<int32_t> t;
    const int64_t n = q3tts_tokenize(tok_, text.data(), (int64_t)text.size(), nullptr, 0);
    if (n <= 0) return t;
    t.resize((size_t)n);
    q3tts_tokenize(tok_, text.data(), (int64_t)text.size(), t.data(), n);
    return t;
}

std::vector<float> TTSEngine::synthesize(const std::string& text, Language lang, const SamplingParams& params) {
    if (!ready_) return {};
    std::vector<int64_t> ids;
    if (!wrap_text(text, ids)) return {};
    return synthesize_tokens(ids, lang, params);
}

std::vector<std::vector<float>> TTSEngine::synthesize_batch(const std::vector<std::string>& texts, Language lang,
                                                            const SamplingParams& params) {
    std::vector<std::vector<int64_t>> ids(texts.size());
    if (!ready_) return std::vector<std::vector<float>>(texts.size());
    for (size_t i = 0; i < texts.size(); ++i)
        if (!wrap_text(texts[i], ids[i])) return std::vector<std::vector<float>>(texts.size());
    return synthesize_tokens_batch(ids, lang, params);
}

bool TTSEngine::has_speaker_encoder() const { return h_ && q3tts_has_speaker_encoder(h_); }

std::vector<float> TTSEngine::synthesize_clone(const std::string& text, const std::string& ref_audio_path, Language lang,
                                               const SamplingParams& params) { // reference tts_onnx.cpp:264-318
    if (!ready_) return {};
    if (!has_speaker_encoder()) {
        std::cerr << "[TTSEngine] Speaker encoder not available" << std::endl;
        return {};
    }
    const std::vector<float> spk = extract_speaker_embedding(ref_audio_path);
    if (spk.empty()) {
        std::cerr << "[TTSEngine] Failed to extract speaker embedding" << std::endl;
        return {};
    }
    std::vector<int64_t> ids;
    if (!wrap_text(text, ids)) return {};
    return synthesize_tokens_clone(ids, spk, lang, params);
}

std::vector<float> TTSEngine::synthesize_tokens_clone(const std::vector<int64_t>& token_ids, const std::vector<float>& speaker_embed,
                                                      Language lang, const SamplingParams& params) {
    if (!ready_) return {};
    if (!speaker_embed.empty() && (int)speaker_embed.size() != cfg_hidden_) {   // the row is spliced into the prompt as one talker-width embedding
        std::cerr << "[TTSEngine] Synthesis error: speaker embedding has " << speaker_embed.size() << " values, the model needs " << cfg_hidden_ << std::endl;
        return {};
    }
    q3tts_sampling sp{ params.temperature, params.top_p, params.top_k, params.repetition_penalty, params.max_new_tokens };
    const int32_t offs[2] = { 0, (int32_t)token_ids.size() };
    const int64_t cap = (int64_t)params.max_new_tokens * 1920 + 1920;
    std::vector<float> pcm((size_t)cap);
    float* ptr = pcm.data();
    const float* spk = speaker_embed.empty() ? nullptr : speaker_embed.data();
    int64_t len = 0;
    int32_t frames = 0;
    if (q3tts_synthesize_clone_batch_host(h_, 1, token_ids.data(), offs, lang_index(lang), &spk, &sp, seed_, 0, &ptr, cap, &len, &frames, nullptr) != 0) {
        std::cerr << "[TTSEngine] Synthesis error: " << q3tts_last_error(h_) << std::endl;
        return {};
    }
    pcm.resize((size_t)std::min<int64_t>(len, cap));
    return pcm;
}

std::vector<float> TTSEngine::synthesize_speaker(const std::string& text, Speaker, Language lang, const SamplingParams& params) {
    std::cerr << "[TTSEngine] Preset speakers require CustomVoice model (not yet supported)" << std::endl; // :327
    return synthesize(text, lang, params);
}

std::vector<float> TTSEngine::extract_speaker_embedding(const std::string& audio_path) { // reference tts_onnx.cpp:331-365
    if (!has_speaker_encoder()) return {};
    std::vector<float> embed((size_t)spk_dim_);
    if (q3tts_extract_speaker_embedding_host(h_, audio_path.c_str(), embed.data()) != 0) {
        std::cerr << "[TTSEngine] " << q3tts_last_error(h_) << std::endl; // "Failed to read audio: <path>" / "Failed to extract mel spectrogram"
        return {};
    }
    return embed;
}

std::vector<std::vector<float>> TTSEngine::synthesize_tokens_batch(const std::vector<std::vector<int64_t>>& token_ids,
                                                                   Language lang, const SamplingParams& params) {
    std::vector<std::vector<float>> out(token_ids.size());
    if (!ready_ || token_ids.empty()) return out;
    q3tts_config cfg;
    q3tts_sampling sp{ params.temperature, params.top_p, params.top_k, params.repetition_penalty, params.max_new_tokens };
    std::vector<int64_t> flat;
    std::vector<int32_t> offs(1, 0);
    for (const auto& t : token_ids) { flat.insert(flat.end(), t.begin(), t.end()); offs.push_back((int32_t)flat.size()); }
    // capacity: samples of max_new_tokens frames
    q3tts_default_config("0.6b", &cfg);
    const int64_t cap = (int64_t)params.max_new_tokens * 1920 + 1920;
    std::vector<float*> ptrs(token_ids.size());
    for (size_t i = 0; i < token_ids.size(); ++i) { out[i].resize((size_t)cap); ptrs[i] = out[i].data(); }
    std::vector<int64_t> lens(token_ids.size(), 0);
    std::vector<int32_t> frames(token_ids.size(), 0);
    const int rc = q3tts_synthesize_batch_host(h_, (int)token_ids.size(), flat.data(), offs.data(), lang_index(lang), &sp, seed_, 0,
                                               ptrs.data(), cap, lens.data(), frames.data(), nullptr);
    if (rc != 0) { // reference tts_onnx.cpp:432-435: log, return empty
        std::cerr << "[TTSEngine] Synthesis error: " << q3tts_last_error(h_) << std::endl;
        for (auto& v : out) v.clear();
        return out;
    }
    for (size_t i = 0; i < out.size(); ++i) out[i].resize((size_t)std::min<int64_t>(lens[i], cap));
    return out;
}

int TTSEngine::synthesize_tokens_streaming(const std::vector<int64_t>& token_ids, Language lang, const SamplingParams& params, int chunk_frames,
                                           int left_context_frames, const std::function<void(const float*, size_t)>& on_audio) {
    if (!ready_ || chunk_frames < 1) return -1;
    q3tts_config cfg;
    q3tts_default_config("0.6b", &cfg);
    const int H = cfg_hidden_;
    std::vector<float> prompt((size_t)16 * H), trailing((size_t)1024 * H);
    int S = 0, nt = 0;
    q3tts_sampling sp{ params.temperature, params.top_p, params.top_k, params.repetition_penalty, params.max_new_tokens };
    auto fail = [&]() { std::cerr << "[TTSEngine] Synthesis error: " << q3tts_last_error(h_) << std::endl; (void)q3tts_slot_release(h_, 0); return -1; };
    for (int b = 0; b < max_batch_; ++b) (void)q3tts_slot_release(h_, b);
    if (q3tts_build_prompt_host(h_, token_ids.data(), (int)token_ids.size(), lang_index(lang), nullptr, prompt.data(), &S, trailing.data(), 1024, &nt) != 0) return fail();
    if (q3tts_slot_begin(h_, 0, prompt.data(), S, trailing.data(), nt, &sp, seed_, 0, 0) != 0) return fail();
    std::vector<float> pcm((size_t)chunk_frames * 1920 + 1920);
    int done = 0;
    for (;;) {
        const int want = std::min(chunk_frames, params.max_new_tokens - done);
        const int active = want > 0 ? q3tts_decode_steps(h_, want) : 0;
        if (active < 0) return fail();
        int nf = 0, fin = 0;
        if (q3tts_slot_status(h_, 0, &nf, &fin) != 0) return fail();
        if (nf > done) {
            int64_t n = 0;
            const int ctx = left_context_frames < 0 ? nf : left_context_frames;
            if (q3tts_slot_codec_decode_range_host(h_, 0, done, nf, ctx, pcm.data(), (int64_t)pcm.size(), &n) != 0) return fail();
            on_audio(pcm.data(), (size_t)std::min<int64_t>(n, (int64_t)pcm.size()));
            done = nf;
        }
        if (active == 0 || want <= 0) break;
    }
    (void)q3tts_slot_release(h_, 0);
    return done;
}

std::vector<float> TTSEngine::synthesize_tokens(const std::vector<int64_t>& token_ids, Language lang, const SamplingParams& params) {
    if (!ready_) return {};
    auto r = synthesize_tokens_batch({ token_ids }, lang, params);
    return r.empty() ? std::vector<float>() : std::move(r[0]);
}

} // namespace leaxer_qwen
