// tts_engine.h — the reference's public surface (leaxer_qwen::TTSEngine, reference src/tts_onnx.h:
// 29-105, 118-164, 230-238) re-implemented over libq3tts_hip.so.  Same namespace, type names, method
// names, argument meaning and error behaviour (errors never throw: is_ready()/get_error(), empty
// vector on a failed synthesis, messages on stderr prefixed "[TTSEngine]"), so a program written
// against the reference header compiles against this one.
#ifndef LEAXER_QWEN_TTS_ENGINE_H
#define LEAXER_QWEN_TTS_ENGINE_H

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

struct q3tts_engine;
struct q3tts_tokenizer;

namespace leaxer_qwen {

namespace config { // reference src/tts_onnx.h:29-70
constexpr int HIDDEN_SIZE = 1024, NUM_LAYERS = 28, NUM_KV_HEADS = 8, HEAD_DIM = 128, VOCAB_SIZE = 3072;
constexpr int NUM_CODE_GROUPS = 16, SUBCODE_VOCAB_SIZE = 2048;
constexpr int64_t TTS_BOS = 151672, TTS_EOS = 151673, TTS_PAD = 151671;
constexpr int64_t IM_START = 151644, IM_END = 151645, ASSISTANT = 77091;
constexpr int64_t CODEC_BOS = 2149, CODEC_EOS = 2150, CODEC_PAD = 2148, CODEC_THINK = 2154, CODEC_NOTHINK = 2155;
constexpr int64_t CODEC_THINK_BOS = 2156, CODEC_THINK_EOS = 2157;
constexpr int64_t LANG_ENGLISH = 2050, LANG_CHINESE = 2051, LANG_JAPANESE = 2052, LANG_KOREAN = 2053;
constexpr int MAX_NEW_TOKENS = 2048;
constexpr float DEFAULT_TEMPERATURE = 0.8f, DEFAULT_TOP_P = 0.95f;
constexpr int DEFAULT_TOP_K = 50;
constexpr int SAMPLE_RATE = 24000;
} // namespace config

enum class Language { Auto, English, Chinese, Japanese, Korean };
enum class Speaker { None, Serena, Vivian, Uncle_Fu, Dylan, Eric, Ryan, Aiden, Ono_Anna, Sohee };
Speaker parse_speaker(const std::string& name);

struct SamplingParams {
    float temperature = config::DEFAULT_TEMPERATURE;
    float top_p = config::DEFAULT_TOP_P;
    int top_k = config::DEFAULT_TOP_K;
    float repetition_penalty = 1.0f; // never read, as in the reference
    int max_new_tokens = config::MAX_NEW_TOKENS;
};

class TTSEngine {
public:
    // model_dir: a directory holding `model.q3w` (see q3tts_save_weights_file / tools/pack_weights.py),
    // or the literal "synthetic:<seed>" for seeded random 0.6B weights (benchmarks, smoke tests).
    // vocab.json + merges.txt are looked up where the reference looks (<parent of model_dir>/models/
    // Qwen3-TTS-12Hz-0.6B-Base/, tts_onnx.cpp:110-112), then in model_dir.
    explicit TTSEngine(const std::string& model_dir);
    ~TTSEngine();
    TTSEngine(const TTSEngine&) = delete;
    TTSEngine& operator=(const TTSEngine&) = delete;

    std::vector<float> synthesize(const std::string& text, Language lang = Language::Auto,
                                  const SamplingParams& params = SamplingParams());
    std::vector<float> synthesize_clone(const std::string& text, const std::string& ref_audio_path,
                                        Language lang = Language::Auto, const SamplingParams& params = SamplingParams());
    std::vector<float> synthesize_speaker(const std::string& text, Speaker speaker, Language lang = Language::Auto,
                                          const SamplingParams& params = SamplingParams());
    std::vector<float> synthesize_tokens(const std::vector<int64_t>& token_ids, Language lang = Language::Auto,
                                         const SamplingParams& params = SamplingParams());
    std::vector<float> extract_speaker_embedding(const std::string& audio_path);

    // batch extension: independent utterances share one decode loop (one result per utterance)
    std::vector<std::vector<float>> synthesize_tokens_batch(const std::vector<std::vector<int64_t>>& token_ids,
                                                            Language lang = Language::Auto,
                                                            const SamplingParams& params = SamplingParams());
    std::vector<std::vector<float>> synthesize_batch(const std::vector<std::string>& texts, Language lang = Language::Auto,
                                                     const SamplingParams& params = SamplingParams());
    // streaming extension: `on_audio` receives each chunk's samples as soon as its frames exist (exactly the
    // samples the whole-utterance decode would return for them when left_context_frames covers the history; < 0 = all of it).
    // Returns the number of frames generated, -1 on error.
    int synthesize_tokens_streaming(const std::vector<int64_t>& token_ids, Language lang, const SamplingParams& params, int chunk_frames,
                                    int left_context_frames, const std::function<void(const float*, size_t)>& on_audio);
    void set_seed(uint64_t seed) { seed_ = seed; }
    // ids of `text` from the loaded tokenizer (reference io::tokenize, src/io/tokenizer.h:22)
    std::vector<int32_t> tokenize(const std::string& text) const;

    // clone with a ready speaker embedding (what synthesize_clone does after extract_speaker_embedding)
    std::vector<float> synthesize_tokens_clone(const std::vector<int64_t>& token_ids, const std::vector<float>& speaker_embed,
                                               Language lang = Language::Auto, const SamplingParams& params = SamplingParams());
    bool has_speaker_encoder() const; // true when the weight file carries the spk.* tensors (reference: speaker_encoder.onnx present)
    bool is_ready() const { return ready_; }
    const std::string& get_error() const { return error_msg_; }

private:
    bool wrap_text(const std::string& text, std::vector<int64_t>& ids) const;
    q3tts_engine* h_ = nullptr;
    q3tts_tokenizer* tok_ = nullptr;
    bool ready_ = false;
    std::string error_msg_;
    uint64_t seed_ = 0;
    int max_batch_ = 1;
    int spk_dim_ = 0;
    int cfg_hidden_ = 1024;
};

inline int64_t language_to_codec_id(Language lang) { // reference src/tts_onnx.h:230-238
    switch (lang) {
    case Language::English: return config::LANG_ENGLISH;
    case Language::Chinese: return config::LANG_CHINESE;
    case Language::Japanese: return config::LANG_JAPANESE;
    case Language::Korean: return config::LANG_KOREAN;
    default: return 0;
    }
}

} // namespace leaxer_qwen
#endif
