// leaxer-tts — command line of the MI355X engine.  Flag set of the reference CLI
// (reference src/main_onnx.cpp:60-77, 99-124): -m -p -o --lang --ref --temp --top-k --top-p --max-tokens -h,
// unknown flags ignored, 16-bit mono WAV at 24 kHz (clip to [-1,1], truncate x*32767).  Additions:
// --tokens "id,id,..." (pre-tokenised text between TTS_BOS and TTS_EOS, bypassing vocab.json/merges.txt),
// --seed N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "tts_engine.h"

using namespace leaxer_qwen;

static bool put(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n; }

static int save_wav16(const char* path, const std::vector<float>& audio, uint32_t rate) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    const uint32_t data = (uint32_t)(audio.size() * 2), riff = 36 + data, fmt = 16, bytes_per_s = rate * 2;
    const uint16_t pcm = 1, ch = 1, align = 2, bits = 16;
    bool ok = put(f, "RIFF", 4) && put(f, &riff, 4) && put(f, "WAVEfmt ", 8) && put(f, &fmt, 4) && put(f, &pcm, 2) && put(f, &ch, 2) &&
              put(f, &rate, 4) && put(f, &bytes_per_s, 4) && put(f, &align, 2) && put(f, &bits, 2) && put(f, "data", 4) && put(f, &data, 4);
    std::vector<int16_t> s(audio.size());
    for (size_t i = 0; i < audio.size(); ++i) {
        float v = audio[i] > 1.0f ? 1.0f : (audio[i] < -1.0f ? -1.0f : audio[i]);
        s[i] = (int16_t)(v * 32767.0f);
    }
    ok = ok && put(f, s.data(), s.size() * 2);
    fclose(f);
    return ok ? 0 : -1;
}

static void usage(const char* prog) {
    printf("Usage: %s [options]\n\nQwen3-TTS synthesis on MI355X (HIP)\n\nOptions:\n", prog);
    printf("  -m, --model DIR       model directory holding model.q3w, or synthetic:<seed> / synthetic-1.7b:<seed> (required)\n");
    printf("  -p, --prompt TEXT     text to synthesize (needs vocab.json + merges.txt, see README)\n");
    printf("      --tokens IDS      comma-separated text token ids (framed as IM_START ASSISTANT TTS_BOS ids TTS_EOS IM_END)\n");
    printf("  -o, --output PATH     output WAV file (default: output.wav)\n");
    printf("  --lang LANG           auto, en, zh, ja, ko (default: auto)\n");
    printf("  --ref PATH            reference audio for voice clone (WAV; resampled to 24 kHz, ECAPA speaker encoder on the GPU)\n");
    printf("  --temp FLOAT          temperature (default: 0.8; 0 samples at T=1 like the reference, use --top-k 1 for greedy)\n");
    printf("  --top-k N             top-k (default: 50)\n  --top-p FLOAT         top-p (default: 0.95)\n");
    printf("  --max-tokens N        max codec frames (default: 2048)\n  --seed N              sampling seed (default: 0)\n");
    printf("  --stream-chunk N      with --tokens: decode audio every N frames while generating (same samples as the one-shot decode)\n  -h, --help\n");
}

static Language lang_of(const std::string& s) {
    if (s == "en" || s == "english") return Language::English;
    if (s == "zh" || s == "chinese") return Language::Chinese;
    if (s == "ja" || s == "japanese") return Language::Japanese;
    if (s == "ko" || s == "korean") return Language::Korean;
    return Language::Auto;
}

int main(int argc, char** argv) {
    std::string model, prompt, tokens, output = "output.wav", lang = "auto", ref;
    bool have_prompt = false;
    SamplingParams sp;
    uint64_t seed = 0;
    int stream_chunk = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const bool more = i + 1 < argc;
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if ((a == "-m" || a == "--model") && more) model = argv[++i];
        else if ((a == "-p" || a == "--prompt") && more) { prompt = argv[++i]; have_prompt = true; }
        else if (a == "--tokens" && more) tokens = argv[++i];
        else if ((a == "-o" || a == "--output") && more) output = argv[++i];
        else if (a == "--lang" && more) lang = argv[++i];
        else if (a == "--ref" && more) ref = argv[++i];
        else if (a == "--temp" && more) sp.temperature = (float)atof(argv[++i]);
        else if (a == "--top-k" && more) sp.top_k = atoi(argv[++i]);
        else if (a == "--top-p" && more) sp.top_p = (float)atof(argv[++i]);
        else if (a == "--max-tokens" && more) sp.max_new_tokens = atoi(argv[++i]);
        else if (a == "--seed" && more) seed = strtoull(argv[++i], nullptr, 10);
        else if (a == "--stream-chunk" && more) stream_chunk = atoi(argv[++i]);
    }
    if (model.empty() || (!have_prompt && tokens.empty())) {
        fprintf(stderr, "Error: --model and --prompt (or --tokens) are required\n");
        usage(argv[0]);
        return 1;
    }
    struct stat stbuf;
    if (model.rfind("synthetic:", 0) != 0 && model.rfind("synthetic-1.7b:", 0) != 0 && stat(model.c_str(), &stbuf) != 0) {
        fprintf(stderr, "Error: model directory not found: %s\n", model.c_str());
        return 1;
    }
    printf("Model: %s\n", model.c_str());
    if (have_prompt) printf("Text: %s\n", prompt.c_str());
    if (!ref.empty()) printf("Reference: %s\n", ref.c_str());
    printf("Language: %s\nOutput: %s\n\n", lang.c_str(), output.c_str());

    TTSEngine engine(model);
    if (!engine.is_ready()) { fprintf(stderr, "Error: %s\n", engine.get_error().c_str()); return 1; }
    engine.set_seed(seed);
    printf("Synthesizing...\n");
    std::vector<float> audio;
    if (!ref.empty()) {
        if (!engine.has_speaker_encoder()) { fprintf(stderr, "Error: speaker encoder not available for voice clone\n"); return 1; }
    }
    std::vector<int64_t> ids;
    if (!tokens.empty()) {
        ids = { config::IM_START, config::ASSISTANT, config::TTS_BOS };
        for (char* tok = strtok(&tokens[0], ", "); tok; tok = strtok(nullptr, ", ")) ids.push_back(strtoll(tok, nullptr, 10));
        ids.push_back(config::TTS_EOS);
        ids.push_back(config::IM_END);
    }
    if (!ref.empty() && !ids.empty()) {
        const std::vector<float> spk = engine.extract_speaker_embedding(ref);
        if (spk.empty()) { fprintf(stderr, "[TTSEngine] Failed to extract speaker embedding\n"); }
        else audio = engine.synthesize_tokens_clone(ids, spk, lang_of(lang), sp);
    } else if (!ref.empty()) {
        audio = engine.synthesize_clone(prompt, ref, lang_of(lang), sp);
    } else if (!ids.empty() && stream_chunk > 0) {   // chunks of audio as their frames are generated; the file holds their concatenation
        size_t chunks = 0;
        const int nf = engine.synthesize_tokens_streaming(ids, lang_of(lang), sp, stream_chunk, -1, [&](const float* p, size_t n) {
            if (chunks++ == 0) printf("First %.2f seconds of audio ready\n", (float)n / config::SAMPLE_RATE);
            audio.insert(audio.end(), p, p + n);
        });
        if (nf < 0) audio.clear();
        else printf("Streamed %d frames in %zu chunks\n", nf, chunks);
    } else if (!ids.empty()) {
        audio = engine.synthesize_tokens(ids, lang_of(lang), sp);
    } else audio = engine.synthesize(prompt, lang_of(lang), sp);
    if (audio.empty()) { fprintf(stderr, "Error: synthesis failed\n"); return 1; }
    printf("Generated %.2f seconds of audio\n", (float)audio.size() / config::SAMPLE_RATE);
    if (save_wav16(output.c_str(), audio, config::SAMPLE_RATE) != 0) { fprintf(stderr, "Error: failed to write WAV\n"); return 1; }
    printf("Saved to: %s\n", output.c_str());
    return 0;
}
