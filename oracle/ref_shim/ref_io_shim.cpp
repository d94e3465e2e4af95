// extern "C" access to the REFERENCE's own I/O code (compiled from /root/reference/src/io/*.cpp where it
// lies, by oracle/build_ref.py, into oracle/_ref/).  Test infrastructure only: this is the checker the
// product's tokenizer / wav / mel code is compared with.  Nothing here is reference source; it only calls
// the reference's public functions (src/io/tokenizer.h, wav_reader.h, mel.h).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "io/mel.h"
#include "io/tokenizer.h"
#include "io/wav_reader.h"

extern "C" {

int ref_tok_load(const char* vocab, const char* merges) {
    const bool a = leaxer_qwen::io::load_vocab(vocab);
    const bool b = leaxer_qwen::io::load_merges(merges);
    return (a ? 1 : 0) | (b ? 2 : 0);
}
int ref_tok_ready() { return leaxer_qwen::io::is_tokenizer_ready() ? 1 : 0; }
int ref_tokenize(const char* text, int len, int32_t* out, int cap) {
    const std::vector<int32_t> ids = leaxer_qwen::io::tokenize(std::string(text, (size_t)len));
    const int n = (int)ids.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = ids[i];
    return n;
}
int ref_read_wav(const char* path, float* out, int cap, int* sample_rate) {
    int sr = 0;
    const std::vector<float> a = leaxer_qwen::io::read_wav(path, sr);
    *sample_rate = sr;
    const int n = (int)a.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = a[i];
    return n;
}
int ref_resample(const float* in, int n_in, int src_rate, int dst_rate, float* out, int cap) {
    const std::vector<float> a = leaxer_qwen::io::resample(std::vector<float>(in, in + n_in), src_rate, dst_rate);
    const int n = (int)a.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = a[i];
    return n;
}
// mel with the settings TTSEngine::extract_speaker_embedding uses (src/tts_onnx.cpp:347-357)
int ref_mel(const float* audio, int n, float* out, int cap) {
    leaxer_qwen::io::MelConfig c;
    c.sample_rate = 24000; c.n_fft = 1024; c.hop_size = 256; c.win_size = 1024; c.num_mels = 128; c.fmin = 0.0f; c.fmax = 12000.0f;
    leaxer_qwen::io::MelExtractor ex(c);
    const std::vector<float> m = ex.extract(std::vector<float>(audio, audio + n));
    const int k = (int)m.size();
    for (int i = 0; i < k && i < cap; ++i) out[i] = m[i];
    return k;
}

} // extern "C"
