// Sanitizer driver for the host-only parsers of the path (tokenizer files, WAV files, text): build with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I leaxer-qwen3-tts_amd/csrc \
//       tools/host_sanitize.cpp leaxer-qwen3-tts_amd/csrc/q3_bpe.cpp leaxer-qwen3-tts_amd/csrc/q3_audio.cpp -o /tmp/host_sanitize
// and run (tools/host_sanitize.sh does both).  Inputs are seeded random mutations of well-formed files: the parsers read files a user
// supplies (vocab.json, merges.txt, reference audio), so they must survive garbage without touching memory they do not own.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "q3_audio.h"
#include "q3_bpe.h"

static void put(const std::string& path, const std::string& bytes) { std::ofstream f(path, std::ios::binary); f.write(bytes.data(), (std::streamsize)bytes.size()); }

static std::string wav(int rate, int channels, int bits, int fmt, int frames, std::mt19937& rng) {
    std::string d;
    const int bps = bits / 8;
    for (int i = 0; i < frames * channels * bps; ++i) d.push_back((char)(rng() & 0xFF));
    auto u32 = [](uint32_t v) { return std::string((const char*)&v, 4); };
    auto u16 = [](uint16_t v) { return std::string((const char*)&v, 2); };
    std::string fmtc = u16((uint16_t)fmt) + u16((uint16_t)channels) + u32((uint32_t)rate) + u32((uint32_t)(rate * channels * bps)) + u16((uint16_t)(channels * bps)) + u16((uint16_t)bits);
    std::string body = "WAVE" + std::string("fmt ") + u32((uint32_t)fmtc.size()) + fmtc + "LIST" + u32(4) + "abcd" + "data" + u32((uint32_t)d.size()) + d;
    return "RIFF" + u32((uint32_t)body.size()) + body;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    std::mt19937 rng(12345);
    // ---- tokenizer: a small consistent vocab + merges, then mutated copies ----
    std::string vocab = "{";
    for (int b = 0; b < 256; ++b) {
        const std::string& s = q3::BpeTokenizer::symbol((unsigned char)b);
        std::string esc;
        for (unsigned char ch : s) { if (ch == '"' || ch == '\\') { esc.push_back('\\'); esc.push_back((char)ch); } else esc.push_back((char)ch); }
        vocab += (b ? ",\"" : "\"") + esc + "\":" + std::to_string(b);
    }
    vocab += ",\"he\":256,\"ll\":257,\"hell\":258,\"hello\":259,\"\\u0120w\":260,\"\\u4e2d\":261}";
    const std::string merges = "#version: 0.2\nh e\nl l\nhe ll\nhell o\n\xC4\xA0 w\n";
    const std::string texts[] = { "hello world", "", "  multiple   spaces\n\nnewlines\t\ttabs", "123456 7.5e-3 it's we'll I'M", "\xE4\xB8\xAD\xE6\x96\x87 mixed \xF0\x9F\x98\x80 emoji",
                                  std::string("nul\0inside", 10), std::string(5000, 'a'), "\xFF\xFE\xFD broken utf8 \xC3" };
    long long checksum = 0;
    for (int r = 0; r < rounds; ++r) {
        std::string v = vocab, m = merges;
        if (r > 0) {   // round 0 is the clean pair
            for (int k = 0, n = 1 + (int)(rng() % 6); k < n; ++k) {
                std::string& t = (rng() & 1) ? v : m;
                if (t.empty()) continue;
                const size_t pos = rng() % t.size();
                switch (rng() % 4) {
                case 0: t[pos] = (char)(rng() & 0xFF); break;
                case 1: t.erase(pos, 1 + rng() % 8); break;
                case 2: t.insert(pos, std::string(1 + rng() % 4, (char)(rng() & 0xFF))); break;
                default: t.resize(pos); break;
                }
            }
        }
        put(dir + "/hs_vocab.json", v);
        put(dir + "/hs_merges.txt", m);
        q3::BpeTokenizer tk;
        const bool okv = tk.load_vocab(dir + "/hs_vocab.json"), okm = tk.load_merges(dir + "/hs_merges.txt");
        for (const std::string& t : texts) {
            std::vector<int32_t> ids;
            tk.encode(t.data(), t.size(), ids);
            for (int32_t id : ids) checksum += id;
            std::vector<std::pair<uint32_t, uint32_t>> pieces;
            q3::BpeTokenizer::split(t.data(), t.size(), pieces);
            size_t covered = 0;
            for (auto& p : pieces) { if (p.first != covered || p.second <= p.first || p.second > t.size()) { fprintf(stderr, "split: bad piece\n"); return 2; } covered = p.second; }
            if (covered != t.size()) { fprintf(stderr, "split: text not covered\n"); return 2; }
        }
        checksum += okv + 2 * okm;
    }
    // ---- audio: well-formed variants, then mutated / truncated files ----
    for (int r = 0; r < rounds; ++r) {
        const int bits = (int[]){ 8, 16, 24, 32 }[rng() % 4], fmt = (rng() % 5 == 0) ? 3 : 1, ch = 1 + (int)(rng() % 3);
        const int rates[] = { 8000, 16000, 22050, 24000, 44100, 48000 };
        std::string w = wav(rates[rng() % 6], ch, fmt == 3 ? 32 : bits, fmt, (int)(rng() % 3000), rng);
        if (r % 3) {
            for (int k = 0, n = 1 + (int)(rng() % 5); k < n && !w.empty(); ++k) {
                const size_t pos = rng() % w.size();
                switch (rng() % 3) {
                case 0: w[pos] = (char)(rng() & 0xFF); break;
                case 1: w.resize(pos); break;
                default: w.insert(pos, std::string(1 + rng() % 16, (char)(rng() & 0xFF))); break;
                }
            }
        }
        put(dir + "/hs.wav", w);
        int sr = 0;
        std::vector<float> a = q3::read_wav(dir + "/hs.wav", &sr);
        if (!a.empty()) {
            if (sr <= 0) { fprintf(stderr, "read_wav: samples without a rate\n"); return 2; }
            if (sr <= 384000 && a.size() < 200000) {
                std::vector<float> rs = q3::resample_linear(a, sr, 24000);
                int frames = 0;
                std::vector<float> mel = q3::log_mel(rs, q3::MelSpec(), &frames);
                checksum += frames + (long long)mel.size();
            }
        }
    }
    printf("host_sanitize: %d rounds each, checksum %lld, no sanitizer report\n", rounds, checksum);
    return 0;
}
