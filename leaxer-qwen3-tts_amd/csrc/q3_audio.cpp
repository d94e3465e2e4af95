// q3_audio.cpp — see q3_audio.h.
#include "q3_audio.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace q3 {

// ---------------------------------------------------------------------------------------------
// RIFF/WAVE (reference wav_reader.cpp:28-143).  Chunks are walked in file order; "fmt " supplies
// format tag / channels / rate / bits, the LAST "data" chunk supplies the samples, everything else is
// skipped by its size (no pad-byte handling, as the reference).  A data chunk that runs past the end of
// the file is zero-filled to its declared size.  Accepted: PCM 8 (unsigned) / 16 / 24 / 32 bit and
// 32-bit IEEE float; channels are averaged.
// ---------------------------------------------------------------------------------------------
namespace {
struct File {
    FILE* f;
    explicit File(const char* p) : f(fopen(p, "rb")) {}
    ~File() { if (f) fclose(f); }
    bool get(void* dst, size_t n) { return fread(dst, 1, n, f) == n; }
};
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }
} // namespace

std::vector<float> read_wav(const std::string& path, int* sample_rate) {
    File fl(path.c_str());
    if (!fl.f) return {};
    uint8_t hdr[12];
    if (!fl.get(hdr, 4) || memcmp(hdr, "RIFF", 4) != 0) return {};
    (void)fl.get(hdr + 4, 4); // RIFF size: not used
    if (!fl.get(hdr + 8, 4) || memcmp(hdr + 8, "WAVE", 4) != 0) return {};

    unsigned tag = 0, channels = 0, bits = 0;
    uint32_t rate = 0;
    std::vector<uint8_t> data;
    for (;;) {
        uint8_t ch[8];
        if (!fl.get(ch, 4) || !fl.get(ch + 4, 4)) break;
        const uint32_t size = le32(ch + 4);
        if (memcmp(ch, "fmt ", 4) == 0) {
            uint8_t fm[16] = { 0 };
            const size_t got = fread(fm, 1, 16, fl.f);
            // fields that were not read keep their previous value, like a sequence of unchecked freads
            if (got >= 2) tag = le16(fm);
            if (got >= 4) channels = le16(fm + 2);
            if (got >= 8) rate = le32(fm + 4);
            if (got >= 16) bits = le16(fm + 14);
            if (size > 16) fseek(fl.f, (long)(size - 16), SEEK_CUR);
        } else if (memcmp(ch, "data", 4) == 0) {
            try {
                data.assign((size_t)size, 0);
            } catch (...) { return {}; }
            if (size) (void)!fread(data.data(), 1, (size_t)size, fl.f);
        } else {
            fseek(fl.f, (long)size, SEEK_CUR);
        }
        if (feof(fl.f)) break;
    }
    if (tag != 1 && tag != 3) return {};
    if (!channels || !rate || !bits) return {};
    const size_t width = bits / 8;
    if (!width) return {};
    *sample_rate = (int)rate;

    const size_t n = data.size() / (channels * width);
    std::vector<float> out;
    out.reserve(n);
    const uint8_t* p = data.data();
    for (size_t i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (unsigned c = 0; c < channels; ++c, p += width) {
            if (tag == 3) {
                if (bits == 32) { float v; memcpy(&v, p, 4); acc += v; }
            } else if (bits == 16) {
                acc += (float)(int16_t)le16(p) / 32768.0f;
            } else if (bits == 24) {
                int32_t v = (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16);
                if (v & 0x800000) v -= 0x1000000;
                acc += (float)v / 8388608.0f;
            } else if (bits == 32) {
                acc += (float)(int32_t)le32(p) / 2147483648.0f;
            } else if (bits == 8) {
                acc += (float)((int)p[0] - 128) / 128.0f;
            }
        }
        out.push_back(acc / (float)channels);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Linear-interpolation resampler (reference wav_reader.cpp:145-164): out[i] samples the input at
// i * src/dst, positions and weights in double, floor(n * dst/src) output samples, no anti-alias filter.
// ---------------------------------------------------------------------------------------------
std::vector<float> resample_linear(const std::vector<float>& a, int src_rate, int dst_rate) {
    if (src_rate == dst_rate || a.empty()) return a;
    const double ratio = (double)dst_rate / src_rate;
    const size_t n_out = (size_t)((double)a.size() * ratio);
    std::vector<float> out(n_out);
    const size_t last = a.size() - 1;
    for (size_t i = 0; i < n_out; ++i) {
        const double pos = (double)i / ratio;
        const size_t k = (size_t)pos;
        const double w = pos - (double)k;
        out[i] = (float)((double)a[k] * (1.0 - w) + (double)a[std::min(k + 1, last)] * w);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// log-mel (reference mel.cpp:13-80 filterbank, :182-236 framing).  NOT librosa's: symmetric Hann
// window, no centre padding, frames = (n - win)/hop + 1 (one zero-padded frame for shorter clips),
// power spectrum, HTK mel scale, triangle corners snapped to FFT bins floor((n_fft+1) f / sr) with
// integer-bin slopes, natural log of (energy + 1e-10).  The corner bins come out of single-precision
// arithmetic in the reference; the same precision is used here so the corners land on the same bins.
// ---------------------------------------------------------------------------------------------
namespace {

struct MelPlan {
    MelSpec spec;
    int n = 0, bins = 0;              // FFT length (power of two >= n_fft), kept bins
    std::vector<float> window;
    std::vector<float> tw_re, tw_im;  // exp(-2 pi i k / n), k < n/2
    std::vector<uint32_t> rev;
    std::vector<int> lo, mid, hi;     // triangle corners per mel band
};

float hz_to_mel(float hz) { return 2595.0f * log10f(1.0f + hz / 700.0f); }
float mel_to_hz(float m) { return 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f); }

MelPlan make_plan(const MelSpec& s) {
    MelPlan p;
    p.spec = s;
    p.n = 1;
    while (p.n < s.n_fft) p.n <<= 1;
    p.bins = s.n_fft / 2 + 1;
    p.window.resize((size_t)s.win);
    for (int i = 0; i < s.win; ++i) p.window[(size_t)i] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * i / (s.win - 1))));
    p.tw_re.resize((size_t)p.n / 2);
    p.tw_im.resize((size_t)p.n / 2);
    for (int k = 0; k < p.n / 2; ++k) {
        p.tw_re[(size_t)k] = (float)cos(-2.0 * M_PI * k / p.n);
        p.tw_im[(size_t)k] = (float)sin(-2.0 * M_PI * k / p.n);
    }
    int lg = 0;
    while ((1 << lg) < p.n) ++lg;
    p.rev.resize((size_t)p.n);
    for (int i = 0; i < p.n; ++i) {
        uint32_t r = 0;
        for (int b = 0; b < lg; ++b) r |= (uint32_t)((i >> b) & 1) << (lg - 1 - b);
        p.rev[(size_t)i] = r;
    }
    const float m_lo = hz_to_mel(s.fmin), m_hi = hz_to_mel(s.fmax);
    std::vector<int> corner((size_t)s.n_mels + 2);
    for (int i = 0; i < s.n_mels + 2; ++i) {
        const float mel = m_lo + (m_hi - m_lo) * i / (s.n_mels + 1);
        const float hz = mel_to_hz(mel);
        corner[(size_t)i] = std::min((int)floorf((s.n_fft + 1) * hz / s.sample_rate), p.bins - 1);
    }
    p.lo.assign(corner.begin(), corner.end() - 2);
    p.mid.assign(corner.begin() + 1, corner.end() - 1);
    p.hi.assign(corner.begin() + 2, corner.end());
    return p;
}

// in-place decimation-in-time radix-2 FFT on bit-reversed input
void fft_pow2(const MelPlan& p, float* re, float* im) {
    const int n = p.n;
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int k = 0; k < half; ++k) {
                const float wr = p.tw_re[(size_t)(k * step)], wi = p.tw_im[(size_t)(k * step)];
                float& ar = re[base + k];
                float& ai = im[base + k];
                float& br = re[base + k + half];
                float& bi = im[base + k + half];
                const float tr = wr * br - wi * bi, ti = wr * bi + wi * br;
                br = ar - tr; bi = ai - ti;
                ar += tr; ai += ti;
            }
        }
    }
}

} // namespace

std::vector<float> log_mel(const std::vector<float>& audio, const MelSpec& spec, int* frames_out) {
    *frames_out = 0;
    if (audio.empty()) return {};
    const MelPlan p = make_plan(spec);
    const long n_audio = (long)audio.size();
    const int frames = n_audio < spec.win ? 1 : (int)((n_audio - spec.win) / spec.hop + 1);
    std::vector<float> out((size_t)spec.n_mels * (size_t)frames);
    std::vector<float> re((size_t)p.n), im((size_t)p.n), power((size_t)p.bins);
    for (int t = 0; t < frames; ++t) {
        const long start = (long)t * spec.hop;
        std::fill(im.begin(), im.end(), 0.0f);
        for (int i = 0; i < p.n; ++i) {
            const long src = start + i;
            const float v = (i < spec.win && i < spec.n_fft && src < n_audio) ? audio[(size_t)src] * p.window[(size_t)i] : 0.0f;
            re[p.rev[(size_t)i]] = v;
        }
        fft_pow2(p, re.data(), im.data());
        const int kept = std::min(p.bins, p.n / 2 + 1);
        for (int k = 0; k < kept; ++k) power[(size_t)k] = re[(size_t)k] * re[(size_t)k] + im[(size_t)k] * im[(size_t)k];
        for (int k = kept; k < p.bins; ++k) power[(size_t)k] = 0.0f;
        for (int m = 0; m < spec.n_mels; ++m) {
            const int a = p.lo[(size_t)m], b = p.mid[(size_t)m], c = p.hi[(size_t)m];
            float e = 0.0f;
            for (int k = a; k < b; ++k) e += (float)(k - a) / (float)(b - a) * power[(size_t)k];
            for (int k = b; k < c; ++k) e += (float)(c - k) / (float)(c - b) * power[(size_t)k];
            out[(size_t)m * (size_t)frames + (size_t)t] = logf(e + 1e-10f);
        }
    }
    *frames_out = frames;
    return out;
}

} // namespace q3
