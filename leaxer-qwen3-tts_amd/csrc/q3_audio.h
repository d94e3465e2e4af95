// q3_audio.h — reference-audio front end of the voice-clone path: RIFF/WAVE reader,
// linear resampler and the 128-bin log-mel extractor whose output feeds the speaker encoder.
// Behavioural contract = reference src/io/wav_reader.cpp:28-164 and src/io/mel.cpp:13-236 as driven by
// TTSEngine::extract_speaker_embedding (src/tts_onnx.cpp:331-365).  Host code: one clip per utterance,
// microseconds to milliseconds of work.
#ifndef Q3_AUDIO_H
#define Q3_AUDIO_H

#include <cstdint>
#include <string>
#include <vector>

namespace q3 {

// mono float samples in [-1, 1); empty on any failure (as the reference).  *sample_rate is set on success.
std::vector<float> read_wav(const std::string& path, int* sample_rate);
std::vector<float> resample_linear(const std::vector<float>& audio, int src_rate, int dst_rate);

struct MelSpec { // the settings of tts_onnx.cpp:347-354
    int sample_rate = 24000, n_fft = 1024, hop = 256, win = 1024, n_mels = 128;
    float fmin = 0.0f, fmax = 12000.0f;
};
// log-mel, layout [n_mels][frames] like the reference's MelExtractor::extract; *frames receives the frame count
std::vector<float> log_mel(const std::vector<float>& audio, const MelSpec& spec, int* frames);

} // namespace q3
#endif
