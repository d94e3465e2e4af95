// which library call kills rocprofv3 --pmc?  prints progress to stderr
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/q3tts.h"
int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4;
    const int batch = argc > 2 ? atoi(argv[2]) : 1;   // armed slots: 1 = configs[1], 64 = configs[2]
    const int ctx = argc > 3 ? atoi(argv[3]) : 0;     // > 0: the slots are moved to this talker context first (q3tts_measure_skip_frames)
    const int bf16 = argc > 4 ? atoi(argv[4]) : 0;    // 1: Q3TTS_FLAG_KV_BF16
    const char* model = argc > 5 ? argv[5] : "0.6b";  // "1.7b": BASELINE configs[4]'s dims
    q3tts_config c;
    if (q3tts_default_config(model, &c) != 0) { fprintf(stderr, "unknown model %s\n", model); return 1; }
    const size_t H = (size_t)c.hidden, V = (size_t)c.vocab;
    fprintf(stderr, "[1] create\n");
    q3tts_engine* e = q3tts_create(&c, 0, batch, ctx + steps + 64, (ctx > 0 ? Q3TTS_FLAG_TEST_HOOKS : 0u) | (bf16 ? Q3TTS_FLAG_KV_BF16 : 0u));
    if (!e) { fprintf(stderr, "create failed: %s\n", q3tts_last_error(nullptr)); return 1; }
    fprintf(stderr, "[2] fill\n");
    q3tts_fill_synthetic(e, 0);
    fprintf(stderr, "[3] finalize\n");
    q3tts_finalize(e);
    fprintf(stderr, "[4] text_project\n");
    int64_t ids[3] = {1, 2, 3}; std::vector<float> out(3 * H);
    q3tts_text_project_host(e, ids, 3, out.data());
    fprintf(stderr, "[5] prefill\n");
    std::vector<float> x(8 * H, 0.01f), lg(8 * V), lh(H);
    q3tts_talker_prefill_host(e, 0, x.data(), 8, lg.data(), lh.data());
    fprintf(stderr, "[6] decode\n");
    q3tts_talker_decode_host(e, 0, x.data(), lg.data(), lh.data());
    fprintf(stderr, "[7] slot_begin\n");
    q3tts_sampling sp{0.8f, 0.95f, 50, 1.0f, ctx + steps + 8};
    std::vector<float> tr(4 * H, 0.01f);
    for (int b = 0; b < batch; ++b) q3tts_slot_begin(e, b, x.data(), 8, tr.data(), 4, &sp, 1, (uint32_t)b, 1);
    if (ctx > 8) {
        fprintf(stderr, "[7b] skip to context %d\n", ctx);
        if (q3tts_measure_skip_frames(e, ctx - 8) != 0) { fprintf(stderr, "skip failed: %s\n", q3tts_last_error(e)); return 1; }
    }
    fprintf(stderr, "[8] decode_steps\n");
    int act = q3tts_decode_steps(e, steps);
    fprintf(stderr, "[9] active=%d codec\n", act);
    int64_t n = 0;
    if (ctx == 0) {   // the long-context passes count the decode step only
        std::vector<float> pcm((size_t)(steps + 8) * 1920);
        q3tts_slot_codec_decode_host(e, 0, pcm.data(), (int64_t)pcm.size(), &n);
    }
    fprintf(stderr, "[10] done n=%lld\n", (long long)n);
    q3tts_destroy(e);
    return 0;
}
