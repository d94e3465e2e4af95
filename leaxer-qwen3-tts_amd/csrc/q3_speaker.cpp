// q3_speaker.cpp — ECAPA-TDNN speaker encoder on the GPU (voice-clone path).
// Replaces run_speaker_encoder (reference src/tts_onnx.cpp:367-403, an ORT session over speaker_encoder.onnx).
// Network [HINT: transformers qwen2_5_omni ECAPA_TimeDelayNet], pinned by tests/golden/hf_speaker.npz through the
// oracle: TDNN(k5) -> 3 x SE-Res2Net(k3, dilation 2/3/4) -> concat -> TDNN(k1) -> attentive statistics pooling -> 1x1.
#include <algorithm>

#include "q3_engine.h"

namespace q3 {

struct SpkConv {
    float* w = nullptr;       // [k][Cin][Cout]
    const float* b = nullptr;
    int cin = 0, cout = 0, k = 1;
};
struct SpeakerW {
    SpkConv tdnn0, tdnn1[3], res[3][16], tdnn2[3], se1[3], se2[3], mfa, asp_tdnn, asp_conv, fc;
};

void Engine::speaker_free() {
    delete spk;
    spk = nullptr;
}

void Engine::speaker_finalize() {
    if (!has_speaker()) return;
    if (!spk) spk = new SpeakerW();
    auto pack = [&](SpkConv& cv, const std::string& n) {
        const Tensor& w = T(n + ".w");
        cv.cout = (int)w.shape[0]; cv.cin = (int)w.shape[1]; cv.k = (int)w.shape[2];
        if (!cv.w) cv.w = (float*)dmalloc((size_t)w.numel * sizeof(float));
        launch_spk_repack((const float*)w.dev, cv.w, cv.cout, cv.cin, cv.k, stream);
        cv.b = (const float*)T(n + ".b").dev;
    };
    pack(spk->tdnn0, "spk.tdnn0");
    for (int i = 0; i < 3; ++i) {
        const std::string p = "spk.blocks." + std::to_string(i) + ".";
        pack(spk->tdnn1[i], p + "tdnn1");
        for (int j = 0; j < c.spk_scale - 1; ++j) pack(spk->res[i][j], p + "res2net." + std::to_string(j));
        pack(spk->tdnn2[i], p + "tdnn2");
        pack(spk->se1[i], p + "se1");
        pack(spk->se2[i], p + "se2");
    }
    pack(spk->mfa, "spk.mfa");
    pack(spk->asp_tdnn, "spk.asp.tdnn");
    pack(spk->asp_conv, "spk.asp.conv");
    pack(spk->fc, "spk.fc");
}

void Engine::speaker_encode(const float* mel, int T_, float* out) {
    if (!has_speaker()) throw Error("model has no speaker encoder");
    if (!finalized || !spk) throw Error("weights not finalized");
    if (T_ < 5) throw Error("speaker encoder needs at least 5 mel frames (reflect padding), got " + std::to_string(T_));
    if (T_ > 16384) throw Error("reference clip too long for the speaker encoder (more than 16384 mel frames)");
    const int T = T_, SC = c.spk_channels, sub = SC / c.spk_scale, C3 = 3 * SC;
    // one arena per call: the encoder runs once per cloned voice
    const size_t n_floats = (size_t)c.spk_mel * T + (size_t)T * (4 * SC + 2 * C3 + 3 * C3 + c.spk_att + C3) + 2 * SC + c.spk_se + 2 * C3 + 2 * C3 + c.spk_enc_dim + 64;
    float* arena = nullptr;
    Q3_HIP_CHECK(hipMalloc((void**)&arena, n_floats * sizeof(float)));
    struct Free { float* p; ~Free() { (void)hipFree(p); } } guard{ arena };
    float* cur = arena;
    auto take = [&](size_t n) { float* p = cur; cur += n; return p; };
    float* mel_d = take((size_t)c.spk_mel * T);
    float* h = take((size_t)T * SC);
    float* a = take((size_t)T * SC);
    float* r2 = take((size_t)T * SC);
    float* y = take((size_t)T * SC);
    float* cat = take((size_t)T * C3);
    float* mf = take((size_t)T * C3);
    float* att_in = take((size_t)T * 3 * C3);
    float* at = take((size_t)T * c.spk_att);
    float* sc = take((size_t)T * C3);
    float* mean = take(SC);
    float* gate = take(SC);
    float* s1 = take(c.spk_se);
    float* mu3 = take(C3);
    float* sd3 = take(C3);
    float* pooled = take(2 * C3);
    float* out_d = take(c.spk_enc_dim);
    Q3_HIP_CHECK(hipMemcpyAsync(mel_d, mel, (size_t)c.spk_mel * T * sizeof(float), hipMemcpyHostToDevice, stream));

    auto conv = [&](const SpkConv& cv, const float* x, int ldx, const float* x2, int ldx2, int Tn, int dil, int act, float* yo, int ldy, int chan_major = 0) {
        SpkConvArgs g;
        g.x = x; g.ldx = ldx; g.x2 = x2; g.ldx2 = ldx2; g.x_channel_major = chan_major;
        g.T = Tn; g.Cin = cv.cin; g.Cout = cv.cout; g.k = cv.k; g.dil = dil; g.act = act; g.W = cv.w; g.bias = cv.b; g.y = yo; g.ldy = ldy;
        launch_spk_conv(g, stream);
    };
    conv(spk->tdnn0, mel_d, T, nullptr, 0, T, 1, 1, h, SC, 1);
    for (int i = 0; i < 3; ++i) {
        conv(spk->tdnn1[i], h, SC, nullptr, 0, T, 1, 1, a, SC);
        // Res2Net: chunk 0 passes through, chunk 1 = f(chunk 1), chunk j = f(chunk j + out j-1)
        Q3_HIP_CHECK(hipMemcpy2DAsync(r2, (size_t)SC * sizeof(float), a, (size_t)SC * sizeof(float), (size_t)sub * sizeof(float), (size_t)T, hipMemcpyDeviceToDevice, stream));
        for (int j = 1; j < c.spk_scale; ++j)
            conv(spk->res[i][j - 1], a + (size_t)j * sub, SC, j >= 2 ? r2 + (size_t)(j - 1) * sub : nullptr, SC, T, i + 2, 1, r2 + (size_t)j * sub, SC);
        conv(spk->tdnn2[i], r2, SC, nullptr, 0, T, 1, 1, y, SC);
        launch_spk_colstats(y, SC, T, SC, mean, nullptr, stream);
        conv(spk->se1[i], mean, SC, nullptr, 0, 1, 1, 1, s1, c.spk_se);
        conv(spk->se2[i], s1, c.spk_se, nullptr, 0, 1, 1, 0, gate, SC);
        launch_spk_se_gate(y, gate, h, cat + (size_t)i * SC, C3, T, SC, stream);
    }
    conv(spk->mfa, cat, C3, nullptr, 0, T, 1, 1, mf, C3);
    launch_spk_colstats(mf, C3, T, C3, mu3, sd3, stream);
    launch_spk_asp_input(mf, mu3, sd3, att_in, T, C3, stream);
    conv(spk->asp_tdnn, att_in, 3 * C3, nullptr, 0, T, 1, 2, at, c.spk_att);
    conv(spk->asp_conv, at, c.spk_att, nullptr, 0, T, 1, 0, sc, C3);
    launch_spk_asp_pool(sc, mf, T, C3, pooled, stream);
    conv(spk->fc, pooled, 2 * C3, nullptr, 0, 1, 1, 0, out_d, c.spk_enc_dim);
    Q3_HIP_CHECK(hipMemcpyAsync(out, out_d, (size_t)c.spk_enc_dim * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
}

} // namespace q3
