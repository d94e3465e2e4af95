// q3_codec.cpp — orchestration of the 12 Hz codec decoder on the GPU: the reference's
// run_vocoder / tokenizer12hz_decode.onnx session (src/tts_onnx.cpp:759-776).  [HINT] architecture:
// transformers Qwen3OmniMoeCode2Wav (modeling_qwen3_omni_moe.py:3180-3697), pinned by
// tests/golden/hf_code2wav.npz.  Whole-utterance decode like the reference (tts_onnx.cpp:430):
// with 288 GB of HBM the largest activation (F=2048: 3.9 M samples x 96 ch fp32 = 1.5 GB) needs no
// chunking.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "q3_engine.h"

namespace q3 {

struct PackedConv { const float* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 0; };
struct SnakeP { const float *alpha = nullptr, *beta = nullptr; int C = 0; };

struct CodecW {
    struct Layer { const float *in_norm, *post_norm, *qkv, *o, *gate, *up, *down, *attn_scale, *mlp_scale; };
    std::vector<Layer> layers;
    const float *norm = nullptr, *code_embed = nullptr;
    struct Up { PackedConv tconv; const float *dw_w, *dw_b, *ln_w, *ln_b, *pw1_w, *pw1_b, *pw2_w, *pw2_b, *gamma; };
    std::vector<Up> up;
    PackedConv conv_in, conv_out;
    struct Res { SnakeP a1, a2; PackedConv c1, c2; };
    struct Block { SnakeP act; PackedConv tconv; Res res[3]; };
    std::vector<Block> blocks;
    SnakeP snake_out;
    std::vector<float*> packed; // owned
    // (hi, lo) bf16 planes of every conv / linear weight, keyed by the fp32 pointer the layer list holds (owned)
    // lo_zero: the weight is exact in fp16 (bf16- / fp16-origin): two products instead of three.  cm: the same planes chunk-major
    // ([plane][tap][C_in / 32][C_out][32], ConvArgs::Whc; null when C_in is not a multiple of 32); fh / fl: a 96 x 96 second conv of a
    // fused residual unit in B-fragment order (ConvArgs::W2fh / W2fl; one allocation)
    struct Planes { bf16_t* hi; bf16_t* lo; float scale_inv; bool lo_zero; bf16_t* cm = nullptr; bf16_t* fh = nullptr; bf16_t* fl = nullptr; };
    int n_lo_zero = 0, n_lo_used = 0;
    std::unordered_map<const float*, Planes> planes;
    // SnakeBeta constants [2][C] (exp(alpha) | 1 / (exp(beta) + 1e-9)) keyed by the alpha pointer (owned; split-precision path only)
    std::unordered_map<const float*, float*> snake_pre;
    // run-time workspace: one bump arena per codec stream (lane 0 = the engine's own stream)
    static constexpr int NLANE = 32;   // capacity; `nlane` of them are used (Q3TTS_CODEC_LANES)
    int nlane = 9;                      // lane 0 = the engine's stream (synchronous entry points), the others decode asynchronously
    char* arena[NLANE] = {}; size_t arena_bytes[NLANE] = {};
    hipStream_t lane_stream[NLANE] = {};
    float* pinned[NLANE] = {}; size_t pinned_floats[NLANE] = {};
    float *rope_cos = nullptr, *rope_sin = nullptr; int rope_P = 0;
    int* page_table = nullptr;
    // asynchronous per-utterance decodes over the side lanes: result parked in the lane's pinned buffer until the lane is drained
    struct Item { float* user; int64_t n, cap, off; int64_t* len; };   // off: float offset of the utterance's samples in the lane's pinned buffer
    struct Pending { std::vector<Item> items; int frames = 0; bool busy = false; };
    Pending pend[NLANE];
    int32_t* job_codes = nullptr; size_t job_codes_n = 0;
    const float* dbg_sx = nullptr; const float* dbg_pcm = nullptr; int dbg_T = 0, dbg_C = 0, dbg_nb = 0;   // test hook: the last batched group's final conv
    char* batch_arena = nullptr; size_t batch_arena_bytes = 0;   // batched pre-transformer of a job (codec_pre_batch)
    int* batch_pages = nullptr; int batch_pages_n = 0;           // identity page table, one cache block per utterance   // codes of a scheduler job's finished utterances, [utterance][max_new][groups]
    hipEvent_t lane_done[NLANE] = {};
    hipEvent_t fork = nullptr, win0 = nullptr, win1 = nullptr;
    bool window_open = false, win0_recorded = false;
    int rr = 0;
    int submits = 0, fail_at_submit = 0;   // fault injection for the tests: the fail_at_submit-th submit of a job throws
    // carried-state streaming decode (Engine::codec_stream_*): per stream the pre-transformer's K / V rows of every layer ([layer][k | v]
    // [head][P][d]) and its output rows [cap][hidden]; work buffers of one push in a grow-only arena of their own
    // Both are SLIDING buffers (round 5; they used to hold every position up to the stream's capacity: 268 MB of K / V per stream at
    // max_ctx 2112).  kv: [layer][k | v][head][P][d] with P = pow2 >= 2 (window - 1) + the largest push so far; `kv_rows` rows are stored,
    // the last of them position n_done - 1; when a push would not fit, the newest window - 1 rows move to the front (the only ones a
    // later query can reach).  hpost: [h_cap][hidden], `h_rows` stored, the newest codec_stage_b_context() rows kept the same way.
    struct Stream { bool used = false; int cap = 0, P = 0, pshift = 0, n_done = 0, kv_rows = 0, h_cap = 0, h_rows = 0; float* kv = nullptr; float* hpost = nullptr; };
    std::vector<Stream> streams;
    char* stream_arena = nullptr; size_t stream_arena_bytes = 0;
    std::vector<int> slot_stream;          // slot -> stream id (-1: none): q3tts_slot_codec_decode_range_host's implicit stream
};

void Engine::codec_free() {
    if (!codec) return;
    for (float* p : codec->packed) (void)hipFree(p);
    for (auto& kv : codec->planes) {
        (void)hipFree(kv.second.hi);   // lo lives in the same allocation
        if (kv.second.cm) (void)hipFree(kv.second.cm);
        if (kv.second.fh) (void)hipFree(kv.second.fh);   // fl lives in the same allocation
    }
    for (auto& kv : codec->snake_pre) (void)hipFree(kv.second);
    for (int i = 0; i < CodecW::NLANE; ++i) {
        if (codec->arena[i]) (void)hipFree(codec->arena[i]);
        if (codec->pinned[i]) (void)hipHostFree(codec->pinned[i]);
        if (codec->lane_done[i]) (void)hipEventDestroy(codec->lane_done[i]);
        if (i > 0 && codec->lane_stream[i] && !null_stream) (void)hipStreamDestroy(codec->lane_stream[i]);
    }
    for (hipEvent_t ev : { codec->fork, codec->win0, codec->win1 }) if (ev) (void)hipEventDestroy(ev);
    if (codec->job_codes) (void)hipFree(codec->job_codes);
    if (codec->batch_arena) (void)hipFree(codec->batch_arena);
    if (codec->batch_pages) (void)hipFree(codec->batch_pages);
    if (codec->rope_cos) (void)hipFree(codec->rope_cos);
    if (codec->rope_sin) (void)hipFree(codec->rope_sin);
    if (codec->page_table) (void)hipFree(codec->page_table);
    for (auto& s : codec->streams) { if (s.kv) (void)hipFree(s.kv); if (s.hpost) (void)hipFree(s.hpost); }
    if (codec->stream_arena) (void)hipFree(codec->stream_arena);
    delete codec;
    codec = nullptr;
}

void Engine::codec_plane_stats(int* two_product, int* three_product) const {
    if (two_product) *two_product = codec ? codec->n_lo_zero : 0;
    if (three_product) *three_product = codec ? codec->n_lo_used : 0;
}

void Engine::codec_finalize() {
    codec_free();
    codec = new CodecW;
    CodecW& W = *codec;
    auto fp = [&](const std::string& n) { return (const float*)T(n).dev; };
    auto pack = [&](const std::string& prefix, int cin, int cout, int k, bool transposed) {
        float* out = nullptr;
        Q3_HIP_CHECK(hipMalloc((void**)&out, (size_t)cin * cout * k * sizeof(float)));
        W.packed.push_back(out);
        launch_repack_conv(fp(prefix + ".w"), out, cin, cout, k, transposed ? 1 : 0, stream);
        PackedConv p; p.w = out; p.b = fp(prefix + ".b"); p.cin = cin; p.cout = cout; p.k = k;
        return p;
    };
    auto snake = [&](const std::string& prefix) { SnakeP s; s.alpha = fp(prefix + ".alpha"); s.beta = fp(prefix + ".beta"); s.C = (int)T(prefix + ".alpha").numel; return s; };
    const int CH = c.cd_hidden, D = c.cd_decoder_dim;
    for (int i = 0; i < c.cd_layers; ++i) {
        const std::string p = "cd.layers." + std::to_string(i) + ".";
        CodecW::Layer L;
        L.in_norm = fp(p + "input_norm"); L.post_norm = fp(p + "post_norm"); L.qkv = fp(p + "q_proj"); L.o = fp(p + "o_proj");
        L.gate = fp(p + "gate_proj"); L.up = fp(p + "up_proj"); L.down = fp(p + "down_proj");
        L.attn_scale = fp(p + "attn_scale"); L.mlp_scale = fp(p + "mlp_scale");
        W.layers.push_back(L);
    }
    W.norm = fp("cd.norm"); W.code_embed = fp("cd.code_embed");
    for (int s = 0; s < c.cd_n_up; ++s) {
        const std::string p = "cd.up." + std::to_string(s) + ".";
        CodecW::Up u;
        u.tconv = pack(p + "tconv", CH, CH, c.cd_up_ratios[s], true);
        u.dw_w = fp(p + "cnx.dw.w"); u.dw_b = fp(p + "cnx.dw.b"); u.ln_w = fp(p + "cnx.ln.w"); u.ln_b = fp(p + "cnx.ln.b");
        u.pw1_w = fp(p + "cnx.pw1.w"); u.pw1_b = fp(p + "cnx.pw1.b"); u.pw2_w = fp(p + "cnx.pw2.w"); u.pw2_b = fp(p + "cnx.pw2.b");
        u.gamma = fp(p + "cnx.gamma");
        W.up.push_back(u);
    }
    W.conv_in = pack("cd.dec.conv_in", CH, D, 7, false);
    for (int i = 0; i < c.cd_n_blocks; ++i) {
        const int cin = D >> i, cout = D >> (i + 1), r = c.cd_up_rates[i];
        const std::string p = "cd.dec.blocks." + std::to_string(i) + ".";
        CodecW::Block B;
        B.act = snake(p + "snake");
        B.tconv = pack(p + "tconv", cin, cout, 2 * r, true);
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + "res." + std::to_string(u) + ".";
            B.res[u].a1 = snake(q + "act1"); B.res[u].a2 = snake(q + "act2");
            B.res[u].c1 = pack(q + "conv1", cout, cout, 7, false);
            B.res[u].c2 = pack(q + "conv2", cout, cout, 1, false);
        }
        W.blocks.push_back(B);
    }
    W.snake_out = snake("cd.dec.snake_out");
    W.conv_out = pack("cd.dec.conv_out", D >> c.cd_n_blocks, 1, 7, false);
    if (!(flags & Q3TTS_FLAG_FP32_CODEC)) { // fp16 hi/lo planes for the split-precision matrix-core path
        unsigned* amax_d = (unsigned*)dmalloc(sizeof(unsigned));
        auto split = [&](const float* w, size_t n, int taps = 0, int cout = 0, int cin = 0) {
            if (!w || W.planes.count(w)) return;
            // power-of-two pre-scale: largest |w| lands in [2^11, 2^12) so the low plane of ordinary weights stays normal
            unsigned bits = 0;
            Q3_HIP_CHECK(hipMemsetAsync(amax_d, 0, sizeof(unsigned), stream));
            launch_absmax(w, n, amax_d, stream);
            Q3_HIP_CHECK(hipMemcpyAsync(&bits, amax_d, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            sync();
            float amax;
            memcpy(&amax, &bits, sizeof amax);
            int e = 0;
            if (amax > 0.f && std::isfinite(amax)) (void)frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1)
            const int k = amax > 0.f ? 12 - e : 0;
            bf16_t *hi = nullptr, *lo = nullptr;   // one allocation, lo right behind hi: k_conv_split addresses both planes as base + 32-bit offset
            const size_t np = (n + 7) / 8 * 8;
            Q3_HIP_CHECK(hipMalloc((void**)&hi, 2 * np * sizeof(bf16_t)));
            lo = hi + np;
            launch_split_planes(w, hi, lo, n, ldexpf(1.0f, k), stream);
            unsigned lo_bits = 0;
            Q3_HIP_CHECK(hipMemsetAsync(amax_d, 0, sizeof(unsigned), stream));
            launch_or_mag16(lo, n, amax_d, stream);
            Q3_HIP_CHECK(hipMemcpyAsync(&lo_bits, amax_d, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            sync();
            CodecW::Planes pl{ hi, lo, ldexpf(1.0f, -k), lo_bits == 0 };
            if (taps > 0 && cin % 32 == 0 && (size_t)taps * cout * cin == n) {   // the chunk-major copy for the 32-wide-chunk kernels
                Q3_HIP_CHECK(hipMalloc((void**)&pl.cm, 2 * np * sizeof(bf16_t)));
                launch_repack_planes_cm(hi, pl.cm, taps, cout, cin, np, stream);
            }
            if (taps == 1 && cout == 96 && cin == 96) {                           // a fused unit's second conv: B fragments
                Q3_HIP_CHECK(hipMalloc((void**)&pl.fh, 2 * n * sizeof(bf16_t)));
                pl.fl = pl.fh + n;
                launch_pack_w2_frags(hi, pl.fh, 96, stream);
                launch_pack_w2_frags(lo, pl.fl, 96, stream);
            }
            W.planes[w] = pl;
            (lo_bits == 0 ? W.n_lo_zero : W.n_lo_used) += 1;
        };
        const size_t FFn = (size_t)c.cd_ffn;
        for (const CodecW::Layer& L : W.layers) {
            split(L.qkv, (size_t)3 * CH * CH, 1, 3 * CH, CH); split(L.o, (size_t)CH * CH, 1, CH, CH);
            split(L.gate, FFn * CH, 1, (int)FFn, CH); split(L.up, FFn * CH, 1, (int)FFn, CH); split(L.down, FFn * CH, 1, CH, (int)FFn);
        }
        for (const CodecW::Up& u : W.up) {
            split(u.tconv.w, (size_t)u.tconv.cin * u.tconv.cout * u.tconv.k, u.tconv.k, u.tconv.cout, u.tconv.cin);
            split(u.pw1_w, (size_t)4 * CH * CH, 1, 4 * CH, CH); split(u.pw2_w, (size_t)4 * CH * CH, 1, CH, 4 * CH);
        }
        auto pc = [&](const PackedConv& p) { split(p.w, (size_t)p.cin * p.cout * p.k, p.k, p.cout, p.cin); };
        pc(W.conv_in);
        for (const CodecW::Block& B : W.blocks) { pc(B.tconv); for (int u = 0; u < 3; ++u) { pc(B.res[u].c1); pc(B.res[u].c2); } }
        auto pre = [&](const SnakeP& sp) {   // exp(alpha), 1 / (exp(beta) + 1e-9) once per activation instead of once per 32-row block of every launch
            if (!sp.alpha || !sp.beta || sp.C <= 0 || W.snake_pre.count(sp.alpha)) return;
            float* d = nullptr;
            Q3_HIP_CHECK(hipMalloc((void**)&d, (size_t)2 * sp.C * sizeof(float)));
            launch_snake_pre(sp.alpha, sp.beta, d, sp.C, stream);
            W.snake_pre[sp.alpha] = d;
        };
        for (const CodecW::Block& B : W.blocks) { pre(B.act); for (int u = 0; u < 3; ++u) { pre(B.res[u].a1); pre(B.res[u].a2); } }
        pre(W.snake_out);
    }
    if (const char* ev = getenv("Q3TTS_CODEC_LANES")) W.nlane = std::max(1, std::min((int)CodecW::NLANE, atoi(ev)));
    W.lane_stream[0] = stream;
    for (int i = 1; i < W.nlane; ++i) {
        if (null_stream) W.lane_stream[i] = nullptr;
        else {
            int lo = 0, hi = 0;
            Q3_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            Q3_HIP_CHECK(hipStreamCreateWithPriority(&W.lane_stream[i], hipStreamNonBlocking, lo));
        }
    }
    int zero = 0;
    Q3_HIP_CHECK(hipMalloc((void**)&W.page_table, sizeof(int)));
    Q3_HIP_CHECK(hipMemcpy(W.page_table, &zero, sizeof(int), hipMemcpyHostToDevice));
    sync();
}

static int tconv_out_len(const q3tts_config& c, int T, int k, int s, int* left_out) {
    const int pad = k - s, left = c.cd_tconv_trim == 0 ? pad : 0;
    if (left_out) *left_out = left;
    return (T - 1) * s + k - left - pad;
}

// RoPE tables of the codec's pre-transformer for positions [0, P), oracle formula (fp32 libm); grow-only, shared by every lane and stream
void Engine::codec_rope_tables(int P) {
    CodecW& W = *codec;
    if (W.rope_P >= P) return;
    const int HD = c.cd_head_dim;
    for (int i = 0; i < W.nlane; ++i) Q3_HIP_CHECK(hipStreamSynchronize(W.lane_stream[i])); // tables may be in use
    if (W.rope_cos) (void)hipFree(W.rope_cos);
    if (W.rope_sin) (void)hipFree(W.rope_sin);
    const int half = HD / 2;
    std::vector<float> cs((size_t)P * half), sn((size_t)P * half);
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / powf(c.cd_rope_theta, (float)(2 * i) / (float)HD);
            const float ang = (float)p * inv;
            cs[(size_t)p * half + i] = cosf(ang); sn[(size_t)p * half + i] = sinf(ang);
        }
    Q3_HIP_CHECK(hipMalloc((void**)&W.rope_cos, cs.size() * sizeof(float)));
    Q3_HIP_CHECK(hipMalloc((void**)&W.rope_sin, sn.size() * sizeof(float)));
    Q3_HIP_CHECK(hipMemcpy(W.rope_cos, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
    Q3_HIP_CHECK(hipMemcpy(W.rope_sin, sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice));
    W.rope_P = P;
}

int64_t Engine::codec_run(const int32_t* codes_dev, int F, float** pcm_dev, int lane, const float* h_in, int h_stage, int nbatch, size_t h_ustride) {
    if (!codec) throw Error("codec decoder not finalized");
    CodecW& W = *codec;
    if (lane < 0 || lane >= W.nlane) throw Error("codec: bad lane");
    hipStream_t stream = W.lane_stream[lane]; // shadows the engine stream for every launch below
    if (nbatch < 1 || (nbatch > 1 && !(h_in && h_stage == 2))) throw Error("codec: a batched decode starts from codec_pre_batch's upsampled rows");
    const size_t nbz = (size_t)nbatch;   // sequences per launch of the conv decoder (rows [sequence][T] everywhere)
    const int CH = c.cd_hidden, NH = c.cd_heads, HD = c.cd_head_dim, FF = c.cd_ffn, D = c.cd_decoder_dim;
    if (NH * HD != CH) throw Error("codec: heads*head_dim must equal hidden");
    int P = 1, pshift = 0;
    while (P < F) { P <<= 1; ++pshift; }
    codec_rope_tables(P);

    int64_t n_pcm = 0;
    float* pcm = nullptr;
    for (int pass = 0; pass < 2; ++pass) { // pass 0 sizes the arena, pass 1 launches
        const bool plan = pass == 0;
        size_t off = 0;
        auto take = [&](size_t nfloat) -> float* {
            const size_t bytes = (nfloat * sizeof(float) + 255) & ~(size_t)255;
            float* p = plan ? nullptr : (float*)(W.arena[lane] + off);
            off += bytes;
            return p;
        };
        const size_t kslab_floats = (size_t)32 * 128 * 4096;   // split-K partial sums of the short pre-transformer / ConvNeXt GEMMs
        float* kslab = take(kslab_floats);
        auto conv = [&](ConvArgs a) {
            if (plan) return;
            a.slab = kslab; a.slab_floats = kslab_floats; a.batch = nbatch;
            const auto it = W.planes.find(a.W);
            if (it != W.planes.end()) { a.Wh = it->second.hi; a.Wl = it->second.lo; a.Whc = it->second.cm; a.w_scale_inv = it->second.scale_inv; a.w_lo_zero = it->second.lo_zero; }
            if (a.snake_alpha) { const auto sp = W.snake_pre.find(a.snake_alpha); if (sp != W.snake_pre.end()) a.snake_pre = sp->second; }
            if (a.mid_alpha) { const auto sp = W.snake_pre.find(a.mid_alpha); if (sp != W.snake_pre.end()) a.mid_pre = sp->second; }
            launch_conv(a, stream);
        };
        auto gemm = [&](const float* in, int T, int Cin, const float* Wm, const float* bias, int Cout, float* out) {
            ConvArgs a; a.in = in; a.T_in = T; a.C_in = Cin; a.out = out; a.T_out = T; a.C_out = Cout; a.W = Wm; a.bias = bias;
            return a;
        };
        // ---- code embedding mean + pre-transformer ----
        float* h = take((size_t)F * CH);
        float* hn = take((size_t)F * CH);
        float* qkvb = take((size_t)F * 3 * CH);
        float* att = take((size_t)F * CH);
        float* ub = take((size_t)F * FF);
        float* gb = take((size_t)F * FF);
        float* kc = take((size_t)NH * P * HD);
        float* vc = take((size_t)NH * P * HD);
        if (!plan && !h_in) launch_code_embed_mean(W.code_embed, codes_dev, F, c.n_groups, c.cd_codebook, CH, h, stream);
        for (int l = 0; l < (h_in ? 0 : c.cd_layers); ++l) {   // h_in: the pre-transformer of this utterance already ran in codec_pre_batch
            const CodecW::Layer& L = W.layers[l];
            if (!plan) launch_rmsnorm_rows(h, L.in_norm, c.cd_rms_eps, F, CH, hn, stream);
            conv(gemm(hn, F, CH, L.qkv, nullptr, 3 * CH, qkvb));
            if (!plan) {
                launch_rope_store(qkvb, 3 * CH, F, NH, NH, HD, W.rope_cos, W.rope_sin, kc, vc, P, stream);
                AttnArgs a;
                a.qkv = qkvb; a.ld_qkv = 3 * CH; a.out = att; a.ld_out = CH; a.kcache = kc; a.vcache = vc;
                a.page_table = W.page_table; a.pages_per_slot = 1; a.page_shift = pshift; a.layer = 0; a.n_layers = 1;
                a.pos_scalar = 0; a.slot_offset = 0; a.nb = 1; a.n_new = F; a.nq = NH; a.nkv = NH; a.d = HD;
                a.scale = 1.0f / sqrtf((float)HD); a.window = c.cd_window; a.new_from_raw = 0;
                launch_attn(a, stream);
            }
            { ConvArgs a = gemm(att, F, CH, L.o, nullptr, CH, h); a.res_scale = L.attn_scale; a.res = h; conv(a); }
            if (!plan) launch_rmsnorm_rows(h, L.post_norm, c.cd_rms_eps, F, CH, hn, stream);
            conv(gemm(hn, F, CH, L.up, nullptr, FF, ub));
            { ConvArgs a = gemm(hn, F, CH, L.gate, nullptr, FF, gb); a.act = 2; a.mul = ub; conv(a); }
            { ConvArgs a = gemm(gb, F, FF, L.down, nullptr, CH, h); a.res_scale = L.mlp_scale; a.res = h; conv(a); }
        }
        if (!plan && !h_in) launch_rmsnorm_rows(h, W.norm, c.cd_rms_eps, F, CH, h, stream);
        // ---- ConvNeXt upsampling stages ----
        float* cur = h_in ? const_cast<float*>(h_in) : h;
        int Tc = F;
        if (h_in && h_stage == 2)   // the ConvNeXt upsampling stages ran in codec_pre_batch too
            for (int s = 0; s < c.cd_n_up; ++s) Tc = tconv_out_len(c, Tc, c.cd_up_ratios[s], c.cd_up_ratios[s], nullptr);
        for (int s = 0; s < (h_in && h_stage == 2 ? 0 : c.cd_n_up); ++s) {
            const CodecW::Up& U = W.up[s];
            const int f = c.cd_up_ratios[s];
            int left = 0;
            const int To = tconv_out_len(c, Tc, f, f, &left);
            float* y = take((size_t)To * CH);
            float* ln = take((size_t)To * CH);
            float* a4 = take((size_t)To * 4 * CH);
            { ConvArgs a; a.in = cur; a.T_in = Tc; a.C_in = CH; a.out = y; a.T_out = To; a.C_out = CH; a.W = U.tconv.w; a.bias = U.tconv.b;
              a.taps = f; a.transposed = 1; a.stride = f; a.left = left; conv(a); }
            if (!plan) launch_dwconv_ln(y, To, CH, U.dw_w, U.dw_b, U.ln_w, U.ln_b, ln, stream);
            { ConvArgs a = gemm(ln, To, CH, U.pw1_w, U.pw1_b, 4 * CH, a4); a.act = 1; conv(a); }
            { ConvArgs a = gemm(a4, To, 4 * CH, U.pw2_w, U.pw2_b, CH, y); a.res_scale = U.gamma; a.res = y; conv(a); }
            cur = y; Tc = To;
        }
        // ---- SnakeBeta decoder: conv_in, 4 x (snake, transposed conv, 3 residual units), snake, conv_out ----
        // SnakeBeta outputs travel between the decoder's convs as (hi, lo) fp16 pairs (ConvArgs::in_planes / out2_planes): the producer's
        // epilogue splits each element once, the consumers stage it without converting.  Needs every decoder conv on the split-precision
        // path with 96-multiple widths (0.6B / 1.7B: 1536 .. 96); the last activation feeds the fp32 C_out = 1 conv and stays fp32.
        const bool fp32_act = knob("Q3TTS_CONV_FP32_ACT") != nullptr || knob("Q3TTS_CONV_GENERIC_EPILOGUE") != nullptr;   // A/B switches
        bool act_planes = !fp32_act && !W.planes.empty() && CH % 32 == 0 && W.planes.count(W.conv_in.w) && !W.snake_pre.empty();
        for (int i = 0, Cw = D; i <= c.cd_n_blocks; ++i, Cw /= 2) act_planes = act_planes && Cw % 96 == 0;
        for (const CodecW::Block& B : W.blocks) {
            act_planes = act_planes && W.planes.count(B.tconv.w);
            for (int u = 0; u < 3; ++u) act_planes = act_planes && W.planes.count(B.res[u].c1.w) && W.planes.count(B.res[u].c2.w);
        }
        float* x = take(nbz * Tc * D);
        float* sx = take(nbz * Tc * D);
        { ConvArgs a; a.in = cur; a.T_in = Tc; a.C_in = CH; a.out = x; a.T_out = Tc; a.C_out = D; a.W = W.conv_in.w; a.bias = W.conv_in.b; a.taps = 7;
          a.in_ustride = nbatch > 1 ? h_ustride : 0;   // the batched front pads every utterance to the job's longest; a group runs at its own longest
          a.out2 = sx; a.snake_alpha = W.blocks[0].act.alpha; a.snake_beta = W.blocks[0].act.beta; a.out2_planes = act_planes; conv(a); }
        int C = D;
        static const int dil[3] = { 1, 3, 9 };
        for (int i = 0; i < c.cd_n_blocks; ++i) {
            const CodecW::Block& B = W.blocks[i];
            const int r = c.cd_up_rates[i], Co = C / 2;
            int left = 0;
            const int To = tconv_out_len(c, Tc, 2 * r, r, &left);
            float* nx = take(nbz * To * Co);
            float* ns = take(nbz * To * Co);
            float* nt = take(nbz * To * Co);
            { ConvArgs a; a.in = sx; a.T_in = Tc; a.C_in = C; a.out = nx; a.T_out = To; a.C_out = Co; a.W = B.tconv.w; a.bias = B.tconv.b;
              a.taps = 2 * r; a.transposed = 1; a.stride = r; a.left = left; a.in_planes = act_planes;
              a.out2 = ns; a.snake_alpha = B.res[0].a1.alpha; a.snake_beta = B.res[0].a1.beta; a.out2_planes = act_planes; conv(a); }
            for (int u = 0; u < 3; ++u) {
                const CodecW::Res& R = B.res[u];
                const SnakeP nxt = u < 2 ? B.res[u + 1].a1 : (i + 1 < c.cd_n_blocks ? W.blocks[i + 1].act : W.snake_out);
                const bool nxt_planes = act_planes && !(u == 2 && i + 1 == c.cd_n_blocks);   // the very last activation feeds the fp32 conv_out
                // 96-channel block: the whole residual unit (7-tap conv, SnakeBeta, 1x1 conv, + x) in one launch — the intermediate stays in
                // the CU (k_conv_split<..., F2>): 4 activation passes through HBM instead of 6, one launch instead of two.  The unit reads
                // snake(x) with a tap halo reaching into its neighbours' rows, so the next layer's snake(x') goes to the OTHER buffer
                // (ns <-> nt); x itself is updated in place (every element is read and written by the same thread).
                const auto p2 = W.planes.find(R.c2.w);
                if (Co == 96 && p2 != W.planes.end() && W.planes.count(R.c1.w) && !knob("Q3TTS_NO_FUSED_RES")) {
                    ConvArgs a; a.in = ns; a.T_in = To; a.C_in = Co; a.out = nx; a.T_out = To; a.C_out = Co; a.W = R.c1.w; a.bias = R.c1.b;
                    a.taps = 7; a.dil = dil[u]; a.mid_alpha = R.a2.alpha; a.mid_beta = R.a2.beta;
                    a.W2 = R.c2.w; a.W2h = p2->second.hi; a.W2l = p2->second.lo; a.W2fh = p2->second.fh; a.W2fl = p2->second.fl; a.w2_scale_inv = p2->second.scale_inv; a.w2_lo_zero = p2->second.lo_zero; a.bias2 = R.c2.b;
                    a.res = nx; a.out2 = nt; a.snake_alpha = nxt.alpha; a.snake_beta = nxt.beta;
                    a.in_planes = act_planes; a.out2_planes = nxt_planes;
                    conv(a);
                    std::swap(ns, nt);
                    continue;
                }
                { ConvArgs a; a.in = ns; a.T_in = To; a.C_in = Co; a.out = nullptr; a.T_out = To; a.C_out = Co; a.W = R.c1.w; a.bias = R.c1.b;
                  a.taps = 7; a.dil = dil[u]; a.out2 = nt; a.snake_alpha = R.a2.alpha; a.snake_beta = R.a2.beta;
                  a.in_planes = act_planes; a.out2_planes = act_planes; conv(a); }
                { ConvArgs a; a.in = nt; a.T_in = To; a.C_in = Co; a.out = nx; a.T_out = To; a.C_out = Co; a.W = R.c2.w; a.bias = R.c2.b;
                  a.taps = 1; a.res = nx; a.out2 = ns; a.snake_alpha = nxt.alpha; a.snake_beta = nxt.beta;
                  a.in_planes = act_planes; a.out2_planes = nxt_planes; conv(a); }
            }
            x = nx; sx = ns; Tc = To; C = Co;
        }
        pcm = take(nbz * Tc);
        { ConvArgs a; a.in = sx; a.T_in = Tc; a.C_in = C; a.out = pcm; a.T_out = Tc; a.C_out = 1; a.W = W.conv_out.w; a.bias = W.conv_out.b;
          a.taps = 7; a.clamp = 1; conv(a); }
        if (!plan && nbatch > 1) { W.dbg_sx = sx; W.dbg_pcm = pcm; W.dbg_T = Tc; W.dbg_C = C; W.dbg_nb = nbatch; }
        n_pcm = Tc;
        if (plan && off > W.arena_bytes[lane]) {
            Q3_HIP_CHECK(hipStreamSynchronize(stream));
            if (W.arena[lane]) (void)hipFree(W.arena[lane]);
            W.arena[lane] = nullptr;
            Q3_HIP_CHECK(hipMalloc((void**)&W.arena[lane], off));
            W.arena_bytes[lane] = off;
        }
    }
    *pcm_dev = pcm;
    return n_pcm;
}


// ------------------------------------------------------------------------------------------------
// Pre-transformer of a whole job in one pass: the utterances' frames are laid out as [utterance][Fp] rows (Fp = the longest, shorter
// ones padded: attention is causal, so padding rows never reach a real one), every linear layer is ONE matrix-core GEMM over all rows
// (weights streamed once, full grids) instead of one short GEMM per utterance, attention runs with the utterance as its batch
// dimension.  Returns the normalised hidden rows [n][Fp][cd_hidden] (valid until the next call); codes: [n][codes_stride_frames][groups].
// ------------------------------------------------------------------------------------------------
const float* Engine::codec_pre_batch(const int32_t* codes_dev, int codes_stride_frames, int n, int Fp, bool with_upsampling, int* rows_per_utt_out,
                                     const int* perm_host) {
    if (!codec) throw Error("codec decoder not finalized");
    CodecW& W = *codec;
    const int CH = c.cd_hidden, NH = c.cd_heads, HD = c.cd_head_dim, FF = c.cd_ffn;
    int P = 1, pshift = 0;
    while (P < Fp) { P <<= 1; ++pshift; }
    if (W.rope_P < P) throw Error("codec_pre_batch: RoPE tables not prepared");
    const size_t rows = (size_t)n * Fp;
    auto bytes_of = [](size_t nfloat) { return (nfloat * sizeof(float) + 255) & ~(size_t)255; };
    const size_t kslab_floats = (size_t)32 * 128 * 4096;
    size_t need = bytes_of(rows * CH) * 3 + bytes_of(rows * 3 * CH) + bytes_of(rows * FF) * 2 + bytes_of((size_t)n * NH * P * HD) * 2 + bytes_of(kslab_floats);
    if (with_upsampling) {   // per stage: y, ln (CH each) and the 4x wide hidden, at the stage's output rate
        size_t r = rows;
        for (int s2 = 0; s2 < c.cd_n_up; ++s2) { r *= (size_t)c.cd_up_ratios[s2]; need += bytes_of(r * CH) * 2 + bytes_of(r * 4 * CH); }
    }
    if (W.batch_arena_bytes < need) {
        sync();
        if (W.batch_arena) (void)hipFree(W.batch_arena);
        W.batch_arena = nullptr;
        Q3_HIP_CHECK(hipMalloc((void**)&W.batch_arena, need));
        W.batch_arena_bytes = need;
    }
    if (W.batch_pages_n < n) {   // [0, n): identity page table; [n_cap, 2 n_cap): the job's utterance order (perm)
        sync();
        if (W.batch_pages) (void)hipFree(W.batch_pages);
        std::vector<int> idt((size_t)n);
        for (int i = 0; i < n; ++i) idt[(size_t)i] = i;
        Q3_HIP_CHECK(hipMalloc((void**)&W.batch_pages, (size_t)2 * n * sizeof(int)));
        Q3_HIP_CHECK(hipMemcpy(W.batch_pages, idt.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        W.batch_pages_n = n;
    }
    const int* perm_d = nullptr;
    if (perm_host) {   // row block y of every buffer holds utterance perm[y] (the scheduler sorts by length so that decoder groups pad little)
        Q3_HIP_CHECK(hipMemcpyAsync(W.batch_pages + W.batch_pages_n, perm_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
        perm_d = W.batch_pages + W.batch_pages_n;
    }
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = (float*)(W.batch_arena + off); off += bytes_of(nfloat); return p; };
    float* h = take(rows * CH);
    float* hn = take(rows * CH);
    float* att = take(rows * CH);
    float* qkvb = take(rows * 3 * CH);
    float* ub = take(rows * FF);
    float* gb = take(rows * FF);
    float* kc = take((size_t)n * NH * P * HD);
    float* vc = take((size_t)n * NH * P * HD);
    float* kslab = take(kslab_floats);
    const int T = (int)rows;
    if (!W.win0_recorded) { Q3_HIP_CHECK(hipEventRecord(W.win0, stream)); W.win0_recorded = true; }   // the vocoder window starts here
    auto conv = [&](ConvArgs a) {
        a.slab = kslab; a.slab_floats = kslab_floats;
        const auto it = W.planes.find(a.W);
        if (it != W.planes.end()) { a.Wh = it->second.hi; a.Wl = it->second.lo; a.Whc = it->second.cm; a.w_scale_inv = it->second.scale_inv; a.w_lo_zero = it->second.lo_zero; }
        if (a.snake_alpha) { const auto sp = W.snake_pre.find(a.snake_alpha); if (sp != W.snake_pre.end()) a.snake_pre = sp->second; }
        launch_conv(a, stream);
    };
    auto gemm = [&](const float* in, int Cin, const float* Wm, int Cout, float* out) {
        ConvArgs a; a.in = in; a.T_in = T; a.C_in = Cin; a.out = out; a.T_out = T; a.C_out = Cout; a.W = Wm;
        return a;
    };
    launch_code_embed_mean(W.code_embed, codes_dev, Fp, c.n_groups, c.cd_codebook, CH, h, stream, n, (size_t)codes_stride_frames * c.n_groups, perm_d);
    for (int l = 0; l < c.cd_layers; ++l) {
        const CodecW::Layer& L = W.layers[l];
        launch_rmsnorm_rows(h, L.in_norm, c.cd_rms_eps, T, CH, hn, stream);
        conv(gemm(hn, CH, L.qkv, 3 * CH, qkvb));
        launch_rope_store(qkvb, 3 * CH, Fp, NH, NH, HD, W.rope_cos, W.rope_sin, kc, vc, P, stream, n);
        AttnArgs a;
        a.qkv = qkvb; a.ld_qkv = 3 * CH; a.out = att; a.ld_out = CH; a.kcache = kc; a.vcache = vc;
        a.page_table = W.batch_pages; a.pages_per_slot = 1; a.page_shift = pshift; a.layer = 0; a.n_layers = 1;
        a.pos_scalar = 0; a.slot_offset = 0; a.nb = n; a.n_new = Fp; a.nq = NH; a.nkv = NH; a.d = HD;
        a.scale = 1.0f / sqrtf((float)HD); a.window = c.cd_window; a.new_from_raw = 0;
        launch_attn(a, stream);
        { ConvArgs g = gemm(att, CH, L.o, CH, h); g.res_scale = L.attn_scale; g.res = h; conv(g); }
        launch_rmsnorm_rows(h, L.post_norm, c.cd_rms_eps, T, CH, hn, stream);
        conv(gemm(hn, CH, L.up, FF, ub));
        { ConvArgs g = gemm(hn, CH, L.gate, FF, gb); g.act = 2; g.mul = ub; conv(g); }
        { ConvArgs g = gemm(gb, FF, L.down, CH, h); g.res_scale = L.mlp_scale; g.res = h; conv(g); }
    }
    launch_rmsnorm_rows(h, W.norm, c.cd_rms_eps, T, CH, h, stream);
    if (rows_per_utt_out) *rows_per_utt_out = Fp;
    if (!with_upsampling) return h;
    // ---- ConvNeXt upsampling stages over all utterances: the transposed convs have kernel == stride (every input row expands on its
    // own), the depthwise causal conv stops at each utterance's first row, the pointwise layers are plain GEMMs ----
    float* cur = h;
    int Tc = T, per = Fp;
    for (int s2 = 0; s2 < c.cd_n_up; ++s2) {
        const CodecW::Up& U = W.up[s2];
        const int f = c.cd_up_ratios[s2];
        if (U.tconv.k != f) throw Error("codec_pre_batch: upsampling kernel != stride");
        const int To = Tc * f;
        float* y = take((size_t)To * CH);
        float* ln = take((size_t)To * CH);
        float* a4 = take((size_t)To * 4 * CH);
        { ConvArgs a; a.in = cur; a.T_in = Tc; a.C_in = CH; a.out = y; a.T_out = To; a.C_out = CH; a.W = U.tconv.w; a.bias = U.tconv.b;
          a.taps = f; a.transposed = 1; a.stride = f; a.left = 0; conv(a); }
        per *= f;
        launch_dwconv_ln(y, To, CH, U.dw_w, U.dw_b, U.ln_w, U.ln_b, ln, stream, per);
        { ConvArgs a; a.in = ln; a.T_in = To; a.C_in = CH; a.out = a4; a.T_out = To; a.C_out = 4 * CH; a.W = U.pw1_w; a.bias = U.pw1_b; a.act = 1; conv(a); }
        { ConvArgs a; a.in = a4; a.T_in = To; a.C_in = 4 * CH; a.out = y; a.T_out = To; a.C_out = CH; a.W = U.pw2_w; a.bias = U.pw2_b; a.res_scale = U.gamma; a.res = y; conv(a); }
        cur = y; Tc = To;
    }
    if (rows_per_utt_out) *rows_per_utt_out = per;
    return cur;
}

// ------------------------------------------------------------------------------------------------
// Vocoder side of the scheduler (q3tts_synthesize_schedule_host).  A slot that finishes copies its codes into the job's buffer on the
// engine stream (codec_stash) and is free at once; when the decode queue has run dry the utterances are vocoded over the side lanes
// (codec_async_submit_dev / codec_async_drain), their small-grid kernels overlapping each other.  Vocoding while other slots still
// decode was measured and dropped: the conv kernels own whole CUs and stream gigabytes through L2 / MALL, so every decode step of the
// latency-bound chain got 12 % longer and the job 10 % slower than with the two phases back to back (profiles/r01_negative_results.txt).
// ------------------------------------------------------------------------------------------------
// Forget every queued vocoder result without delivering it: after a failure inside the vocoder phase the pending items still hold the
// failed job's host pointers (pcm_out rows, pcm_len entries), which a later job's drain must never write through.
void Engine::codec_async_abort() {
    if (!codec) return;
    CodecW& W = *codec;
    for (int i = 0; i < W.nlane; ++i) {
        if (i > 0 && W.lane_stream[i]) (void)hipStreamSynchronize(W.lane_stream[i]);
        W.pend[i].items.clear();
        W.pend[i].frames = 0;
        W.pend[i].busy = false;
    }
    if (stream || null_stream) (void)hipStreamSynchronize(stream);
    W.window_open = false; W.win0_recorded = false;
    W.submits = 0;
}

void Engine::codec_async_prepare(int max_frames, int n_utt) {
    if (!codec) throw Error("codec decoder not finalized");
    CodecW& W = *codec;
    codec_async_abort();   // a previous job that failed mid-way must not leak its pending items into this one
    // fault injection for the tests (tests/test_gpu_edges.py): only an engine created with Q3TTS_FLAG_TEST_HOOKS reads the variable, so a
    // stray environment setting cannot fail a production job
    const char* fv = (flags & Q3TTS_FLAG_TEST_HOOKS) ? getenv("Q3TTS_TEST_FAIL_VOCODER_SUBMIT") : nullptr;
    W.fail_at_submit = fv ? atoi(fv) : 0;
    int P = 1;
    while (P < max_frames) P <<= 1;
    if (W.rope_P < P) {   // grow the shared RoPE tables before any lane is in flight
        int32_t* tmp = nullptr; float* dummy = nullptr;
        Q3_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)max_frames * c.n_groups * sizeof(int32_t)));
        Q3_HIP_CHECK(hipMemset(tmp, 0, (size_t)max_frames * c.n_groups * sizeof(int32_t)));
        codec_run(tmp, max_frames, &dummy, 0);
        sync();
        (void)hipFree(tmp);
    }
    const size_t need = (size_t)n_utt * max_frames * c.n_groups;
    if (W.job_codes_n < need) {
        sync();
        if (W.job_codes) (void)hipFree(W.job_codes);
        Q3_HIP_CHECK(hipMalloc((void**)&W.job_codes, need * sizeof(int32_t)));
        W.job_codes_n = need;
    }
    W.rr = 0;   // a job's first utterances always land on the same lanes (their arenas are already sized)
    W.win0_recorded = false;
    if (!W.fork) { Q3_HIP_CHECK(hipEventCreateWithFlags(&W.fork, hipEventDisableTiming)); Q3_HIP_CHECK(hipEventCreate(&W.win0)); Q3_HIP_CHECK(hipEventCreate(&W.win1)); }
}

// slot's first nf frames -> row `utt` of the job buffer (stride = row_frames frames); ordered on the engine stream, so the slot may be re-armed next
const int32_t* Engine::codec_stash(int slot, int nf, int utt, int row_frames) {
    CodecW& W = *codec;
    const int G = c.n_groups;
    int32_t* dst = W.job_codes + (size_t)utt * row_frames * G;
    if (nf > 0) Q3_HIP_CHECK(hipMemcpyAsync(dst, codes_d + (size_t)slot * max_frames_cap * G, (size_t)nf * G * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    return dst;
}

// host codes [n_utt][row_frames][n_groups] -> the job buffer (after codec_async_prepare(row_frames, n_utt)); returns once they are in HBM
void Engine::codec_job_upload(const int32_t* host, int n_utt, int row_frames) {
    CodecW& W = *codec;
    const size_t n = (size_t)n_utt * row_frames * c.n_groups;
    if (n > W.job_codes_n) throw Error("codec_job_upload: job buffer not prepared");
    Q3_HIP_CHECK(hipMemcpyAsync(W.job_codes, host, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    sync();
}

const int32_t* Engine::codec_job_codes(int utt, int row_frames) { return codec->job_codes + (size_t)utt * row_frames * c.n_groups; }

void Engine::codec_async_drain_lane(int lane) {
    CodecW& W = *codec;
    CodecW::Pending& p = W.pend[lane];
    if (!p.busy) return;
    Q3_HIP_CHECK(hipStreamSynchronize(W.lane_stream[lane]));
    for (const CodecW::Item& it : p.items) {
        const int64_t m = std::min(it.n, it.cap);
        if (it.user && m > 0) memcpy(it.user, W.pinned[lane] + it.off, (size_t)m * sizeof(float));
        if (it.len) *it.len = it.n;
    }
    total_codec_frames += p.frames;
    p.items.clear();
    p.busy = false;
}

void Engine::codec_async_submit_dev(const int32_t* codes_dev, int nf, float* user_pcm, int64_t cap, int64_t* len_out, const float* h_in, int h_stage) {
    CodecW& W = *codec;
    if (len_out) *len_out = 0;
    if (nf <= 0) return;   // the reference returns an empty vector when no frame was generated (tts_onnx.cpp:418)
    const int lane = W.nlane > 1 ? 1 + (W.rr++ % (W.nlane - 1)) : 0;
    codec_async_drain_lane(lane);
    if (W.fail_at_submit > 0 && ++W.submits == W.fail_at_submit) throw Error("vocoder submit failed (injected by Q3TTS_TEST_FAIL_VOCODER_SUBMIT)");
    if (!W.window_open) {   // the lanes start behind everything the engine stream has queued (the stashed codes among it)
        if (!W.win0_recorded) { Q3_HIP_CHECK(hipEventRecord(W.win0, stream)); W.win0_recorded = true; }
        Q3_HIP_CHECK(hipEventRecord(W.fork, stream));
        for (int i = 1; i < W.nlane; ++i) Q3_HIP_CHECK(hipStreamWaitEvent(W.lane_stream[i], W.fork, 0));
        W.window_open = true;
    }
    if (!W.lane_done[lane]) Q3_HIP_CHECK(hipEventCreateWithFlags(&W.lane_done[lane], hipEventDisableTiming));
    hipStream_t ls = W.lane_stream[lane];
    float* pcm_d = nullptr;
    const int64_t n = codec_run(codes_dev, nf, &pcm_d, lane, h_in, h_stage);
    const int64_t m = std::min(n, cap);
    if ((size_t)m > W.pinned_floats[lane]) {
        if (W.pinned[lane]) (void)hipHostFree(W.pinned[lane]);
        Q3_HIP_CHECK(hipHostMalloc((void**)&W.pinned[lane], (size_t)m * sizeof(float)));
        W.pinned_floats[lane] = (size_t)m;
    }
    if (m > 0) Q3_HIP_CHECK(hipMemcpyAsync(W.pinned[lane], pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ls));
    Q3_HIP_CHECK(hipEventRecord(W.lane_done[lane], ls));
    CodecW::Pending& p = W.pend[lane];
    p.items.assign(1, CodecW::Item{ user_pcm, n, cap, 0, len_out });
    p.frames = nf; p.busy = true;
}

// ------------------------------------------------------------------------------------------------
// Streaming decode with CARRIED state (the reference decodes the whole utterance in one run_vocoder call,
// /root/reference/src/tts_onnx.cpp:759-776, :430).  The decoder is causal.  Its only long memory is the pre-transformer (8 layers of
// 72-frame sliding-window attention: 568 frames of receptive field): a stream keeps every layer's K / V rows and the transformer's
// output rows, so a push of n new frames runs the transformer on n rows (positions n_done .. n_done + n, attention over the cached
// window) instead of over the history.  Everything behind it — ConvNeXt upsampling, the conv decoder — looks back a few frames only
// (stage_b_context: depthwise k7 convs, the 2r-tap transposed convs, three 7-tap dilated convs per block): it is decoded over the
// window [n_done - context, n_done + n) of the kept transformer rows and the new frames' samples are cut out, exactly as the windowed
// range decode does for the whole network.  Work per push: O(n + context), exact (same kernels as the one-shot decode on the same rows;
// only the GEMM tile shapes a short launch picks differ).
// ------------------------------------------------------------------------------------------------
int Engine::codec_stage_b_context() const {
    // frames the stages behind the pre-transformer look back, from the config: rows of lookback at each layer's rate / rows per frame
    double frames = 0.0, rate = 1.0;
    for (int s = 0; s < c.cd_n_up; ++s) { rate *= c.cd_up_ratios[s]; frames += 6.0 / rate; }            // ConvNeXt: depthwise causal k7 at the stage's output rate
    frames += 6.0 / rate;                                                                                    // conv_in k7
    for (int i = 0; i < c.cd_n_blocks; ++i) {
        frames += 1.0 / rate;                                                                                // transposed conv k = 2 r: one input row back
        rate *= c.cd_up_rates[i];
        frames += 6.0 * (1 + 3 + 9) / rate;                                                                  // three 7-tap convs, dilations 1, 3, 9
    }
    frames += 6.0 / rate;                                                                                    // conv_out k7
    return (int)frames + 3;                                                                                  // + margin for the transposed convs' trimming
}

int Engine::codec_stream_begin(int max_frames) {
    if (!codec) throw Error("codec decoder not finalized");
    if (max_frames < 1 || max_frames > (1 << 20)) throw Error("codec_stream_begin: max_frames out of range");
    CodecW& W = *codec;
    int sid = -1;
    for (size_t i = 0; i < W.streams.size(); ++i) if (!W.streams[i].used) { sid = (int)i; break; }
    if (sid < 0) { W.streams.emplace_back(); sid = (int)W.streams.size() - 1; }
    CodecW::Stream& S = W.streams[(size_t)sid];
    // buffers sized for pushes of up to 64 frames (16.8 MB of K / V + 0.4 MB of rows at 0.6B dims, whatever max_frames is); a larger
    // push grows them (codec_stream_fit).  Kept across streams: begin / end of a pooled stream allocate nothing.
    S.kv_rows = 0; S.h_rows = 0;                       // nothing of the buffers' previous stream is kept
    codec_stream_fit(sid, std::min(max_frames, 64));
    S.cap = max_frames;
    S.used = true; S.n_done = 0;
    int P = 1;
    while (P < max_frames) P <<= 1;
    codec_rope_tables(P);   // positions are absolute in the RoPE tables (grow-only, shared)
    return sid;
}

// make room for a push of n frames: K / V rows [kv_rows, kv_rows + n) and hpost rows [h_rows, h_rows + n) exist afterwards, the rows a
// later query / the stages behind the transformer still need are kept (moved to the front of the buffer, or into a larger one)
void Engine::codec_stream_fit(int sid, int n) {
    CodecW& W = *codec;
    CodecW::Stream& S = W.streams[(size_t)sid];
    const int CH = c.cd_hidden, NH = c.cd_heads, HD = c.cd_head_dim;
    const int keep = std::max(c.cd_window - 1, 0), ctx = codec_stage_b_context();
    const size_t blocks = (size_t)c.cd_layers * 2 * NH;   // (layer, k | v, head) row blocks of P rows each
    if (S.kv == nullptr || S.kv_rows + n > S.P) {
        const int keepN = std::min(keep, S.kv_rows);
        if (S.kv == nullptr || 2 * keep + n > S.P) {       // grow (or first allocation): P rows so that an in-place move never overlaps
            int P = 64, pshift = 6;
            while (P < 2 * keep + n) { P <<= 1; ++pshift; }
            float* nk = nullptr;
            Q3_HIP_CHECK(hipMalloc((void**)&nk, blocks * P * HD * sizeof(float)));
            if (S.kv && keepN > 0)
                Q3_HIP_CHECK(hipMemcpy2DAsync(nk, (size_t)P * HD * sizeof(float), S.kv + (size_t)(S.kv_rows - keepN) * HD, (size_t)S.P * HD * sizeof(float),
                                              (size_t)keepN * HD * sizeof(float), blocks, hipMemcpyDeviceToDevice, stream));
            if (S.kv) { sync(); (void)hipFree(S.kv); }
            S.kv = nk; S.P = P; S.pshift = pshift;
        } else if (keepN > 0) {                            // kv_rows > P - n >= 2 keep: source rows [kv_rows - keepN, kv_rows) lie behind the destination rows [0, keepN)
            Q3_HIP_CHECK(hipMemcpy2DAsync(S.kv, (size_t)S.P * HD * sizeof(float), S.kv + (size_t)(S.kv_rows - keepN) * HD, (size_t)S.P * HD * sizeof(float),
                                          (size_t)keepN * HD * sizeof(float), blocks, hipMemcpyDeviceToDevice, stream));
        }
        S.kv_rows = keepN;
    }
    if (S.hpost == nullptr || S.h_rows + n > S.h_cap) {
        const int ctxN = std::min(ctx, S.h_rows);
        if (S.hpost == nullptr || 2 * ctx + n > S.h_cap) {
            const int cap = 2 * ctx + std::max(n, 64);
            float* nh = nullptr;
            Q3_HIP_CHECK(hipMalloc((void**)&nh, (size_t)cap * CH * sizeof(float)));
            if (S.hpost && ctxN > 0)
                Q3_HIP_CHECK(hipMemcpyAsync(nh, S.hpost + (size_t)(S.h_rows - ctxN) * CH, (size_t)ctxN * CH * sizeof(float), hipMemcpyDeviceToDevice, stream));
            if (S.hpost) { sync(); (void)hipFree(S.hpost); }
            S.hpost = nh; S.h_cap = cap;
        } else if (ctxN > 0) {                             // h_rows > h_cap - n >= 2 ctx: no overlap
            Q3_HIP_CHECK(hipMemcpyAsync(S.hpost, S.hpost + (size_t)(S.h_rows - ctxN) * CH, (size_t)ctxN * CH * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
        S.h_rows = ctxN;
    }
}

void Engine::codec_stream_end(int sid) {
    if (!codec || sid < 0 || sid >= (int)codec->streams.size() || !codec->streams[(size_t)sid].used) throw Error("codec_stream_end: no such stream");
    codec->streams[(size_t)sid].used = false;        // buffers are kept for the next stream of this size
    codec->streams[(size_t)sid].n_done = 0;
}

int Engine::codec_stream_frames(int sid) const {
    if (!codec || sid < 0 || sid >= (int)codec->streams.size() || !codec->streams[(size_t)sid].used) throw Error("codec_stream: no such stream");
    return codec->streams[(size_t)sid].n_done;
}

// n new frames (codes_dev: int32 [n][n_groups] on the device) -> the samples they own, *pcm_dev pointing into the engine's decode arena
// (valid until the next decode on the engine's stream); returns the sample count
int64_t Engine::codec_stream_push_dev(int sid, const int32_t* codes_dev, int n, float** pcm_dev) {
    if (!codec || sid < 0 || sid >= (int)codec->streams.size() || !codec->streams[(size_t)sid].used) throw Error("codec_stream_push: no such stream");
    CodecW& W = *codec;
    CodecW::Stream& S = W.streams[(size_t)sid];
    if (n < 1) throw Error("codec_stream_push: no frames");
    if (S.n_done + n > S.cap) throw Error("codec_stream_push: more frames than the stream was opened for");
    const int CH = c.cd_hidden, NH = c.cd_heads, HD = c.cd_head_dim, FF = c.cd_ffn;
    const int a0 = S.n_done, b0 = a0 + n, T = n;
    codec_stream_fit(sid, n);
    const int k0 = S.kv_rows, h0 = S.h_rows;   // where this push's rows go in the sliding buffers (position a0 = K / V row k0 = hpost row h0)
    auto bytes_of = [](size_t nfloat) { return (nfloat * sizeof(float) + 255) & ~(size_t)255; };
    const size_t kslab_floats = (size_t)32 * 128 * 4096;
    const size_t need = bytes_of((size_t)T * CH) * 3 + bytes_of((size_t)T * 3 * CH) + bytes_of((size_t)T * FF) * 2 + bytes_of(kslab_floats);
    if (W.stream_arena_bytes < need) {
        sync();
        if (W.stream_arena) (void)hipFree(W.stream_arena);
        W.stream_arena = nullptr;
        Q3_HIP_CHECK(hipMalloc((void**)&W.stream_arena, need));
        W.stream_arena_bytes = need;
    }
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = (float*)(W.stream_arena + off); off += bytes_of(nfloat); return p; };
    float* h = take((size_t)T * CH);
    float* hn = take((size_t)T * CH);
    float* att = take((size_t)T * CH);
    float* qkvb = take((size_t)T * 3 * CH);
    float* ub = take((size_t)T * FF);
    float* gb = take((size_t)T * FF);
    float* kslab = take(kslab_floats);
    auto conv = [&](ConvArgs a) {
        a.slab = kslab; a.slab_floats = kslab_floats;
        const auto it = W.planes.find(a.W);
        if (it != W.planes.end()) { a.Wh = it->second.hi; a.Wl = it->second.lo; a.Whc = it->second.cm; a.w_scale_inv = it->second.scale_inv; a.w_lo_zero = it->second.lo_zero; }
        launch_conv(a, stream);
    };
    auto gemm = [&](const float* in, int Cin, const float* Wm, int Cout, float* out) {
        ConvArgs a; a.in = in; a.T_in = T; a.C_in = Cin; a.out = out; a.T_out = T; a.C_out = Cout; a.W = Wm;
        return a;
    };
    // ---- the pre-transformer on the new rows: positions [a0, b0), keys / values of earlier frames from the stream's caches ----
    const int half = HD / 2;
    const size_t layer_kv = (size_t)NH * S.P * HD;
    launch_code_embed_mean(W.code_embed, codes_dev, T, c.n_groups, c.cd_codebook, CH, h, stream);
    for (int l = 0; l < c.cd_layers; ++l) {
        const CodecW::Layer& L = W.layers[l];
        float* kc = S.kv + (size_t)l * 2 * layer_kv;
        float* vc = kc + layer_kv;
        launch_rmsnorm_rows(h, L.in_norm, c.cd_rms_eps, T, CH, hn, stream);
        conv(gemm(hn, CH, L.qkv, 3 * CH, qkvb));
        // row t of this push is position a0 + t: the tables and the cache rows start there
        launch_rope_store(qkvb, 3 * CH, T, NH, NH, HD, W.rope_cos + (size_t)a0 * half, W.rope_sin + (size_t)a0 * half, kc + (size_t)k0 * HD, vc + (size_t)k0 * HD, S.P, stream);
        AttnArgs a;
        a.qkv = qkvb; a.ld_qkv = 3 * CH; a.out = att; a.ld_out = CH; a.kcache = kc; a.vcache = vc;
        a.page_table = W.page_table; a.pages_per_slot = 1; a.page_shift = S.pshift; a.layer = 0; a.n_layers = 1;
        a.pos_scalar = k0;   // cache rows are addressed relative to the sliding buffer; every row a window can reach is in it (RoPE is already applied, absolute)
        a.slot_offset = 0; a.nb = 1; a.n_new = T; a.nq = NH; a.nkv = NH; a.d = HD;
        a.scale = 1.0f / sqrtf((float)HD); a.window = c.cd_window; a.new_from_raw = 0;
        launch_attn(a, stream);
        { ConvArgs g = gemm(att, CH, L.o, CH, h); g.res_scale = L.attn_scale; g.res = h; conv(g); }
        launch_rmsnorm_rows(h, L.post_norm, c.cd_rms_eps, T, CH, hn, stream);
        conv(gemm(hn, CH, L.up, FF, ub));
        { ConvArgs g = gemm(hn, CH, L.gate, FF, gb); g.act = 2; g.mul = ub; conv(g); }
        { ConvArgs g = gemm(gb, FF, L.down, CH, h); g.res_scale = L.mlp_scale; g.res = h; conv(g); }
    }
    launch_rmsnorm_rows(h, W.norm, c.cd_rms_eps, T, CH, S.hpost + (size_t)h0 * CH, stream);
    // ---- everything behind the transformer over the window [sB, b0) of its kept output rows ----
    const int sB = std::max(0, a0 - codec_stage_b_context());   // h0 >= a0 - sB: codec_stream_fit keeps that many rows
    int64_t up = 1;
    for (int i = 0; i < c.cd_n_up; ++i) up *= c.cd_up_ratios[i];
    for (int i = 0; i < c.cd_n_blocks; ++i) up *= c.cd_up_rates[i];
    auto len_of = [&](int nfr) -> int64_t { return nfr <= 0 ? 0 : q3tts_codec_decode_len(&c, nfr); };
    const int64_t first = len_of(a0) - up * sB, n_own = len_of(b0) - len_of(a0);
    float* pcm_d = nullptr;
    if (h0 < a0 - sB) throw Error("codec_stream_push: the kept transformer rows do not cover the look-back window");
    const int64_t n_win = codec_run(nullptr, b0 - sB, &pcm_d, 0, S.hpost + (size_t)(h0 - (a0 - sB)) * CH, 1);
    if (first < 0 || first + n_own != n_win) throw Error("codec_stream_push: window arithmetic does not match the decoder length formula");
    S.n_done = b0; S.kv_rows = k0 + n; S.h_rows = h0 + n;
    if (pcm_dev) *pcm_dev = pcm_d + first;
    return n_own;
}

int64_t Engine::codec_stream_push_host(int sid, const int64_t* codes, int n, float* pcm, int64_t cap) {
    if (n < 1) throw Error("codec_stream_push: no frames");
    const int G = c.n_groups;
    std::vector<int32_t> tmp((size_t)n * G);
    for (size_t i = 0; i < tmp.size(); ++i) {
        if (codes[i] < 0 || codes[i] >= c.cd_codebook) throw Error("codec_decode: code out of range");
        tmp[i] = (int32_t)codes[i];
    }
    if (n > max_frames_cap) throw Error("codec_stream_push: more frames in one push than the engine's frame capacity");
    Q3_HIP_CHECK(hipMemcpyAsync(codes_scratch_d, tmp.data(), tmp.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n_own = codec_stream_push_dev(sid, codes_scratch_d, n, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    const int64_t m = std::min(n_own, cap);
    if (m > 0 && pcm) Q3_HIP_CHECK(hipMemcpyAsync(pcm, pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += n;
    return n_own;
}

// the implicit stream behind q3tts_slot_codec_decode_range_host: consecutive exact ranges of a slot ([0, b1), [b1, b2), ...) are pushes
int64_t Engine::slot_codec_stream_range(int slot, int a, int b, float* pcm, int64_t cap) {
    CodecW& W = *codec;
    if ((int)W.slot_stream.size() < B) W.slot_stream.assign((size_t)B, -1);
    int sid = W.slot_stream[(size_t)slot];
    if (sid < 0 || !W.streams[(size_t)sid].used || W.streams[(size_t)sid].n_done > a) {   // first range of the utterance, or a restart
        if (sid >= 0 && W.streams[(size_t)sid].used) codec_stream_end(sid);
        sid = codec_stream_begin(max_frames_cap);
        W.slot_stream[(size_t)slot] = sid;
    }
    const int32_t* codes = codes_d + (size_t)slot * max_frames_cap * c.n_groups;
    CodecW::Stream& S = W.streams[(size_t)sid];
    if (S.n_done < a) codec_stream_push_dev(sid, codes + (size_t)S.n_done * c.n_groups, a - S.n_done, nullptr);   // frames nobody asked the audio of
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n_own = codec_stream_push_dev(sid, codes + (size_t)a * c.n_groups, b - a, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    const int64_t m = std::min(n_own, cap);
    if (m > 0 && pcm) Q3_HIP_CHECK(hipMemcpyAsync(pcm, pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += b - a;
    return n_own;
}
void Engine::slot_codec_stream_reset(int slot) {
    if (!codec) return;
    CodecW& W = *codec;
    if (slot < 0 || slot >= (int)W.slot_stream.size()) return;
    const int sid = W.slot_stream[(size_t)slot];
    if (sid >= 0 && W.streams[(size_t)sid].used) codec_stream_end(sid);
    W.slot_stream[(size_t)slot] = -1;
}

// Test hook (Q3TTS_FLAG_TEST_HOOKS engines): every byte of the vocoder's reusable workspace — lane arenas, the batched front's arena,
// the streaming arena, the pinned PCM staging buffers, the job's code rows — becomes 0xFF (NaN as fp32 and as fp16), so a decode that
// reads anything it did not write in the same call shows up as NaN instead of as yesterday's plausible samples.
void Engine::codec_poison() {
    if (!(flags & Q3TTS_FLAG_TEST_HOOKS)) throw Error("codec_poison needs Q3TTS_FLAG_TEST_HOOKS");
    if (!codec) return;
    CodecW& W = *codec;
    for (int i = 0; i < W.nlane; ++i) Q3_HIP_CHECK(hipStreamSynchronize(W.lane_stream[i]));
    for (int i = 0; i < CodecW::NLANE; ++i) {
        if (W.arena[i]) Q3_HIP_CHECK(hipMemset(W.arena[i], 0xFF, W.arena_bytes[i]));
        if (W.pinned[i]) memset(W.pinned[i], 0xFF, W.pinned_floats[i] * sizeof(float));
    }
    if (W.batch_arena) Q3_HIP_CHECK(hipMemset(W.batch_arena, 0xFF, W.batch_arena_bytes));
    if (W.stream_arena) Q3_HIP_CHECK(hipMemset(W.stream_arena, 0xFF, W.stream_arena_bytes));
    if (W.job_codes) Q3_HIP_CHECK(hipMemset(W.job_codes, 0xFF, W.job_codes_n * sizeof(int32_t)));   // -1: clamped to code 0 by the gather, never out of the table
    Q3_HIP_CHECK(hipDeviceSynchronize());
}

// Test hook: input rows [nb][T][C] and output [nb][T] of the last batched group's final conv, as they lie in the lane's arena now
void Engine::codec_debug_group(float* sx_out, float* pcm_out, int64_t cap_floats, int* T, int* C, int* nb) {
    if (!(flags & Q3TTS_FLAG_TEST_HOOKS)) throw Error("codec_debug_group needs Q3TTS_FLAG_TEST_HOOKS");
    CodecW& W = *codec;
    *T = W.dbg_T; *C = W.dbg_C; *nb = W.dbg_nb;
    if (!W.dbg_sx) return;
    Q3_HIP_CHECK(hipDeviceSynchronize());
    const int64_t n = (int64_t)W.dbg_nb * W.dbg_T * W.dbg_C;
    if (sx_out && n <= cap_floats) Q3_HIP_CHECK(hipMemcpy(sx_out, W.dbg_sx, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (pcm_out) Q3_HIP_CHECK(hipMemcpy(pcm_out, W.dbg_pcm, (size_t)W.dbg_nb * W.dbg_T * sizeof(float), hipMemcpyDeviceToHost));
}
int64_t Engine::codec_debug_partials(float* out, int64_t cap_floats) {
    const float* p = nullptr; size_t n = 0;
    cout1_debug_buffer(&p, &n);
    if (out && p && (int64_t)n <= cap_floats) { Q3_HIP_CHECK(hipDeviceSynchronize()); Q3_HIP_CHECK(hipMemcpy(out, p, n * sizeof(float), hipMemcpyDeviceToHost)); }
    return (int64_t)n;
}

bool Engine::codec_batchable() const {
    if (!codec || codec->planes.empty()) return false;                     // exact-fp32 codec: no split-precision planes
    const int D = c.cd_decoder_dim, OD = D >> c.cd_n_blocks;
    if (c.cd_hidden % 32 || OD % 32 || OD < 32) return false;             // every channel count of the decoder is then a multiple of 32
    if ((size_t)(128 + 6) * (OD + 1) * sizeof(float) > 60 * 1024) return false;   // last conv (C_out = 1) in LDS
    for (int i = 0; i < c.cd_n_blocks; ++i) if (c.cd_up_rates[i] < 1) return false;
    return true;
}

// The conv decoder of g consecutive utterances of codec_pre_batch's output in ONE set of launches (rows [utterance][Tp] in every layer; the
// kernels' row tiles never straddle utterances, and causality keeps the padding out of the real samples): h_group = the group's upsampled
// rows (h_ustride floats apart), Fg = frames of the group's longest utterance.  Each utterance's sample count comes from its own frame count.
void Engine::codec_async_submit_group(const float* h_group, size_t h_ustride, int Fg, int g, const int* nf, float* const* user_pcm, int64_t cap, int64_t* const* len_out) {
    CodecW& W = *codec;
    // a group's launches fill the chip on their own: two lanes are enough to overlap one group's tail with the next one's start (and each
    // lane owns a group-sized arena)
    const int lane = W.nlane > 1 ? 1 + (W.rr++ % std::min(W.nlane - 1, 2)) : 0;
    codec_async_drain_lane(lane);
    if (W.fail_at_submit > 0 && ++W.submits == W.fail_at_submit) throw Error("vocoder submit failed (injected by Q3TTS_TEST_FAIL_VOCODER_SUBMIT)");
    if (!W.window_open) {
        if (!W.win0_recorded) { Q3_HIP_CHECK(hipEventRecord(W.win0, stream)); W.win0_recorded = true; }
        Q3_HIP_CHECK(hipEventRecord(W.fork, stream));
        for (int i = 1; i < W.nlane; ++i) Q3_HIP_CHECK(hipStreamWaitEvent(W.lane_stream[i], W.fork, 0));
        W.window_open = true;
    }
    if (!W.lane_done[lane]) Q3_HIP_CHECK(hipEventCreateWithFlags(&W.lane_done[lane], hipEventDisableTiming));
    hipStream_t ls = W.lane_stream[lane];
    if (lane != 0) {   // this group's rows come from the batched pass just queued on the engine stream (a later block than the window's first)
        Q3_HIP_CHECK(hipEventRecord(W.fork, stream));
        Q3_HIP_CHECK(hipStreamWaitEvent(ls, W.fork, 0));
    }
    float* pcm_d = nullptr;
    const int64_t Tp = codec_run(nullptr, Fg, &pcm_d, lane, h_group, 2, g, h_ustride);   // padded samples per utterance (Fg = the group's longest)
    CodecW::Pending& p = W.pend[lane];
    p.items.clear();
    int64_t total = 0;
    int frames = 0;
    for (int u = 0; u < g; ++u) {
        const int64_t n = nf[u] > 0 ? q3tts_codec_decode_len(&c, nf[u]) : 0;
        p.items.push_back(CodecW::Item{ user_pcm ? user_pcm[u] : nullptr, n, cap, total, len_out ? len_out[u] : nullptr });
        total += std::min(n, cap);
        frames += std::max(nf[u], 0);
    }
    if ((size_t)total > W.pinned_floats[lane]) {
        if (W.pinned[lane]) (void)hipHostFree(W.pinned[lane]);
        Q3_HIP_CHECK(hipHostMalloc((void**)&W.pinned[lane], (size_t)total * sizeof(float)));
        W.pinned_floats[lane] = (size_t)total;
    }
    for (int u = 0; u < g; ++u) {
        const CodecW::Item& it = p.items[(size_t)u];
        const int64_t m = std::min(it.n, it.cap);
        if (m > 0) Q3_HIP_CHECK(hipMemcpyAsync(W.pinned[lane] + it.off, pcm_d + (size_t)u * Tp, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ls));
    }
    Q3_HIP_CHECK(hipEventRecord(W.lane_done[lane], ls));
    p.frames = frames; p.busy = true;
}

// the engine stream waits for every lane's queued work (no host synchronisation): what it launches next may overwrite what the lanes read
void Engine::codec_lanes_join() {
    if (!codec) return;
    CodecW& W = *codec;
    for (int i = 1; i < W.nlane; ++i)
        if (W.pend[i].busy && W.lane_done[i]) Q3_HIP_CHECK(hipStreamWaitEvent(stream, W.lane_done[i], 0));
}

void Engine::codec_async_drain() {
    if (!codec) return;
    CodecW& W = *codec;
    if (!W.window_open) return;
    for (int i = 0; i < W.nlane; ++i)
        if (W.pend[i].busy && i != 0) Q3_HIP_CHECK(hipStreamWaitEvent(stream, W.lane_done[i], 0));
    Q3_HIP_CHECK(hipEventRecord(W.win1, stream));
    for (int i = 0; i < W.nlane; ++i) codec_async_drain_lane(i);
    sync();
    float ms = 0.f;
    Q3_HIP_CHECK(hipEventElapsedTime(&ms, W.win0, W.win1));
    total_codec_ms += ms; last_codec_ms = ms;
    W.window_open = false; W.win0_recorded = false;
}

} // namespace q3
