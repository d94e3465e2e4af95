// q3_engine.cpp — device-resident engine: weight registry, workspaces, paged KV cache, the
// per-frame launch sequence (captured into a hipGraph), and the mirrored host logic of the
// reference's generation path (src/tts_onnx.cpp:442-539, 782-872).
#include "q3_engine.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace q3 {

// ---- host-logic constants, reference src/tts_onnx.h:40-62 ----
static const int64_t TTS_BOS = 151672, TTS_EOS = 151673, TTS_PAD = 151671;
static const int64_t CODEC_BOS = 2149, CODEC_PAD = 2148, CODEC_THINK = 2154, CODEC_NOTHINK = 2155;
static const int64_t CODEC_THINK_BOS = 2156, CODEC_THINK_EOS = 2157, LANG_ENGLISH = 2050;

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

void* Engine::dmalloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    Q3_HIP_CHECK(hipMalloc(&p, bytes));
    allocs.push_back(p);
    return p;
}
void Engine::sync() { Q3_HIP_CHECK(hipStreamSynchronize(stream)); }

Tensor& Engine::T(const std::string& n) {
    auto it = tindex.find(n);
    if (it == tindex.end()) throw Error("unknown tensor '" + n + "'");
    return tensors[it->second];
}

// ------------------------------------------------------------------------------------------------
// tensor registry (names/shapes shared with the oracle and DESIGN.md section 3) — host-only, no HIP call: the importer and
// q3tts_config_tensor_info walk it without a GPU; the engine allocates from it.
// ------------------------------------------------------------------------------------------------
std::vector<TensorSpec> tensor_specs(const q3tts_config& c) {
    std::vector<TensorSpec> out;
    // counts that index fixed arrays or drive loops: an out-of-range config has no registry (callers report it)
    if (c.n_layers < 0 || c.n_layers > 1024 || c.cp_layers < 0 || c.cp_layers > 1024 || c.cd_layers < 0 || c.cd_layers > 1024 ||
        c.n_groups < 2 || c.n_groups > 32 || c.cd_n_up < 0 || c.cd_n_up > 4 || c.cd_n_blocks < 0 || c.cd_n_blocks > 8 || c.spk_scale > 64) return out;
    auto add = [&](const std::string& name, std::vector<int64_t> shape, int kind, bool bf, int fuse = 0, float sstd = 0.02f) {
        TensorSpec t;
        t.name = name; t.ndim = (int)shape.size(); t.kind = kind; t.bf16 = bf; t.fuse = fuse; t.synth_std = sstd;
        for (int i = 0; i < t.ndim; ++i) t.shape[i] = shape[i];
        out.push_back(t);
    };
    auto add_layers = [&](const std::string& prefix, int n, int H, int nq, int nkv, int d, int ffn, bool qk, bool ls, bool bf) {
        for (int i = 0; i < n; ++i) {
            const std::string p = prefix + ".layers." + std::to_string(i) + ".";
            add(p + "input_norm", {H}, TK_NORM, false);
            add(p + "q_proj", {(int64_t)nq * d, H}, TK_W, bf, 1);   // q | k | v rows contiguous in HBM: one GEMV
            add(p + "k_proj", {(int64_t)nkv * d, H}, TK_W, bf, 2);
            add(p + "v_proj", {(int64_t)nkv * d, H}, TK_W, bf, 3);
            add(p + "o_proj", {H, (int64_t)nq * d}, TK_W, bf);
            if (qk) { add(p + "q_norm", {d}, TK_NORM, false); add(p + "k_norm", {d}, TK_NORM, false); }
            add(p + "post_norm", {H}, TK_NORM, false);
            add(p + "gate_proj", {ffn, H}, TK_W, bf);
            add(p + "up_proj", {ffn, H}, TK_W, bf);
            add(p + "down_proj", {H, ffn}, TK_W, bf);
            if (ls) { add(p + "attn_scale", {H}, TK_SCALE, false); add(p + "mlp_scale", {H}, TK_SCALE, false); }
        }
    };
    const int H = c.hidden;
    add_layers("talker", c.n_layers, H, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn, true, false, true);
    add("talker.norm", {H}, TK_NORM, false);
    add("talker.codec_head", {c.vocab, H}, TK_W, true);
    add("talker.codec_embed", {c.vocab, H}, TK_W, true);
    add("text.embed", {c.text_vocab, c.text_hidden}, TK_W, true);
    add("text.fc1.w", {c.text_hidden, c.text_hidden}, TK_W, true);
    add("text.fc1.b", {c.text_hidden}, TK_BIAS, false);
    add("text.fc2.w", {H, c.text_hidden}, TK_W, true);
    add("text.fc2.b", {H}, TK_BIAS, false);
    // predictor width: the talker's (0.6B) or narrower behind cp.proj (1.7B: 2048 -> 1024); its embeddings stay talker-wide
    const int Hc = c.cp_hidden > 0 ? c.cp_hidden : c.hidden;
    add_layers("cp", c.cp_layers, Hc, c.cp_heads, c.cp_kv_heads, c.cp_head_dim, c.cp_ffn, true, false, true);
    add("cp.norm", {Hc}, TK_NORM, false);
    if (Hc != H) { add("cp.proj.w", {Hc, H}, TK_W, true); add("cp.proj.b", {Hc}, TK_BIAS, false); }
    for (int j = 0; j < c.n_groups - 1; ++j) add("cp.head." + std::to_string(j), {c.sub_vocab, Hc}, TK_W, true);
    for (int j = 0; j < c.n_groups - 1; ++j) add("cp.embed." + std::to_string(j), {c.sub_vocab, H}, TK_W, true);
    const int CH = c.cd_hidden;
    add_layers("cd", c.cd_layers, CH, c.cd_heads, c.cd_heads, c.cd_head_dim, c.cd_ffn, false, true, false);
    add("cd.norm", {CH}, TK_NORM, false);
    add("cd.code_embed", {(int64_t)c.n_groups * c.cd_codebook, CH}, TK_W, false);
    for (int s = 0; s < c.cd_n_up; ++s) {
        const int f = c.cd_up_ratios[s];
        const std::string p = "cd.up." + std::to_string(s) + ".";
        add(p + "tconv.w", {CH, CH, f}, TK_W, false);
        add(p + "tconv.b", {CH}, TK_BIAS, false);
        add(p + "cnx.dw.w", {CH, 1, 7}, TK_W, false);
        add(p + "cnx.dw.b", {CH}, TK_BIAS, false);
        add(p + "cnx.ln.w", {CH}, TK_NORM, false);
        add(p + "cnx.ln.b", {CH}, TK_BIAS, false);
        add(p + "cnx.pw1.w", {4 * CH, CH}, TK_W, false);
        add(p + "cnx.pw1.b", {4 * CH}, TK_BIAS, false);
        add(p + "cnx.pw2.w", {CH, 4 * CH}, TK_W, false);
        add(p + "cnx.pw2.b", {CH}, TK_BIAS, false);
        add(p + "cnx.gamma", {CH}, TK_SCALE, false);
    }
    const int D = c.cd_decoder_dim;
    add("cd.dec.conv_in.w", {D, CH, 7}, TK_W, false);
    add("cd.dec.conv_in.b", {D}, TK_BIAS, false);
    for (int i = 0; i < c.cd_n_blocks; ++i) {
        const int cin = D >> i, cout = D >> (i + 1), r = c.cd_up_rates[i];
        const std::string p = "cd.dec.blocks." + std::to_string(i) + ".";
        add(p + "snake.alpha", {cin}, TK_SNAKE, false);
        add(p + "snake.beta", {cin}, TK_SNAKE, false);
        add(p + "tconv.w", {cin, cout, 2 * r}, TK_W, false);
        add(p + "tconv.b", {cout}, TK_BIAS, false);
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + "res." + std::to_string(u) + ".";
            add(q + "act1.alpha", {cout}, TK_SNAKE, false);
            add(q + "act1.beta", {cout}, TK_SNAKE, false);
            add(q + "conv1.w", {cout, cout, 7}, TK_W, false);
            add(q + "conv1.b", {cout}, TK_BIAS, false);
            add(q + "act2.alpha", {cout}, TK_SNAKE, false);
            add(q + "act2.beta", {cout}, TK_SNAKE, false);
            add(q + "conv2.w", {cout, cout, 1}, TK_W, false);
            add(q + "conv2.b", {cout}, TK_BIAS, false);
        }
    }
    const int OD = D >> c.cd_n_blocks;
    add("cd.dec.snake_out.alpha", {OD}, TK_SNAKE, false);
    add("cd.dec.snake_out.beta", {OD}, TK_SNAKE, false);
    add("cd.dec.conv_out.w", {1, OD, 7}, TK_W, false, 0, 0.002f);
    add("cd.dec.conv_out.b", {1}, TK_BIAS, false);
    if (c.spk_enc_dim > 0) { // ECAPA-TDNN speaker encoder (clone path), torch Conv1d layout [out][in][k], fp32
        if (c.spk_scale < 1 || c.spk_channels < 1) return out;   // malformed: the engine constructor reports it
        const int SC = c.spk_channels, sub = SC / c.spk_scale;
        auto conv = [&](const std::string& n, int cout, int cin, int k) {
            add(n + ".w", {cout, cin, k}, TK_W, false, 0, 1.0f / sqrtf((float)(cin * k)));
            add(n + ".b", {cout}, TK_BIAS, false);
        };
        conv("spk.tdnn0", SC, c.spk_mel, 5);
        for (int i = 0; i < 3; ++i) {
            const std::string p = "spk.blocks." + std::to_string(i) + ".";
            conv(p + "tdnn1", SC, SC, 1);
            for (int j = 0; j < c.spk_scale - 1; ++j) conv(p + "res2net." + std::to_string(j), sub, sub, 3);
            conv(p + "tdnn2", SC, SC, 1);
            conv(p + "se1", c.spk_se, SC, 1);
            conv(p + "se2", SC, c.spk_se, 1);
        }
        conv("spk.mfa", 3 * SC, 3 * SC, 1);
        conv("spk.asp.tdnn", c.spk_att, 9 * SC, 1);
        conv("spk.asp.conv", 3 * SC, c.spk_att, 1);
        conv("spk.fc", c.spk_enc_dim, 6 * SC, 1);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// A/B knobs: environment variables honoured only while a Q3TTS_FLAG_TEST_HOOKS engine is alive (q3_common.h)
// ------------------------------------------------------------------------------------------------
static std::atomic<int> g_hook_engines{0};
const char* knob(const char* name) { return g_hook_engines.load(std::memory_order_relaxed) > 0 ? getenv(name) : nullptr; }
void KnobScope::enable() { if (!on) { on = true; g_hook_engines.fetch_add(1, std::memory_order_relaxed); } }
KnobScope::~KnobScope() { if (on) g_hook_engines.fetch_sub(1, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------------
Engine::Engine(const q3tts_config& cfg, int device_, int max_batch, int max_ctx_, uint32_t flags_, int64_t kv_pool_tokens)
    : c(cfg), device(device_), B(max_batch), max_ctx(max_ctx_), flags(flags_) {
    if (flags & Q3TTS_FLAG_TEST_HOOKS) knob_scope.enable();   // before the first knob() below
    if (B < 1 || B > 1024) throw Error("max_batch out of range");
    if (kv_pool_tokens < 0) throw Error("kv_pool_tokens must be >= 0");
    if (c.cp_hidden < 0) throw Error("cp_hidden must be >= 0");
    if (c.cp_hidden == c.hidden) c.cp_hidden = 0;
    if (c.hidden % 8 || c.ffn % 8 || c.text_hidden % 8 || c.cp_ffn % 8 || cp_width() % 8) throw Error("hidden/ffn sizes must be multiples of 8");
    if (c.vocab > 4096 || c.sub_vocab > 4096) throw Error("codec vocabularies larger than 4096 are not supported");
    if (c.n_groups < 2 || c.n_groups > 32) throw Error("n_groups out of range");
    Q3_HIP_CHECK(hipSetDevice(device));
    // Q3TTS_NULL_STREAM=1 (profiling aid): rocprofv3 --pmc crashes on user-created streams on this ROCm; run on the
    // default stream instead (forces eager launches: the default stream cannot be captured)
    if (const char* mr = knob("Q3TTS_MFMA_MIN_ROWS")) mfma_min_rows = std::max(3, atoi(mr));   // A/B knob for the GEMV <-> GEMM crossover
    if (const char* sv = knob("Q3TTS_SEAM")) seam_on = atoi(sv) != 0;
    if (const char* sv = knob("Q3TTS_SEAM_SPIN")) seam_spin = std::max(1, atoi(sv));
    attn_keep_splits = knob("Q3TTS_ATTN_KEEP_SPLITS") != nullptr;
    if (const char* sv = knob("Q3TTS_ATTN_STREAM")) attn_stream = atoi(sv) != 0;
    if (const char* sv = knob("Q3TTS_ATTN_STREAM_ONE")) attn_stream_one = atoi(sv) != 0;
    null_stream = getenv("Q3TTS_NULL_STREAM") && getenv("Q3TTS_NULL_STREAM")[0] == '1';
    if (null_stream) { stream = nullptr; flags |= Q3TTS_FLAG_NO_GRAPH; }
    else if (const char* cm = knob("Q3TTS_STREAM_CU_MASK")) {   // experiment aid (tools/overlap_probe.py): this engine's stream on a subset of the CUs;
        const uint32_t pat = (uint32_t)strtoul(cm, nullptr, 16);    // the 32-bit pattern is repeated over the 256-CU mask
        if (pat == 0) throw Error("Q3TTS_STREAM_CU_MASK must be a non-zero hexadecimal CU pattern (a stream with no CU never runs)");
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = pat;
        Q3_HIP_CHECK(hipExtStreamCreateWithCUMask(&stream, 8, mask));
    }
    else {   // the decode chain is latency-bound: its launches go ahead of the vocoder lanes' (created at the lowest priority)
        int lo = 0, hi = 0;
        Q3_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        Q3_HIP_CHECK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));
    }
    Q3_HIP_CHECK(hipEventCreate(&ev0));
    Q3_HIP_CHECK(hipEventCreate(&ev1));

    if (c.spk_enc_dim > 0 && (c.spk_scale < 2 || c.spk_scale > 16 || c.spk_channels % c.spk_scale || c.spk_mel < 1 || c.spk_se < 1 || c.spk_att < 1))
        throw Error("speaker encoder dims out of range");
    // the encoder's output is spliced into the prompt as ONE talker-width row (build_prompts copies `hidden` floats from it)
    if (c.spk_enc_dim > 0 && c.spk_enc_dim != c.hidden) throw Error("speaker encoder output width must equal the talker width (spk_enc_dim == hidden)");
    {   // allocate the registry (q | k | v of a layer share one block so that their rows are contiguous)
        char* fused = nullptr; size_t fused_off = 0;
        const std::vector<TensorSpec> specs = tensor_specs(c);
        if (specs.empty()) throw Error("model config out of range (layer / group / block counts)");
        for (size_t i = 0; i < specs.size(); ++i) {
            const TensorSpec& sp = specs[i];
            Tensor t;
            t.name = sp.name; t.ndim = sp.ndim; t.kind = sp.kind; t.bf16 = sp.bf16; t.synth_std = sp.synth_std; t.numel = 1;
            for (int k = 0; k < sp.ndim; ++k) { t.shape[k] = sp.shape[k]; t.numel *= sp.shape[k]; }
            const size_t bytes = (size_t)t.numel * (sp.bf16 ? 2 : 4);
            if (sp.fuse == 1) {
                size_t total = bytes;
                for (size_t k = i + 1; k < specs.size() && specs[k].fuse > 1; ++k) {
                    size_t n = 1;
                    for (int d = 0; d < specs[k].ndim; ++d) n *= (size_t)specs[k].shape[d];
                    total += n * (specs[k].bf16 ? 2 : 4);
                }
                fused = (char*)dmalloc(total); fused_off = 0;
            }
            if (sp.fuse > 0) { t.dev = fused + fused_off; fused_off += bytes; }
            else t.dev = dmalloc(bytes);
            tindex[t.name] = (int)tensors.size();
            tensors.push_back(t);
        }
    }
    const int H = c.hidden, Hc = cp_width();

    // ---- workspaces ----
    rows_max = std::max(2 * B, 16);
    max_trailing = 1024;
    max_frames_cap = max_ctx;
    const int QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, QKVp = (c.cp_heads + 2 * c.cp_kv_heads) * c.cp_head_dim;
    const int AO = std::max(c.n_heads * c.head_dim, c.cp_heads * c.cp_head_dim);
    auto fm = [&](size_t n) { float* p = (float*)dmalloc(n * sizeof(float)); Q3_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(float), stream)); return p; };
    x_talk = fm((size_t)B * H);
    qkv = fm((size_t)rows_max * std::max(QKV, QKVp));
    attn = fm((size_t)rows_max * AO);
    act = fm((size_t)rows_max * std::max(c.ffn, c.cp_ffn));
    logits_t = fm((size_t)B * c.vocab);
    logits_cp = fm((size_t)B * c.sub_vocab);
    cp_logit_slab_d = fm((size_t)4 * B * c.sub_vocab);
    x_cp = fm((size_t)B * 2 * H);
    x_cp1 = fm((size_t)B * H);
    if (cp_projected()) x_cpp = fm((size_t)std::max(rows_max, 16) * Hc);
    sum = fm((size_t)B * H);
    xp = fm((size_t)std::max(16, rows_max) * H);   // one prefill group: up to rows_max prompt rows
    hn = fm((size_t)std::max(16, rows_max) * H);
    logits_p = fm((size_t)16 * std::max(c.vocab, c.sub_vocab));
    trailing_d = fm((size_t)B * max_trailing * H);
    tts_pad_d = fm(H);
    text_tmp = fm((size_t)16 * c.text_hidden);
    text_tmp2 = fm((size_t)16 * c.text_hidden);
    ldp = (std::max(std::max(H, AO), std::max(c.ffn, c.cp_ffn)) + 7) / 8 * 8;
    pl0h = (bf16_t*)dmalloc((size_t)rows_max * ldp * 2); pl0l = (bf16_t*)dmalloc((size_t)rows_max * ldp * 2);
    pl1h = (bf16_t*)dmalloc((size_t)rows_max * ldp * 2); pl1l = (bf16_t*)dmalloc((size_t)rows_max * ldp * 2);
    slab_d = fm((size_t)16 * rows_max * H);
    qkv_slab_d = fm((size_t)4 * rows_max * std::max(QKV, QKVp));
    {   // seam flag lines: <= 3 seam launches per layer pass, <= (ffn / 64) x 2 row blocks units of 16 words (one 64-byte line) each
        const size_t per_launch = (size_t)(std::max(std::max(c.ffn, c.cp_ffn), H) / 64 + 1) * 2 * 16;
        const size_t launches = (size_t)3 * ((size_t)c.n_layers + (size_t)(c.n_groups - 1) * c.cp_layers) + 8;
        seam_cnt_words = per_launch * launches;
        seam_cnt_d = (unsigned*)dmalloc(seam_cnt_words * sizeof(unsigned));
        Q3_HIP_CHECK(hipMemsetAsync(seam_cnt_d, 0, seam_cnt_words * sizeof(unsigned), stream));
        seam_gen_d = (unsigned*)dmalloc(64);
        Q3_HIP_CHECK(hipMemsetAsync(seam_gen_d, 0, 64, stream));
        ssq_a_d = fm((size_t)rows_max * 64);
        ssq_b_d = fm((size_t)rows_max * 64);
    }
    gu_slab_d = fm((size_t)2 * 8 * rows_max * std::max(c.ffn, c.cp_ffn));   // gate | up halves of up to 8 K slices each
    ids_d = (int64_t*)dmalloc(64 * sizeof(int64_t));
    tok_d = (int64_t*)dmalloc(sizeof(int64_t));
    codes_d = (int32_t*)dmalloc((size_t)B * max_frames_cap * c.n_groups * sizeof(int32_t));
    codes_scratch_d = (int32_t*)dmalloc((size_t)max_frames_cap * c.n_groups * sizeof(int32_t));
    talker_pos_d = (int32_t*)dmalloc((size_t)B * sizeof(int32_t));
    slot_map_d = (int*)dmalloc(128 * sizeof(int));
    logits_g = fm((size_t)128 * c.vocab);
    Q3_HIP_CHECK(hipMemsetAsync(talker_pos_d, 0, (size_t)B * sizeof(int32_t), stream));
    st_d = (SlotState*)dmalloc((size_t)B * sizeof(SlotState));
    st_h.assign(B, SlotState{});
    Q3_HIP_CHECK(hipMemsetAsync(st_d, 0, (size_t)B * sizeof(SlotState), stream));
    active_d = (int32_t*)dmalloc(sizeof(int32_t));
    Q3_HIP_CHECK(hipHostMalloc((void**)&active_h, sizeof(int32_t)));

    // ---- paged KV caches (fp32).  Talker: 64-token pages handed out by kv_reserve / kv_release.  With the default pool (every slot can
    // reach max_ctx at once) slot b owns the fixed run [b*pps, (b+1)*pps): the table is the identity, never changes, and k_attn computes
    // the page ids instead of reading them (one memory round less in front of the K/V batch).  A bounded pool (kv_pool_tokens > 0) hands
    // pages out from a free list; page 0 is then a scratch page every unowned table entry points at, so masked rows of unarmed slots
    // write somewhere harmless.  Code predictor: 32 tokens per slot, rewritten every frame — one fixed page per slot.
    auto setup_stack = [&](DecStack& S, int Hh, int L, int nq, int nkv, int d, int ffn, float eps, int shift, int ctx, float theta, bool nt, int64_t pool_tokens, bool talker_stack) {
        S.H = Hh; S.L = L; S.nq = nq; S.nkv = nkv; S.d = d; S.ffn = ffn; S.eps = eps; S.page_shift = shift; S.nt = nt;
        const int ptok = 1 << shift;
        S.pages_per_slot = (ctx + ptok - 1) / ptok;
        size_t n_pages = (size_t)B * S.pages_per_slot;
        bool pooled = false;
        if (talker_stack) {
            kv.init(B, S.pages_per_slot, shift, pool_tokens > 0 ? (pool_tokens + ptok - 1) / ptok : 0);
            pooled = !kv.identity;
            n_pages = (size_t)kv.device_pages();
        }
        S.identity_pages = !pooled;
        S.kv_bf16 = talker_stack && (flags & Q3TTS_FLAG_KV_BF16);
        S.kv_round = talker_stack && !S.kv_bf16 && (flags & Q3TTS_FLAG_KV_ROUND_BF16);
        const size_t page_elems = (size_t)L * nkv * ptok * d, esz = S.kv_bf16 ? 2 : sizeof(float);
        S.kc = (float*)dmalloc(n_pages * page_elems * esz);
        S.vc = (float*)dmalloc(n_pages * page_elems * esz);
        std::vector<int> pt((size_t)B * S.pages_per_slot);
        for (size_t i = 0; i < pt.size(); ++i) pt[i] = talker_stack ? kv.table[i] : (int)i;
        S.page_table = (int*)dmalloc(pt.size() * sizeof(int));
        Q3_HIP_CHECK(hipMemcpy(S.page_table, pt.data(), pt.size() * sizeof(int), hipMemcpyHostToDevice));
        if (pooled) {   // the scratch page
            Q3_HIP_CHECK(hipMemsetAsync(S.kc, 0, page_elems * esz, stream));
            Q3_HIP_CHECK(hipMemsetAsync(S.vc, 0, page_elems * esz, stream));
        }
        // RoPE tables with the oracle's formula (fp32 libm): inv = 1/powf(theta, 2i/d); ang = pos*inv
        const int half = d / 2, npos = S.pages_per_slot * ptok;
        std::vector<float> cs((size_t)npos * half), sn((size_t)npos * half);
        for (int p = 0; p < npos; ++p)
            for (int i = 0; i < half; ++i) {
                const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)d);
                const float ang = (float)p * inv;
                cs[(size_t)p * half + i] = cosf(ang);
                sn[(size_t)p * half + i] = sinf(ang);
            }
        S.rope_cos = (float*)dmalloc(cs.size() * sizeof(float));
        S.rope_sin = (float*)dmalloc(sn.size() * sizeof(float));
        Q3_HIP_CHECK(hipMemcpy(S.rope_cos, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
        Q3_HIP_CHECK(hipMemcpy(S.rope_sin, sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice));
    };
    setup_stack(talker, H, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn, c.rms_eps, 6, max_ctx, c.rope_theta, true, kv_pool_tokens, true);
    setup_stack(cp, Hc, c.cp_layers, c.cp_heads, c.cp_kv_heads, c.cp_head_dim, c.cp_ffn, c.cp_rms_eps, 5, 32, c.cp_rope_theta, false, 0, false);
    // split-T attention: 128 cache tokens per workgroup (16 lane groups x 8 tokens in flight)
    // 128 cache tokens per workgroup; 64 for engines of one or two slots, where the attention launch is 72 workgroups at a 1041-token context
    // and each one's K/V ingest (128 KB) and softmax loop set its length: b=1 step 2.178 -> 2.156 ms.  Q3TTS_ATTN_CHUNK is the A/B knob.
    talker.chunk = B <= 2 ? 64 : 128;
    if (const char* ck = knob("Q3TTS_ATTN_CHUNK")) talker.chunk = atoi(ck) == 64 ? 64 : 128;
    talker.n_splits = (max_ctx + talker.chunk - 1) / talker.chunk;
    if (talker.n_splits > 64) { talker.n_splits = 64; talker.chunk = ((max_ctx + 63) / 64 + 127) / 128 * 128; }
    // k_attn_stream (batched step): splits of whole 64-token pages — 128 KB of K/V per split (256 tokens fp32, 512 bf16: measured against
    // 128 / 256 / 512 / 1024, profiles/r04_attn_stream_ab.txt) unless Q3TTS_ATTN_STREAM_CHUNK says otherwise
    talker.chunk_stream = (talker.kv_bf16 || talker.kv_round) ? 512 : 256;   // kv_round (test aid) mirrors the bf16 cache's splits
    if (const char* ck = knob("Q3TTS_ATTN_STREAM_CHUNK")) { const int v = atoi(ck); if (v >= 64 && v % 64 == 0) talker.chunk_stream = v; }
    talker.n_splits_stream = (max_ctx + talker.chunk_stream - 1) / talker.chunk_stream;
    for (DecStack* S : { &talker, &cp }) {
        const int ns = std::max(S->n_splits, S->n_splits_stream);
        S->po = fm((size_t)rows_max * S->nq * ns * S->d);
        S->pm = fm((size_t)rows_max * S->nq * ns);
        S->pl = fm((size_t)rows_max * S->nq * ns);
    }
    sync();
}

Engine::~Engine() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    if (proj_ids_d) (void)hipFree(proj_ids_d);
    if (proj_out_d) (void)hipFree(proj_out_d);
    codec_free();
    speaker_free();
    free_packed_weights();
    for (void* p : allocs) (void)hipFree(p);
    if (active_h) (void)hipHostFree(active_h);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (stream && !null_stream) (void)hipStreamDestroy(stream);
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
void Engine::set_tensor(const std::string& name, const float* data, int64_t n) {
    Tensor& t = T(name);
    if (t.numel != n) throw Error("tensor '" + name + "': expected " + std::to_string(t.numel) + " elements, got " + std::to_string(n));
    if (t.bf16) {
        std::vector<bf16_t> tmp((size_t)n);
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = f32_to_bf16(data[i]);
        Q3_HIP_CHECK(hipMemcpy(t.dev, tmp.data(), (size_t)n * 2, hipMemcpyHostToDevice));
    } else {
        Q3_HIP_CHECK(hipMemcpy(t.dev, data, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    finalized = false;
}

void Engine::get_tensor(const std::string& name, float* out, int64_t n) {
    Tensor& t = T(name);
    if (t.numel != n) throw Error("tensor '" + name + "': expected " + std::to_string(t.numel) + " elements, got " + std::to_string(n));
    sync();
    if (t.bf16) {
        std::vector<bf16_t> tmp((size_t)n);
        Q3_HIP_CHECK(hipMemcpy(tmp.data(), t.dev, (size_t)n * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) out[i] = bf16_to_f32(tmp[(size_t)i]);
    } else {
        Q3_HIP_CHECK(hipMemcpy(out, t.dev, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
}

// SURVEY.md section 8d synthetic weights: matrices N(0, 0.02^2) bf16-representable, norm gains 1,
// SnakeBeta alpha = beta = 0, LayerScale / ConvNeXt gamma 0.01, biases N(0, 0.02^2).
void Engine::fill_synthetic(uint64_t seed) {
    for (Tensor& t : tensors) {
        float mean = 0.f, sd = 0.f;
        switch (t.kind) {
        case TK_W: sd = t.synth_std; break;
        case TK_NORM: mean = 1.f; break;
        case TK_BIAS: sd = 0.02f; break;
        case TK_SCALE: mean = 0.01f; break;
        default: break;
        }
        launch_fill_synth(t.dev, t.bf16 ? 1 : 0, t.numel, mix64(seed ^ fnv1a(t.name)), mean, sd, stream);
    }
    sync();
    finalized = false;
}

// Fragment-packed copies of every matrix the matrix-core decode kernels stream (q3_common.h: launch_pack_mfma_b): + one copy of the
// bf16 projection weights (1.05 GB at 0.6B dims, 3 GB at 1.7B), registered under the row-major pointer the launch sites keep using.
void Engine::free_packed_weights() {
    for (auto& pr : packed_w) { unregister_packed_weight(pr.first); (void)hipFree(pr.second); }
    packed_w.clear();
}
void Engine::pack_mfma_weights() {
    free_packed_weights();
    auto pack = [&](const bf16_t* W, int N, int K) {
        if (W == nullptr || N < 1 || K < 32 || K % 32 != 0) return;
        for (auto& pr : packed_w) if (pr.first == W) return;
        bf16_t* P = nullptr;
        Q3_HIP_CHECK(hipMalloc((void**)&P, packed_mfma_b_elems(N, K) * sizeof(bf16_t)));
        launch_pack_mfma_b(W, P, N, K, stream);
        packed_w.emplace_back(W, P);
    };
    auto stack = [&](const DecStack& S) {
        for (const DecLayerW& w : S.layers) {
            pack(w.qkv, (S.nq + 2 * S.nkv) * S.d, S.H); pack(w.o, S.H, S.nq * S.d);
            pack(w.gate, S.ffn, S.H); pack(w.up, S.ffn, S.H); pack(w.down, S.H, S.ffn);
        }
    };
    stack(talker); stack(cp);
    pack(codec_head, c.vocab, c.hidden);
    for (const bf16_t* h : cp_head) pack(h, c.sub_vocab, cp_width());
    if (cp_proj_w) pack(cp_proj_w, cp_width(), c.hidden);
    pack(fc1_w, c.text_hidden, c.text_hidden); pack(fc2_w, c.hidden, c.text_hidden);
    sync();
    for (auto& pr : packed_w) register_packed_weight(pr.first, pr.second);
}

void Engine::finalize() {
    auto fp = [&](const std::string& n) { return (const float*)T(n).dev; };
    auto bp = [&](const std::string& n) { return (const bf16_t*)T(n).dev; };
    auto fill_stack = [&](DecStack& S, const std::string& prefix) {
        S.layers.resize(S.L);
        for (int i = 0; i < S.L; ++i) {
            const std::string p = prefix + ".layers." + std::to_string(i) + ".";
            DecLayerW& w = S.layers[i];
            w.in_norm = fp(p + "input_norm"); w.post_norm = fp(p + "post_norm");
            w.q_norm = fp(p + "q_norm"); w.k_norm = fp(p + "k_norm");
            w.qkv = bp(p + "q_proj"); w.o = bp(p + "o_proj");
            w.gate = bp(p + "gate_proj"); w.up = bp(p + "up_proj"); w.down = bp(p + "down_proj");
        }
    };
    fill_stack(talker, "talker");
    fill_stack(cp, "cp");
    talker_norm = fp("talker.norm"); codec_head = bp("talker.codec_head"); codec_embed_w = bp("talker.codec_embed");
    text_embed = bp("text.embed"); fc1_w = bp("text.fc1.w"); fc2_w = bp("text.fc2.w"); fc1_b = fp("text.fc1.b"); fc2_b = fp("text.fc2.b");
    cp_norm = fp("cp.norm");
    cp_proj_w = cp_projected() ? bp("cp.proj.w") : nullptr;
    cp_proj_b = cp_projected() ? fp("cp.proj.b") : nullptr;
    cp_head.clear(); cp_embed_w.clear();
    for (int j = 0; j < c.n_groups - 1; ++j) { cp_head.push_back(bp("cp.head." + std::to_string(j))); cp_embed_w.push_back(bp("cp.embed." + std::to_string(j))); }
    codec_finalize();
    speaker_finalize();
    pack_mfma_weights();
    finalized = true;
    // tts_pad_embed_ = text_project(TTS_PAD) (tts_onnx.cpp:459-463), model-wide constant kept on device
    if (TTS_PAD < c.text_vocab) {
        std::vector<float> pad(c.hidden);
        int64_t id = TTS_PAD;
        text_project(&id, 1, pad.data());
        Q3_HIP_CHECK(hipMemcpy(tts_pad_d, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
}

// ------------------------------------------------------------------------------------------------
// decoder stack: five launches per layer
// ------------------------------------------------------------------------------------------------
static int pick_ksplit(int K) { // K slices per GEMM: 256 (two LDS chunks) per workgroup, 128 for small K
    int ks = K <= 512 ? K / 128 : K / 256;
    if (ks < 1) ks = 1;
    if (ks > 16) ks = 16;
    while (ks > 1 && K % (128 * ks) != 0) --ks;
    return ks;
}

// K slices of the gate/up GEMM under the seam: 4 (256-wide at K = 1024).  Q3TTS_SEAM_GU_KS=8 is the A/B knob for 128-wide slices (twice the
// workgroups, half the body each, 8 slab pairs per seam): measured slower, 4.90 vs 4.78 ms per b=64 step (profiles/r03_negative_results.txt).
static int seam_gu_ksplit(int K) {
    const int forced = knob("Q3TTS_SEAM_GU_KS") ? atoi(knob("Q3TTS_SEAM_GU_KS")) : 0;
    const int ks4 = std::min(4, pick_ksplit(K));
    if (forced == 2 && K == 1024) return 2;   // 512-wide slices (A/B knob; measured in round 5, profiles/r05_negative_results.txt)
    return forced == 8 && K % (128 * 8) == 0 && K / 8 == 128 ? 8 : ks4;
}


bool Engine::seam_applies(const DecStack& W, int M, float* x, int ldx, bool has_slot_map) const {
    const int AO = W.nq * W.d, NTH = W.H / 64;
    const bool mfma = M >= mfma_min_rows && W.H % 128 == 0 && AO % 128 == 0 && W.ffn % 128 == 0 && W.H <= 4096;
    if (!mfma || !seam_step || !seam_on || has_slot_map || M > 128 || NTH > 64 || NTH % 4 != 0 || W.L < 1) return false;
    const int ks_q = seam_gu_ksplit(W.H);
    GemmArgs t1, t2, t3;
    t1.seam = 1; t1.epi = EPI_SLAB; t1.M = M; t1.N = W.H; t1.K = AO; t1.ldo = W.H; t1.ldx = ldp; t1.sx = x; t1.sldx = ldx; t1.sgamma = W.layers[0].post_norm;
    t1.ssq_out = ssq_a_d; t1.ssq_nt = NTH; t1.seam_cnt = seam_cnt_d; t1.seam_gen = seam_gen_d; t1.oh = pl0h; t1.ol = pl0l; t1.ldp = ldp;
    t3 = t1; t3.K = W.ffn;
    t2.seam = 2; t2.epi = EPI_SLAB2; t2.M = M; t2.N = W.ffn; t2.K = W.H; t2.ldo = W.ffn; t2.ldx = ldp; t2.ssq_in = ssq_a_d; t2.ssq_in_nt = NTH;
    t2.seam_cnt = seam_cnt_d; t2.seam_gen = seam_gen_d; t2.oh = pl1h; t2.ol = pl1l; t2.ldp = ldp;
    return gemm_seam_ok(t1, pick_ksplit(AO)) && gemm_seam_ok(t2, ks_q) && gemm_seam_ok(t3, pick_ksplit(W.ffn));
}

bool Engine::run_layers(const DecStack& W, float* x, int ldx, int nb, int n_new, int slot_offset, const int* pos_dev, int pos_scalar,
                        const float* final_gamma, float final_eps, float* final_xn, int final_ld_xn, const int* slot_map) {
    const int M = nb * n_new, QKV = (W.nq + 2 * W.nkv) * W.d, AO = W.nq * W.d;
    // M >= mfma_min_rows: bf16-MFMA skinny GEMM over (hi, lo) activation planes; below it the single-pass GEMV family
    const bool mfma = M >= mfma_min_rows && W.H % 128 == 0 && AO % 128 == 0 && W.ffn % 128 == 0 && W.H <= 4096;
    const int ks_q = mfma ? std::min(4, pick_ksplit(W.H)) : 1;
    // Split-K seam (decode step only): the three finish launches of a layer pass fold into their GEMMs — o_proj and down reduce their
    // slabs in-launch and leave gamma * x planes plus per-tile sums of squares (the consumer applies 1 / rms: deferred RMSNorm), gate/up
    // reduces and applies SwiGLU.  The last layer's down projection keeps the finish launch (it also applies the stack's final norm).
    // Measured, b=64 step: 4.97 ms with the finish launches, 4.77 ms with this seam (flag line + static chunk owners); a first protocol with
    // an arrival ticket and a claim counter — two returning atomics on the critical path — took 5.57 ms (profiles/r03_negative_results.txt).
    const int NTH = W.H / 64;
    const bool seam = mfma && seam_applies(W, M, x, ldx, slot_map != nullptr);
    const bool planes_in = planes_in_ready;   // the sampler made planes0 + ssq_b (record_step checked seam_applies for this pass)
    planes_in_ready = false;
    if (planes_in && !seam) throw Error("run_layers: input planes announced without the seam");
    auto seam_counters = [&](int n_tiles) -> unsigned* {   // this launch's counter region
        const size_t need = (size_t)n_tiles * 2 * 16;
        if (seam_cnt_used + need > seam_cnt_words) throw Error("split-K seam: counter buffer too small for this step");
        unsigned* p = seam_cnt_d + seam_cnt_used;
        seam_cnt_used += need;
        return p;
    };
    if (mfma && !planes_in) // planes0 = RMSNorm(in_norm[0])(x)
        launch_finish(x, ldx, nullptr, 0, 0, 0, W.layers[0].in_norm, W.eps, M, W.H, pl0h, pl0l, ldp, nullptr, 0, stream);
    for (int l = 0; l < W.L; ++l) {
        const DecLayerW& w = W.layers[l];
        if (mfma) {
            GemmArgs g; // split-K slabs; the attention prologue sums them
            g.W = w.qkv; g.xh = pl0h; g.xl = pl0l; g.ldx = ldp; g.out = qkv_slab_d; g.ldo = QKV; g.M = M; g.N = QKV; g.K = W.H; g.epi = EPI_SLAB; g.nt = W.nt;
            launch_gemm2(g, ks_q, 4, stream);
        } else {
            GemvArgs g;
            g.W = w.qkv; g.x = x; g.ldx = ldx; g.gamma = w.in_norm; g.eps = W.eps; g.out = qkv; g.ldo = QKV;
            g.M = M; g.N = QKV; g.K = W.H; g.epi = EPI_STORE; g.nt = W.nt;
            launch_gemv(g, stream);
        }
        if (!mfma && nb == 1 && pos_dev == nullptr && slot_map == nullptr && W.n_splits == 1 && W.pages_per_slot == 1 && !W.kv_bf16 && !W.kv_round &&
            !(flags & Q3TTS_FLAG_NO_FUSED_CP)) {
            // code predictor at b = 1: attention + o_proj + residual in one launch (identity page table: slot s owns page s).  The gate is
            // stack-agnostic — a talker with max_ctx <= 64 matches it too — and the fused kernel reads and writes an fp32 cache without
            // rounding, so the bf16 / rounded-bf16 cache modes (talker only) stay on the general path.
            CpAttnOprojArgs f;
            const int ptok = 1 << W.page_shift;
            const size_t coff = (((size_t)slot_offset * W.L + l) * W.nkv) * ptok * W.d;
            f.qkv = qkv; f.ld_qkv = QKV; f.kc = W.kc + coff; f.vc = W.vc + coff; f.page_tokens = ptok; f.base = pos_scalar;
            f.q_norm = w.q_norm; f.k_norm = w.k_norm; f.eps = W.eps; f.rope_cos = W.rope_cos; f.rope_sin = W.rope_sin;
            f.scale = 1.0f / sqrtf((float)W.d); f.nq = W.nq; f.nkv = W.nkv; f.d = W.d; f.W = w.o; f.K = AO; f.N = W.H; f.x = x; f.ldx = ldx;
            if (cp_attn_oproj_ok(f, n_new)) {
                launch_cp_attn_oproj(f, n_new, stream);
                GemvArgs fg;
                fg.W = w.gate; fg.W2 = w.up; fg.x = x; fg.ldx = ldx; fg.gamma = w.post_norm; fg.eps = W.eps; fg.out = act; fg.ldo = W.ffn;
                fg.M = M; fg.N = W.ffn; fg.K = W.H; fg.epi = EPI_SWIGLU; fg.nt = W.nt;
                launch_gemv(fg, stream);
                GemvArgs dg;
                dg.W = w.down; dg.x = act; dg.ldx = W.ffn; dg.res = x; dg.ldres = ldx; dg.out = x; dg.ldo = ldx;
                dg.M = M; dg.N = W.H; dg.K = W.ffn; dg.epi = EPI_RESIDUAL; dg.nt = W.nt;
                launch_gemv(dg, stream);
                continue;
            }
        }
        AttnArgs a;
        a.qkv = qkv; a.ld_qkv = QKV; a.out = attn; a.ld_out = AO; a.kcache = W.kc; a.vcache = W.vc; a.kv_bf16 = W.kv_bf16; a.kv_round = W.kv_round;
        if (mfma) { a.qkv = qkv_slab_d; a.qkv_nslab = ks_q; a.qkv_slab_stride = (size_t)M * QKV; }
        if (seam && (l > 0 || planes_in)) { a.ssq_in = ssq_b_d; a.ssq_nt = NTH; a.ssq_K = W.H; a.ssq_eps = W.eps; }   // planes0 came from the previous layer's down seam (or the sampler): gamma * x, 1 / rms deferred
        a.page_table = W.page_table; a.pages_per_slot = W.pages_per_slot; a.page_shift = W.page_shift; a.identity_pages = W.identity_pages;
        a.layer = l; a.n_layers = W.L; a.q_norm = w.q_norm; a.k_norm = w.k_norm; a.eps = W.eps;
        a.rope_cos = W.rope_cos; a.rope_sin = W.rope_sin; a.pos_dev = pos_dev; a.pos_scalar = pos_scalar;
        a.slot_offset = slot_offset; a.slot_map = slot_map; a.nb = nb; a.n_new = n_new; a.nq = W.nq; a.nkv = W.nkv; a.d = W.d;
        a.scale = 1.0f / sqrtf((float)W.d); a.window = 0; a.new_from_raw = 1;
        a.n_splits = W.n_splits; a.chunk = W.chunk; a.po = W.po; a.pm = W.pm; a.pl = W.pl;
        // Split-T exists to fill the chip at small batch.  With >= 256 (row, kv head) workgroups already and a context of a few hundred
        // tokens, one split walks the whole context (page ids are arithmetic with a fixed run of pages per slot) and writes the planes
        // itself: no partials, no combine launch (28 launches per b=64 step).
        if (mfma && W.n_splits > 1 && W.identity_pages && !attn_keep_splits && (size_t)nb * W.nkv >= 256 && (W.pages_per_slot << W.page_shift) <= 512) {
            a.n_splits = 1; a.chunk = 1 << 30;
        }
        // Long contexts in the batched step: the launch is bound by the KV bytes it streams — k_attn_stream (page-aligned splits walked
        // through a two-deep register ring, 3-4 workgroups per CU) instead of k_attn's one-batch 128-token splits.
        const int grp_w = W.nkv > 0 ? W.nq / W.nkv : 0;
        const bool stream_shape = attn_stream && mfma && n_new == 1 && W.d == 128 && slot_map == nullptr && W.page_shift == 6 && W.n_splits_stream > 0 &&
                                  W.nq % W.nkv == 0 && grp_w == 2;   // the kernel is built for two query heads per kv head (0.6B and 1.7B talkers)
        if (stream_shape && a.n_splits > 1) { a.stream = 1; a.n_splits = W.n_splits_stream; a.chunk = W.chunk_stream; }
        else if (stream_shape && attn_stream_one && a.n_splits == 1) a.stream = 1;
        const bool direct_planes = mfma && a.n_splits == 1;     // one split: the attention kernel normalises and writes the planes itself
        if (direct_planes) { a.out = nullptr; a.po = nullptr; a.pm = nullptr; a.pl = nullptr; a.oh = pl1h; a.ol = pl1l; a.ldp = ldp; }
        const bool direct_rows = !mfma && a.n_splits == 1;      // likewise for the GEMV family: normalised fp32 rows, nothing to combine
        if (direct_rows) { a.po = nullptr; a.pm = nullptr; a.pl = nullptr; }
        launch_attn(a, stream);
        if (mfma) {
            if (!direct_planes) {
                a.out = nullptr; a.oh = pl1h; a.ol = pl1l; a.ldp = ldp;
                launch_attn_combine(a, stream);                  // partials -> (hi, lo) planes
            }
            const int ks_o = pick_ksplit(AO), ks_d = pick_ksplit(W.ffn);
            GemmArgs o;
            o.W = w.o; o.xh = pl1h; o.xl = pl1l; o.ldx = ldp; o.out = slab_d; o.ldo = W.H; o.M = M; o.N = W.H; o.K = AO; o.epi = EPI_SLAB; o.nt = W.nt;
            if (seam) {   // x += sum(slabs); planes0 = gamma(post_norm) * x; ssq_a = per-tile sums of squares of x
                o.seam = 1; o.seam_gen = seam_gen_d; o.seam_spin = seam_spin; o.seam_cnt = seam_counters(W.H / 64); o.sx = x; o.sldx = ldx; o.sgamma = w.post_norm;
                o.oh = pl0h; o.ol = pl0l; o.ldp = ldp; o.ssq_out = ssq_a_d; o.ssq_nt = NTH;
            }
            launch_gemm2(o, ks_o, 4, stream);
            // x += sum(slabs); planes0 = RMSNorm(post_norm)(x)
            if (!seam) launch_finish(x, ldx, slab_d, ks_o, (size_t)M * W.H, W.H, w.post_norm, W.eps, M, W.H, pl0h, pl0l, ldp, nullptr, 0, stream);
            GemmArgs f; // gate and up as split-K slab pairs, SwiGLU applied by the finish kernel
            f.W = w.gate; f.W2 = w.up; f.xh = pl0h; f.xl = pl0l; f.ldx = ldp; f.out = gu_slab_d; f.out2 = gu_slab_d + (size_t)4 * rows_max * W.ffn; f.ldo = W.ffn;
            f.M = M; f.N = W.ffn; f.K = W.H; f.epi = EPI_SLAB2; f.nt = W.nt;
            const int ks_gu = seam ? seam_gu_ksplit(W.H) : ks_q;
            f.out2 = gu_slab_d + (size_t)ks_gu * rows_max * W.ffn;
            if (seam) {   // planes1 = SwiGLU of the slab sums scaled by 1 / rms(x) (from ssq_a)
                f.seam = 2; f.seam_gen = seam_gen_d; f.seam_spin = seam_spin; f.seam_cnt = seam_counters(W.ffn / 64); f.oh = pl1h; f.ol = pl1l; f.ldp = ldp;
                f.ssq_in = ssq_a_d; f.ssq_in_nt = NTH; f.seps = W.eps;
            }
            launch_gemm2(f, ks_gu, 4, stream);
            if (!seam) launch_finish_swiglu(gu_slab_d, gu_slab_d + (size_t)ks_gu * rows_max * W.ffn, ks_q, (size_t)M * W.ffn, M, W.ffn, pl1h, pl1l, ldp, stream);
            GemmArgs d;
            d.W = w.down; d.xh = pl1h; d.xl = pl1l; d.ldx = ldp; d.out = slab_d; d.ldo = W.H; d.M = M; d.N = W.H; d.K = W.ffn; d.epi = EPI_SLAB; d.nt = W.nt;
            if (seam && l + 1 < W.L) {   // x += sum(slabs); planes0 = gamma(next input norm) * x; ssq_b for the next layer's attention
                d.seam = 1; d.seam_gen = seam_gen_d; d.seam_spin = seam_spin; d.seam_cnt = seam_counters(W.H / 64); d.sx = x; d.sldx = ldx; d.sgamma = W.layers[l + 1].in_norm;
                d.oh = pl0h; d.ol = pl0l; d.ldp = ldp; d.ssq_out = ssq_b_d; d.ssq_nt = NTH;
                launch_gemm2(d, ks_d, 4, stream);
                continue;
            }
            launch_gemm2(d, ks_d, 4, stream);
            // x += sum(slabs); planes0 = RMSNorm(next layer's input norm | the stack's final norm)(x)
            const bool last = l + 1 == W.L;
            const float* ng = last ? final_gamma : W.layers[l + 1].in_norm;
            launch_finish(x, ldx, slab_d, ks_d, (size_t)M * W.H, W.H, ng, last ? final_eps : W.eps, M, W.H, pl0h, pl0l, ldp,
                          last ? final_xn : nullptr, last ? final_ld_xn : 0, stream);
            continue;
        }
        GemvArgs o;
        o.W = w.o; o.x = attn; o.ldx = AO; o.res = x; o.ldres = ldx; o.out = x; o.ldo = ldx;
        o.M = M; o.N = W.H; o.K = AO; o.epi = EPI_RESIDUAL; o.nt = W.nt;
        o.po = W.po; o.pm = W.pm; o.pl = W.pl; o.pS = W.n_splits; o.pchunk = W.chunk; o.pn_new = n_new; o.pslot_offset = slot_offset;
        o.pheads = W.nq; o.pd = W.d; o.ppos_dev = pos_dev; o.ppos_scalar = pos_scalar;
        if (direct_rows) { o.po = nullptr; o.pm = nullptr; o.pl = nullptr; }
        else if (!gemv_fast_path(o)) { // partials -> attn rows, then a GEMV without the combine prologue
            launch_attn_combine(a, stream);
            o.po = nullptr; o.pm = nullptr; o.pl = nullptr;
        }
        launch_gemv(o, stream);
        GemvArgs f;
        f.W = w.gate; f.W2 = w.up; f.x = x; f.ldx = ldx; f.gamma = w.post_norm; f.eps = W.eps; f.out = act; f.ldo = W.ffn;
        f.M = M; f.N = W.ffn; f.K = W.H; f.epi = EPI_SWIGLU; f.nt = W.nt;
        launch_gemv(f, stream);
        GemvArgs d;
        d.W = w.down; d.x = act; d.ldx = W.ffn; d.res = x; d.ldres = ldx; d.out = x; d.ldo = ldx;
        d.M = M; d.N = W.H; d.K = W.ffn; d.epi = EPI_RESIDUAL; d.nt = W.nt;
        launch_gemv(d, stream);
    }
    return mfma && final_gamma != nullptr;
}

// final RMSNorm + output head (codec_head / cp.head.j); optionally keeps the normalised rows
int Engine::head_proj(const bf16_t* Wm, const float* x, int ldx, const float* gamma, float eps, float* xn_out, int ld_xn,
                      float* out, int ldo, int M, int N, int K, bool nt, bool planes_ready, int plane_row0, int plane_row_stride, float* slab_out) {
    if (planes_ready || (M >= mfma_min_rows && K % 128 == 0 && K <= 4096)) {
        if (!planes_ready) {
            launch_finish(const_cast<float*>(x), ldx, nullptr, 0, 0, 0, gamma, eps, M, K, pl0h, pl0l, ldp, xn_out, ld_xn, stream);
            plane_row0 = 0; plane_row_stride = 1;
        }
        GemmArgs g;
        g.W = Wm; g.xh = pl0h + (size_t)plane_row0 * ldp; g.xl = pl0l + (size_t)plane_row0 * ldp; g.ldx = ldp * plane_row_stride;
        g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.epi = EPI_STORE; g.nt = nt;
        // a head of N columns is N / 64 workgroups walking all of K (32 for the predictor's 2048 columns: 14.8 us per launch at 64 rows);
        // with the consumer summing 4 K slices it is 4x the workgroups on a quarter of the bytes each
        if (slab_out != nullptr && M <= 128 && (K == 512 || K == 1024) && ldo % 4 == 0 && !knob("Q3TTS_NO_HEAD_SLABS")) {
            g.out = slab_out; g.epi = EPI_SLAB; g.slab_rows = M;
            launch_gemm2(g, 4, 4, stream);
            return 4;
        }
        launch_gemm2(g, 1, 4, stream);
        return 1;
    }
    GemvArgs g;
    g.W = Wm; g.x = x; g.ldx = ldx; g.gamma = gamma; g.eps = eps; g.xn_out = xn_out; g.ld_xn = ld_xn;
    g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.epi = EPI_STORE; g.nt = nt;
    launch_gemv(g, stream);
    return 1;
}

// ------------------------------------------------------------------------------------------------
// session-shaped ops
// ------------------------------------------------------------------------------------------------
// All rows in one go: ids uploaded once, 16-row blocks (the last one padded with a repeat of the last id) through gather + fc1 + fc2
// without a host round trip in between, one copy back.  Every block has the same shape, so a token's row does not depend on where in
// the list — or in which job — it was projected.
void Engine::text_project(const int64_t* ids, int n, float* out) {
    if (!finalized) throw Error("weights not finalized");
    if (n <= 0) return;
    const int TH = c.text_hidden, H = c.hidden;
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= c.text_vocab) throw Error("text id out of range");
    const int np = (n + 15) / 16 * 16;
    if (proj_cap < (size_t)np) {
        sync();
        if (proj_ids_d) (void)hipFree(proj_ids_d);
        if (proj_out_d) (void)hipFree(proj_out_d);
        proj_cap = (size_t)std::max(np, 256);
        Q3_HIP_CHECK(hipMalloc((void**)&proj_ids_d, proj_cap * sizeof(int64_t)));
        Q3_HIP_CHECK(hipMalloc((void**)&proj_out_d, proj_cap * H * sizeof(float)));
    }
    std::vector<int64_t> padded((size_t)np, ids[n - 1]);
    memcpy(padded.data(), ids, (size_t)n * sizeof(int64_t));
    Q3_HIP_CHECK(hipMemcpyAsync(proj_ids_d, padded.data(), (size_t)np * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    for (int i0 = 0; i0 < np; i0 += 16) {
        launch_gather_rows_bf16(text_embed, TH, proj_ids_d + i0, 16, text_tmp, TH, stream);
        GemvArgs a;
        a.W = fc1_w; a.x = text_tmp; a.ldx = TH; a.bias = fc1_b; a.out = text_tmp2; a.ldo = TH; a.M = 16; a.N = TH; a.K = TH; a.epi = EPI_BIAS_SILU;
        launch_gemv(a, stream);
        GemvArgs b;
        b.W = fc2_w; b.x = text_tmp2; b.ldx = TH; b.bias = fc2_b; b.out = proj_out_d + (size_t)i0 * H; b.ldo = H; b.M = 16; b.N = H; b.K = TH; b.epi = EPI_BIAS;
        launch_gemv(b, stream);
    }
    Q3_HIP_CHECK(hipMemcpyAsync(out, proj_out_d, (size_t)n * H * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();   // also keeps `padded` alive until the upload is done
}

void Engine::codec_embed(const int64_t* ids, int n, float* out) {
    if (!finalized) throw Error("weights not finalized");
    const int H = c.hidden;
    for (int i0 = 0; i0 < n; i0 += 16) {
        const int m = std::min(16, n - i0);
        for (int i = 0; i < m; ++i) if (ids[i0 + i] < 0 || ids[i0 + i] >= c.vocab) throw Error("codec id out of range");
        Q3_HIP_CHECK(hipMemcpyAsync(ids_d, ids + i0, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        launch_gather_rows_bf16(codec_embed_w, H, ids_d, m, xp, H, stream);
        Q3_HIP_CHECK(hipMemcpyAsync(out + (size_t)i0 * H, xp, (size_t)m * H * sizeof(float), hipMemcpyDeviceToHost, stream));
        sync();
    }
}

void Engine::cp_embed(int64_t id, int step, float* out) {
    if (!finalized) throw Error("weights not finalized");
    if (step < 0 || step >= c.n_groups - 1 || id < 0 || id >= c.sub_vocab) throw Error("cp_embed out of range");
    Q3_HIP_CHECK(hipMemcpyAsync(ids_d, &id, sizeof(int64_t), hipMemcpyHostToDevice, stream));
    launch_gather_rows_bf16(cp_embed_w[step], c.hidden, ids_d, 1, xp, c.hidden, stream);
    Q3_HIP_CHECK(hipMemcpyAsync(out, xp, (size_t)c.hidden * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
}

// run_prefill's device work for one slot whose S prompt rows sit in xp: layers, final norm + codec head on every row (logits_p [S][V],
// normalised rows hn [S][H]), the fused path armed with the last row, position = S.  Shared by the host and the device-pointer entry.
void Engine::prefill_rows_in_xp(int slot, int S) {
    const int H = c.hidden, V = c.vocab;
    const bool pr = run_layers(talker, xp, H, 1, S, slot, nullptr, 0, talker_norm, c.rms_eps, hn, H);
    // final norm + codec head on every row; normalised rows kept for last_hidden
    head_proj(codec_head, xp, H, talker_norm, c.rms_eps, hn, H, logits_p, V, S, V, H, true, pr);
    // arm the fused path: logits of the last row -> logits_t[slot], last_hidden -> x_cp[slot][0]
    launch_copy_rows(logits_p + (size_t)(S - 1) * V, V, logits_t + (size_t)slot * V, V, 1, V, stream);
    launch_copy_rows(hn + (size_t)(S - 1) * H, H, x_cp + (size_t)slot * 2 * H, H, 1, H, stream);
    st_h[slot].prompt_len = S;
    st_h[slot].n_frames = 0;
    int32_t pos = S;
    Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d + slot, &pos, sizeof(int32_t), hipMemcpyHostToDevice, stream));
}

void Engine::talker_prefill(int slot, const float* embeds, int S, float* logits, float* last_hidden) {
    if (!finalized) throw Error("weights not finalized");
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    if (S < 1 || S > 16) throw Error("prefill length must be 1..16 rows");
    const int H = c.hidden, V = c.vocab;
    kv_reserve(slot, S, false);
    Q3_HIP_CHECK(hipMemcpyAsync(xp, embeds, (size_t)S * H * sizeof(float), hipMemcpyHostToDevice, stream));
    prefill_rows_in_xp(slot, S);
    if (logits) Q3_HIP_CHECK(hipMemcpyAsync(logits, logits_p, (size_t)S * V * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (last_hidden) Q3_HIP_CHECK(hipMemcpyAsync(last_hidden, hn + (size_t)(S - 1) * H, (size_t)H * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
}

void Engine::talker_decode(int slot, const float* embed, float* logits, float* last_hidden) {
    if (!finalized) throw Error("weights not finalized");
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    const int H = c.hidden, V = c.vocab;
    int32_t pos = 0;
    Q3_HIP_CHECK(hipMemcpy(&pos, talker_pos_d + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (pos >= max_ctx) throw Error("KV cache full");
    kv_reserve(slot, pos + 1, false);
    Q3_HIP_CHECK(hipMemcpyAsync(xp, embed, (size_t)H * sizeof(float), hipMemcpyHostToDevice, stream));
    run_layers(talker, xp, H, 1, 1, slot, nullptr, pos);
    GemvArgs g;
    g.W = codec_head; g.x = xp; g.ldx = H; g.gamma = talker_norm; g.eps = c.rms_eps; g.xn_out = hn; g.ld_xn = H;
    g.out = logits_p; g.ldo = V; g.M = 1; g.N = V; g.K = H; g.epi = EPI_STORE; g.nt = true;
    launch_gemv(g, stream);
    launch_copy_rows(logits_p, V, logits_t + (size_t)slot * V, V, 1, V, stream);
    launch_copy_rows(hn, H, x_cp + (size_t)slot * 2 * H, H, 1, H, stream);
    pos += 1;
    Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d + slot, &pos, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (logits) Q3_HIP_CHECK(hipMemcpyAsync(logits, logits_p, (size_t)V * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (last_hidden) Q3_HIP_CHECK(hipMemcpyAsync(last_hidden, hn, (size_t)H * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
}

// ------------------------------------------------------------------------------------------------
// batch-first session ops on device pointers: row b of a call is slot b of the engine
// ------------------------------------------------------------------------------------------------
void Engine::stream_join(hipStream_t caller) {
    if (caller == stream) return;
    if (!ev_join) { Q3_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming)); Q3_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); }
    Q3_HIP_CHECK(hipEventRecord(ev_join, caller));
    Q3_HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
}
void Engine::stream_fork(hipStream_t caller) {
    if (caller == stream) return;
    if (!ev_fork) { Q3_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming)); Q3_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); }
    Q3_HIP_CHECK(hipEventRecord(ev_fork, stream));
    Q3_HIP_CHECK(hipStreamWaitEvent(caller, ev_fork, 0));
}
void Engine::dev_scratch(int nb) {
    if (nb < 1 || nb > B) throw Error("batch must be 1..max_batch (row b of a device-pointer call is slot b)");
    if (dev_logits_d) return;
    dev_logits_d = (float*)dmalloc((size_t)B * c.vocab * sizeof(float));
    dev_flags_d = (int*)dmalloc((size_t)B * sizeof(int));
    dev_pos_d = (int32_t*)dmalloc((size_t)B * sizeof(int32_t));
    dev_pos_dummy_d = (int32_t*)dmalloc((size_t)B * sizeof(int32_t));
    dev_st_d = (SlotState*)dmalloc((size_t)B * sizeof(SlotState));
    dev_codes_d = (int32_t*)dmalloc((size_t)B * c.n_groups * sizeof(int32_t));
}

// run_prefill (tts_onnx.cpp:615-665) for nb slots at once, rows in HBM: embeds [nb][S][H] (row block b holds lens[b] <= S prompt rows,
// lens == null: S for all) -> logits_last [nb][V] (the last prompt row's logits, all the reference consumes, :797-798) and
// last_hidden [nb][H].  Each slot's KV cache restarts at its prompt.  Consecutive slots with equal lengths share one pass through the
// layers (the scheduler's batched prefill, 128-row MFMA groups); a slot on its own takes exactly q3tts_talker_prefill_host's launches.
void Engine::talker_prefill_dev(const float* embeds, int nb, int S, const int32_t* lens, float* logits_last, float* last_hidden) {
    if (!finalized) throw Error("weights not finalized");
    dev_scratch(nb);
    if (S < 1 || S > 16) throw Error("prefill length must be 1..16 rows");
    if (!embeds) throw Error("talker_prefill_dev: null input");
    const int H = c.hidden, V = c.vocab;
    for (int b = 0; b < nb; ++b) if (lens && (lens[b] < 1 || lens[b] > S)) throw Error("talker_prefill_dev: lens[b] must be 1..S");
    auto len_of = [&](int b) { return lens ? (int)lens[b] : S; };
    const bool mfma_ok = H % 128 == 0 && (c.n_heads * c.head_dim) % 128 == 0 && c.ffn % 128 == 0 && H <= 4096;   // run_layers' MFMA condition
    std::vector<int32_t> pos_h;
    int b0 = 0;
    while (b0 < nb) {
        const int L = len_of(b0), cap = mfma_ok ? std::max(1, std::min(std::min(rows_max, 128) / L, 128)) : 1;
        int g = 1;
        while (b0 + g < nb && g < cap && len_of(b0 + g) == L) ++g;
        for (int k = 0; k < g; ++k) kv_reserve(b0 + k, L, false);
        if (g == 1 || g * L < mfma_min_rows) {
            for (int k = 0; k < g; ++k) {
                const int slot = b0 + k;
                Q3_HIP_CHECK(hipMemcpyAsync(xp, embeds + (size_t)slot * S * H, (size_t)L * H * sizeof(float), hipMemcpyDeviceToDevice, stream));
                prefill_rows_in_xp(slot, L);
                if (logits_last) launch_copy_rows(logits_p + (size_t)(L - 1) * V, V, logits_last + (size_t)slot * V, V, 1, V, stream);
                if (last_hidden) launch_copy_rows(hn + (size_t)(L - 1) * H, H, last_hidden + (size_t)slot * H, H, 1, H, stream);
            }
        } else {
            for (int k = 0; k < g; ++k)
                Q3_HIP_CHECK(hipMemcpyAsync(xp + (size_t)k * L * H, embeds + (size_t)(b0 + k) * S * H, (size_t)L * H * sizeof(float), hipMemcpyDeviceToDevice, stream));
            const bool pr = run_layers(talker, xp, H, g, L, b0, nullptr, 0, talker_norm, c.rms_eps, hn, H);
            if (!pr) throw Error("batched prefill expects the MFMA path");
            head_proj(codec_head, xp, H, talker_norm, c.rms_eps, nullptr, 0, logits_t + (size_t)b0 * V, V, g, V, H, true, true, L - 1, L);
            launch_copy_rows(hn + (size_t)(L - 1) * H, L * H, x_cp + (size_t)b0 * 2 * H, 2 * H, g, H, stream);
            if (logits_last) launch_copy_rows(logits_t + (size_t)b0 * V, V, logits_last + (size_t)b0 * V, V, g, V, stream);
            if (last_hidden) launch_copy_rows(hn + (size_t)(L - 1) * H, L * H, last_hidden + (size_t)b0 * H, H, g, H, stream);
            pos_h.assign((size_t)g, L);
            Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d + b0, pos_h.data(), (size_t)g * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            for (int k = 0; k < g; ++k) { st_h[b0 + k].prompt_len = L; st_h[b0 + k].n_frames = 0; }
            sync();   // pos_h is reused by the next group
        }
        b0 += g;
    }
    // every entry point returns after its launches and copies have completed (include/q3tts.h): the single-slot path above queues
    // asynchronous copies from host memory (a stack-local position in prefill_rows_in_xp, the page-table mirror row in kv_upload_row)
    sync();
}

// run_decode (tts_onnx.cpp:667-732) for nb slots in one pass: embeds [nb][H] -> logits [nb][V], last_hidden [nb][H]; every active row's
// token is appended to its slot's cache and the slot's position advances.  Rows with active[b] == 0 are carried through the launches
// (the batch keeps its shape) but leave no trace: outputs, cache contents that matter, position and the fused path's state of that slot
// are untouched.  The same launches as the fused step's talker stage at nb rows (GEMV family / k_gemv16 / MFMA slab GEMMs by row count).
void Engine::talker_decode_dev(const float* embeds, int nb, const uint8_t* active, float* logits, float* last_hidden) {
    if (!finalized) throw Error("weights not finalized");
    dev_scratch(nb);
    if (!embeds) throw Error("talker_decode_dev: null input");
    const int H = c.hidden, V = c.vocab;
    std::vector<int32_t> pos((size_t)nb), run((size_t)nb), flag((size_t)nb);
    Q3_HIP_CHECK(hipMemcpyAsync(pos.data(), talker_pos_d, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    sync();
    for (int b = 0; b < nb; ++b) {
        const bool on = !active || active[b] != 0;
        flag[(size_t)b] = on ? 1 : 0;
        if (on) {
            if (pos[(size_t)b] >= max_ctx) throw Error("KV cache full");
            kv_reserve(b, pos[(size_t)b] + 1, false);
        }
        // a masked row still writes one K / V row: at its own next position (overwritten by its next real token), or — full slot — over its last
        run[(size_t)b] = std::min(pos[(size_t)b], max_ctx - 1);
    }
    Q3_HIP_CHECK(hipMemcpyAsync(dev_pos_d, run.data(), (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    Q3_HIP_CHECK(hipMemcpyAsync(dev_flags_d, flag.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, stream));
    Q3_HIP_CHECK(hipMemcpyAsync(x_talk, embeds, (size_t)nb * H * sizeof(float), hipMemcpyDeviceToDevice, stream));
    const bool pr = run_layers(talker, x_talk, H, nb, 1, 0, dev_pos_d, 0, talker_norm, c.rms_eps, hn, H);
    head_proj(codec_head, x_talk, H, talker_norm, c.rms_eps, hn, H, dev_logits_d, V, nb, V, H, true, pr);
    // active rows: into the fused path's state (as q3tts_talker_decode_host leaves it) and into the caller's buffers
    launch_copy_rows_masked(dev_logits_d, V, logits_t, V, nb, V, dev_flags_d, stream);
    launch_copy_rows_masked(hn, H, x_cp, 2 * H, nb, H, dev_flags_d, stream);
    if (logits) launch_copy_rows_masked(dev_logits_d, V, logits, V, nb, V, dev_flags_d, stream);
    if (last_hidden) launch_copy_rows_masked(hn, H, last_hidden, H, nb, H, dev_flags_d, stream);
    for (int b = 0; b < nb; ++b) pos[(size_t)b] += flag[(size_t)b];
    Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d, pos.data(), (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    sync();   // the host arrays above are read by the copies
}

// predict_subcodes (tts_onnx.cpp:851-872) for nb utterances, fused: last_hidden [nb][H] (run_decode's output) and code0 [nb] (int64 ids, as the
// reference holds them) -> sub [nb][n_groups - 1] int32.  15 KV-cached predictor passes with on-device sampling (sample_token, :878-950):
// row b draws its j-th sub-code with q3tts_rng_uniform(seed, stream0 + b, frame, j + 1) — the fused generation loop's stream for
// utterance stream0 + b at that frame, so the sub-codes are those q3tts_decode_steps emits from the same state (its first pass takes its
// input planes from the code0 sampler with the RMSNorm deferred, this call normalises them itself: equal up to fp32 rounding).  Uses the
// per-slot workspaces of slots 0..nb-1 (predictor cache rows, predictor input rows): not for slots in the middle of a fused generation.
void Engine::code_predictor_dev(const float* last_hidden, const int64_t* code0, int nb, const q3tts_sampling& p, uint64_t seed, uint32_t stream0,
                                uint32_t frame, int32_t* sub) {
    if (!finalized) throw Error("weights not finalized");
    dev_scratch(nb);
    if (!last_hidden || !code0 || !sub) throw Error("code_predictor_dev: null argument");
    const int H = c.hidden, Hc = cp_width(), G = c.n_groups;
    std::vector<SlotState> st((size_t)nb);
    for (int b = 0; b < nb; ++b) {
        SlotState& s = st[(size_t)b];
        memset(&s, 0, sizeof s);
        s.n_frames = (int32_t)frame; s.finished = 0; s.active = 1; s.prompt_len = 0; s.trailing_len = 0; s.max_frames = (int32_t)frame + 1;
        s.top_k = p.top_k; s.ignore_eos = 1; s.temperature = p.temperature; s.top_p = p.top_p; s.stream_id = stream0 + (uint32_t)b; s.seed = seed;
    }
    Q3_HIP_CHECK(hipMemcpyAsync(dev_st_d, st.data(), (size_t)nb * sizeof(SlotState), hipMemcpyHostToDevice, stream));
    // rows [last_hidden, embed(code0)] per utterance and the frame's running embedding sum = embed(code0) (what the code0 sampler leaves)
    launch_copy_rows(last_hidden, H, x_cp, 2 * H, nb, H, stream);
    launch_gather_rows_bf16(codec_embed_w, H, code0, nb, x_cp + H, 2 * H, stream);
    launch_gather_rows_bf16(codec_embed_w, H, code0, nb, sum, H, stream);
    SampleArgs s0;
    s0.nb = nb; s0.sup_begin = c.suppress_begin; s0.sup_end = c.suppress_end; s0.eos_id = c.codec_eos;
    s0.group = 0; s0.n_groups = G; s0.st = dev_st_d; s0.embed = codec_embed_w; s0.H = H;
    s0.x_next = x_cp + H; s0.ld_xnext = 2 * H; s0.sum = sum; s0.x_talk = x_talk; s0.trailing = trailing_d; s0.max_trailing = max_trailing;
    s0.tts_pad = tts_pad_d; s0.talker_pos = dev_pos_dummy_d;
    // the sampler records group g of row b at codes[(b * cap + frame) * G + g]: cap = 1 and the base moved back by `frame` rows
    s0.codes = dev_codes_d - (size_t)frame * G; s0.max_frames_cap = 1;
    struct SeamScope { Engine& e; explicit SeamScope(Engine& en) : e(en) { e.seam_step = true; e.seam_cnt_used = 0; } ~SeamScope() { e.seam_step = false; } } seam_scope(*this);
    if (seam_gen_d) launch_bump_u32(seam_gen_d, stream);     // the generation this call's seam flags carry (the fused step's first sampler does this)
    const bool no_sp = knob("Q3TTS_NO_SAMPLER_PLANES") != nullptr;
    const bool sp_ok = !no_sp && !cp_projected() && H <= 2048 && H % 256 == 0;
    const bool spn = sp_ok && seam_applies(cp, nb, x_cp1, Hc, false);
    predictor_passes(nb, s0, false, spn, [] {});
    Q3_HIP_CHECK(hipMemcpy2DAsync(sub, (size_t)(G - 1) * sizeof(int32_t), dev_codes_d + 1, (size_t)G * sizeof(int32_t), (size_t)(G - 1) * sizeof(int32_t), (size_t)nb,
                                   hipMemcpyDeviceToDevice, stream));
    sync();   // `st` is read by the upload
}

// sample_token (tts_onnx.cpp:878-950) for nb rows: logits [nb][V] and u [nb] (uniforms in [0, 1), one per row) in HBM -> ids [nb] int64
void Engine::sample_dev(const float* logits, int nb, int V, const q3tts_sampling& p, const float* u, int suppress, int64_t* ids) {
    if (nb < 1 || nb > 65535) throw Error("sample_dev: batch out of range");
    if (V < 1 || V > 4096) throw Error("sample: n out of range");
    if (!logits || !u || !ids) throw Error("sample_dev: null argument");
    SampleArgs a;
    a.logits = logits; a.ld = V; a.V = V; a.nb = nb; a.sup_begin = c.suppress_begin; a.sup_end = c.suppress_end; a.eos_id = c.codec_eos;
    a.temperature = p.temperature; a.top_p = p.top_p; a.top_k = p.top_k; a.u_dev = u; a.suppress = suppress; a.token_out = ids;
    launch_sample(a, stream);
}

void Engine::code_predictor(const float* seq, int n, int step, float* logits) {
    if (!finalized) throw Error("weights not finalized");
    if (n < 1 || n > 16 || step < 0 || step >= c.n_groups - 1) throw Error("code_predictor arguments out of range");
    const int H = c.hidden, Hc = cp_width(), SV = c.sub_vocab;
    Q3_HIP_CHECK(hipMemcpyAsync(xp, seq, (size_t)n * H * sizeof(float), hipMemcpyHostToDevice, stream));
    float* xc = cp_project(xp, H, n);
    run_layers(cp, xc, Hc, 1, n, 0, nullptr, 0); // full causal re-run over the n rows (reference call pattern)
    GemvArgs g;
    g.W = cp_head[step]; g.x = xc + (size_t)(n - 1) * Hc; g.ldx = Hc; g.gamma = cp_norm; g.eps = c.cp_rms_eps;
    g.out = logits_p; g.ldo = SV; g.M = 1; g.N = SV; g.K = Hc; g.epi = EPI_STORE;
    launch_gemv(g, stream);
    Q3_HIP_CHECK(hipMemcpyAsync(logits, logits_p, (size_t)SV * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
}

void Engine::sample(const float* logits, int n, const q3tts_sampling& p, float u, int suppress, int64_t* tok) {
    if (n < 1 || n > 4096) throw Error("sample: n out of range");
    Q3_HIP_CHECK(hipMemcpyAsync(logits_p, logits, (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
    SampleArgs a;
    a.logits = logits_p; a.ld = n; a.V = n; a.nb = 1; a.sup_begin = c.suppress_begin; a.sup_end = c.suppress_end; a.eos_id = c.codec_eos;
    a.temperature = p.temperature; a.top_p = p.top_p; a.top_k = p.top_k; a.u = u; a.suppress = suppress; a.token_out = tok_d;
    launch_sample(a, stream);
    Q3_HIP_CHECK(hipMemcpyAsync(tok, tok_d, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    sync();
}

// build_prompt_embeddings, reference src/tts_onnx.cpp:442-539, for one utterance: the batch builder below with a batch of one
void Engine::build_prompt(const int64_t* ids, int n_ids, int lang, const float* speaker, float* prompt, int* S,
                          float* trailing, int cap_rows, int* n_trailing) {
    // the reference indexes input_ids[0..3] unguarded (:493, :518): 4 ids is the least it can take.  Empty text = the 5-token frame:
    // TTS_EOS lands in the "first text token" slot and the trailing block is just [tts_eos].
    if (n_ids < 4) throw Error("token sequence too short: need at least 4 ids (the reference indexes input_ids[3])");
    if (std::max(0, n_ids - 6) + 1 > cap_rows) throw Error("text too long for the trailing buffer");
    const int32_t offsets[2] = { 0, n_ids };
    const size_t toff = 0;
    const float* spk[1] = { speaker };
    build_prompts(ids, offsets, 1, lang, speaker ? spk : nullptr, prompt, S, trailing, &toff, n_trailing);
}

// ------------------------------------------------------------------------------------------------
// fused generation: one frame = sampler + (n_groups-1) predictor passes + talker decode
// ------------------------------------------------------------------------------------------------
// build_prompt_embeddings for every utterance of a job: ONE projection pass over all the text ids the prompts need (the three tts
// specials first) and one codec-embedding gather (the control rows depend on the language only), then the reference's fp32 row sums on
// the host (tts_onnx.cpp:442-539, same order as build_prompt).  prompts: [n_utt][16][hidden]; trailing rows of utterance u start at row
// toff[u] of `trailing`.
void Engine::build_prompts(const int64_t* ids, const int32_t* offsets, int n_utt, int lang, const float* const* speakers,
                           float* prompts, int* S_out, float* trailing, const size_t* toff, int* nt_out) {
    const int H = c.hidden;
    std::vector<int64_t> all = { TTS_BOS, TTS_EOS, TTS_PAD };
    std::vector<size_t> first((size_t)n_utt);
    for (int u = 0; u < n_utt; ++u) {
        const int n_ids = offsets[u + 1] - offsets[u];
        if (n_ids < 4) throw Error("token sequence too short: need at least 4 ids (the reference indexes input_ids[3])");
        first[(size_t)u] = all.size();
        const int used = std::max(4, n_ids - 2);                  // ids[0..2] role, ids[3] first text, ids[4 .. n-2) trailing
        if (used - 4 + 1 > max_trailing) throw Error("text too long for the trailing buffer");
        all.insert(all.end(), ids + offsets[u], ids + offsets[u] + used);
    }
    std::vector<float> proj(all.size() * (size_t)H);
    text_project(all.data(), (int)all.size(), proj.data());
    const float *tts_bos = proj.data(), *tts_eos = proj.data() + H, *tts_pad = proj.data() + 2 * (size_t)H;
    std::vector<int64_t> cpf;
    if (lang == 0) cpf = { CODEC_NOTHINK, CODEC_THINK_BOS, CODEC_THINK_EOS };                      // :467-469
    else cpf = { CODEC_THINK, CODEC_THINK_BOS, LANG_ENGLISH + (lang - 1), CODEC_THINK_EOS };       // :470-474
    cpf.push_back(CODEC_PAD); cpf.push_back(CODEC_BOS);                                            // :475-476
    const int ncp = (int)cpf.size();
    std::vector<float> ce0((size_t)ncp * H), ce((size_t)(ncp + 1) * H);
    codec_embed(cpf.data(), ncp, ce0.data());                                                      // :478
    for (int u = 0; u < n_utt; ++u) {
        const int n_ids = offsets[u + 1] - offsets[u];
        const float* speaker = speakers ? speakers[u] : nullptr;
        const float* rows = proj.data() + first[(size_t)u] * H;   // projections of ids[0], ids[1], ...
        float* prompt = prompts + (size_t)u * 16 * H;
        memcpy(ce.data(), ce0.data(), ce0.size() * sizeof(float));
        if (speaker) { // speaker row inserted before the last (BOS) row, :481-490
            memmove(ce.data() + (size_t)ncp * H, ce.data() + (size_t)(ncp - 1) * H, (size_t)H * sizeof(float));
            memcpy(ce.data() + (size_t)(ncp - 1) * H, speaker, (size_t)H * sizeof(float));
        }
        memcpy(prompt, rows, (size_t)3 * H * sizeof(float));                                        // :493-494
        int row = 3;
        const int pad_count = ncp - 2 + (speaker ? 1 : 0);                                         // :497-498
        for (int i = 0; i <= pad_count; ++i, ++row) {                                              // :506-512
            const float* t = i < pad_count ? tts_pad : tts_bos;
            for (int j = 0; j < H; ++j) prompt[(size_t)row * H + j] = t[j] + ce[(size_t)i * H + j];
        }
        const float* ft = rows + (size_t)3 * H;                                                    // first text token, :518
        for (int j = 0; j < H; ++j) prompt[(size_t)row * H + j] = ft[j] + ce[(size_t)(pad_count + 1) * H + j]; // :519-520
        ++row;
        S_out[u] = row;
        int nt = (n_ids - 2) - 4;                                                                  // text_end - (text_start + 1), :531-534
        if (nt < 0) nt = 0;
        float* tr = trailing + toff[u] * H;
        if (nt > 0) memcpy(tr, rows + (size_t)4 * H, (size_t)nt * H * sizeof(float));
        memcpy(tr + (size_t)nt * H, tts_eos, (size_t)H * sizeof(float));                           // :535
        nt_out[u] = nt + 1;                                                                        // :536
    }
}

// predictor input rows: talker-width rows through cp.proj (bias) when the predictor is narrower, else the rows themselves
float* Engine::cp_project(float* rows, int ld, int M) {
    if (!cp_projected()) return rows;
    GemvArgs g;
    g.W = cp_proj_w; g.x = rows; g.ldx = ld; g.bias = cp_proj_b; g.out = x_cpp; g.ldo = cp_width(); g.M = M; g.N = cp_width(); g.K = c.hidden; g.epi = EPI_BIAS;
    launch_gemv(g, stream);
    return x_cpp;
}

// predict_subcodes (tts_onnx.cpp:851-872), KV-cached, for nb utterances: x_cp holds [last_hidden, embed(code0)] per utterance and `sum`
// embed(code0); 15 x {predictor pass, head j, sampler of group j + 1} — the sampler gathers the sampled code's embedding as the next
// pass's input and keeps the frame's running embedding sum.  sp0 / spn: the sampler in front of pass 0 / of a later pass also wrote that
// pass's input planes (split-K seam, deferred RMSNorm).  Shared by the fused step (record_step) and q3tts_code_predictor_dev.
void Engine::predictor_passes(int nb, const SampleArgs& s0, bool sp0, bool spn, const std::function<void()>& mark) {
    const int H = c.hidden, Hc = cp_width(), SV = c.sub_vocab, G = c.n_groups;
    auto with_planes = [&](SampleArgs& s, int mul, int add, const float* lh, int ld_lh) {
        s.pl_h = pl0h; s.pl_l = pl0l; s.pl_ldp = ldp; s.gamma0 = cp.layers[0].in_norm; s.ssq_out = ssq_b_d; s.ssq_nt = Hc / 64;
        s.pl_row_mul = mul; s.pl_row_add = add; s.lh = lh; s.ld_lh = ld_lh;
    };
    for (int j = 0; j < G - 1; ++j) {
        // pass 0: rows [last_hidden, embed(code0)] of every utterance; later passes: the embedding of the code just sampled
        float* xin = j == 0 ? cp_project(x_cp, H, nb * 2) : cp_project(x_cp1, H, nb);
        bool pr;
        planes_in_ready = j == 0 ? sp0 : spn;                   // the sampler in front of this pass made its input planes
        if (j == 0) pr = run_layers(cp, xin, Hc, nb, 2, 0, nullptr, 0, cp_norm, c.cp_rms_eps);
        else pr = run_layers(cp, xin, Hc, nb, 1, 0, nullptr, j + 1, cp_norm, c.cp_rms_eps);
        // head j on the last row of every utterance (pass 0 holds two rows per utterance: planes row b*2+1)
        // a traced step (step_logits) keeps plain logits rows: the head then takes the unsplit GEMM at any batch
        const int nsl = head_proj(cp_head[j], j == 0 ? xin + Hc : xin, j == 0 ? 2 * Hc : Hc, cp_norm, c.cp_rms_eps, nullptr, 0, logits_cp, SV, nb, SV, Hc, false,
                                  pr, j == 0 ? 1 : 0, j == 0 ? 2 : 1, trace_d ? nullptr : cp_logit_slab_d);
        mark();
        if (trace_d) launch_copy_rows(logits_cp + (size_t)trace_slot * SV, SV, trace_d + (size_t)(j + 1) * trace_cols, trace_cols, 1, SV, stream);
        SampleArgs s = s0;
        s.logits = nsl > 1 ? cp_logit_slab_d : logits_cp; s.nslab = nsl; s.slab_stride = (size_t)nb * SV;
        s.ld = SV; s.V = SV; s.group = j + 1; s.embed = cp_embed_w[j];
        s.x_next = j + 1 < G - 1 ? x_cp1 : nullptr; s.ld_xnext = H;
        s.pl_h = nullptr; s.lh = nullptr; s.step_gen = nullptr;
        if (spn && s.x_next) with_planes(s, 1, 0, nullptr, 0);
        launch_sample(s, stream);
        mark();
    }
}

void Engine::record_step(int nb) {
    const int H = c.hidden, Hc = cp_width(), V = c.vocab, SV = c.sub_vocab, G = c.n_groups;
    SampleArgs s0;
    s0.logits = logits_t; s0.ld = V; s0.V = V; s0.nb = nb; s0.sup_begin = c.suppress_begin; s0.sup_end = c.suppress_end; s0.eos_id = c.codec_eos;
    s0.group = 0; s0.n_groups = G; s0.st = st_d; s0.embed = codec_embed_w; s0.H = H;
    s0.x_next = x_cp + H; s0.ld_xnext = 2 * H; s0.sum = sum; s0.x_talk = x_talk; s0.trailing = trailing_d; s0.max_trailing = max_trailing;
    s0.tts_pad = tts_pad_d; s0.codes = codes_d; s0.max_frames_cap = max_frames_cap; s0.talker_pos = talker_pos_d;
    // stage_profile(): events between the stages of the step (eager launches only)
    size_t mk = 0;
    auto mark = [&]() { if (!stage_ev.empty()) Q3_HIP_CHECK(hipEventRecord(stage_ev[mk++], stream)); };
    // split-K seam flag lines of this step's GEMM launches: one region per launch; their words carry the step's generation (s0.step_gen:
    // the first sampler launch bumps it), so nothing has to be zeroed between steps
    struct SeamScope { Engine& e; explicit SeamScope(Engine& en) : e(en) { e.seam_step = true; e.seam_cnt_used = 0; } ~SeamScope() { e.seam_step = false; } } seam_scope(*this);
    s0.step_gen = seam_gen_d;
    mark();
    if (trace_d) launch_copy_rows(logits_t + (size_t)trace_slot * V, V, trace_d, trace_cols, 1, V, stream);
    // With the seam's deferred RMSNorm in place the sampler also writes the next predictor pass's input planes (gamma0 * row + the row's
    // sum of squares): the pass then starts at its QKV projection, without an RMSNorm launch in front (15 launches per step).
    const bool no_sp = knob("Q3TTS_NO_SAMPLER_PLANES") != nullptr;   // A/B knob
    const bool sp_ok = !no_sp && !cp_projected() && H <= 2048 && H % 256 == 0;
    const bool sp0 = sp_ok && seam_applies(cp, nb * 2, x_cp, Hc, false), spn = sp_ok && seam_applies(cp, nb, x_cp1, Hc, false);
    auto with_planes = [&](SampleArgs& s, int mul, int add, const float* lh, int ld_lh) {
        s.pl_h = pl0h; s.pl_l = pl0l; s.pl_ldp = ldp; s.gamma0 = cp.layers[0].in_norm; s.ssq_out = ssq_b_d; s.ssq_nt = Hc / 64;
        s.pl_row_mul = mul; s.pl_row_add = add; s.lh = lh; s.ld_lh = ld_lh;
    };
    if (sp0) with_planes(s0, 2, 1, x_cp, 2 * H);
    launch_sample(s0, stream);                                  // code0 (tts_onnx.cpp:803-812)
    mark();
    predictor_passes(nb, s0, sp0, spn, mark);                   // predict_subcodes (:851-872), KV-cached
    const bool pr = run_layers(talker, x_talk, H, nb, 1, 0, talker_pos_d, 0, talker_norm, c.rms_eps, x_cp, 2 * H);  // run_decode (:845)
    head_proj(codec_head, x_talk, H, talker_norm, c.rms_eps, x_cp, 2 * H, logits_t, V, nb, V, H, true, pr);
    mark();
}

// Per-stage device time of the decode step (north_star: "achieved fraction of roofline reported per stage"): n_steps eager steps of the
// armed slots with HIP events at the stage boundaries.  out[0] sampler (n_groups launches), out[1] code predictor (layer passes + heads),
// out[2] talker decode (layers + codec head), out[3] the whole step — milliseconds per step.  Eager launches carry a little more launch
// gap than the hipGraph replay bench.py times, so out[3] reads slightly above decode_ms_per_frame_step.
void Engine::stage_profile(int n_steps, double* out) {
    if (!finalized) throw Error("weights not finalized");
    const int nb = nb_in_use(), G = c.n_groups;
    if (nb == 0 || n_steps < 1) throw Error("stage_profile: no armed slot");
    const size_t n_marks = 2 + 2 * (size_t)(G - 1) + 1;   // before/after sample0, after each {predictor pass, its sample}, after the talker
    stage_ev.resize(n_marks);
    for (auto& e : stage_ev) Q3_HIP_CHECK(hipEventCreate(&e));
    double acc[3] = { 0.0, 0.0, 0.0 };
    try {
        for (int i = 0; i < n_steps; ++i) {
            record_step(nb);
            sync();
            for (size_t k = 0; k + 1 < n_marks; ++k) {
                float ms = 0.f;
                Q3_HIP_CHECK(hipEventElapsedTime(&ms, stage_ev[k], stage_ev[k + 1]));
                const int kind = k + 2 == n_marks ? 2 : (k % 2 == 0 ? 0 : 1);   // S, (P, S) x (G-1), T
                acc[kind] += ms;
            }
        }
    } catch (...) { for (auto& e : stage_ev) (void)hipEventDestroy(e); stage_ev.clear(); throw; }
    for (auto& e : stage_ev) (void)hipEventDestroy(e);
    stage_ev.clear();
    for (int k = 0; k < 3; ++k) out[k] = acc[k] / n_steps;
    out[3] = out[0] + out[1] + out[2];
}

// Device time of run_prefill (tts_onnx.cpp:615-665) for nb slots x S prompt rows: `reps` batched prefill passes over synthetic prompt rows
// that are already in HBM (the same launches slots_begin issues for a job's equal-length prompts: groups of up to 128 rows share one pass
// over the weights), HIP events on the engine's stream around each pass.  Slots 0..nb-1 must be free; they are released again afterwards.
void Engine::prefill_profile(int nb, int S, int reps, double* ms_per_pass) {
    if (!finalized) throw Error("weights not finalized");
    if (nb < 1 || nb > B || S < 1 || S > 16 || reps < 1) throw Error("prefill_profile: bad shape");
    for (int b = 0; b < nb; ++b) if (st_h[b].active) throw Error("prefill_profile: slots in use");
    float* emb = nullptr;
    const size_t n = (size_t)nb * S * c.hidden;
    Q3_HIP_CHECK(hipMalloc((void**)&emb, n * sizeof(float)));
    double acc = 0.0;
    try {
        Q3_HIP_CHECK(hipMemsetAsync(emb, 0x3C, n * sizeof(float), stream));   // 0x3C3C3C3C = 0.0115f in every element
        for (int r = -1; r < reps; ++r) {                                     // pass -1 warms the caches / lazy allocations, untimed
            Q3_HIP_CHECK(hipEventRecord(ev0, stream));
            talker_prefill_dev(emb, nb, S, nullptr, nullptr, nullptr);
            Q3_HIP_CHECK(hipEventRecord(ev1, stream));
            sync();
            float ms = 0.f;
            Q3_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            if (r >= 0) acc += ms;
            for (int b = 0; b < nb; ++b) slot_release(b);
        }
    } catch (...) { (void)hipFree(emb); for (int b = 0; b < nb; ++b) { try { slot_release(b); } catch (...) { } } throw; }
    (void)hipFree(emb);
    *ms_per_pass = acc / reps;
}

void Engine::step_logits(int slot, float* out, int cols) {
    if (!finalized) throw Error("weights not finalized");
    const int nb = nb_in_use(), G = c.n_groups;
    if (nb == 0 || slot < 0 || slot >= nb) throw Error("step_logits: slot not armed");
    if (cols < std::max(c.vocab, c.sub_vocab)) throw Error("step_logits: row buffer too narrow");
    float* buf = nullptr;
    Q3_HIP_CHECK(hipMalloc((void**)&buf, (size_t)G * cols * sizeof(float)));
    Q3_HIP_CHECK(hipMemsetAsync(buf, 0, (size_t)G * cols * sizeof(float), stream));
    trace_d = buf; trace_slot = slot; trace_cols = cols;
    try { record_step(nb); sync(); } catch (...) { trace_d = nullptr; (void)hipFree(buf); throw; }
    trace_d = nullptr;
    hipError_t e = hipMemcpy(out, buf, (size_t)G * cols * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    Q3_HIP_CHECK(e);
}

void Engine::kv_reserve(int slot, int tokens, bool exact) {
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    if (tokens < 0 || tokens > max_ctx) throw Error("KV reservation exceeds max_ctx");
    std::string err;
    const int rc = kv.reserve(slot, tokens, exact, &err);
    if (rc < 0) throw Error(err);
    if (rc > 0) kv_upload_row(slot);
}

// The slot is not in flight here (every entry point that steps it synchronises before returning); the mirror row outlives the copy.
void Engine::kv_upload_row(int slot) {
    if (kv.identity) return;
    Q3_HIP_CHECK(hipMemcpyAsync(talker.page_table + (size_t)slot * talker.pages_per_slot, kv.row(slot), (size_t)talker.pages_per_slot * sizeof(int), hipMemcpyHostToDevice, stream));
}

void Engine::kv_release(int slot) { kv_reserve(slot, 0, true); }

int Engine::nb_in_use() const {
    int nb = 0;
    for (int b = 0; b < B; ++b) if (st_h[b].active) nb = b + 1;
    return nb;
}

void Engine::slot_begin(int slot, const float* prompt, int S, const float* trailing, int n_trailing,
                        const q3tts_sampling& p, uint64_t seed, uint32_t stream_id, int ignore_eos) {
    SlotInit in;
    in.slot = slot; in.prompt = prompt; in.S = S; in.trailing = trailing; in.n_trailing = n_trailing; in.stream_id = stream_id;
    slots_begin(&in, 1, p, seed, ignore_eos);
}

// Prefill of several slots at once: slots with equal prompt length share one pass through the talker stack (rows = slots x S,
// groups of at most 128 rows on the MFMA path), only the last row of each prompt goes through the codec head.  One
// synchronisation for the whole set instead of two per slot.
void Engine::slots_begin(const SlotInit* in, int n, const q3tts_sampling& p, uint64_t seed, int ignore_eos) {
    if (!finalized) throw Error("weights not finalized");
    const int H = c.hidden, V = c.vocab;
    for (int i = 0; i < n; ++i) {
        if (in[i].slot < 0 || in[i].slot >= B) throw Error("slot out of range");
        if (in[i].n_trailing < 0 || in[i].n_trailing > max_trailing) throw Error("too many trailing text rows");
        if (in[i].S < 1 || in[i].S > 16) throw Error("prefill length must be 1..16 rows");
        if (p.max_new_tokens < 1 || in[i].S + p.max_new_tokens > max_ctx) throw Error("prompt + max_new_tokens exceeds max_ctx");
    }
    {   // KV pages for the prompt and every frame the slot may generate, all or nothing: nothing is armed if the pool cannot hold the set
        int need = 0;
        auto tokens_of = [&](const SlotInit& q) {
            const int all = q.S + (q.max_frames > 0 ? std::min(q.max_frames, p.max_new_tokens) : p.max_new_tokens);
            return q.kv_tokens > 0 ? std::min(all, std::max(q.kv_tokens, q.S)) : all;   // a caller that grows the share itself (the scheduler)
        };
        auto want_of = [&](const SlotInit& q) { return kv_pages_for(tokens_of(q)); };
        for (int i = 0; i < n; ++i) need += want_of(in[i]) - kv_slot_pages(in[i].slot);
        if (need > kv_free_pages()) {
            char msg[160];
            snprintf(msg, sizeof msg, "KV page pool exhausted: %d slots need %d more pages, %d of %d free", n, need, kv_free_pages(), kv_total_pages());
            throw Error(msg);
        }
        for (int pass = 0; pass < 2; ++pass)   // slots that give pages back first
            for (int i = 0; i < n; ++i)
                if ((want_of(in[i]) <= kv_slot_pages(in[i].slot)) == (pass == 0)) kv_reserve(in[i].slot, tokens_of(in[i]), true);
    }
    int i0 = 0;
    std::vector<int32_t> pos_h, map_h;
    while (i0 < n) {
        // a group: entries with the same S, at most rows_max / S of them (and <= 128 rows); consecutive slot ids write their results in
        // place, scattered ones (the scheduler re-arming whatever finished) go through a slot map and a scatter of the head's rows
        const bool mfma_ok = H % 128 == 0 && (c.n_heads * c.head_dim) % 128 == 0 && c.ffn % 128 == 0 && H <= 4096;   // run_layers' MFMA condition
        const int S = in[i0].S, cap = mfma_ok ? std::max(1, std::min(std::min(rows_max, 128) / S, 128)) : 1;
        int g = 1;
        bool consecutive = true;
        while (i0 + g < n && g < cap && in[i0 + g].S == S) { consecutive = consecutive && in[i0 + g].slot == in[i0].slot + g; ++g; }
        for (int k = 1; k < g; ++k)   // a slot may appear once per group (its cache rows are written by one row group only)
            for (int j = 0; j < k; ++j) if (in[i0 + k].slot == in[i0 + j].slot) throw Error("slots_begin: slot listed twice");
        if (g == 1 || g * S < mfma_min_rows) {
            for (int k = 0; k < g; ++k) talker_prefill(in[i0 + k].slot, in[i0 + k].prompt, S, nullptr, nullptr);
        } else {
            const int slot0 = in[i0].slot;
            for (int k = 0; k < g; ++k)
                Q3_HIP_CHECK(hipMemcpyAsync(xp + (size_t)k * S * H, in[i0 + k].prompt, (size_t)S * H * sizeof(float), hipMemcpyHostToDevice, stream));
            pos_h.assign((size_t)g, S);
            if (consecutive) {
                const bool pr = run_layers(talker, xp, H, g, S, slot0, nullptr, 0, talker_norm, c.rms_eps, hn, H);
                if (!pr) throw Error("batched prefill expects the MFMA path");   // M = g*S >= mfma_min_rows by construction
                // codec head on the last row of every prompt straight into the fused path's logits; normalised last rows -> predictor input
                head_proj(codec_head, xp, H, talker_norm, c.rms_eps, nullptr, 0, logits_t + (size_t)slot0 * V, V, g, V, H, true, true, S - 1, S);
                launch_copy_rows(hn + (size_t)(S - 1) * H, S * H, x_cp + (size_t)slot0 * 2 * H, 2 * H, g, H, stream);
                Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d + slot0, pos_h.data(), (size_t)g * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            } else {
                map_h.resize((size_t)g);
                for (int k = 0; k < g; ++k) map_h[(size_t)k] = in[i0 + k].slot;
                Q3_HIP_CHECK(hipMemcpyAsync(slot_map_d, map_h.data(), (size_t)g * sizeof(int), hipMemcpyHostToDevice, stream));
                const bool pr = run_layers(talker, xp, H, g, S, 0, nullptr, 0, talker_norm, c.rms_eps, hn, H, slot_map_d);
                if (!pr) throw Error("batched prefill expects the MFMA path");
                head_proj(codec_head, xp, H, talker_norm, c.rms_eps, nullptr, 0, logits_g, V, g, V, H, true, true, S - 1, S);
                for (int k = 0; k < g; ++k) {
                    const int sl = in[i0 + k].slot;
                    launch_copy_rows(logits_g + (size_t)k * V, V, logits_t + (size_t)sl * V, V, 1, V, stream);
                    launch_copy_rows(hn + (size_t)(k * S + S - 1) * H, H, x_cp + (size_t)sl * 2 * H, 2 * H, 1, H, stream);
                    Q3_HIP_CHECK(hipMemcpyAsync(talker_pos_d + sl, pos_h.data(), sizeof(int32_t), hipMemcpyHostToDevice, stream));
                }
            }
            sync();   // pos_h / map_h are reused by the next group
        }
        i0 += g;
    }
    for (int i = 0; i < n; ++i) {
        const SlotInit& q = in[i];
        if (q.n_trailing > 0)
            Q3_HIP_CHECK(hipMemcpyAsync(trailing_d + (size_t)q.slot * max_trailing * H, q.trailing, (size_t)q.n_trailing * H * sizeof(float), hipMemcpyHostToDevice, stream));
        slot_codec_stream_reset(q.slot);   // a new utterance: its streaming vocoder state starts over
        SlotState& s = st_h[q.slot];
        s.n_frames = 0; s.finished = 0; s.active = 1; s.prompt_len = q.S; s.trailing_len = q.n_trailing;
        s.max_frames = q.max_frames > 0 ? std::min(q.max_frames, p.max_new_tokens) : p.max_new_tokens;
        s.top_k = p.top_k; s.ignore_eos = ignore_eos; s.temperature = p.temperature; s.top_p = p.top_p; s.stream_id = q.stream_id; s.pad0 = 0; s.seed = seed;
        Q3_HIP_CHECK(hipMemcpyAsync(st_d + q.slot, &s, sizeof(SlotState), hipMemcpyHostToDevice, stream));
    }
    sync();
}

int Engine::decode_steps(int n_steps) {
    if (!finalized) throw Error("weights not finalized");
    int nb = nb_in_use();
    if (nb == 0) return 0;
    // One captured graph per batch width: past 16 rows widths are rounded up to a multiple of 8 (the extra slots are unarmed, their rows
    // masked like any finished slot's), so a queue that drains from 64 slots to 17 replays 7 graphs instead of capturing 48.
    if (nb > 16) nb = std::min(B, (nb + 7) / 8 * 8);
    hipGraphExec_t exec = nullptr;
    if (!(flags & Q3TTS_FLAG_NO_GRAPH)) {
        auto it = graphs.find(nb);
        if (it == graphs.end()) {
            hipGraph_t graph = nullptr;
            Q3_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            try { record_step(nb); } catch (...) { (void)hipStreamEndCapture(stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
            Q3_HIP_CHECK(hipStreamEndCapture(stream, &graph));
            Q3_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            Q3_HIP_CHECK(hipGraphDestroy(graph));
            graphs[nb] = exec;
        } else exec = it->second;
    }
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    for (int i = 0; i < n_steps; ++i) {
        if (exec) Q3_HIP_CHECK(hipGraphLaunch(exec, stream));
        else record_step(nb);
    }
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    launch_count_active(st_d, nb, active_d, stream);
    Q3_HIP_CHECK(hipMemcpyAsync(active_h, active_d, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_decode_ms, ev0, ev1));
    last_decode_steps = n_steps;
    total_decode_ms += last_decode_ms; total_decode_steps += n_steps;
    return *active_h;
}

// Measurement aid (q3tts_measure_skip_frames): every armed slot is moved n frames ahead WITHOUT generating them — frame counters and
// positions advance, the skipped frames' codes are zero, and the talker's whole KV cache is refilled with seeded synthetic rows (finite,
// |x| ~ 0.5), so what the slots decode afterwards is numerically meaningless but the step streams a context of the requested depth.  For
// profiling the decode step at a long context (rocprofv3 passes that cannot afford 1000 eager steps of warm-up).
void Engine::measure_skip_frames(int n) {
    if (!(flags & Q3TTS_FLAG_TEST_HOOKS)) throw Error("measure_skip_frames needs Q3TTS_FLAG_TEST_HOOKS");
    if (n < 1) throw Error("measure_skip_frames: n must be positive");
    const int nb = nb_in_use();
    if (nb == 0) throw Error("measure_skip_frames: no armed slot");
    sync();
    std::vector<SlotState> st;
    slots_state(nb, st);
    std::vector<int32_t> pos((size_t)nb, 0);
    Q3_HIP_CHECK(hipMemcpy(pos.data(), talker_pos_d, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) {
        if (!st[(size_t)b].active || st[(size_t)b].finished) continue;
        if (st[(size_t)b].n_frames + n >= st[(size_t)b].max_frames || pos[(size_t)b] + n >= max_ctx) throw Error("measure_skip_frames: past the slot's frame cap or max_ctx");
        kv_reserve(b, pos[(size_t)b] + n + 1, false);
        st[(size_t)b].n_frames += n; st_h[b].n_frames = st[(size_t)b].n_frames;
        pos[(size_t)b] += n;
    }
    Q3_HIP_CHECK(hipMemcpy(st_d, st.data(), (size_t)nb * sizeof(SlotState), hipMemcpyHostToDevice));
    Q3_HIP_CHECK(hipMemcpy(talker_pos_d, pos.data(), (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice));
    const size_t page_elems = (size_t)talker.L * talker.nkv * ((size_t)1 << talker.page_shift) * talker.d;
    const int64_t total = (int64_t)((size_t)kv.device_pages() * page_elems);
    launch_fill_synth(talker.kc, talker.kv_bf16 ? 1 : 0, total, 0x5EEDull, 0.f, 0.5f, stream);
    launch_fill_synth(talker.vc, talker.kv_bf16 ? 1 : 0, total, 0x5EEEull, 0.f, 0.5f, stream);
    sync();
}

void Engine::slots_state(int nb, std::vector<SlotState>& out) {   // one copy for the whole batch (the scheduler polls it between step chunks)
    out.resize((size_t)nb);
    if (nb > 0) Q3_HIP_CHECK(hipMemcpy(out.data(), st_d, (size_t)nb * sizeof(SlotState), hipMemcpyDeviceToHost));
}

void Engine::slot_status(int slot, int* n_frames, int* finished) {
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    SlotState s;
    Q3_HIP_CHECK(hipMemcpy(&s, st_d + slot, sizeof(SlotState), hipMemcpyDeviceToHost));
    if (n_frames) *n_frames = s.n_frames;
    if (finished) *finished = s.finished || s.n_frames >= s.max_frames;
}

void Engine::slot_codes(int slot, int64_t* codes, int cap_frames) {
    int nf = 0;
    slot_status(slot, &nf, nullptr);
    const int n = std::min(nf, cap_frames), G = c.n_groups;
    std::vector<int32_t> tmp((size_t)n * G);
    if (n > 0) Q3_HIP_CHECK(hipMemcpy(tmp.data(), codes_d + (size_t)slot * max_frames_cap * G, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) codes[i] = tmp[i];
}

void Engine::slot_logits(int slot, float* logits, float* last_hidden) {
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    sync();
    if (logits) Q3_HIP_CHECK(hipMemcpy(logits, logits_t + (size_t)slot * c.vocab, (size_t)c.vocab * sizeof(float), hipMemcpyDeviceToHost));
    if (last_hidden) Q3_HIP_CHECK(hipMemcpy(last_hidden, x_cp + (size_t)slot * 2 * c.hidden, (size_t)c.hidden * sizeof(float), hipMemcpyDeviceToHost));
}

void Engine::slot_release(int slot) {
    if (slot < 0 || slot >= B) throw Error("slot out of range");
    st_h[slot].active = 0;
    Q3_HIP_CHECK(hipMemcpy(st_d + slot, &st_h[slot], sizeof(SlotState), hipMemcpyHostToDevice));
    kv_release(slot);
    slot_codec_stream_reset(slot);
}

int64_t Engine::slot_codec_decode(int slot, float* pcm, int64_t cap) {
    int nf = 0;
    slot_status(slot, &nf, nullptr);
    if (nf <= 0) return 0; // reference returns an empty vector when no frame was generated (tts_onnx.cpp:418)
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n = codec_run(codes_d + (size_t)slot * max_frames_cap * c.n_groups, nf, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    const int64_t m = std::min(n, cap);
    if (m > 0 && pcm) Q3_HIP_CHECK(hipMemcpyAsync(pcm, pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += nf;
    return n;
}

// ------------------------------------------------------------------------------------------------
// streaming / chunked codec decode.  The decoder is causal, so the samples a chunk of frames [a, b) owns —
// full-utterance indices [L(a), L(b)), L(n) = q3tts_codec_decode_len(n), L(0) = 0 — are final once frame b-1 exists, and a
// decode of the window [a - left_context, b) reproduces them at offset L(a) - total_upsample * (a - left_context) (every
// layer is shift-invariant).  With left_context covering the history the result equals the whole-utterance decode the
// reference performs (tts_onnx.cpp:430) up to RoPE rounding; a shorter context trades exactness for bounded work
// (transformers' chunked_decode uses 25 frames — and also drops L(1)-many samples per chunk, which this does not).
// ------------------------------------------------------------------------------------------------
static int64_t codec_len_of(const q3tts_config& c, int n) { return n <= 0 ? 0 : q3tts_codec_decode_len(&c, n); }

int64_t Engine::codec_decode_range_dev(const int32_t* codes_dev, int a, int b, int left_context, float* pcm, int64_t cap) {
    if (!finalized) throw Error("weights not finalized");
    if (a < 0 || b <= a || b > max_frames_cap) throw Error("codec_decode_range: frame range out of range");
    if (left_context < 0) throw Error("codec_decode_range: negative left context");
    const int s = std::max(0, a - left_context);
    int64_t up = 1;
    for (int i = 0; i < c.cd_n_up; ++i) up *= c.cd_up_ratios[i];
    for (int i = 0; i < c.cd_n_blocks; ++i) up *= c.cd_up_rates[i];
    const int64_t first = codec_len_of(c, a) - up * s, n_own = codec_len_of(c, b) - codec_len_of(c, a);
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n_win = codec_run(codes_dev + (size_t)s * c.n_groups, b - s, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    if (first < 0 || first + n_own != n_win) throw Error("codec_decode_range: window arithmetic does not match the decoder length formula");
    const int64_t m = std::min(n_own, cap);
    if (m > 0 && pcm) Q3_HIP_CHECK(hipMemcpyAsync(pcm, pcm_d + first, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += b - s;
    return n_own;
}

int64_t Engine::slot_codec_decode_range(int slot, int a, int b, int left_context, float* pcm, int64_t cap) {
    int nf = 0;
    slot_status(slot, &nf, nullptr);
    if (b > nf) throw Error("codec_decode_range: frames [" + std::to_string(a) + ", " + std::to_string(b) + ") are not generated yet (" + std::to_string(nf) + " so far)");
    // exact mode (the context covers the history): the slot's carried-state stream — O(new frames) per call instead of O(history)
    const bool no_carry = knob("Q3TTS_CODEC_NO_CARRY") != nullptr;   // A/B knob, read per call: the windowed decode of the whole history
    if (left_context >= a && a >= 0 && b > a && !no_carry) return slot_codec_stream_range(slot, a, b, pcm, cap);
    return codec_decode_range_dev(codes_d + (size_t)slot * max_frames_cap * c.n_groups, a, b, left_context, pcm, cap);
}

int64_t Engine::codec_decode_chunked_host(const int64_t* codes, int F, int chunk, int left_context, float* pcm, int64_t cap) {
    if (F < 1 || F > max_frames_cap) throw Error("codec_decode: F out of range");
    if (chunk < 1) throw Error("codec_decode_chunked: chunk must be positive");
    const int G = c.n_groups;
    std::vector<int32_t> tmp((size_t)F * G);
    for (size_t i = 0; i < tmp.size(); ++i) {
        if (codes[i] < 0 || codes[i] >= c.cd_codebook) throw Error("codec_decode: code out of range");
        tmp[i] = (int32_t)codes[i];
    }
    const bool no_carry = knob("Q3TTS_CODEC_NO_CARRY") != nullptr;
    if (left_context >= F && !no_carry) {   // exact mode: one carried-state stream, every chunk a push
        const int sid = codec_stream_begin(F);
        int64_t total = 0;
        try {
            for (int a = 0; a < F; a += chunk) {
                const int b = std::min(F, a + chunk);
                total += codec_stream_push_host(sid, codes + (size_t)a * G, b - a, pcm ? pcm + total : nullptr, std::max<int64_t>(0, cap - total));
            }
        } catch (...) { codec_stream_end(sid); throw; }
        codec_stream_end(sid);
        return total;
    }
    Q3_HIP_CHECK(hipMemcpyAsync(codes_scratch_d, tmp.data(), tmp.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    int64_t total = 0;
    for (int a = 0; a < F; a += chunk) {
        const int b = std::min(F, a + chunk);
        const int64_t n = codec_decode_range_dev(codes_scratch_d, a, b, left_context, pcm ? pcm + total : nullptr, std::max<int64_t>(0, cap - total));
        total += n;
    }
    return total;
}

int64_t Engine::codec_decode_host(const int64_t* codes, int F, float* pcm, int64_t cap) {
    if (!finalized) throw Error("weights not finalized");
    if (F < 1 || F > max_frames_cap) throw Error("codec_decode: F out of range");
    const int G = c.n_groups;
    std::vector<int32_t> tmp((size_t)F * G);
    for (size_t i = 0; i < tmp.size(); ++i) {
        if (codes[i] < 0 || codes[i] >= c.cd_codebook) throw Error("codec_decode: code out of range");
        tmp[i] = (int32_t)codes[i];
    }
    int32_t* cd = codes_scratch_d;
    Q3_HIP_CHECK(hipMemcpyAsync(cd, tmp.data(), tmp.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n = codec_run(cd, F, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    const int64_t m = std::min(n, cap);
    if (m > 0 && pcm) Q3_HIP_CHECK(hipMemcpyAsync(pcm, pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += F;
    return n;
}

// run_vocoder with everything already in HBM: codes int32 [F][n_groups] and the PCM destination are device pointers of the caller
// (e.g. torch tensors); out-of-range codes are clamped by the gather kernel (the host entry rejects them instead).
int64_t Engine::codec_decode_dev(const int32_t* codes_dev, int F, float* pcm_dev, int64_t cap) {
    if (!finalized) throw Error("weights not finalized");
    if (!codes_dev) throw Error("codec_decode_dev: null codes");
    if (F < 1 || F > max_frames_cap) throw Error("codec_decode: F out of range");
    float* pcm_d = nullptr;
    Q3_HIP_CHECK(hipEventRecord(ev0, stream));
    const int64_t n = codec_run(codes_dev, F, &pcm_d);
    Q3_HIP_CHECK(hipEventRecord(ev1, stream));
    const int64_t m = std::min(n, cap);
    if (m > 0 && pcm_dev) Q3_HIP_CHECK(hipMemcpyAsync(pcm_dev, pcm_d, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, stream));
    sync();
    Q3_HIP_CHECK(hipEventElapsedTime(&last_codec_ms, ev0, ev1));
    total_codec_ms += last_codec_ms; total_codec_frames += F;
    return n;
}

// Algorithmic bytes of one decode step: every talker weight once, every
// predictor weight once per pass, plus the KV entries the attention kernels read.
void Engine::step_bytes(double* wbytes, double* kvbytes) {
    const double H = c.hidden, Hc = cp_width();
    auto layer = [&](double Hw, int nq, int nkv, int d, int ffn) { return 2.0 * (Hw * (nq + 2.0 * nkv) * d + Hw * nq * d + 3.0 * Hw * ffn); };
    double w = c.n_layers * layer(H, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn) + 2.0 * H * c.vocab;
    const int P = c.n_groups - 1;
    w += P * (c.cp_layers * layer(Hc, c.cp_heads, c.cp_kv_heads, c.cp_head_dim, c.cp_ffn) + 2.0 * Hc * c.sub_vocab + (cp_projected() ? 2.0 * H * Hc : 0.0));
    double kv = 0;
    for (int b = 0; b < B; ++b) {
        if (!st_h[b].active) continue;
        int nf = 0;
        slot_status(b, &nf, nullptr);
        const double Tt = st_h[b].prompt_len + nf;
        kv += Tt * c.n_layers * 2.0 * c.n_kv_heads * c.head_dim * (talker.kv_bf16 ? 2.0 : 4.0);   // fp32 cache, bf16 under Q3TTS_FLAG_KV_BF16
        double tp = 0;
        for (int j = 0; j < P; ++j) tp += j + 2;
        kv += tp * c.cp_layers * 2.0 * c.cp_kv_heads * c.cp_head_dim * 4.0;
    }
    *wbytes = w; *kvbytes = kv;
}

} // namespace q3
